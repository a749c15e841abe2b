"""The depth attacks of the reference's patched ``torchattacks`` package that the training path uses
(torchattacks/__init__.py:6-8).  The stock classification attacks and the evaluation-only physical
variants are out of scope."""
from .attack import Attack
from .attacks.pgd_depth import PGD_depth
from .attacks.phy_obj_atk import Phy_obj_atk
from .attacks.phy_obj_atk_l0 import Phy_obj_atk_l0

__all__ = ["Attack", "PGD_depth", "Phy_obj_atk", "Phy_obj_atk_l0"]
