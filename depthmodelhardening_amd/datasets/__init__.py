from .synthetic import SyntheticKITTIDataset, kitti_like, make_object

__all__ = ["SyntheticKITTIDataset", "kitti_like", "make_object"]
