from .depth_decoder import DepthDecoder
from .resnet_encoder import ResnetEncoder

__all__ = ["ResnetEncoder", "DepthDecoder"]
