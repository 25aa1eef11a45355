"""ImageFE (database side), drop-in for reference network/image_fe.py: adds the resnet50 branch
(:47-59, last_dim 512/1024/2048)."""
from ..network_mm.image_fe import ImageFE as _ImageFE


class ImageFE(_ImageFE):
    _ALLOWED = ("resnet18", "resnet34", "resnet50")
