from .params import CameraParams
from .projection import project_to_image

__all__ = ["project_to_image", "CameraParams"]
