"""Camera models (explicit re-exports; the reference uses a pkgutil walk,
pyslam/sensors/__init__.py:1-14)."""
from .stereo_camera import StereoCamera
from .rgbd_camera import RGBDCamera

__all__ = ["StereoCamera", "RGBDCamera"]
