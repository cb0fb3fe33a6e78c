"""Import shim: ``from liegroups import SE3`` resolves to pyslam_amd.liegroups.

The reference examples import the third-party ``liegroups`` package
(e.g. reference examples/stereo_ba.py:3); it is not installable here, so the
build ships its own implementation under this name.
"""
from pyslam_amd.liegroups import SO2, SE2, SO3, SE3

__all__ = ["SO2", "SE2", "SO3", "SE3"]
