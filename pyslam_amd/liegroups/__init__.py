"""liegroups-compatible SO(2)/SE(2)/SO(3)/SE(3) (numpy, fp64).

Drop-in for the surface of github.com/utiasSTARS/liegroups that pyslam's hot
path, examples and tests use.
"""
from .so2 import SO2
from .so3 import SO3
from .se import SE2, SE3

__all__ = ["SO2", "SE2", "SO3", "SE3"]
