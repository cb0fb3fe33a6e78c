from pyslam_amd.sensors import *  # noqa: F401,F403
from pyslam_amd.sensors import __all__  # noqa: F401
