from pyslam_amd.residuals import *  # noqa: F401,F403
from pyslam_amd.residuals import __all__  # noqa: F401
