from pyslam_amd.pipelines.ransac import *  # noqa: F401,F403
from pyslam_amd.pipelines.ransac import FrameToFrameRANSAC, compute_transform_fast, SE3_SHAPE  # noqa: F401
