"""``pyslam.pipelines``: only the frame-to-frame RANSAC has a device implementation (the cv2 / viso2
front-ends of the reference's pipelines are out of scope)."""
from pyslam_amd.pipelines.ransac import FrameToFrameRANSAC, compute_transform_fast  # noqa: F401
