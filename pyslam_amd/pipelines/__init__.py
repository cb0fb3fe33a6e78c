"""Pipeline steps with a device implementation.  Only the frame-to-frame RANSAC of the reference's
sparse VO pipeline (pyslam/pipelines/ransac.py) is in scope: it is the step that feeds the
motion-only solve; the cv2 / viso2 front-ends around it are not (out of scope)."""
from .ransac import FrameToFrameRANSAC, compute_transform_fast  # noqa: F401
