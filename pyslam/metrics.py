"""``pyslam.metrics`` of the reference (pyslam/metrics.py), backed by pyslam_amd."""
from pyslam_amd.metrics import TrajectoryMetrics  # noqa: F401
