from pyslam_amd.utils import invsqrt, stackmul  # noqa: F401
