from pyslam_amd.utils import invsqrt, stackmul, bilinear_interpolate  # noqa: F401
