from pyslam_amd.problem import Options, Problem  # noqa: F401
