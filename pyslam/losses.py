from pyslam_amd.losses import (L2Loss, L1Loss, CauchyLoss, HuberLoss, TukeyLoss,  # noqa: F401
                               TDistributionLoss)
