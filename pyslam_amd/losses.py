"""Robust loss functions: element-wise rho(x) / psi(x) / w(x).

Mirrors the loss protocol of reference pyslam/losses.py:8-214 (``loss``,
``influence``, ``weight`` on an (m,) residual; ctor argument ``k``).  The
reference evaluates these through numba ufuncs; here they are plain numpy
expressions on the host, and the device restatement lives in
csrc/ps_math.h (ps_loss_rho / ps_loss_weight, selected by ``LOSS_ID`` during lowering;
only the six classes below THEMSELVES are lowered -- a subclass may override loss() /
weight(), so it takes the host-evaluated path: pyslam_amd/lowering.py:_loss_id_k).

Reference quirks kept on purpose:
* ``L2Loss.weight`` returns ``np.ones(x.size)`` (flat), losses.py:16-17;
* ``L1Loss`` returns NaN where |x| is close to zero, losses.py:26-33;
* ``HuberLoss.influence`` hands back the ufunc itself, losses.py:83-84.
"""
import numpy as np

# ids shared with the HIP side (csrc/ps_math.h)
LOSS_L2, LOSS_L1, LOSS_CAUCHY, LOSS_HUBER, LOSS_TUKEY, LOSS_TDIST = range(6)


def _arr(x):
    return np.asarray(x, dtype=float)


class L2Loss:
    LOSS_ID = LOSS_L2
    k = 0.

    def loss(self, x):
        return 0.5 * x * x

    def influence(self, x):
        return x

    def weight(self, x):
        return np.ones(np.size(x))


class L1Loss:
    LOSS_ID = LOSS_L1
    k = 0.

    def loss(self, x):
        return np.abs(x)

    def influence(self, x):
        x = _arr(x)
        out = np.sign(x)
        out[np.isclose(np.abs(x), 0.)] = np.nan
        return out

    def weight(self, x):
        x = _arr(x)
        with np.errstate(divide='ignore'):
            out = 1. / np.abs(x)
        out[np.isclose(np.abs(x), 0.)] = np.nan
        return out


def cauchy_rho(k, x):
    x = _arr(x)
    return (0.5 * k ** 2) * np.log(1. + (x / k) ** 2)


def cauchy_psi(k, x):
    x = _arr(x)
    return x / (1. + (x / k) ** 2)


def cauchy_w(k, x):
    x = _arr(x)
    return 1. / (1. + (x / k) ** 2)


class CauchyLoss:
    LOSS_ID = LOSS_CAUCHY

    def __init__(self, k):
        self.k = k

    def loss(self, x):
        return cauchy_rho(self.k, x)

    def influence(self, x):
        return cauchy_psi(self.k, x)

    def weight(self, x):
        return cauchy_w(self.k, x)


def huber_rho(k, x):
    x = _arr(x)
    a = np.abs(x)
    return np.where(a <= k, 0.5 * x * x, k * (a - 0.5 * k))


def huber_psi(k, x):
    x = _arr(x)
    return np.where(np.abs(x) <= k, x, k * np.sign(x))


def huber_w(k, x):
    x = _arr(x)
    a = np.abs(x)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(a <= k, 1., k / a)


class HuberLoss:
    LOSS_ID = LOSS_HUBER

    def __init__(self, k):
        self.k = k

    def loss(self, x):
        return huber_rho(self.k, x)

    def influence(self, x):
        # reference behaviour (losses.py:83-84): the function object, uncalled
        return huber_psi

    def weight(self, x):
        return huber_w(self.k, x)


def tukey_rho(k, x):
    x = _arr(x)
    c = k ** 2 / 6.
    return np.where(np.abs(x) <= k, c * (1. - (1. - (x / k) ** 2) ** 3), c)


def tukey_psi(k, x):
    x = _arr(x)
    return np.where(np.abs(x) <= k, x * (1. - (x / k) ** 2), 0.)


def tukey_w(k, x):
    x = _arr(x)
    return np.where(np.abs(x) <= k, 1. - (x / k) ** 2, 0.)


class TukeyLoss:
    LOSS_ID = LOSS_TUKEY

    def __init__(self, k):
        self.k = k

    def loss(self, x):
        return tukey_rho(self.k, x)

    def influence(self, x):
        return tukey_psi(self.k, x)

    def weight(self, x):
        return tukey_w(self.k, x)


def tdist_rho(k, x):
    x = _arr(x)
    return 0.5 * (k + 1.) * np.log(1. + x * x / k)


def tdist_psi(k, x):
    x = _arr(x)
    return (k + 1.) * x / (k + x * x)


def tdist_w(k, x):
    x = _arr(x)
    return (k + 1.) / (k + x * x)


class TDistributionLoss:
    LOSS_ID = LOSS_TDIST

    def __init__(self, k):
        self.k = k  # t-distribution degrees of freedom

    def loss(self, x):
        return tdist_rho(self.k, x)

    def influence(self, x):
        return tdist_psi(self.k, x)

    def weight(self, x):
        return tdist_w(self.k, x)
