"""Dense photometric alignment: oracle and product class against goldens produced by
the reference's PhotometricResidualSE3 / Problem (oracle/gen_golden.py case_photometric), and the HIP path
(ps_photometric_*, through the C ABI) against the oracle."""
import os

import numpy as np
import pytest
import scipy.ndimage

from oracle import gn_oracle as orc
from oracle import photo_oracle as po
from pyslam_amd import synthetic
from pyslam_amd.liegroups import SE3, SO3
from pyslam_amd.losses import CauchyLoss, HuberLoss, L2Loss, TukeyLoss
from pyslam_amd.problem import Options, Problem
from pyslam_amd.residuals import PhotometricResidualSE3, QuadraticResidual
from pyslam_amd.sensors import RGBDCamera, StereoCamera
from pyslam_amd.utils import bilinear_interpolate

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'photometric.npz')
TAGS = ('stereo', 'rgbd')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


def camera_of(cam, rgbd):
    cu, cv, fu, fv, b, w, h = cam
    c = RGBDCamera(cu, cv, fu, fv, int(w), int(h)) if rgbd else StereoCamera(cu, cv, fu, fv, b, int(w), int(h))
    c.compute_pixel_grid()
    return c


def block_of(g, tag):
    return PhotometricResidualSE3(camera_of(g[tag + '_cam'], tag == 'rgbd'), g[tag + '_im_ref'], g[tag + '_depth_ref'],
                                  g[tag + '_im_track'], g[tag + '_im_jac'], float(g[tag + '_intensity_stiffness']),
                                  float(g[tag + '_depth_stiffness']), min_grad=float(g[tag + '_min_grad']))


def tables_of(g, tag):
    return po.tables(g[tag + '_cam'], g[tag + '_im_ref'], g[tag + '_depth_ref'], g[tag + '_im_jac'],
                     float(g[tag + '_min_grad']), tag == 'rgbd')


def dense_options():
    o = Options()                                   # reference pipelines/dense.py:31-36
    o.allow_nondecreasing_steps = True
    o.max_nondecreasing_steps = 5
    o.min_cost_decrease = 0.99
    o.max_iters = 30
    o.linesearch_max_iters = 0
    return o


# ---------------------------------------------------------------- CPU: oracle and host class vs the reference
@pytest.mark.parametrize('tag', TAGS)
def test_oracle_matches_the_reference_residual_and_jacobian(gold, tag):
    tb = tables_of(gold, tag)
    assert tb['pt_ref'].shape[0] == int(gold[tag + '_num_pixels'])
    np.testing.assert_allclose(tb['pt_ref'], gold[tag + '_pt_ref'], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(tb['tri_jac_d'], gold[tag + '_triang_jac'][:, :, 2], rtol=1e-13, atol=1e-13)
    for k in (0, 1):
        T = gold[tag + '_T%d' % k]
        r, J, _ = po.evaluate(tb, gold[tag + '_im_track'], 1.0, 0.25, T[:3, :3], T[:3, 3])
        assert r.shape == gold[tag + '_r%d' % k].shape          # the same pixels survive is_valid_measurement
        np.testing.assert_allclose(r, gold[tag + '_r%d' % k], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(J, gold[tag + '_J%d' % k], rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize('tag', TAGS)
def test_block_protocol_matches_the_reference(gold, tag):
    blk = block_of(gold, tag)
    for k in (0, 1):
        T = SE3.from_matrix(gold[tag + '_T%d' % k])
        r, J = blk.evaluate([T], [True])
        np.testing.assert_allclose(r, gold[tag + '_r%d' % k], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(J[0], gold[tag + '_J%d' % k], rtol=1e-11, atol=1e-11)
        r2, J2 = blk.evaluate([T.rot, T.trans], [True, True])
        np.testing.assert_array_equal(r, r2)
        np.testing.assert_allclose(J2[0], gold[tag + '_Jrot%d' % k], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(J2[1], gold[tag + '_Jtrans%d' % k], rtol=1e-11, atol=1e-11)
        assert blk.evaluate([T]).shape == r.shape and blk.evaluate([T], [False])[1] == [None]
        assert blk.evaluate([T.rot, T.trans], [False, True])[1][0] is None
    with pytest.raises(ValueError):
        blk.evaluate([1, 2, 3])


def test_bilinear_lookup():
    rng = np.random.default_rng(0)
    im = rng.uniform(0, 255, (13, 17))
    x, y = rng.uniform(0, 17, 500), rng.uniform(0, 13, 500)       # everything is_valid_measurement lets through
    want = scipy.ndimage.map_coordinates(im, [y, x], order=1, mode='nearest')
    np.testing.assert_allclose(bilinear_interpolate(im, x, y), want, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(po.bilinear(im, x, y), want, rtol=1e-12, atol=1e-12)
    # integer coordinates return the pixel; the last row / column repeat
    assert bilinear_interpolate(im, [3.], [4.]) == im[4, 3]
    assert bilinear_interpolate(im, [16.7], [12.2]) == pytest.approx(im[12, 16])
    # multi-channel images come back (N, channels) (reference utils.py:16-24)
    rgb = np.stack([im, 2 * im, 3 * im], axis=2)
    out = bilinear_interpolate(rgb, x[:5], y[:5])
    assert out.shape == (5, 3) and np.allclose(out[:, 2], 3 * out[:, 0])


def test_bilinear_lookup_is_the_reference_kernel_body():
    """tests/golden/bilinear.npz: outputs of the reference's own ``_bilinear_interpolate`` (pyslam/utils.py:27-75) with the
    four spellings that keep it from running repaired and nothing else (oracle/gen_golden.py: reference_bilinear_body; the
    repairs are recorded in the fixture) -- 320 coordinates incl. pixel centres, the last row / column and points up to
    2.5 pixels outside the image on every side.  Host function and oracle restate its arithmetic in its order: equal to
    the last bit."""
    from conftest import load_golden
    g = load_golden('bilinear')
    assert list(g['repairs']) == ['x = x[1] -> x = x[0]', 'y = y[1] -> y = y[0]', 'out[1] = -> out[0] =', 'np.int( -> int(']
    assert ((g['x'] < 0) | (g['x'] > g['im'].shape[1] - 1) | (g['y'] < 0) | (g['y'] > g['im'].shape[0] - 1)).sum() >= 60
    np.testing.assert_array_equal(bilinear_interpolate(g['im'], g['x'], g['y']), g['out'])
    np.testing.assert_array_equal(po.bilinear(g['im'], g['x'], g['y']), g['out'])


@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('form', ['se3', 'split'])
def test_oracle_gauss_newton_reproduces_the_reference_solve(gold, tag, form):
    tb, im = tables_of(gold, tag), gold[tag + '_im_track']
    hist = gold['%s_solve_%s_cost_history' % (tag, form)]
    R, t = np.eye(3), np.zeros(3)
    # linesearch_max_iters = 0: every iteration reports the cost of its linearisation point (problem.py:188-192)
    for k in range(1, len(hist)):
        dx, R, t, cost = po.gn_step(tb, im, 1.0, 0.25, R, t, 3, 10.0, split=form == 'split')
        assert cost == pytest.approx(hist[k], rel=1e-9)
    assert hist[0] == pytest.approx(hist[1], rel=1e-12)


# ---------------------------------------------------------------- GPU: the HIP path through the C ABI
def device_of(blk, loss, split=False):
    from pyslam_amd.device import PhotometricDevice
    return PhotometricDevice(blk, loss, split)


@pytest.mark.gpu
def test_device_image_lookup_against_the_reference_kernel_body():
    """photo_bilinear (csrc/ps_photo.h) through the C ABI, one pixel at a time: a one-pixel problem at the identity pose whose
    point projects to a golden coordinate, reference intensity 0, unit intensity variance, no depth variance, L2 loss has
    cost = I(u, v)^2 / 2.  Every coordinate of tests/golden/bilinear.npz that is a valid measurement (strictly inside the
    image: all the device ever looks up) must give the reference body's value; the others must be counted invalid."""
    from conftest import load_golden
    from pyslam_amd.device import PhotometricDevice
    g = load_golden('bilinear')
    im = np.ascontiguousarray(g['im'])
    h, w = im.shape
    cu, cv, fu, fv, b = 8.0, 6.0, 16.0, 16.0, 0.25

    class OnePixel:
        def __init__(self, x, y):
            z = 2.0                                             # disparity fu b / z = 2: a valid stereo measurement
            self.t = dict(pt_ref=np.array([[(x - cu) * z / fu, (y - cv) * z / fv, z]]), im_ref=np.zeros(1), im_jac=np.zeros((1, 2)),
                          tri_jac_d=np.zeros((1, 3)), im_track=im, cam=(cu, cv, fu, fv, b), cam_type=0, cam_w=w, cam_h=h,
                          intensity_covar=1.0, depth_covar=0.0)

        def device_tables(self):
            return self.t

    checked = rejected = 0
    for x, y, want in zip(g['x'], g['y'], g['out']):
        dev = PhotometricDevice(OnePixel(x, y), L2Loss(), False)
        dev.set_pose(np.eye(3), np.zeros(3))
        cost = dev.eval_cost()
        u, v = fu * ((x - cu) * 2.0 / fu) / 2.0 + cu, fv * ((y - cv) * 2.0 / fv) / 2.0 + cv     # what the device projects to
        if 0 < u < w and 0 < v < h:
            assert dev.num_valid == 1
            look = po.bilinear(im, np.array([u]), np.array([v]))[0]                  # (u, v) can differ from (x, y) by an ulp
            assert abs(look - want) <= 1e-11 * max(1., abs(want))
            assert abs(np.sqrt(2. * cost) - abs(want)) <= 1e-11 * max(1., abs(want)), (x, y)
            checked += 1
        else:
            assert dev.num_valid == 0 and cost == 0.
            rejected += 1
        dev.close()
    assert checked >= 240 and rejected >= 40


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('loss', [L2Loss(), HuberLoss(10.0), CauchyLoss(5.0), TukeyLoss(25.0)], ids=lambda l: type(l).__name__)
def test_normal_equations_match_the_oracle(gold, tag, loss):
    blk, tb = block_of(gold, tag), tables_of(gold, tag)
    dev = device_of(blk, loss)
    assert dev.num_pixels == tb['pt_ref'].shape[0]
    for k in (0, 1):
        T = gold[tag + '_T%d' % k]
        dev.set_pose(T[:3, :3], T[:3, 3])
        H, b, cost, nvalid = dev.normal_equations()
        Ho, bo, co, no = po.normal_equations(tb, gold[tag + '_im_track'], 1.0, 0.25, T[:3, :3], T[:3, 3],
                                             loss.LOSS_ID, getattr(loss, 'k', 1.0))
        assert nvalid == no
        assert np.linalg.norm(H - Ho) <= 1e-12 * np.linalg.norm(Ho)
        assert np.linalg.norm(b - bo) <= 1e-11 * np.linalg.norm(bo)
        assert cost == pytest.approx(co, rel=1e-12)
        assert dev.eval_cost() == pytest.approx(co, rel=1e-12) and dev.num_valid == no
        assert np.array_equal(H, H.T)
        H2 = dev.normal_equations()[0]
        assert np.array_equal(H, H2)                              # fixed reduction order: bitwise reproducible


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('split', [False, True])
def test_iteration_matches_the_oracle_step(gold, tag, split):
    blk, tb = block_of(gold, tag), tables_of(gold, tag)
    dev = device_of(blk, HuberLoss(10.0), split)
    T = gold[tag + '_T1']
    R, t = T[:3, :3].copy(), T[:3, 3].copy()
    dev.set_pose(R, t)
    for it in range(3):
        dxo, Ro, to, co = po.gn_step(tb, gold[tag + '_im_track'], 1.0, 0.25, R, t, 3, 10.0, split=split)
        dx, cost = dev.step(linesearch=False)
        assert np.linalg.norm(dx - dxo) <= 1e-9 * np.linalg.norm(dxo)
        assert cost == pytest.approx(co, rel=1e-12)
        R, t = dev.get_pose()
        assert np.abs(R - Ro).max() < 1e-11 and np.abs(t - to).max() < 1e-11
    # line-search mode reports the cost AFTER the step
    dev.set_pose(T[:3, :3], T[:3, 3])
    dx, cost = dev.step(linesearch=True)
    assert cost == pytest.approx(dev.eval_cost(), rel=1e-14)
    dev.snapshot(); dev.step(False); dev.restore()
    Rb, tb2 = dev.get_pose()
    dev.set_pose(T[:3, :3], T[:3, 3]); dev.step(True)
    assert np.array_equal(dev.get_pose()[0], Rb) and np.array_equal(dev.get_pose()[1], tb2)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('form', ['se3', 'split'])
def test_problem_solve_reproduces_the_reference(gold, tag, form):
    blk = block_of(gold, tag)
    prob = Problem(dense_options())
    if form == 'se3':
        prob.add_residual_block(blk, ['T_1_0'], loss=HuberLoss(10.0))
        prob.initialize_params({'T_1_0': SE3.identity()})
    else:
        prob.add_residual_block(blk, ['R_1_0', 't_1_0_1'], loss=HuberLoss(10.0))
        prob.initialize_params({'R_1_0': SO3.identity(), 't_1_0_1': np.zeros(3)})
    params = prob.solve()
    from pyslam_amd.device import PhotometricDevice
    assert isinstance(prob._device, PhotometricDevice)               # the HIP path ran, not the host block protocol
    hist = gold['%s_solve_%s_cost_history' % (tag, form)]
    assert len(prob._cost_history) == len(hist)
    np.testing.assert_allclose(prob._cost_history, hist, rtol=1e-9)
    T = params['T_1_0'] if form == 'se3' else SE3(params['R_1_0'], params['t_1_0_1'])
    np.testing.assert_allclose(T.as_matrix(), gold['%s_solve_%s_T' % (tag, form)], atol=1e-9)
    # the reference's own acceptance: close to the true motion (interpolation-limited)
    err = SE3.from_matrix(gold[tag + '_T_true']).dot(T.inv()).log()
    assert np.linalg.norm(err) < 1e-2
    assert prob.eval_cost() == pytest.approx(
        float(np.sum(HuberLoss(10.0).loss(blk.evaluate([T])))), rel=1e-10)
    # one more step through solve_one_iter leaves the parameters alone and agrees with the host protocol
    before = T.as_matrix().copy()
    dx, cost = prob.solve_one_iter()
    assert np.array_equal(T.as_matrix(), before) and dx.shape == (6,)
    prob.compute_covariance()
    key = 'T_1_0' if form == 'se3' else 't_1_0_1'
    blockcov = prob.get_covariance_block(key, key)
    r, J = blk.evaluate([T], [True])
    s = np.sqrt(HuberLoss(10.0).weight(r))
    cov = np.linalg.inv((J[0] * s[:, None]).T @ (J[0] * s[:, None]))
    want = cov if form == 'se3' else cov[:3, :3]
    np.testing.assert_allclose(blockcov, want, rtol=1e-8)


@pytest.mark.gpu
def test_full_resolution_alignment_converges_and_matches_the_oracle():
    """640 x 480 (307 200 pixels, the reference cameras' resolution): size-independent properties -- the normal
    equations agree with the oracle's on the same pose, Gauss-Newton from identity recovers the rendered motion."""
    sc = synthetic.photometric_scene(h=480, w=640, seed=9, xi_true=(0.02, -0.01, 0.03, 0.004, -0.006, 0.008), noise=0.2)
    cam = camera_of(sc['cam'], False)
    blk = PhotometricResidualSE3(cam, sc['im_ref'], sc['depth_ref'], sc['im_track'], sc['im_jac'], 1.0, 2.0, min_grad=0.02)
    dev = device_of(blk, HuberLoss(10.0))
    tb = po.tables(sc['cam'], sc['im_ref'], sc['depth_ref'], sc['im_jac'], 0.02, False)
    assert dev.num_pixels == tb['pt_ref'].shape[0] > 250000
    T0 = SE3.exp(np.array([0.01, 0.0, 0.01, 0.002, 0.0, 0.003]))
    dev.set_pose(T0.rot.as_matrix(), T0.trans)
    H, b, cost, nvalid = dev.normal_equations()
    Ho, bo, co, no = po.normal_equations(tb, sc['im_track'], 1.0, 0.25, T0.rot.as_matrix(), np.asarray(T0.trans), 3, 10.0)
    assert nvalid == no and cost == pytest.approx(co, rel=1e-12)
    assert np.linalg.norm(H - Ho) <= 1e-12 * np.linalg.norm(Ho) and np.linalg.norm(b - bo) <= 1e-10 * np.linalg.norm(bo)
    dev.set_pose(np.eye(3), np.zeros(3))
    for _ in range(10):
        dx, cost = dev.step(False)
    assert np.linalg.norm(dx) < 1e-6
    R, t = dev.get_pose()
    err = SE3.from_matrix(sc['T_true']).dot(SE3(SO3(R), t).inv()).log()
    assert np.linalg.norm(err) < 2e-3


@pytest.mark.gpu
def test_other_problem_shapes_keep_the_block_protocol(gold):
    """A photometric block next to another block, or on a constant parameter, is a generic problem: evaluated
    through evaluate() on the host, normal equations solved on the device (pyslam_amd/problem.py)."""
    blk = block_of(gold, 'stereo')
    prob = Problem(dense_options())
    prob.add_residual_block(blk, ['T'], loss=L2Loss())
    prob.add_residual_block(QuadraticResidual(1.0, 2.0, 1.0), ['a'])
    prob.initialize_params({'T': SE3.identity(), 'a': np.array([0.5, 0.1, 0.2])})
    assert prob._photometric_form() is None
    opt = dense_options(); opt.lm_lambda = 1e-3
    p2 = Problem(opt)
    p2.add_residual_block(blk, ['T'], loss=L2Loss())
    p2.initialize_params({'T': SE3.identity()})
    with pytest.raises(ValueError):
        p2.solve()
