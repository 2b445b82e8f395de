"""The Jacobian entry points of RobotSdfCollisionNet (mlp_learn/sdf/robot_sdf.py:68-110, 117-158) on the device: omds_mlp_jacobian
through the C-ABI and compute_signed_distance_wgrad / _wgrad2 / dist_grad_closest through the reference-shaped class, against the
vectors the reference itself produced (tests/golden/wgrad_*.npz <- tools/make_golden_wgrad.py) and against the oracle for the
layouts no fixture holds (skip concatenations, 128-wide, a wide network)."""
import numpy as np
import pytest
import torch

from helpers import OWN, RTOL, assert_close, load, weights_path
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu

KINDS = ["franka", "planar7", "franka_tanh"]
LAYERS = {"franka": [256] * 4, "planar7": [256] * 4, "franka_tanh": [256] * 3}


def _facade(kind):
    from optimalmodulationds_amd import RobotSdfCollisionNet
    fx = load("wgrad_" + kind)
    m = orc.Mlp.from_npz(weights_path(kind))
    nn_model = RobotSdfCollisionNet(in_channels=fx["x"].shape[1], out_channels=m.W[-1].shape[0], layers=LAYERS[kind], skips=[])
    nn_model.load_weights(weights_path(kind), {})
    return nn_model, fx, m


@pytest.mark.parametrize("kind", KINDS)
def test_compute_signed_distance_wgrad_like_the_reference(kind):
    nn_model, fx, m = _facade(kind)
    x = torch.from_numpy(fx["x"])
    safe = (fx["min_abs_preact"] > 1e-4) | (m.act != "relu")
    floor = float(np.abs(fx["all_grads"]).max())
    d, g, mi = nn_model.compute_signed_distance_wgrad(x.clone(), "all")
    assert g.shape == fx["all_grads"].shape
    assert_close(d.numpy(), fx["all_dist"], RTOL, "distances", floor=OWN)
    assert_close(g.numpy()[safe], fx["all_grads"][safe], 2e-5, "all Jacobian columns", floor=floor)
    d, g, mi = nn_model.compute_signed_distance_wgrad(x.clone(), [int(c) for c in fx["cols"]])
    assert_close(g.numpy()[safe], fx["cols_grads"][safe], 2e-5, "listed Jacobian columns", floor=floor)
    d, g, mi = nn_model.compute_signed_distance_wgrad(x.clone(), "closest")
    assert g.shape == fx["closest_grads"].shape and mi.dtype == torch.int64
    assert (mi.numpy() == fx["closest_idx"]).all()
    assert_close(d.numpy(), fx["closest_dist"], RTOL, "closest: distances", floor=OWN)
    assert_close(g.numpy()[safe], fx["closest_grads"][safe], 2e-5, "closest gradient", floor=floor)
    if "w2_grads" in fx:
        d, g2, mi = nn_model.compute_signed_distance_wgrad2(x.clone())
        assert g2.shape == fx["w2_grads"].shape and (mi.numpy() == fx["w2_idx"]).all()
        assert_close(g2.numpy()[safe], fx["w2_grads"][safe], 2e-5, "wgrad2 gradient", floor=floor)
    nb = fx["dgc_dist"].shape[0]
    nn_model.allocate_gradients(nb, {})
    d, g, mi = nn_model.dist_grad_closest(x.clone())
    assert d.shape == fx["dgc_dist"].shape and g.shape == fx["dgc_grads"].shape      # truncated to maxInputSize rows
    assert (mi.numpy() == fx["dgc_idx"]).all()
    assert_close(g.numpy()[safe[:nb]], fx["dgc_grads"][safe[:nb]], 2e-5, "dist_grad_closest gradient", floor=floor)
    # the caller's link order: columns, arg-min and Jacobian columns are those of the re-ordered outputs
    nn_model.set_link_order([int(c) for c in fx["order"]])
    d, g, mi = nn_model.compute_signed_distance_wgrad(x.clone(), "closest")
    assert_close(d.numpy(), fx["ord_closest_dist"], RTOL, "re-ordered distances", floor=OWN)
    assert (mi.numpy() == fx["ord_closest_idx"]).all()
    assert_close(g.numpy()[safe], fx["ord_closest_grads"][safe], 2e-5, "closest gradient, re-ordered", floor=floor)
    d, g, mi = nn_model.compute_signed_distance_wgrad(x.clone(), [int(c) for c in fx["cols"]])
    assert_close(g.numpy()[safe], fx["ord_cols_grads"][safe], 2e-5, "listed columns, re-ordered", floor=floor)
    assert_close(nn_model.compute_signed_distance(x).numpy(), fx["ord_closest_dist"], RTOL, "compute_signed_distance", floor=OWN)


def _wide():
    rng = np.random.RandomState(5)
    dims = [30, 384, 320, 9]
    W = [rng.uniform(-1, 1, (dims[i + 1], dims[i])).astype(np.float32) / np.sqrt(dims[i]) for i in range(3)]
    b = [rng.uniform(-1, 1, dims[i + 1]).astype(np.float32) / np.sqrt(dims[i]) for i in range(3)]
    W[-1] *= 40.0
    return orc.Mlp(W, b, "relu")


@pytest.mark.parametrize("kind", ["franka_skip", "planar7_128", "planar2", "wide"])
def test_jacobian_columns_through_the_c_abi(kind):
    """omds_mlp_jacobian vs the oracle on the layouts without a reference vector; the arg-min column of the Jacobian is the
    gradient omds_mlp_forward_vjp returns, bit for bit (same kernel, another seed)."""
    from optimalmodulationds_amd.engine import Engine
    m = _wide() if kind == "wide" else orc.Mlp.from_npz(weights_path(kind))
    d = m.W[0].shape[1] // 3
    C = m.W[-1].shape[0]
    rng = np.random.RandomState(3)
    B = 150
    x = rng.uniform(-2.5, 2.5, (B, d)).astype(np.float32)
    x[:, -3:] = rng.uniform(-0.2, 1.0, (B, 3))
    eng = Engine(d - 3, 64, 1, 3, max_obs=8)                       # capacity n_traj * n_closest = 192 rows
    eng.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
    cols = list(range(C))
    y, J = eng.mlp_jacobian(x, cols)
    oy, oJ = orc.mlp_jacobian(m, x, cols)
    assert_close(y, oy, RTOL, "raw forward", floor=OWN)
    margin = orc.relu_margin(m, x) if m.act == "relu" else np.full(B, 1.0)
    ok = margin >= 5e-6
    assert ok.mean() > 0.9
    assert_close(J[ok], oJ[ok], 2e-5, "Jacobian", floor=float(np.abs(oJ).max()))
    y2, g, mi = eng.mlp_forward_vjp(x)
    assert np.array_equal(y, y2)
    assert np.array_equal(J[np.arange(B), :, mi], g)
    _, J1 = eng.mlp_jacobian(x, [C - 1])                            # a single column == that column of the full call
    assert np.array_equal(J1[:, :, 0], J[:, :, C - 1])
    for bad in ([C], [-1], list(range(17))):
        with pytest.raises(Exception):
            eng.mlp_jacobian(x, bad)
    eng.close()
