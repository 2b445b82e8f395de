"""GPU tests of skip-connection networks (MLPRegression(..., skips=[...]), network_macros_mod.py:113-146): the encoded input
concatenated behind a hidden layer, forward and backward, through the C-ABI (omds_set_mlp_ex) and the façade class.  The
fixtures mlp_franka_skip / franka_skip_shelf_K4 were captured from the reference's own class with seeded synthetic weights
(tools/make_golden.py --only-skip); the generic fixture tests of test_gpu_parity.py run the same scenario too."""
import numpy as np
import pytest

from helpers import RTOL, assert_close, load, weights_path
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu


def _random_skip_net(rng, layers, skips, act="relu", n=7, C=9):
    from optimalmodulationds_amd.robot_sdf import linear_plan
    shapes, _, skip_after = linear_plan(3 * (n + 3), C, layers, skips)
    W = [(rng.standard_normal((o, i)) * np.sqrt(2.0 / i)).astype(np.float32) for i, o in shapes]
    b = [(0.1 * rng.standard_normal(o)).astype(np.float32) for _, o in shapes]
    W[-1] *= 40.0
    b[-1] += 12.0
    return orc.Mlp(W, b, act, tuple(skip_after))


@pytest.mark.parametrize("layers,skips,act", [([256] * 4, [2], "relu"), ([256] * 4, [1], "relu"), ([256] * 4, [3], "relu"),
                                             ([256] * 5, [1, 3], "relu"), ([128] * 5, [2], "relu"), ([256] * 4, [2], "tanh")])
def test_forward_and_vjp_of_skip_layouts(layers, skips, act):
    """Every position a concatenation can take (behind layer 1, in the middle, in front of the last layer, two of them,
    a narrower network) against the oracle: raw outputs, arg-min link, input gradient."""
    from optimalmodulationds_amd.engine import Engine
    rng = np.random.RandomState(len(layers) * 10 + sum(skips))
    m = _random_skip_net(rng, layers, skips, act)
    x = rng.uniform(-2.5, 2.5, (96, 10)).astype(np.float32)
    x[:, 7:] = rng.uniform(-0.2, 1.0, (96, 3))
    eng = Engine(7, 128, 1, 1, max_obs=8)
    eng.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
    y, g, mi = eng.mlp_forward_vjp(x)
    yo, go, mio = orc.mlp_vjp_argmin(m, x)
    assert_close(y, yo, RTOL, "raw outputs")
    assert (mi == mio).all()
    safe = orc.relu_margin(m, x) > 1e-5
    assert safe.mean() > 0.8
    assert_close(g[safe], go[safe], 2e-5, "input gradient", floor=float(np.abs(go).max()))
    eng.close()


def test_facade_class_with_skips_matches_the_reference_vectors():
    """RobotSdfCollisionNet(in, out, skips=[2], layers=[256]*4) as the reference's drivers would build it."""
    from optimalmodulationds_amd import RobotSdfCollisionNet
    fx = load("mlp_franka_skip")
    net = RobotSdfCollisionNet(in_channels=10, out_channels=9, skips=[2], layers=[256] * 4)
    assert [w.shape for w in net.model.W] == [(256, 30), (226, 256), (256, 256), (9, 256)]
    net.load_weights(weights_path("franka_skip"), None)
    y = net.model_jit.forward(fx["x"]).numpy()
    assert_close(y, fx["y"], RTOL, "forward vs reference")
    y2, g, mi = net.functorch_vjp(fx["x"])
    assert (mi.numpy() == fx["min_idx"]).all()
    safe = fx["min_abs_preact"] > 1e-4
    assert_close(g.numpy()[safe], fx["grad"][safe], 2e-5, "vjp vs reference", floor=float(np.abs(fx["grad"]).max()))


def test_skip_network_rollouts_all_step_variants_agree():
    """The fused fp32 step, the step as stand-alone kernels, and the SCREENED step -- forced on, and chosen by the library itself
    at this size (k_screen's concatenation stage + k_exact's masks + k_tail_sel) -- give the same rollouts, the screened ones bit
    for bit; pass 1 (feature tables) and pass 2 (features recomputed) see the same network: the pass-1 value of the closest
    obstacle equals the pass-2 distance when no link is ignored."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd import _lib as L
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path("franka_skip"))
    obs, q0, qf = scenes.shelf_scene(), scenes.FRANKA_Q0, scenes.FRANKA_QF
    outs = []
    for flags, mode in ((0, 0), (L.FLAG_UNFUSED_STEP, 0), (0, 1), (0, 2)):
        e = Engine(7, 1024, 4, 5, max_obs=512, flags=flags)
        e.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
        e.set_obstacles(obs)
        e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0
        e.push_params()
        e.set_ds(qf)
        e.set_screening(mode)
        e.sample_policy(None, None, None, 0, 0, 0, 0, seed=3)
        e.propagate(q0)
        st = e.screen_stats()
        assert st["active"] == (mode != 0) and st["fallbacks"] == 0, st
        outs.append(e.get_rollouts())
        if flags == 0 and mode == 0:
            q = outs[0]["all_traj"][:64, 2]
            d, g, mind, idx = e.dist_grad(q, want_mindist=True, want_idx=True)
            do, go, mindo, idxo = orc.distance_repulsion_nn(m, q, obs, 5, [])
            assert_close(mind, mindo, RTOL, "pass-1 matrix vs oracle", floor=0.0)
            assert_close(d, do, 2e-5, "blended distance vs oracle", floor=0.0)
        e.close()
    for key in ("all_traj", "closest_dist_all", "dot_products", "qdot"):
        assert_close(outs[1][key], outs[0][key], 2e-4, "unfused vs fused " + key)
        assert np.array_equal(outs[2][key], outs[0][key]), key
        assert np.array_equal(outs[3][key], outs[0][key]), key


def test_set_mlp_ex_rejects_inconsistent_layouts():
    from optimalmodulationds_amd import _lib as L
    from optimalmodulationds_amd.engine import Engine
    rng = np.random.RandomState(0)
    m = _random_skip_net(rng, [256] * 4, [2])
    eng = Engine(7, 8, 2, 1, max_obs=8)
    with pytest.raises(L.OmdsError, match="input width"):
        eng.set_mlp(m.W, m.b, skip_after=())                 # 226 -> 256 without the concatenation
    with pytest.raises(L.OmdsError, match="above 256"):
        eng.set_mlp(m.W, m.b, skip_after=(0,))               # concatenation in the wrong place: 256 + 30 columns
    with pytest.raises(L.OmdsError, match="skip_after"):
        eng.set_mlp(m.W, m.b, skip_after=(3,))               # behind the output layer
    full = _random_skip_net(rng, [256] * 4, [])
    W = [full.W[0], np.zeros((256, 286), np.float32)] + full.W[2:]
    with pytest.raises(L.OmdsError, match="above 256"):
        eng.set_mlp(W, full.b, skip_after=(0,))              # 256 + 30 columns do not fit
    eng.set_mlp(m.W, m.b, skip_after=m.skip_after)
    eng.close()
