"""GPU tests of the fp16 screening of pass 1 (csrc/screen_kernel.hip): the screening values stay within the calibrated
bound of the fp32 pass-1 matrix, and a screened propagate is BIT-IDENTICAL to the fp32-pass-1 propagate -- the selected
obstacle sets are the fp32 ones, so every downstream number is the same -- on more than 10^6 (rollout, step) states."""
import numpy as np
import pytest

from helpers import SCENARIOS, load, weights_path
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu

KEYS = ("all_traj", "closest_dist_all", "kernel_val_all", "dot_products", "kernel_activations", "qdot", "normal")


def _engine(N, H, kind="franka", k=5, obs=None):
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path(kind))
    if kind == "franka":
        obs = scenes.shelf_scene() if obs is None else obs
        q0, qf = scenes.FRANKA_Q0, scenes.FRANKA_QF
        dst_thr, dt, ign = 0.01, 0.5, 0b111
    else:
        obs = scenes.planar7_scene() if obs is None else obs
        q0 = np.zeros(7, np.float32); q0[0] = np.pi / 2
        qf = np.zeros(7, np.float32); qf[0] = -np.pi / 2
        dst_thr, dt, ign = 0.25, 0.3, 0
    e = Engine(7, N, H, k, max_obs=max(64, obs.shape[0]))
    e.set_mlp(m.W, m.b)
    e.set_obstacles(obs)
    e.params.dt, e.params.dst_thr, e.params.ignored_links = dt, dst_thr, ign
    e.push_params()
    e.set_ds(qf)
    return e, m, obs, q0, qf


def test_screening_values_within_calibrated_bound():
    e, m, obs, q0, qf = _engine(512, 4)
    rng = np.random.RandomState(1)
    q = (q0 + (qf - q0) * rng.rand(512, 1) + 0.3 * rng.standard_normal((512, 7))).astype(np.float32)
    _, _, ref, _ = e.dist_grad(q, want_mindist=True)          # fp32 pass 1
    apx = e.screen_mindist(q)
    err = np.abs(apx - ref)
    assert np.isfinite(apx).all()
    assert err.max() < 5e-3, err.max()                        # fp16 network, fp32 accumulation: ~1e-3 m on the shelf scene
    # calibration happens at the first screened propagate
    e.set_screening(1)
    e.sample_policy(None, None, None, 0, 0, 0, 0, seed=1)
    e.propagate(q0)
    st = e.screen_stats()
    assert st["active"] and st["fallbacks"] == 0
    assert err.max() <= st["eps"], (err.max(), st)            # eps = 4 x calibration maximum covers an independent batch
    assert st["max_err_seen"] <= 0.5 * st["eps"]
    assert 5.0 <= st["candidates_per_rollout_step"] <= 64.0, st
    e.close()


def _run(e, q, mode, K, mu_c, sg_c, al_c, seed):
    e.set_screening(mode)
    e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=seed)
    e.propagate(q)
    return e.get_rollouts()


@pytest.mark.parametrize("N,H,iters", [(1024, 32, 32), (4096, 32, 8), (256, 6, 4)])
def test_screened_propagate_is_bit_identical(N, H, iters):
    """>= 10^6 (rollout, step) states per full-size shape: the screened propagate reproduces the fp32-pass-1 propagate bit
    for bit (free-running rollouts from moving start states, K = 10 sampled kernels).  The small shape is one where the
    unscreened step picks 16-row pass-2 tiles (v_mfma_f32_16x16x4, another order of the k sum) while the screened step
    always works on 32-row tiles: same selected obstacles, numbers equal to fp32 rounding instead of bit for bit."""
    e, m, obs, q0, qf = _engine(N, H)
    K = 10
    rng = np.random.RandomState(5)
    s = (np.arange(K) + 0.5) / K
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
    sg_c = np.ones(K, np.float32)
    al_c = rng.standard_normal((K, 7)).astype(np.float32)
    q = q0.copy()
    states = 0
    for it in range(iters):
        a = _run(e, q, 0, K, mu_c, sg_c, al_c, seed=100 + it)
        b = _run(e, q, 1, K, mu_c, sg_c, al_c, seed=100 + it)
        for key in KEYS:
            assert np.array_equal(a[key], b[key]), (it, key, float(np.abs(a[key] - b[key]).max()))
        states += N * H
        q = (q + 0.04 * (qf - q0) + 0.02 * rng.standard_normal(7)).astype(np.float32)
    st = e.screen_stats()
    # the start state drifts into the shelf over the iterations: the error bound widens with the errors seen on the way, and a
    # sudden doubling may cost one fp32 re-run (bit-identical by construction) -- never more than a couple
    assert st["fallbacks"] <= 2 and st["max_err_seen"] <= 0.5 * st["eps"], st
    assert iters < 8 or states >= 10 ** 6
    e.close()


def test_screened_teacher_forced_fixture():
    """The reference-captured Franka shelf fixture through the screened path (per-rollout start states): the screened step
    returns the bits of the all-fp32 step -- with the tile shapes each launcher picks for this small batch (N = 64: different
    heights for the two steps) and with both forced to 32-row tiles (omds_debug_force_tile_rows), not merely equal to rounding."""
    fx = load("franka_shelf_K6")
    from optimalmodulationds_amd import _lib
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path("franka"))
    N, H, K = int(fx["N"]), int(fx["H"]), int(fx["K"])
    hooks = _lib.load_test_hooks()
    for forced in ((0, 0), (32, 32)):
        outs = []
        for mode in (0, 1):
            e = Engine(7, N, H, int(fx["k"]), max_obs=512, lib=hooks)
            try:
                e.debug_force_tile_rows(*forced)
                e.set_mlp(m.W, m.b)
                e.set_obstacles(fx["obs"])
                e.params.dt, e.params.dst_thr = float(fx["dt"]), float(fx["dst_thr"])
                e.params.ignored_links = sum(1 << int(l) for l in fx["ignored_links"])
                e.push_params()
                e.set_ds(fx["qf"])
                e.set_screening(mode)
                e.set_policy_samples(fx["it0_mu_tmp"][:, :K], fx["it0_sigma_tmp"][:, :K], fx["it0_alpha_tmp"][:, :K])
                e.propagate(fx["it0_q_cur"])
                outs.append(e.get_rollouts())
                assert e.screen_stats()["fallbacks"] == 0
            finally:
                e.debug_force_tile_rows(0, 0)
                e.close()
        for key in KEYS:
            assert np.array_equal(outs[0][key], outs[1][key]), (forced, key, float(np.nanmax(np.abs(outs[0][key] - outs[1][key]))))


def test_guards_trip_on_a_bound_that_is_too_small():
    """A caller-supplied bound far below the real screening error: the run-time guards (largest error on the re-evaluated
    rows; per-rollout slack tau - D*_k >= eps) must reject the screened propagate, the fp32 re-run must give the all-fp32
    result bit for bit, and the bound must have been widened afterwards."""
    e, m, obs, q0, qf = _engine(1024, 8)
    K = 10
    rng = np.random.RandomState(9)
    s = (np.arange(K) + 0.5) / K
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
    sg_c, al_c = np.ones(K, np.float32), rng.standard_normal((K, 7)).astype(np.float32)
    a = _run(e, q0, 0, K, mu_c, sg_c, al_c, seed=7)
    e.set_screening(1, 2e-4)
    e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=7)
    e.propagate(q0)
    b = e.get_rollouts()
    st = e.screen_stats()
    assert st["fallbacks"] == 1 and st["eps"] >= 4 * st["max_err_seen"] > 8e-4, st
    for key in KEYS:
        assert np.array_equal(a[key], b[key]), key
    e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=7)     # with the widened bound the same propagate is accepted
    e.propagate(q0)
    c = e.get_rollouts()
    assert e.screen_stats()["fallbacks"] == 1
    for key in KEYS:
        assert np.array_equal(a[key], c[key]), key
    e.close()


@pytest.mark.parametrize("scene", ["cross", "cloud600", "sub40"])
def test_screened_propagate_is_bit_identical_on_other_scenes(scene):
    """The same bit-identity on scenes with other obstacle statistics than the shelf: the 28-sphere cross (few, clustered),
    a random cloud of 600 spheres around the arm (many near-ties), and 40 shelf spheres (every rollout's candidates are a
    large share of its obstacles) -- screening forced on, the bound calibrated per scene at the first propagate."""
    from optimalmodulationds_amd import scenes
    if scene == "cross":
        obs = scenes.cross_scene(0.45)
    elif scene == "sub40":
        obs = scenes.shelf_scene()[::7][:40]
    else:
        rng = np.random.RandomState(3)
        p = rng.uniform([-0.2, -0.7, 0.0], [0.9, 0.7, 1.1], (600, 3))
        obs = np.c_[p, rng.uniform(0.02, 0.08, 600)].astype(np.float32)
    N, H, K = 1024, 8, 10
    e, m, obs, q0, qf = _engine(N, H, obs=obs)
    rng = np.random.RandomState(5)
    s = (np.arange(K) + 0.5) / K
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
    sg_c, al_c = np.ones(K, np.float32), rng.standard_normal((K, 7)).astype(np.float32)
    q = q0.copy()
    for it in range(6):
        a = _run(e, q, 0, K, mu_c, sg_c, al_c, seed=300 + it)
        b = _run(e, q, 1, K, mu_c, sg_c, al_c, seed=300 + it)
        for key in KEYS:
            assert np.array_equal(a[key], b[key]), (scene, it, key, float(np.abs(a[key] - b[key]).max()))
        q = (q + 0.05 * (qf - q0) + 0.02 * rng.standard_normal(7)).astype(np.float32)
    st = e.screen_stats()
    assert st["active"] and st["fallbacks"] <= 2 and st["max_err_seen"] <= 0.5 * st["eps"], st
    e.close()


def _net_engine(kind, N, H, obs):
    """Franka context with one of the weight sets under tests/golden/weights (activation and skip layout from the file)."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path(kind))
    e = Engine(7, N, H, 5, max_obs=max(64, obs.shape[0]))
    e.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
    e.set_obstacles(obs)
    e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0b111
    e.push_params()
    e.set_ds(scenes.FRANKA_QF)
    return e


@pytest.mark.parametrize("kind,N,H,iters", [("franka_tanh", 1024, 8, 6), ("franka_skip", 1024, 8, 6), ("franka_tanh", 4096, 32, 2),
                                            ("franka_skip", 4096, 16, 2)])
def test_screened_tanh_and_skip_networks_are_bit_identical(kind, N, H, iters):
    """BASELINE configs[2] as worded (256-256-256 tanh) and the reference's skip-connection layout
    (MLPRegression(skips=[2]), network_macros_mod.py:117-146) through the screened step: k_screen with the tanh epilogue /
    the concatenation stage; ReLU + skip keeps k_exact's masks and k_tail_sel, tanh takes the matrix route (k_exact writes
    the exact values into Dmin, k_tail runs its own forward).  Two contexts (fp32 pass 1 / screened), every returned array
    equal, free-running from drifting start states."""
    from optimalmodulationds_amd import scenes
    obs = scenes.shelf_scene()
    e0, e1 = _net_engine(kind, N, H, obs), _net_engine(kind, N, H, obs)
    e0.set_screening(0)
    e1.set_screening(1)
    q0, qf = scenes.FRANKA_Q0, scenes.FRANKA_QF
    K = 10
    rng = np.random.RandomState(5)
    s = (np.arange(K) + 0.5) / K
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
    sg_c, al_c = np.ones(K, np.float32), rng.standard_normal((K, 7)).astype(np.float32)
    q = q0.copy()
    for it in range(iters):
        outs = []
        for e in (e0, e1):
            e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=70 + it)
            e.propagate(q)
            outs.append(e.get_rollouts())
        for key in KEYS:
            assert np.array_equal(outs[0][key], outs[1][key]), (kind, it, key, float(np.nanmax(np.abs(outs[0][key] - outs[1][key]))))
        q = (q + 0.05 * (qf - q0) + 0.02 * rng.standard_normal(7)).astype(np.float32)
    st = e1.screen_stats()
    print(kind, N, H, st)
    assert st["active"] and not st["suspended"] and st["fallbacks"] <= 2, st
    assert st["max_err_seen"] <= 0.5 * st["eps"] and st["audit_max_err"] <= 0.5 * st["eps"], st
    assert st["audit_rows_per_rollout_step"] > 1.0, st
    e0.close()
    e1.close()


@pytest.mark.parametrize("name,kind", [("franka_tanh_shelf_K4", "franka_tanh"), ("franka_skip_shelf_K4", "franka_skip")])
def test_screened_tanh_and_skip_fixtures(name, kind):
    """The reference-generated tanh / skip rollout fixtures (its own MLPRegression with act_fn=Tanh / skips=[2]) through
    the screened step, teacher-forced from the fixture's start states: the same numbers as the fp32 step."""
    fx = load(name)
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path(kind))
    N, H, K = int(fx["N"]), int(fx["H"]), int(fx["K"])
    outs = []
    for mode in (0, 1):
        e = Engine(7, N, H, int(fx["k"]), max_obs=512)
        e.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
        e.set_obstacles(fx["obs"])
        e.params.dt, e.params.dst_thr = float(fx["dt"]), float(fx["dst_thr"])
        e.params.ignored_links = sum(1 << int(l) for l in fx["ignored_links"])
        e.push_params()
        e.set_ds(fx["qf"])
        e.set_screening(mode)
        e.set_policy_samples(fx["it0_mu_tmp"][:, :K], fx["it0_sigma_tmp"][:, :K], fx["it0_alpha_tmp"][:, :K])
        e.propagate(fx["it0_q_cur"])
        outs.append(e.get_rollouts())
        if mode:
            st = e.screen_stats()
            assert st["active"] and st["fallbacks"] == 0, st
        e.close()
    for key in KEYS:   # small N: the unscreened relu step may pick 16-row pass-2 tiles where k_tail_sel picks its own height -> the same bits either way (gemm16)
        assert np.array_equal(outs[0][key], outs[1][key]), (key, float(np.nanmax(np.abs(outs[0][key] - outs[1][key]))))
    # and against the reference's own rollouts of the fixture (first step from the fixture's start states)
    ref = fx["it0_all_traj"]
    assert np.abs(outs[1]["all_traj"][:, 1] - ref[:, 1]).max() <= 2e-4, float(np.abs(outs[1]["all_traj"][:, 1] - ref[:, 1]).max())


def test_rows_longer_than_a_workgroups_result_buffer_take_the_matrix_route():
    """O = 6000 obstacles: a rollout's screening values no longer fit one workgroup's LDS (5120), so k_screen writes the
    matrix and k_select -- its long-row form, values re-read instead of held in registers -- does the selection: same bits."""
    rng = np.random.RandomState(8)
    p = rng.uniform([-0.3, -0.8, 0.0], [1.0, 0.8, 1.2], (6000, 3))
    obs = np.c_[p, rng.uniform(0.01, 0.04, 6000)].astype(np.float32)
    N, H, K = 96, 3, 4
    e, m, obs, q0, qf = _engine(N, H, obs=obs)
    rng = np.random.RandomState(5)
    mu_c = (q0 + 0.2 * rng.standard_normal((K, 7))).astype(np.float32)
    sg_c, al_c = np.ones(K, np.float32), rng.standard_normal((K, 7)).astype(np.float32)
    a = _run(e, q0, 0, K, mu_c, sg_c, al_c, seed=3)
    b = _run(e, q0, 1, K, mu_c, sg_c, al_c, seed=3)
    for key in KEYS:
        assert np.array_equal(a[key], b[key]), (key, float(np.abs(a[key] - b[key]).max()))
    st = e.screen_stats()
    assert st["active"] and st["fallbacks"] <= 1 and st["audit_rows_per_rollout_step"] > 10, st
    e.close()


@pytest.mark.parametrize("kind,n,k", [("planar7", 7, 1), ("planar7", 7, 3), ("planar2", 2, 2), ("planar7_128", 7, 2)])
def test_screened_planar_networks_are_bit_identical(kind, n, k):
    """The planar robots' networks (15 / 30 inputs, 2 / 7 output channels, distances not divided by 100, no ignored links, the
    128-wide net zero-padded to 256) on a cloud of 700 discs: the screened step of the 2- and 7-DoF tails, other k."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path(kind))
    rng = np.random.RandomState(12)
    reach = 6.5 if n == 2 else 7.5
    obs = np.c_[rng.uniform(-reach, reach, (700, 2)), np.zeros(700), rng.uniform(0.2, 0.6, 700)].astype(np.float32)
    q0 = np.zeros(n, np.float32); q0[0] = np.pi / 2
    qf = np.zeros(n, np.float32); qf[0] = -np.pi / 2
    N, H, K = 512, 6, 4
    outs, stats = [], None
    mu_c = (q0 + 0.3 * rng.standard_normal((K, n))).astype(np.float32)
    sg_c, al_c = np.full(K, 0.5, np.float32), rng.standard_normal((K, n)).astype(np.float32)
    for mode in (0, 1):
        e = Engine(n, N, H, k, max_obs=1024)
        e.set_mlp(m.W, m.b)
        e.set_obstacles(obs)
        e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.3, 0.25, 0
        e.push_params()
        e.set_ds(qf)
        e.set_screening(mode)
        q = q0.copy()
        runs = []
        for it in range(3):
            e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 0.75, K, seed=20 + it)
            e.propagate(q)
            runs.append(e.get_rollouts())
            q = (q + 0.1 * (qf - q0)).astype(np.float32)
        outs.append(runs)
        if mode:
            stats = e.screen_stats()
        e.close()
    for it in range(3):
        for key in KEYS:
            assert np.array_equal(outs[0][it][key], outs[1][it][key]), (kind, it, key, float(np.nanmax(np.abs(outs[0][it][key] - outs[1][it][key]))))
    print(kind, k, stats)
    assert stats["active"] and not stats["suspended"] and stats["fallbacks"] <= 1, stats


@pytest.mark.parametrize("kind,n,k,N", [("planar2", 2, 1, 512), ("planar2", 2, 2, 512), ("planar2", 2, 5, 300), ("planar7", 7, 3, 512),
                                         ("franka", 7, 5, 1024), ("franka", 7, 4, 250), ("franka_dup", 7, 7, 600), ("franka_dup", 7, 10, 333)])
def test_every_tile_shape_computes_the_same_bits(kind, n, k, N):
    """The tail kernels choose a tile shape from the batch (32-row, 16-row, 4-row groups: the backward of the screened step, forward and
    backward of the unscreened one); the choice must not show in the results.  Forces each shape in turn (omds_debug_force_tile_rows) for the screened
    and the unscreened step and compares every rollout tensor bit for bit -- the 2-DoF tail with the 4-row groups is the
    case in which a cross-statement multiply-add contraction once differed between two instantiations of the same source."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    dup = kind == "franka_dup"   # every 4th obstacle twice: exactly equal distances, the top-k's tie rule (lower index) in every shape
    kind = "franka" if dup else kind
    m = orc.Mlp.from_npz(weights_path(kind))
    rng = np.random.RandomState(5)
    if kind == "franka":
        obs, q0, qf = scenes.shelf_scene(), scenes.FRANKA_Q0, scenes.FRANKA_QF
        if dup:
            obs = np.concatenate([obs, obs[::4]]).astype(np.float32)
        dt, thr, ign = 0.5, 0.01, 0b111
    else:
        reach = 6.5 if n == 2 else 7.5
        obs = np.c_[rng.uniform(-reach, reach, (700, 2)), np.zeros(700), rng.uniform(0.2, 0.6, 700)].astype(np.float32)
        q0 = np.zeros(n, np.float32); q0[0] = np.pi / 2
        qf = np.zeros(n, np.float32); qf[0] = -np.pi / 2
        dt, thr, ign = 0.3, 0.25, 0
    H, K = 6, 4
    mu_c = (q0 + 0.3 * rng.standard_normal((K, n))).astype(np.float32)
    sg_c, al_c = np.full(K, 0.5, np.float32), rng.standard_normal((K, n)).astype(np.float32)

    from optimalmodulationds_amd import _lib
    hooks = _lib.load_test_hooks()   # libomds_hip_test.so: the product's objects + the hooks of include/omds_test.h

    def run(mode, sel_rows, tail_rows):
        e = Engine(n, N, H, k, max_obs=1024, lib=hooks)
        try:
            e.debug_force_tile_rows(sel_rows, tail_rows)
            e.set_mlp(m.W, m.b)
            e.set_obstacles(obs)
            e.params.dt, e.params.dst_thr, e.params.ignored_links = dt, thr, ign
            e.push_params()
            e.set_ds(qf)
            e.set_screening(mode)
            out = []
            q = q0.copy()
            for it in range(2):
                e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 0.75, K, seed=40 + it)
                e.propagate(q)
                out.append(e.get_rollouts())
                q = (q + 0.1 * (qf - q0)).astype(np.float32)
            st = e.screen_stats()
            return out, st
        finally:
            e.debug_force_tile_rows(0, 0)
            e.close()

    ref, _ = run(0, 0, 32)
    variants = [("unscreened 16-row", 0, 0, 16), ("unscreened 4-row groups", 0, 0, 4), ("screened 32-row", 1, 32, 0), ("screened 16-row", 1, 16, 0), ("screened 4-row groups", 1, 4, 0)]
    for name, mode, sel_rows, tail_rows in variants:
        if sel_rows == 16 and k > 16:
            continue
        got, st = run(mode, sel_rows, tail_rows)
        if mode:
            assert st["active"] and st["fallbacks"] == 0, (name, st)   # a fallback would compare the fp32 step with itself
        for it in range(2):
            for key in KEYS:
                assert np.array_equal(ref[it][key], got[it][key], equal_nan=True), (kind, k, name, it, key, float(np.nanmax(np.abs(ref[it][key] - got[it][key]))))


@pytest.mark.parametrize("kind", ["franka", "planar7_128"])
def test_unit_reorder_changes_no_result(kind):
    """Behind a calibration the hidden units of the fp16 pack are sorted by how often they fire (omds_screen_order_stats) so that
    k_screen's zero test finds whole k-chunks dead.  The order is a property of the SCREENING pack only: the screening values stay
    inside the bound, and every number a propagate returns is the all-fp32 step's -- over five propagates, the fourth of which
    calibrates (and reorders) again.  The 128-wide network is zero-padded to 256: half of its chunks are dead in
    any order, and none of its padded units may ever count as having fired."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.engine import Engine
    m = orc.Mlp.from_npz(weights_path(kind))
    N, H = 256, 6
    if kind == "franka":
        obs, q0, qf, dst_thr, dt, ign = scenes.shelf_scene(), scenes.FRANKA_Q0, scenes.FRANKA_QF, 0.01, 0.5, 0b111
    else:
        obs = np.tile(scenes.planar7_scene(), (40, 1)).astype(np.float32)      # enough pairs for the screened step (N * O >= 65536)
        obs[:, :2] += np.random.RandomState(4).uniform(-3, 3, (obs.shape[0], 2)).astype(np.float32)
        q0 = np.zeros(7, np.float32); q0[0] = np.pi / 2
        qf = np.zeros(7, np.float32); qf[0] = -np.pi / 2
        dst_thr, dt, ign = 0.25, 0.3, 0

    def ctx(mode):
        e = Engine(7, N, H, 5, max_obs=max(64, obs.shape[0]))
        e.set_mlp(m.W, m.b)
        e.set_obstacles(obs)
        e.params.dt, e.params.dst_thr, e.params.ignored_links = dt, dst_thr, ign
        e.push_params()
        e.set_ds(qf)
        e.set_screening(mode)
        return e
    a, b = ctx(1), ctx(0)
    rng = np.random.RandomState(7)
    q = np.asarray(q0, np.float32).copy()
    nre = []
    for it in range(5):
        if it == 3:
            a.set_screening(1, -1.0)                     # discard the calibration: the next propagate calibrates and reorders again
        for e in (a, b):
            e.sample_policy(None, None, None, 0, 0, 0, 0, seed=100 + it)
            e.propagate(q)
        ra, rb = a.get_rollouts(), b.get_rollouts()
        for key in KEYS:
            assert np.array_equal(ra[key], rb[key]), (it, key)
        st = a.screen_stats()
        assert st["active"] and (st["fallbacks"] == 0 or kind != "franka"), st
        nre.append(st["unit_reorders"])
        q = (q + 0.05 * (np.asarray(qf) - q) + 0.02 * rng.standard_normal(7)).astype(np.float32)
    assert nre == [2, 2, 2, 4, 4], nre                   # two per calibration: on its batch, then on the states the rollouts reached
    st = a.screen_stats()
    never = st["units_never_fired"]
    assert len(never) == len(m.W) - 1
    if kind == "planar7_128":
        assert all(v >= 128 for v in never), never       # the zero padding of a 128-wide layer never fires
    else:
        assert sum(never) >= 150, never                  # the shipped network: a third of its hidden units are silent on this scene
    # the reordered pack still computes the screening function: values within the bound of the fp32 matrix
    qq = (np.asarray(q0) + (np.asarray(qf) - np.asarray(q0)) * rng.rand(128, 1) + 0.3 * rng.standard_normal((128, 7))).astype(np.float32)
    _, _, ref, _ = b.dist_grad(qq, want_mindist=True)
    apx = a.screen_mindist(qq)
    assert np.abs(apx - ref).max() <= st["eps"], (np.abs(apx - ref).max(), st["eps"])
    assert b.screen_stats()["unit_reorders"] == 0
    a.close(); b.close()
