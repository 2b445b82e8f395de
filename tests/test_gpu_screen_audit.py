"""GPU tests of what the screened step MEASURES about itself (csrc/capi.hip omds_propagate): the audit sample
(a pseudo-random subset of the pairs that are NOT re-evaluated, drawn anew every step and evaluated in fp32 by k_audit at the
end of the horizon loop), the
recalibration of the bound when the scene changes, and the fp32 fallback.  Every case runs the same planner iterations on
two contexts -- screening off / screening on, each set ONCE, so the bound persists over the run as it does in production --
and demands array equality of everything a propagate returns.

Scenes: the reference's streamed obstacle sets (obstacleStreamer.py:28-142, obstacleStreamerBenchmark.py:30-81) restated in
optimalmodulationds_amd/scenes.py; the moving shelf is the streamer's sinusoid (obstacleStreamer.py:120-137)."""
import numpy as np
import pytest

from helpers import weights_path
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu

KEYS = ("all_traj", "closest_dist_all", "kernel_val_all", "dot_products", "kernel_activations", "qdot", "normal")


def _pair(N, H, obs, k=5, max_obs=None, audit=None):
    """Two Franka contexts on the same scene: [fp32 pass 1, screened]."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    from optimalmodulationds_amd.engine import Engine
    from optimalmodulationds_amd import _lib
    m = orc.Mlp.from_npz(weights_path("franka"))
    out = []
    for mode in (0, 1):
        # the screened context comes from libomds_hip_test.so (the product's objects + the damage hook of include/omds_test.h)
        e = Engine(7, N, H, k, max_obs=max_obs or max(64, obs.shape[0]), lib=_lib.load_test_hooks() if mode else None)
        e.set_mlp(m.W, m.b)
        e.set_obstacles(obs)
        e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0b111
        e.push_params()
        e.set_ds(scenes.FRANKA_QF)
        e.set_cost(scenes.franka_dh_params(), np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32))
        e.set_screening(mode)
        if audit is not None:
            e.set_screening_audit(audit)
        out.append(e)
    return out


def _policy(K=10, seed=5):
    from optimalmodulationds_amd import scenes
    q0, qf = scenes.FRANKA_Q0, scenes.FRANKA_QF
    rng = np.random.RandomState(seed)
    s = (np.arange(K) + 0.5) / K
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, 7))).astype(np.float32)
    return mu_c, np.ones(K, np.float32), rng.standard_normal((K, 7)).astype(np.float32)


def _step_both(engines, q, pol, seed, label):
    mu_c, sg_c, al_c = pol
    outs = []
    for e in engines:
        e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, mu_c.shape[0], seed=seed)
        e.propagate(q)
        outs.append(e.get_rollouts())
    for key in KEYS:
        assert np.array_equal(outs[0][key], outs[1][key]), (label, key, float(np.nanmax(np.abs(outs[0][key] - outs[1][key]))))
    return outs[1]


def test_moving_shelf_1024x32_is_bit_identical_and_audited():
    """BASELINE configs[4] per GPU at bench.py's franka_dynamic_1024x32 shape: 32 planner iterations, update_obstacles before
    every one (the shelf rides the streamer's sinusoid), ONE calibration, the bound persisting over the run."""
    from optimalmodulationds_amd import scenes
    shelf = scenes.shelf_scene()
    e0, e1 = engines = _pair(1024, 32, shelf)
    pol = _policy()
    q = scenes.FRANKA_Q0.copy()
    rng = np.random.RandomState(11)
    for it in range(32):
        moved = shelf.copy()
        moved[:, 1] += 0.05 * np.sin(0.3 * it)
        for e in engines:
            e.set_obstacles(moved)
        r = _step_both(engines, q, pol, 500 + it, f"iteration {it}")
        q = (q + 0.05 * r["qdot"][0] + 0.01 * rng.standard_normal(7)).astype(np.float32)
    st = e1.screen_stats()
    assert st["active"] and not st["suspended"], st
    assert st["calibrations"] == 1, st                       # +-0.05 m stays inside the 0.1 m signature tolerance
    assert st["fallbacks"] <= 2, st
    assert st["audit_one_in"] == 128 and 1.5 <= st["audit_rows_per_rollout_step"] <= 3.0, st   # (294 - ~7.4) / 128
    assert st["audit_max_err"] <= 0.5 * st["eps"] and st["max_err_seen"] <= 0.5 * st["eps"], st
    for e in engines:
        e.close()


def test_scene_swaps_recalibrate_and_stay_bit_identical():
    """A driver that starts on far-away placeholder spheres (obstacleStreamer.py:133-134) and then receives
    shelf -> cross -> ring -> I-shape -> wall -> line: every swap is recalibrated against the new scene
    (omds_set_obstacles compares with the calibrated set), and every propagate equals the fp32 one."""
    from optimalmodulationds_amd import scenes
    N, H = 1024, 8
    engines = _pair(N, H, scenes.placeholder_scene(5), max_obs=64)       # grows to 294 on the way
    pol = _policy()
    q = scenes.FRANKA_Q0.copy()
    seq = [("placeholder", scenes.placeholder_scene(5)), ("shelf", scenes.shelf_scene()), ("cross", scenes.cross_scene()),
           ("ring", scenes.ring_scene()), ("tshape", scenes.tshape_scene()), ("wall", scenes.wall_scene()),
           ("line", scenes.line_scene()), ("cross-lowered", scenes.cross_scene(0.35))]
    it = 0
    for name, obs in seq:
        for e in engines:
            e.set_obstacles(obs)
        for rep in range(3):
            r = _step_both(engines, q, pol, 900 + it, f"{name} #{rep}")
            q = (q + 0.04 * r["qdot"][0]).astype(np.float32)
            it += 1
    st = engines[1].screen_stats()
    assert st["calibrations"] == len(seq), st                # one per scene, none in between
    assert not st["suspended"], st
    assert engines[1].max_obs >= 294
    for e in engines:
        e.close()


def test_out_of_range_obstacles_and_uniform_start_states():
    """Obstacles outside anything the network was trained on (5 m away, radius 0.3 m) mixed into the shelf, and per-rollout
    start states uniform in the joint box instead of near the q0 -> qf line."""
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    rng = np.random.RandomState(3)
    far = np.zeros((40, 4), np.float32)
    ang = rng.uniform(0, 2 * np.pi, 40)
    far[:, 0], far[:, 1], far[:, 2], far[:, 3] = 5.0 * np.cos(ang), 5.0 * np.sin(ang), rng.uniform(-1, 2, 40), 0.3
    obs = np.vstack((scenes.shelf_scene(), far)).astype(np.float32)
    N, H = 1024, 8
    engines = _pair(N, H, obs)
    pol = _policy()
    lo, hi = np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32)
    for it in range(4):
        q = (lo + (hi - lo) * rng.rand(N, 7)).astype(np.float32)
        _step_both(engines, q, pol, 40 + it, f"uniform starts #{it}")
    st = engines[1].screen_stats()
    assert st["active"] and not st["suspended"] and st["fallbacks"] <= 2, st
    assert st["audit_max_err"] <= 0.5 * st["eps"], st
    for e in engines:
        e.close()


def test_corrupted_weight_fragment_is_caught_and_suspends_screening():
    """One 1-KiB fragment of the fp16 weight pack zeroed: every screening value moves.  The propagate must notice (candidate
    and audit errors), be redone in fp32 -- results still the fp32 ones -- and after three such propagates the context stays
    on the fp32 step; a fresh omds_set_mlp restores screening."""
    from optimalmodulationds_amd import scenes
    N, H = 512, 6
    engines = _pair(N, H, scenes.shelf_scene())
    pol = _policy()
    q = scenes.FRANKA_Q0.copy()
    _step_both(engines, q, pol, 1, "before")
    st0 = engines[1].screen_stats()
    assert st0["fallbacks"] == 0 and st0["active"]
    engines[1].screen_debug_corrupt(0, 16 * 5 + 3)          # slice 5 = a row block of the first hidden->hidden layer
    for it in range(4):
        _step_both(engines, q, pol, 2 + it, f"corrupted #{it}")
    st = engines[1].screen_stats()
    assert st["fallbacks"] == 3 and st["suspended"] and not st["active"], st     # the 4th propagate ran fp32 directly
    assert st["audit_max_err"] > 0.5 * st0["eps"] or st["max_err_seen"] > 0.5 * st0["eps"], st
    m = orc.Mlp.from_npz(weights_path("franka"))
    engines[1].set_mlp(m.W, m.b)                             # new packs, new calibration
    _step_both(engines, q, pol, 9, "restored")
    st = engines[1].screen_stats()
    assert st["active"] and not st["suspended"] and st["fallbacks"] == 3, st
    for e in engines:
        e.close()


def test_only_the_audit_rows_can_see_a_misplaced_far_looking_obstacle():
    """The failure the audit exists for: the closest obstacle of the start state is shifted by 1.5 m in the SCREENING inputs
    only, so the fp16 network calls it far, it is no candidate, and no candidate-side check can notice.  With the audit
    sample the propagate is caught and redone in fp32; without it (one_in = 0) the wrong obstacle set goes through."""
    from optimalmodulationds_amd import scenes
    N, H = 1024, 8
    obs = scenes.shelf_scene()
    q = scenes.FRANKA_Q0.copy()
    pol = _policy()
    engines = _pair(N, H, obs)
    _, _, _, idx = engines[0].dist_grad(q[None], want_idx=True)
    victim = int(idx[0, 0])
    _step_both(engines, q, pol, 1, "before")
    eps0 = engines[1].screen_stats()["eps"]
    engines[1].screen_debug_corrupt(1, victim, 1.5)
    _step_both(engines, q, pol, 2, "audited")                # identical because the audit forced the fp32 redo
    st = engines[1].screen_stats()
    assert st["fallbacks"] == 1, st
    assert st["audit_max_err"] > 0.5 * eps0, st              # the audit rows measured it ...
    assert st["max_err_seen"] <= 0.5 * eps0, st              # ... the candidates could not
    for e in engines:
        e.close()
    # the same damage with the audit switched off goes unnoticed: this is what the audit rows are for (sweeps off too: the first
    # propagate after the pack's re-sort would carry one, and a sweep sees every pair of its step)
    engines = _pair(N, H, obs, audit=0)
    engines[1].set_screening_sweep(0)
    mu_c, sg_c, al_c = pol
    outs = []
    for e in engines:
        e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, mu_c.shape[0], seed=2)
        if e is engines[1]:
            e.propagate(q)                                   # calibrates on the intact tables
            e.screen_debug_corrupt(1, victim, 1.5)
        e.propagate(q)
        outs.append(e.get_rollouts())
    st = engines[1].screen_stats()
    assert st["fallbacks"] == 0 and st["audit_rows_per_rollout_step"] == 0.0, st
    assert not np.array_equal(outs[0]["closest_dist_all"], outs[1]["closest_dist_all"])
    for e in engines:
        e.close()


def test_mode_changes_keep_the_bound_and_negative_eps_recalibrates():
    from optimalmodulationds_amd import scenes
    e0, e1 = _pair(256, 4, scenes.shelf_scene())
    q = scenes.FRANKA_Q0.copy()
    e1.sample_policy(None, None, None, 0, 0, 0, 0, seed=1)
    e1.propagate(q)
    st = e1.screen_stats()
    assert st["calibrations"] == 1 and st["eps"] > 0
    e1.set_screening(0)
    e1.set_screening(1)                                      # eps = 0: mode only
    e1.propagate(q)
    st2 = e1.screen_stats()
    assert st2["calibrations"] == 1 and st2["eps"] >= st["eps"]
    e1.set_screening(1, -1.0)                                # forget: measured again
    e1.propagate(q)
    assert e1.screen_stats()["calibrations"] == 2
    e1.set_screening(1, 0.02)                                # the caller's bound: used as is
    e1.set_obstacles(scenes.cross_scene())
    e1.propagate(q)
    st3 = e1.screen_stats()
    assert st3["calibrations"] == 2 and st3["eps"] == pytest.approx(0.02)
    e0.close()
    e1.close()


def test_sweep_checks_every_pair_of_a_step_and_catches_what_a_disabled_audit_misses():
    """The sweep (every 32nd screened propagate by default: ALL N x O pairs of the last horizon step in fp32 beside their
    screening values) on a clean run stays below the bound; and with the audit sample switched off and a sweep on every
    propagate, the misplaced far-looking obstacle is caught by the sweep alone."""
    from optimalmodulationds_amd import scenes
    N, H = 1024, 8
    obs = scenes.shelf_scene()
    q = scenes.FRANKA_Q0.copy()
    pol = _policy()
    engines = _pair(N, H, obs)
    for it in range(34):
        r = _step_both(engines, q, pol, 700 + it, f"clean #{it}")
        q = (q + 0.02 * r["qdot"][0]).astype(np.float32)
    st = engines[1].screen_stats()
    assert st["sweep_every"] == 32 and st["sweeps"] == 3, st            # propagates 0 and 32, and 1: the first on the re-sorted pack
    assert 0.0 < st["sweep_max_err"] <= 0.5 * st["eps"] and st["fallbacks"] == 0, st
    for e in engines:
        e.close()
    engines = _pair(N, H, obs, audit=0)
    engines[1].set_screening_sweep(1)
    q = scenes.FRANKA_Q0.copy()
    _, _, _, idx = engines[0].dist_grad(q[None], want_idx=True)
    _step_both(engines, q, pol, 1, "before")
    eps0 = engines[1].screen_stats()["eps"]
    engines[1].screen_debug_corrupt(1, int(idx[0, 0]), 1.5)
    _step_both(engines, q, pol, 2, "swept")                             # identical because the sweep forced the fp32 redo
    st = engines[1].screen_stats()
    assert st["fallbacks"] == 1 and st["sweep_max_err"] > 0.5 * eps0 and st["audit_rows_per_rollout_step"] == 0.0, st
    for e in engines:
        e.close()


def test_all_steps_sweep_counts_every_unevaluated_pair_of_1e8():
    """The soak mode of tools/sweep_soak.py at test size: EVERY horizon step of every propagate swept -- all N x O pairs in fp32
    beside their screening values -- over > 1e8 (rollout, obstacle) pairs on the moving shelf at 1024 x 32.  The assumption of the
    selection rule (omds.h) is about the pairs that are NOT re-evaluated: Da - D <= eps.  Here that population is counted
    exhaustively, not sampled: none above eps / 2 (the acceptance margin), none above eps (a possible miss), none non-finite; and
    every propagate is bit-identical to the fp32 step, as the proof says it must be under that condition."""
    from optimalmodulationds_amd import scenes
    N, H = 1024, 32
    shelf = scenes.shelf_scene()
    engines = _pair(N, H, shelf)
    engines[1].set_screening_sweep(1, all_steps=True)
    pol = _policy()
    q = scenes.FRANKA_Q0.copy()
    rng = np.random.RandomState(3)
    for it in range(11):
        moved = shelf.copy()
        moved[:, 1] += 0.05 * np.sin(0.3 * it)
        for e in engines:
            e.set_obstacles(moved)
        r = _step_both(engines, q, pol, 900 + it, f"swept iteration {it}")
        q = (q + 0.05 * r["qdot"][0] + 0.01 * rng.standard_normal(7)).astype(np.float32)
    st, hs = engines[1].screen_stats(), engines[1].sweep_hist()
    print({k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in hs.items()}, st)
    assert hs["steps"] == 11 * H and hs["pairs"] == 11 * H * N * shelf.shape[0] and hs["pairs"] > 1e8, hs
    cand = st["candidates_per_rollout_step"] * N * H * 11
    assert abs((hs["pairs"] - hs["non_candidates"]) - cand) <= 1e-6 * hs["pairs"], (hs, cand)   # the sweep's candidate test = the selection's
    assert hs["above_half_eps"] == 0 and hs["above_eps"] == 0 and hs["non_finite"] == 0, hs
    assert 0.0 < hs["max_pos"] <= 0.5 * st["eps"] and hs["max_abs"] <= 0.5 * st["eps"], (hs, st)
    assert hs["max_abs"] == pytest.approx(st["sweep_max_err"]) and st["fallbacks"] == 0, (hs, st)
    assert int(hs["pos"].sum() + hs["neg"].sum()) == hs["non_candidates"], hs
    assert int(hs["ratio"][len(hs["ratio"]) // 2:].sum()) == 0, hs                     # nothing in the upper half of [0, eps)
    for e in engines:
        e.close()


def test_large_radii_and_in_collision_rollouts():
    """Spheres of radius 0.1 .. 0.3 m scattered through the arm's workspace: most rollouts spend steps in collision (negative
    thresholded distance, the in-collision branch of the modulation) and the pass-1 values are dominated by the radii."""
    from optimalmodulationds_amd import scenes
    rng = np.random.RandomState(21)
    p = rng.uniform([-0.4, -0.8, 0.0], [0.9, 0.8, 1.2], (250, 3))
    obs = np.c_[p, rng.uniform(0.1, 0.3, 250)].astype(np.float32)
    N, H = 1024, 8
    engines = _pair(N, H, obs)
    pol = _policy()
    q = scenes.FRANKA_Q0.copy()
    ncoll = 0
    for it in range(5):
        r = _step_both(engines, q, pol, 60 + it, f"large radii #{it}")
        ncoll += int((r["closest_dist_all"] < 0).sum())
        q = (q + 0.05 * (scenes.FRANKA_QF - scenes.FRANKA_Q0)).astype(np.float32)
    st = engines[1].screen_stats()
    assert ncoll > 0.2 * 5 * N * H, ncoll                                 # the scene does put the rollouts into collision
    assert st["active"] and not st["suspended"] and st["fallbacks"] <= 1, st
    assert st["audit_max_err"] <= 0.5 * st["eps"] and st["sweep_max_err"] <= 0.5 * st["eps"], st
    for e in engines:
        e.close()
