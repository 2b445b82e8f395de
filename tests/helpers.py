"""Shared helpers for the parity tests: fixture loading and tolerant comparisons."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_ALL = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith(("mlp_", "fk_", "bases_", "seds_", "train_", "wgrad_")))
SCENARIOS = [s for s in _ALL if not s.startswith("toy")]          # MPPI.py fixtures (tools/make_golden.py)
TOY_SCENARIOS = [s for s in _ALL if s.startswith("toy")]          # MPPI_toy.py fixtures (tools/make_golden_toy.py)
SEDS_FILES = ["seds_left10", "seds_right", "seds_sine10", "seds_2d"]   # SEDS.get_velocity known answers (tools/make_golden_seds.py)
MLP_KINDS = ["franka", "planar7", "planar2", "franka_tanh", "planar7_128", "franka_skip"]
# tolerance named by BASELINE.json's north_star: 1e-5 relative fp32 on modulated velocities
RTOL = 1e-5


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def weights_path(kind):
    return os.path.join(GOLDEN, "weights", kind + ".npz")


OWN = 0.0   # floor of distance / velocity comparisons: the tensor's own largest magnitude is the scale


def rel_err(a, b, floor=1.0):
    """max |a-b| / max(floor, max|b|), NaN-pattern must agree.  The default floor 1.0 is the natural scale of joint angles
    (radians), unit normals, dot products, activations and RBF values; DISTANCE and VELOCITY tensors pass ``floor=OWN``: Franka
    distances are <= 0.5 m and modulated velocities often <= 0.1, so that "1e-5" means 1e-5 of the values compared, not 1e-5
    absolute."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    na, nb = np.isnan(a), np.isnan(b)
    assert (na == nb).all(), f"NaN pattern differs ({na.sum()} vs {nb.sum()})"
    if a.size == 0:
        return 0.0
    d = np.abs(np.where(na, 0, a) - np.where(nb, 0, b))
    scale = max(floor, float(np.abs(np.where(nb, 0, b)).max()), 1e-30)
    return float(d.max()) / scale


def assert_close(a, b, tol, what, floor=1.0):
    e = rel_err(a, b, floor)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e


DIST_ULP = 5e-7     # x max(1, largest network distance of the step): admissible difference between two fp32 evaluations of
                    # the distance network (tools/studies/accuracy_study.py, profiles/r02_distance_accuracy.txt: the reference's
                    # own distance is this far from the float64 value of the same network)


def velocity_envelope(q, qf, d_raw, grads, mu, sg, al, prm, delta):
    """Per component [lo, hi] of the modulated velocity when the network distance moves within +-delta, for each of the given
    blended gradients / normals (only their direction enters the step)."""
    from oracle import omds_oracle as orc
    us = [orc.modulation_step(q, qf, (d_raw + np.float32(s * delta)).astype(np.float32), g, mu, sg, al, prm)["u"]
          for s in (-1.0, -0.5, 0.0, 0.5, 1.0) for g in grads]
    us = np.stack(us)
    return us.min(axis=0), us.max(axis=0)


def assert_velocity_in_envelope(u_dev, q, qf, d_raw, grads, mu, sg, al, prm, d_scale, what, pad=0.0, family=None):
    """The device's modulated velocity must lie, per component, in the interval the oracle's modulation spans when the network
    distance moves by +-DIST_ULP * max(1, d_scale) -- MPPI.py:149-155 multiplies a distance difference by sigmoid slopes of up
    to 100, so two valid fp32 evaluations of the network differ by more than 1e-5 in the velocity near an obstacle -- for each
    of the given normals (the oracle's and the device's own, equal to 2e-5), widened by RTOL x the velocity scale (+ pad, for
    velocities recovered as (q_next - q) / dt).  Returns the largest excess over the un-widened envelope."""
    delta = DIST_ULP * max(1.0, float(d_scale))
    lo, hi = velocity_envelope(q, qf, d_raw, grads, mu, sg, al, prm, delta)
    uscale = max(1.0, float(np.abs(hi).max()))
    tol = RTOL * uscale + pad
    u_dev = np.asarray(u_dev)
    excess = float(np.maximum(np.maximum(lo - u_dev, u_dev - hi), 0).max()) if u_dev.size else 0.0
    if family is not None and u_dev.size:      # the plain bar beside it: the same rows against the oracle's own step, no envelope
        from oracle import omds_oracle as orc
        u_orc = orc.modulation_step(q, qf, np.asarray(d_raw, np.float32), grads[0], mu, sg, al, prm)["u"]
        in_env = ((u_dev >= lo - tol) & (u_dev <= hi + tol)).all(axis=-1)
        log_plain_bar(family, what, "oracle", plain_bar(u_dev, u_orc, in_env)[0])
    assert excess <= tol, f"{what}: modulated velocity {excess:.3e} outside the +-{delta:.1e} distance envelope (allowed {tol:.1e})"
    return excess


def seds_of(fx):
    """The SEDS nominal DS of a scenario fixture as oracle.Params.seds / Engine.set_ds_seds keyword arguments, or None."""
    if "seds_mu_in" not in fx:
        return None
    return dict(mu_in=fx["seds_mu_in"], b=fx["seds_b"], sigma_inv=fx["seds_sigma_inv"], A=fx["seds_A"], prior=fx["seds_prior"],
                den=fx["seds_den"], lin_thr=float(fx["seds_lin_thr"]), seds_thr=float(fx["seds_thr"]))


# ---- the plain north-star bar -------------------------------------------------------------------------------------------------
# BASELINE.json: "matching reference modulated velocities within 1e-5 rel-fp32".  The parity tests hold every row to that bar through
# two admissions (the +-DIST_ULP distance envelope; for rows with a hidden pre-activation within 5e-6 of zero, any admissible ReLU-mask
# assignment).  plain_bar() counts how many rows need NEITHER: |u_device - u_reference| <= 1e-5 x the velocity scale of the batch,
# no envelope, no alternatives.  Every GPU test that compares velocities files its counts here; test_plain_bar_summary (the last test
# of tests/test_gpu_parity.py) prints the table per fixture family and asserts floors on the plain fraction.
PLAIN_LOG = os.path.join(ROOT, "gpurun_out", "parity_plain_bar.jsonl")


def plain_bar(u_dev, u_ref, in_envelope=None):
    """Per-row classification of a velocity comparison: 'plain' = max_j |u_dev - u_ref| <= RTOL x max(|u_ref| over the batch) (the
    tensor's own scale, never clamped up to 1); of the others 'envelope' = inside the single-assignment +-DIST_ULP envelope
    (``in_envelope`` [rows] bool, from velocity_envelope), 'mask' = the rest (they pass only under another admissible ReLU-mask
    assignment -- the calling test asserts that they do).  Returns (counts dict, per-row error / scale)."""
    u_dev, u_ref = np.asarray(u_dev, np.float64), np.asarray(u_ref, np.float64)
    if u_dev.size == 0:
        return dict(rows=0, plain=0, envelope=0, mask=0, worst_plain=0.0, worst=0.0), np.zeros(0)
    scale = max(float(np.abs(np.nan_to_num(u_ref)).max()), 1e-30)
    e = np.abs(np.nan_to_num(u_dev) - np.nan_to_num(u_ref)).max(axis=-1) / scale
    plain = e <= RTOL
    env = ~plain & (np.ones_like(plain) if in_envelope is None else np.asarray(in_envelope, bool))
    return dict(rows=int(e.size), plain=int(plain.sum()), envelope=int(env.sum()), mask=int((~plain & ~env).sum()),
                worst_plain=float(e[plain].max()) if plain.any() else 0.0, worst=float(e.max())), e


def log_plain_bar(family, what, against, counts):
    """Appends one record to PLAIN_LOG (scratch; copied to profiles/ by hand) -- family = fixture family, what = the test,
    against = 'reference' (the fixture's own velocity) or 'oracle'."""
    import json
    try:
        os.makedirs(os.path.dirname(PLAIN_LOG), exist_ok=True)
        with open(PLAIN_LOG, "a") as f:
            f.write(json.dumps(dict(counts, family=family, what=what, against=against)) + "\n")
    except OSError:
        pass
