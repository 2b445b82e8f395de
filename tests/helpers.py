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


def assert_velocity_plain(u_dev, q, qf, d_raw, g, mu, sg, al, prm, what, pad=0.0, family=None):
    """north_star's bar on EVERY row, nothing admitted: |u_dev - u_oracle| <= RTOL x max|u_oracle| (+ pad, for velocities recovered as
    (q_next - q) / dt, which lose ulp(q) / dt), u_oracle = the oracle's modulation step on the oracle's own network distance
    ``d_raw`` and blended gradient ``g``.  The device evaluates the network in the oracle's arithmetic (bit-identical distances
    and ReLU masks, tools/studies/device_bits_check.py), so no row needs an envelope or another mask assignment any more.
    Returns the worst row's error / scale; files the counts in the plain-bar table when ``family`` is given."""
    from oracle import omds_oracle as orc
    u_orc = orc.modulation_step(q, qf, np.asarray(d_raw, np.float32), g, mu, sg, al, prm)["u"]
    u_dev = np.asarray(u_dev)
    if u_dev.size == 0:
        return 0.0
    counts, e = plain_bar(u_dev, u_orc)
    if family is not None:
        log_plain_bar(family, what, "oracle", counts)
    scale = max(float(np.abs(np.nan_to_num(u_orc)).max()), 1e-30)
    worst = float(e.max())
    assert worst <= RTOL + pad / scale, f"{what}: modulated velocity {worst:.3e} of its scale off the oracle's (allowed {RTOL + pad / scale:.1e}), row {int(e.argmax())}"
    return worst


def seds_of(fx):
    """The SEDS nominal DS of a scenario fixture as oracle.Params.seds / Engine.set_ds_seds keyword arguments, or None."""
    if "seds_mu_in" not in fx:
        return None
    return dict(mu_in=fx["seds_mu_in"], b=fx["seds_b"], sigma_inv=fx["seds_sigma_inv"], A=fx["seds_A"], prior=fx["seds_prior"],
                den=fx["seds_den"], lin_thr=float(fx["seds_lin_thr"]), seds_thr=float(fx["seds_thr"]))


# ---- the plain north-star bar -------------------------------------------------------------------------------------------------
# BASELINE.json: "matching reference modulated velocities within 1e-5 rel-fp32".  plain_bar() counts the rows with
# |u_device - u_reference| <= 1e-5 x the velocity scale of the batch -- nothing else is admitted anywhere in the test suite.  Every
# test that compares velocities files its counts here; tests/conftest.py prints the table per fixture family at the end of a session
# (the CPU session shows the ORACLE's rows against the reference, the GPU session the device's beside them).
PLAIN_LOG = os.path.join(ROOT, "gpurun_out", "parity_plain_bar.jsonl")


def plain_bar(u_dev, u_ref, in_envelope=None):
    """Per-row count of a velocity comparison: 'plain' = max_j |u_dev - u_ref| <= RTOL x max(|u_ref| over the batch) (the tensor's
    own scale, never clamped up to 1); the others are MISSES ('mask'; 'envelope' counts those of them a caller marks through
    ``in_envelope`` and is zero everywhere since round 6 -- both keys remain for the records' format).  Returns (counts dict,
    per-row error / scale)."""
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
