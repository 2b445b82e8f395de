"""Shared helpers for the parity tests: fixture loading and tolerant comparisons."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_ALL = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(p).startswith(("mlp_", "fk_", "bases_", "seds_")))
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


def rel_err(a, b, floor=1.0):
    """max |a-b| / max(floor, max|b|): relative to the tensor's scale, NaN-pattern must agree."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    na, nb = np.isnan(a), np.isnan(b)
    assert (na == nb).all(), f"NaN pattern differs ({na.sum()} vs {nb.sum()})"
    if a.size == 0:
        return 0.0
    d = np.abs(np.where(na, 0, a) - np.where(nb, 0, b))
    scale = max(floor, float(np.abs(np.where(nb, 0, b)).max()))
    return float(d.max()) / scale


def assert_close(a, b, tol, what, floor=1.0):
    e = rel_err(a, b, floor)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e


def seds_of(fx):
    """The SEDS nominal DS of a scenario fixture as oracle.Params.seds / Engine.set_ds_seds keyword arguments, or None."""
    if "seds_mu_in" not in fx:
        return None
    return dict(mu_in=fx["seds_mu_in"], b=fx["seds_b"], sigma_inv=fx["seds_sigma_inv"], A=fx["seds_A"], prior=fx["seds_prior"],
                den=fx["seds_den"], lin_thr=float(fx["seds_lin_thr"]), seds_thr=float(fx["seds_thr"]))
