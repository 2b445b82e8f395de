#!/usr/bin/env python3
"""Golden vectors for the SEDS nominal DS (ds_mppi/functions/SEDS.py) by RUNNING THE REFERENCE (container-only, like
tools/make_golden.py).  Two kinds:
  * seds_<name>.npz      : SEDS(<content/ds/file>.mat).get_velocity on seeded states, one state per call -- the reference's
                           normalisation lines (SEDS.py:69-73) only broadcast for a single state, which is also why every
                           use of SEDS inside the N-rollout planner is commented out in the reference's drivers
                           (frankaIntegrator.py:70-71); the mixture parameters (data of the .mat file) and the quantities
                           SEDS.__init__ / GMR derive from them with torch travel in the fixture;
  * franka_seds_integrator_N1.npz : the integrator shape (MPPI with N_traj = 1, H = 2, frankaIntegrator.py:73) with
                           DS_ARRAY = [SEDS(seds_left10.mat, q_f)], captured like the other scenario fixtures.
Usage:  MPLBACKEND=Agg python tools/make_golden_seds.py"""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
REF = "/root/reference/python_scripts"
sys.path[:0] = [REF + "/ds_mppi/functions", REF + "/mlp_learn"]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

import numpy as np
import torch

from SEDS import SEDS  # noqa: E402  (reference)

import make_golden as mg  # noqa: E402  (this repo's generator: scenario machinery)
from optimalmodulationds_amd import scenes  # noqa: E402
from optimalmodulationds_amd.seds import SEDS as OurSEDS  # noqa: E402  (parameter derivation only, checked below)

OUT = mg.OUT
DS_DIR = REF + "/ds_mppi/content/ds/"


def t2n(x):
    return x.detach().cpu().numpy().copy()


def seds_vectors(name, seed):
    ds = SEDS(DS_DIR + name + ".mat")
    d = ds.dof
    torch.manual_seed(seed)
    B = 160
    x = ds.q_goal.reshape(1, d) + 0.5 * torch.randn(B, d)
    x[:8] = ds.q_goal.reshape(1, d) + 3e-3 * torch.randn(8, d)          # inside lin_thr: the mixture's raw output
    x[8:40] = ds.Mu[:d, torch.randint(0, ds.n_gaussians, (32,))].t() + ds.q_goal.reshape(1, d) + 0.05 * torch.randn(32, d)   # near the components
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        y = torch.cat([ds.get_velocity(x[i:i + 1].clone()) for i in range(B)])
    ours = OurSEDS(DS_DIR + name + ".mat")
    mu_in, b, s_inv, A, prior, den = ours.device_params()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), Mu=t2n(ds.Mu), Sigma=t2n(ds.Sigma), Priors=t2n(ds.Priors), xT=t2n(ds.q_goal),
                        mu_in=mu_in, b=b, sigma_inv=s_inv, A=A, prior=prior, den=den, lin_thr=np.float32(ds.lin_thr),
                        seds_thr=np.float32(ds.seds_thr), x=t2n(x), y=t2n(y))
    print(f"{name}: dof {d}, {ds.n_gaussians} components, |y| in [{float(y.norm(dim=1).min()):.3f}, {float(y.norm(dim=1).max()):.3f}], "
          f"linear fallback on {buf.getvalue().count('lin!')} of {B} states")


def main():
    for i, name in enumerate(("seds_left10", "seds_right", "seds_sine10", "seds_2d")):
        seds_vectors(name, 40 + i)
    # integrator-shaped planner with a SEDS nominal DS
    model = mg.load_model("franka")
    qf = torch.from_numpy(np.asarray(scenes.FRANKA_QF, np.float32))
    ds = SEDS(DS_DIR + "seds_left10.mat", qf.unsqueeze(1))
    ours = OurSEDS(DS_DIR + "seds_left10.mat", qf.unsqueeze(1))
    mu_in, b, s_inv, A, prior, den = ours.device_params()
    extra = {"seds_mu_in": mu_in, "seds_b": b, "seds_sigma_inv": s_inv, "seds_A": A, "seds_prior": prior, "seds_den": den,
             "seds_lin_thr": np.float32(ds.lin_thr), "seds_thr": np.float32(ds.seds_thr)}
    mg.run_scenario("franka_seds_integrator_N1", kind="franka", nn_model=model, N=1, H=2, obs=scenes.shelf_scene(), k=5, K=5, seed=31,
                    dt=0.01, q0=scenes.FRANKA_Q0, qf=scenes.FRANKA_QF, dst_thr=0.03, ker_thr=0.1, alpha_s=0.0, sigma_nom=1.0,
                    ds_array=[ds], extra=extra)


if __name__ == "__main__":
    main()
