"""The drop-in boundary used from plain C: examples/c_abi_planner.c includes nothing but include/omds.h, is compiled with gcc as
C99, linked against libomds_hip.so and runs one planner iteration (sample -> propagate -> cost -> update -> qdot).  Its numbers must
equal the same calls made through the ctypes binding: the boundary carries no Python / torch / C++ types."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from helpers import ROOT, weights_path
from oracle import omds_oracle as orc

CSRC = os.path.join(ROOT, "optimalmodulationds_amd", "csrc")


def _compile(tmp_path):
    exe = str(tmp_path / "c_abi_planner")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_planner.c"), "-o", exe, "-L" + CSRC, "-lomds_hip", "-Wl,-rpath," + CSRC, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_headers_are_c99_and_the_library_links_from_c(tmp_path):
    """CPU part: both headers compile as strict C99 and a C program links against the product library."""
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    import __graft_entry__ as g
    g.build()
    src = tmp_path / "hdr.c"
    src.write_text('#include "omds.h"\n#include "omds_test.h"\n'
                   'int main(void) { omds_params p; omds_default_params(&p); return (omds_version() > 0 && p.softmax_k == -10.f) ? 0 : 1; }\n')
    exe = str(tmp_path / "hdr")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe,
                        "-L" + CSRC, "-lomds_hip", "-Wl,-rpath," + CSRC], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([exe]).returncode == 0
    _compile(tmp_path)


@pytest.mark.gpu
def test_planner_iteration_from_c_equals_the_ctypes_binding(tmp_path):
    from optimalmodulationds_amd import scenes
    from optimalmodulationds_amd.cost import FRANKA_Q_MAX, FRANKA_Q_MIN
    from optimalmodulationds_amd.engine import Engine
    exe = _compile(tmp_path)
    m = orc.Mlp.from_npz(weights_path("franka"))
    obs = scenes.shelf_scene()
    q0, qf, dh = scenes.FRANKA_Q0, scenes.FRANKA_QF, scenes.franka_dh_params()
    qmin, qmax = np.array(FRANKA_Q_MIN, np.float32), np.array(FRANKA_Q_MAX, np.float32)
    N, H, K, n = 1024, 8, 6, 7
    rng = np.random.RandomState(8)
    s = (np.arange(K) + 0.5) / K
    mu_c = (q0 + s[:, None] * (qf - q0) + 0.15 * rng.standard_normal((K, n))).astype(np.float32)
    sg_c, al_c = np.ones(K, np.float32), rng.standard_normal((K, n)).astype(np.float32)
    dims = [m.W[0].shape[1]] + [w.shape[0] for w in m.W]
    blob = tmp_path / "model.bin"
    with open(blob, "wb") as f:
        f.write(struct.pack("<ii", n, len(m.W)) + struct.pack("<%di" % len(dims), *dims) + struct.pack("<ii", obs.shape[0], K))
        for w, b in zip(m.W, m.b):
            f.write(np.ascontiguousarray(w, np.float32).tobytes() + np.ascontiguousarray(b, np.float32).tobytes())
        for a in (obs, q0, qf, dh, qmin, qmax, mu_c, sg_c, al_c):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())
    r = subprocess.run([exe, str(blob), str(N), str(H)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = {l.split()[0]: [float(x) for x in l.split()[1:]] for l in r.stdout.strip().splitlines()}
    # the same calls through ctypes
    e = Engine(n, N, H, 5, max_obs=max(64, obs.shape[0]))
    e.set_mlp(m.W, m.b)
    e.set_obstacles(obs)
    e.params.dt, e.params.dst_thr, e.params.ignored_links = 0.5, 0.01, 0b111
    e.push_params()
    e.set_ds(qf)
    e.set_cost(dh, qmin, qmax)
    e.sample_policy(mu_c, sg_c, al_c, 0.0, 0.0, 3.0, K, seed=4242, rollout_offset=0)
    e.propagate(q0)
    cost = e.cost()
    mu, sg, al, mask, _ = e.weighted_update(0.1, 0.1, mu_c, sg_c, al_c)
    qd = e.get_qdot("weighted")
    e.close()
    assert int(out["version"][0]) == e.lib.omds_version()
    assert int(out["updated"][0]) == int(mask.sum())
    assert np.array_equal(np.array(out["qdot"], np.float32), qd)                       # printed with 9 significant digits: exact float32
    assert np.array_equal(np.array(out["mu0"], np.float32), mu[0])
    assert abs(out["cost_sum"][0] - float(cost.astype(np.float64).sum())) <= 1e-6 * abs(out["cost_sum"][0])
