"""CPU-only checks of the C-ABI boundary: the in-tree library loads, exports every symbol that
include/omds.h declares, the ctypes binding covers them all, the context-free host entry points
work, and context creation fails LOUDLY without a GPU (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import ROOT, SCENARIOS, assert_close, load
from oracle import omds_oracle as orc


def _declared(header="omds.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"OMDS_API\s+[\w\s\*]+?\b(omds_\w+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from optimalmodulationds_amd import _lib
    return _lib.load()


def test_header_symbols_exported_and_bound(lib):
    from optimalmodulationds_amd import _lib
    names = _declared()
    assert len(names) >= 25
    raw = C.CDLL(_lib.LIB_PATH)
    for nme in names:
        assert hasattr(raw, nme), f"{nme} declared in omds.h but not exported"
        assert nme in _lib.SIGNATURES, f"{nme} declared in omds.h but missing from the ctypes binding"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.omds_version() >= 100


def test_test_hooks_live_in_the_test_library_only(lib):
    """include/omds_test.h: the hooks that damage the screening inputs / force a tile shape are exported by
    libomds_hip_test.so and NOT by the product library; and the product library reads no experiment environment variable
    (only OMDS_SCREEN, OMDS_ROCTX, OMDS_RCCL_LIB appear among its strings)."""
    from optimalmodulationds_amd import _lib
    hooks = _declared("omds_test.h")
    assert hooks == sorted(_lib.TEST_HOOK_SIGNATURES) and len(hooks) == 4
    raw, raw_test = C.CDLL(_lib.LIB_PATH), C.CDLL(_lib.TEST_LIB_PATH)
    for nme in hooks:
        assert not hasattr(raw, nme), f"{nme} is a test hook and must not be exported by the product library"
        assert hasattr(raw_test, nme)
    for nme in _declared():
        assert hasattr(raw_test, nme)
    bound = _lib.load_test_hooks()
    assert bound.omds_version() == lib.omds_version()
    env_names = set(re.findall(rb"OMDS_[A-Z0-9_]+", open(_lib.LIB_PATH, "rb").read()))
    knobs = {e for e in env_names if not re.match(rb"OMDS_(ERR|ACT|MAX|WIDTH|FROW|CPAD|FLAG|COST|VARIANT|OK|HIP|API|H$|TEST|EXP|DBG|SWEEP_HIST)", e)}
    assert knobs <= {b"OMDS_SCREEN", b"OMDS_ROCTX", b"OMDS_RCCL_LIB"}, knobs


def test_default_params_are_the_reference_constants(lib):
    from optimalmodulationds_amd import _lib
    p = _lib.default_params()
    assert list(p.lvel) == [0, 1, -1, 0, 10]                                   # MPPI.py:132
    assert np.allclose(list(p.ln), [0, 1, 0, 0.1, 100])                        # MPPI.py:149-153
    assert np.allclose(list(p.ltau), [5, 1, 0, 0.1, 100])                      # MPPI.py:155
    assert (p.dst_thr, p.goal_act_cut, p.norm_clamp) == (0.5, 0.5, 0.5)
    assert np.allclose([p.lin_thr, p.coll_slow, p.coll_repulse, p.softmax_k, p.rbf_p], [0.015, 0.1, 0.1, -10, 2])


def test_create_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from optimalmodulationds_amd import _lib
    from optimalmodulationds_amd.engine import Engine
    with pytest.raises(_lib.OmdsError, match="no CPU fallback"):
        Engine(7, 16, 2, 1, max_obs=8)
    # the reference-shaped facade fails the same way (no silent fallback anywhere)
    from optimalmodulationds_amd import MPPI, LinDS, RobotSdfCollisionNet, scenes
    nn = RobotSdfCollisionNet(10, 9, [], [256] * 4)
    with pytest.raises(_lib.OmdsError):
        MPPI(scenes.FRANKA_Q0, scenes.FRANKA_QF, scenes.franka_dh_params(), scenes.shelf_scene(), 0.5, 4, 16,
             [LinDS(scenes.FRANKA_QF)], None, nn, 5)


def _packed_from_oracle(fx, pre, lo, hi, include0):
    """Partial-sum buffer of rollouts [lo, hi) in the layout of omds_local_sums, via the oracle."""
    from optimalmodulationds_amd.engine import red_layout
    cost = fx[pre + "cost"]
    K, n = int(fx["K"]), fx["q0"].shape[0]
    beta = np.float32(cost.mean(dtype=np.float32) / np.float32(50))
    w = np.exp(np.float32(-1) / beta * cost[lo:hi]).astype(np.float32)
    lay = red_layout(K, n)
    red = np.zeros(lay["size"], np.float32)
    red[0] = w.sum()
    red[lay["mu"]:lay["sigma"]] = (w[:, None, None] * fx[pre + "mu_tmp"][lo:hi]).sum(0).reshape(-1)
    red[lay["sigma"]:lay["alpha"]] = (w[:, None] * fx[pre + "sigma_tmp"][lo:hi]).sum(0)
    red[lay["alpha"]:lay["maxact"]] = (w[:, None, None] * fx[pre + "alpha_tmp"][lo:hi]).sum(0).reshape(-1)
    with np.errstate(invalid="ignore"):
        prod = fx[pre + "kernel_val_all"][lo:hi] * fx[pre + "kernel_activations"][lo:hi, :, None]
        mx = np.where(np.isnan(prod).any(axis=1), np.nan, np.nanmax(np.where(np.isnan(prod), -np.inf, prod), axis=1))
    red[lay["maxact"]:lay["phi0"]] = mx.sum(0)
    if include0:
        red[lay["phi0"]:lay["qdot"]] = fx[pre + "kernel_val_all"][0].sum(0)
    red[lay["qdot"]:lay["best"]] = (w[:, None] * fx[pre + "qdot"][lo:hi]).sum(0)
    b = int(np.argmin(cost[lo:hi]))
    red[lay["best"]] = cost[lo + b]
    red[lay["best"] + 1:] = fx[pre + "qdot"][lo + b]
    return red


@pytest.mark.parametrize("name", SCENARIOS)
def test_apply_update_matches_reference(lib, name):
    """omds_apply_update (host C) on oracle-built partial sums reproduces the reference's
    shift_policy_means outputs captured in the fixtures."""
    from optimalmodulationds_amd.engine import apply_update, red_layout
    fx = load(name)
    K, n, H, N = int(fx["K"]), fx["q0"].shape[0], int(fx["H"]), int(fx["N"])
    for it in range(int(fx["n_iter"])):
        pre = f"it{it}_"
        red = _packed_from_oracle(fx, pre, 0, N, True)
        mu, sg, al, mask = apply_update(K, n, H, red, float(N), float(fx["policy_upd_rate"]), float(fx["ker_thr"]),
                                        fx[pre + "mu_c"], fx[pre + "sigma_c"], fx[pre + "alpha_c"])
        assert int(mask.sum()) == int(fx[pre + "n_updated"])
        assert_close(mu, fx[pre + "mu_c_new"], 1e-5, "mu_c")
        assert_close(sg, fx[pre + "sigma_c_new"], 1e-5, "sigma_c")
        assert_close(al, fx[pre + "alpha_c_new"], 1e-5, "alpha_c")
        lay = red_layout(K, n)
        assert_close(red[lay["qdot"]:lay["best"]] / red[0], fx[pre + "qdot_weighted"], 1e-5, "weighted qdot")


def test_fk_num_mirror_matches_reference_vectors():
    """optimalmodulationds_amd.fk_num (host helper of the drivers' visualisation payloads) against vectors captured
    from the reference's numeric_fk_model(_vec) (tools/make_golden.py fk_vectors), and the oracle's link end points
    (the FK cost's restatement) against the same vectors."""
    import torch
    from helpers import load
    from optimalmodulationds_amd.fk_num import dh_fk, numeric_fk_model, numeric_fk_model_vec
    from oracle import omds_oracle as orc
    fx = load("fk_num")
    for kind in ("franka", "planar2"):
        q, dh = torch.from_numpy(fx[kind + "_q"]), torch.from_numpy(fx[kind + "_dh"])
        links, pint = numeric_fk_model_vec(q, dh, 4)
        assert np.abs(links.numpy() - fx[kind + "_links4"]).max() < 2e-6
        assert np.abs(pint.numpy() - fx[kind + "_int4"]).max() < 1e-7
        l1, p1 = numeric_fk_model(q[0], dh, 2)
        assert np.abs(l1.numpy() - fx[kind + "_links2_q0"]).max() < 2e-6 and np.abs(p1.numpy() - fx[kind + "_int2_q0"]).max() < 1e-7
        assert len(dh_fk(q[0], dh)) == q.shape[1] + 1
        for b in range(q.shape[0]):   # last sample point of every link = the end point the FK cost compares
            assert np.abs(orc.link_endpoints(fx[kind + "_q"][b], fx[kind + "_dh"]) - fx[kind + "_links4"][b, :, -1, :]).max() < 2e-6


def test_missing_rccl_is_an_error_code_not_a_crash():
    """A host without a loadable librccl must end in OMDS_ERR_RCCL with a message (the documented 'no fallback, raises'),
    never in a crash inside the loader.  Own process: the RCCL loader is a per-process singleton."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from optimalmodulationds_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = (C.c_uint8 * 128)()\n"
        "rc = lib.omds_comm_unique_id(buf)\n"
        "msg = (lib.omds_comm_last_error() or b'').decode()\n"
        "print('RC', rc, '|', msg)\n"
        "from optimalmodulationds_amd.engine import Engine\n"
        "try:\n"
        "    Engine.comm_unique_id()\n"
        "except _lib.OmdsError as e:\n"
        "    print('RAISED', e)\n"
    )
    env = dict(os.environ, OMDS_RCCL_LIB="/nonexistent/librccl-not-here.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RC 3 |" in r.stdout and "RCCL not available" in r.stdout and "librccl-not-here" in r.stdout, r.stdout
    assert "RAISED" in r.stdout


def test_stale_library_version_is_reported(tmp_path, lib):
    """A build of another ABI version is refused with a message that says how to rebuild (not a bare AttributeError)."""
    from optimalmodulationds_amd import _lib
    assert lib.omds_version() == _lib.ABI_VERSION
    import subprocess
    import sys
    code = (
        "import sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from optimalmodulationds_amd import _lib\n"
        "_lib.ABI_VERSION = _lib.ABI_VERSION - 1\n"
        "try:\n"
        "    _lib.load()\n"
        "except _lib.OmdsError as e:\n"
        "    print('RAISED', e)\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "RAISED" in r.stdout and f"ABI version {_lib.ABI_VERSION}" in r.stdout and "rebuild" in r.stdout, r.stdout + r.stderr


def test_example_driver_accepts_the_reference_config_keys():
    """examples/franka_planner_loop.py --config: the reference's YAML keys (ds_mppi/config.yaml, read at frankaPlanner.py:20-21,
    43-88, 129).  examples/config.yaml carries every key the loop reads; in the build container the reference's own config files
    are parsed too and must define every one of those keys (so that the loop takes them unchanged)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("franka_planner_loop", os.path.join(ROOT, "examples", "franka_planner_loop.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.load_config(os.path.join(ROOT, "examples", "config.yaml"))
    for sec, keys in mod.DEFAULTS.items():
        assert set(keys) <= set(cfg[sec]), sec
    assert cfg["planner"]["n_trajectories"] == 40 and cfg["planner"]["kernel_adding_dotproduct_thr"] == -0.9
    assert mod.load_config(None) == {k: dict(v) for k, v in mod.DEFAULTS.items()}
    assert os.path.exists(mod.weights_file(cfg["collision_model"]["fname"]))
    for ref in ("config.yaml", "config_real.yaml"):
        path = os.path.join("/root/reference/python_scripts/ds_mppi", ref)
        if not os.path.exists(path):      # the GPU box has no reference tree
            continue
        import yaml
        raw = yaml.safe_load(open(path))
        for sec, keys in mod.DEFAULTS.items():
            assert set(keys) <= set(raw[sec]), (ref, sec, set(keys) - set(raw[sec]))
        got = mod.load_config(path)
        assert got["planner"]["horizon"] == raw["planner"]["horizon"] and got["general"]["q_f"] == raw["general"]["q_f"]
