"""Host code of the library under AddressSanitizer + UBSan ("-fsanitize=address host build"), on the CPU.

`make asan` instruments the translation units that hold the host logic -- capi.hip (argument validation, the MFMA pack builders of
omds_set_mlp(_ex), omds_apply_update), comm.hip (the RCCL loader) and train.hip (the trainer's host side); device code is the release
build (GPU sanitizers are not available on this pool).  A child process with the sanitizer runtime preloaded drives everything
that needs no GPU: the update arithmetic, the pack builders on every network layout the fixtures hold (256x4 ReLU, 128-wide, skip
concatenation, tanh 256x3, the toy networks' planar inputs) and their error paths, context creation failing without a device,
and the RCCL-missing path.  Any sanitizer report fails the test; the packs' checksums must equal the uninstrumented library's."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

CSRC = os.path.join(ROOT, "optimalmodulationds_amd", "csrc")
ASAN_LIB = os.path.join(CSRC, "libomds_hip_asan.so")

CHILD = r'''
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, ROOT)
from optimalmodulationds_amd import _lib
lib = _lib.load(os.environ["OMDS_ASAN_LIB"], _lib.TEST_HOOK_SIGNATURES)
out = {}

# ---- omds_apply_update: pure host arithmetic on the reduced buffer (K = 0, 1, 50; NaN sums; masks)
rng = np.random.RandomState(0)
for K, n, H in ((0, 7, 4), (1, 2, 16), (50, 7, 32)):
    rs = 1 + K * (2 * n + 3) + n + 1 + n
    red = rng.uniform(0.1, 1.0, rs).astype(np.float32)
    if K:
        red[1 + 3] = np.nan
    mu = rng.standard_normal((K, n)).astype(np.float32); sg = np.ones(K, np.float32); al = rng.standard_normal((K, n)).astype(np.float32)
    mask = np.zeros(K, np.int32)
    for variant in (0, 2):
        rc = lib.omds_apply_update(K, n, H, red.ctypes.data, 64.0, 0.1, 1e-3, variant, mu.ctypes.data if K else None,
                                   sg.ctypes.data if K else None, al.ctypes.data if K else None, mask.ctypes.data if K else None)
        assert rc == 0, rc
assert lib.omds_apply_update(3, 7, 4, None, 64.0, 0.1, 1e-3, 0, None, None, None, None) == 1      # invalid argument, no crash
out["apply_update"] = "ok"

# ---- pack builders on every layout of the fixtures
def packs(kind, n_dof, act, mutate=None):
    z = np.load(os.path.join(ROOT, "tests", "golden", "weights", kind + ".npz"))
    nl = len([k for k in z.files if k.startswith("W")])
    W = [np.ascontiguousarray(z[f"W{i}"], dtype=np.float32) for i in range(nl)]
    b = [np.ascontiguousarray(z[f"b{i}"], dtype=np.float32) for i in range(nl)]
    skips = np.ascontiguousarray(z["skip_after"], dtype=np.int32) if "skip_after" in z.files else np.zeros(0, np.int32)
    ins = np.array([w.shape[1] for w in W], np.int32); outs = np.array([w.shape[0] for w in W], np.int32)
    if mutate:
        mutate(ins, outs, skips)
    Wp = (C.c_void_p * nl)(*[w.ctypes.data for w in W]); bp = (C.c_void_p * nl)(*[x.ctypes.data for x in b])
    cs, nb = C.c_uint64(), C.c_int64()
    rc = lib.omds_test_pack_mlp(n_dof, nl, ins.ctypes.data, outs.ctypes.data, Wp, bp, act, 100.0 if outs[-1] == 9 else 1.0,
                                int(skips.size), skips.ctypes.data if skips.size else None, C.byref(cs), C.byref(nb))
    return rc, cs.value, nb.value
for kind, n_dof, act in (("franka", 7, 0), ("planar7", 7, 0), ("planar2", 2, 0), ("planar7_128", 7, 0), ("franka_skip", 7, 0),
                         ("franka_tanh", 7, 1), ("toy2", 2, 0)):
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "weights", kind + ".npz")):
        continue
    rc, cs, nb = packs(kind, n_dof, act)
    assert rc == 0, (kind, rc, lib.omds_last_error(None))
    out["pack_" + kind] = [cs, nb]
# error paths: every one must come back as a status code with a message, not as a sanitizer report
def widen(ins, outs, skips): outs[1] = 300; ins[2] = 300
def wrong_in(ins, outs, skips): ins[0] = 31
def bad_chain(ins, outs, skips): ins[2] = 255
errs = {}
for name, (kind, n_dof, mut) in {"too_wide": ("franka", 7, widen), "bad_input_width": ("franka", 7, wrong_in), "bad_chain": ("franka", 7, bad_chain),
                                 "wrong_dof": ("franka", 6, None)}.items():
    rc, _, _ = packs(kind, n_dof, 0, mut)
    assert rc in (1, 5), (name, rc)
    errs[name] = [rc, lib.omds_last_error(None).decode()]
rc = lib.omds_test_pack_mlp(7, 5, None, None, None, None, 0, 1.0, 0, None, None, None)
assert rc == 1
out["pack_errors"] = errs

# ---- no device: creation fails loudly and leaves nothing behind; the trainer likewise
cfg = _lib.OmdsConfig(7, 64, 4, 50, 64, 5, 0, 0)
h = C.c_void_p()
rc = lib.omds_create(C.byref(cfg), C.byref(h))
out["create_rc"] = rc
out["create_err"] = lib.omds_last_error(None).decode()
bad = _lib.OmdsConfig(9, 64, 4, 50, 64, 5, 0, 0)
assert lib.omds_create(C.byref(bad), C.byref(h)) == 1
p = _lib.OmdsParams(); lib.omds_default_params(C.byref(p)); lib.omds_default_params(None)
dims = np.array([15, 64, 2], np.int32); tr = C.c_void_p()
out["trainer_rc"] = lib.omds_trainer_create(0, 2, dims.ctypes.data, 0, C.byref(tr))

# ---- RCCL missing: the loader's failure path (dlopen of a path that does not exist), twice (the cached result)
buf = (C.c_uint8 * 128)()
for _ in range(2):
    rc = lib.omds_comm_unique_id(buf)
    out["rccl_rc"] = rc
    out["rccl_err"] = (lib.omds_comm_last_error() or b"").decode()
assert lib.omds_comm_unique_id(None) != 0
print("RESULT " + json.dumps(out))
'''


def _runtime():
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        return None
    p = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_code_is_clean_under_asan_and_ubsan():
    rt = _runtime()
    if rt is None:
        pytest.skip("the ROCm clang sanitizer runtime is not in this image")
    r = subprocess.run(["make", "-C", CSRC, "-j4", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", OMDS_ASAN_LIB=ASAN_LIB,
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMDS_RCCL_LIB="/nonexistent/librccl.so", HIP_VISIBLE_DEVICES="-1",
               ROCR_VISIBLE_DEVICES="", OMDS_NO_AUTOBUILD="1")
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, r.stdout[-2000:]
    out = json.loads(line[0][7:])
    assert out["create_rc"] == 2 and "no CPU fallback" in out["create_err"]          # OMDS_ERR_HIP, loudly
    assert out["trainer_rc"] != 0
    assert out["rccl_rc"] == 3 and "OMDS_RCCL_LIB" in out["rccl_err"]                  # OMDS_ERR_RCCL
    for name, (rc, msg) in out["pack_errors"].items():
        assert msg, name
    # the same packs, bit for bit, from the uninstrumented test library: nothing in the builders depends on undefined behaviour
    import ctypes as C
    import numpy as np
    from optimalmodulationds_amd import _lib
    ref = _lib.load_test_hooks()
    kinds = [k[5:] for k in out if k.startswith("pack_") and k != "pack_errors"]
    assert {"franka", "planar7_128", "franka_skip", "franka_tanh"} <= set(kinds)
    for kind in kinds:
        z = np.load(os.path.join(ROOT, "tests", "golden", "weights", kind + ".npz"))
        nl = len([k for k in z.files if k.startswith("W")])
        W = [np.ascontiguousarray(z[f"W{i}"], dtype=np.float32) for i in range(nl)]
        b = [np.ascontiguousarray(z[f"b{i}"], dtype=np.float32) for i in range(nl)]
        skips = np.ascontiguousarray(z["skip_after"], dtype=np.int32) if "skip_after" in z.files else np.zeros(0, np.int32)
        ins = np.array([w.shape[1] for w in W], np.int32)
        outs = np.array([w.shape[0] for w in W], np.int32)
        Wp = (C.c_void_p * nl)(*[w.ctypes.data for w in W])
        bp = (C.c_void_p * nl)(*[x.ctypes.data for x in b])
        cs, nb = C.c_uint64(), C.c_int64()
        n_dof = int(ins[0]) // 3 - (2 if kind.startswith("toy") else 3)
        rc = ref.omds_test_pack_mlp(n_dof, nl, ins.ctypes.data, outs.ctypes.data, Wp, bp, 1 if kind.endswith("tanh") else 0,
                                    100.0 if outs[-1] == 9 else 1.0, int(skips.size), skips.ctypes.data if skips.size else None,
                                    C.byref(cs), C.byref(nb))
        assert rc == 0 and [cs.value, nb.value] == out["pack_" + kind], kind
