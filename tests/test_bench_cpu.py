"""bench.py's launcher logic on a box without a GPU: it must REFUSE -- never print a line for a rank count it did not run."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

BENCH = os.path.join(ROOT, "bench.py")
CLEAN = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}


def _run(args, env=None, timeout=300):
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env or CLEAN)


def _no_result(r):
    return r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_more_ranks_than_gpus_is_refused_before_anything_runs():
    """`python bench.py --gpus N` without a launcher environment starts its own ranks -- and exits non-zero with a message when the box
    has fewer than N GPUs (64 is more than any box; on this CPU box 2 already is), instead of running one GPU and printing n_gpus: 1."""
    import __graft_entry__ as g
    g.build()
    r = _run(["--gpus", "64", "--steps", "1", "--warmup", "0"])
    assert _no_result(r) and "--gpus 64 needs 64 GPU(s)" in r.stderr and "refusing" in r.stderr, r.stdout[-1000:] + r.stderr[-2000:]
    from optimalmodulationds_amd import _lib
    if _lib.device_count() < 2:
        r = _run(["--gpus", "2"])
        assert _no_result(r) and "--gpus 2 needs 2 GPU(s)" in r.stderr, r.stdout[-1000:] + r.stderr[-2000:]
    if _lib.device_count() < 1:
        r = _run([])
        assert _no_result(r) and "needs a GPU" in r.stderr, r.stdout[-1000:] + r.stderr[-2000:]


def test_a_launcher_world_size_that_disagrees_with_gpus_is_refused():
    """The driver's form (torch.distributed.run sets RANK / WORLD_SIZE): --gpus must equal the number of ranks the launcher started;
    a 2-rank run is never filed as 8 GPUs, nor a 1-rank run as 2."""
    env = dict(CLEAN, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = _run(["--gpus", "8"], env=env)
    assert _no_result(r) and "WORLD_SIZE=2" in r.stderr and "--gpus 8" in r.stderr, r.stderr[-2000:]
    env["WORLD_SIZE"] = "1"
    r = _run(["--gpus", "2"], env=env)
    assert _no_result(r) and "WORLD_SIZE=1" in r.stderr, r.stderr[-2000:]


def test_self_launcher_relays_a_failure_of_its_ranks(monkeypatch, capsys):
    """The self-launcher with the GPU pre-check out of the way: it spawns `torch.distributed.run` with two ranks as a CHILD process;
    here every rank fails (no GPU), and the launcher must turn that into its own non-zero exit and print no result line."""
    from optimalmodulationds_amd import _lib
    if _lib.device_count() >= 1:
        pytest.skip("needs a box without a GPU: the ranks are meant to fail")
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "visible_devices", lambda: 2)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    args = type("A", (), {"gpus": 2, "share_gpu": False})()
    with pytest.raises(SystemExit) as e:
        bench.self_launch(args, ["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"])
    assert e.value.code not in (0, None) and "2-rank run failed" in str(e.value.code)
    assert not [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]


def test_physical_cores_come_from_cpuinfo_not_from_psutil():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.physical_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
