"""The exact zero-skip of the fp32 pass 1 (include/omds.h: omds_pass1_skip_stats): k_pass1 stores every hidden level of a tile
compacted to the units that fire in it and multiplies only those.  The SAME BITS as the dense kernel (OMDS_FLAG_DENSE_PASS1) and as
the oracle, on the data the shipped networks see and on batches from all over the input box (where other units fire)."""
import numpy as np
import pytest

from helpers import weights_path
from oracle import omds_oracle as orc

pytestmark = pytest.mark.gpu


def _engine(m, N, obs, flags=0, k=5):
    from optimalmodulationds_amd.engine import Engine
    e = Engine(m.W[0].shape[1] // 3 - 3, N, 2, k, max_obs=max(64, obs.shape[0]), flags=flags)
    e.set_mlp(m.W, m.b, act=m.act, skip_after=m.skip_after)
    e.set_obstacles(obs)
    e.params.ignored_links = 0b111 if m.W[-1].shape[0] == 9 else 0
    e.push_params()
    return e


@pytest.mark.parametrize("kind,N", [("franka", 1024), ("franka", 150), ("planar7_128", 512), ("planar7", 700), ("planar2", 1200)])
def test_compacted_pass_is_bit_identical_to_the_dense_pass(kind, N):
    from optimalmodulationds_amd import _lib as L, scenes
    m = orc.Mlp.from_npz(weights_path(kind))
    n = m.W[0].shape[1] // 3 - 3
    rng = np.random.RandomState(3)
    if kind == "franka":
        obs = scenes.shelf_scene()
        q0, qf = np.asarray(scenes.FRANKA_Q0, np.float32), np.asarray(scenes.FRANKA_QF, np.float32)
        near = (q0 + rng.rand(N, 1).astype(np.float32) * (qf - q0) + 0.3 * rng.standard_normal((N, n))).astype(np.float32)
    else:
        obs = np.concatenate([scenes.planar7_scene(4), np.c_[rng.uniform(-7, 7, (120, 2)), np.zeros(120), np.full(120, 0.5)]]).astype(np.float32)
        near = (0.8 * rng.standard_normal((N, n))).astype(np.float32)
    far = rng.uniform(-np.pi, np.pi, (N, n)).astype(np.float32)                 # all over the joint box
    wild_obs = np.c_[rng.uniform(-9, 9, (obs.shape[0], 3)), np.full(obs.shape[0], 0.1)].astype(np.float32)
    ign = [0, 1, 2] if m.W[-1].shape[0] == 9 else []
    dy, de = _engine(m, N, obs), _engine(m, N, obs, flags=L.FLAG_DENSE_PASS1)
    assert dy.pass1_skip_stats()["active"] and not de.pass1_skip_stats()["active"]
    for q, scene in ((near, obs), (far, obs), (far, wild_obs), (near, obs)):
        de.set_obstacles(scene); dy.set_obstacles(scene)
        b, c = (e.dist_grad(q, want_mindist=True, want_idx=True) for e in (de, dy))
        for y, z, what in zip(b, c, ("distance", "gradient", "pass-1 matrix", "indices")):
            assert np.array_equal(z, y), f"{kind}: {what} differs between the compacted and the dense pass"
        _, _, mo, _ = orc.distance_repulsion_nn(m, q[:64], scene, 5, ign)
        assert np.array_equal(c[2][:64], mo), f"{kind}: pass-1 matrix is not the oracle's bits"
    st = dy.pass1_skip_stats()
    print(kind, N, st)
    if N * obs.shape[0] > 32 * 128:     # (tiny batches run 16-row tiles, which are not compacted)
        assert st["tiles"] > 0 and all(0 < u <= 256 for u in st["units"]) and all(c <= 32 for c in st["chunks"])
    if kind == "franka" and N == 1024:
        assert st["units"][2] < 200 and st["units"][3] < 180, st     # the shipped network is sparse under ReLU
    dy.close(); de.close()


def test_networks_that_do_not_qualify_run_dense():
    from optimalmodulationds_amd import scenes
    for kind in ("franka_tanh", "franka_skip"):
        m = orc.Mlp.from_npz(weights_path(kind))
        e = _engine(m, 64, scenes.shelf_scene())
        assert e.pass1_skip_stats()["active"] is False
        e.close()
