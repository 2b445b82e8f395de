"""Synthetic obstacle scenes and robot descriptions used by the parity tests and bench.

Every scene is a float32 array ``[O, 4]`` of spheres ``(x, y, z, r)`` -- the obstacle tensor
format the reference streams between its processes (reference
``python_scripts/ds_mppi/obstacleStreamer.py:84-109`` for the shelf,
``obstacleStreamerBenchmark.py:30-51`` for the cross, ``obstacleStreamer.py:28-82`` for the I-shape,
ring, wall and line, ``scripts/standalonePlanar7d.py:68-73``
and ``scripts/standalonePlanar2d.py:76-77`` for the planar robots).  The geometry is restated
here from those descriptions; nothing is imported from the reference.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


def _segment(a, b, n):
    """n spheres evenly spaced from a to b (inclusive), rows (x, y, z, r)."""
    s = np.linspace(0.0, 1.0, n, dtype=F32).reshape(-1, 1)
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    return (a + s * (b - a)).astype(F32)


def shelf_scene() -> np.ndarray:
    """The 294-sphere Franka shelf (r = 0.03 m): 6 depth slices, each with a vertical
    divider, a horizontal divider, a top wall and a bottom wall of 12 spheres."""
    r = 0.03
    n_pts = 12
    length = max(1, 2 * n_pts - 2) * r * 1.5
    z0, x0, y0 = 0.15, 0.45, 0.0
    pos_a = np.array([x0, y0, z0 + length, r], dtype=F32)
    pos_b = pos_a + np.array([length / 3, 0, 0, 0], dtype=F32)
    line = _segment(pos_a, pos_b, n_pts // 2)
    parts = [line]
    for sphere in line:
        down = _segment(sphere, sphere - np.array([0, 0, length, 0], dtype=F32), n_pts)
        left = sphere + np.array([0, -length / 2, -length / 2, 0], dtype=F32)
        right = sphere + np.array([0, length / 2, -length / 2, 0], dtype=F32)
        lr = _segment(left, right, n_pts)
        top = lr + np.array([0, 0, length / 2, 0], dtype=F32)
        bottom = lr + np.array([0, 0, -length / 2, 0], dtype=F32)
        parts += [down, lr, top, bottom]
    return np.vstack(parts).astype(F32)


def cross_scene(z_drop: float = 0.0) -> np.ndarray:
    """The 28-sphere cross of the dynamic-obstacle benchmark (r = 0.05 m), optionally lowered
    by ``z_drop`` metres (the benchmark streamer lowers it 5 cm per robot reset)."""
    r = 0.05
    n_pts = 7
    z0, x0, y0 = 0.9, 0.3, 0.0
    length = max(1, 2 * n_pts - 2) * r
    c = np.array([x0, y0, z0 - z_drop, r], dtype=F32)
    top = c + np.array([0, 0, length, 0], dtype=F32)
    bottom = c + np.array([0, 0, -length, 0], dtype=F32)
    left = c + np.array([0, -length, 0, 0], dtype=F32)
    right = c + np.array([0, length, 0, 0], dtype=F32)
    return np.vstack((_segment(top, bottom, 2 * n_pts), _segment(left, right, 2 * n_pts))).astype(F32)


def tshape_scene() -> np.ndarray:
    """The 60-sphere I-shape in front of the robot (r = 0.05 m): a top bar and a bottom bar of 20 spheres across y at
    x = 0.4 m, joined by a vertical bar of 20 (``obstacleStreamer.py:28-51``, the streamer's 'tshape')."""
    x, half_w, z_lo, height, r, n_bar = 0.4, 0.4, 0.1, 0.75, 0.05, 20
    top = _segment([x, -half_w, z_lo + height, r], [x, half_w, z_lo + height, r], n_bar)
    bottom = top - np.array([0, 0, height, 0], dtype=F32)
    middle = _segment([x, 0.0, z_lo, r], [x, 0.0, z_lo + height, r], n_bar)
    return np.vstack((top, middle, bottom)).astype(F32)


def ring_scene() -> np.ndarray:
    """21 spheres (r = 0.03 m) on a circle of radius 0.2 m in the y-z plane at x = 0.55, z = 0.6; first and last coincide,
    as the streamer's closed ``linspace(0, 2 pi, 21)`` makes them (``obstacleStreamer.py:53-64``)."""
    ang = np.linspace(0.0, 2.0 * math.pi, 21, dtype=F32)
    ring = np.zeros((21, 4), dtype=F32)
    ring[:, 0] = 0.55
    ring[:, 1] = 0.2 * np.cos(ang)
    ring[:, 2] = 0.6 + 0.2 * np.sin(ang)
    ring[:, 3] = 0.03
    return ring


def wall_scene() -> np.ndarray:
    """The streamer's 'wall' (``obstacleStreamer.py:66-82``): two spheres (r = 0.05 m) 0.1 m apart along x at z = 0.6,
    each followed by a column of two (itself again and the sphere 0.1 m below) -- 6 rows, two of them duplicates."""
    r, n_pts = 0.05, 2
    length = max(1, 2 * n_pts - 2) * r
    line = _segment([0.6, 0.0, 0.5 + length, r], [0.6 + length, 0.0, 0.5 + length, r], n_pts)
    parts = [line]
    for sphere in line:
        parts.append(_segment(sphere, sphere - np.array([0, 0, length, 0], dtype=F32), n_pts))
    return np.vstack(parts).astype(F32)


def line_scene() -> np.ndarray:
    """What the streamer sends for 'line': by the time of its main loop the name is bound to the shelf's top row -- 6
    spheres (r = 0.03 m) along x above the shelf (``obstacleStreamer.py:93-94,129-130``)."""
    return shelf_scene()[:6].copy()


def placeholder_scene(n: int = 1) -> np.ndarray:
    """The far-away placeholder sphere(s) drivers start with before the first real scene arrives
    (``obstacleStreamer.py:133-134``: one sphere of 1 cm at x = 10.3 m; ``n`` copies for ``n_closest_obs`` > 1)."""
    return np.tile(np.array([[10.3, 0.0, 0.0, 0.01]], dtype=F32), (n, 1))


STREAMED_SCENES = {"shelf": shelf_scene, "tshape": tshape_scene, "ring": ring_scene, "wall": wall_scene, "line": line_scene,
                   "cross": cross_scene}


def planar7_scene(n_extra: int = 4, seed: int = 7) -> np.ndarray:
    """Planar 7-link scene: the driver's 3 spheres + its far 'dummy' sphere, plus ``n_extra``
    seeded spheres in [-7, 7]^2 (BASELINE config 2 asks for 8 obstacles)."""
    obs = [[6, 2, 0, .5], [4., -1, 0, .5], [5, 0, 0, .5], [6, 6, 6, .1]]
    rng = np.random.RandomState(seed)
    for _ in range(n_extra):
        x, y = rng.uniform(-7, 7, size=2)
        obs.append([x, y, 0.0, 0.5])
    return np.asarray(obs, dtype=F32)


def planar2_scene(n_obs: int = 1) -> np.ndarray:
    obs = [[6.0, 0.0, 0, .5], [0.0, 4.5, 0, .5]]
    return np.asarray(obs[:n_obs], dtype=F32)


def franka_dh_params() -> np.ndarray:
    """Modified-DH table rows (d, theta, a, alpha) of the Franka Panda, 8 rows (7 joints +
    flange), as the planner builds it (reference ``frankaPlanner.py:58-61``)."""
    a = np.array([0, 0, 0, 0.0825, -0.0825, 0, 0.088, 0], dtype=F32)
    d = np.array([0.333, 0, 0.316, 0, 0.384, 0, 0, 0.107], dtype=F32)
    alpha = np.array([0, -math.pi / 2, math.pi / 2, math.pi / 2, -math.pi / 2, math.pi / 2,
                      math.pi / 2, 0], dtype=F32)
    return np.stack((d, a * 0, a, alpha), axis=1).astype(F32)


def planar_dh_params(n_dof: int, link_len: float) -> np.ndarray:
    """Planar chain: a[0] = 0, a[1:] = link length; d = theta = alpha = 0
    (reference ``standalonePlanar2d.py:67-69``)."""
    a = np.zeros(n_dof + 1, dtype=F32)
    a[1:] = link_len
    z = np.zeros_like(a)
    return np.stack((z, z, a, z), axis=1).astype(F32)


FRANKA_Q0 = np.array([-0.88, 0.38, 0.5, -1, 0.45, 1.9, 0.31], dtype=F32)      # config.yaml:30
FRANKA_QF = np.array([-1.24, 1.53, 1.22, -1.21, -0.21, 1.55, 0.08], dtype=F32)  # config.yaml:31
