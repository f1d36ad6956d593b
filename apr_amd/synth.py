"""Deterministic synthetic KITTI-shaped LiDAR scans and scan pairs.

There is no dataset on the build or GPU box, so every test and the bench use
this generator: a 64-beam spinning LiDAR
ray-cast against a ground plane plus 40 random axis-aligned boxes, with range
noise, giving ~117 k returns per frame at the default 64 x 1875 rays.  A pair
is the same scene cast from two sensor poses, so the ground-truth rigid
transform between the frames is known.

Shapes follow what the reference's data loader hands to the hot path
(`FCGF_APR/lib/complement_data_loader.py:358-361`: float32 `[N,3]` xyz per
frame; `:788-812`: voxelise at 0.3 m, unit features).
"""
from __future__ import annotations

import numpy as np

SENSOR_HEIGHT = 1.73
MAX_RANGE = 80.0


def make_scene(seed: int, n_boxes: int = 40):
    """Random static scene: ground plane z=-1.73 and `n_boxes` axis-aligned boxes."""
    rng = np.random.default_rng(seed)
    centres = np.empty((n_boxes, 2))
    k = 0
    while k < n_boxes:
        c = rng.uniform(-60.0, 60.0, size=2)
        if np.hypot(c[0], c[1]) < 6.0:
            continue
        centres[k] = c
        k += 1
    half = rng.uniform(1.0, 8.0, size=(n_boxes, 2))
    height = 2.0 * rng.uniform(0.8, 6.0, size=n_boxes)
    lo = np.concatenate([centres - half, np.full((n_boxes, 1), -SENSOR_HEIGHT)], 1)
    hi = np.concatenate([centres + half, (-SENSOR_HEIGHT + height)[:, None]], 1)
    return lo, hi


def _ray_dirs(n_beams: int, n_azimuth: int):
    elev = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))
    azim = np.linspace(0.0, 2.0 * np.pi, n_azimuth, endpoint=False)
    ce, se = np.cos(elev)[:, None], np.sin(elev)[:, None]
    d = np.stack([ce * np.cos(azim)[None, :], ce * np.sin(azim)[None, :],
                  np.broadcast_to(se, (n_beams, n_azimuth))], -1)
    return d.reshape(-1, 3)


def raycast(scene, origin, yaw, rng, n_beams=64, n_azimuth=1875, noise=0.02):
    """Cast the scan pattern from `origin` (world) with heading `yaw`.

    Returns float32 [N,3] points in the *sensor* frame, in ray order
    (beam-major), like a KITTI .bin file.
    """
    lo, hi = scene
    d_s = _ray_dirs(n_beams, n_azimuth)
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    d = d_s @ R.T
    o = np.asarray(origin, dtype=np.float64)
    t_hit = np.full(len(d), np.inf)
    # ground plane
    dz = d[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (-SENSOR_HEIGHT - o[2]) / dz
    ok = (dz < 0) & (tg > 0)
    t_hit[ok] = tg[ok]
    # boxes (slab test), a few boxes at a time to bound memory
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
    for b in range(len(lo)):
        t0 = (lo[b] - o) * inv
        t1 = (hi[b] - o) * inv
        tmin = np.minimum(t0, t1).max(1)
        tmax = np.maximum(t0, t1).min(1)
        hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0)
        t_hit = np.where(hit & (tmin < t_hit), tmin, t_hit)
    keep = t_hit < MAX_RANGE
    pts_s = d_s[keep] * t_hit[keep, None]
    pts_s = pts_s + rng.normal(0.0, noise, size=pts_s.shape)
    return pts_s.astype(np.float32)


def make_frame(seed: int, n_beams=64, n_azimuth=1875):
    """One scan from the scene origin (seed 0 -> ~117 k points)."""
    rng = np.random.default_rng(seed)
    scene = make_scene(seed)
    return raycast(scene, (0.0, 0.0, 0.0), 0.0, rng, n_beams, n_azimuth)


def make_pair(seed: int, n_beams=64, n_azimuth=1875, n_beams1=None, dist=None):
    """Two scans of one scene.  Returns (xyz0, xyz1, T) with xyz1 ~= xyz0 @ R.T + t.

    The second pose is d in U[5,20] m along +x with yaw in U[-15,15] deg
    (SURVEY 8(d)); `dist` overrides d (config 5 uses 40 m) and `n_beams1`
    the beam count of the *first* frame (30 k vs 120 k density ratio).
    """
    rng = np.random.default_rng(seed)
    scene = make_scene(seed)
    d = rng.uniform(5.0, 20.0) if dist is None else float(dist)
    yaw = np.deg2rad(rng.uniform(-15.0, 15.0))
    xyz0 = raycast(scene, (0.0, 0.0, 0.0), 0.0, rng,
                   n_beams if n_beams1 is None else n_beams1, n_azimuth)
    xyz1 = raycast(scene, (d, 0.0, 0.0), yaw, rng, n_beams, n_azimuth)
    c, s = np.cos(yaw), np.sin(yaw)
    R1 = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    T = np.eye(4)
    T[:3, :3] = R1.T
    T[:3, 3] = -R1.T @ np.array([d, 0.0, 0.0])
    return xyz0, xyz1, T


def make_small_frame(seed: int):
    """BASELINE config 1: 16 beams x 1250 azimuths = 20 k rays."""
    return make_frame(seed, n_beams=16, n_azimuth=1250)
