"""Off-screen rendering for `BatchedModularVecEnv.get_images()` (reference src/subproc_vec_env.py:70-73 -> every worker's
`env.render(mode='rgb_array')`, MuJoCo's OpenGL renderer through gym / mujoco-py: third-party, absent here).

A ray caster (csrc/render.hip, C ABI include/sgrl_render.h) over the morphology's own geoms -- ground plane, spheres,
capsules at the bodies' world poses -- seen through the reference's camera (`viewer_setup`, <env>.py:166-170: track body 2,
distance = half the model extent, look-at height 1.15, elevation -20 degrees; MuJoCo's default azimuth 90 and fovy 45).
Pixel parity with MuJoCo's rasteriser is neither claimed nor pinned: it shows what the engine simulates.  The scene
(forward kinematics of the requested environments) is assembled on the host from the engine's state records: this is a
visualisation path, not part of the rollout.
"""
import ctypes

import numpy as np
import torch

from . import _lib, mjcf

GEOM_FLOATS, CAM_FLOATS = 16, 13
_PALETTE = np.array([[0.8, 0.6, 0.4], [0.85, 0.45, 0.35], [0.4, 0.6, 0.85], [0.45, 0.75, 0.5], [0.8, 0.75, 0.4]])


def model_extent(m):
    """Radius-like size of the model at qpos0 (stand-in for MuJoCo's model.stat.extent): the largest distance between the
    bounding spheres of two of its non-plane geoms."""
    xpos, xquat, _, _ = mjcf.kinematics_np(m, m.qpos0)
    pts, rad = [], []
    for g in range(m.ngeom):
        if m.geom_type[g] == 0:
            continue
        b = m.geom_body[g]
        pts.append(xpos[b] + mjcf.quat_to_mat(xquat[b]) @ m.geom_pos[g])
        rad.append(m.geom_size[g][0] + (m.geom_size[g][1] if m.geom_type[g] == 3 else 0.0))
    pts, rad = np.array(pts), np.array(rad)
    d = np.linalg.norm(pts[:, None] - pts[None], axis=-1) + rad[:, None] + rad[None]
    return float(max(d.max(), 2 * rad.max()))


def scene_of(m, qpos):
    """(geom records [ngeom, 16], camera record [13]) of morphology `m` at configuration `qpos`."""
    xpos, xquat, _, _ = mjcf.kinematics_np(m, qpos)
    recs = np.zeros((m.ngeom, GEOM_FLOATS), dtype=np.float32)
    for g in range(m.ngeom):
        b = m.geom_body[g]
        rb = mjcf.quat_to_mat(xquat[b])
        pos = xpos[b] + rb @ m.geom_pos[g]
        axis = rb @ mjcf.quat_to_mat(m.geom_quat[g])[:, 2]
        t = int(m.geom_type[g])
        recs[g, 0] = t
        recs[g, 1:4] = pos
        recs[g, 4:7] = axis
        recs[g, 7] = 0.0 if t == 0 else m.geom_size[g][0]
        recs[g, 8] = m.geom_size[g][1] if t == 3 else 0.0
        recs[g, 9:12] = [0.75, 0.8, 0.7] if t == 0 else _PALETTE[b % len(_PALETTE)]
    # camera: reference viewer_setup (<env>.py:166-170) on MuJoCo's tracking camera (azimuth 90, fovy 45)
    track = min(2, m.nbody - 1)
    lookat = np.array([xpos[track][0], xpos[track][1], 1.15])
    dist, el, az = 0.5 * model_extent(m) * 2.2, np.deg2rad(-20.0), np.deg2rad(90.0)
    fwd = np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    eye = lookat - dist * fwd
    right = np.cross(fwd, [0.0, 0.0, 1.0])
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    cam = np.concatenate([eye, fwd, right, up, [np.tan(np.deg2rad(45.0) / 2)]]).astype(np.float32)
    return recs, cam


def render(scenes, width=256, height=256, device="cuda:0"):
    """scenes: list of (geom records, camera record) -> uint8 [n, height, width, 3] (a CUDA tensor)."""
    if not torch.cuda.is_available():
        raise _lib.SgrlError("rendering needs an MI355X (no CPU fallback)")
    L = _lib.lib()
    L.sgrl_render.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_void_p, ctypes.c_void_p]
    n = len(scenes)
    mg = max(s[0].shape[0] for s in scenes)
    geoms = np.zeros((n, mg, GEOM_FLOATS), dtype=np.float32)
    counts = np.zeros(n, dtype=np.int32)
    cams = np.zeros((n, CAM_FLOATS), dtype=np.float32)
    for i, (g, c) in enumerate(scenes):
        geoms[i, :g.shape[0]] = g
        counts[i] = g.shape[0]
        cams[i] = c
    dev = torch.device(device)
    gd, cd, kd = torch.from_numpy(geoms).to(dev), torch.from_numpy(cams).to(dev), torch.from_numpy(counts).to(dev)
    out = torch.empty((n, height, width, 3), dtype=torch.uint8, device=dev)
    rc = L.sgrl_render(ctypes.c_void_p(gd.data_ptr()), ctypes.c_void_p(kd.data_ptr()), mg, ctypes.c_void_p(cd.data_ptr()), n, int(width),
                       int(height), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise _lib.SgrlError("sgrl_render failed (%d)" % rc)
    return out
