"""Seeded synthetic surfel scenes + camera set-up for benches and parity tests.

The recipe follows SURVEY.md section 8(d) / BASELINE.md section 2.  `setup_camera` computes
the same camera quantities as the reference adapter render/render_2dgs.py:6-31
(tan-fov from intrinsics, OpenGL-style projection with near .01 / far 100,
transposed = column-major matrices, camera centre from inv(w2c)), but on the
CPU in float64->float32 so the oracle and the HIP op can be fed identical bits.
"""
import math
from typing import NamedTuple

import numpy as np
import torch


class Camera(NamedTuple):
    W: int
    H: int
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor  # [4,4] float32 = w2c^T (column-major w2c)
    projmatrix: torch.Tensor  # [4,4] float32 = (P @ w2c)^T
    campos: torch.Tensor      # [3]
    K: torch.Tensor           # [3,3]
    w2c: torch.Tensor         # [4,4]


def intrinsics_for(W, H):
    """TUM-like pinhole scaled from 640x480 (fx=fy=525, principal point at the centre)."""
    f = 525.0 * W / 640.0
    return torch.tensor([[f, 0.0, (W - 1) / 2.0], [0.0, f, (H - 1) / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float32)


def setup_camera(W, H, K, w2c, near=0.01, far=100.0):
    """Same math as render/render_2dgs.py:6-31, float32 torch ops on the CPU."""
    K = K.float()
    w2c = w2c.float()
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    cam_center = torch.inverse(w2c)[:3, 3]
    view = w2c.unsqueeze(0).transpose(1, 2)
    opengl_proj = torch.tensor([[2 * fx / W, 0.0, -(W - 2 * cx) / W, 0.0],
                                [0.0, 2 * fy / H, -(H - 2 * cy) / H, 0.0],
                                [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                                [0.0, 0.0, 1.0, 0.0]]).float().unsqueeze(0).transpose(1, 2)
    full_proj = view.bmm(opengl_proj)
    return Camera(W=W, H=H, tanfovx=float(W / (2 * fx)), tanfovy=float(H / (2 * fy)),
                  viewmatrix=view[0].contiguous(), projmatrix=full_proj[0].contiguous(),
                  campos=cam_center.contiguous(), K=K, w2c=w2c)


def _rotmat_to_quat_wxyz(R):
    """Batched rotation matrix -> unit quaternion (w,x,y,z), float64 numpy."""
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    q = np.empty((R.shape[0], 4))
    q[:, 0] = np.sqrt(np.maximum(0, 1 + m00 + m11 + m22)) / 2
    q[:, 1] = np.sqrt(np.maximum(0, 1 + m00 - m11 - m22)) / 2
    q[:, 2] = np.sqrt(np.maximum(0, 1 - m00 + m11 - m22)) / 2
    q[:, 3] = np.sqrt(np.maximum(0, 1 - m00 - m11 + m22)) / 2
    q[:, 1] = np.copysign(q[:, 1], R[:, 2, 1] - R[:, 1, 2])
    q[:, 2] = np.copysign(q[:, 2], R[:, 0, 2] - R[:, 2, 0])
    q[:, 3] = np.copysign(q[:, 3], R[:, 1, 0] - R[:, 0, 1])
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def random_w2c(rng, max_rot_deg=10.0, max_trans=0.3):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = math.radians(rng.uniform(0.3, 1.0) * max_rot_deg)
    Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + math.sin(ang) * Kx + (1 - math.cos(ang)) * Kx @ Kx
    t = rng.uniform(-max_trans, max_trans, size=3)
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = t
    return torch.from_numpy(M).float()


def make_scene(P, W, H, seed=0, regime="tracking", cull_frac=0.03, scale_lo=0.7, scale_hi=4.0,
               max_tilt_deg=75.0):
    """Returns dict(means3D[P,3], scales[P,2], rotations[P,4] wxyz, opacities[P,1], colors[P,3], cam).

    regime 'tracking': camera-space Gaussians + identity view (render/__init__.py:23-40);
    regime 'mapping' : world-space Gaussians + a general w2c (render/__init__.py:59-71)."""
    rng = np.random.default_rng(seed)
    K = intrinsics_for(W, H)
    f, cx, cy = float(K[0, 0]), float(K[0, 2]), float(K[1, 2])
    u = rng.uniform(-0.05, 1.05, P) * W
    v = rng.uniform(-0.05, 1.05, P) * H
    z = rng.uniform(0.5, 6.0, P)
    ncull = int(round(cull_frac * P))
    if ncull > 0:
        half = ncull // 2
        z[:half] = rng.uniform(-1.0, 0.19, half)           # behind the near plane
        u[half:ncull] = rng.uniform(1.5, 3.0, ncull - half) * W  # far off-screen
    mean_cam = np.stack([(u - cx) / f * z, (v - cy) / f * z, z], 1)
    s = np.exp(rng.uniform(math.log(scale_lo), math.log(scale_hi), (P, 2)))
    scales = (np.abs(z)[:, None] + 1e-3) / f * s
    # normal = direction towards the camera, tilted by <= max_tilt_deg
    to_cam = -mean_cam / (np.linalg.norm(mean_cam, axis=1, keepdims=True) + 1e-12)
    helper = np.where(np.abs(to_cam[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    t1 = np.cross(to_cam, helper)
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(to_cam, t1)
    tilt = np.radians(rng.uniform(0, max_tilt_deg, P))
    az = rng.uniform(0, 2 * math.pi, P)
    n = (np.cos(tilt)[:, None] * to_cam + np.sin(tilt)[:, None] * (np.cos(az)[:, None] * t1 + np.sin(az)[:, None] * t2))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    h2 = np.where(np.abs(n[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    a1 = np.cross(n, h2)
    a1 /= np.linalg.norm(a1, axis=1, keepdims=True)
    a2 = np.cross(n, a1)
    spin = rng.uniform(0, 2 * math.pi, P)
    e1 = np.cos(spin)[:, None] * a1 + np.sin(spin)[:, None] * a2
    e2 = np.cross(n, e1)
    Rm = np.stack([e1, e2, n], 2)  # columns: tangent u, tangent v, normal
    opac = np.clip(1.0 / (1.0 + np.exp(-rng.normal(0, 1.5, P))), 0.02, 0.99)
    colors = rng.uniform(0, 1, (P, 3))
    if regime == "mapping":
        w2c = random_w2c(rng)
        c2w = np.linalg.inv(w2c.double().numpy())
        means = mean_cam @ c2w[:3, :3].T + c2w[:3, 3]
        Rm = np.einsum("ij,njk->nik", c2w[:3, :3], Rm)
    else:
        w2c = torch.eye(4)
        means = mean_cam
    quat = _rotmat_to_quat_wxyz(Rm)
    cam = setup_camera(W, H, K, w2c)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
    return dict(means3D=t(means), scales=t(scales), rotations=t(quat), opacities=t(opac[:, None]),
                colors=t(colors), cam=cam)


def make_upstream_grads(W, H, seed=1, channels=(0, 1, 5, 6)):
    """N(0,1)/HW upstream gradients on color and on the allmap channels SLAM's losses
    touch (tracking: 0,1; mapping adds 6; 5 = median depth) -- SURVEY.md section 8(d)."""
    g = torch.Generator().manual_seed(seed)
    dcolor = torch.randn(3, H, W, generator=g) / (H * W)
    dall = torch.zeros(7, H, W)
    for c in channels:
        dall[c] = torch.randn(H, W, generator=g) / (H * W)
    return dcolor, dall
