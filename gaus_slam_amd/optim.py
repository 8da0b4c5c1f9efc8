"""Gaussian SoA + fused dense Adam.

The reference keeps five `nn.Parameter`s (`_xyz, _opacity, _scaling, _rotation, _rgb`, scene/Gaussians.py:106-114)
and steps them with `torch.optim.Adam(l, lr=0.0, eps=1e-15)` using one learning rate per group (`:121-137`); prune /
densify rebuild the parameters AND their moments (`prune_optimizer :143-160`, `cat_tensors_to_optimizer :162-184`).

Here the five tensors are [P,k] views of ONE flat fp32 buffer in the all-reduce bucket layout of `ba_shard.GradBucket`
(xyz 3 | opacity 1 | scaling 2 | rotation 4 | rgb 3), with flat `exp_avg` / `exp_avg_sq` beside it, and a step is one
HIP launch (`gs2d_adam_step`, csrc/gs2d_adam.hip).  The reduced gradient bucket feeds the step directly.
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib
from .rasterizer import _on_device, _stream_ptr
from .ba_shard import BUCKET_FIELDS, BUCKET_FLOATS

# reference name (optimizer group "name", scene/Gaussians.py:124-135) for each bucket field
GROUP_NAMES = OrderedDict([("means3D", "xyz"), ("opacities", "opacity"), ("scales", "scaling"), ("rotations", "rotation"),
                           ("colors", "rgb")])


def _views(flat, P):
    out, o = OrderedDict(), 0
    for name, k in BUCKET_FIELDS.items():
        out[name] = flat[o:o + k * P].view(P, k)
        o += k * P
    return out


class GaussianSoA:
    """Flat [13*P] parameter buffer with per-field [P,k] views (leaf tensors that share storage with `flat`)."""

    generation = 0  # bumped whenever the flat buffer is re-allocated (prune / cat): older leaves() are then stale

    def __init__(self, fields):
        self.generation = self.generation + 1
        P = fields["means3D"].shape[0]
        dev = fields["means3D"].device
        self.P = P
        self.flat = torch.empty(BUCKET_FLOATS * P, dtype=torch.float32, device=dev)
        self.views = _views(self.flat, P)
        for name, v in self.views.items():
            v.copy_(fields[name].detach().reshape(v.shape))

    def leaves(self, requires_grad=True):
        """Per-field autograd leaves aliasing the flat buffer (what the rasterizer is called with).  They are valid until
        the next prune / cat, which re-allocates the buffer: fetch fresh leaves (and rebuild whatever holds the old ones,
        e.g. a KeyframeShardedBA) afterwards -- `assert_current` tells the two apart."""
        out = OrderedDict((n, v.detach().requires_grad_(requires_grad)) for n, v in self.views.items())
        for t in out.values():
            t._gs2d_generation = self.generation
        return out

    def assert_current(self, leaves):
        """Raise if any tensor of `leaves` (a dict from leaves()) no longer aliases this buffer."""
        for n, t in leaves.items():
            v = self.views[n]
            if getattr(t, "_gs2d_generation", None) != self.generation or t.data_ptr() != v.data_ptr() or t.shape != v.shape:
                raise RuntimeError(f"stale Gaussian leaf {n!r}: the SoA was re-allocated by prune()/cat(); call leaves() again")


class FusedGaussianAdam:
    """Adam over a GaussianSoA: `lrs` maps reference group names (xyz, opacity, scaling, rotation, rgb) or bucket field
    names to learning rates (configs/*/config*.py `training_args`: `<name>_lr`)."""

    def __init__(self, soa, lrs, betas=(0.9, 0.999), eps=1e-15):
        self.soa = soa
        self.betas, self.eps = betas, eps
        self.lr = [float(lrs.get(GROUP_NAMES[f], lrs.get(f, 0.0))) for f in BUCKET_FIELDS]
        self.exp_avg = torch.zeros_like(soa.flat)
        self.exp_avg_sq = torch.zeros_like(soa.flat)
        self.step_count = 0

    def _group_end(self):
        ends, o = [], 0
        for k in BUCKET_FIELDS.values():
            o += k * self.soa.P
            ends.append(o)
        return ends

    def step(self, grad_flat, leaves=None):
        """grad_flat: [13*P] fp32 in bucket layout (e.g. `GradBucket.flat` after the all-reduce).
        leaves (optional): the dict the gradients were rendered from; a stale one (older than the last prune / cat) raises
        instead of silently updating parameters nobody renders from."""
        soa = self.soa
        if leaves is not None:
            soa.assert_current(leaves)
        if grad_flat.numel() != soa.flat.numel() or grad_flat.dtype != torch.float32 or not grad_flat.is_contiguous():
            raise RuntimeError("grad_flat must be a contiguous fp32 [13*P] tensor in bucket layout")
        if not soa.flat.is_cuda or grad_flat.device != soa.flat.device:
            raise RuntimeError("FusedGaussianAdam needs CUDA tensors on one device (no CPU fallback)")
        self.step_count += 1
        n = len(BUCKET_FIELDS)
        ends = (C.c_ulonglong * n)(*self._group_end())
        lrs = (C.c_float * n)(*self.lr)
        with _on_device(soa.flat.device):
            rc = _lib.lib().gs2d_adam_step(n, ends, lrs, self.betas[0], self.betas[1], self.eps, self.step_count,
                                           soa.flat.numel(), soa.flat.data_ptr(), grad_flat.data_ptr(), self.exp_avg.data_ptr(),
                                           self.exp_avg_sq.data_ptr(), _stream_ptr(soa.flat.device))
        if rc != 0:
            raise RuntimeError("gs2d_adam_step failed (bad group table or misaligned buffers)")

    # -- topology changes keep parameters and moments aligned (scene/Gaussians.py:143-184) --------------------------
    def _rebuild(self, new_fields, new_m, new_v):
        self.soa.__init__(new_fields)
        P = self.soa.P
        self.exp_avg = torch.empty_like(self.soa.flat)
        self.exp_avg_sq = torch.empty_like(self.soa.flat)
        for (name, mv), (_, vv) in zip(_views(self.exp_avg, P).items(), _views(self.exp_avg_sq, P).items()):
            mv.copy_(new_m[name]); vv.copy_(new_v[name])

    def prune(self, keep_mask):
        """Keep rows where keep_mask is True, moments included (prune_optimizer, scene/Gaussians.py:143-160)."""
        P = self.soa.P
        m, v = _views(self.exp_avg, P), _views(self.exp_avg_sq, P)
        self._rebuild({n: t[keep_mask] for n, t in self.soa.views.items()}, {n: t[keep_mask] for n, t in m.items()},
                      {n: t[keep_mask] for n, t in v.items()})

    def cat(self, new_fields):
        """Append Gaussians with zero moments (cat_tensors_to_optimizer, scene/Gaussians.py:162-184)."""
        P = self.soa.P
        m, v = _views(self.exp_avg, P), _views(self.exp_avg_sq, P)
        z = {n: torch.zeros_like(new_fields[n].reshape(-1, k)) for n, k in BUCKET_FIELDS.items()}
        self._rebuild({n: torch.cat([t, new_fields[n].reshape(-1, t.shape[1])]) for n, t in self.soa.views.items()},
                      {n: torch.cat([t, z[n]]) for n, t in m.items()}, {n: torch.cat([t, z[n]]) for n, t in v.items()})
