"""simple_knn.distCUDA2 host wrapper: mean squared distance to the 3 nearest other points (call sites
scene/Gaussians.py:77,218 of the reference).  Compute is the HIP kernel set in csrc/sknn.hip."""
import ctypes as C

import torch

from . import _lib
from .rasterizer import _Chunk


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("points must be a CUDA tensor")
    if points.ndimension() != 2 or points.size(1) != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    pts = points.detach().float().contiguous()
    N = pts.size(0)
    out = torch.zeros((N,), dtype=torch.float32, device=pts.device)
    if N == 0:
        return out
    ws = _Chunk(pts.device)
    with torch.cuda.device(pts.device):
        rc = _lib.lib().sknn_dist2(N, pts.data_ptr(), out.data_ptr(), ws.cb, ws.user,
                                   C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream))
    ws.release()
    if rc < 0:
        raise RuntimeError("sknn_dist2 failed: " + _lib.last_error())
    return out
