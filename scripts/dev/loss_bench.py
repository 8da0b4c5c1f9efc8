"""Time the fused post-op + loss (gaus_slam_amd/loss.py) against the PyTorch formulation of the same formulas on the GPU."""
import sys, time
import torch
sys.path.insert(0, ".")
from gaus_slam_amd import loss as gl
from oracle import loss_ref  # timing comparison only (dev script)

W, H = 640, 480
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
color = torch.rand(3, H, W, generator=g).to(dev)
allmap = torch.rand(7, H, W, generator=g).to(dev); allmap[0] *= 4
gt_color = torch.rand(H, W, 3, generator=g).to(dev)
gt_depth = (0.5 + 3 * torch.rand(H, W, 1, generator=g)).to(dev)


def run(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


for mode in (0, 1):
    def fused():
        c = color.clone().requires_grad_(True); a = allmap.clone().requires_grad_(True)
        l = gl.tracking_loss(c, a, gt_color, gt_depth, 0.5, 1.0) if mode == 0 else gl.mapping_loss(c, a, gt_color, gt_depth, 0.5, 1.0, 0.1)
        l.backward()
    def plain():
        c = color.clone().requires_grad_(True); a = allmap.clone().requires_grad_(True)
        l = loss_ref.post_and_loss(c, a, gt_color, gt_depth, mode, 0.5, 1.0, 0.1)
        l.backward()
    print(f"mode {mode}: fused {run(fused):.3f} ms, pytorch {run(plain):.3f} ms (both incl. 2 clones)")
