"""Dev: one-shape driver for rocprofv3 --kernel-trace of simple_knn.distCUDA2 (depth-map cloud, 640x480)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simple_knn._C import distCUDA2
g = torch.Generator().manual_seed(0)
ys, xs = torch.meshgrid(torch.arange(480.0), torch.arange(640.0), indexing="ij")
z = 2.0 + 0.5 * torch.sin(xs / 40.0) * torch.cos(ys / 55.0) + 0.02 * torch.rand(480, 640, generator=g)
pts = torch.stack([(xs - 319.5) / 525.0 * z, (ys - 239.5) / 525.0 * z, z], -1).reshape(-1, 3).cuda()
for _ in range(5): d = distCUDA2(pts)
torch.cuda.synchronize()
