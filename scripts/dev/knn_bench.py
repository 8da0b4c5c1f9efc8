"""Dev: time simple_knn.distCUDA2 (sknn_dist2) on point clouds the size the reference feeds it (one point per pixel)."""
import sys, time, torch
sys.path.insert(0, ".")
from simple_knn._C import distCUDA2
for N in (76800, 307200, 1000000):
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(N, 3, generator=g) * torch.tensor([4.0, 3.0, 5.0])).cuda()
    for _ in range(3): d = distCUDA2(pts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): d = distCUDA2(pts)
    torch.cuda.synchronize()
    print(N, "points: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3), "mean d2 %.3e" % float(d.mean()))
