"""Dev: time simple_knn.distCUDA2 (sknn_dist2) on point clouds the size the reference feeds it (one point per pixel):
a uniform volume and a back-projected depth map (what scene/Gaussians.py:77,218 would pass).  Prints one JSON line."""
import json, sys, time, torch
sys.path.insert(0, ".")
from simple_knn._C import distCUDA2
out = {}
for name, N in (("volume_76800", 76800), ("volume_307200", 307200), ("volume_1000000", 1000000), ("depthmap_640x480", 307200)):
    g = torch.Generator().manual_seed(0)
    if name.startswith("volume"):
        pts = (torch.rand(N, 3, generator=g) * torch.tensor([4.0, 3.0, 5.0])).cuda()
    else:
        ys, xs = torch.meshgrid(torch.arange(480.0), torch.arange(640.0), indexing="ij")
        z = 2.0 + 0.5 * torch.sin(xs / 40.0) * torch.cos(ys / 55.0) + 0.02 * torch.rand(480, 640, generator=g)
        pts = torch.stack([(xs - 319.5) / 525.0 * z, (ys - 239.5) / 525.0 * z, z], -1).reshape(-1, 3).cuda()
    for _ in range(3): d = distCUDA2(pts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): d = distCUDA2(pts)
    torch.cuda.synchronize()
    out[name] = {"ms": round((time.perf_counter() - t0) / 10 * 1e3, 3), "mean_d2": float(d.mean())}
print(json.dumps({"bench": "simple_knn.distCUDA2 (sknn_dist2)", "results": out}))
