"""Dev: how many (Gaussian, tile) instances of the reference's 3-sigma square rect are provably empty, and how many of
them a tighter per-Gaussian tile rect would already drop.  Reads the cull bits of a forward (private scratch layout:
hits follow the point list).  Run on the GPU box; prints one JSON line.  usage: prune_estimate.py [P] [W] [H] [regime]"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
from gaus_slam_amd.scene_synth import make_scene

P = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 640
H = int(sys.argv[3]) if len(sys.argv) > 3 else 480
regime = sys.argv[4] if len(sys.argv) > 4 else "mapping"
sc = make_scene(P, W, H, seed=0, regime=regime)
fw = util.hip_forward(sc, debug=True)
R = fw["num_rendered"]
b = fw["buffers"][1].cpu().numpy()
off = (4 * R + 255) // 256 * 256
hits = np.frombuffer(b.tobytes()[off:off + 8 * R], dtype=np.uint64)  # 64 group bits per instance
pl, rng = fw["point_list"].astype(np.int64), fw["ranges"].astype(np.int64)
gx = (W + 15) // 16
tile = np.repeat(np.arange(len(rng)), rng[:, 1] - rng[:, 0])
tx, ty = tile % gx, tile // gx
nz = hits != 0
big = 1 << 20
lo_x = np.full(P, big); hi_x = np.full(P, -1); lo_y = np.full(P, big); hi_y = np.full(P, -1)
np.minimum.at(lo_x, pl[nz], tx[nz]); np.maximum.at(hi_x, pl[nz], tx[nz])
np.minimum.at(lo_y, pl[nz], ty[nz]); np.maximum.at(hi_y, pl[nz], ty[nz])
inside = (tx >= lo_x[pl]) & (tx <= hi_x[pl]) & (ty >= lo_y[pl]) & (ty <= hi_y[pl])
print(json.dumps({"P": P, "W": W, "H": H, "regime": regime, "instances": int(R), "empty_frac": float((~nz).mean()),
                  "dropped_by_tight_rect_frac": float((~inside).mean()),
                  "empty_left_inside_tight_rect_frac": float((inside & ~nz).mean())}))
