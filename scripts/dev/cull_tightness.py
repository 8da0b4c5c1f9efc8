"""Dev: how tight is the sub-block cull?  Compares the library's 16 cull bits per instance (read back from the binning
chunk) with the exact bits (any pixel of the sub-block passes the alpha test, brute force on the CPU) on the bench scene.
Needs scripts/dev/exact_bits.so: gcc -O2 -fopenmp -shared -fPIC oracle/cull_exact.c -o scripts/dev/exact_bits.so -lm"""
import ctypes as C, sys, json
import numpy as np
sys.path.insert(0, ".")
import torch
from tests import util
from oracle import gs2d_oracle as orc
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
W, H = 640, 480
orc.set_threads(64)
sc = util.make_scene(P, W, H, seed=0, regime="mapping")
o = util.oracle_forward(orc, sc, use_sa=True)
h = util.hip_forward(sc, use_sa=True)
R = h["num_rendered"]
binning = h["buffers"][1].cpu().numpy()
al = lambda n: (n + 255) // 256 * 256
off = al(4 * R) + al(8 * R)  # BinLayout: point_list, group bits (8 B), then the row bits (4 B, one byte per quadrant)
hits = np.frombuffer(binning.tobytes()[off:off + 4 * R], dtype=np.uint32)
L = C.CDLL("scripts/dev/exact_bits.so")
ex = np.zeros(R, np.uint32)
p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
keep = [np.ascontiguousarray(o[k]) for k in ("ranges", "point_list", "means2D", "transMats", "normal_opacity")]
L.exact_bits(W, H, *[p(a) for a in keep], p(ex))
goff = al(4 * R)  # the forward's 64 group bits per instance
ghits = np.frombuffer(binning.tobytes()[goff:goff + 8 * R], dtype=np.uint64)
gex = np.zeros(R, np.uint64)
L.exact_group_bits(W, H, *[p(a) for a in keep], p(gex))
pc = lambda a: int(np.unpackbits(a.view(np.uint8)).sum())
missed = int(np.count_nonzero(ex & ~hits))
print(json.dumps({"instances": R, "exact_row_pairs": pc(ex), "cull_row_pairs": pc(hits), "looseness": round(pc(hits) / pc(ex), 4),
                  "instances_exact_nonzero": int(np.count_nonzero(ex)), "instances_cull_nonzero": int(np.count_nonzero(hits)),
                  "exact_bits_missed_by_cull": missed,
                  "group_bits": {"exact": pc(gex), "cull": pc(ghits), "looseness": round(pc(ghits) / max(pc(gex), 1), 4),
                                 "missed": int(np.count_nonzero(gex & ~ghits))}}))
