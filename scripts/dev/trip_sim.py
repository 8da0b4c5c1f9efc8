"""Dev tool: trip counts of the blend loop under different queue granularities on the bench scene (CPU only)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, ".")
from tests import util
from oracle import gs2d_oracle as orc
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
W, H = 640, 480
sc = util.make_scene(P, W, H, seed=0, regime="mapping")
st = util.oracle_forward(orc, sc, use_sa=True)
L = C.CDLL("/tmp/trip_sim.so")
out = np.zeros(12)
p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
keep = [np.ascontiguousarray(st[k]) for k in ("ranges", "point_list", "means2D", "transMats", "normal_opacity", "n_contrib")]
L.trip_sim(W, H, *[p(a) for a in keep], p(out))
names = ["S4 sync", "S2 sync", "S4 decoupled", "S2 decoupled", "pairs S4", "pairs S2", "passing pairs", "wave-chunks", "S2 sync128", "S8 (whole quadrant)", "S4 compacted batches (<=64 touched)", "S4 exact 64-slot batches (split chunks)"]
for n, v in zip(names, out):
    print(f"{n:22s} {v:14.0f}")
