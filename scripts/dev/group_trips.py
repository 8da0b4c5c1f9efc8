"""Dev tool: trips per frame of the blend loops for 4 / 8 / 16 pixel groups per quadrant and batch sizes 64 / 56 / 48 on the
bench scene (CPU only; see group_trips.c).  Output kept as profiles/group_trips_r02.txt."""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util
from oracle import gs2d_oracle as orc
orc.set_threads(os.cpu_count() or 1)
P, W, H = 500000, 640, 480
sc = util.make_scene(P, W, H, seed=0, regime="mapping")
st = util.oracle_forward(orc, sc, use_sa=True)
p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
keep = [np.ascontiguousarray(st[k]) for k in ("ranges", "point_list", "means2D", "transMats", "normal_opacity", "n_contrib")]
td = tempfile.mkdtemp()
print(f"# trips per frame, {W}x{H} / {P} Gaussians, mapping regime (kernel counters: forward 957 499 / backward 957 155 with four 4x4 groups, BS 64;")
print("# forward 716 435 with sixteen 2x2 groups, BS 64)")
for bs in (64,):
    so = os.path.join(td, f"gt{bs}.so")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", f"-DBS={bs}", os.path.join(ROOT, "scripts", "dev", "group_trips.c"), "-o", so, "-lm"])
    out = np.zeros(9)
    C.CDLL(so).trip8(W, H, *[p(a) for a in keep], p(out))
    print(f"batch {bs}: 4 groups of 4x4 {int(out[0])} | 8 groups of 4x2 {int(out[1])} | 8 groups of 2x4 {int(out[2])} | 16 groups of 2x2 {int(out[3])} | 32 groups of 2x1 {int(out[5])} | 64 single pixels {int(out[6])} | staged splats {int(out[4])} | group-level last contributor: 4x4 {int(out[7])}, 2x2 {int(out[8])}")
