"""Dev tool (CPU): what each algebraic rewrite of the HIP backward costs in rounding, against the float64 evaluation.
Runs the float32 oracle with the kernel's forms switched on one by one (oracle/gs2d_oracle_forms.c) on the 320x240 / 20k
scene of tests/test_gpu_round3.py and prints, per tensor, max-norm error / mid-magnitude max / mid-magnitude rms vs backward_f64.
usage: form_costs.py [use_sa=1] > profiles/form_costs_r04.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import gs2d_oracle as orc
from tests import util
use_sa = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
orc.set_threads(os.cpu_count() or 1)
W, H, P = 320, 240, 20000
sc = util.make_scene(P, W, H, seed=33, regime="mapping")
o = util.oracle_forward(orc, sc, use_sa=use_sa)
stable = (o["stability"] > 2e-5).reshape(H, W)
dc, da = util.make_upstream_grads(W, H, channels=(0, 1, 2, 3, 4, 5, 6))
dc, da = (dc * W * H).numpy(), (da * W * H).numpy()
dc[:, ~stable] = 0; da[:, ~stable] = 0
g64 = orc.backward_f64(o, dc, da)
keys = ["dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dtransMat", "dL_dscales", "dL_drotations"]
names = {0: "oracle order", 1: "EXP2 (exp2 of a rounded argument)", 2: "RCP (+-1 ulp reciprocals)", 4: "MERGED (one blend recurrence)",
         8: "CLOSED (closed-form opacity-map term)", 16: "CONF (mm + conf dm)", 32: "EXPAND (-dk, -dl direct)", 63: "all six (the kernel's order)",
         60: "the four algebraic rewrites (no EXP2, no RCP)", 3: "EXP2 + RCP (the hardware instructions' error bounds)"}
assert all(np.array_equal(orc.backward(o, dc, da, forms=0)[k], orc.backward(o, dc, da)[k]) for k in keys)
print(f"# 320x240 / 20k, use_sa={use_sa}: error of the float32 backward vs its float64 evaluation, by form (max-norm | mid-magnitude max | mid rms)")
print(f"{'form':52s} " + " ".join(f"{k[3:]:>28s}" for k in keys))
for f in (0, 1, 2, 4, 8, 16, 32, 3, 60, 63):
    g = orc.backward(o, dc, da, forms=f)
    cells = []
    for k in keys:
        ref = g64[k].ravel(); a = g[k].astype(np.float64).ravel()
        sel = np.abs(ref) >= 1e-3 * np.abs(ref).max()
        r = np.abs(a[sel] - ref[sel]) / np.abs(ref[sel])
        cells.append(f"{util.grad_err(a, ref):.1e}|{r.max():.1e}|{np.sqrt((r ** 2).mean()):.1e}")
    print(f"{names[f]:52s} " + " ".join(f"{c:>28s}" for c in cells))
