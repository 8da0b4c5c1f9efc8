"""Dev: tests/test_gpu_batch.py::test_batch_equals_separate_calls (K keyframes in one batched call against K separate calls: images,
lists, ranges and pixel state bit for bit; gradients, accumulated gradients, a second backward on the same forward) on random
(K, Gaussians, image size, use_sa).  usage: fuzz_batch.py [cases=60] [seconds=300] [seed=3]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_gpu_batch import test_batch_equals_separate_calls as check  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 3)
t0 = time.time()
done = 0
for it in range(n_cases):
    if time.time() - t0 > limit:
        break
    K = int(rng.integers(2, 9))  # (the checker rebuilds the accumulated sum from frames 1..K-1)
    W, H = int(rng.integers(48, 800)), int(rng.integers(48, 560))
    P = int(rng.choice([500, 3000, 20000, 90000, 250000]))
    use_sa = bool(rng.integers(2))
    check(K, P, W, H, use_sa)
    done += 1
    print(f"case {it:3d}: K={K} P={P} {W}x{H} sa={int(use_sa)} ok", flush=True)
print(f"{done} cases in {time.time() - t0:.0f} s: all checks passed")
