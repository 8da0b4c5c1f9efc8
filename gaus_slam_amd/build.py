"""Builds the gfx950 shared library (C ABI in include/gs2d_rasterizer.h) with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as well as on the GPU box.
The .so is written in-tree (gaus_slam_amd/lib/) so it travels with the source snapshot.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
# GS2D_LIB_PATH: load a pre-built variant instead (kernel experiments, scripts/dev/); never set in normal use
LIB_PATH = os.environ.get("GS2D_LIB_PATH") or os.path.join(LIB_DIR, "libgs2d_hip.so")
SOURCES = ["gs2d_preprocess.hip", "gs2d_binning.hip", "gs2d_blend.hip", "gs2d_det.hip", "gs2d_api.hip", "sknn.hip", "gs2d_loss.hip", "gs2d_adam.hip"]
# -ffp-contract=off: the per-Gaussian geometry (tile rectangles, depth keys) must be reproducible on the host.
# -fno-slp-vectorize: the SLP pass packs scalar fp32 ops into v_pk_* pairs; on gfx950 a packed op costs ~1.85 plain ones
# (scripts/dev/issue_bench.hip) and assembling the register pairs took ~90 v_mov and 8 extra spills in blend_bwd.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17"]


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from: every file under csrc/ plus the C-ABI
    header, in name order (name and content).  Compiled into the library (gs2d_build_info) and written into every JSON bench.py
    emits, so that a kept artifact says which kernels produced it (tests/test_host.py checks the profiles of the current round)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.listdir(CSRC))
    for f in files:
        h.update(f.encode() + b"\0")
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(_HERE, "..", "include", "gs2d_rasterizer.h"), "rb") as fh:
        h.update(b"gs2d_rasterizer.h\0" + fh.read())
    return h.hexdigest()[:16]


def _stale():
    if os.environ.get("GS2D_LIB_PATH"):
        return False
    if not os.path.exists(LIB_PATH):
        return True
    try:  # the hash the library was built from (sidecar written by build())
        with open(LIB_PATH + ".hash") as fh:
            if fh.read().strip() != source_hash():
                return True
    except OSError:
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "gs2d_rasterizer.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    sh = source_hash()
    cmd = [hipcc] + FLAGS + [f'-DGS2D_SOURCE_HASH="{sh}"', "-o", LIB_PATH] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(LIB_PATH + ".hash", "w") as fh:
        fh.write(sh + "\n")
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB_PATH)
