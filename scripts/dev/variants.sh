#!/bin/bash
# Dev tool: build experiment variants of the library (never the product build).  usage: scripts/dev/variants.sh name "-DFLAG ..."
set -e
cd "$(dirname "$0")/../../gaus_slam_amd/csrc"
mkdir -p ../../scripts/dev/variants
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 $2 -o ../../scripts/dev/variants/lib$1.so \
  gs2d_preprocess.hip gs2d_binning.hip gs2d_blend.hip gs2d_det.hip gs2d_api.hip sknn.hip gs2d_loss.hip gs2d_adam.hip 2>&1 | grep -E "error" -A5 || true
ls -la ../../scripts/dev/variants/lib$1.so
