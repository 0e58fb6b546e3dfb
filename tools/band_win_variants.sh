#!/bin/bash
# timing experiments on the window-resident fronts (results of the variants that skip work are WRONG by construction; only the phase clock is read)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_tmp
hipcc --offload-arch=gfx950 -O3 -c -x hip tools/microbench/band_bench.cpp -o build_tmp/band_bench.o 2>/dev/null
for v in NOPRIO NOCHUNK NOEXTRA; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DWIN_X_$v -c cube_slam_wu_amd/csrc/ba_kernels.hip -o build_tmp/ba_kernels_$v.o
  hipcc --offload-arch=gfx950 build_tmp/band_bench.o build_tmp/ba_kernels_$v.o -o build_tmp/band_bench_$v
done
