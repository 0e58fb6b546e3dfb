#!/bin/bash
# builds build_tmp/band_bench against the current ba_kernels.o (run `make -C cube_slam_wu_amd/csrc` first)
set -e
cd "$(dirname "$0")/../.."
mkdir -p build_tmp
hipcc --offload-arch=gfx950 -O3 -c -x hip tools/microbench/band_bench.cpp -o build_tmp/band_bench.o
hipcc --offload-arch=gfx950 build_tmp/band_bench.o cube_slam_wu_amd/csrc/ba_kernels.o cube_slam_wu_amd/csrc/bcr_kernels.o -o build_tmp/band_bench
# the same harness over the band kernels built with plain stores + agent release / acquire fences instead of the write-through hand-offs
# (BAND_WT = 0): tests/test_ba_gpu.py holds the two builds to bit-identical factors and solutions
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DBAND_WT=0 -c cube_slam_wu_amd/csrc/ba_kernels.hip -o build_tmp/ba_kernels_fence.o
hipcc --offload-arch=gfx950 build_tmp/band_bench.o build_tmp/ba_kernels_fence.o cube_slam_wu_amd/csrc/bcr_kernels.o -o build_tmp/band_bench_fence
# counter calibration (tools/pmc_calib.sh)
hipcc --offload-arch=gfx950 -O3 -x hip tools/microbench/pmc_calib.cpp -o build_tmp/pmc_calib
# the diagonal-block routine of the banded solver alone
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -x hip tools/microbench/potf2_bench.cpp -o build_tmp/potf2_bench
# host-only check of the symbolic phase of the general sparse reduced solve (tests/test_ba_oracle.py runs it)
hipcc -O2 -std=c++17 tools/microbench/sparse_plan_check.cpp -o build_tmp/sparse_plan_check
# host-only: the error bound of fastAtan2 that the LSD host stage's region growing relies on (tests/test_capi_symbols.py runs it with a stride)
g++ -O2 -std=c++17 tools/microbench/lsd_atan_bound.cpp -o build_tmp/lsd_atan_bound -pthread
# where the time of bcr_factor_kernel goes (stamps at its barriers; development tool)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -x hip tools/microbench/bcr_factor_probe.cpp -o build_tmp/bcr_factor_probe
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -x hip tools/microbench/potf4_bench.cpp -o build_tmp/potf4_bench
