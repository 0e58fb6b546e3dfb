# distance-transform prefetch depth (rows of operands in flight): single-frame latency and the batch figure, per variant library (run on the GPU box)
cp cube_slam_wu_amd/libcubeslam_hip.so /tmp/lib_D4.so
for D in 4 8 12; do
  [ $D = 4 ] && cp /tmp/lib_D4.so cube_slam_wu_amd/libcubeslam_hip.so || cp build_tmp/libcubeslam_hip_D$D.so cube_slam_wu_amd/libcubeslam_hip.so
  echo "== DT_AHEAD $D"
  bash tools/latency_quick.sh | grep gray | head -1
  python bench.py --steps 2 --warmup 1 --ba none --no-cpu-baseline --rp-frames 0 --latency-calls 0 --lines-images 0 --no-measure-traffic 2>/dev/null | tail -1 | python -c "import sys,json; print('edge device_ms', json.loads(sys.stdin.read())['edge_front_end']['device_ms'])"
done
cp /tmp/lib_D4.so cube_slam_wu_amd/libcubeslam_hip.so
