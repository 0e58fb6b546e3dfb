# scratch: time the detect path with each library variant under exp_libs/ (run on the GPU box; the box copy is disposable)
for f in exp_libs/lib_*.so; do
  cp $f cube_slam_wu_amd/libcubeslam_hip.so
  n=$(basename $f .so)
  python bench.py --steps 10 --warmup 2 --ba none --no-cpu-baseline --no-edge --rp-frames 0 --inflight ${INFLIGHT:-1} 2>/dev/null | tail -1 > gpurun_out/exp_$n.json
done
echo done
