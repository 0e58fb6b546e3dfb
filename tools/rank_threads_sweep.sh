ARGS="--no-measure-traffic --ba none --no-cpu-baseline --no-edge --rp-frames 0 --latency-calls 0 --lines-images 0 --steady-steps 0 --image-in-steps 0"
for nt in 256 128 64; do
  export CS_RANK_THREADS=$nt
  echo "== rank threads $nt"
  for rep in 1 2 3; do python bench.py $ARGS --steps 80 --warmup 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   frames/s %.0f  ms/step %.3f' % (d['value'], d['ms_per_step']))"; done
done
