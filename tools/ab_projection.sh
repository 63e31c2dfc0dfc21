# A/B of the node-projection kernels inside the whole forward (bench.py --tuning 2=<variant>; the empty variant = the defaults)
for i in 1 2; do
for t in ${VARIANTS:-"" "2=5"}; do
  for w in ${WORKLOADS:-c2 10m}; do
  python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timers ${t:+--tuning $t} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', '$t', round(d['ms_per_step'],4))"
  done
done
done
