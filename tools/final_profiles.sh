#!/bin/bash
# round-end evidence: rocprofv3 kernel stats (infer + train), PMC traffic of the dominant kernel, bench lines.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
for mode in infer train; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o r01 -- python bench.py --mode $mode --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers > $O/prof_$mode.bench.json 2> $O/prof_$mode.err
  DB=$(find $O/prof_$mode -name "*.db" | head -1)
  python tools/rocpd_summary.py "$DB" > $O/kernel_stats_$mode.md 2>&1
  find $O/prof_$mode -name "*.db" -delete
  head -14 $O/kernel_stats_$mode.md
done
bash tools/pmc_traffic.sh $O/pmc
python tools/pmc_to_json.py $O/pmc $O/gate_pmc.json k_edge_gate_bf | tail -8
find $O/pmc -name "*.db" -delete
timeout 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 500 $O/bench_c2.json
for w in 10m parity64 c4shard; do timeout 400 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
timeout 400 python bench.py --kind uniform --no-cpu-baseline > $O/bench_c2_uniform.json 2>/dev/null
timeout 400 python bench.py --mode train --steps 10 --warmup 3 > $O/bench_train.json 2>/dev/null
for f in c2 10m parity64 c4shard c2_uniform train; do python - "$f" <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads([l for l in open(f"gpurun_out/final/bench_{f}.json") if l.startswith("{")][-1])
    r=d.get("roofline",{})
    print(f, round(d["ms_per_step"],3), "ms", round(d["value"]/1e6,1), "M edges/s | gate", round(r.get("avg_launch_ms",0),4), r.get("bound"), round(r.get("frac",0),3), "| cold", d.get("cold_ms_incl_graph_views"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as ex:
    print(f, "FAILED", ex)
PY
done
