#!/bin/bash
# The bench lines of tools/final_profiles.sh alone (a host-side change: the library, and with it the PMC / MFMA / kernel-table files, stay as they are).
# usage: tools/final_bench_lines.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r06}; O=gpurun_out/final; mkdir -p $O
timeout 900 python bench.py > $O/${TAG}_bench_c2.json 2> $O/bench_c2.err; tail -c 300 $O/${TAG}_bench_c2.json
for w in 10m parity64 c4shard; do timeout 400 python bench.py --workload $w --no-cpu-baseline --no-extras > $O/${TAG}_bench_$w.json 2>/dev/null; done
timeout 400 python bench.py --workload ecoli > $O/${TAG}_bench_ecoli.json 2>/dev/null
for k in permuted uniform; do timeout 400 python bench.py --kind $k --no-cpu-baseline --no-extras > $O/${TAG}_bench_c2_$k.json 2>/dev/null; done
for f in c2 10m parity64 c4shard c2_uniform c2_permuted ecoli; do python - "$f" "$TAG" <<'PY'
import json,sys
f,tag=sys.argv[1],sys.argv[2]
d=json.loads([l for l in open(f"gpurun_out/final/{tag}_bench_{f}.json") if l.startswith("{")][-1])
p=d.get("placement") or {}
print(f, round(d["ms_per_step"],3), "ms", round(d["value"]/1e6,1), "M edges/s | placement", p.get("tuned"), p.get("start_ms"), p.get("kept_ms"))
PY
done
