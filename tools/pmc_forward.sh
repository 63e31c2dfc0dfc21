#!/bin/bash
# HBM traffic of every kernel of one forward (FETCH_SIZE / WRITE_SIZE, one counter per pass).  usage: tools/pmc_forward.sh <outdir> [workload] [infer|train]
OUT=$1; W=${2:-c2}; M=${3:-infer}; R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/$C -o p -- python $R/bench.py --workload $W --mode $M --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-extras > /dev/null 2>&1 || echo "$C pass failed"
done
python - "$R/$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for k, C in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    with open(f"{out}/{C}/p_counter_collection.csv") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0][-60:]
            tot[name][k] += float(row["Counter_Value"])
            if k == 0:
                tot[name][2] += 1
print("kernel | launches | HBM read MB/launch (FETCH_SIZE x2) | HBM write MB/launch")
for name, (f, w, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:10]:
    print(f"{name} | {n} | {2 * f * 1024 / n / 1e6:.1f} | {w * 1024 / n / 1e6:.1f}")
PY
