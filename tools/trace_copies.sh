#!/bin/bash
# which HIP API calls does one forward make?  (hip trace of a short E. coli-sized run; no PMC)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/trace; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d $O/t -o r -- python bench.py --workload ecoli --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-extras > /dev/null 2> $O/err.log
F=$(find $O/t -name "*hip_api_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
c = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    c[row.get("Function") or row.get("Name")] += 1
for k, v in c.most_common(25): print(v, k)
PY
M=$(find $O/t -name "*memory_copy_trace.csv" | head -1)
python - "$M" <<'PY'
import csv, sys, collections
c = collections.Counter()
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "memory copies; columns:", list(rows[0].keys()) if rows else None)
for r in rows: c[(r.get("Direction"), r.get("Size") or r.get("Bytes"))] += 1
for k, v in c.most_common(15): print(v, k)
PY
rm -rf $O/t
