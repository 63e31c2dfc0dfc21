#!/bin/bash
# HBM bytes fetched by the aggregation kernel inside whole forwards, per aggregation variant (gnnome_set_tuning key 7).
# usage: tools/pmc_agg_variant.sh <outdir> "<variants>"   (GPU box)
OUT=$1; VARS=${2:-"0 3"}; R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for V in $VARS; do
  timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/v$V -o p -- python $R/tools/forward_ab.py $V 7 > /dev/null 2>&1 || echo "variant $V failed"
  python - "$R/$OUT/v$V" "$V" <<'PY'
import csv, glob, re, sys
tot, n = 0.0, 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if re.search(r"k_node_aggregate<\d+, \d+, \d+, false", row["Kernel_Name"]):   # the regular launch, not the hub finish
            tot += float(row["Counter_Value"]); n += 1
print(f"aggregation variant {sys.argv[2]}: {2 * tot * 1024 / max(n, 1) / 1e6:.1f} MB fetched per launch (FETCH_SIZE x 2, {n} launches)")
PY
done
