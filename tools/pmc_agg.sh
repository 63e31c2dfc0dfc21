#!/bin/bash
# PMC counter groups on the aggregation kernel (one group per pass).  usage: tools/pmc_agg.sh <outdir> [H]   (GPU box)
OUT=$1; H=${2:-128}; R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
i=0
for G in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/$OUT/g$i -o p -- python $R/tools/agg_time.py $H once > /dev/null 2>&1 || echo "group $i failed: $G"
done
python - "$R/$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if "aggregate" in k or "gate" in k or "linear" in k:
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:36s} {sum(v) / len(v):16.1f}   (n={len(v)})")
PY
