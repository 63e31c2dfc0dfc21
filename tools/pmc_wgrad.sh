#!/bin/bash
# usage: tools/pmc_wgrad.sh <outdir> <ablation> -- SQ counters of the weight-gradient kernels (GPU box only, one --pmc pass per group)
OUT=$1; A=${2:-0}; R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 100 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/p$i -o p -- python $R/tools/wgrad_time.py 2500000 256 256 $A > /dev/null 2>&1 || echo "pass $i timed out / failed"
done
python - "$R/$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "wgrad" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} mean {sum(v)/len(v):16.0f}  n={len(v)}")
PY
