#!/bin/bash
# PMC counter groups on the two aggregation kernels at configs[1] (one group per pass, --kernel-trace only).  usage: tools/pmc_stream.sh <outdir>   (GPU box)
OUT=$1; R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for which in gather stream; do
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/$OUT/$which/g$i -o p -- python $R/tools/stream_agg_time.py 128 100000 1000000 banded 0 once_$which > /dev/null 2>&1 || echo "$which group $i failed: $G"
done
done
python - "$R/$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for which in ("gather", "stream"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + f"/{which}/g*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].split("<")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        if "aggregate" in k or "stream_finish" in k:
            res[f"{which}:{k}"] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
