#!/bin/bash
# usage: tools/pmc_gate.sh <variant> <outdir> [ablation]
# GPU box only.  One rocprofv3 --pmc pass per counter group, each under its own timeout.
# (TA_* counters are left out: a pass with TA_*_sum never returned on this pool.)
V=$1; OUT=$2; A=${3:-0}; R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
         "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum" \
         "TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 75 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/p$i -o p -- python $R/tools/gate_only.py --variant $V --ablation $A --reps 3 > /dev/null 2>&1 || echo "pass $i timed out / failed"
done
ls $R/$OUT
