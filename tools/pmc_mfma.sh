#!/bin/bash
# MFMA utilisation of the dense kernels from PMC counters (north_star: "MFMA utilisation ... against CDNA4 peak").
# One rocprofv3 --pmc pass per workload, --kernel-trace only (no other trace domains).  usage: tools/pmc_mfma.sh <outdir>  (GPU box)
OUT=$1; R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
run() {   # name, bench arguments
  local name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/$name -o p -- \
      python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-extras "$@" > /dev/null 2> $R/$OUT/$name.err || echo "$name: pass failed"
}
mkdir -p $R/$OUT
run c2_infer
run c4shard_infer --workload c4shard
run c2_train --mode train
run c4shard_train --workload c4shard --mode train
ls $R/$OUT
