#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, one counter per pass (FETCH_SIZE takes 3 of the 4 TCC slots,
# WRITE_SIZE 2 - MI355X_MICROARCH.md, "rocprofv3 PMC slots").  usage: tools/pmc_traffic.sh <outdir>  (GPU box)
OUT=$1; R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 90 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/$C -o p -- python $R/tools/gate_only.py --reps 3 > /dev/null 2>&1 || echo "$C pass failed"
done
ls $R/$OUT
