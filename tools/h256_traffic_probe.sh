#!/bin/bash
# Round 5 (VERDICT r4 item 2 / missing 6): what reaches HBM at H = 256 in the gate k_edge_tile_f16<0>: launch time and FETCH_SIZE of the shipped form against
# compile-time variants (gnnome_set_tuning(1, 100 + mask): 64 = the pair of column-half workgroups walks its tiles in the same order, 128 = nontemporal e'
# stores, 192 = both).   usage: <outdir>  (GPU box)
OUT=$1; R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/$OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/tools/gate_time.py --hidden 256 --edges 2500000 --variants 0 --ablations 0,164,228,292 2>/dev/null | sed "s/^/gate time: /"
for A in 0 164 228 292; do
  for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/gate_${A}_$C -o p -- python $R/tools/gate_only.py --hidden 256 --edges 2500000 --reps 3 --ablation $A > /dev/null 2>&1
  python - $R/$OUT/gate_${A}_$C/p_counter_collection.csv $A $C <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_edge_tile_f16" in r["Kernel_Name"]]
m = (2 if sys.argv[3] == "FETCH_SIZE" else 1) * sum(v) / max(len(v), 1) * 1024 / 1e6
print(f"gate ablation {sys.argv[2]}: {sys.argv[3]} = {m:.1f} MB per launch ({len(v)} launches; e rows = 2560 MB)")
PY
  done
done
