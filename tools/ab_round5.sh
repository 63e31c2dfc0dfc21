#!/bin/bash
# Same-box A/B of two ROUNDS: round 5's final tree (git archive 1bbcadc, built under build/ab_r05 - it travels with the snapshot) against this tree,
# the same bench.py command alternating between the two on ONE GPU box (boxes of the pool differ by up to 4 %: lines taken on different boxes do not
# show a 2 % change).  usage: tools/ab_round5.sh <rounds>     (prepare: mkdir -p build/ab_r05 && git archive 1bbcadc | tar -x -C build/ab_r05 &&
# make -C build/ab_r05/gnnome_amd/csrc -j8)
cd "$GRAFT_REPO_ROOT" || exit 1
R=${1:-3}
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4))"; }
for spec in "c2|" "c4shard|--workload c4shard" "10m|--workload 10m" "parity64|--workload parity64" "ecoli|--workload ecoli" "train_c2|--mode train" "train_c4shard|--workload c4shard --mode train --steps 5 --warmup 2"; do
  name=${spec%%|*}; args=${spec#*|}
  for rnd in $(seq 1 $R); do
    a=$(cd build/ab_r05 && timeout 600 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | ms)
    b=$(timeout 600 python bench.py --no-cpu-baseline --no-extras $args 2>/dev/null | ms)
    echo "$name round $rnd: round-5 tree $a ms   this tree $b ms"
  done
done
