#!/bin/bash
# Same-box kernel tables of round 5's final tree (build/ab_r05, see tools/ab_round5.sh) and of this tree (one library call per forward, and call by call):
# rocprofv3 --kernel-trace of the same short bench.py run in each.   usage: tools/ab_round5_kernels.sh [workload]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
W=${1:-c2}; O=$GRAFT_REPO_ROOT/gpurun_out/r06/abk; rm -rf $O; mkdir -p $O
run() {   # name, directory, environment assignment
  (cd $2 && env $3 timeout 400 rocprofv3 --kernel-trace --stats -d $O/$1 -o r -- python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers --no-extras > $O/$1.json 2> $O/$1.err)
  echo "== $1: $(python -c "import sys,json; d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],4), 'ms per forward under the profiler')")"
  python tools/rocpd_summary.py "$(find $O/$1 -name '*.db' | head -1)" 2>&1 | sed -n 4,12p | cut -c1-150
  rm -rf $O/$1
}
for rnd in 1 2; do
  run r05_tree build/ab_r05 X=1
  run this_tree . X=1
  run this_tree_call_by_call . GNNOME_ONE_CALL_FORWARD=0
done
