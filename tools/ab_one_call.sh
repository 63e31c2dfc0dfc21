#!/bin/bash
# Same-box A/B, unprofiled: round 5's tree (build/ab_r05) / this tree with one library call per forward / this tree call by call, with and without the
# per-forward HIP events of bench.py's live roofline; the one call also with all its buffers in ONE block (GNNOME_FORWARD_BUFFERS=block, round 6's first form).   usage: tools/ab_one_call.sh <rounds> [bench arguments]
cd "$GRAFT_REPO_ROOT" || exit 1
R=${1:-3}; shift
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4))"; }
for rnd in $(seq 1 $R); do
  a=$(cd build/ab_r05 && timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | ms)
  b=$(timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | ms)
  c=$(GNNOME_ONE_CALL_FORWARD=0 timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | ms)
  d=$(timeout 600 python bench.py --no-cpu-baseline --no-extras --no-kernel-timers "$@" 2>/dev/null | ms)
  e=$(GNNOME_ONE_CALL_FORWARD=0 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-kernel-timers "$@" 2>/dev/null | ms)
  f=$(cd build/ab_r05 && timeout 600 python bench.py --no-cpu-baseline --no-extras --no-kernel-timers "$@" 2>/dev/null | ms)
  g=$(GNNOME_FORWARD_BUFFERS=block timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | ms)
  echo "round $rnd: round-5 tree $a | one call $b | one call on ONE workspace block $g | call by call $c | no events: one call $d, call by call $e, round-5 tree $f"
done
