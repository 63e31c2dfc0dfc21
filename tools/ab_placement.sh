#!/bin/bash
# Same-box A/B: the forward's buffers placed by gnnome_amd.ops._placed_buffers (default) against as the allocator hands them out (GNNOME_TUNE_PLACEMENT=0),
# alternating bench.py processes.   usage: tools/ab_placement.sh <rounds> [bench arguments]
cd "$GRAFT_REPO_ROOT" || exit 1
R=${1:-4}; shift
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d.get('placement') or {}; print(round(d['ms_per_step'],4), 'ms', ('(start %s kept %s)' % (p.get('start_ms'), p.get('kept_ms'))) if p.get('tuned') else '')"; }
for rnd in $(seq 1 $R); do
  a=$(timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | ms)
  b=$(GNNOME_TUNE_PLACEMENT=0 timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | ms)
  echo "round $rnd: placed $a | as allocated $b"
done
