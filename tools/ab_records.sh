#!/bin/bash
# Same-box A/B in alternating bench.py processes: the aggregation's record form on up to H = 64 (default) / up to 128 / off.   usage: tools/ab_records.sh <rounds> [bench arguments]
cd "$GRAFT_REPO_ROOT" || exit 1
R=${1:-3}; shift
ms() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4))"; }
for rnd in $(seq 1 $R); do
  a=$(GNNOME_NODE_RECORDS=0 timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | ms)
  b=$(GNNOME_NODE_RECORDS=64 timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | ms)
  c=$(GNNOME_NODE_RECORDS=128 timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | ms)
  echo "round $rnd: records off $a | up to H = 64 $b | up to H = 128 $c ms"
done
