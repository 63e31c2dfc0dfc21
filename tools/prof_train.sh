#!/bin/bash
# rocprofv3 kernel trace of the training step (bench.py --mode train) -> gpurun_out/<tag>/kernel_stats.md + small-kernel summary
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-prof_train}; O=gpurun_out/$TAG; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline $PROF_ARGS > $O/bench.json 2> $O/err.log
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" > $O/kernel_stats.md 2>&1
rm -rf $O/prof
python - "$O/kernel_stats.md" <<'PY'
import re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\| `(.*?)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", l)
    if m:
        rows.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4))))
steps = max(1, [r[1] for r in rows if "k_edge_gate_bf<4, 1, 1" in r[0] or "k_edge_gate_bf<2, 2, 1" in r[0]][0] // 8)
small = [r for r in rows if r[3] < 50]
print(f"steps traced {steps}: {sum(r[2] for r in rows) / steps / 1e3:.2f} ms of kernels per step in {sum(r[1] for r in rows) / steps:.0f} launches; "
      f"kernels under 50 us: {sum(r[2] for r in small) / steps / 1e3:.2f} ms in {sum(r[1] for r in small) / steps:.0f} launches")
for r in sorted(rows, key=lambda r: -r[2])[:28]:
    print("  %-72s %6.1f /step %7.3f ms/step  avg %.1f us" % (r[0][:72], r[1] / steps, r[2] / steps / 1e3, r[3]))
PY
