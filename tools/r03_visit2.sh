#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v2; mkdir -p $O
timeout 900 python -m pytest tests/test_overlap_similarity.py tests/test_gfa_and_mask.py tests/test_hip_parity.py -m gpu -x -q -k "overlap or similarit or gfa or rejects or layer_and_predictor or pipeline or cluster" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/mfma_order_probe.hip -o /tmp/mfma_probe 2>/dev/null && /tmp/mfma_probe | tee $O/mfma_order.txt
python - <<'PY' 2>&1 | tee $O/overlap_time.txt
import random, time, torch, sys
sys.path.insert(0, '.')
from gnnome_amd import overlap
rng = random.Random(0)
R = 4000
genome = "".join(rng.choice("ACGT") for _ in range(R * 2500 + 20000))
reads = [genome[r * 2500: r * 2500 + rng.randrange(12000, 18000)] for r in range(R)]
src, dst, ol = [], [], []
for r in range(R - 6):
    for t in range(1, 6):
        o = len(reads[r]) - 2500 * t
        if o > 500:
            src.append(2 * r); dst.append(2 * (r + t)); ol.append(min(o, len(reads[r + t])))
            src.append(2 * (r + t) + 1); dst.append(2 * r + 1); ol.append(min(o, len(reads[r + t])))
packed = overlap.pack_reads(reads)
dev = torch.device('cuda', 0)
d, s = overlap.edit_distances(packed, src, dst, ol, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
d, s = overlap.edit_distances(packed, src, dst, ol, device=dev)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
cells = sum(o * o for o in ol)
print(f"{len(ol)} overlaps, mean length {sum(ol)/len(ol):.0f}, {dt*1e3:.1f} ms, {cells/dt/1e12:.2f} T cell updates/s, {len(ol)/dt:.0f} overlaps/s, max dist {int(d.max())}")
PY
