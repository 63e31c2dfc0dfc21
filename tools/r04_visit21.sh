#!/bin/bash
# round-4 visit 21: raw rows through registers instead of LDS-DMA in the fp16x3 H = 256 gate (experiment, tuning 1 = 200)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v21; rm -rf $O; mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, sys
sys.path.insert(0, '.')
from gnnome_amd import ops
from gnnome_amd.synth import make_graph
dev = torch.device("cuda", 0)
for e in (33, 70_001, 300_007):
    n, H = max(e // 10, 8), 256
    g = make_graph(n - n % 2, e, seed=1)
    n = n - n % 2
    views = ops.GraphViews(g["src"].to(dev), g["dst"].to(dev), n)
    gen = torch.Generator(device=dev).manual_seed(0)
    ee = torch.randn(e, H, device=dev, generator=gen); P = torch.randn(n, 2 * H, device=dev, generator=gen)
    W3 = torch.randn(H, H, device=dev, generator=gen) / 16; sc, sh = torch.rand(H, device=dev, generator=gen), torch.randn(H, device=dev, generator=gen)
    a = ops.edge_gate(ee, P[:, :H], P[:, H:], views, W3, 0, sc, sh, out=torch.empty_like(ee))
    ops.set_tuning(1, 200)
    b = ops.edge_gate(ee, P[:, :H], P[:, H:], views, W3, 0, sc, sh, out=torch.full_like(ee, float("nan")))
    ops.set_tuning(1, 0)
    print("E", e, "register path equals the DMA path bit for bit:", torch.equal(a, b))
PY
timeout 600 python tools/gate_time.py --hidden 256 --edges 2500000 --variants 0 --ablations 0,200 --reps 30 2>&1 | grep -v amdgpu.ids | tee $O/gate_reg_ab.txt
for t in "" "1=200"; do
timeout 400 python bench.py --workload c4shard --no-cpu-baseline --no-extras ${t:+--tuning $t} > $O/b.json 2> $O/b.err
python - $O/b.json "c4shard [$t]" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], round(d["ms_per_step"],3),"ms gate", round(d["roofline"]["avg_launch_ms"],4))
except Exception as ex: print(sys.argv[2],"FAILED",ex)
PY
done
