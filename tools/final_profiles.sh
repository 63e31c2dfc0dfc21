#!/bin/bash
# round-end evidence in one GPU-box visit: rocprofv3 kernel stats (infer + train), PMC traffic of the dominant kernel and of
# every kernel of one forward, bench lines for every workload.   usage: tools/final_profiles.sh <tag, e.g. r02>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-r03}
O=gpurun_out/final; rm -rf $O; mkdir -p $O
for mode in infer train; do
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o r -- python bench.py --mode $mode --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-extras > $O/prof_$mode.bench.json 2> $O/prof_$mode.err
  DB=$(find $O/prof_$mode -name "*.db" | head -1)
  python tools/rocpd_summary.py "$DB" > $O/${TAG}_c2_$mode.kernel_stats.md 2>&1
  rm -rf $O/prof_$mode
  head -14 $O/${TAG}_c2_$mode.kernel_stats.md
done
# PMC traffic of every kernel of one forward / one training step, stamped with the build (round 5: bench.py quotes `roofline.traffic` for whichever
# kernel it names from profiles/<tag>_forward_pmc_<workload>.json)
for w in c2 c4shard; do
  bash tools/pmc_forward.sh $O/pmc_fwd_$w $w > $O/${TAG}_forward_hbm_traffic_$w.md 2>&1; tail -8 $O/${TAG}_forward_hbm_traffic_$w.md
  python tools/pmc_forward_json.py $O/pmc_fwd_$w $O/${TAG}_forward_pmc_$w.json $w > /dev/null; cp $O/${TAG}_forward_pmc_$w.json profiles/; rm -rf $O/pmc_fwd_$w
done
bash tools/pmc_forward.sh $O/pmc_train c2 train > $O/${TAG}_train_hbm_traffic_c2.md 2>&1; python tools/pmc_forward_json.py $O/pmc_train $O/${TAG}_train_pmc_c2.json c2-train > /dev/null; rm -rf $O/pmc_train
# MFMA-busy counters of this build BEFORE the bench lines: bench.py quotes them as `mfma_util` when the stamp matches the library it runs
bash tools/pmc_mfma.sh $O/pmc_mfma > $O/pmc_mfma.log 2>&1; python tools/pmc_mfma_to_json.py $O/pmc_mfma $O/${TAG}_mfma_busy.json > $O/${TAG}_mfma_busy.txt 2>&1
cp $O/${TAG}_mfma_busy.json profiles/; rm -rf $O/pmc_mfma; tail -12 $O/${TAG}_mfma_busy.txt
timeout 900 python bench.py > $O/${TAG}_bench_c2.json 2> $O/bench_c2.err; tail -c 400 $O/${TAG}_bench_c2.json
for w in 10m parity64 c4shard; do timeout 400 python bench.py --workload $w --no-cpu-baseline --no-extras > $O/${TAG}_bench_$w.json 2>/dev/null; done
# the whole configs[3] graph and the configs[4] graph (inference) on ONE GPU: H = 256, 20M / 50M edges
timeout 400 python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/${TAG}_bench_c4_one_gpu.json 2>/dev/null
timeout 600 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_bench_c5_one_gpu.json 2>/dev/null
timeout 600 python bench.py --mode train > $O/${TAG}_bench_c3_train_step.json 2>/dev/null
timeout 600 python bench.py --mode train --storage bf16 --no-cpu-baseline > $O/${TAG}_bench_c3_train_bf16_storage.json 2>/dev/null
timeout 600 python bench.py --mode train --symmetry --no-cpu-baseline > $O/${TAG}_bench_c3_train_symmetry.json 2>/dev/null
timeout 300 python bench.py --gpus 2 --one-gpu-gloo --workload c2 --steps 5 --warmup 2 > $O/${TAG}_bench_n2_plumbing.json 2>/dev/null   # (spawns its own two ranks)
timeout 400 python bench.py --gpus 2 --one-gpu-gloo --mode train --workload c2 --steps 5 --warmup 2 > $O/${TAG}_bench_n2_train_plumbing.json 2>/dev/null
timeout 400 python bench.py --gpus 2 --one-gpu-gloo --mode train --workload c4shard --steps 3 --warmup 1 > $O/${TAG}_bench_n2_train_h256_plumbing.json 2>/dev/null
timeout 400 python bench.py --workload c4shard --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_train_c4shard_h256.json 2>/dev/null
# what inference.py users run: E. coli-sized graph, shipped checkpoint, auto arithmetic (layer 0 in the reference's order) + its kernel table
timeout 400 python bench.py --workload ecoli > $O/${TAG}_bench_ecoli.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_ecoli -o r -- python bench.py --workload ecoli --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-extras > /dev/null 2> $O/prof_ecoli.err
python tools/rocpd_summary.py "$(find $O/prof_ecoli -name '*.db' | head -1)" > $O/${TAG}_ecoli.kernel_stats.md 2>&1; rm -rf $O/prof_ecoli
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_c4t -o r -- python bench.py --workload c4shard --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-extras > /dev/null 2> $O/prof_c4t.err
python tools/rocpd_summary.py "$(find $O/prof_c4t -name '*.db' | head -1)" > $O/${TAG}_c4shard_train.kernel_stats.md 2>&1; rm -rf $O/prof_c4t
for w in c2 10m c4quarter; do timeout 300 python bench.py --gpus 8 --one-gpu-gloo --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_n8_plumbing_$w.json 2>/dev/null; done
timeout 300 python bench.py --gpus 8 --one-gpu-gloo --mode train --workload c2 --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_n8_plumbing_train_c2.json 2>/dev/null
timeout 300 python bench.py --gpus 8 --one-gpu-gloo --mode train --workload c4quarter --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_n8_plumbing_train_c4quarter.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o r -- python bench.py --workload c4shard --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-extras > /dev/null 2> $O/prof_c4.err
python tools/rocpd_summary.py "$(find $O/prof_c4 -name '*.db' | head -1)" > $O/${TAG}_c4shard_infer.kernel_stats.md 2>&1; rm -rf $O/prof_c4
timeout 200 python tools/gate_phase_profile.py --hidden 256 --edges 2500000 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gate256_phases.txt
for k in permuted uniform; do timeout 400 python bench.py --kind $k --no-cpu-baseline --no-extras > $O/${TAG}_bench_c2_$k.json 2>/dev/null; done
timeout 200 python tools/clock_sample.py > $O/${TAG}_clock_power_samples.jsonl 2>/dev/null
for f in c2 10m parity64 c4shard c4_one_gpu c5_one_gpu c2_uniform c2_permuted c3_train_step c3_train_bf16_storage c3_train_symmetry n2_plumbing n2_train_plumbing n2_train_h256_plumbing train_c4shard_h256 ecoli; do python - "$f" "$TAG" <<'PY'
import json,sys
f,tag=sys.argv[1],sys.argv[2]
try:
    d=json.loads([l for l in open(f"gpurun_out/final/{tag}_bench_{f}.json") if l.startswith("{")][-1])
    r=d.get("roofline",{})
    print(f, round(d["ms_per_step"],3), "ms", round(d["value"]/1e6,1), "M edges/s | roofline", r.get("kernel","")[:16], round(r.get("avg_launch_ms",0),4), r.get("bound"), round(r.get("frac",0),3), "traffic", r.get("traffic"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as ex:
    print(f, "FAILED", ex)
PY
done
