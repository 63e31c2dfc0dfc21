#!/bin/bash
# round 3, first GPU visit: tests, default bench, N=2 plumbing (self-spawned), ecoli line + kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v1; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for mode in infer train; do
  timeout 500 python bench.py --gpus 2 --steps 5 --warmup 2 --mode $mode --workload c2 --one-gpu-gloo > $O/bench_n2_$mode.json 2> $O/bench_n2_$mode.err; echo "n2 $mode rc=$?"
  tail -c 700 $O/bench_n2_$mode.json; tail -3 $O/bench_n2_$mode.err
done
timeout 500 python bench.py --gpus 2 --steps 3 --warmup 1 --mode train --workload c4shard --one-gpu-gloo > $O/bench_n2_train_h256.json 2> $O/bench_n2_train_h256.err; echo "n2 train h256 rc=$?"
tail -c 500 $O/bench_n2_train_h256.json; tail -3 $O/bench_n2_train_h256.err
timeout 300 python bench.py --workload ecoli > $O/bench_ecoli.json 2> $O/bench_ecoli.err; echo "ecoli rc=$?"
tail -c 1500 $O/bench_ecoli.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python bench.py --workload ecoli --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-extras > $O/ecoli_prof.json 2> $O/ecoli_prof.err
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" > $O/ecoli.kernel_stats.md 2>&1
find $O/prof -name "*.db" -delete
head -24 $O/ecoli.kernel_stats.md
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 1200 $O/bench_default.json
