#!/bin/bash
# round-4 closing visit: full GPU suite, smoke, the default bench line (timed), then tools/final_profiles.sh r04 + the MFMA-busy passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/fin
python -m pytest tests -q -m gpu > gpurun_out/fin/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/fin/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench_default.err ) 2>&1 | grep real
bash tools/final_profiles.sh r04 2>&1 | tail -60
bash tools/pmc_mfma.sh gpurun_out/final/mfma > gpurun_out/final/mfma.log 2>&1
python tools/pmc_mfma_to_json.py gpurun_out/final/mfma gpurun_out/final/r04_mfma_busy.json | head -40
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/final/prof_c4t -o r -- python bench.py --workload c4shard --mode train --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-extras > /dev/null 2> gpurun_out/final/prof_c4t.err
python tools/rocpd_summary.py "$(find gpurun_out/final/prof_c4t -name '*.db' | head -1)" > gpurun_out/final/r04_c4shard_train.kernel_stats.md 2>&1; rm -rf gpurun_out/final/prof_c4t
timeout 900 python bench.py --mode train --workload c5shard --no-cpu-baseline > gpurun_out/final/r04_bench_train_c5shard.json 2>/dev/null; tail -c 300 gpurun_out/final/r04_bench_train_c5shard.json
