#!/bin/bash
# round-6 GPU-box visits (one leg or several per call; logs under gpurun_out/r06/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
for leg in "$@"; do case $leg in
  bench)      timeout 900 python bench.py > $O/r06_bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; tail -c 600 $O/r06_bench_c2.json ;;
  kstats)     for w in c2 c4shard; do
                timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$w -o r -- python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-extras > $O/prof_$w.bench.json 2> $O/prof_$w.err
                python tools/rocpd_summary.py "$(find $O/prof_$w -name '*.db' | head -1)" > $O/r06_${w}_infer.kernel_stats.md 2>&1; rm -rf $O/prof_$w; head -16 $O/r06_${w}_infer.kernel_stats.md
              done ;;
  kstats_train) timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_train -o r -- python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-extras > $O/prof_train.bench.json 2> $O/prof_train.err
              python tools/rocpd_summary.py "$(find $O/prof_train -name '*.db' | head -1)" > $O/r06_c2_train.kernel_stats.md 2>&1; rm -rf $O/prof_train; head -20 $O/r06_c2_train.kernel_stats.md ;;
  mfma)       bash tools/pmc_mfma.sh $O/pmc_mfma > $O/pmc_mfma.log 2>&1; python tools/pmc_mfma_to_json.py $O/pmc_mfma $O/r06_mfma_busy.json > $O/r06_mfma_busy.txt 2>&1; tail -40 $O/r06_mfma_busy.txt ;;
  pmc)        for w in c2 c4shard; do bash tools/pmc_forward.sh $O/pmc_fwd_$w $w > $O/r06_forward_hbm_traffic_$w.md 2>&1; tail -12 $O/r06_forward_hbm_traffic_$w.md
                python tools/pmc_forward_json.py $O/pmc_fwd_$w $O/r06_forward_pmc_$w.json $w > /dev/null; rm -rf $O/pmc_fwd_$w; done ;;
  tests)      timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log ;;
  *) echo "unknown leg $leg" ;;
esac; done
