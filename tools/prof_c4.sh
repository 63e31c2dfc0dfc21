cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/pc4
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/pc4 -o r01 -- python bench.py --workload c4shard --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers > gpurun_out/pc4/bench.json 2> gpurun_out/pc4/err.log
DB=$(find gpurun_out/pc4 -name "*.db" | head -1); python tools/rocpd_summary.py "$DB" | head -14; find gpurun_out/pc4 -name "*.db" -delete
