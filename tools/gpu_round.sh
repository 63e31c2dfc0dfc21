#!/bin/bash
# one GPU-box visit: tests, default bench, N>1 plumbing checks.  Every leg bounded.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/round
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/round/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/round/pytest.log
tail -5 gpurun_out/round/pytest.log
timeout 400 python bench.py > gpurun_out/round/bench_default.json 2> gpurun_out/round/bench_default.err; echo "bench rc=$?"
tail -c 600 gpurun_out/round/bench_default.json
for mode in infer train; do
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --mode $mode --one-gpu-gloo > gpurun_out/round/bench_n2_$mode.json 2> gpurun_out/round/bench_n2_$mode.err; echo "n2 $mode rc=$?"
  tail -c 400 gpurun_out/round/bench_n2_$mode.json; tail -5 gpurun_out/round/bench_n2_$mode.err
done
timeout 400 python bench.py --mode train --steps 10 --warmup 3 > gpurun_out/round/bench_train.json 2> gpurun_out/round/bench_train.err; echo "train rc=$?"
tail -c 300 gpurun_out/round/bench_train.json
