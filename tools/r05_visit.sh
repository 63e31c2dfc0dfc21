#!/bin/bash
# one GPU-box visit of round 5: `tools/r05_visit.sh <leg> [<leg> ...]`, every leg bounded, logs under gpurun_out/r05/.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05; mkdir -p $out
for leg in "$@"; do
  case $leg in
    stream_tests) timeout 900 python -m pytest tests/test_stream_aggregate.py -x -q > $out/stream_tests.log 2>&1; echo "stream_tests rc=$?"; tail -15 $out/stream_tests.log ;;
    stream_time)  for a in "128 100000 1000000 banded 0,256,512,1024" "128 1000000 10000000 banded 0" "256 250000 2500000 banded 0" "64 100000 1000000 banded 0" "128 100000 1000000 uniform 512"; do
                    timeout 300 python tools/stream_agg_time.py $a >> $out/stream_time.jsonl 2>> $out/stream_time.err; done; echo "stream_time rc=$?"; cat $out/stream_time.jsonl; tail -5 $out/stream_time.err ;;
    tests)        timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest.log ;;
    bench)        timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; tail -c 1500 $out/bench_default.json; tail -3 $out/bench_default.err ;;
    bench_ab)     for v in off auto; do GNNOME_STREAM_AGGREGATE=$v timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $out/bench_stream_$v.json 2> $out/bench_stream_$v.err; echo "bench $v rc=$?"; tail -c 700 $out/bench_stream_$v.json; done ;;
    agg_scaling)  timeout 400 python tools/agg_scaling.py 128 > $out/agg_scaling.jsonl 2> $out/agg_scaling.err; echo "agg_scaling rc=$?"; cat $out/agg_scaling.jsonl; tail -3 $out/agg_scaling.err ;;
    pmc_stream)   timeout 1500 bash tools/pmc_stream.sh $out/pmc_stream > $out/pmc_stream.log 2>&1; echo "pmc_stream rc=$?"; tail -120 $out/pmc_stream.log ;;
    pmc_forward)  for w in c2 c4shard; do rm -rf $out/pmc_fwd_$w; timeout 600 bash tools/pmc_forward.sh $out/pmc_fwd_$w $w > $out/r05_forward_hbm_traffic_$w.md 2>&1
                    python tools/pmc_forward_json.py $out/pmc_fwd_$w profiles/r05_forward_pmc_$w.json $w; cp profiles/r05_forward_pmc_$w.json $out/; done ;;
    clocks)       timeout 300 python tools/clock_sample.py > $out/clock_samples.jsonl 2> $out/clock_samples.err; echo "clocks rc=$?"; cut -c1-700 $out/clock_samples.jsonl; tail -2 $out/clock_samples.err ;;
    node_order)   timeout 900 python -m pytest tests/test_node_order.py tests/test_overlap_similarity.py tests/test_edge_tile_f16.py -m gpu -x -q > $out/node_order_tests.log 2>&1; echo "node_order rc=$?"; tail -6 $out/node_order_tests.log
                  for k in banded permuted uniform; do timeout 300 python bench.py --kind $k --no-cpu-baseline --no-extras --steps 50 --warmup 5 > $out/bench_c2_$k.json 2> $out/bench_c2_$k.err; python - $out/bench_c2_$k.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], round(d["ms_per_step"], 4), "ms", d["config"].get("node_order"), d["cold"].get("node_order_ms"))
PY
                  done ;;
    *) echo "unknown leg $leg" ;;
  esac
done
