#!/bin/bash
# round-3 closing visit: full GPU suite, the default bench line (timed), then tools/final_profiles.sh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fin
python -m pytest tests -q -m gpu > gpurun_out/fin/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/fin/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench_default.err ) 2>&1 | grep real
bash tools/final_profiles.sh r03 2>&1 | tail -60
