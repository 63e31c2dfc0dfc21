#!/bin/bash
# round-4 visit 12: new tests (ADVICE fixes, bf16 storage at 1M edges), MFMA-busy PMC passes, gate traffic PMC, kernel tables
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/v12; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_overlap_similarity.py tests/test_hip_training.py tests/test_hip_parity.py tests/test_gfa_and_mask.py -m gpu -q -k "endpoints_out_of_range or configs2_size or hub_split or gfa or capture or hipgraph or graph" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -6 $O/pytest.log | cut -c1-250
bash tools/pmc_mfma.sh $O/mfma > $O/mfma.log 2>&1; tail -3 $O/mfma.log
python tools/pmc_to_json.py /dev/null /dev/null > /dev/null 2>&1
python tools/pmc_mfma_to_json.py $O/mfma $O/r04_mfma_busy.json 2>&1 | tail -40
bash tools/pmc_traffic.sh $O/traffic > $O/traffic.log 2>&1
python tools/pmc_to_json.py $O/traffic $O/r04_gate_pmc.json 2>&1 | tail -12
