#!/bin/bash
# the whole GPU suite + smoke once more on the final tree (Python-side changes after the closing visit)
mkdir -p gpurun_out/v42
python -m pytest tests -q -m gpu > gpurun_out/v42/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/v42/pytest.log
tail -3 gpurun_out/v42/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
