#!/bin/bash
# round 4 call I: the whole GPU suite + smoke on the current build (I2: the files behind the first failure)
R="$GRAFT_REPO_ROOT"; cd $R; O=$R/gpurun_out/R4i; mkdir -p $O
timeout 1500 python -m pytest ${SUITE_FILES:-tests/} -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -5 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -2 $O/smoke.log
