#!/bin/bash
# round 3 final validation of the committed build: smoke, whole GPU suite, the driver's bench command
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/R3s; mkdir -p $O
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; tail -3 $O/pytest_all.log
timeout 600 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-300
