#!/bin/bash
# round-2 call W: full GPU suite, smoke, full bench line, rocprofv3 summaries (kernel trace + PMC traffic passes) of the final build
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/r2w3; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -rA --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|ERROR|bf16 6-view" $O/pytest.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json; echo
bash tools/profile_round.sh r02d
