#!/bin/bash
# round-2 call A: full GPU suite (old + new tests), baseline bench, per-layer conv/wgrad timing at B=16 and B=64
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -rA --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
for B in 16 64; do
  BENCH_B=$B timeout 200 python tools/bench_conv.py fwd 20 > $O/conv_fwd_B$B.txt 2>&1
  BENCH_B=$B timeout 200 python tools/bench_conv.py wgrad 20 > $O/conv_wgrad_B$B.txt 2>&1
done
tail -3 $O/conv_fwd_B16.txt $O/conv_fwd_B64.txt
