#!/bin/bash
# round-2 call 3J: graph replay vs eager launching of the same train step (what N > 1 pays for launching eagerly)
R="$GRAFT_REPO_ROOT"; cd $R; export TMPDIR=/tmp
for F in "" "--no-graph"; do timeout 300 python bench.py --steps 50 --warmup 10 --no-predict --no-cpu-baseline --no-peaks --no-kernel-events $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flags=[$F]', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], d['config'].get('launch'))"; done
