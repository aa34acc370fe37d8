# R5-fold: BatchNorm finalize folded into the apply pass where the producer left <= 64 partial rows: bitwise equality with the separate
# launches (tests), then the graphed configs[1] step with MPU_BN_FOLD=1 / 0, three alternations, and the per-launch sequence
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5fold; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_unet.py -q -x -k "bn_finalize_folded or f32_train_step or bf16_forward_and_step or graphed or fused_pool" 2>&1 | tail -4
B="python bench.py --no-predict --no-cpu-baseline --no-kernel-events --no-peaks --no-e2e"
for rep in 1 2 3; do for v in 1 0; do
  echo -n "graphed step fold=$v: "; MPU_BN_FOLD=$v $B --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"
done; done
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  MPU_BN_FOLD=$v rocprofv3 --kernel-trace --stats -d $O/s$v -o s -- python $R/bench.py --no-predict --no-cpu-baseline --no-kernel-events --no-peaks --no-e2e --no-graph --steps 10 --warmup 3 > /dev/null 2>&1
  python $R/tools/rocpd_sequence.py $(find $O/s$v -name "*.db" | head -1) > $O/seq$v.txt 2>&1; rm -rf $O/s$v
  echo "== fold=$v"; grep -E "bn_" $O/seq$v.txt | awk '{s+=$NF; n+=1} END {print n, "bn launches", s, "us"}'; tail -1 $O/seq$v.txt
done
grep -E "bn_fold|bn_bwd_fold" $O/seq1.txt | head -20
