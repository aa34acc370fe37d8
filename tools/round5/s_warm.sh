# R5s: do warm weights pay? A touch kernel streams every >= 1-MB packed weight block in front of the conv launch that reads it
# (MPU_WARM_WEIGHTS=1): per-launch sequence with / without, and the graphed step A/B (the touch launches included).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5s; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-predict --no-cpu-baseline --no-kernel-events --no-peaks --no-e2e"
for v in 0 1; do
  MPU_WARM_WEIGHTS=$v rocprofv3 --kernel-trace --stats -d $O/s$v -o s -- $B --no-graph --steps 10 --warmup 3 > /dev/null 2>&1
  S=$(find $O/s$v -name "*.db" | head -1)
  python $R/tools/rocpd_sequence.py $S > $O/seq$v.txt 2>&1
  rm -rf $O/s$v
  tail -1 $O/seq$v.txt
done
python - <<PY
import re
def rows(p):
    out=[]
    for l in open(p):
        m=re.match(r'\s*(\d+) (\w+).*grid=\s*(\d+)\s+([\d.]+)',l)
        if m: out.append((m.group(2),int(m.group(3)),float(m.group(4))))
    return out
a=rows("$O/seq0.txt"); b=rows("$O/seq1.txt")
ia=0; touch=0.0; pend=0.0
for k,g,t in b:
    if k=="touch_kernel": pend=t; touch+=t; continue
    while ia < len(a) and a[ia][0]!=k: ia+=1
    if ia>=len(a): break
    if pend: print(f"{k:28s} grid={g:5d} cold {a[ia][2]:7.2f}  warm {t:7.2f}  touch {pend:6.2f}  delta {t-a[ia][2]:+7.2f}")
    pend=0.0; ia+=1
print("touch total", touch)
PY
for rep in 1 2; do for v in 0 1; do
  echo -n "graphed step warm=$v: "; MPU_WARM_WEIGHTS=$v $B --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"
done; done
