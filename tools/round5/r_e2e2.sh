# R5r: why train_e2e drops to 0.6 of the headline when the predict leg ran before it (R5a, R5p) and not otherwise (R5b, R5q)
# (historical: MPU_BENCH_KEEP_CACHE existed in that experiment build of bench.py only)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for keep in 1 0 1 0; do
  MPU_BENCH_KEEP_CACHE=$keep python bench.py --no-cpu-baseline --no-peaks --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); e = d['train_e2e']
print('keep_cache=$keep: ms_per_step', d['ms_per_step'], 'e2e', e['value'], 'frac', e['fraction_of_headline'], 'serial', e['serial_slices_per_s'], 'sampler', e['sampler_alone_slices_per_s'], 'predict s', d['predict_fuse']['seconds'], 'stream latency us', e.get('producer_stream_latency_us'))"
done
