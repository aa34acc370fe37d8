# R5q: train_e2e under variants of the producer stream (priority), three repetitions each, same box
# (historical: MPU_PIPE_PRIORITY was a switch of that experiment build only; pipeline.pick_side_stream replaced it)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2 3; do
  for prio in -1 0; do
    MPU_PIPE_PRIORITY=$prio python bench.py --e2e-only --steps 40 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); e = d['train_e2e']
print('prio $prio rep $rep: headline', d['headline_slices_per_s'], 'e2e', e['value'], 'frac', e['fraction_of_headline'], 'serial', e['serial_slices_per_s'], 'sampler', e['sampler_alone_slices_per_s'], 'reads/batch', e['sampler_host_reads_per_batch'])"
  done
done
