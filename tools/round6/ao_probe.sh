R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2; do timeout 300 python bench.py --e2e-only --steps 60 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); e=d["train_e2e"]; print("e2e-only: fraction", e["fraction_of_headline"], "probe latency", e["producer_stream_latency_us"], e["producer_stream_candidates_ms_per_step"])'; done
timeout 300 python - <<'PY'
import torch, time
from multiplanarunet_amd.pipeline import pick_side_streams
dev = torch.device("cuda:0")
for k in range(3):
    r = pick_side_streams(dev)
    print("fresh process, call %d: candidate latencies (us):" % k, [round(l) for _, l in r])
# after a graph capture + many streams
from multiplanarunet_amd.unet import UNet
q = lambda *a, **k: None
m = UNet(n_classes=3, dim=128, n_channels=1, depth=4, complexity_factor=1, dtype="bf16", logger=q, seed=0, device=dev)
m.compile("Adam", "SparseCategoricalCrossentropy")
x = torch.randn(16, 128, 128, 1, device=dev); y = torch.randint(0, 3, (16, 128 * 128, 1), device=dev, dtype=torch.uint8)
rep = m.make_graphed_train_step(x, y, None)
for _ in range(5): rep()
torch.cuda.synchronize()
r = pick_side_streams(dev)
print("after a graphed step (library side stream exists):", [round(l) for _, l in r])
PY
