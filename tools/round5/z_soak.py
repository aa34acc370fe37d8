"""R5z: soak of the overlapped graphed pipeline against the serial eager loop: 400 steps, an epoch read every 50, a learning-rate
change every 100, an eager validation-like forward on a larger batch in between -- parameters compared bit for bit at every read."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from test_gpu_pipeline import _model_and_sampler
from multiplanarunet_amd.pipeline import TrainPipeline

m0, s0 = _model_and_sampler(31, elastic=True)
m1, s1 = _model_and_sampler(31, elastic=True)
p0 = TrainPipeline(m0, s0, graphed=False, overlap=False)
p1 = TrainPipeline(m1, s1)
xb = torch.randn(32, 64, 64, 1, device="cuda")
ok = True
for ep in range(8):
    a, b = p0.run_epoch(50), p1.run_epoch(50)
    torch.cuda.synchronize()
    same = (a == b) and torch.equal(m0.params, m1.params) and torch.equal(m0.bn_state, m1.bn_state)
    va, vb = m0.predict_on_batch(xb), m1.predict_on_batch(xb)
    same = same and torch.equal(va, vb)
    print(ep, "%.6f %.6f" % (a, b), "EQ" if same else "DIFF", flush=True)
    ok = ok and same
    if ep % 2 == 1:
        for m in (m0, m1):
            m.optimizer_kwargs["lr"] *= 0.9
print("SOAK", "OK" if ok else "FAILED")
