"""R5x: where the overlapped graphed pipeline and the serial eager loop part ways (per-step batch checksums and losses)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_gpu_pipeline import _model_and_sampler
from multiplanarunet_amd.pipeline import TrainPipeline

def trace(graphed, overlap, n=12, read_at=(7,)):
    m, s = _model_and_sampler(11)
    p = TrainPipeline(m, s, graphed=graphed, overlap=overlap)
    out = []
    for i in range(n):
        p.step()
        torch.cuda.synchronize()
        out.append((float(p.gx.double().sum()), int(p.gy.long().sum()), float(p.loss_sum.item()), float(m.params.double().sum())))
        if (i + 1) in read_at:
            p.epoch_loss()
    return out

a = trace(False, False); b = trace(True, True); c = trace(True, False); d = trace(False, True)
for i, rows in enumerate(zip(a, b, c, d)):
    print(i + 1, " | ".join("%.6f %d %.9f %.9f" % r for r in rows), "EQ" if len(set(rows)) == 1 else "DIFF")
