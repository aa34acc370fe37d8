import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multiplanarunet_amd.unet import UNet
q = lambda *a, **k: None
for dtype, cf, C in (("f32", 1, 1), ("bf16", 1, 1), ("bf16", 2, 2)):
    rng = np.random.RandomState(8)
    B, H, D = 2, 32, 2
    x = torch.tensor(rng.randn(B, H, H, C).astype(np.float32), device="cuda")
    y = torch.tensor(rng.randint(0, 3, (B, H * H, 1)).astype(np.uint8), device="cuda")
    a = UNet(n_classes=3, dim=H, n_channels=C, depth=D, complexity_factor=cf, dtype=dtype, logger=q, seed=0)
    b = UNet(n_classes=3, dim=H, n_channels=C, depth=D, complexity_factor=cf, dtype=dtype, logger=q, seed=0)
    for m in (a, b):
        m.compile("Adam", "SparseCategoricalCrossentropy", optimizer_kwargs=dict(lr=1e-3))
    for step in range(2):
        a.forward_backward(x, y, None, want_loss=False); a.apply_gradients(fused=True)
        b.forward_backward(x, y, None, want_loss=False); b.apply_gradients(fused=False)
        torch.cuda.synchronize()
        print(dtype, cf, C, "step", step, "grads equal", torch.equal(a.grads, b.grads), "params equal", torch.equal(a.params, b.params),
              "m", torch.equal(a._adam_m, b._adam_m), "v", torch.equal(a._adam_v, b._adam_v),
              "packed", torch.equal(a.packed, b.packed))
        if not torch.equal(a.params, b.params):
            pa, pb = a.params.cpu().numpy(), b.params.cpu().numpy()
            bad = np.nonzero(pa != pb)[0]
            print("   differing params:", bad.size, "of", pa.size, "first", bad[:8], "max |d|", np.abs(pa - pb).max())
            for name, (kind, off, ps, ls) in a._tensors.items():
                if kind != 0: continue
                n = int(np.prod(ps)); k = np.count_nonzero(pa[off:off + n] != pb[off:off + n])
                if k: print("     ", name, ps, k, "of", n)
        if not torch.equal(a.packed, b.packed):
            qa, qb = a.packed.cpu().numpy(), b.packed.cpu().numpy()
            bad = np.nonzero(qa != qb)[0]
            print("   differing packed bytes:", bad.size, "first", bad[:8], "last", bad[-3:])
