"""Markdown table "which kernel runs each layer of the configs[1] train step, how long, at what rate" from a step-sequence file
(tools/rocpd_sequence.py output). usage: layer_table.py train_step_sequence.txt   Dev / documentation tool (round 6)."""
import re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\s*(\d+)\s+(\S+?)[<(].*grid=\s*(\d+)\s+([\d.]+)\s*$", l)
    if m:
        rows.append((m.group(2), int(m.group(3)), float(m.group(4)), l))
B, K = 16, 3
def gf(px, cin, cout, taps): return 2.0 * B * px * px * cin * cout * taps / 1e9
F = [64, 128, 256, 512, 1024]
layers = []                                            # (name, shape text, GF)
for i in range(4):
    px = 128 >> i; cin = 1 if i == 0 else F[i - 1]
    layers.append(("encoder_L%d_conv1" % i, "%d->%d @ %d^2" % (cin, F[i], px), gf(px, cin, F[i], 9)))
    layers.append(("encoder_L%d_conv2" % i, "%d->%d @ %d^2" % (F[i], F[i], px), gf(px, F[i], F[i], 9)))
layers.append(("bottom_conv1", "512->1024 @ 8^2", gf(8, 512, 1024, 9)))
layers.append(("bottom_conv2", "1024->1024 @ 8^2", gf(8, 1024, 1024, 9)))
for j in range(4):
    lvl = 3 - j; px = 128 >> lvl; f = F[lvl]
    layers.append(("upsample_L%d_conv1 (2x2 up)" % j, "%d->%d @ %d^2" % (F[lvl + 1], f, px), gf(px, F[lvl + 1], f, 4)))
    layers.append(("upsample_L%d_conv2 (concat)" % j, "%d->%d @ %d^2" % (2 * f, f, px), gf(px, 2 * f, f, 9)))
    layers.append(("upsample_L%d_conv3" % j, "%d->%d @ %d^2" % (f, f, px), gf(px, f, f, 9)))
CONV = ("conv_c8", "conv_ws", "conv_halo8", "conv_halo", "conv_deepk", "conv_pipe", "conv_glds")
def short(n):
    return n.replace("_kernel", "")
# split the sequence at head_forward: forward convs before, data gradients after
hf = next(i for i, r in enumerate(rows) if r[0].startswith(("head_forward", "head_bn_forward")))
tail = next(i for i, r in enumerate(rows) if r[0].startswith("wgrad_c8_kernel"))
def conv_groups(seq):
    out = []
    for name, grid, us, _ in seq:
        if name.startswith(CONV):
            out.append([short(name), us, 1])
        elif name.startswith("splitk_finish") and out:
            out[-1][0] += " + splitk_finish"; out[-1][1] += us; out[-1][2] += 1
    return out
fw = conv_groups(rows[:hf]); bw = conv_groups(rows[hf:tail])
assert len(fw) == 22 and len(bw) == 25, (len(fw), len(bw))
# backward order -> per layer (list of launches)
order = []
for j in (3, 2, 1, 0):
    order += ["upsample_L%d_conv3" % j, "upsample_L%d_conv2 (concat)" % j, "upsample_L%d_conv2 (concat)" % j, "upsample_L%d_conv1 (2x2 up)" % j]
order += ["bottom_conv2", "bottom_conv1"]
for i in (3, 2, 1):
    order += ["encoder_L%d_conv2" % i, "encoder_L%d_conv1" % i]
order += ["encoder_L0_conv2"]
dg = {}
for name, g in zip(order, bw):
    d = dg.setdefault(name, [[], 0.0])
    d[0].append(g[0]); d[1] += g[1]
print("| layer | shape (16 slices) | forward kernel | us | TFLOP/s | data-gradient kernel(s) | us | TFLOP/s |")
print("|---|---|---|---|---|---|---|---|")
tf = tb = gfw = gbw = 0.0
for (name, shape, g), f in zip(layers, fw):
    d = dg.get(name)
    dk = "--" if d is None else (" ; ".join(sorted(set(d[0]))) + (" x%d" % len(d[0]) if len(d[0]) > 1 else ""))
    print("| %s | %s | %s | %.1f | %.0f | %s | %s | %s |" % (name, shape, f[0], f[1], g / f[1] * 1e3, dk,
          "--" if d is None else "%.1f" % d[1], "--" if d is None else "%.0f" % (g / d[1] * 1e3)))
    tf += f[1]; gfw += g
    if d is not None: tb += d[1]; gbw += g
print("| **all 22 / 21 layers** | | | **%.0f** | **%.0f** | | **%.0f** | **%.0f** |" % (tf, gfw / tf * 1e3, tb, gbw / tb * 1e3))
# everything else of the step, by kernel
other = {}
for name, grid, us, _ in rows:
    if name.startswith(CONV) or name.startswith("splitk_finish"): continue
    k = short(name); o = other.setdefault(k, [0, 0.0]); o[0] += 1; o[1] += us
print()
print("| other launches of the step | count | us |")
print("|---|---|---|")
for k, (n, us) in sorted(other.items(), key=lambda kv: -kv[1][1]):
    print("| %s | %d | %.1f |" % (k, n, us))
