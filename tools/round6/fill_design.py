"""Regenerates the GENERATED blocks of DESIGN.md (between `<!-- BEGIN name -->` / `<!-- END name -->` markers) from the committed
profile artifacts of one tag. usage: python tools/round6/fill_design.py r06b        Dev / documentation tool (round 6).

blocks: step_summary, layer_table (tools/round6/layer_table.py on profiles/TAG_train_step_sequence.txt), predict_table
(profiles/TAG_predict_kernel_stats.txt), numbers (profiles/TAG_bench_line*.json, TAG_hbm_traffic_pmc.json)."""
import json, os, re, subprocess, sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
P = lambda n: os.path.join(R, "profiles", "%s_%s" % (tag, n))
line = json.load(open(P("bench_line.json")))
seq = open(P("train_step_sequence.txt")).read()
nlaunch = len(re.findall(r"^\s*\d+\s+\S+.*grid=", seq, re.M))
blocks = {}

rf, wg, pf, e2e = line["roofline"], line["wgrad"], line["predict_fuse"], line["train_e2e"]
blocks["step_summary"] = (
    "**Train step, configs[1]** (16 slices of 128 × 128 × 1, depth 4, 64 filters, bf16, Adam): **%d launches**, %.3f ms per step by\n"
    "`bench.py`'s clock on the box of `profiles/%s_bench_line.json` (launch form: %s)." % (nlaunch, line["ms_per_step"], tag, line["config"].get("launch", "?")))

blocks["layer_table"] = subprocess.run([sys.executable, os.path.join(R, "tools/round6/layer_table.py"), P("train_step_sequence.txt")],
                                       capture_output=True, text=True, check=True).stdout.strip()

# ---- predict table: classes of the kernel names ------------------------------------------------------------------------------------
CLASSES = [
    (r"conv_halo16p_kernelILb\dELb0E", "`conv_halo16p<·,false>` (persistent, 16-row tiles)", "3×3 layers with ≥ 128 filters, one source"),
    (r"conv_halo16p_kernelILb\dELb1E", "`conv_halo16p<·,true>` (same, two sources)", "the concat layers with ≥ 128 filters"),
    (r"conv_halo_kernelItLi64ELi8ELi3ELi0E", "`conv_halo<64,8>` 3×3", "the level-0 concat layer (128 → 64 at 256²)"),
    (r"conv_halo_kernelItLi\d+ELi\dELi3ELi1E", "`conv_halo<·,·>` up-conv form", "the four 2×2 up-convolutions"),
    (r"conv_glds_kernel", "`conv_glds<128,128,64,64>`", "the bottom layers (16² maps)"),
    (r"conv_ws_kernel", "`conv_ws<4>` (± fused max-pool)", "the 64-channel level-0 layers"),
    (r"conv_c8_kernel", "`conv_c8<2>`", "first layer"),
    (r"sample_fast_kernel|cast_pad_kernel|head_combine_kernel|fuse|map_", "`sample_fast`, `cast_pad`, `head_combine` (softmax + back-mapping + fusion)", "geometry: cutting the planes, mapping and fusing"),
]
tot, acc, n_pred = 0.0, {}, None
for l in open(P("predict_kernel_stats.txt")):
    m = re.match(r"(\S+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
    if not m or "probe_clock" in m.group(1):
        continue
    name, calls, us = m.group(1), int(m.group(2)), float(m.group(3))
    if "cast_pad" in name:
        n_pred = calls
    tot += us
    for i, (pat, _, _) in enumerate(CLASSES):
        if re.search(pat, name):
            acc[i] = acc.get(i, 0.0) + us
            break
    else:
        acc[-1] = acc.get(-1, 0.0) + us
batches = 18                                              # plane batches per predict at configs[2] (6 views x 3 batches of 92)
n_pred = (n_pred or batches * 7) / batches
rows = ["| kernel | serves | ms per predict | share |", "|---|---|---|---|"]
for i, (_, k, s) in enumerate(CLASSES):
    if i in acc:
        rows.append("| %s | %s | %.1f | %.1f %% |" % (k, s, acc[i] / n_pred / 1e3, 100 * acc[i] / tot))
if -1 in acc:
    rows.append("| other | | %.1f | %.1f %% |" % (acc[-1] / n_pred / 1e3, 100 * acc[-1] / tot))
rows.append("| **all kernels of a predict** | | **%.1f** | |" % (tot / n_pred / 1e3))
rows.append("")
rows.append("The U-Net of a predict runs at %.0f TFLOP/s algorithmic (%.3f of the bf16 peak; its large grids are power-bound at ≈ 1.6 PFLOP/s on\n"
            "random operands, A.4); a predict takes %.4f s by `bench.py`'s clock (median of %d)." %
            (pf["unet_tflops_algorithmic"], pf["unet_frac_of_mfma_peak"], pf["seconds"], pf["reps"]))
blocks["predict_table"] = "\n".join(rows)

# ---- numbers ---------------------------------------------------------------------------------------------------------------------------
def opt(n):
    try:
        return json.load(open(P(n)))
    except Exception:
        return None
c3, c4, tr = opt("bench_line_configs3.json"), opt("bench_line_configs4.json"), opt("hbm_traffic_pmc.json")
traffic = "not taken"
if tr:
    step_gb = sum(k["launches"] * k["hbm_MB_per_launch"] for k in tr["kernels"].values()) / 1e3
    steps = next(k["launches"] for n, k in tr["kernels"].items() if "head_forward" in n or "head_bn_forward" in n)      # one head per step
    traffic = "%.2f GB per step (conv family %.1f MB per launch, weight gradients %.0f MB per launch)" % (
        step_gb / steps, tr["classes"]["conv_igemm"]["hbm_bytes_per_launch"] / 1e6, tr["classes"]["wgrad_igemm"]["hbm_bytes_per_launch"] / 1e6)
cb = line["cpu_baseline"]
N = [
    ("train step, configs[1] (headline)", "**%.4f ms = %.0f slices/s** (launch form: %s; 2.21–2.32 ms across the boxes of the pool with the last build: `r06e` is a fast one, `r06g` a slow one; 2.27–2.42 before the last session)" % (line["ms_per_step"], line["value"], line["config"].get("launch")), "`%s_bench_line.json`" % tag),
    ("conv family (`roofline`)", "%.0f TFLOP/s = **%.3f** of peak, %.1f µs per launch, %.3f ms per step" % (rf["achieved"], rf["frac"], rf["avg_launch_us"], rf["kernel_ms_per_step"]), "same"),
    ("weight gradients (`wgrad`)", "%.0f TFLOP/s = %.3f of peak, %.3f ms per step" % (wg["achieved"], wg["frac"], wg["kernel_ms_per_step"]), "same"),
    ("HBM traffic of a step (PMC)", traffic, "`%s_hbm_traffic_pmc.json`" % tag),
    ("launches per step", "%d (round 5: 119)" % nlaunch, "`%s_train_step_sequence.txt`" % tag),
    ("`mp train` loop", "%.0f slices/s = %.3f of the headline" % (e2e["value"], e2e["fraction_of_headline"]), "`train_e2e`"),
    ("f32 parity mode / bf16x3", "%.2f ms / %.2f ms per step" % (line["f32_mode"]["ms_per_step"], line["bf16x3_mode"]["ms_per_step"]), "`f32_mode`, `bf16x3_mode`"),
    ("6-view predict + fuse, 256³ (configs[2])", "%.4f s = %.1f Mvox/s (U-Net %.0f TFLOP/s, map + fuse %.0f GB/s algorithmic)" % (pf["seconds"], pf["value"] / 1e6, pf["unet_tflops_algorithmic"], pf["map_fuse_GBs_algorithmic"]), "`predict_fuse`"),
]
if c3:
    N.append(("configs[3] train step (32 × 256², 1 GPU)", "%.2f ms = %.0f slices/s" % (c3["ms_per_step"], c3["value"]), "`%s_bench_line_configs3.json`" % tag))
if c4:
    N.append(("configs[4] predict (512³ × 2, K = 5)", "%.3f s = %.1f Mvox/s" % (c4["ms_per_step"] / 1e3, c4["value"] / 1e6), "`%s_bench_line_configs4.json`" % tag))
N.append(("CPU port of the step", "%.2f slices/s on %d cores" % (cb["value"], cb["cores"]), "`cpu_baseline`"))
blocks["numbers"] = "\n".join(["| quantity | value | source |", "|---|---|---|"] + ["| %s | %s | %s |" % r for r in N])

path = os.path.join(R, "DESIGN.md")
s = open(path).read()
for k, v in blocks.items():
    pat = re.compile(r"(<!-- BEGIN %s -->\n).*?(<!-- END %s -->)" % (k, k), re.S)
    assert pat.search(s), k
    s = pat.sub(lambda m: m.group(1) + v + "\n" + m.group(2), s)
open(path, "w").write(s)
print("DESIGN.md: blocks", ", ".join(blocks), "from profiles/%s_*" % tag)
