#!/usr/bin/env python3
"""Refresh the measured numbers in profiles/README.md and README.md from profiles/<tag>_bench.json and
profiles/<tag>_train_step_kernel_stats.csv (run after tools/profile_round.sh + copying its output into profiles/)."""
import collections, csv, json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
d = json.load(open(f"{root}/profiles/{tag}_bench.json")); r = d["roofline"]; nk = r["named_kernel"]
rows = list(csv.DictReader(open(f"{root}/profiles/{tag}_train_step_kernel_stats.csv")))
cat, avg = {}, {}
def add(k, v): cat[k] = cat.get(k, 0) + v
for x in rows:
    n = x["Name"]; ms = float(x["TotalDurationNs"]) / 1e7
    avg[n] = float(x["AverageNs"]) / 1e3
    m = re.search(r"conv3x3_halo_kernel<(\d+), (\d+), (\d)", n)
    if m: add("halo" + m.group(3), ms)
    elif "conv_shift_kernel" in n or "conv_dma_kernel" in n: add("halo3", ms)
    elif "wgrad_tr_reduce" in n: add("wtr_red", ms)
    elif "wgrad_tr_kernel" in n: add("wtr", ms)
    elif "wgrad1x1_tr_reduce" in n: add("w1_red", ms)
    elif "wgrad1x1_tr" in n: add("w1", ms)
    elif "wgrad_s2" in n: add("ws2", ms)
    elif "wgrad" in n: add("wother", ms)
    elif "gn_mish_fwd" in n or "gn_stats" in n: add("gnf", ms)
    elif "gn_mish_bwd" in n: add("gnb", ms)
    elif "linattn" in n: add("attn", ms)
    elif "chan_ln" in n: add("ln", ms)
    elif "igemm" in n: add("igemm", ms)
    elif "colsum" in n: add("colsum", ms)
    elif "adam" in n: add("adam", ms)
    elif "pack" in n: add("pack", ms)
    elif "f32_to_bf16" in n: add("cvt", ms)
    elif "small_c" in n or "partial_sum" in n: add("smallc", ms)
    elif "small_gemm" in n: add("sgemm", ms)
    else: add("other", ms)
c = {k: round(v, 2) for k, v in cat.items()}; tot = round(sum(cat.values()), 2)
p = f"{root}/profiles/README.md"; s = open(p).read()
def sub(pat, rep):
    global s
    s2 = re.sub(pat, rep, s, count=1, flags=re.S)
    assert s2 != s or re.search(pat, s, flags=re.S), pat
    s = s2
sub(r"\| train images/s \(1 GPU\) \| \*\*\d+\*\* \([\d.]+ ms/step, [\d.]+ TFLOP/s of algorithmic work = [\d.]+ %",
    f"| train images/s (1 GPU) | **{d['value']:.0f}** ({d['ms_per_step']} ms/step, {d['train_tflops']} TFLOP/s of algorithmic work = {d['train_tflops'] / 25:.1f} %")
sub(r"\| denoise steps/s at B=64 \(hipGraph\) \| [\d.]+ \([\d.]+ TFLOP/s\)", f"| denoise steps/s at B=64 (hipGraph) | {d['denoise_steps_per_sec']} ({d['denoise_tflops']} TFLOP/s)")
sub(r"fp32 storage\) \| [\d.]+ images/s", f"fp32 storage) | {d['fp32_mode']['value']} images/s")
sub(r"B=128 train step\) \| [\d.]+ images/s; one `p_sample` at B=64: [\d.]+ steps/s",
    f"B=128 train step) | {d['cpu_baseline']['value']} images/s; one `p_sample` at B=64: {d['cpu_baseline']['legs']['p_sample_b64']['value']} steps/s")
i = s.index("## Where a training step goes now"); j = s.index("## Dominant kernel (`roofline` in")
s = s[:i] + f"""## Where a training step goes now (ms/step, `{tag}_train_step_kernel_stats.csv`, {tot} ms under the profiler)

3x3 conv fwd + dgrad {c.get('halo3')} · 1x1 conv fwd + dgrad {c.get('halo1')} · 3x3 weight gradient {c.get('wtr')} + reduce {c.get('wtr_red')} (round 1: 1.30 + 0.40) ·
1x1 weight gradient {c.get('w1')} + {c.get('w1_red')} (0.54) · stride-2 weight gradients {c.get('ws2')} (0.37) · other weight gradients (3-channel ends,
Linear) {c.get('wother')} · stride-2 / transposed convs {c.get('igemm')} · GroupNorm+Mish fwd {c.get('gnf')} / bwd {c.get('gnb')} · LinearAttention {c.get('attn')} ·
LayerNorm {c.get('ln')} · Adam {c.get('adam')} · 3-channel ends {c.get('smallc')} · bias column sums + bf16 conversions {round(c.get('colsum', 0) + c.get('cvt', 0), 2)} · time-MLP GEMMs {c.get('sgemm')} ·
weight pack {c.get('pack')} (0.10) · the rest {c.get('other')}.

""" + s[j:]
ak = r["all_kernels"]
def prof_avg(sym):
    key = sym.split("<")[0]
    targs = sym[sym.index("<"):-1] if "<" in sym else ""
    for n, v in avg.items():
        if key in n and (not targs or targs[1:] in n): return v
    return float("nan")
traf = json.load(open(f"{root}/profiles/{tag}_pmc_traffic.json"))
top = sorted(ak.items(), key=lambda kv: -kv[1]["ms_per_step"])[:3]
def trow(sym, e):
    t = traf.get(sym)
    tr = (f"; canonical shape under the counters ({t['shape']}): {t['hbm_bytes_per_launch'] / 1e6:.1f} MB of HBM traffic vs "
          f"{t['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic = {t['ratio']}x") if t else ""
    return (f"* `{sym}`: {e['launches_per_step']} launches/step, {e['ms_per_step']} ms, {e['tflops']} TFLOP/s by events "
            f"({prof_avg(sym):.1f} us AverageNs in the rocprof summary){tr}")
i = s.index("## Dominant kernel (`roofline` in"); j = s.index("## The named kernel")
tline = f"HBM traffic {r['traffic'] / 1e6:.1f} MB per launch of its canonical shape (PMC, `{tag}_pmc_traffic.json`)." if r.get("traffic") else "No PMC pass for this symbol."
s = s[:i] + f"""## Dominant kernel (`roofline` in `{tag}_bench.json`)

`{r['kernel']}`: {r['launches_per_step']} launches/step, {r['avg_launch_us']} us/launch by HIP events ({prof_avg(r['kernel']):.1f} us AverageNs in
`{tag}_train_step_kernel_stats.csv`), {r['avg_gflop_per_launch']} GFLOP/launch => {r['achieved']} TFLOP/s = {100 * r['frac']:.1f} % of the dense bf16 MFMA peak;
{tline}
The top of the per-symbol table (`roofline.all_kernels`, HIP events around every launch of one instrumented step):
""" + "\n".join(trow(k, e) for k, e in top) + """

(`conv_shift_kernel<4, NI, out16>` = the 3x3 conv / data gradient of the bf16-stored layers, 8 waves, 256 pixels x 64*NI channels per workgroup;
`wgrad_tr_kernel` = eight Block-conv weight gradients per launch, LDS-DMA + transposing reads: canonical 8-layer launch 1.26 PFLOP/s,
`SQ_VALU_MFMA_BUSY_CYCLES` 60 %, `SQ_LDS_BANK_CONFLICT` 0; the register-staged `conv3x3_halo_kernel` now runs the 1x1 convs, the < 128-channel
layers and the dual-output / fused entry points.)

""" + s[j:]
f = nk["fp32_storage"]; b = nk["bf16_storage"]
i = s.index("| storage | two launches"); j = s.index("The unit is compute-bound")
def cell(e, key):
    x = e[key]
    if key == "fused": return f"{x['unit_us']} us (+ statistics {x['statistics_pass_us']}), hbm_frac {x['hbm_frac']}, mfma_frac {x['mfma_frac']}"
    if key == "fused_epilogue_stats": return f"{x['unit_us']} us, hbm_frac {x['hbm_frac']}; producer conv {x['producer_conv_us']} -> {x['producer_conv_with_sums_us']} us"
    return f"{x['unit_us']} us, hbm_frac {x['hbm_frac']}, mfma_frac {x['mfma_frac']}"
s = s[:i] + f"""| storage | two launches (GroupNorm kernel, conv) | fused (statistics pass + fused conv) | fused, statistics from the producing conv's epilogue |
|---|---|---|---|
| fp32 | {cell(f, 'two_pass')} | {cell(f, 'fused')} | {cell(f, 'fused_epilogue_stats')} |
| bf16 | {cell(b, 'two_pass')} | {cell(b, 'fused')} | {cell(b, 'fused_epilogue_stats')} |

""" + s[j:]
open(p, "w").write(s)
p = f"{root}/README.md"; s = open(p).read()
s = re.sub(r"bf16 block- and attention-internal tensors\): \d+ training images/s in the committed run \([\d.]+ ms/step;",
           f"bf16 block- and attention-internal tensors): {d['value']:.0f} training images/s in the committed run ({d['ms_per_step']} ms/step;", s)
s = re.sub(r"round 1: 16.4–16.9k\), \d+ hipGraph", f"round 1: 16.4–16.9k), {d['denoise_steps_per_sec']:.0f} hipGraph", s)
open(p, "w").write(s)
print("profiles/README.md, README.md refreshed:", d["value"], d["ms_per_step"], d["denoise_steps_per_sec"])
