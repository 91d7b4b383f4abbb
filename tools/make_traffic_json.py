"""profiles/<tag>_pmc_traffic.json from the <tag>_pmc_traffic_*.txt passes of tools/profile_round.sh (bench.py reads it for
roofline.traffic).  FETCH_SIZE is doubled as MI355X_MICROARCH.md's HBM/rocprofv3 section prescribes for gfx950; both counters are
in KB.  usage: python tools/make_traffic_json.py r02"""
import glob, json, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_source_sha            # (path, sha256) of the csrc file behind a symbol: bench.py flags stale entries with it

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")

# bench_one.py case -> (symbol bench.py reports, shape text, algorithmic bytes per launch)
def conv_bytes(N, H, Ci, Co, b_in, b_out): return N * H * H * (Ci * b_in + Co * b_out) + 9 * Ci * Co * 2
CASES = {
    "halo_128_8_512_512_bf16": ("conv3x3_halo_kernel<128, 64, 3, false, 3, 8>", "[128,8,8,512]->512 bf16 storage", conv_bytes(128, 8, 512, 512, 2, 2)),
    "halo_128_32_128_128_bf16": ("conv3x3_halo_kernel<256, 64, 3, false, 3, 8>", "[128,32,32,128]->128 bf16 storage", conv_bytes(128, 32, 128, 128, 2, 2)),
    "halo_128_32_128_128_fp32": ("conv3x3_halo_kernel<256, 64, 3, false, 0, 8>", "[128,32,32,128]->128 fp32 storage", conv_bytes(128, 32, 128, 128, 4, 4)),
    "shift_128_8_512_512_bf16": ("conv_shift_kernel<4, 1, true>", "[128,8,8,512]->512 bf16 storage", conv_bytes(128, 8, 512, 512, 2, 2)),
    "shift_128_32_128_128_bf16": ("conv_shift_kernel<4, 2, true>", "[128,32,32,128]->128 bf16 storage", conv_bytes(128, 32, 128, 128, 2, 2)),
    "pw_128_8_512_512_bf16": ("conv_pw_kernel<true, 0, 0, 128>", "[128,8,8,512]->512 bf16 storage", conv_bytes(128, 8, 512, 512, 2, 2)),
    "pw_128_32_128_128_bf16": ("conv_pw_kernel<true, 0, 0, 128> [level 0]", "[128,32,32,128]->128 bf16 storage", conv_bytes(128, 32, 128, 128, 2, 2)),
    "pw_128_16_256_256_bf16": ("conv_pw_kernel<true, 0, 0, 128> [16x16]", "[128,16,16,256]->256 bf16 storage", conv_bytes(128, 16, 256, 256, 2, 2)),
    "fusedpw_128_32_128_128_bf16": ("conv_pw_kernel<true, 2, 0, 128>", "fused GN+Mish+conv [128,32,32,128]->128 bf16 storage (private-weight-stream kernel)",
                                    conv_bytes(128, 32, 128, 128, 2, 2) + 3 * 128 * 128 * 4),
    "fusedpw_128_32_128_128_fp32": ("conv_pw_kernel<false, 2, 0, 128, true>", "fused GN+Mish+conv [128,32,32,128]->128 fp32 storage (private-weight-stream kernel, round 4)",
                                    conv_bytes(128, 32, 128, 128, 4, 4) + 3 * 128 * 128 * 4),
    "gtdown_128_32_128_128_bf16": ("conv_gt_kernel<false, 128> [Downsample]", "Downsample 3x3 / stride 2 [128,32,32,128] bf16 -> [128,16,16,128] fp32 (tap-gather kernel)",
                                   128 * 32 * 32 * 128 * 2 + 128 * 16 * 16 * 128 * 4 + 9 * 128 * 128 * 2),
    "gtup_128_16_128_128_bf16": ("conv_gt_kernel<false, 128> [Upsample]", "Upsample ConvTranspose2d(4, 2, 1) [128,16,16,128] bf16 -> [128,32,32,128] fp32 (tap-gather kernel)",
                                 128 * 16 * 16 * 128 * 2 + 128 * 32 * 32 * 128 * 4 + 16 * 128 * 128 * 2),
    "wgrad_128_32_128_128_bf16": ("wgrad_tr_kernel[single]", "[128,32,32,128]x[128,32,32,128] bf16 operands, one layer per launch",
                                  2 * 128 * 32 * 32 * 128 * 2 + 9 * 128 * 128 * 4),
    "wgrad_128_8_512_512_bf16": ("wgrad_tr_kernel[single 8x8]", "[128,8,8,512]x[128,8,8,512] bf16 operands, one layer per launch",
                                 2 * 128 * 8 * 8 * 512 * 2 + 9 * 512 * 512 * 4),
    "wgradq_128_0_0_0_bf16": ("wgrad_tr_kernel", "the eight layers of backward's first group in one launch (tools/bench_one.py wgradq: 4x 128->128 @32x32, 3x 256->256 + 512->128 @16x16), bf16 operands",
                              sum(128 * h * h * (ci + co) * 2 + 9 * ci * co * 4
                                  for h, ci, co in [(32, 128, 128)] * 4 + [(16, 256, 256)] * 2 + [(16, 512, 128), (16, 256, 256)])),
    "fused_128_32_128_128_bf16": ("conv3x3_halo_kernel<256, 64, 3, false, 3, 8, true>", "fused GN+Mish+conv [128,32,32,128]->128 bf16 storage",
                                  conv_bytes(128, 32, 128, 128, 2, 2) + 3 * 128 * 128 * 4),
    "fused_128_32_128_128_fp32": ("conv3x3_halo_kernel<256, 64, 3, false, 0, 8, true>", "fused GN+Mish+conv [128,32,32,128]->128 fp32 storage",
                                  conv_bytes(128, 32, 128, 128, 4, 4) + 3 * 128 * 128 * 4),
}
MAIN = {"pw": "conv_pw_kernel", "fusedpw": "conv_pw_kernel", "gtdown": "conv_gt_kernel", "gtup": "conv_gt_kernel", "shift": "conv_shift_kernel", "halo": "conv3x3_halo_kernel", "fused": "conv3x3_halo_kernel", "wgrad": "wgrad_tr_kernel(", "wgradq": "wgrad_tr_kernel("}

# HBM-bound kernels (tools/bench_one.py gn / ln / attn / c1x1 at level 0, B = 128, bf16 storage): every kernel of the pass gets a row
E0 = 128 * 32 * 32 * 128
HBM_CASES = {
    "gn_128_32_128_128_bf16": {"gn_mish_fwd_kernel": ("GroupNorm+Mish fwd [128,32,32,128] bf16 -> bf16", E0 * 4),
                               "gn_mish_bwd_kernel": ("GroupNorm+Mish bwd, same tensor (x, dout in, dx out, bf16)", E0 * 6)},
    "ln_128_32_128_128_bf16": {"chan_ln_fwd_kernel": ("channel LayerNorm fwd [128,32,32,128] fp32 -> bf16", E0 * 6),
                               "chan_ln_bwd_kernel": ("channel LayerNorm bwd (x fp32, dy bf16 in; dx fp32 read-modify-write)", E0 * 14)},
    "attn_128_32_128_128_bf16": {"linattn_fwd_kernel": ("LinearAttention fwd, qkv [128,32,32,384] bf16 (k twice: max pass)", E0 * 2 * 5),
                                 "linattn_bwd_kernel": ("LinearAttention bwd (q, dout, then dout, k, v in; dq, dk, dv out; bf16)", E0 * 2 * 8)},
    "c1x1_128_32_128_384_bf16": {"conv1x1_pw_kernel": ("to_qkv 1x1 conv 128 -> 384 @32x32, bf16 in / out", E0 * 2 * 4 + 128 * 384 * 2)},
}

out = {}
for path in sorted(glob.glob(os.path.join(root, f"{tag}_pmc_traffic_*.txt"))):
    case = os.path.basename(path)[len(tag) + len("_pmc_traffic_"):-4]
    if case not in CASES:
        continue
    want = MAIN[case.split("_")[0]]
    vals = {}
    for line in open(path):
        m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE) per-dispatch.*:\s*(\d+)\s*$", line)
        if m and want in m.group(1) and "reduce" not in m.group(1):
            vals[m.group(2)] = int(m.group(3))
    if len(vals) != 2:
        continue
    sym, shape, alg = CASES[case]
    hbm = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
    src, sha = kernel_source_sha(sym)
    out[sym] = {"shape": shape, "source": src, "source_sha256": sha, "fetch_size_kb_raw": vals["FETCH_SIZE"], "write_size_kb_raw": vals["WRITE_SIZE"],
                "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": round(hbm / alg, 2),
                "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_traffic.sh) on {shape}; "
                        "FETCH_SIZE doubled per MI355X_MICROARCH.md"}
for case, kernels in HBM_CASES.items():
    path = os.path.join(root, f"{tag}_pmc_traffic_{case}.txt")
    if not os.path.exists(path):
        continue
    vals = {}
    for line in open(path):
        m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE) per-dispatch.*:\s*(\d+)\s*$", line)
        if m:
            for key in kernels:
                if key in m.group(1):
                    vals.setdefault(key, {})[m.group(2)] = int(m.group(3))
        m = re.match(r"(.*?)\s+DURATION_US\s+([0-9.]+)\s*$", line)
        if m:
            for key in kernels:
                if key in m.group(1):
                    vals.setdefault(key, {})["us"] = float(m.group(2))
    for key, (shape, alg) in kernels.items():
        v = vals.get(key, {})
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        hbm = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        extra = {}
        if "us" in v:
            extra = {"duration_us_under_pmc": v["us"], "hbm_gbs_from_pmc": round(hbm / v["us"] / 1e3, 1),
                     "algorithmic_gbs": round(alg / v["us"] / 1e3, 1), "hbm_frac_of_8tbs": round(hbm / v["us"] / 1e3 / 8000.0, 3)}
        src, sha = kernel_source_sha(key)
        out[f"{key} [{case}]"] = {"shape": shape, "source": src, "source_sha256": sha, "fetch_size_kb_raw": v["FETCH_SIZE"], "write_size_kb_raw": v["WRITE_SIZE"],
                                  "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": round(hbm / alg, 2), **extra,
                                  "note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_traffic.sh) on {shape}; "
                                          "FETCH_SIZE doubled per MI355X_MICROARCH.md"}
with open(os.path.join(root, f"{tag}_pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
for k, v in out.items():
    print(f"{k:60s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB  x{v['ratio']}")
