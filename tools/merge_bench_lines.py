#!/usr/bin/env python3
"""bench.py prints two JSON lines since round 6 (detail, then the record).  This merges them back into ONE object with the layout of
gpurun_out/bench_detail.json (roofline.all_kernels / named_kernel inside roofline, fp32_mode, dp_path_n1, cfg3_b32 ... at the top level):
    python tools/merge_bench_lines.py two_lines.txt > merged.json"""
import json
import sys

lines = [ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")]
rec = json.loads(lines[-1])
det = next((json.loads(ln) for ln in lines[:-1] if '"detail"' in ln[:20]), {})
full = dict(rec)
roof = dict(rec.get("roofline") or {})
for k in ("all_kernels", "named_kernel", "named_kernel_cfg3_level0", "traffic_note"):
    if det.get(k) is not None:
        roof[k] = det[k]
full["roofline"] = roof
if rec.get("cpu_baseline") and det.get("cpu_baseline_legs"):
    full["cpu_baseline"] = dict(rec["cpu_baseline"], legs=det["cpu_baseline_legs"])
for k in ("fp32_mode", "dp_path_n1", "cfg3_b32", "comm", "rank_ms_per_step"):
    full[k] = det.get(k)
c = rec.get("config", {})
full.update({"denoise_steps_per_sec": c.get("denoise_steps_per_sec"), "denoise_tflops": c.get("denoise_tflops"), "train_tflops": c.get("train_tflops")})
print(json.dumps(full, indent=1))
