#!/usr/bin/env python3
"""Row-unit timeline of conv_pw_kernel (profiling build: conv_pw.hip compiled with -DMI_PW_TIMING, MI_DDPM_LIB=.../libmi_ddpm_pwt.so).
Usage: pw_timeline.py H Ci Co [tile] [out16].  Prints per-phase cycles (shader clock), the clock rate, per-unit durations."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-generation-models_amd")]
import numpy as np
import torch
from src.ops import functional as K
H, Ci, Co = (int(v) for v in sys.argv[1:4])
tile = int(sys.argv[4]) if len(sys.argv) > 4 else 128
out16 = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
B = int(os.environ.get("B", 128))
lib = K.load_library(); K.PW_MIN_TILES = 0
fn = lib.mi_debug_pw_ts; fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
lib.mi_debug_conv_pw_tile(tile)
x = torch.randn(B, H, H, Ci, device="cuda").bfloat16()
w = torch.randn(3, 3, Ci, Co, device="cuda") * 0.05
table, nent, tiles = K.pack_table([(0, 9, Ci, Co)], "cuda")
wd, wf, wdq, wfq = (torch.zeros(w.numel(), device="cuda", dtype=torch.bfloat16) for _ in range(4))
K.pack_weights_bf16(table, nent, tiles, w.reshape(-1), wd, wf, wdq, wfq)
y = torch.empty(B, H, H, Co, device="cuda", dtype=torch.bfloat16 if out16 else torch.float32)
run = lambda: K.conv3x3_bf16w(x, wf, K=Ci, Nc=Co, flip=False, out=y, wq=wfq)
for _ in range(10): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
buf = np.zeros(1024 * 4 * 5 * 64, dtype=np.uint32)
assert fn(buf.ctypes.data) == 0
nwg = min(1024, (B * H * H // tile) * ((Co + 127) // 128))
T = buf.reshape(1024, 4, 5, 64)[:nwg].astype(np.int64)
BH = tile // 32; U = BH + 1; nch = Ci // 64
aux = T[:, :, 4, :]
t_last, t_loop_end, t_end, t_entry, t_loop0 = (aux[..., i] for i in range(5))
hwid, xcc, wc0, wc1 = aux[..., 5], aux[..., 6], aux[..., 7], aux[..., 8]
d32 = lambda a, b: (a - b) & 0xffffffff
# point times: slot p holds time of point p-1 (p = (ch*4+ks)*U+u in program order); point -1 = loop entry
P = nch * 4 * U
pts = np.zeros((nwg, 4, P + 1), dtype=np.int64)        # index 0 = loop entry, 1.. = points
for ch in range(nch):
    for ks in range(4):
        for u in range(U):
            p = (ch * 4 + ks) * U + u
            pts[:, :, p] = T[:, :, ks, ch * U + u]
pts[:, :, P] = t_last
dur = d32(pts[:, :, 1:], pts[:, :, :-1])                  # dur[p] = time(point p) - time(point p-1): p=0 -> loop entry..first stamp
unit = np.zeros((nwg, 4, P), dtype=np.int64)
unit[:, :, :-1] = dur[:, :, 1:]                           # unit p lasts from point p to point p+1
unit[:, :, -1] = d32(t_loop_end, t_last)
cyc_total = d32(t_end, t_entry); wall = d32(wc1, wc0) * 10.0      # ns
print(f"{H}x{H} {Ci}->{Co} tile {tile} out16 {out16}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {nwg} workgroups, {nch} chunks")
print(f"  clock: {np.median(cyc_total / np.maximum(wall, 1)):.2f} GHz (cycles / wall per wave, median); wave lifetime {np.median(wall)/1e3:.2f} us (min {wall.min()/1e3:.2f} max {wall.max()/1e3:.2f})")
w0 = wc0.min()
print(f"  wave start spread {d32(wc0, w0).max()*10/1e3:.2f} us; last end {d32(wc1, w0).max()*10/1e3:.2f} us after the first start")
t_loop0 = pts[:, :, 0]
pro, loop, epi = d32(t_loop0, t_entry), d32(t_loop_end, t_loop0), d32(t_end, t_loop_end)
print(f"  cycles per wave (median): prologue {np.median(pro):.0f}  main loop {np.median(loop):.0f}  epilogue {np.median(epi):.0f};  MFMA-only floor of the loop {nch*4*9*BH*32} ({nch*4*9*BH*32/np.median(loop)*100:.0f} % of it)")
m = lambda a, b: np.median(d32(aux[..., a], aux[..., b]))
print(f"  prologue: entry -> first request {m(10, 3):.0f}, -> own loads landed {m(11, 10):.0f}, -> barrier passed {m(12, 11):.0f};  epilogue: loop end -> barrier {m(13, 1):.0f}, -> tile in LDS + sync {m(14, 13):.0f}, -> stores issued {m(15, 14):.0f}, -> stores landed {m(2, 15):.0f}")
um = unit.reshape(nwg, 4, nch, 4, U)
print("  unit durations (cycles, median over workgroups and waves) [chunk][step]: units 0..BH; ideal per unit: unit 0 = 192, others " + str(9 * 32) + " (x waves per SIMD)")
for ch in range(nch):
    print("   ch%d " % ch + " | ".join(" ".join(f"{np.median(um[:, :, ch, ks, u]):5.0f}" for u in range(U)) for ks in range(4)))
# CU sharing: how many workgroups per (xcc, se, sh?, cu)
cu = (xcc[:, 0] & 0xf) * 4096 + ((hwid[:, 0] >> 8) & 0xf) + ((hwid[:, 0] >> 12) & 0x1) * 16 + ((hwid[:, 0] >> 13) & 0x7) * 32
uniq, cnt = np.unique(cu, return_counts=True)
print(f"  distinct CUs {len(uniq)}; workgroups per CU: min {cnt.min()} max {cnt.max()}")
# start order on one CU
if cnt.max() > 1:
    c = uniq[np.argmax(cnt)]
    idx = np.where(cu == c)[0]
    print("  one CU's workgroups (start us, end us):", [(round(float(d32(wc0[i, 0], w0)) * 0.01, 2), round(float(d32(wc1[i, 0], w0)) * 0.01, 2)) for i in idx])
np.save(os.environ.get("PW_TS_OUT", "/tmp/pw_ts.npy"), T)
