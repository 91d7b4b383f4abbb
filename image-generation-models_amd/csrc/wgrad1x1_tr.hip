// Weight gradient of the 1x1 convolutions (LinearAttention's to_qkv / to_out and ResnetBlock's res_conv, reference
// src/models/ddpm.py:134,151-152):   dW[ci][co] += sum_p X[p][ci] * dY[p][co]   (+ dbias[co] += sum_p dY[p][co])
// A [Ci x Cj] output from a contraction over all N*H*W pixels: 0.4-13 GFLOP against 50-270 MB of operands, i.e. bound by HBM, and
// with a tiny output a full-chip launch per layer drowns in partial tiles (256 k-slices x the whole output).  So, like the 3x3
// kernel of wgrad_tr.hip: several layers share one launch, each on its share of the workgroups (few k-slices per layer), operands
// go L2 -> LDS by LDS-DMA in the layout the transposing ds_read_b64_tr_b16 wants ([pixel block][32-channel group][pixel][32 ch]),
// and the pixel axis is streamed in steps of 64 pixels through a ring that keeps PF steps in flight (counted s_waitcnt vmcnt).
// X is always bf16-stored (LayerNorm output, attention output, the bf16 copy of a ResnetBlock input).  dY is bf16 (d qkv) or the
// fp32 residual-stream gradient: fp32 rows are DMA'd raw and converted LDS -> LDS by the workgroup (v_cvt_pk_bf16_f32), which
// is also where the bias gradient is summed.  One workgroup (8 waves) = a 64 x 128 (ci x co) tile, a wave = one 32 x 32 MFMA tile.
#include "tr_common.h"

namespace {

struct W1Args {
    const uint16_t* P; const uint16_t* P2; const void* Q;
    float* ws; float* dW; float* dbias;
    int Ci, Cj, I1, ldp, ldp2, ldq, q32;
    int total, sps, splits, gx, gy, wg0, tile0, xcd_map;
};
struct W1Batch { W1Args p[MAXP]; int n; };

// PF steps are requested ahead of the one being multiplied (3 for bf16 dY, 2 when the raw fp32 rows take 32 KB per step); a step is
// requested right AFTER the barrier that ends the previous step's reads, into the slot that step just freed: ring = PF + 1 slots
constexpr int XSTEP = 64 * 64 * 2;    // 64 pixels x 64 channels bf16
constexpr int YSTEP = 64 * 128 * 2;   // 64 pixels x 128 channels bf16
constexpr int YRAW = 64 * 128 * 4;    // ... as fp32 rows

template <bool Q32>
__device__ __forceinline__ void wgrad1_body(const W1Args& a, const int wg, uint8_t* lds_raw) {
    // LDS: [X ring][dY ring (bf16, Q32: one converted buffer)][Q32: raw fp32 ring]
    constexpr int PF = Q32 ? 2 : 3, RING = PF + 1;
    constexpr int YOFF = RING * XSTEP;
    constexpr int RAWOFF = YOFF + YSTEP;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wv >> 2, wj = wv & 3;
    const int ntiles = a.gx * a.gy;
    // k-slice and tile.  The tiles of one k-slice read the same X / dY rows, and consecutive workgroup ids go to different XCDs (id % 8),
    // each with its own L2: rank the problem's workgroups by (XCD, order on that XCD) and hand out (slice, tile) in rank order, so
    // that a slice's tiles sit on one XCD and its rows come from HBM once (this kernel is HBM-bound: 2.2x less traffic for to_qkv).
    int rank = wg;
    if (a.xcd_map) {
        const int W = ntiles * a.splits, x = (a.wg0 + wg) & 7;
        rank = (wg - ((x - a.wg0) & 7)) >> 3;
        for (int xx = 0; xx < x; ++xx) rank += (W - ((xx - a.wg0) & 7) + 7) >> 3;
    }
    const int split = rank / ntiles, tile = rank - split * ntiles;
    const int ci0 = (tile % a.gx) * 64, co0 = (tile / a.gx) * 128;
    const int sb = split * a.sps, se = min(a.total, sb + a.sps);
    const int last = a.total - 1;

    // DMA sources (lane -> piece as in wgrad_tr.hip); fp32 dY rows: 2 pixels of 128 channels per wave instruction, lane -> pixel
    // lane >> 5, 4-channel piece lane & 31
    const bool second = ci0 >= a.I1;
    const int ldx = second ? a.ldp2 : a.ldp;
    const uint16_t* xsrc = (second ? a.P2 : a.P) + (size_t)((l >> 2) & 7) * ldx +
                           min((second ? ci0 - a.I1 : ci0) + (l >> 5) * 32 + (l & 3) * 8, (second ? a.Ci - a.I1 : a.I1) - 8);
    const uint16_t* ysrc16 = reinterpret_cast<const uint16_t*>(a.Q) + (size_t)((l >> 2) & 3) * a.ldq + min(co0 + (l >> 4) * 32 + (l & 3) * 8, a.Cj - 8);
    const float* ysrc32 = reinterpret_cast<const float*>(a.Q) + (size_t)(l >> 5) * a.ldq + min(co0 + (l & 31) * 4, a.Cj - 4);
    auto stage = [&](int step) {
        const size_t pix0 = (size_t)min(step, last) * 64;
        const int slot = step % RING;
        glds16(xsrc + (pix0 + wv * 8) * ldx, lds0 + slot * XSTEP + wv * 1024);                    // 8 X blocks of 8 pixels, one per wave
        if constexpr (Q32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                                            // 32 pixel pairs, four per wave
                const int i = wv + 8 * k;
                glds16(ysrc32 + (pix0 + i * 2) * a.ldq, lds0 + RAWOFF + slot * YRAW + i * 1024);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {                                                            // 16 dY blocks of 4 pixels, two per wave
                const int i = wv + 8 * k;
                glds16(ysrc16 + (pix0 + i * 4) * a.ldq, lds0 + YOFF + slot * YSTEP + i * 1024);
            }
        }
    };
    constexpr int PER_STEP = Q32 ? 5 : 3;                // DMA instructions per wave and step

    const int half = l >> 5, psub = (l & 15) >> 2;
    const int lane_b = ((l >> 4) & 1) * 32 + (l & 3) * 8;
    const int fa = half * 1024 + psub * 64 + wi * 512 + lane_b;           // X: 8-pixel block 2j + half, pixel 4r + psub
    const int fb = half * 2048 + psub * 64 + wj * 256 + lane_b;           // dY: 4-pixel block 4j + 2*half + r
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // convert pass (Q32): thread -> 4-channel piece cq = t & 31, pixels (t >> 5) + 16k
    const int cq = t & 31, pg = t >> 5;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = Q32 && a.dbias != nullptr && ci0 == 0;

    static_for<0, PF>([&](auto kc) { stage(sb + decltype(kc)::value); });
    for (int s = sb; s < se; ++s) {
        // all but the PF - 1 newest steps have landed (this wave's pieces; the barrier makes it every wave's) ...
        if constexpr (Q32) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        static_assert(PER_STEP * (PF - 1) == (Q32 ? 5 : 6), "vmcnt immediates above");
        __builtin_amdgcn_s_barrier();                 // ... and every wave is done with step s-1, whose slot the next request refills
        asm volatile("" ::: "memory");
        stage(s + PF);
        const int slot = s % RING;
        uint32_t yb;
        if constexpr (Q32) {
            const uint8_t* raw = lds_raw + RAWOFF + slot * YRAW;
            uint8_t* dst = lds_raw + YOFF;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int px = pg + 16 * k;
                const f32x4 v = *reinterpret_cast<const f32x4*>(raw + px * 512 + cq * 16);
                bsum += v;
                *reinterpret_cast<uint2*>(dst + (px >> 2) * 1024 + (cq >> 3) * 256 + (px & 3) * 64 + (cq & 7) * 8) =
                    make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
            }
            __syncthreads();
            yb = lds0 + YOFF;
        } else {
            yb = lds0 + YOFF + slot * YSTEP;
        }
        const uint32_t xb = lds0 + slot * XSTEP;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 af = tr_pair(xb + fa + j * 2048, xb + fa + j * 2048 + 256);
            const bf16x8 bf = tr_pair(yb + fb + j * 4096, yb + fb + j * 4096 + 1024);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (do_bias) {
        // lanes l and l + 32 hold the same channel piece (pixels pg, pg + ... of different parity); then one atomic per wave and piece
        bsum.x += __shfl_xor(bsum.x, 32, 64); bsum.y += __shfl_xor(bsum.y, 32, 64);
        bsum.z += __shfl_xor(bsum.z, 32, 64); bsum.w += __shfl_xor(bsum.w, 32, 64);
        const int c = co0 + cq * 4;
        if (l < 32 && c < a.Cj) {
            atomicAdd(a.dbias + c, bsum.x); atomicAdd(a.dbias + c + 1, bsum.y);
            atomicAdd(a.dbias + c + 2, bsum.z); atomicAdd(a.dbias + c + 3, bsum.w);
        }
    }
    if (a.splits == 1) {
        const int ci = ci0 + wi * 32 + 4 * (l >> 5), co = co0 + wj * 32 + (l & 31);
        if (co < a.Cj) {
            float* o = a.dW + (size_t)ci * a.Cj + co;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * a.Cj] += acc[r];
        }
        return;
    }
    float* out = a.ws + (size_t)(split * ntiles + tile) * (4 * 2048) + t * 4;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<f32x4*>(out + rq * 2048) = f32x4{acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]};
}

__global__ __launch_bounds__(512, 1) void wgrad1x1_tr_kernel(const W1Batch b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    int p = 0;
#pragma unroll
    for (int q = 1; q < MAXP; ++q)
        if (q < b.n && (int)blockIdx.x >= b.p[q].wg0) p = q;
    const W1Args& a = b.p[p];
    const int wg = blockIdx.x - a.wg0;
    if (a.q32) wgrad1_body<true>(a, wg, lds_raw); else wgrad1_body<false>(a, wg, lds_raw);
}

// dW[ci][co] += sum over k-slices (fixed order); grid = (4 position groups, 4 register quads, tiles with k-slices), 256 threads =
// 2 slice groups x 128 float4 positions
__global__ __launch_bounds__(256) void wgrad1x1_tr_reduce_kernel(const W1Batch b) {
    __shared__ f32x4 red[128];
    int pi = 0;
#pragma unroll
    for (int q = 1; q < MAXP; ++q)
        if (q < b.n && b.p[q].splits > 1 && (int)blockIdx.z >= b.p[q].tile0) pi = q;
    const W1Args& a = b.p[pi];
    const int ntiles = a.gx * a.gy, splits = a.splits;
    const int it = threadIdx.x & 127, grp = threadIdx.x >> 7;
    const int tt = blockIdx.x * 128 + it;
    const int rq = blockIdx.y, tile = blockIdx.z - a.tile0;
    const float* p = a.ws + (size_t)tile * (4 * 2048) + (size_t)rq * 2048 + tt * 4;
    const size_t stride = (size_t)ntiles * (4 * 2048);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int sp = grp;
    for (; sp + 2 < splits; sp += 4) {
        s0 += *reinterpret_cast<const f32x4*>(p + (size_t)sp * stride);
        s1 += *reinterpret_cast<const f32x4*>(p + (size_t)(sp + 2) * stride);
    }
    for (; sp < splits; sp += 2) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)sp * stride);
    f32x4 s = s0 + s1;
    if (grp == 1) red[it] = s;
    __syncthreads();
    if (grp != 0) return;
    s += red[it];
    const int wv = tt >> 6, l = tt & 63;
    const int ci = (tile % a.gx) * 64 + (wv >> 2) * 32 + 8 * rq + 4 * (l >> 5);
    const int co = (tile / a.gx) * 128 + (wv & 3) * 32 + (l & 31);
    if (co >= a.Cj) return;
    float* out = a.dW + (size_t)ci * a.Cj + co;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (ci + e < a.Ci) out[(size_t)e * a.Cj] += s[e];
}

bool w1_ok(const MiWgradDesc* d, int q32) {
    if (d->KH != 1 || d->KW != 1 || d->pad != 0 || d->stride != 1 || !d->gather_i || d->mode != 1) return false;
    if (d->GH != d->DH || d->GW != d->DW) return false;
    if (((long)d->N * d->DH * d->DW) % 64) return false;
    if (d->Ci % 64 || d->I1 % 64 || d->Cj % 32 || d->Cj < 32) return false;
    if (d->ldp % 8 || (d->I1 != d->Ci && d->ldp2 % 8)) return false;
    return q32 ? d->ldq % 4 == 0 : d->ldq % 8 == 0;
}

int g_w1_phase = 0, g_w1_blocks = 0;

void w1_plan(const MiWgradDesc* d, W1Args& a, long wgs) {
    a.Ci = d->Ci; a.Cj = d->Cj; a.I1 = d->I1;
    a.gx = d->Ci / 64; a.gy = (d->Cj + 127) / 128;
    a.total = (int)((long)d->N * d->DH * d->DW / 64);
    const int ntiles = a.gx * a.gy;
    long splits = wgs / ntiles;
    if (splits < 1) splits = 1;
    if (splits > a.total) splits = a.total;
    a.sps = (int)((a.total + splits - 1) / splits);
    a.splits = (a.total + a.sps - 1) / a.sps;
}

// workgroups per problem in proportion to the bytes it streams (memory-bound), every problem at least its tiles
void w1_shares(int n, const MiWgradDesc* d, const int* q32, long* wgs) {
    double tot = 0, by[MAXP];
    for (int i = 0; i < n; ++i) {
        const double tiles_ci = d[i].Ci / 64, tiles_co = (d[i].Cj + 127) / 128;
        by[i] = (double)d[i].N * d[i].DH * d[i].DW * (tiles_co * d[i].Ci * 2.0 + tiles_ci * d[i].Cj * (q32[i] ? 4.0 : 2.0));
        tot += by[i];
    }
    const long target = g_w1_blocks > 0 ? g_w1_blocks : 256;
    static const int greedy = (int)mi_knob("MI_W1_BALANCE", 1);
    if (greedy) {               // whole k-slices to whoever carries the most bytes per workgroup (tr_common.h)
        long tiles[MAXP];
        for (int i = 0; i < n; ++i) tiles[i] = (long)(d[i].Ci / 64) * ((d[i].Cj + 127) / 128);
        balance_shares(n, by, tiles, target, wgs);
        return;
    }
    for (int i = 0; i < n; ++i) {
        const long tiles = (long)(d[i].Ci / 64) * ((d[i].Cj + 127) / 128);
        long w = (long)(target * by[i] / tot + 0.5);
        w = w / tiles * tiles;
        wgs[i] = w < tiles ? tiles : w;
    }
}

size_t w1_ws_floats(const W1Args& a) { return a.splits > 1 ? (size_t)a.splits * a.gx * a.gy * 4 * 2048 : 0; }

}  // namespace

extern "C" int mi_conv1x1_wgrad_tr_supported(const MiWgradDesc* d, int q_is_fp32) { return (d && w1_ok(d, q_is_fp32)) ? 1 : 0; }

extern "C" size_t mi_conv1x1_wgrad_tr_batch_workspace(int n, const MiWgradDesc* descs, const int* q_is_fp32) {
    if (!descs || !q_is_fp32 || n < 1 || n > MAXP) return 0;
    long wgs[MAXP];
    for (int i = 0; i < n; ++i) if (!w1_ok(&descs[i], q_is_fp32[i])) return 0;
    w1_shares(n, descs, q_is_fp32, wgs);
    size_t fl = 0;
    for (int i = 0; i < n; ++i) { W1Args a; w1_plan(&descs[i], a, wgs[i]); fl += w1_ws_floats(a); }
    return fl * sizeof(float) + 256;
}

extern "C" int mi_debug_wgrad1x1_tr_phase(int phase) {
    if (phase < 0 || phase > 2) return mi_set_error(-1, "mi_debug_wgrad1x1_tr_phase: phase in 0..2");
    g_w1_phase = phase;
    return 0;
}
extern "C" int mi_debug_wgrad1x1_tr_blocks(int blocks) { g_w1_blocks = blocks > 0 ? blocks : 0; return 0; }

extern "C" int mi_conv1x1_wgrad_tr_batch(int n, const MiWgradDesc* descs, const int* q_is_fp32, const void* const* P,
                                         const void* const* P2, const void* const* Q, float* const* dW, float* const* dbias,
                                         void* workspace, size_t ws_bytes, void* stream) {
    MI_REQUIRE(n >= 1 && n <= MAXP && descs && q_is_fp32 && P && Q && dW, "1..8 problems, non-null arrays");
    W1Batch b;
    b.n = n;
    long wgs[MAXP];
    bool any32 = false;
    for (int i = 0; i < n; ++i) {
        MI_REQUIRE(w1_ok(&descs[i], q_is_fp32[i]), "descriptor not supported by the LDS-DMA 1x1 weight-gradient kernel");
        MI_REQUIRE(P[i] && Q[i] && dW[i], "null operand");
        MI_REQUIRE(descs[i].I1 == descs[i].Ci || (P2 && P2[i]), "two-source split without P2");
        MI_REQUIRE((((uintptr_t)P[i] | (uintptr_t)Q[i] | (uintptr_t)((P2 && P2[i]) ? P2[i] : P[i])) & 15) == 0, "operands must be 16-byte aligned");
        MI_REQUIRE(!(dbias && dbias[i]) || q_is_fp32[i], "the bias gradient is summed in the fp32 conversion pass (fp32 dY only)");
        any32 = any32 || q_is_fp32[i];
    }
    w1_shares(n, descs, q_is_fp32, wgs);
    size_t off = 0;
    int wg = 0, tile = 0;
    for (int i = 0; i < n; ++i) {
        W1Args& a = b.p[i];
        w1_plan(&descs[i], a, wgs[i]);
        a.P = (const uint16_t*)P[i]; a.P2 = (const uint16_t*)((P2 && P2[i]) ? P2[i] : P[i]); a.Q = Q[i];
        a.dW = dW[i]; a.dbias = dbias ? dbias[i] : nullptr; a.q32 = q_is_fp32[i] ? 1 : 0;
        a.ldp = descs[i].ldp; a.ldp2 = (P2 && P2[i]) ? descs[i].ldp2 : descs[i].ldp; a.ldq = descs[i].ldq;
        a.ws = (float*)workspace + off;
        off += w1_ws_floats(a);
        static const int xcd_env = (int)mi_knob("MI_W1_XCD", 1);
        a.xcd_map = xcd_env && a.gx * a.gy > 1 && a.splits > 1;
        a.wg0 = wg; wg += a.gx * a.gy * a.splits;
        a.tile0 = tile; if (a.splits > 1) tile += a.gx * a.gy;
    }
    MI_REQUIRE(off == 0 || (workspace && ((uintptr_t)workspace & 15) == 0 && ws_bytes >= off * sizeof(float)),
               "workspace too small (mi_conv1x1_wgrad_tr_batch_workspace)");
    hipStream_t st = (hipStream_t)stream;
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)wgrad1x1_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)once;
    const size_t lds16 = (size_t)4 * XSTEP + (size_t)4 * YSTEP, lds32 = (size_t)3 * XSTEP + YSTEP + (size_t)3 * YRAW;
    const size_t lds = any32 && lds32 > lds16 ? lds32 : lds16;
    if (g_w1_phase != 2) hipLaunchKernelGGL(wgrad1x1_tr_kernel, dim3((unsigned)wg), dim3(512), lds, st, b);
    if (g_w1_phase != 1 && tile > 0) hipLaunchKernelGGL(wgrad1x1_tr_reduce_kernel, dim3(4, 4, tile), dim3(256), 0, st, b);
    MI_LAUNCH_CHECK();
    return 0;
}
