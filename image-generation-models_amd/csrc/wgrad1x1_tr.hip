// Weight gradient of the 1x1 convolutions (LinearAttention's to_qkv / to_out and ResnetBlock's res_conv, reference
// src/models/ddpm.py:134,151-152):   dW[ci][co] += sum_p X[p][ci] * dY[p][co]   (+ dbias[co] += sum_p dY[p][co])
// A [Ci x Cj] output from a contraction over all N*H*W pixels: 0.4-13 GFLOP against 50-270 MB of operands, i.e. bound by HBM, and
// with a tiny output a full-chip launch per layer drowns in partial tiles (256 k-slices x the whole output).  So, like the 3x3
// kernel of wgrad_tr.hip: several layers share one launch, each on its share of the workgroups (few k-slices per layer), operands
// go L2 -> LDS by LDS-DMA in the layout the transposing ds_read_b64_tr_b16 wants ([pixel block][32-channel group][pixel][32 ch]),
// and the pixel axis is streamed in steps of 64 pixels through a ring that keeps PF steps in flight (counted s_waitcnt vmcnt).
// X is always bf16-stored (LayerNorm output, attention output, the bf16 copy of a ResnetBlock input).  dY is bf16 (d qkv) or the
// fp32 residual-stream gradient: fp32 rows are DMA'd raw and converted LDS -> LDS by the workgroup (v_cvt_pk_bf16_f32), which
// is also where the bias gradient is summed.  One workgroup (8 waves) = a (64 NI) x 128 (ci x co) tile, a wave = NI 32 x 32 MFMA tiles
// that share the dY fragment.  Round 4: NI = 2 wherever Ci % 128 == 0 -- the kernel is bound by the bytes the CUs pull through
// L2 (every ci tile re-reads the layer's dY rows, every co tile its X rows), and a 128-wide ci tile halves the dY re-reads: to_qkv
// 128 -> 384 moves 1536 instead of 2304 bytes per pixel, to_out 128 -> 128 (fp32 dY) 768 = its algorithmic bytes instead of 1280.
// fp32 dY rows are no longer converted LDS -> LDS: a lane reads the 8 pixels of its output column straight from the raw rows
// (8 conflict-free ds_read_b32: 32 lanes = 32 consecutive channels of one pixel), rounds them with four v_cvt_pk_bf16_f32 and has
// its MFMA B fragment -- no second buffer, no second barrier per step, and the bias gradient is the sum of the same registers.
#include "tr_common.h"

namespace {

struct W1Args {
    const uint16_t* P; const uint16_t* P2; const void* Q;
    float* ws; float* dW; float* dbias;
    int Ci, Cj, I1, ldp, ldp2, ldq, q32, ni, f32;
    int total, sps, splits, gx, gy, wg0, tile0, xcd_map;
    // round 6, exact-fp32 mode only: ONE TAP of a stride-2 / transposed conv's weight gradient as a gathered 1x1 problem -- the pixels m of the
    // small (dense) grid [N][1 << (lghw - lgw)][1 << lgw] meet pixel (2 y + gdy, 2 x + gdx) of the big [N][GH][GW] tensor (zero outside)
    int gmode;          // 0: plain 1x1; 1: P is the big (gathered) tensor; 2: Q is
    int gdy, gdx, lgw, lghw, GH, GW;
};
__device__ __attribute__((aligned(16))) uint32_t g_w1_zero[128];      // 512 zero bytes: what a tap reads outside the big tensor
struct W1Batch { W1Args p[MAXP]; int n; };

// PF steps are requested ahead of the one being multiplied (3 for bf16 dY, 2 when the raw fp32 rows take 32 KB per step); a step is
// requested right AFTER the barrier that ends the previous step's reads, into the slot that step just freed: ring = PF + 1 slots
constexpr int XSTEP = 64 * 64 * 2;    // 64 pixels x 64 channels bf16
constexpr int YSTEP = 64 * 128 * 2;   // 64 pixels x 128 channels bf16
constexpr int YRAW = 64 * 128 * 4;    // ... as fp32 rows

template <bool Q32, int NI>
__device__ __forceinline__ void wgrad1_body(const W1Args& a, const int wg, uint8_t* lds_raw) {
    // LDS: [X ring: RING x NI x 8 KB][dY ring: bf16 16 KB per step, or the raw fp32 rows 32 KB per step]
    constexpr int PF = Q32 ? 2 : 3, RING = PF + 1;
    constexpr int XSLOT = NI * XSTEP;
    constexpr int YOFF = RING * XSLOT;
    constexpr int YSLOT = Q32 ? YRAW : YSTEP;
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
    const int ci0 = (tile % a.gx) * (64 * NI), co0 = (tile / a.gx) * 128;
    const int sb = split * a.sps, se = min(a.total, sb + a.sps);
    const int last = a.total - 1;

    // DMA sources (lane -> piece as in wgrad_tr.hip); fp32 dY rows: 2 pixels of 128 channels per wave instruction, lane -> pixel
    // lane >> 5, 4-channel piece lane & 31.  The NI 64-channel regions of the ci tile are staged independently (each may lie in
    // either source of a two-source layer: I1 % 64 == 0).
    const uint16_t* xsrc[NI];
    int ldx[NI];
#pragma unroll
    for (int r = 0; r < NI; ++r) {
        const int cr = ci0 + 64 * r;
        const bool second = cr >= a.I1;
        ldx[r] = second ? a.ldp2 : a.ldp;
        xsrc[r] = (second ? a.P2 : a.P) + (size_t)((l >> 2) & 7) * ldx[r] +
                  min((second ? cr - a.I1 : cr) + (l >> 5) * 32 + (l & 3) * 8, (second ? a.Ci - a.I1 : a.I1) - 8);
    }
    const uint16_t* ysrc16 = reinterpret_cast<const uint16_t*>(a.Q) + (size_t)((l >> 2) & 3) * a.ldq + min(co0 + (l >> 4) * 32 + (l & 3) * 8, a.Cj - 8);
    const float* ysrc32 = reinterpret_cast<const float*>(a.Q) + (size_t)(l >> 5) * a.ldq + min(co0 + (l & 31) * 4, a.Cj - 4);
    auto stage = [&](int step) {
        const size_t pix0 = (size_t)min(step, last) * 64;
        const int slot = step % RING;
#pragma unroll
        for (int r = 0; r < NI; ++r)                                                                 // 8 X blocks of 8 pixels per region, one per wave
            glds16(xsrc[r] + (pix0 + wv * 8) * ldx[r], lds0 + slot * XSLOT + r * XSTEP + wv * 1024);
        if constexpr (Q32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                                            // 32 pixel pairs, four per wave
                const int i = wv + 8 * k;
                glds16(ysrc32 + (pix0 + i * 2) * a.ldq, lds0 + YOFF + slot * YRAW + i * 1024);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {                                                            // 16 dY blocks of 4 pixels, two per wave
                const int i = wv + 8 * k;
                glds16(ysrc16 + (pix0 + i * 4) * a.ldq, lds0 + YOFF + slot * YSTEP + i * 1024);
            }
        }
    };
    constexpr int PER_STEP = NI + (Q32 ? 4 : 2);                // DMA instructions per wave and step

    const int half = l >> 5, psub = (l & 15) >> 2;
    const int lane_b = ((l >> 4) & 1) * 32 + (l & 3) * 8;
    const int fa = half * 1024 + psub * 64 + wi * 512 + lane_b;           // X: 8-pixel block 2j + half, pixel 4r + psub
    const int fb = half * 2048 + psub * 64 + wj * 256 + lane_b;           // dY: 4-pixel block 4j + 2*half + r
    const int fr = half * (8 * 512) + (wj * 32 + (l & 31)) * 4;           // raw fp32 dY: pixel 16j + 8*half + e, this lane's output column
    f32x16 acc[NI];
#pragma unroll
    for (int r = 0; r < NI; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][q] = 0.f;
    float bsum = 0.f;
    const bool do_bias = Q32 && a.dbias != nullptr && ci0 == 0 && wi == 0;

    static_for<0, PF>([&](auto kc) { stage(sb + decltype(kc)::value); });
    for (int s = sb; s < se; ++s) {
        // all but the PF - 1 newest steps have landed (this wave's pieces; the barrier makes it every wave's) ...
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PER_STEP * (PF - 1)) : "memory");
        __builtin_amdgcn_s_barrier();                 // ... and every wave is done with step s-1, whose slot the next request refills
        asm volatile("" ::: "memory");
        stage(s + PF);
        const int slot = s % RING;
        const uint32_t xb = lds0 + slot * XSLOT;
        const uint32_t yb = lds0 + YOFF + slot * YSLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x8 bf;
            if constexpr (Q32) {
                const uint8_t* rp = lds_raw + YOFF + slot * YRAW + fr + j * (16 * 512);
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(rp + e * 512);
                if (do_bias) bsum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                const u32x4 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
                bf = __builtin_bit_cast(bf16x8, pk);
            } else {
                bf = tr_pair(yb + fb + j * 4096, yb + fb + j * 4096 + 1024);
            }
#pragma unroll
            for (int r = 0; r < NI; ++r) {
                const bf16x8 af = tr_pair(xb + r * XSTEP + fa + j * 2048, xb + r * XSTEP + fa + j * 2048 + 256);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[r], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (do_bias) {
        // lanes l and l + 32 hold the same output column (pixels of the two halves of every 16-pixel step); one atomic per column
        bsum += __shfl_xor(bsum, 32, 64);
        const int c = co0 + wj * 32 + l;
        if (l < 32 && c < a.Cj) atomicAdd(a.dbias + c, bsum);
    }
    if (a.splits == 1) {
#pragma unroll
        for (int r = 0; r < NI; ++r) {
            const int ci = ci0 + 64 * r + wi * 32 + 4 * (l >> 5), co = co0 + wj * 32 + (l & 31);
            if (co < a.Cj) {
                float* o = a.dW + (size_t)ci * a.Cj + co;
#pragma unroll
                for (int q = 0; q < 16; ++q) o[(size_t)((q & 3) + 8 * (q >> 2)) * a.Cj] += acc[r][q];
            }
        }
        return;
    }
    float* out = a.ws + (size_t)(split * ntiles + tile) * (NI * 4 * 2048) + t * 4;
#pragma unroll
    for (int r = 0; r < NI; ++r)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            *reinterpret_cast<f32x4*>(out + r * 8192 + rq * 2048) = f32x4{acc[r][4 * rq], acc[r][4 * rq + 1], acc[r][4 * rq + 2], acc[r][4 * rq + 3]};
}

// Exact-fp32 mode (Unet.compute_mode = "fp32"; round 4): X and dY are fp32 tensors, v_mfma_f32_32x32x2_f32 -- bit-equal to an fp32 fmaf
// chain over the pixels of a k-slice.  Same tiling (64 ci x 128 co per workgroup, wave (wi, wj) = one 32 x 32 tile), same k-slices,
// partial tiles and reduce.  No transposition is needed: the fp32 MFMA takes ONE k per lane half, i.e. lane (i, half) reads channel i of
// pixel 2s + half straight from the raw rows (32 lanes = 128 consecutive bytes: conflict-free).  Rows arrive by LDS-DMA: X 64 pixels x
// 64 channels (16 KB: four pixels per wave instruction), dY 64 pixels x 128 channels (32 KB: two pixels per instruction); three slots.
// Per 64-pixel step a wave issues 32 MFMAs (2048 clocks) for 6 DMA instructions: bound by the fp32 matrix pipe as long as the tiles of a
// k-slice share their rows in one XCD's L2 (xcd_map, as for the bf16 kernel).  Before: the generic wgrad_kernel<0>, 46 TFLOP/s.
constexpr int XRAW = 64 * 64 * 4;     // 64 pixels x 64 channels fp32
__device__ __forceinline__ void wgrad1_f32_body(const W1Args& a, const int wg, uint8_t* lds_raw) {
    constexpr int PF = 2, RING = PF + 1;
    constexpr int YOFF = RING * XRAW;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wv >> 2, wj = wv & 3;
    const int ntiles = a.gx * a.gy;
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
    const bool second = ci0 >= a.I1;                                   // the ci tile lies in one source of a two-source layer (I1 % 64 == 0)
    const int ldx = second ? a.ldp2 : a.ldp;
    const float* xsrc = reinterpret_cast<const float*>(second ? (const void*)a.P2 : (const void*)a.P) + (size_t)(l >> 4) * ldx +
                        (second ? ci0 - a.I1 : ci0) + (l & 15) * 4;
    const float* ysrc = reinterpret_cast<const float*>(a.Q) + (size_t)(l >> 5) * a.ldq + min(co0 + (l & 31) * 4, a.Cj - 4);
    // gathered operand (gmode): dense pixel m -> its big-tensor pixel for this problem's tap, or -1 outside the tensor
    auto gpix = [&](int m) -> long {
        const int n = m >> a.lghw, r = m & ((1 << a.lghw) - 1);
        const int gy = 2 * (r >> a.lgw) + a.gdy, gx = 2 * (r & ((1 << a.lgw) - 1)) + a.gdx;
        return ((unsigned)gy < (unsigned)a.GH && (unsigned)gx < (unsigned)a.GW) ? ((long)n * a.GH + gy) * a.GW + gx : -1L;
    };
    const float* xch = xsrc - (size_t)(l >> 4) * ldx;                  // channel part only (the pixel part is computed per request)
    const float* ych = ysrc - (size_t)(l >> 5) * a.ldq;
    const float* zsrc = reinterpret_cast<const float*>(g_w1_zero);
    auto stage = [&](int step) {
        const size_t pix0 = (size_t)min(step, last) * 64;
        const int slot = step % RING;
#pragma unroll
        for (int k = 0; k < 2; ++k) {                                  // 16 blocks of 4 pixels, two per wave
            const int i = wv + 8 * k;
            if (a.gmode == 1) {
                const long g = gpix((int)pix0 + i * 4 + (l >> 4));
                glds16(g >= 0 ? xch + (size_t)g * ldx : zsrc + (l & 15) * 4, lds0 + slot * XRAW + i * 1024);
            } else glds16(xsrc + (pix0 + i * 4) * ldx, lds0 + slot * XRAW + i * 1024);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                  // 32 pixel pairs, four per wave
            const int i = wv + 8 * k;
            if (a.gmode == 2) {
                const long g = gpix((int)pix0 + i * 2 + (l >> 5));
                glds16(g >= 0 ? ych + (size_t)g * a.ldq : zsrc + (l & 31) * 4, lds0 + YOFF + slot * YRAW + i * 1024);
            } else glds16(ysrc + (pix0 + i * 2) * a.ldq, lds0 + YOFF + slot * YRAW + i * 1024);
        }
    };
    constexpr int PER_STEP = 6;
    const int half = l >> 5;
    const int fa = half * 256 + (wi * 32 + (l & 31)) * 4;              // X: pixel 2s + half, channel wi*32 + lane
    const int fb = half * 512 + (wj * 32 + (l & 31)) * 4;              // dY: pixel 2s + half, channel wj*32 + lane
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float bsum = 0.f;
    const bool do_bias = a.dbias != nullptr && ci0 == 0 && wi == 0;
    static_for<0, PF>([&](auto kc) { stage(sb + decltype(kc)::value); });
    for (int s = sb; s < se; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PER_STEP * (PF - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(s + PF);
        const int slot = s % RING;
        const uint8_t* xb = lds_raw + slot * XRAW + fa;
        const uint8_t* yb = lds_raw + YOFF + slot * YRAW + fb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float av[8], bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                av[e] = *reinterpret_cast<const float*>(xb + (8 * j + e) * 512);
                bv[e] = *reinterpret_cast<const float*>(yb + (8 * j + e) * 1024);
            }
            if (do_bias) bsum += ((bv[0] + bv[1]) + (bv[2] + bv[3])) + ((bv[4] + bv[5]) + (bv[6] + bv[7]));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc, 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (do_bias) {
        bsum += __shfl_xor(bsum, 32, 64);
        const int c = co0 + wj * 32 + l;
        if (l < 32 && c < a.Cj) atomicAdd(a.dbias + c, bsum);
    }
    if (a.splits == 1) {
        const int ci = ci0 + wi * 32 + 4 * (l >> 5), co = co0 + wj * 32 + (l & 31);
        if (co < a.Cj) {
            float* o = a.dW + (size_t)ci * a.Cj + co;
#pragma unroll
            for (int q = 0; q < 16; ++q) o[(size_t)((q & 3) + 8 * (q >> 2)) * a.Cj] += acc[q];
        }
        return;
    }
    float* out = a.ws + (size_t)(split * ntiles + tile) * (4 * 2048) + t * 4;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<f32x4*>(out + rq * 2048) = f32x4{acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]};
}

__global__ __launch_bounds__(512, 1) void wgrad1x1_f32_kernel(const W1Batch b) {
    MI_PRIO_UP();
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    int p = 0;
#pragma unroll
    for (int q = 1; q < MAXP; ++q)
        if (q < b.n && (int)blockIdx.x >= b.p[q].wg0) p = q;
    const W1Args& a = b.p[p];
    wgrad1_f32_body(a, blockIdx.x - a.wg0, lds_raw);
}

__global__ __launch_bounds__(512, 1) void wgrad1x1_tr_kernel(const W1Batch b) {
    MI_PRIO_UP();
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    int p = 0;
#pragma unroll
    for (int q = 1; q < MAXP; ++q)
        if (q < b.n && (int)blockIdx.x >= b.p[q].wg0) p = q;
    const W1Args& a = b.p[p];
    const int wg = blockIdx.x - a.wg0;
    if (a.ni == 2) { if (a.q32) wgrad1_body<true, 2>(a, wg, lds_raw); else wgrad1_body<false, 2>(a, wg, lds_raw); }
    else           { if (a.q32) wgrad1_body<true, 1>(a, wg, lds_raw); else wgrad1_body<false, 1>(a, wg, lds_raw); }
}

// dW[ci][co] += sum over k-slices (fixed order); grid = ((512 / IB) position groups x 64-channel regions, 4 register quads, tiles with
// k-slices), 256 threads = G slice groups x IB = 256 / G float4 positions; each thread keeps 8 independent 16-byte loads in flight.
// (Round 4: it was 2 slice groups with 2 loads in flight -- a chain of 16 L2 round trips for the 64 k-slices of a to_qkv layer, 15 us per
// launch and six launches per step.)
template <int G>
__global__ __launch_bounds__(256) void wgrad1x1_tr_reduce_kernel(const W1Batch b) {
    MI_PRIO_UP();
    constexpr int IB = 256 / G, PB = 512 / IB;               // positions per workgroup, workgroups per (region, quad)
    __shared__ f32x4 red[G][IB];
    int pi = 0;
#pragma unroll
    for (int q = 1; q < MAXP; ++q)
        if (q < b.n && b.p[q].splits > 1 && (int)blockIdx.z >= b.p[q].tile0) pi = q;
    const W1Args& a = b.p[pi];
    const int reg = blockIdx.x / PB;
    if (reg >= a.ni) return;
    const int ntiles = a.gx * a.gy, splits = a.splits;
    const int it = threadIdx.x % IB, grp = threadIdx.x / IB;
    const int tt = (blockIdx.x % PB) * IB + it;
    const int rq = blockIdx.y, tile = blockIdx.z - a.tile0;
    const size_t tsz = (size_t)a.ni * (4 * 2048);
    const float* p = a.ws + (size_t)tile * tsz + (size_t)reg * 8192 + (size_t)rq * 2048 + tt * 4;
    const size_t stride = (size_t)ntiles * tsz;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int sp = grp;
    for (; sp + 7 * G < splits; sp += 8 * G) {
        f32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + q * G) * stride);
        s0 += v[0] + v[4]; s1 += v[1] + v[5]; s2 += v[2] + v[6]; s3 += v[3] + v[7];
    }
    {   // the tail: up to seven more, requested together
        f32x4 v[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) v[q] = (sp + q * G < splits) ? *reinterpret_cast<const f32x4*>(p + (size_t)(sp + q * G) * stride) : f32x4{0.f, 0.f, 0.f, 0.f};
        s0 += v[0] + v[4]; s1 += v[1] + v[5]; s2 += v[2] + v[6]; s3 += v[3];
    }
    f32x4 s = (s0 + s1) + (s2 + s3);
    red[grp][it] = s;
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int g = 1; g < G; ++g) s += red[g][it];
    const int wv = tt >> 6, l = tt & 63;
    const int ci = (tile % a.gx) * (64 * a.ni) + reg * 64 + (wv >> 2) * 32 + 8 * rq + 4 * (l >> 5);
    const int co = (tile / a.gx) * 128 + (wv & 3) * 32 + (l & 31);
    if (co >= a.Cj) return;
    float* out = a.dW + (size_t)ci * a.Cj + co;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (ci + e < a.Ci) out[(size_t)e * a.Cj] += s[e];
}

bool w1_ok(const MiWgradDesc* d, int q32) {
    if (d->KH != 1 || d->KW != 1 || d->pad != 0 || d->stride != 1 || !d->gather_i || (d->mode != 1 && d->mode != 0)) return false;
    if (d->mode == 0) {          // exact-fp32 mode: fp32 X and dY rows, 16-byte pieces
        if (d->GH != d->DH || d->GW != d->DW || ((long)d->N * d->DH * d->DW) % 64) return false;
        if (d->Ci % 64 || d->I1 % 64 || d->Cj % 32 || d->Cj < 32) return false;
        return q32 && d->ldp % 4 == 0 && (d->I1 == d->Ci || d->ldp2 % 4 == 0) && d->ldq % 4 == 0;
    }
    if (d->GH != d->DH || d->GW != d->DW) return false;
    if (((long)d->N * d->DH * d->DW) % 64) return false;
    if (d->Ci % 64 || d->I1 % 64 || d->Cj % 32 || d->Cj < 32) return false;
    if (d->ldp % 8 || (d->I1 != d->Ci && d->ldp2 % 8)) return false;
    return q32 ? d->ldq % 4 == 0 : d->ldq % 8 == 0;
}

int g_w1_phase = 0, g_w1_blocks = 0;

// 64-channel regions per ci tile: 2 wherever the layer allows it (the dY rows are then re-read Ci / 128 instead of Ci / 64 times)
int w1_ni(const MiWgradDesc* d) {
    static const int wide = (int)mi_knob("MI_W1_NI", 2);
    return (wide >= 2 && d->Ci % 128 == 0 && d->mode != 0) ? 2 : 1;      // (exact-fp32 mode: 48 KB of raw rows per step and slot already)
}

void w1_plan(const MiWgradDesc* d, W1Args& a, long wgs) {
    a.Ci = d->Ci; a.Cj = d->Cj; a.I1 = d->I1;
    a.ni = w1_ni(d);
    a.gx = d->Ci / (64 * a.ni); a.gy = (d->Cj + 127) / 128;
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
        const double tiles_ci = d[i].Ci / (64 * w1_ni(&d[i])), tiles_co = (d[i].Cj + 127) / 128;
        by[i] = (double)d[i].N * d[i].DH * d[i].DW * (tiles_co * d[i].Ci * (d[i].mode == 0 ? 4.0 : 2.0) + tiles_ci * d[i].Cj * (q32[i] ? 4.0 : 2.0));
        tot += by[i];
    }
    const long target = g_w1_blocks > 0 ? g_w1_blocks : 256;
    static const int greedy = (int)mi_knob("MI_W1_BALANCE", 1);
    if (greedy) {               // whole k-slices to whoever carries the most bytes per workgroup (tr_common.h)
        long tiles[MAXP];
        for (int i = 0; i < n; ++i) tiles[i] = (long)(d[i].Ci / (64 * w1_ni(&d[i]))) * ((d[i].Cj + 127) / 128);
        balance_shares(n, by, tiles, target, wgs);
        return;
    }
    for (int i = 0; i < n; ++i) {
        const long tiles = (long)(d[i].Ci / (64 * w1_ni(&d[i]))) * ((d[i].Cj + 127) / 128);
        long w = (long)(target * by[i] / tot + 0.5);
        w = w / tiles * tiles;
        wgs[i] = w < tiles ? tiles : w;
    }
}

size_t w1_ws_floats(const W1Args& a) { return a.splits > 1 ? (size_t)a.splits * a.gx * a.gy * a.ni * 4 * 2048 : 0; }
// dynamic LDS of one problem: X ring + dY ring (see wgrad1_body)
size_t w1_lds(const W1Args& a) {
    if (a.f32) return (size_t)3 * (XRAW + YRAW);
    const size_t ring = a.q32 ? 3 : 4;
    return ring * ((size_t)a.ni * XSTEP + (a.q32 ? YRAW : YSTEP));
}

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
    W1Batch b{};                                          // (every field a kernel reads is defined: gmode = 0 means no gather)
    b.n = n;
    long wgs[MAXP];
    for (int i = 0; i < n; ++i) {
        MI_REQUIRE(w1_ok(&descs[i], q_is_fp32[i]), "descriptor not supported by the LDS-DMA 1x1 weight-gradient kernel");
        MI_REQUIRE(descs[i].mode == descs[0].mode, "one numeric mode per batch");
        MI_REQUIRE(P[i] && Q[i] && dW[i], "null operand");
        MI_REQUIRE(descs[i].I1 == descs[i].Ci || (P2 && P2[i]), "two-source split without P2");
        MI_REQUIRE((((uintptr_t)P[i] | (uintptr_t)Q[i] | (uintptr_t)((P2 && P2[i]) ? P2[i] : P[i])) & 15) == 0, "operands must be 16-byte aligned");
        MI_REQUIRE(!(dbias && dbias[i]) || q_is_fp32[i], "the bias gradient is summed in the fp32 conversion pass (fp32 dY only)");
    }
    w1_shares(n, descs, q_is_fp32, wgs);
    size_t off = 0, lds = 0;
    int wg = 0, tile = 0, nimax = 1, max_splits = 1;
    for (int i = 0; i < n; ++i) {
        W1Args& a = b.p[i];
        w1_plan(&descs[i], a, wgs[i]);
        a.P = (const uint16_t*)P[i]; a.P2 = (const uint16_t*)((P2 && P2[i]) ? P2[i] : P[i]); a.Q = Q[i];
        a.dW = dW[i]; a.dbias = dbias ? dbias[i] : nullptr; a.q32 = q_is_fp32[i] ? 1 : 0; a.f32 = descs[i].mode == 0 ? 1 : 0;
        a.ldp = descs[i].ldp; a.ldp2 = (P2 && P2[i]) ? descs[i].ldp2 : descs[i].ldp; a.ldq = descs[i].ldq;
        a.ws = (float*)workspace + off;
        off += w1_ws_floats(a);
        static const int xcd_env = (int)mi_knob("MI_W1_XCD", 1);
        a.xcd_map = xcd_env && a.gx * a.gy > 1 && a.splits > 1;
        a.wg0 = wg; wg += a.gx * a.gy * a.splits;
        if (a.splits > max_splits) max_splits = a.splits;
        a.tile0 = tile; if (a.splits > 1) { tile += a.gx * a.gy; nimax = a.ni > nimax ? a.ni : nimax; }
        lds = w1_lds(a) > lds ? w1_lds(a) : lds;
    }
    MI_REQUIRE(off == 0 || (workspace && ((uintptr_t)workspace & 15) == 0 && ws_bytes >= off * sizeof(float)),
               "workspace too small (mi_conv1x1_wgrad_tr_batch_workspace)");
    hipStream_t st = (hipStream_t)stream;
    static MiPerDevice once;
    once.run([] {
        (void)hipFuncSetAttribute((const void*)wgrad1x1_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad1x1_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (g_w1_phase != 2) {
        if (descs[0].mode == 0) hipLaunchKernelGGL(wgrad1x1_f32_kernel, dim3((unsigned)wg), dim3(512), lds, st, b);
        else hipLaunchKernelGGL(wgrad1x1_tr_kernel, dim3((unsigned)wg), dim3(512), lds, st, b);
    }
    if (g_w1_phase != 1 && tile > 0) {
        if (max_splits >= 32) hipLaunchKernelGGL(wgrad1x1_tr_reduce_kernel<8>, dim3(16 * nimax, 4, tile), dim3(256), 0, st, b);
        else hipLaunchKernelGGL(wgrad1x1_tr_reduce_kernel<2>, dim3(4 * nimax, 4, tile), dim3(256), 0, st, b);
    }
    MI_LAUNCH_CHECK();
    return 0;
}

// ---- round 6, exact-fp32 mode: the weight gradients of Downsample = Conv2d(C, C, 3, 2, 1) and Upsample = ConvTranspose2d(C, C, 4, 2, 1)
//      (reference src/models/ddpm.py:70,79) as k x k GATHERED 1x1 problems of wgrad1x1_f32_kernel, up to eight taps per launch:
//          dW[ky][kx][ci][co] += sum_m P[pix_p(m)][ci] * Q[pix_q(m)][co],   m over the small grid [N][DH][DW],
//      the big tensor read at (2 y + ky - 1, 2 x + kx - 1) (gather_i: P is the big one, else Q).  Replaces the generic wgrad_kernel<0> for
//      these four layers (0.87 ms of fp32 mode's 21.8 ms step at 0.40 of the fp32 matrix peak).
namespace {
bool s2f_ok(const MiWgradDesc* d) {
    if (!d || d->mode != 0 || d->stride != 2 || d->pad != 1 || d->KH != d->KW || (d->KH != 3 && d->KH != 4)) return false;
    if (d->I1 != d->Ci || d->Ci % 64 || d->Cj % 32 || d->Cj < 32 || d->ldp % 4 || d->ldq % 4) return false;
    if (d->GH != 2 * d->DH || d->GW != 2 * d->DW || (d->DH & (d->DH - 1)) || (d->DW & (d->DW - 1))) return false;
    return ((long)d->N * d->DH * d->DW) % 64 == 0;
}
MiWgradDesc s2f_tap_desc(const MiWgradDesc* d) {          // the dense 1x1 problem one tap is planned as
    MiWgradDesc t = *d;
    t.KH = t.KW = 1; t.pad = 0; t.stride = 1; t.gather_i = 1; t.GH = d->DH; t.GW = d->DW;
    return t;
}
int s2f_groups(int ntap, int* first, int* count) {          // taps per launch: 9 -> 5 + 4, 16 -> 8 + 8
    const int ng = (ntap + MAXP - 1) / MAXP;
    int t0 = 0;
    for (int g = 0; g < ng; ++g) { first[g] = t0; count[g] = (ntap - t0 + (ng - g) - 1) / (ng - g); t0 += count[g]; }
    return ng;
}
size_t s2f_group_ws(const MiWgradDesc* d, int n) {
    const MiWgradDesc t = s2f_tap_desc(d);
    MiWgradDesc ds[MAXP]; int q32[MAXP]; long wgs[MAXP];
    for (int i = 0; i < n; ++i) { ds[i] = t; q32[i] = 1; }
    w1_shares(n, ds, q32, wgs);
    size_t fl = 0;
    for (int i = 0; i < n; ++i) { W1Args a{}; w1_plan(&t, a, wgs[i]); fl += w1_ws_floats(a); }
    return fl;
}
}  // namespace
extern "C" int mi_conv_s2_wgrad_f32_supported(const MiWgradDesc* d) { return s2f_ok(d) ? 1 : 0; }
extern "C" size_t mi_conv_s2_wgrad_f32_workspace(const MiWgradDesc* d) {
    if (!s2f_ok(d)) return 0;
    int first[4], count[4];
    const int ng = s2f_groups(d->KH * d->KW, first, count);
    size_t fl = 0;
    for (int g = 0; g < ng; ++g) { const size_t f = s2f_group_ws(d, count[g]); fl = f > fl ? f : fl; }
    return fl * sizeof(float) + 256;
}
extern "C" int mi_conv_s2_wgrad_f32(const MiWgradDesc* d, const float* P, const float* Q, float* dW, void* workspace, size_t ws_bytes, void* stream) {
    MI_REQUIRE(s2f_ok(d), "descriptor not supported (exact-fp32 mode, 3x3 or 4x4, stride 2, pad 1, one source, Ci % 64, Cj % 32, power-of-two small grid)");
    MI_REQUIRE(P && Q && dW && ((((uintptr_t)P | (uintptr_t)Q) & 15) == 0), "null or misaligned operand");
    hipStream_t st = (hipStream_t)stream;
    static MiPerDevice once;
    once.run([] { (void)hipFuncSetAttribute((const void*)wgrad1x1_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    const MiWgradDesc t = s2f_tap_desc(d);
    int lgw = 0, lgh = 0;
    while ((1 << lgw) < d->DW) ++lgw;
    while ((1 << lgh) < d->DH) ++lgh;
    int first[4], count[4];
    const int ng = s2f_groups(d->KH * d->KW, first, count);
    for (int g = 0; g < ng; ++g) {
        const int n = count[g];
        MiWgradDesc ds[MAXP]; int q32[MAXP]; long wgs[MAXP];
        for (int i = 0; i < n; ++i) { ds[i] = t; q32[i] = 1; }
        w1_shares(n, ds, q32, wgs);
        W1Batch b{};                                          // (every field a kernel reads is defined: gmode = 0 means no gather)
        b.n = n;
        size_t off = 0, lds = 0;
        int wg = 0, tile = 0, max_splits = 1;
        for (int i = 0; i < n; ++i) {
            const int tap = first[g] + i, ky = tap / d->KW, kx = tap % d->KW;
            W1Args& a = b.p[i];
            a = W1Args{};
            w1_plan(&t, a, wgs[i]);
            a.P = (const uint16_t*)P; a.P2 = (const uint16_t*)P; a.Q = Q;
            a.dW = dW + (size_t)tap * d->Ci * d->Cj; a.dbias = nullptr; a.q32 = 1; a.f32 = 1;
            a.ldp = d->ldp; a.ldp2 = d->ldp; a.ldq = d->ldq;
            a.ws = (float*)workspace + off;
            off += w1_ws_floats(a);
            a.xcd_map = a.gx * a.gy > 1 && a.splits > 1;
            a.wg0 = wg; wg += a.gx * a.gy * a.splits;
            if (a.splits > max_splits) max_splits = a.splits;
            a.tile0 = tile; if (a.splits > 1) tile += a.gx * a.gy;
            a.gmode = d->gather_i ? 1 : 2; a.gdy = ky - d->pad; a.gdx = kx - d->pad; a.lgw = lgw; a.lghw = lgw + lgh; a.GH = d->GH; a.GW = d->GW;
            lds = w1_lds(a) > lds ? w1_lds(a) : lds;
        }
        MI_REQUIRE(off == 0 || (workspace && ((uintptr_t)workspace & 15) == 0 && ws_bytes >= off * sizeof(float)),
                   "workspace too small (mi_conv_s2_wgrad_f32_workspace)");
        hipLaunchKernelGGL(wgrad1x1_f32_kernel, dim3((unsigned)wg), dim3(512), lds, st, b);
        if (tile > 0) {
            if (max_splits >= 32) hipLaunchKernelGGL(wgrad1x1_tr_reduce_kernel<8>, dim3(16, 4, tile), dim3(256), 0, st, b);
            else hipLaunchKernelGGL(wgrad1x1_tr_reduce_kernel<2>, dim3(4, 4, tile), dim3(256), 0, st, b);
        }
        MI_LAUNCH_CHECK();
    }
    return 0;
}
