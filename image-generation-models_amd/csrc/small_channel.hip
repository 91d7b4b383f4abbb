// The 3-channel ends of the UNet: Conv2d(3, C, 3, padding=1) and the 1x1 res_conv of downs.0.0
// (ddpm.py:116,134,208) and final_conv's Conv2d(C, 3, 1) (ddpm.py:236), forward / dgrad / wgrad.
// With 3 (or 3x9 = 27) contraction elements an MFMA tile would be 90 % padding, so these are fp32
// VALU kernels bound by the one wide tensor they stream (the C-channel activation or its gradient).
// Layout of every kernel: a thread owns one channel quad of the wide side (16-byte loads / stores, 512-byte
// rows per 32 lanes) and a pixel lane; the small side is a broadcast load; weights live in registers.
// Weight gradients are reduced per workgroup (shuffle + LDS) and leave as one partial tile per workgroup
// (summed by partial_sum_kernel) -- or as one atomic per output and workgroup when no workspace is given.
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// the wide tensor of these kernels may be stored as bf16 (round 4: the first Block's conv output / its gradient and the final Block's
// output / its gradient, like every other block-internal tensor): a thread's channel quad is one 8-byte load / store
template <bool B16> __device__ __forceinline__ f32x4 ldq(const void* base, size_t off) {
    if constexpr (B16) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + off);
        return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
    } else return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
}
template <bool B16> __device__ __forceinline__ void stq(void* base, size_t off, const f32x4& v) {
    if constexpr (B16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + off) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
    else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = v;
}

constexpr int WG_BLOCKS = 768;          // weight-gradient workgroups (3 per CU): the number of partial tiles

// m -> (image, row, column); POW2: W and H*W are powers of two (shifts), else divisions
template <bool POW2>
__device__ __forceinline__ void decode_px(int m, int H, int W, int w_sh, int hw_sh, int& n, int& yy, int& xx) {
    if constexpr (POW2) {
        n = m >> hw_sh; const int rem = m & ((1 << hw_sh) - 1);
        yy = rem >> w_sh; xx = rem & ((1 << w_sh) - 1);
    } else {
        n = m / (H * W); const int rem = m - n * (H * W);
        yy = rem / W; xx = rem - yy * W;
    }
}

// the <= 4 input channels of one pixel (zero when the tap falls outside the image)
template <int CIN, bool VEC>      // VEC: 16-byte aligned pixels (ldx % 4 == 0) -> one load per tap, no branch around it
__device__ __forceinline__ f32x4 load_small(const float* __restrict__ x, int ldx, int n, int iy, int ix, int H, int W) {
    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
    const float* p = x + (size_t)((n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * ldx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC) v = *reinterpret_cast<const f32x4*>(p);
    else { v.x = p[0]; if (CIN > 1) v.y = p[1]; if (CIN > 2) v.z = p[2]; if (CIN > 3) v.w = p[3]; }
    const float k = ok ? 1.f : 0.f;
    return v * k;
}

// y[px][co] = b[co] + sum_{tap,ci<CIN} x[px+tap][ci] * w[tap][ci][co];   KS = 3 (pad 1) or 1.  Two pixels per
// iteration: both pixels' tap loads are in flight before the first FMA.
template <int CIN, int KS, bool POW2, bool VEC>
__global__ __launch_bounds__(256) void small_cin_fwd_kernel(int N, int H, int W, int Cout, const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, int ldy, int w_sh, int hw_sh) {
    constexpr int NT = KS * KS;
    const int nq = Cout / 4;                              // channel quads
    const int q = threadIdx.x % nq, psub = threadIdx.x / nq, pp = 256 / nq;
    f32x4 wr[NT * CIN];
#pragma unroll
    for (int i = 0; i < NT * CIN; ++i) wr[i] = *reinterpret_cast<const f32x4*>(w + (size_t)i * Cout + 4 * q);
    const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int M = N * H * W, step = gridDim.x * pp;
    for (int m0 = blockIdx.x * pp + psub; m0 < M; m0 += 2 * step) {
        f32x4 xv[2][NT];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int n, yy, xx;
            decode_px<POW2>(min(m0 + u * step, M - 1), H, W, w_sh, hw_sh, n, yy, xx);
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) xv[u][tp] = load_small<CIN, VEC>(x, ldx, n, yy + tp / KS - KS / 2, xx + tp % KS - KS / 2, H, W);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 acc = b4;
#pragma unroll
            for (int tp = 0; tp < NT; ++tp)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) acc += xv[u][tp][ci] * wr[tp * CIN + ci];
            if (m0 + u * step < M) *reinterpret_cast<f32x4*>(y + (size_t)(m0 + u * step) * ldy + 4 * q) = acc;
        }
    }
}

// The same 3x3 conv with the input taps served from LDS.  In the kernel above every thread issues 9 global tap loads per pixel and
// waits for them (200 registers of weights and taps, two waves per SIMD: a latency chain of ~1 M load instructions per launch at
// B = 128).  Here a workgroup owns 64-pixel tiles (whole image rows), the tile's (rows + 2) x (W + 2) halo of padded pixels is
// fetched by one load per thread -- the NEXT tile's while this one is computed -- and the taps are ds_read_b128.
// DUAL: the ResnetBlock's res_conv (Conv2d(Cin, Cout, 1) on the same input, reference ddpm.py:134,143) rides along -- its input is the centre
// tap: 3 more FMAs per output quad and a second (fp32) output instead of a launch that reads the image again.
template <int CIN, bool Y16 = false, bool DUAL = false>
__global__ __launch_bounds__(256) void small_cin3x3_fwd_tiled_kernel(int N, int H, int W, int Cout, const float* __restrict__ x, int ldx,
                                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                                     void* __restrict__ y, int ldy, int w_sh, int ntiles,
                                                                     const float* __restrict__ w1 = nullptr, const float* __restrict__ bias1 = nullptr,
                                                                     float* __restrict__ y1 = nullptr, int ldy1 = 0,
                                                                     unsigned long long* __restrict__ zero = nullptr, int nzero = 0,
                                                                     const float* __restrict__ gsrc = nullptr, const long long* __restrict__ gidx = nullptr,
                                                                     float* __restrict__ gdst = nullptr, int grow = 0, int gB = 0) {
    // (DUAL, inference: this is the first launch of a forward -- it also does the forward's two chores whose consumers are LATER launches: it clears
    //  the pool of GroupNorm sums the conv epilogues add into, and it gathers the step's time-bias rows gdst[b][:] = gsrc[gidx[b]][:] (the sampler's
    //  precomputed table): two launches less per denoise step)
    if constexpr (DUAL) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < nzero; i += gridDim.x * 256) zero[i] = 0ull;
        const int rq = grow >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < gB * rq; i += gridDim.x * 256) {
            const int b = i / rq, q = i - b * rq;
            *reinterpret_cast<f32x4*>(gdst + (size_t)b * grow + 4 * q) = *reinterpret_cast<const f32x4*>(gsrc + (size_t)gidx[b] * grow + 4 * q);
        }
    }
    extern __shared__ __attribute__((aligned(16))) float halo_s[];          // 2 x [(rows + 2) * (W + 2)] float4
    const int nq = Cout / 4;
    const int q = threadIdx.x % nq, psub = threadIdx.x / nq, pp = 256 / nq;
    const int rows = 64 >> w_sh, W2 = W + 2, HP = (rows + 2) * W2, tiles_per_img = H / rows;
    f32x4 wr[9 * CIN];
#pragma unroll
    for (int i = 0; i < 9 * CIN; ++i) wr[i] = *reinterpret_cast<const f32x4*>(w + (size_t)i * Cout + 4 * q);
    const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 wr1[DUAL ? CIN : 1], b1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (DUAL) {
#pragma unroll
        for (int i = 0; i < CIN; ++i) wr1[i] = *reinterpret_cast<const f32x4*>(w1 + (size_t)i * Cout + 4 * q);
        if (bias1) b1 = *reinterpret_cast<const f32x4*>(bias1 + 4 * q);
    }
    // this thread's halo pixel (HP <= 256): position -> (row, column) of the tile's halo
    const int hp = threadIdx.x, hy = hp / W2, hx = hp - hy * W2;
    auto fetch = [&](int tile) -> f32x4 {
        const int n = tile / tiles_per_img, y0 = (tile - n * tiles_per_img) * rows;
        const int iy = y0 + hy - 1, ix = hx - 1;
        const bool ok = hp < HP && tile < ntiles && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)((n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * (ok ? ldx : 0));
        return v * (ok ? 1.f : 0.f);
    };
    int tile = blockIdx.x, buf = 0;
    f32x4 nxt = fetch(tile);
    if (hp < HP) *reinterpret_cast<f32x4*>(halo_s + (size_t)hp * 4) = nxt;
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        nxt = fetch(tile + gridDim.x);                                        // in flight under this tile's arithmetic
        const float* hs = halo_s + (size_t)buf * HP * 4;
        const size_t m0 = (size_t)tile * 64;
        for (int px = psub; px < 64; px += pp) {
            const int ty = px >> w_sh, tx = px & (W - 1);
            f32x4 acc = b4, acc1 = b1;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(hs + (size_t)((ty + tp / 3) * W2 + tx + tp % 3) * 4);
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) acc += xv[ci] * wr[tp * CIN + ci];
                if constexpr (DUAL) {
                    if (tp == 4) {
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) acc1 += xv[ci] * wr1[ci];
                    }
                }
            }
            stq<Y16>(y, (m0 + px) * ldy + 4 * q, acc);
            if constexpr (DUAL) stq<false>(y1, (m0 + px) * ldy1 + 4 * q, acc1);
        }
        buf ^= 1;
        if (hp < HP) *reinterpret_cast<f32x4*>(halo_s + ((size_t)buf * HP + hp) * 4) = nxt;
        __syncthreads();
    }
}

// Sum of a per-thread array of NA float4 over the pixel lanes of a workgroup (threads with equal t % nq), result in
// threads 0..nq-1.  nq is 16, 32 or 64: the pixel lanes inside a wave are combined by shuffles, the four waves via LDS.
template <int NA>
__device__ __forceinline__ void block_reduce_quads(f32x4 (&a)[NA], int nq, float* red /* [3][NA][nq][4] */) {
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    for (int o = nq; o < 64; o <<= 1) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            a[i].x += __shfl_xor(a[i].x, o, 64); a[i].y += __shfl_xor(a[i].y, o, 64);
            a[i].z += __shfl_xor(a[i].z, o, 64); a[i].w += __shfl_xor(a[i].w, o, 64);
        }
    }
    if (wv > 0 && l < nq) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&red[(((wv - 1) * NA + i) * nq + l) * 4]) = a[i];
    }
    __syncthreads();
    if (wv == 0 && l < nq) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) a[i] += *reinterpret_cast<const f32x4*>(&red[((k * NA + i) * nq + l) * 4]);
    }
}

// dW[tap][ci][co] += sum_px x[px+tap][ci] * dy[px][co]: thread = (co quad, pixel lane), KS*KS*CIN float4 accumulators
template <int CIN, int KS, bool POW2, bool VEC>
__global__ __launch_bounds__(256) void small_cin_wgrad_kernel(int N, int H, int W, int Cout, const float* __restrict__ x, int ldx,
                                                              const float* __restrict__ dy, int lddy, float* __restrict__ dW,
                                                              float* __restrict__ ws, int w_sh, int hw_sh) {
    constexpr int NT = KS * KS, NA = NT * CIN;
    extern __shared__ float red[];
    const int nq = Cout / 4;
    const int q = threadIdx.x % nq, psub = threadIdx.x / nq, pp = 256 / nq;
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int M = N * H * W;
    const int per = (M + gridDim.x - 1) / gridDim.x;
    const int mb = blockIdx.x * per, me = min(M, mb + per);
    int m = mb + psub;
    f32x4 g = *reinterpret_cast<const f32x4*>(dy + (size_t)min(m, M - 1) * lddy + 4 * q);
    for (; m < me; m += pp) {
        const f32x4 gnext = *reinterpret_cast<const f32x4*>(dy + (size_t)min(m + pp, M - 1) * lddy + 4 * q);   // one pixel ahead
        int n, yy, xx;
        decode_px<POW2>(m, H, W, w_sh, hw_sh, n, yy, xx);
#pragma unroll
        for (int tp = 0; tp < NT; ++tp) {
            const f32x4 xv = load_small<CIN, VEC>(x, ldx, n, yy + tp / KS - KS / 2, xx + tp % KS - KS / 2, H, W);
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) acc[tp * CIN + ci] += xv[ci] * g;
        }
        g = gnext;
    }
    block_reduce_quads<NA>(acc, nq, red);
    if (threadIdx.x < nq) {
        if (ws) {
            float* out = ws + (size_t)blockIdx.x * NA * Cout;
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(out + (size_t)i * Cout + 4 * q) = acc[i];
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(dW + (size_t)i * Cout + 4 * q + k, acc[i][k]);
        }
    }
}

// The 3x3 weight gradient with the input taps in LDS (same tiling as small_cin3x3_fwd_tiled_kernel): a workgroup walks 64-pixel
// row tiles, a thread = (co quad, pixel lane) requests its tile's dy rows (64 / pp of them) up front, the taps are ds_read_b128.
template <int CIN, bool DY16 = false>
__global__ __launch_bounds__(256) void small_cin3x3_wgrad_tiled_kernel(int N, int H, int W, int Cout, const float* __restrict__ x, int ldx,
                                                                       const void* __restrict__ dy, int lddy, float* __restrict__ dW,
                                                                       float* __restrict__ ws, int w_sh, int ntiles) {
    constexpr int NA = 9 * CIN;
    extern __shared__ __attribute__((aligned(16))) float smem[];             // [reduce area 3 * NA * nq * 4][2 halo buffers]
    const int nq = Cout / 4;
    const int q = threadIdx.x % nq, psub = threadIdx.x / nq, pp = 256 / nq;
    const int rows = 64 >> w_sh, W2 = W + 2, HP = (rows + 2) * W2, tiles_per_img = H / rows;
    float* red = smem;
    float* halo_s = smem + (size_t)3 * NA * nq * 4;
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int hp = threadIdx.x, hy = hp / W2, hx = hp - hy * W2;
    auto fetch = [&](int tile) -> f32x4 {
        const int n = tile / tiles_per_img, y0 = (tile - n * tiles_per_img) * rows;
        const int iy = y0 + hy - 1, ix = hx - 1;
        const bool ok = hp < HP && tile < ntiles && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)((n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * (ok ? ldx : 0));
        return v * (ok ? 1.f : 0.f);
    };
    // contiguous range of tiles per workgroup (one partial tile per workgroup leaves at the end)
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int tb = blockIdx.x * per, te = min(ntiles, tb + per);
    int buf = 0;
    f32x4 nxt = fetch(tb < te ? tb : ntiles);
    if (hp < HP) *reinterpret_cast<f32x4*>(halo_s + (size_t)hp * 4) = nxt;
    __syncthreads();
    constexpr int MAXPX = 8;                                                   // 64 / pp <= 8 for Cout >= 128; Cout = 64: two passes of 8... handled by the loop
    for (int tile = tb; tile < te; ++tile) {
        nxt = fetch(tile + 1 < te ? tile + 1 : ntiles);
        const float* hs = halo_s + (size_t)buf * HP * 4;
        const size_t m0 = (size_t)tile * 64;
        for (int base = psub; base < 64; base += pp * MAXPX) {
            f32x4 g[MAXPX];
#pragma unroll
            for (int k = 0; k < MAXPX; ++k) {
                const int px = base + k * pp;
                g[k] = ldq<DY16>(dy, (m0 + (px < 64 ? px : 63)) * lddy + 4 * q);
            }
#pragma unroll
            for (int k = 0; k < MAXPX; ++k) {
                const int px = base + k * pp;
                if (px < 64) {
                    const int ty = px >> w_sh, tx = px & (W - 1);
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(hs + (size_t)((ty + tp / 3) * W2 + tx + tp % 3) * 4);
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) acc[tp * CIN + ci] += xv[ci] * g[k];
                    }
                }
            }
        }
        buf ^= 1;
        if (hp < HP) *reinterpret_cast<f32x4*>(halo_s + ((size_t)buf * HP + hp) * 4) = nxt;
        __syncthreads();
    }
    block_reduce_quads<NA>(acc, nq, red);
    if (threadIdx.x < nq) {
        if (ws) {
            float* out = ws + (size_t)blockIdx.x * NA * Cout;
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(out + (size_t)i * Cout + 4 * q) = acc[i];
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(dW + (size_t)i * Cout + 4 * q + k, acc[i][k]);
        }
    }
}

// out[map(o)] += sum_p ws[p * nout + o]; rows4 > 0: the partial tiles are [rows][4] and out is [rows][rows4] (rows4 <= 4)
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ ws, int nparts, int nout, float* __restrict__ out, int rows4) {
    __shared__ float red[8][32];
    const int ol = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int o = blockIdx.x * 32 + ol;
    float s = 0.f;
    if (o < nout) {                                       // eight independent loads in flight per thread
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int p = grp;
        for (; p + 56 < nparts; p += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(p + 8 * u) * nout + o];
            s0 += v[0] + v[4]; s1 += v[1] + v[5]; s2 += v[2] + v[6]; s3 += v[3] + v[7];
        }
        for (; p < nparts; p += 8) s0 += ws[(size_t)p * nout + o];
        s = (s0 + s1) + (s2 + s3);
    }
    red[grp][ol] = s;
    __syncthreads();
    if (grp == 0 && o < nout) {
#pragma unroll
        for (int g = 1; g < 8; ++g) s += red[g][ol];
        if (rows4 > 0) { if ((o & 3) < rows4) out[(o >> 2) * rows4 + (o & 3)] += s; }
        else out[o] += s;
    }
}

// ---- 1x1 conv to/from a few channels (Cs <= 4 "small" side, C wide side), weights w[c][j] (c < C, j < Cs)
// forward: y[px][j] = b[j] + sum_c x[px][c] w[c][j]; 8 lanes per pixel, each 4*CK channels, weights in registers
template <int CK, bool X16 = false>
__global__ __launch_bounds__(256) void small_cout_fwd_kernel(int M, int Cs, const void* __restrict__ x, int ldx,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ y, int ldy) {
    const int sub = threadIdx.x & 7, pl = threadIdx.x >> 3;          // 32 pixels per workgroup and iteration
    f32x4 wr[CK][4];                                                 // wr[k][e][j] = w[c = 4*(sub+8k)+e][j]
#pragma unroll
    for (int k = 0; k < CK; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* wp = w + (size_t)(4 * (sub + 8 * k) + e) * Cs;
            wr[k][e] = f32x4{wp[0], Cs > 1 ? wp[1] : 0.f, Cs > 2 ? wp[2] : 0.f, Cs > 3 ? wp[3] : 0.f};
        }
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) { b4.x = bias[0]; if (Cs > 1) b4.y = bias[1]; if (Cs > 2) b4.z = bias[2]; if (Cs > 3) b4.w = bias[3]; }
    // the next 32 pixels' rows are requested before this iteration's arithmetic (a workgroup runs several iterations: the 48 scalar
    // weight loads above used to be paid once per 32 pixels)
    f32x4 xn[CK];
    {
        const int m = min((int)blockIdx.x * 32 + pl, M - 1);
#pragma unroll
        for (int k = 0; k < CK; ++k) xn[k] = ldq<X16>(x, (size_t)m * ldx + 4 * (sub + 8 * k));
    }
    for (int m0 = blockIdx.x * 32; m0 < M; m0 += gridDim.x * 32) {
        const int m = min(m0 + pl, M - 1);
        f32x4 xv[CK];
#pragma unroll
        for (int k = 0; k < CK; ++k) xv[k] = xn[k];
        {
            const int mn = min(m0 + (int)gridDim.x * 32 + pl, M - 1);
#pragma unroll
            for (int k = 0; k < CK; ++k) xn[k] = ldq<X16>(x, (size_t)mn * ldx + 4 * (sub + 8 * k));
        }
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < CK; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) a += xv[k][e] * wr[k][e];
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64);
            a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
        }
        if (sub == 0 && m0 + pl < M) {
            a += b4;
            float* yp = y + (size_t)m * ldy;
            // (the padding channels Cs .. ldy - 1 of a padded pixel are written too -- as zeros: the caller needs no fill)
            yp[0] = a.x; if (Cs > 1 || ldy > 1) yp[1] = Cs > 1 ? a.y : 0.f; if (Cs > 2 || ldy > 2) yp[2] = Cs > 2 ? a.z : 0.f;
            if (Cs > 3 || ldy > 3) yp[3] = Cs > 3 ? a.w : 0.f;
        }
    }
}
// Inference, final_conv (reference ddpm.py:232-235: Block = conv -> GroupNorm -> Mish, then Conv2d(dim, channels, 1)): the Block's GroupNorm-apply
// + Mish ride in the 1x1 conv's load.  x = the Block conv's bf16 output, the statistics come from the sums its epilogue left (mi_conv3x3_pw_gnsums;
// mi_gn_coef_from_sums' arithmetic per lane); the normalised tensor is never written and never rounded to bf16.  A workgroup takes a contiguous run
// of ppw pixels (one sample when ppw divides H * W): the coefficients of a lane's 4 * CK channels are resolved once per sample.
template <int CK>
__global__ __launch_bounds__(256) void small_cout_fwd_gn_kernel(int M, int HW, int Cs, const uint16_t* __restrict__ x, int ldx,
                                                                const long long* __restrict__ sums, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int G, float eps, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y, int ldy, int ppw) {
    constexpr int C = 32 * CK;
    const int sub = threadIdx.x & 7, pl = threadIdx.x >> 3;          // 32 pixels per iteration, 8 lanes per pixel
    f32x4 wr[CK][4];                                                 // wr[k][e][j] = w[c = 4*(sub+8k)+e][j]
    f32x4 ga[CK], be[CK];
#pragma unroll
    for (int k = 0; k < CK; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* wp = w + (size_t)(4 * (sub + 8 * k) + e) * Cs;
            wr[k][e] = f32x4{wp[0], Cs > 1 ? wp[1] : 0.f, Cs > 2 ? wp[2] : 0.f, Cs > 3 ? wp[3] : 0.f};
        }
        ga[k] = *reinterpret_cast<const f32x4*>(gamma + 4 * (sub + 8 * k));
        be[k] = *reinterpret_cast<const f32x4*>(beta + 4 * (sub + 8 * k));
    }
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (bias) { b4.x = bias[0]; if (Cs > 1) b4.y = bias[1]; if (Cs > 2) b4.z = bias[2]; if (Cs > 3) b4.w = bias[3]; }
    const int Cg = C / G, nslab = Cg / 16;
    const double icnt = 1.0 / (MI_GSUM_SCALE * (double)HW * (double)Cg);
    const int mb = blockIdx.x * ppw, me = min(M, mb + ppw);
    int ncur = -1;
    f32x4 sc[CK], sh[CK];
    // (the next 32 pixels' rows are requested before this iteration's arithmetic, as in small_cout_fwd_kernel)
    typedef unsigned int u32x2q __attribute__((ext_vector_type(2)));
    u32x2q xn[CK];
    {
        const int m = min(mb + pl, M - 1);
#pragma unroll
        for (int k = 0; k < CK; ++k) xn[k] = *reinterpret_cast<const u32x2q*>(x + (size_t)m * ldx + 4 * (sub + 8 * k));
    }
    for (int m0 = mb; m0 < me; m0 += 32) {
        const int m = min(m0 + pl, M - 1);
        f32x4 xv[CK];
#pragma unroll
        for (int k = 0; k < CK; ++k)
            xv[k] = f32x4{__uint_as_float(xn[k].x << 16), __uint_as_float(xn[k].x & 0xffff0000u), __uint_as_float(xn[k].y << 16), __uint_as_float(xn[k].y & 0xffff0000u)};
        {
            const int mn = min(m0 + 32 + pl, M - 1);
#pragma unroll
            for (int k = 0; k < CK; ++k) xn[k] = *reinterpret_cast<const u32x2q*>(x + (size_t)mn * ldx + 4 * (sub + 8 * k));
        }
        const int n = m / HW;
        if (n != ncur) {
            ncur = n;
#pragma unroll
            for (int k = 0; k < CK; ++k) {
                const int g = (4 * (sub + 8 * k)) / Cg;              // (a channel quad lies in one group: Cg % 16 == 0)
                long long si = 0, qi = 0;
                bool poisoned = false;
                for (int j = 0; j < nslab; ++j) {
                    const size_t p = ((size_t)n * (C / 16) + g * nslab + j) * 2;
                    const long long r0 = sums[p], r1 = sums[p + 1];
                    poisoned |= r0 >= (1LL << 60) || r0 <= -(1LL << 60) || r1 >= (1LL << 60) || r1 <= -(1LL << 60);
                    si += r0; qi += r1;
                }
                const double mean = poisoned ? __builtin_nan("") : (double)si * icnt;
                double var = (double)qi * icnt - mean * mean;
                if (var < 0.0) var = 0.0;
                const float rstd = 1.0f / sqrtf((float)var + eps), mf = (float)mean;
                sc[k] = ga[k] * rstd;
                sh[k] = be[k] - mf * sc[k];
            }
        }
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < CK; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) a += mish_fast_f(xv[k][e] * sc[k][e] + sh[k][e]) * wr[k][e];
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64);
            a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
        }
        if (sub == 0 && m0 + pl < me) {
            a += b4;                                                 // (channels Cs .. 3 of the padded pixel: exactly 0 -- their weights and bias are)
            *reinterpret_cast<f32x4*>(y + (size_t)m * ldy) = a;
        }
    }
}
// dgrad: dx[px][c] (+)= sum_j dy[px][j] w[c][j]; thread = (channel quad, pixel lane)
template <bool DX16 = false>
__global__ __launch_bounds__(256) void small_cout_dgrad_kernel(int M, int C, int Cs, const float* __restrict__ dy, int lddy,
                                                               const float* __restrict__ w, void* __restrict__ dx, int lddx, int acc) {
    const int nq = C / 4;
    const int q = threadIdx.x % nq, psub = threadIdx.x / nq, pp = 256 / nq;
    f32x4 wr[4];                                                     // wr[e][j] = w[4q+e][j]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float* wp = w + (size_t)(4 * q + e) * Cs;
        wr[e] = f32x4{wp[0], Cs > 1 ? wp[1] : 0.f, Cs > 2 ? wp[2] : 0.f, Cs > 3 ? wp[3] : 0.f};
    }
    for (int m = blockIdx.x * pp + psub; m < M; m += gridDim.x * pp) {
        const float* gp = dy + (size_t)m * lddy;
        f32x4 g = {gp[0], Cs > 1 ? gp[1] : 0.f, Cs > 2 ? gp[2] : 0.f, Cs > 3 ? gp[3] : 0.f};
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const f32x4 pr = g * wr[e]; o[e] = (pr.x + pr.y) + (pr.z + pr.w); }
        const size_t po = (size_t)m * lddx + 4 * q;
        if (acc) o += ldq<DX16>(dx, po);
        stq<DX16>(dx, po, o);
    }
}
// wgrad: dW[c][j] += sum_px x[px][c] dy[px][j]; thread = (channel quad, pixel lane), partial tile [C][4]
// DG (round 6): the data gradient of the same layer rides along -- same thread mapping, same dy: dx[px][4q + e] (+)= sum_j dy[px][j] w[4q + e][j],
// small_cout_dgrad_kernel's arithmetic; one pass over (x, dy) instead of two launches
template <bool X16 = false, bool DG = false, bool DX16 = false>
__global__ __launch_bounds__(256) void small_cout_wgrad_kernel(int M, int C, int Cs, const void* __restrict__ x, int ldx,
                                                               const float* __restrict__ dy, int lddy, float* __restrict__ dW,
                                                               float* __restrict__ ws, const float* __restrict__ w = nullptr,
                                                               void* __restrict__ dx = nullptr, int lddx = 0, int dacc = 0) {
    extern __shared__ float red[];
    const int nq = C / 4;
    const int q = threadIdx.x % nq, psub = threadIdx.x / nq, pp = 256 / nq;
    f32x4 acc[4];                                                    // acc[e][j] for channel 4q+e
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 wr[DG ? 4 : 1];                                            // wr[e][j] = w[4q+e][j]
    if constexpr (DG) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* wp = w + (size_t)(4 * q + e) * Cs;
            wr[e] = f32x4{wp[0], Cs > 1 ? wp[1] : 0.f, Cs > 2 ? wp[2] : 0.f, Cs > 3 ? wp[3] : 0.f};
        }
    }
    const int per = (M + gridDim.x - 1) / gridDim.x;
    const int mb = blockIdx.x * per, me = min(M, mb + per);
    int m = mb + psub;
    f32x4 xv = ldq<X16>(x, (size_t)min(m, M - 1) * ldx + 4 * q);
    for (; m < me; m += pp) {
        const f32x4 xnext = ldq<X16>(x, (size_t)min(m + pp, M - 1) * ldx + 4 * q);
        const float* gp = dy + (size_t)m * lddy;
        const f32x4 g = {gp[0], Cs > 1 ? gp[1] : 0.f, Cs > 2 ? gp[2] : 0.f, Cs > 3 ? gp[3] : 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += xv[e] * g;
        if constexpr (DG) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const f32x4 pr = g * wr[e]; o[e] = (pr.x + pr.y) + (pr.z + pr.w); }
            const size_t po = (size_t)m * lddx + 4 * q;
            if (dacc) o += ldq<DX16>(dx, po);
            stq<DX16>(dx, po, o);
        }
        xv = xnext;
    }
    block_reduce_quads<4>(acc, nq, red);
    if (threadIdx.x < nq) {
        if (ws) {
            float* out = ws + (size_t)blockIdx.x * C * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x4*>(out + (size_t)(4 * q + e) * 4) = acc[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                for (int j = 0; j < Cs; ++j) atomicAdd(dW + (size_t)(4 * q + e) * Cs + j, acc[e][j]);
        }
    }
}

int log2_exact(int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; }

}  // namespace

extern "C" size_t mi_conv_small_wgrad_workspace(int outputs) { return (size_t)WG_BLOCKS * (size_t)outputs * sizeof(float); }

namespace {
// the whole-row-tile kernels' geometry (small_cin3x3_fwd_tiled_kernel / small_cin3x3_wgrad_tiled_kernel)
bool cin_tiled_geom(int ks, int H, int W, int ldx, const void* x) {
    const int w_sh = log2_exact(W), hw_sh = log2_exact(H * W);
    const int rows = w_sh >= 0 && W <= 64 ? 64 / W : 0;
    return ks == 3 && w_sh >= 0 && hw_sh >= 0 && ldx == 4 && ((uintptr_t)x & 15) == 0 && rows >= 1 && H % rows == 0 && (rows + 2) * (W + 2) <= 256;
}
}  // namespace

// bf16 wide tensors (round 4): 1 when BOTH the forward conv can write its output as bf16 and the weight gradient can read a bf16 dy
// for this layer (the whole-row-tile kernels take it)
extern "C" int mi_conv_small_cin_bf16_supported(int ks, int N, int H, int W, int Cin, int Cout, int ldx) {
    static const int tiled = (int)mi_knob("MI_SMALL_CIN_TILED", 1);
    return (tiled && N > 0 && Cin >= 1 && Cin <= 4 && (Cout == 64 || Cout == 128 || Cout == 256) && ((long)N * H * W) % 64 == 0 && cin_tiled_geom(ks, H, W, ldx, nullptr) &&
            (size_t)3 * 9 * Cin * (Cout / 4) * 16 <= 48 * 1024) ? 1 : 0;
}

extern "C" int mi_conv_small_cin_fwd_io(int ks, int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w,
                                        const float* bias, void* yv, int ldy, int y_bf16, void* stream) {
    MI_REQUIRE(x && w && yv && (ks == 1 || ks == 3) && Cin >= 1 && Cin <= 4 && Cout % 4 == 0 && Cout <= 1024 && 256 % (Cout / 4) == 0 &&
               ldy % 4 == 0 && (((uintptr_t)yv & (y_bf16 ? 7 : 15)) | ((uintptr_t)w & 15)) == 0,
               "needs ks 1|3, Cin <= 4, Cout a multiple of 4 with Cout/4 dividing 256, 16-byte aligned y / w");
    float* y = (float*)yv;
    const int pp = 256 / (Cout / 4);
    long blocks = ((long)N * H * W + 2 * pp - 1) / (2 * pp); if (blocks > 4096) blocks = 4096;
    const int w_sh = log2_exact(W), hw_sh = log2_exact(H * W);
    const bool pow2 = w_sh >= 0 && hw_sh >= 0;
    const bool vec = ldx % 4 == 0 && ((uintptr_t)x & 15) == 0;
    hipStream_t st = (hipStream_t)stream;
    {   // 3x3 on whole-row tiles of 64 pixels with the taps in LDS (see small_cin3x3_fwd_tiled_kernel)
        static const int tiled = (int)mi_knob("MI_SMALL_CIN_TILED", 1);
        const int rows = w_sh >= 0 && W <= 64 ? 64 / W : 0;
        if (tiled && cin_tiled_geom(ks, H, W, ldx, x) && Cout <= 256 && ((long)N * H * W) % 64 == 0) {
            const int ntiles = N * H * W / 64;
            const int grid = ntiles < 768 ? ntiles : 768;             // (one round at three workgroups per CU, as measured for the dual form)
            const size_t lds = (size_t)2 * (rows + 2) * (W + 2) * 16;
#define MI_TILED(CIN) do { \
                if (y_bf16) hipLaunchKernelGGL((small_cin3x3_fwd_tiled_kernel<CIN, true>), dim3(grid), dim3(256), lds, st, N, H, W, Cout, x, ldx, w, bias, yv, ldy, w_sh, ntiles); \
                else hipLaunchKernelGGL((small_cin3x3_fwd_tiled_kernel<CIN, false>), dim3(grid), dim3(256), lds, st, N, H, W, Cout, x, ldx, w, bias, yv, ldy, w_sh, ntiles); } while (0)
            switch (Cin) { case 1: MI_TILED(1); break; case 2: MI_TILED(2); break; case 3: MI_TILED(3); break; default: MI_TILED(4); break; }
#undef MI_TILED
            MI_LAUNCH_CHECK();
            return 0;
        }
    }
    MI_REQUIRE(!y_bf16, "bf16 output: only on the whole-row-tile kernel (mi_conv_small_cin_bf16_supported)");
#define MI_GO(CIN, KS) do { \
        if (pow2 && vec) hipLaunchKernelGGL((small_cin_fwd_kernel<CIN, KS, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, N, H, W, Cout, x, ldx, w, bias, y, ldy, w_sh, hw_sh); \
        else if (vec) hipLaunchKernelGGL((small_cin_fwd_kernel<CIN, KS, false, true>), dim3((unsigned)blocks), dim3(256), 0, st, N, H, W, Cout, x, ldx, w, bias, y, ldy, 0, 0); \
        else hipLaunchKernelGGL((small_cin_fwd_kernel<CIN, KS, false, false>), dim3((unsigned)blocks), dim3(256), 0, st, N, H, W, Cout, x, ldx, w, bias, y, ldy, 0, 0); } while (0)
#define MI_GO_CIN(KS) do { switch (Cin) { case 1: MI_GO(1, KS); break; case 2: MI_GO(2, KS); break; case 3: MI_GO(3, KS); break; default: MI_GO(4, KS); break; } } while (0)
    if (ks == 3) MI_GO_CIN(3); else MI_GO_CIN(1);
#undef MI_GO
    MI_LAUNCH_CHECK();
    return 0;
}
// The first ResnetBlock's block1 conv (3x3) AND its res_conv (1x1) on the same image in one launch: y3 (fp32 or bf16) = conv3x3(x, w3) + bias3,
// y1 (fp32) = conv1x1(x, w1) + bias1.  Whole-row-tile geometry only (mi_conv_small_cin_fwd_dual_supported).
extern "C" int mi_conv_small_cin_fwd_dual_supported(int N, int H, int W, int Cin, int Cout, int ldx) {
    static const int tiled = (int)mi_knob("MI_SMALL_CIN_TILED", 1);
    return (tiled && N > 0 && Cin >= 1 && Cin <= 4 && Cout % 4 == 0 && Cout <= 256 && 256 % (Cout / 4) == 0 && ((long)N * H * W) % 64 == 0 &&
            cin_tiled_geom(3, H, W, ldx, nullptr)) ? 1 : 0;
}
struct CinChores { void* zero; size_t zero_bytes; const float* gsrc; const long long* gidx; float* gdst; int grow, gB; };
static int cin_dual_launch(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w3, const float* bias3,
                           void* y3, int ldy3, int y3_bf16, const float* w1, const float* bias1, float* y1, int ldy1, const CinChores& ch, void* stream);
extern "C" int mi_conv_small_cin_fwd_dual(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w3, const float* bias3,
                                          void* y3, int ldy3, int y3_bf16, const float* w1, const float* bias1, float* y1, int ldy1, void* stream) {
    return cin_dual_launch(N, H, W, Cin, Cout, x, ldx, w3, bias3, y3, ldy3, y3_bf16, w1, bias1, y1, ldy1, CinChores{}, stream);
}
// ... that also zero-fills `zero` (zero_bytes % 8 == 0, 8-byte aligned): the pool of GroupNorm sums the conv epilogues of the forward's LATER launches
// add into (this launch is the first of a forward)
// ... and / or gathers rows gather_dst[b][0 .. row) = gather_src[gather_idx[b]][0 .. row) for b < gather_n (the sampler's time-bias table; row % 4 == 0,
// 16-byte aligned).  Either chore may be absent (null).
extern "C" int mi_conv_small_cin_fwd_dual_chores(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w3, const float* bias3,
                                               void* y3, int ldy3, int y3_bf16, const float* w1, const float* bias1, float* y1, int ldy1,
                                               void* zero, size_t zero_bytes, const float* gather_src, const void* gather_idx, float* gather_dst,
                                               int gather_row, int gather_n, void* stream) {
    MI_REQUIRE(!zero || (zero_bytes % 8 == 0 && ((uintptr_t)zero & 7) == 0 && zero_bytes < (1u << 30)), "zero fill: 8-byte aligned, a multiple of 8 bytes");
    MI_REQUIRE(!gather_dst || (gather_src && gather_idx && gather_row > 0 && gather_row % 4 == 0 && gather_n > 0 && (long)gather_row * gather_n < (1L << 30) &&
                               (((uintptr_t)gather_src | (uintptr_t)gather_dst) & 15) == 0), "row gather: rows of a multiple of 4 floats, 16-byte aligned");
    CinChores ch{zero, zero ? zero_bytes : 0, gather_src, (const long long*)gather_idx, gather_dst, gather_dst ? gather_row : 0, gather_dst ? gather_n : 0};
    return cin_dual_launch(N, H, W, Cin, Cout, x, ldx, w3, bias3, y3, ldy3, y3_bf16, w1, bias1, y1, ldy1, ch, stream);
}
static int cin_dual_launch(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w3, const float* bias3,
                           void* y3, int ldy3, int y3_bf16, const float* w1, const float* bias1, float* y1, int ldy1, const CinChores& ch, void* stream) {
    MI_REQUIRE(x && w3 && y3 && w1 && y1 && mi_conv_small_cin_fwd_dual_supported(N, H, W, Cin, Cout, ldx) && ((uintptr_t)x & 15) == 0 &&
               ldy3 % 4 == 0 && ldy1 % 4 == 0 && (((uintptr_t)y3 & (y3_bf16 ? 7 : 15)) | ((uintptr_t)y1 & 15) | ((uintptr_t)w3 & 15) | ((uintptr_t)w1 & 15)) == 0,
               "needs the whole-row-tile geometry (power-of-two W <= 64, ldx == 4), Cout / 4 dividing 256, aligned operands");
    const int w_sh = log2_exact(W), rows = 64 / W;
    const int ntiles = N * H * W / 64;
    // (three workgroups per CU fit -- ~140 registers of weights per thread; one round of 768: B = 64 18.7 -> 16.6 us, B = 128 29.0 -> 24.6)
    const int grid = ntiles < 768 ? ntiles : 768;
    const size_t lds = (size_t)2 * (rows + 2) * (W + 2) * 16;
    hipStream_t st = (hipStream_t)stream;
#define MI_TILED2(CIN) do { \
        if (y3_bf16) hipLaunchKernelGGL((small_cin3x3_fwd_tiled_kernel<CIN, true, true>), dim3(grid), dim3(256), lds, st, N, H, W, Cout, x, ldx, w3, bias3, y3, ldy3, w_sh, ntiles, w1, bias1, y1, ldy1, (unsigned long long*)ch.zero, (int)(ch.zero_bytes / 8), ch.gsrc, ch.gidx, ch.gdst, ch.grow, ch.gB); \
        else hipLaunchKernelGGL((small_cin3x3_fwd_tiled_kernel<CIN, false, true>), dim3(grid), dim3(256), lds, st, N, H, W, Cout, x, ldx, w3, bias3, y3, ldy3, w_sh, ntiles, w1, bias1, y1, ldy1, (unsigned long long*)ch.zero, (int)(ch.zero_bytes / 8), ch.gsrc, ch.gidx, ch.gdst, ch.grow, ch.gB); } while (0)
    switch (Cin) { case 1: MI_TILED2(1); break; case 2: MI_TILED2(2); break; case 3: MI_TILED2(3); break; default: MI_TILED2(4); break; }
#undef MI_TILED2
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_conv_small_cin_fwd(int ks, int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w,
                                     const float* bias, float* y, int ldy, void* stream) {
    return mi_conv_small_cin_fwd_io(ks, N, H, W, Cin, Cout, x, ldx, w, bias, y, ldy, 0, stream);
}

extern "C" int mi_conv_small_cin_wgrad_io(int ks, int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const void* dyv,
                                          int lddy, int dy_bf16, float* dW, void* workspace, size_t ws_bytes, void* stream) {
    const int na = ks * ks * Cin;
    MI_REQUIRE(x && dyv && dW && (ks == 1 || ks == 3) && Cin >= 1 && Cin <= 4 && (Cout == 64 || Cout == 128 || Cout == 256) && lddy % 4 == 0 &&
               ((uintptr_t)dyv & (dy_bf16 ? 7 : 15)) == 0 && (size_t)3 * na * (Cout / 4) * 16 <= 48 * 1024,
               "needs ks 1|3, Cin <= 4, Cout in {64, 128, 256} (3x3: {64, 128}), 16-byte aligned dy rows");
    const float* dy = (const float*)dyv;
    float* ws = (workspace && ws_bytes >= mi_conv_small_wgrad_workspace(na * Cout)) ? (float*)workspace : nullptr;
    const int w_sh = log2_exact(W), hw_sh = log2_exact(H * W);
    const bool pow2 = w_sh >= 0 && hw_sh >= 0;
    const bool vec = ldx % 4 == 0 && ((uintptr_t)x & 15) == 0;
    const size_t lds = (size_t)3 * na * (Cout / 4) * 4 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    {   // 3x3 on whole-row tiles of 64 pixels with the taps in LDS (small_cin3x3_wgrad_tiled_kernel); same partial-tile contract
        static const int tiled = (int)mi_knob("MI_SMALL_CIN_TILED", 1);
        const int rows = w_sh >= 0 && W <= 64 ? 64 / W : 0;
        if (tiled && cin_tiled_geom(ks, H, W, ldx, x) && (Cout >= 128 || dy_bf16) && ((long)N * H * W) % 64 == 0) {     // (fp32 dy at 64 channels: the untiled kernel, as measured in round 2)
            const int ntiles = N * H * W / 64;
            const size_t lds2 = lds + (size_t)2 * (rows + 2) * (W + 2) * 16;
#define MI_TILED(CIN) do { \
                if (dy_bf16) hipLaunchKernelGGL((small_cin3x3_wgrad_tiled_kernel<CIN, true>), dim3(WG_BLOCKS), dim3(256), lds2, st, N, H, W, Cout, x, ldx, dyv, lddy, dW, ws, w_sh, ntiles); \
                else hipLaunchKernelGGL((small_cin3x3_wgrad_tiled_kernel<CIN, false>), dim3(WG_BLOCKS), dim3(256), lds2, st, N, H, W, Cout, x, ldx, dyv, lddy, dW, ws, w_sh, ntiles); } while (0)
            switch (Cin) { case 1: MI_TILED(1); break; case 2: MI_TILED(2); break; case 3: MI_TILED(3); break; default: MI_TILED(4); break; }
#undef MI_TILED
            if (ws) hipLaunchKernelGGL(partial_sum_kernel, dim3((na * Cout + 31) / 32), dim3(256), 0, st, ws, WG_BLOCKS, na * Cout, dW, 0);
            MI_LAUNCH_CHECK();
            return 0;
        }
    }
    MI_REQUIRE(!dy_bf16, "bf16 dy: only on the whole-row-tile kernel (mi_conv_small_cin_bf16_supported)");
#define MI_GO(CIN, KS) do { \
        if (pow2 && vec) hipLaunchKernelGGL((small_cin_wgrad_kernel<CIN, KS, true, true>), dim3(WG_BLOCKS), dim3(256), lds, st, N, H, W, Cout, x, ldx, dy, lddy, dW, ws, w_sh, hw_sh); \
        else if (vec) hipLaunchKernelGGL((small_cin_wgrad_kernel<CIN, KS, false, true>), dim3(WG_BLOCKS), dim3(256), lds, st, N, H, W, Cout, x, ldx, dy, lddy, dW, ws, 0, 0); \
        else hipLaunchKernelGGL((small_cin_wgrad_kernel<CIN, KS, false, false>), dim3(WG_BLOCKS), dim3(256), lds, st, N, H, W, Cout, x, ldx, dy, lddy, dW, ws, 0, 0); } while (0)
    if (ks == 3) MI_GO_CIN(3); else MI_GO_CIN(1);
#undef MI_GO
#undef MI_GO_CIN
    if (ws) hipLaunchKernelGGL(partial_sum_kernel, dim3((na * Cout + 31) / 32), dim3(256), 0, st, ws, WG_BLOCKS, na * Cout, dW, 0);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_conv_small_cin_wgrad(int ks, int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* dy,
                                       int lddy, float* dW, void* workspace, size_t ws_bytes, void* stream) {
    return mi_conv_small_cin_wgrad_io(ks, N, H, W, Cin, Cout, x, ldx, dy, lddy, 0, dW, workspace, ws_bytes, stream);
}

// the 3x3 forms of the two entry points above (kept for callers of ABI version 1)
extern "C" int mi_conv3x3_small_cin_fwd(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w,
                                        const float* bias, float* y, int ldy, void* stream) {
    return mi_conv_small_cin_fwd(3, N, H, W, Cin, Cout, x, ldx, w, bias, y, ldy, stream);
}
extern "C" int mi_conv3x3_small_cin_wgrad(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* dy,
                                          int lddy, float* dW, void* stream) {
    return mi_conv_small_cin_wgrad(3, N, H, W, Cin, Cout, x, ldx, dy, lddy, dW, nullptr, 0, stream);
}

extern "C" int mi_conv1x1_small_cout_io(int op, int M, int C, int Cs, const void* a, int lda, const float* b, int ldb,
                                        const float* w, const float* bias, void* out, int ldo, int accumulate, int wide_bf16,
                                        void* workspace, size_t ws_bytes, void* stream) {
    // op 0: forward  (a = x[M][C], out = y[M][Cs], bias)        op 1: dgrad (a = dy[M][Cs], out = dx[M][C])
    // op 2: wgrad    (a = x[M][C], b = dy[M][Cs], out = dW[C][Cs])
    // wide_bf16: the C-channel tensor (x of op 0 / 2, dx of op 1) is stored as bf16; the Cs-channel side and the weights stay fp32
    MI_REQUIRE(a && out && (op == 2 || w) && Cs >= 1 && Cs <= 4 && (C == 32 || C == 64 || C == 128 || C == 256) && M > 0,
               "needs Cs <= 4 and C in {32, 64, 128, 256}");
    hipStream_t st = (hipStream_t)stream;
    const uintptr_t wal = wide_bf16 ? 7 : 15;
    if (op == 0) {
        MI_REQUIRE(C <= 128 && lda % 4 == 0 && ((uintptr_t)a & wal) == 0, "forward: C <= 128, 16-byte (bf16: 8-byte) aligned x rows");
        int blocks = (M + 31) / 32; if (blocks > 1024) blocks = 1024;
#define MI_FWD(CK) do { \
            if (wide_bf16) hipLaunchKernelGGL((small_cout_fwd_kernel<CK, true>), dim3(blocks), dim3(256), 0, st, M, Cs, a, lda, w, bias, (float*)out, ldo); \
            else hipLaunchKernelGGL((small_cout_fwd_kernel<CK, false>), dim3(blocks), dim3(256), 0, st, M, Cs, a, lda, w, bias, (float*)out, ldo); } while (0)
        if (C == 128) MI_FWD(4); else if (C == 64) MI_FWD(2); else MI_FWD(1);
#undef MI_FWD
    } else if (op == 1) {
        MI_REQUIRE(ldo % 4 == 0 && ((uintptr_t)out & wal) == 0, "dgrad: 16-byte (bf16: 8-byte) aligned dx rows");
        const int pp = 256 / (C / 4);
        long blocks = ((long)M + pp - 1) / pp; if (blocks > 4096) blocks = 4096;
        if (wide_bf16) hipLaunchKernelGGL(small_cout_dgrad_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, M, C, Cs, (const float*)a, lda, w, out, ldo, accumulate);
        else hipLaunchKernelGGL(small_cout_dgrad_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, M, C, Cs, (const float*)a, lda, w, out, ldo, accumulate);
    } else if (op == 2) {
        MI_REQUIRE(b && C >= 64 && lda % 4 == 0 && ((uintptr_t)a & wal) == 0, "wgrad: needs dy, C in {64, 128, 256}, aligned x rows");
        float* ws = (workspace && ws_bytes >= mi_conv_small_wgrad_workspace(C * 4)) ? (float*)workspace : nullptr;
        const size_t lds = (size_t)3 * 4 * (C / 4) * 4 * sizeof(float);
        if (wide_bf16) hipLaunchKernelGGL(small_cout_wgrad_kernel<true>, dim3(WG_BLOCKS), dim3(256), lds, st, M, C, Cs, a, lda, b, ldb, (float*)out, ws);
        else hipLaunchKernelGGL(small_cout_wgrad_kernel<false>, dim3(WG_BLOCKS), dim3(256), lds, st, M, C, Cs, a, lda, b, ldb, (float*)out, ws);
        if (ws) hipLaunchKernelGGL(partial_sum_kernel, dim3((C * 4 + 31) / 32), dim3(256), 0, st, ws, WG_BLOCKS, C * 4, (float*)out, Cs);
    } else {
        return mi_set_error(-1, "mi_conv1x1_small_cout: op must be 0, 1 or 2");
    }
    MI_LAUNCH_CHECK();
    return 0;
}

// y[M][Cs] = bias + conv1x1(mish(groupnorm(x))), x bf16 [M][ldx] = the producing conv's output, sums = what its epilogue left (see
// small_cout_fwd_gn_kernel).  C = 64 or 128, (C / G) % 16 == 0.
extern "C" int mi_conv1x1_small_cout_gn_supported(int C, int Cs, int G) {
    return ((C == 64 || C == 128) && Cs >= 1 && Cs <= 4 && G > 0 && C % G == 0 && (C / G) % 16 == 0) ? 1 : 0;
}
extern "C" int mi_conv1x1_small_cout_gn_fwd(int M, int HW, int C, int Cs, const void* x_bf16, int ldx, const void* sums, const float* gamma,
                                            const float* beta, int G, float eps, const float* w, const float* bias, float* y, int ldy, void* stream) {
    MI_REQUIRE(x_bf16 && sums && gamma && beta && w && y && M > 0 && HW > 0 && M % HW == 0 && mi_conv1x1_small_cout_gn_supported(C, Cs, G) &&
               ldx % 4 == 0 && ldy == 4 && (((uintptr_t)x_bf16 & 7) | ((uintptr_t)sums & 15) | ((uintptr_t)gamma & 15) | ((uintptr_t)beta & 15) | ((uintptr_t)y & 15)) == 0,
               "needs C in {64, 128}, (C / G) % 16 == 0, Cs <= 4, y as padded 4-channel pixels, aligned operands");
    // contiguous runs of pixels per workgroup: a divisor of H * W where there is one (one sample per workgroup), about 512 workgroups
    // (B = 64, 32x32 x 128: 24.0 us with 2 048 workgroups, 15.5 with 1 024, 12.0 with 512 or 256, 17.9 with 128 -- the per-workgroup set-up, 48 weight
    //  loads and the coefficients in double, wants to be paid a few hundred times, not a few thousand)
    int ppw = 32;
    while ((long)M / ppw > 768 && HW % (2 * ppw) == 0) ppw *= 2;
    const int blocks = (M + ppw - 1) / ppw;
    hipStream_t st = (hipStream_t)stream;
    if (C == 128) hipLaunchKernelGGL((small_cout_fwd_gn_kernel<4>), dim3(blocks), dim3(256), 0, st, M, HW, Cs, (const uint16_t*)x_bf16, ldx, (const long long*)sums, gamma, beta, G, eps, w, bias, y, ldy, ppw);
    else hipLaunchKernelGGL((small_cout_fwd_gn_kernel<2>), dim3(blocks), dim3(256), 0, st, M, HW, Cs, (const uint16_t*)x_bf16, ldx, (const long long*)sums, gamma, beta, G, eps, w, bias, y, ldy, ppw);
    MI_LAUNCH_CHECK();
    return 0;
}

// Backward of y = x w + bias in one pass over (x, dy): dW[c][j] += sum_px x[px][c] dy[px][j] AND dx[px][c] (+)= sum_j dy[px][j] w[c][j] (the two launches
// mi_conv1x1_small_cout_io op 2 / op 1 make; bitwise their results).  x_bf16 / dx_bf16: the wide tensors' storage.  workspace as for op 2.
extern "C" int mi_conv1x1_small_cout_bwd(int M, int C, int Cs, const void* x, int ldx, int x_bf16, const float* dy, int lddy, const float* w,
                                         float* dW, void* dx, int lddx, int dx_bf16, int accumulate_dx, void* workspace, size_t ws_bytes, void* stream) {
    MI_REQUIRE(x && dy && w && dW && dx && Cs >= 1 && Cs <= 4 && (C == 64 || C == 128 || C == 256) && M > 0 && ldx % 4 == 0 && lddx % 4 == 0 &&
               ((uintptr_t)x & (x_bf16 ? 7 : 15)) == 0 && ((uintptr_t)dx & (dx_bf16 ? 7 : 15)) == 0, "needs Cs <= 4, C in {64, 128, 256}, aligned rows");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (workspace && ws_bytes >= mi_conv_small_wgrad_workspace(C * 4)) ? (float*)workspace : nullptr;
    const size_t lds = (size_t)3 * 4 * (C / 4) * 4 * sizeof(float);
#define MI_BWD(X16, D16) hipLaunchKernelGGL((small_cout_wgrad_kernel<X16, true, D16>), dim3(WG_BLOCKS), dim3(256), lds, st, M, C, Cs, x, ldx, dy, lddy, dW, ws, w, dx, lddx, accumulate_dx)
    if (x_bf16) { if (dx_bf16) MI_BWD(true, true); else MI_BWD(true, false); }
    else { if (dx_bf16) MI_BWD(false, true); else MI_BWD(false, false); }
#undef MI_BWD
    if (ws) hipLaunchKernelGGL(partial_sum_kernel, dim3((C * 4 + 31) / 32), dim3(256), 0, st, ws, WG_BLOCKS, C * 4, dW, Cs);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_conv1x1_small_cout_ws(int op, int M, int C, int Cs, const float* a, int lda, const float* b, int ldb,
                                        const float* w, const float* bias, float* out, int ldo, int accumulate,
                                        void* workspace, size_t ws_bytes, void* stream) {
    return mi_conv1x1_small_cout_io(op, M, C, Cs, a, lda, b, ldb, w, bias, out, ldo, accumulate, 0, workspace, ws_bytes, stream);
}

extern "C" int mi_conv1x1_small_cout(int op, int M, int C, int Cs, const float* a, int lda, const float* b, int ldb,
                                     const float* w, const float* bias, float* out, int ldo, int accumulate, void* stream) {
    return mi_conv1x1_small_cout_ws(op, M, C, Cs, a, lda, b, ldb, w, bias, out, ldo, accumulate, nullptr, 0, stream);
}
