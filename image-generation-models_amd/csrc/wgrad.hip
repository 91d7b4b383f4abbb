// Weight gradients on the matrix cores:  dW[tap][i][j] += sum_m P[pix_p(m,tap)][i] * Q[pix_q(m,tap)][j]
//
// The contraction runs over pixels, which is the strided axis of both NHWC operands, so both
// are transposed on their way into LDS: a thread fetches a float4 (4 channels) of two
// consecutive pixels, pairs them along the pixel axis with v_cvt_pk_bf16_f32 (bf16 mode) and
// stores dwords to Ps[channel][pixel]; lanes are arranged 16 pixel-pairs x 2 channel-quads
// per half-wave so those stores hit 32 distinct banks.  One workgroup = one (tap, i-tile,
// j-tile, pixel-slice); slices are combined with fp32 atomics.
// Replaces aten::convolution_backward (weight) and addmm-backward for the layers listed in
// include/mi_ddpm.h.
#include <stdlib.h>
#include "common.h"

namespace {

struct WgradArgs {
    const float* P; const float* P2; const float* Q; float* dW;
    int N, GH, GW, DH, DW, Ci, Cj, KH, KW, stride, pad, gather_i, I1, ldp, ldp2, ldq;
    int Mtot, chunk, splits;
    int vec;
    int xcd_map, gx, gy;       // fast kernel: XCD-aware 1-D grid (gx, gy = tiles along Ci, Cj)
};

template <int MODE> struct WElem { using type = float; static constexpr int PITCH = 36; };
template <> struct WElem<1> { using type = uint16_t; static constexpr int PITCH = 40; };

__device__ __forceinline__ float4 ldv(const float* p, int nvalid, bool vec) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4 && vec) r = *reinterpret_cast<const float4*>(p);
    else if (nvalid > 0) {
        r.x = p[0];
        if (nvalid > 1) r.y = p[1];
        if (nvalid > 2) r.z = p[2];
        if (nvalid > 3) r.w = p[3];
    }
    return r;
}

template <int MODE, int BI, int BJ>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
    MI_PRIO_UP();
    using E = typename WElem<MODE>::type;
    constexpr int PITCH = WElem<MODE>::PITCH;
    constexpr int P_IT = BI / 64, Q_IT = BJ / 64;      // channel quads per thread
    constexpr int WI = BI / 2, WJ = BJ / 2;
    constexpr int MI = WI / 32, NJ = WJ / 32;

    __shared__ __attribute__((aligned(16))) E lds[(BI + BJ) * PITCH];
    E* Ps = lds;
    E* Qs = lds + BI * PITCH;

    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int wi = wv >> 1, wj = wv & 1;
    const int i0 = blockIdx.x * BI, j0 = blockIdx.y * BJ;
    const int tap = blockIdx.z / a.splits, split = blockIdx.z % a.splits;
    const int ky = tap / a.KW, kx = tap % a.KW;
    const int mbeg = split * a.chunk;
    const int mend = min(a.Mtot, mbeg + a.chunk);

    const int kd = t & 15;
    // the two pixels this thread stages: m = mcur + 2*kd + {0,1}; decode incrementally
    int pm[2], pn[2], pyy[2], pxx[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int m = mbeg + 2 * kd + h;
        pm[h] = m;
        int n = m / (a.DH * a.DW);
        int rem = m - n * (a.DH * a.DW);
        pn[h] = n; pyy[h] = rem / a.DW; pxx[h] = rem - pyy[h] * a.DW;
    }

    float4 rp[2 * P_IT], rq[2 * Q_IT];

    auto load_step = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool inr = pm[h] < mend;
            const int gy = pyy[h] * a.stride - a.pad + ky, gx = pxx[h] * a.stride - a.pad + kx;
            const bool gok = inr && gy >= 0 && gy < a.GH && gx >= 0 && gx < a.GW;
            const size_t gpix = (size_t)(pn[h] * a.GH + gy) * a.GW + gx;
            const size_t dpix = (size_t)pm[h];
            const bool pok = a.gather_i ? gok : inr;
            const bool qok = a.gather_i ? inr : gok;
            const size_t ppix = a.gather_i ? gpix : dpix;
            const size_t qpix = a.gather_i ? dpix : gpix;
#pragma unroll
            for (int p = 0; p < P_IT; ++p) {
                int c = i0 + ((t >> 4) + 16 * p) * 4;
                const float* src = a.P; int ld = a.ldp; int cc = c;
                if (c >= a.I1) { src = a.P2; ld = a.ldp2; cc = c - a.I1; }
                int lim = (c < a.I1 ? a.I1 : a.Ci) - c;
                rp[2 * p + h] = ldv(src + (pok ? ppix * ld + cc : 0), pok ? lim : 0, a.vec);
            }
#pragma unroll
            for (int p = 0; p < Q_IT; ++p) {
                int c = j0 + ((t >> 4) + 16 * p) * 4;
                rq[2 * p + h] = ldv(a.Q + (qok ? qpix * a.ldq + c : 0), qok ? a.Cj - c : 0, a.vec);
            }
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            pm[h] += 32; pxx[h] += 32;
            while (pxx[h] >= a.DW) { pxx[h] -= a.DW; ++pyy[h]; }
            while (pyy[h] >= a.DH) { pyy[h] -= a.DH; ++pn[h]; }
        }
    };
    auto store_one = [&](E* dst, const float4& lo4, const float4& hi4, int cq) {
        const float lo[4] = {lo4.x, lo4.y, lo4.z, lo4.w};
        const float hi[4] = {hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int row = cq * 4 + j;
            if constexpr (MODE == 1)
                *reinterpret_cast<uint32_t*>(&dst[row * PITCH + 2 * kd]) = pack_bf16(lo[j], hi[j]);
            else
                *reinterpret_cast<float2*>(&dst[row * PITCH + 2 * kd]) = make_float2(lo[j], hi[j]);
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int p = 0; p < P_IT; ++p) store_one(Ps, rp[2 * p], rp[2 * p + 1], (t >> 4) + 16 * p);
#pragma unroll
        for (int p = 0; p < Q_IT; ++p) store_one(Qs, rq[2 * p], rq[2 * p + 1], (t >> 4) + 16 * p);
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int mcur = mbeg;
    bool more = mcur < mend;
    if (more) { load_step(); store_step(); }
    __syncthreads();
    while (more) {
        mcur += 32;
        more = mcur < mend;
        if (more) { advance(); load_step(); }

        const int arow = wi * WI + (l & 31), brow = wj * WJ + (l & 31), kh = l >> 5;
        if constexpr (MODE == 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[MI], bf[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(&Ps[(arow + 32 * i) * PITCH + ks * 16 + kh * 8]);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8*>(&Qs[(brow + 32 * j) * PITCH + ks * 16 + kh * 8]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                float af[MI], bf[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = Ps[(arow + 32 * i) * PITCH + 2 * kk + kh];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bf[j] = Qs[(brow + 32 * j) * PITCH + 2 * kk + kh];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) { store_step(); __syncthreads(); }
    }

    float* out = a.dW + (size_t)tap * a.Ci * a.Cj;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = i0 + wi * WI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            if (row >= a.Ci) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                int col = j0 + wj * WJ + j * 32 + (l & 31);
                if (col >= a.Cj) continue;
                atomicAdd(out + (size_t)row * a.Cj + col, acc[i][j][r]);
            }
        }
}

template <int MODE, int BI, int BJ>
void launch(const WgradArgs& a, hipStream_t st) {
    dim3 grid((a.Ci + BI - 1) / BI, (a.Cj + BJ - 1) / BJ, a.KH * a.KW * a.splits);
    hipLaunchKernelGGL((wgrad_kernel<MODE, BI, BJ>), grid, dim3(256), 0, st, a);
}


// ---------------------------------------------------------------------------------------------
// Fast path of the same weight gradient for the aligned bf16 layers that matter for throughput
// (the stride-2 Downsample conv and the ConvTranspose2d Upsample, ddpm.py:70,79): power-of-two
// dense grid, 16-byte aligned rows, channel counts that are multiples of 4.  Same LDS layout and MFMA
// schedule as wgrad_kernel, but the pixel walk is straight-line code over a ring of three
// register stages with clamped addresses (padding and the slice tail are AND masks at the LDS
// store), LDS is double-buffered and there is one barrier per 32-pixel step -- every s_waitcnt
// is an exact vmcnt(N).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BI, int BJ>
__global__ __launch_bounds__(256, 2) void wgrad_fast_kernel(const WgradArgs a, int dw_sh, int dhw_sh) {
    MI_PRIO_UP();
    constexpr int PITCH = 40;
    constexpr int P_IT = BI / 64, Q_IT = BJ / 64;      // channel quads per thread
    constexpr int WI = BI / 2, WJ = BJ / 2;
    constexpr int MI = WI / 32, NJ = WJ / 32;
    constexpr int BUF = (BI + BJ) * PITCH;

    __shared__ __attribute__((aligned(16))) uint16_t lds[2 * BUF];

    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int wi = wv >> 1, wj = wv & 1;
    // Workgroup -> (i tile, j tile, tap, pixel slice).  All taps and tiles of one pixel slice read the same rows of P and
    // Q; consecutive workgroup ids go to different XCDs (separate L2s), so with xcd_map the 1-D id xcd + 8*slot is
    // slice xcd + 8*(slot / inner), inner index slot % inner: a slice's workgroups share one L2.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd_map) {
        const int inner = a.gx * a.gy * a.KH * a.KW;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int split_ = xcd + 8 * (slot / inner), in_ = slot % inner;
        if (split_ >= a.splits) return;
        bx = in_ % a.gx; by = (in_ / a.gx) % a.gy; bz = (in_ / (a.gx * a.gy)) * a.splits + split_;
    }
    const int i0 = bx * BI, j0 = by * BJ;
    const int tap = bz / a.splits, split = bz % a.splits;
    const int ky = tap / a.KW, kx = tap % a.KW;
    const int mbeg = split * a.chunk;
    const int mend = min(a.Mtot, mbeg + a.chunk);
    const int nsteps = (mend - mbeg + 31) / 32;
    const int kd = t & 15, cq = t >> 4;

    // channel sources of this thread's quads (clamped: rows / columns past the extent are never written back)
    const float* p_src[P_IT]; int p_ld[P_IT];
#pragma unroll
    for (int p = 0; p < P_IT; ++p) {
        const int c = min(i0 + (cq + 16 * p) * 4, a.Ci - 4);
        const bool second = c >= a.I1;
        p_src[p] = (second ? a.P2 : a.P) + (second ? c - a.I1 : c);
        p_ld[p] = second ? a.ldp2 : a.ldp;
    }
    const float* q_src[Q_IT];
#pragma unroll
    for (int p = 0; p < Q_IT; ++p) q_src[p] = a.Q + min(j0 + (cq + 16 * p) * 4, a.Cj - 4);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 rp[3][2 * P_IT], rq[3][2 * Q_IT]; uint32_t rk[3];
    int lm = mbeg + 2 * kd;                  // first of the two pixels this thread stages in the next stage

    auto load_stage = [&](f32x4 (&RP)[2 * P_IT], f32x4 (&RQ)[2 * Q_IT], uint32_t& keep) {
        uint32_t kp = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = lm + h;
            const bool inr = m < mend;
            const int mm = inr ? m : mbeg;
            const int n = mm >> dhw_sh, rem = mm & ((1 << dhw_sh) - 1);
            const int yy = rem >> dw_sh, xx = rem & ((1 << dw_sh) - 1);
            const int gy = yy * a.stride - a.pad + ky, gx = xx * a.stride - a.pad + kx;
            const bool gok = inr && gy >= 0 && gy < a.GH && gx >= 0 && gx < a.GW;
            const int gpix = gok ? (n * a.GH + gy) * a.GW + gx : 0;
            const int ppix = a.gather_i ? gpix : mm;
            const int qpix = a.gather_i ? mm : gpix;
            kp |= (gok ? 1u : 0u) << h;      // one operand zero is enough: the product vanishes
#pragma unroll
            for (int p = 0; p < P_IT; ++p) RP[2 * p + h] = *reinterpret_cast<const f32x4*>(p_src[p] + (size_t)ppix * p_ld[p]);
#pragma unroll
            for (int p = 0; p < Q_IT; ++p) RQ[2 * p + h] = *reinterpret_cast<const f32x4*>(q_src[p] + (size_t)qpix * a.ldq);
        }
        keep = kp;
        lm += 32;
    };
    auto store_stage = [&](int buf, const f32x4 (&RP)[2 * P_IT], const f32x4 (&RQ)[2 * Q_IT], uint32_t keep) {
        uint16_t* Ps = lds + buf * BUF;
        uint16_t* Qs = Ps + BI * PITCH;
        // lo half = pixel 2*kd, hi half = pixel 2*kd + 1; mask each half separately
        const uint32_t m = ((keep & 1u) ? 0x0000ffffu : 0u) | ((keep & 2u) ? 0xffff0000u : 0u);
#pragma unroll
        for (int p = 0; p < P_IT; ++p) {
            const int row = (cq + 16 * p) * 4;
            const f32x4 lo = RP[2 * p], hi = RP[2 * p + 1];
            *reinterpret_cast<uint32_t*>(&Ps[(row + 0) * PITCH + 2 * kd]) = pack_bf16(lo.x, hi.x) & m;
            *reinterpret_cast<uint32_t*>(&Ps[(row + 1) * PITCH + 2 * kd]) = pack_bf16(lo.y, hi.y) & m;
            *reinterpret_cast<uint32_t*>(&Ps[(row + 2) * PITCH + 2 * kd]) = pack_bf16(lo.z, hi.z) & m;
            *reinterpret_cast<uint32_t*>(&Ps[(row + 3) * PITCH + 2 * kd]) = pack_bf16(lo.w, hi.w) & m;
        }
#pragma unroll
        for (int p = 0; p < Q_IT; ++p) {
            const int row = (cq + 16 * p) * 4;
            const f32x4 lo = RQ[2 * p], hi = RQ[2 * p + 1];
            *reinterpret_cast<uint32_t*>(&Qs[(row + 0) * PITCH + 2 * kd]) = pack_bf16(lo.x, hi.x);
            *reinterpret_cast<uint32_t*>(&Qs[(row + 1) * PITCH + 2 * kd]) = pack_bf16(lo.y, hi.y);
            *reinterpret_cast<uint32_t*>(&Qs[(row + 2) * PITCH + 2 * kd]) = pack_bf16(lo.z, hi.z);
            *reinterpret_cast<uint32_t*>(&Qs[(row + 3) * PITCH + 2 * kd]) = pack_bf16(lo.w, hi.w);
        }
    };
    const int arow = (wi * WI + (l & 31)) * PITCH + (l >> 5) * 8, brow = (wj * WJ + (l & 31)) * PITCH + (l >> 5) * 8;
    auto mma_stage = [&](int buf) {
        const uint16_t* Ps = lds + buf * BUF;
        const uint16_t* Qs = Ps + BI * PITCH;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[MI], bf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&Ps[arow + 32 * i * PITCH + ks * 16]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(&Qs[brow + 32 * j * PITCH + ks * 16]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    load_stage(rp[0], rq[0], rk[0]);
    load_stage(rp[1], rq[1], rk[1]);
    load_stage(rp[2], rq[2], rk[2]);
    store_stage(0, rp[0], rq[0], rk[0]);
    __syncthreads();
    for (int s0 = 0; s0 < nsteps; s0 += 3) {       // stages past the slice are all-zero (m >= mend): up to two idle steps
        load_stage(rp[0], rq[0], rk[0]);
        mma_stage(s0 & 1);
        store_stage((s0 + 1) & 1, rp[1], rq[1], rk[1]);
        __syncthreads();
        load_stage(rp[1], rq[1], rk[1]);
        mma_stage((s0 + 1) & 1);
        store_stage(s0 & 1, rp[2], rq[2], rk[2]);
        __syncthreads();
        load_stage(rp[2], rq[2], rk[2]);
        mma_stage(s0 & 1);
        store_stage((s0 + 1) & 1, rp[0], rq[0], rk[0]);
        __syncthreads();
    }

    float* out = a.dW + (size_t)tap * a.Ci * a.Cj;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = i0 + wi * WI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            if (row >= a.Ci) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                int col = j0 + wj * WJ + j * 32 + (l & 31);
                if (col >= a.Cj) continue;
                atomicAdd(out + (size_t)row * a.Cj + col, acc[i][j][r]);
            }
        }
}

template <int BI, int BJ>
void launch_fast(WgradArgs a, int dw_sh, int dhw_sh, hipStream_t st) {
    static const int xcd_env = (int)mi_knob("MI_WGRAD_XCD", 1);
    a.gx = (a.Ci + BI - 1) / BI; a.gy = (a.Cj + BJ - 1) / BJ;
    a.xcd_map = xcd_env && a.splits >= 8;
    dim3 grid(a.gx, a.gy, a.KH * a.KW * a.splits);
    if (a.xcd_map) grid = dim3((unsigned)(a.gx * a.gy * a.KH * a.KW * ((a.splits + 7) / 8 * 8)), 1, 1);
    hipLaunchKernelGGL((wgrad_fast_kernel<BI, BJ>), grid, dim3(256), 0, st, a, dw_sh, dhw_sh);
}

// out[c] += sum_m x[m*ld + c]
__global__ __launch_bounds__(256) void colsum_kernel(int M, int C, const float* __restrict__ x, int ld,
                                                     float* __restrict__ out, int rows_per_block) {
    __shared__ float red[256];
    const int cw = min(C, 64);                    // columns handled by this block
    const int c0 = blockIdx.y * 64;
    const int tc = threadIdx.x % cw, tr = threadIdx.x / cw, nr = 256 / cw;
    const int c = c0 + tc;
    float s = 0.f;
    if (c < C && tr < nr) {
        int mb = blockIdx.x * rows_per_block, me = min(M, mb + rows_per_block);
        for (int m = mb + tr; m < me; m += nr) s += x[(size_t)m * ld + c];
    }
    red[threadIdx.x] = (tr < nr) ? s : 0.f;
    __syncthreads();
    if (tr == 0 && c < C) {
        float tot = 0.f;
        for (int r = 0; r < nr; ++r) tot += red[r * cw + tc];
        atomicAdd(out + c, tot);
    }
}

}  // namespace

extern "C" int mi_conv_wgrad(const MiWgradDesc* d, const float* P, const float* P2, const float* Q,
                             float* dW, void* stream) {
    MI_REQUIRE(d && P && Q && dW, "null argument");
    MI_REQUIRE(d->mode == 0 || d->mode == 1, "mode must be 0 or 1");
    MI_REQUIRE(d->I1 == d->Ci || (P2 && d->I1 > 0 && d->I1 < d->Ci && d->I1 % 4 == 0), "bad two-source split");
    WgradArgs a;
    a.P = P; a.P2 = P2 ? P2 : P; a.Q = Q; a.dW = dW;
    a.N = d->N; a.GH = d->GH; a.GW = d->GW; a.DH = d->DH; a.DW = d->DW; a.Ci = d->Ci; a.Cj = d->Cj;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.gather_i = d->gather_i;
    a.I1 = d->I1; a.ldp = d->ldp; a.ldp2 = P2 ? d->ldp2 : d->ldp; a.ldq = d->ldq;
    a.Mtot = d->N * d->DH * d->DW;
    a.xcd_map = 0; a.gx = a.gy = 0;
    a.vec = (d->ldp % 4 == 0) && (a.ldp2 % 4 == 0) && (d->ldq % 4 == 0) && (((uintptr_t)P & 15) == 0) &&
            (((uintptr_t)a.P2 & 15) == 0) && (((uintptr_t)Q & 15) == 0);
    const bool big = d->Ci >= 128 && d->Cj >= 128;
    const int B = big ? 128 : 64;
    long base = (long)d->KH * d->KW * ((d->Ci + B - 1) / B) * ((d->Cj + B - 1) / B);
    long splits = (1024 + base - 1) / base;
    long maxs = (a.Mtot + 63) / 64;
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    a.chunk = (int)(((a.Mtot + splits - 1) / splits + 31) / 32 * 32);
    a.splits = (a.Mtot + a.chunk - 1) / a.chunk;
    hipStream_t st = (hipStream_t)stream;
    {   // aligned bf16 layers on a power-of-two dense grid: the straight-line ring kernel
        static const int allow_fast = (int)mi_knob("MI_WGRAD_FAST", 1);
        auto lg = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
        const int dw_sh = lg(d->DW), dhw_sh = lg(d->DH * d->DW);
        if (allow_fast && d->mode == 1 && a.vec && dw_sh >= 0 && dhw_sh >= 0 && d->Ci % 4 == 0 && d->Cj % 4 == 0 && d->I1 % 4 == 0 &&
            d->Ci >= 4 && d->Cj >= 4) {
            if (big) launch_fast<128, 128>(a, dw_sh, dhw_sh, st); else launch_fast<64, 64>(a, dw_sh, dhw_sh, st);
            MI_LAUNCH_CHECK();
            return 0;
        }
    }
    if (d->mode == 1) { if (big) launch<1, 128, 128>(a, st); else launch<1, 64, 64>(a, st); }
    else              { if (big) launch<0, 128, 128>(a, st); else launch<0, 64, 64>(a, st); }
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_f32_to_bf16_colsum(size_t M, int C, const float* x, int ldx, void* y_bf16, int ldy, float* colsum, void* workspace,
                                     size_t ws_bytes, void* stream);

extern "C" int mi_colsum(int M, int C, const float* x, int ld, float* out, void* stream) {
    MI_REQUIRE(M > 0 && C > 0 && x && out, "bad argument");
    if (C % 4 == 0 && C >= 4 && C <= 1024 && ld % 4 == 0 && (((uintptr_t)x) & 15) == 0)       // float4 rows, several loads in flight
        return mi_f32_to_bf16_colsum((size_t)M, C, x, ld, nullptr, 0, out, nullptr, 0, stream);
    int rows = 256;
    dim3 grid((M + rows - 1) / rows, (C + 63) / 64);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, M, C, x, ld, out, rows);
    MI_LAUNCH_CHECK();
    return 0;
}
