// Implicit-GEMM convolution on the gfx950 matrix cores.
//
//   Y[m, j] = sum_{tap, k} A[pixel(m, tap), k] * W(tap, k, j)   (+bias, +residual, +=)
//
// m runs over output pixels (NHWC, so a row of A is a contiguous channel vector), j over
// produced channels.  One 256-thread workgroup (4 waves, 2x2) owns a BM x BN tile and walks
// K in steps of (tap, 32 channels): the fp32 activation rows and weight rows are fetched
// global->registers one step ahead, converted while being staged to LDS
// (bf16 mode: v_cvt_pk_bf16_f32, exact-fp32 mode: as is) and consumed by
// v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32 with fp32 accumulation.
// Replaces aten::convolution / convolution_backward(input) / addmm for ddpm.py:70,79,116,
// 127,134,151-152,190-192,236 (see include/mi_ddpm.h).
#include "common.h"

namespace {

struct IgemmArgs {
    const float* x; const float* x2; const float* w; const float* bias; const float* res; float* y;
    int N, IH, IW, OH, OW, K, Nc, KH, KW, stride, pad, transposed, w_kn, K1;
    int ldx, ldx2, ldy, ldr, accumulate;
    int OHc, OWc;     // per-parity-class output grid (== OH, OW unless transposed with stride > 1)
    int Mc;           // rows per class
    int vecA, vecB;   // 16-byte vector loads legal for activations / weights
    const uint16_t* wb; // optional bf16 weights [tap][Nc][K] (k contiguous); used instead of w in bf16 mode
    int ksplit;       // slices of the channel axis (blockIdx.z = class * ksplit + slice); > 1 => atomic epilogue
};

template <int MODE> struct LdsElem { using type = float; static constexpr int PITCH = 36; };
template <> struct LdsElem<1> { using type = uint16_t; static constexpr int PITCH = 40; };

__device__ __forceinline__ float4 ld4(const float* p, int nvalid, bool vec) {
    // nvalid: how many of p[0..3] are inside the logical extent (<=0 -> zeros)
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4 && vec) {
        r = *reinterpret_cast<const float4*>(p);
    } else if (nvalid > 0) {
        r.x = p[0];
        if (nvalid > 1) r.y = p[1];
        if (nvalid > 2) r.z = p[2];
        if (nvalid > 3) r.w = p[3];
    }
    return r;
}

template <int MODE, int BM, int BN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs a) {
    using E = typename LdsElem<MODE>::type;
    constexpr int PITCH = LdsElem<MODE>::PITCH;
    constexpr int A_IT = BM / 32;          // float4 per thread for the A tile (BM x 32)
    constexpr int B_IT = BN / 32;          // float4 per thread for the B tile (32 x BN)
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MI = WM / 32, NI = WN / 32;

    __shared__ __attribute__((aligned(16))) E lds[(BM + BN) * PITCH];
    E* As = lds;
    E* Bs = lds + BM * PITCH;

    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int s = a.stride;
    const int cls = blockIdx.z / a.ksplit, slice = blockIdx.z % a.ksplit;
    const int py = a.transposed ? cls / s : 0, px = a.transposed ? cls % s : 0;
    // channel range of this k-slice (multiples of 32)
    const int kper = ((a.K + 31) / 32 + a.ksplit - 1) / a.ksplit * 32;
    const int kbeg = slice * kper, kend = min(a.K, kbeg + kper);

    // ---- per-thread A rows: decode once ------------------------------------------------
    const int ac4 = t & 7;                 // which float4 of the 32-channel chunk
    int a_nb[A_IT], a_yb[A_IT], a_xb[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int m = m0 + (t >> 3) + 32 * i;
        a_ok[i] = m < a.Mc;
        int mm = a_ok[i] ? m : 0;
        int n = mm / (a.OHc * a.OWc);
        int rem = mm - n * (a.OHc * a.OWc);
        int yy = rem / a.OWc, xx = rem - yy * a.OWc;
        a_nb[i] = n * a.IH * a.IW;
        a_yb[i] = a.transposed ? yy : yy * s;
        a_xb[i] = a.transposed ? xx : xx * s;
    }

    // ---- K-walk state (uniform) -----------------------------------------------------------
    int ky = 0, kx = -1, kc = kbeg;
    auto tap_ok = [&](int y, int x) -> bool {
        if (!a.transposed) return true;
        return ((py + a.pad - y + s * a.KH) % s) == 0 && ((px + a.pad - x + s * a.KW) % s) == 0;
    };
    auto next_tap = [&]() -> bool {      // advance (ky,kx) to the next valid tap
        for (;;) {
            if (++kx == a.KW) { kx = 0; ++ky; }
            if (ky >= a.KH) return false;
            if (tap_ok(ky, kx)) return true;
        }
    };

    float4 ra[A_IT], rb[B_IT];
    constexpr int BW_IT = BN / 64;          // 16-byte bf16 weight vectors per thread per step
    uint4 rbw[BW_IT];
    const bool usewb = MODE == 1 && a.wb != nullptr;

    auto load_step = [&]() {
        const int dy = a.transposed ? (py + a.pad - ky) / s : ky - a.pad;
        const int dx = a.transposed ? (px + a.pad - kx) / s : kx - a.pad;
        // A: BM rows x 32 channels, float4 along channels
        const int c = kc + ac4 * 4;
        const float* src = a.x; int ld = a.ldx; int cc = c;
        if (c >= a.K1) { src = a.x2; ld = a.ldx2; cc = c - a.K1; }
        const int lim = (c < a.K1 ? a.K1 : a.K) - c;           // valid channels from c on
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int iy = a_yb[i] + dy, ix = a_xb[i] + dx;
            bool ok = a_ok[i] && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
            size_t off = (size_t)(a_nb[i] + iy * a.IW + ix) * ld + cc;
            ra[i] = ld4(src + (ok ? off : 0), ok ? lim : 0, a.vecA);
        }
        // B: 32 k-rows x BN columns of W(tap)
        const int tap = ky * a.KW + kx;
        if (usewb) {
            // pre-converted bf16 [tap][n][k]: 4 x 16-byte vectors per output channel row, no cvt, no transposition
            const int k8 = t & 3;
#pragma unroll
            for (int i = 0; i < BW_IT; ++i) {
                const int n = n0 + (t >> 2) + 64 * i;
                const bool ok = n < a.Nc && kc + k8 * 8 < a.K;
                uint4 v = *reinterpret_cast<const uint4*>(a.wb + ((size_t)tap * a.Nc + (ok ? n : 0)) * a.K + (ok ? kc + k8 * 8 : 0));
                rbw[i] = ok ? v : make_uint4(0, 0, 0, 0);
            }
        } else if (a.w_kn) {
            // memory [tap][k][n]: float4 along n; thread = (k-pair kd, column quad n4).  A half-wave
            // spans 16 k-pairs x 2 quads so the transposing LDS store below is bank-conflict free
            // (pitch 20 dwords: bank = 16*(n4&1) + 20*j + kd mod 32).
            const int kd = t & 15;
#pragma unroll
            for (int p = 0; p < B_IT / 2; ++p) {
                int n4 = (t >> 4) + 16 * p;
                int k = kc + 2 * kd;
                int n = n0 + n4 * 4;
                const float* wp = a.w + ((size_t)tap * a.K + k) * a.Nc + n;
                rb[2 * p]     = ld4(wp,        (k     < a.K) ? a.Nc - n : 0, a.vecB);
                rb[2 * p + 1] = ld4(wp + a.Nc, (k + 1 < a.K) ? a.Nc - n : 0, a.vecB);
            }
        } else {
            // memory [tap][n][k]: float4 along k
            const int k4 = t & 7;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                int n = n0 + (t >> 3) + 32 * i;
                int k = kc + k4 * 4;
                const float* wp = a.w + ((size_t)tap * a.Nc + n) * a.K + k;
                rb[i] = ld4(n < a.Nc ? wp : a.w, n < a.Nc ? a.K - k : 0, a.vecB);
            }
        }
    };

    auto store_step = [&]() {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int r = (t >> 3) + 32 * i;
            if constexpr (MODE == 1) {
                uint2 v = make_uint2(pack_bf16(ra[i].x, ra[i].y), pack_bf16(ra[i].z, ra[i].w));
                *reinterpret_cast<uint2*>(&As[r * PITCH + ac4 * 4]) = v;
            } else {
                *reinterpret_cast<float4*>(&As[r * PITCH + ac4 * 4]) = ra[i];
            }
        }
        if (usewb) {
            if constexpr (MODE == 1) {
                const int k8 = t & 3;
#pragma unroll
                for (int i = 0; i < BW_IT; ++i)
                    *reinterpret_cast<uint4*>(&Bs[((t >> 2) + 64 * i) * PITCH + k8 * 8]) = rbw[i];
            }
        } else if (a.w_kn) {
            const int kd = t & 15;
#pragma unroll
            for (int p = 0; p < B_IT / 2; ++p) {
                int n4 = (t >> 4) + 16 * p;
                int k = 2 * kd;
                const float lo[4] = {rb[2 * p].x, rb[2 * p].y, rb[2 * p].z, rb[2 * p].w};
                const float hi[4] = {rb[2 * p + 1].x, rb[2 * p + 1].y, rb[2 * p + 1].z, rb[2 * p + 1].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int n = n4 * 4 + j;
                    if constexpr (MODE == 1)
                        *reinterpret_cast<uint32_t*>(&Bs[n * PITCH + k]) = pack_bf16(lo[j], hi[j]);
                    else
                        *reinterpret_cast<float2*>(&Bs[n * PITCH + k]) = make_float2(lo[j], hi[j]);
                }
            }
        } else {
            const int k4 = t & 7;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                int n = (t >> 3) + 32 * i;
                if constexpr (MODE == 1) {
                    uint2 v = make_uint2(pack_bf16(rb[i].x, rb[i].y), pack_bf16(rb[i].z, rb[i].w));
                    *reinterpret_cast<uint2*>(&Bs[n * PITCH + k4 * 4]) = v;
                } else {
                    *reinterpret_cast<float4*>(&Bs[n * PITCH + k4 * 4]) = rb[i];
                }
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bool more = kbeg < kend && next_tap();
    if (more) { load_step(); store_step(); }
    __syncthreads();

    while (more) {
        // advance to the next (tap, channel chunk) and prefetch it into registers
        kc += 32;
        if (kc >= kend) { kc = kbeg; more = next_tap(); }
        if (more) load_step();

        const int arow = wm * WM + (l & 31), brow = wn * WN + (l & 31), kh = l >> 5;
        if constexpr (MODE == 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[MI], bf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(&As[(arow + 32 * i) * PITCH + ks * 16 + kh * 8]);
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8*>(&Bs[(brow + 32 * j) * PITCH + ks * 16 + kh * 8]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                float af[MI], bf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = As[(arow + 32 * i) * PITCH + 2 * kk + kh];
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[j] = Bs[(brow + 32 * j) * PITCH + 2 * kk + kh];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) { store_step(); __syncthreads(); }
    }

    // ---- epilogue: bias, residual, accumulate, coalesced 128-B row segments -------------
    const bool remap = a.transposed && s > 1;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            int m = m0 + row;
            if (m >= a.Mc) continue;
            size_t opix = m;
            if (remap) {
                int n = m / (a.OHc * a.OWc);
                int rem = m - n * (a.OHc * a.OWc);
                int yy = rem / a.OWc, xx = rem - yy * a.OWc;
                opix = (size_t)(n * a.OH + yy * s + py) * a.OW + xx * s + px;
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int col = n0 + wn * WN + j * 32 + (l & 31);
                if (col >= a.Nc) continue;
                float v = acc[i][j][r];
                float* yp = a.y + opix * a.ldy + col;
                if (a.ksplit > 1) {                     // output was zeroed (or holds the value to add to)
                    if (slice == 0) { if (a.bias) v += a.bias[col]; if (a.res) v += a.res[opix * a.ldr + col]; }
                    atomicAdd(yp, v);
                    continue;
                }
                if (a.bias) v += a.bias[col];
                if (a.res) v += a.res[opix * a.ldr + col];
                if (a.accumulate) v += *yp;
                *yp = v;
            }
        }
    }
}

template <int MODE, int BM, int BN>
int launch(const IgemmArgs& a, int classes, hipStream_t st) {
    dim3 grid((a.Mc + BM - 1) / BM, (a.Nc + BN - 1) / BN, classes * a.ksplit);
    hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN>), grid, dim3(256), 0, st, a);
    return 0;
}

}  // namespace

static int igemm_dispatch(const MiConvDesc* d, const float* x, const float* x2, const float* w, const uint16_t* wb,
                          const float* bias, const float* residual, float* y, void* stream);

extern "C" int mi_conv_igemm(const MiConvDesc* d, const float* x, const float* x2, const float* w,
                             const float* bias, const float* residual, float* y, void* stream) {
    return igemm_dispatch(d, x, x2, w, nullptr, bias, residual, y, stream);
}

// Same contract, weights given as the bf16 copy laid out [tap][Nc][K] (mi_pack_weights_bf16): used for
// the stride-2 / transposed convolutions the tile kernel does not cover.  Needs mode == 1, K % 8 == 0.
extern "C" int mi_conv_igemm_bf16w(const MiConvDesc* d, const float* x, const float* x2, const void* w_nk_bf16,
                                   const float* bias, const float* residual, float* y, void* stream) {
    if (!d || d->mode != 1 || d->K % 8 || !w_nk_bf16 || ((uintptr_t)w_nk_bf16 & 15))
        return mi_set_error(-1, "mi_conv_igemm_bf16w: needs mode 1, K %% 8 == 0 and 16-byte aligned bf16 weights");
    return igemm_dispatch(d, x, x2, (const float*)w_nk_bf16, (const uint16_t*)w_nk_bf16, bias, residual, y, stream);
}

static int igemm_dispatch(const MiConvDesc* d, const float* x, const float* x2, const float* w, const uint16_t* wb,
                          const float* bias, const float* residual, float* y, void* stream) {
    MI_REQUIRE(d && x && w && y, "null argument");
    MI_REQUIRE(d->N > 0 && d->K > 0 && d->Nc > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0, "bad sizes");
    MI_REQUIRE(d->mode == 0 || d->mode == 1, "mode must be 0 (fp32) or 1 (bf16)");
    MI_REQUIRE(d->K1 == d->K || (x2 && d->K1 > 0 && d->K1 < d->K && d->K1 % 4 == 0), "bad two-source split");
    MI_REQUIRE(d->ldx % 4 == 0 && d->ldy >= d->Nc, "ldx must be a multiple of 4, ldy >= Nc");
    IgemmArgs a;
    a.x = x; a.x2 = x2 ? x2 : x; a.w = w; a.wb = wb; a.bias = bias; a.res = residual; a.y = y;
    a.N = d->N; a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW; a.K = d->K; a.Nc = d->Nc;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.transposed = d->transposed;
    a.w_kn = d->w_kn; a.K1 = d->K1; a.ldx = d->ldx; a.ldx2 = x2 ? d->ldx2 : d->ldx; a.ldy = d->ldy;
    a.ldr = d->ldr; a.accumulate = d->accumulate;
    int classes = 1;
    a.OHc = d->OH; a.OWc = d->OW;
    if (d->transposed && d->stride > 1) {
        MI_REQUIRE(d->OH % d->stride == 0 && d->OW % d->stride == 0, "transposed: OH, OW must be multiples of stride");
        MI_REQUIRE(d->KH >= d->stride && d->KW >= d->stride, "transposed: kernel smaller than stride");
        classes = d->stride * d->stride;
        a.OHc = d->OH / d->stride; a.OWc = d->OW / d->stride;
    }
    a.Mc = d->N * a.OHc * a.OWc;
    // activations: rows are 16-B aligned when ld % 4 == 0; a chunk may still be ragged at the end
    a.vecA = (d->ldx % 4 == 0) && (a.ldx2 % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)a.x2 & 15) == 0);
    a.vecB = (((uintptr_t)w & 15) == 0) && (d->w_kn ? (d->Nc % 4 == 0) : (d->K % 4 == 0));
    if (residual) MI_REQUIRE(d->ldr >= d->Nc, "ldr < Nc");
    hipStream_t st = (hipStream_t)stream;
    a.ksplit = 1;
    {   // long contraction, almost no tiles (Linear dgrad of the fused time-bias GEMM): split the channel axis
        long tiles64 = (long)((a.Mc + 63) / 64) * ((d->Nc + 63) / 64) * classes;
        if (tiles64 <= 16 && d->K >= 1024 && d->KH * d->KW == 1 && (d->accumulate || d->ldy == d->Nc)) {
            int ks = d->K / 128; if (ks > 32) ks = 32;
            a.ksplit = ks;
            if (!d->accumulate) {
                hipError_t e = hipMemsetAsync(y, 0, (size_t)a.Mc * classes * d->ldy * sizeof(float), st);
                if (e != hipSuccess) return mi_set_error((int)e, "mi_conv_igemm: memset: %s", hipGetErrorString(e));
            }
        }
    }

    // tile choice: keep >= ~2 workgroups per CU when the problem allows it
    long tiles128 = (long)((a.Mc + 127) / 128) * ((d->Nc + 127) / 128) * classes;
    bool bn64 = d->Nc <= 64;
    bool bm64 = tiles128 < 384 || a.Mc <= 64;
    if (bn64 == false && bm64 && (long)((a.Mc + 63) / 64) * ((d->Nc + 127) / 128) * classes < 384) bn64 = true;
#define MI_GO(MODE) \
    do { if (!bm64 && !bn64) launch<MODE, 128, 128>(a, classes, st); \
         else if (!bm64 && bn64) launch<MODE, 128, 64>(a, classes, st); \
         else if (bm64 && !bn64) launch<MODE, 64, 128>(a, classes, st); \
         else launch<MODE, 64, 64>(a, classes, st); } while (0)
    if (d->mode == 1) MI_GO(1); else MI_GO(0);
#undef MI_GO
    MI_LAUNCH_CHECK();
    return 0;
}

// Which tile instantiation mi_conv_igemm picks for a descriptor (profiling attribution only).
extern "C" int mi_conv_igemm_tile(const MiConvDesc* d, int* bm, int* bn) {
    MI_REQUIRE(d && bm && bn, "null argument");
    int classes = 1, OHc = d->OH, OWc = d->OW;
    if (d->transposed && d->stride > 1) { classes = d->stride * d->stride; OHc /= d->stride; OWc /= d->stride; }
    int Mc = d->N * OHc * OWc;
    long tiles128 = (long)((Mc + 127) / 128) * ((d->Nc + 127) / 128) * classes;
    bool bn64 = d->Nc <= 64;
    bool bm64 = tiles128 < 384 || Mc <= 64;
    if (bn64 == false && bm64 && (long)((Mc + 63) / 64) * ((d->Nc + 127) / 128) * classes < 384) bn64 = true;
    *bm = bm64 ? 64 : 128; *bn = bn64 ? 64 : 128;
    return 0;
}
