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
#include <stdlib.h>
#include "common.h"

#ifdef MI_HALO_TIMING
// profiling build only (make EXTRA=-DMI_HALO_TIMING): per-workgroup phase timestamps of igemm_fast_kernel
__device__ unsigned long long g_igemm_ts[4 * 4096];
#define MI_TSI(k) do { if (threadIdx.x == 0 && blockIdx.y == 0) { const unsigned b_ = blockIdx.z * gridDim.x + blockIdx.x; if (b_ < 4096) g_igemm_ts[(k) * 4096 + b_] = wall_clock64(); } } while (0)
extern "C" int mi_debug_igemm_ts(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_igemm_ts), sizeof(g_igemm_ts));
}
#else
#define MI_TSI(k) do {} while (0)
#endif

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
    MI_PRIO_UP();
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
                if (yy * s + py >= a.OH || xx * s + px >= a.OW) continue;        // partial last cell of a ragged output grid
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

// ---------------------------------------------------------------------------------------------
// Fast path of the same implicit GEMM for the aligned bf16 cases that matter for throughput: the
// stride-2 Downsample conv, the ConvTranspose2d Upsample and their data gradients (ddpm.py:70,79).
// Requirements: bf16 weight copy [tap][Nc][K], K % 32 == 0, 16-byte aligned activation rows.
// Differences from igemm_kernel: the valid taps of a parity class come from a table in the
// arguments (no tap search in the loop); row validity per tap is a bit mask computed once; the
// K walk is straight-line code over a ring of three register stages (stage s+3 is requested
// while stage s runs on the matrix cores and stage s+1 is written to the other LDS buffer) with
// clamped addresses, padding applied as an AND mask at the LDS store and the tail handled by
// zeroed weight tiles -- so every s_waitcnt is an exact vmcnt(N) and there is one barrier per
// 32-channel step instead of two.
struct FastTaps {                   // dwords, so that the uniform lookups in the loop are scalar loads
    int dy[64], dx[64];             // [class * 16 + i]: input offset of the i-th valid tap of the class
    int dpix[64];                   // dy * IW + dx
    int tap[64];                    // its index ky * KW + kx into the weights
    int ntap[4];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// IN16: the activations are a bf16 tensor (a copy the caller already has: the skip connection's, the one the weight gradient reads).
// A 16-byte load is then 8 channels, so the same loads per thread cover a 64-channel stage: half the barriers per MFMA, half the
// bytes, no pack on the way into LDS.
template <int BM, int BN, bool IN16>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128 ? 2 : 3)) void igemm_fast_kernel(const IgemmArgs a, const FastTaps tt) {
    MI_PRIO_UP();
    constexpr int CK = IN16 ? 64 : 32;     // channels per stage
    constexpr int KS = CK / 16;            // MFMA k-steps per stage
    constexpr int PITCH = CK + 8;
    constexpr int A_IT = BM / 32;          // 16-byte activation loads per thread per stage (BM rows x CK channels)
    constexpr int BROWS = 256 / (CK / 8);  // weight rows covered by one 16-byte load per thread
    constexpr int B_IT = BN / BROWS;       // 16-byte weight loads per thread per stage (BN rows x CK k, bf16)
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int ABUF = (BM + BN) * PITCH;

    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];         // 2 * ABUF

    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int s = a.stride, cls = blockIdx.z;
    const int py = a.transposed ? cls / s : 0, px = a.transposed ? cls % s : 0;
    const int ntap = tt.ntap[cls];
    const int nchunks = a.K / CK;
    const int nsteps = ntap * nchunks;

    MI_TSI(0);
    // ---- per-thread A rows: base pixel and a validity bit per tap, decoded once
    const int ac4 = t & 7;
    int apix[A_IT]; uint32_t amask[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + (t >> 3) + 32 * i;
        const bool ok = m < a.Mc;
        const int mm = ok ? m : 0;
        const int n = mm / (a.OHc * a.OWc);
        const int rem = mm - n * (a.OHc * a.OWc);
        const int yy = rem / a.OWc, xx = rem - yy * a.OWc;
        const int yb = a.transposed ? yy : yy * s, xb = a.transposed ? xx : xx * s;
        apix[i] = (n * a.IH + yb) * a.IW + xb;
        uint32_t mk = 0;
        for (int ti = 0; ti < ntap; ++ti) {
            const int iy = yb + tt.dy[cls * 16 + ti], ix = xb + tt.dx[cls * 16 + ti];
            if (ok && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) mk |= 1u << ti;
        }
        amask[i] = mk;
    }
    const int b_row = t / (CK / 8), b_k8 = t % (CK / 8);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[3][A_IT]; u32x4 rb[3][B_IT]; uint32_t rk[3];      // ra: 4 fp32 or (IN16) 8 bf16 channels
    int lti = 0, lkc = 0;                   // load cursor: (tap index in the class, channel offset) of the next stage

    auto load_stage = [&](u32x4 (&A)[A_IT], u32x4 (&B)[B_IT], uint32_t& keep) {
        const int dpix = tt.dpix[cls * 16 + lti];
        const bool second = lkc >= a.K1;
        const int coff = (second ? lkc - a.K1 : lkc) + ac4 * (IN16 ? 8 : 4);
        const int ld = second ? a.ldx2 : a.ldx;
        uint32_t kp = 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const uint32_t ok = (amask[i] >> lti) & 1u;
            kp |= ok << i;
            const int pix = ok ? apix[i] + dpix : 0;
            if constexpr (IN16) A[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(second ? a.x2 : a.x) + (size_t)pix * ld + coff);
            else A[i] = *reinterpret_cast<const u32x4*>((second ? a.x2 : a.x) + (size_t)pix * ld + coff);
        }
        keep = kp;
        const uint16_t* wsrc = a.wb + (size_t)tt.tap[cls * 16 + lti] * a.Nc * a.K + lkc + b_k8 * 8;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = min(n0 + b_row + BROWS * i, a.Nc - 1);       // rows past Nc: their columns are never written
            B[i] = *reinterpret_cast<const u32x4*>(wsrc + (size_t)n * a.K);
        }
        // advance the cursor; past the end it stays on the last stage (a harmless re-read)
        lkc += CK;
        if (lkc >= a.K) { lkc = 0; ++lti; }
        if (lti >= ntap) { lti = ntap - 1; lkc = a.K - CK; }
    };
    auto store_stage = [&](int buf, const u32x4 (&A)[A_IT], const u32x4 (&B)[B_IT], uint32_t keep, uint32_t live) {
        uint16_t* As = lds + buf * ABUF;
        uint16_t* Bs = As + BM * PITCH;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const uint32_t m = 0u - ((keep >> i) & 1u);
            if constexpr (IN16) {
                *reinterpret_cast<u32x4*>(&As[((t >> 3) + 32 * i) * PITCH + ac4 * 8]) = A[i] & m;
            } else {
                const f32x4 v = __builtin_bit_cast(f32x4, A[i]);
                *reinterpret_cast<u32x2*>(&As[((t >> 3) + 32 * i) * PITCH + ac4 * 4]) = u32x2{pack_bf16(v.x, v.y) & m, pack_bf16(v.z, v.w) & m};
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            *reinterpret_cast<u32x4*>(&Bs[(b_row + BROWS * i) * PITCH + b_k8 * 8]) = B[i] & live;
    };
    const int arow = (wm * WM + (l & 31)) * PITCH + (l >> 5) * 8, brow = (wn * WN + (l & 31)) * PITCH + (l >> 5) * 8;
    // one step = fragment reads of the current buffer, then the LDS stores of the next stage (they drain under the
    // MFMAs), then the MFMAs
    bf16x8 af[KS][MI], bf[KS][NI];
    auto read_frags = [&](int buf) {
        const uint16_t* As = lds + buf * ABUF;
        const uint16_t* Bs = As + BM * PITCH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int i = 0; i < MI; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(&As[arow + 32 * i * PITCH + ks * 16]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[ks][j] = *reinterpret_cast<const bf16x8*>(&Bs[brow + 32 * j * PITCH + ks * 16]);
        }
    };
    auto mma_frags = [&]() {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
    };

    // prologue: stage 0 to LDS buffer 0, stages 1 and 2 into the ring
    load_stage(ra[0], rb[0], rk[0]);
    load_stage(ra[1], rb[1], rk[1]);
    load_stage(ra[2], rb[2], rk[2]);
    store_stage(0, ra[0], rb[0], rk[0], ~0u);
    __syncthreads();
    MI_TSI(1);
    for (int s0 = 0; s0 < nsteps; s0 += 3) {
        // step s0 (slot 0 is free: stage s0 already sits in LDS)
        read_frags(s0 & 1);
        store_stage((s0 + 1) & 1, ra[1], rb[1], rk[1], s0 + 1 < nsteps ? ~0u : 0u);
        load_stage(ra[0], rb[0], rk[0]);                                   // stage s0 + 3
        mma_frags();
        __syncthreads();
        read_frags((s0 + 1) & 1);
        store_stage(s0 & 1, ra[2], rb[2], rk[2], s0 + 2 < nsteps ? ~0u : 0u);
        load_stage(ra[1], rb[1], rk[1]);                                   // stage s0 + 4
        mma_frags();
        __syncthreads();
        read_frags(s0 & 1);
        store_stage((s0 + 1) & 1, ra[0], rb[0], rk[0], s0 + 3 < nsteps ? ~0u : 0u);
        load_stage(ra[2], rb[2], rk[2]);                                   // stage s0 + 5
        mma_frags();
        __syncthreads();
    }

    MI_TSI(2);
    // ---- epilogue: bias, residual, accumulate; 128-byte row segments.  The optional operands are fetched in
    //      batches (one wait per 32-row block) instead of one dependent load per element.
    const bool remap = a.transposed && s > 1;
    float bcol[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = min(n0 + wn * WN + j * 32 + (l & 31), a.Nc - 1);
        bcol[j] = a.bias ? a.bias[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int opix[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), a.Mc - 1);
            opix[r] = m;
            if (remap) {
                const int n = m / (a.OHc * a.OWc);
                const int rem = m - n * (a.OHc * a.OWc);
                const int yy = rem / a.OWc, xx = rem - yy * a.OWc;
                opix[r] = (n * a.OH + yy * s + py) * a.OW + xx * s + px;
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * WN + j * 32 + (l & 31);
            const int colc = min(col, a.Nc - 1);
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bcol[j];
            if (a.res) {
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = a.res[(size_t)opix[r] * a.ldr + colc];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += rv[r];
            }
            if (a.accumulate) {
                float ov[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) ov[r] = a.y[(size_t)opix[r] * a.ldy + colc];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += ov[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (m < a.Mc && col < a.Nc) a.y[(size_t)opix[r] * a.ldy + col] = v[r];
            }
        }
    }
    MI_TSI(3);
}

template <int BM, int BN, bool IN16>
int launch_fast_t(const IgemmArgs& a, const FastTaps& tt, int classes, hipStream_t st) {
    dim3 grid((a.Mc + BM - 1) / BM, (a.Nc + BN - 1) / BN, classes);
    constexpr size_t lds = 2 * (size_t)(BM + BN) * ((IN16 ? 64 : 32) + 8) * sizeof(uint16_t);
    static MiPerDevice once;
    once.run([] {
        (void)hipFuncSetAttribute((const void*)igemm_fast_kernel<BM, BN, IN16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL((igemm_fast_kernel<BM, BN, IN16>), grid, dim3(256), lds, st, a, tt);
    return 0;
}
template <int BM, int BN>
int launch_fast(const IgemmArgs& a, const FastTaps& tt, int classes, bool in16, hipStream_t st) {
    return in16 ? launch_fast_t<BM, BN, true>(a, tt, classes, st) : launch_fast_t<BM, BN, false>(a, tt, classes, st);
}

}  // namespace

static int igemm_dispatch(const MiConvDesc* d, const float* x, const float* x2, const float* w, const uint16_t* wb,
                          const float* bias, const float* residual, float* y, void* stream, bool in16 = false);

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

// ... and bf16-stored activations (x_is_bf16: x / x2 are bf16 tensors, ldx / ldx2 in elements, % 8 == 0, K and K1 % 64 == 0): only
// the shapes the ring kernel takes (the stride-2 conv, the transposed conv and their data gradients); query first.
extern "C" int mi_conv_igemm_bf16w_io_supported(const MiConvDesc* d) {
    if (!d || d->mode != 1 || d->K % 64 || d->K1 % 64 || d->ldx % 8 || (d->K1 != d->K && d->ldx2 % 8)) return 0;
    if (d->KH * d->KW > 16 || d->KH * d->KW == 1) return 0;
    if (d->transposed && d->stride > 1 && (d->OH % d->stride || d->OW % d->stride || d->stride != 2)) return 0;
    return 1;
}
extern "C" int mi_conv_igemm_bf16w_io(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16,
                                      const float* bias, const float* residual, float* y, int x_is_bf16, void* stream) {
    if (!x_is_bf16) return mi_conv_igemm_bf16w(d, (const float*)x, (const float*)x2, w_nk_bf16, bias, residual, y, stream);
    if (!mi_conv_igemm_bf16w_io_supported(d) || !w_nk_bf16 || (((uintptr_t)w_nk_bf16 | (uintptr_t)x | (uintptr_t)(x2 ? x2 : x)) & 15))
        return mi_set_error(-1, "mi_conv_igemm_bf16w_io: bf16 activations need mode 1, K / K1 %% 64 == 0, ldx %% 8 == 0, 16-byte aligned "
                                "operands and a stride-2 / multi-tap layer (mi_conv_igemm_bf16w_io_supported)");
    return igemm_dispatch(d, (const float*)x, (const float*)x2, (const float*)w_nk_bf16, (const uint16_t*)w_nk_bf16, bias, residual, y, stream, true);
}

static int igemm_dispatch(const MiConvDesc* d, const float* x, const float* x2, const float* w, const uint16_t* wb,
                          const float* bias, const float* residual, float* y, void* stream, bool in16) {
    MI_REQUIRE(d && x && w && y, "null argument");
    MI_REQUIRE(d->N > 0 && d->K > 0 && d->Nc > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0, "bad sizes");
    MI_REQUIRE(d->mode == 0 || d->mode == 1, "mode must be 0 (fp32) or 1 (bf16)");
    MI_REQUIRE(d->K1 == d->K || (x2 && d->K1 > 0 && d->K1 < d->K && d->K1 % 4 == 0), "bad two-source split");
    MI_REQUIRE(d->ldx % 4 == 0 && d->ldy >= d->Nc, "ldx must be a multiple of 4, ldy >= Nc");
    MI_REQUIRE(!in16 || d->K1 == d->K || d->K1 % 64 == 0, "bf16 activations: K1 % 64 == 0");
    IgemmArgs a;
    a.x = x; a.x2 = x2 ? x2 : x; a.w = w; a.wb = wb; a.bias = bias; a.res = residual; a.y = y;
    a.N = d->N; a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW; a.K = d->K; a.Nc = d->Nc;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.transposed = d->transposed;
    a.w_kn = d->w_kn; a.K1 = d->K1; a.ldx = d->ldx; a.ldx2 = x2 ? d->ldx2 : d->ldx; a.ldy = d->ldy;
    a.ldr = d->ldr; a.accumulate = d->accumulate;
    int classes = 1;
    a.OHc = d->OH; a.OWc = d->OW;
    bool ragged = false;                      // output extent not a multiple of the stride (7x7 from 4x4, stride 2): rows of the
    if (d->transposed && d->stride > 1) {     // last partial class cell are computed and dropped at the store
        MI_REQUIRE(d->KH >= d->stride && d->KW >= d->stride, "transposed: kernel smaller than stride");
        classes = d->stride * d->stride;
        a.OHc = (d->OH + d->stride - 1) / d->stride; a.OWc = (d->OW + d->stride - 1) / d->stride;
        ragged = d->OH % d->stride != 0 || d->OW % d->stride != 0;
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
                hipError_t e = mi_zero_async(y, (size_t)d->N * d->OH * d->OW * d->ldy * sizeof(float), st);
                if (e != hipSuccess) return mi_set_error((int)e, "mi_conv_igemm: memset: %s", hipGetErrorString(e));
            }
        }
    }

    // tile choice: keep >= ~2 workgroups per CU when the problem allows it
    long tiles128 = (long)((a.Mc + 127) / 128) * ((d->Nc + 127) / 128) * classes;
    bool bn64 = d->Nc <= 64;
    bool bm64 = tiles128 < 384 || a.Mc <= 64;
    if (bn64 == false && bm64 && (long)((a.Mc + 63) / 64) * ((d->Nc + 127) / 128) * classes < 384) bn64 = true;
    // aligned bf16 layers with few taps per class: the straight-line ring kernel
    static const int allow_fast = (int)mi_knob("MI_IGEMM_FAST", 1);
    // (bf16-stored activations exist only on the ring kernel: the profiling switch does not apply to them, so that
    //  mi_conv_igemm_bf16w_io_supported() and this dispatch agree)
    if ((allow_fast || in16) && !ragged && d->mode == 1 && wb && a.ksplit == 1 && d->K % 32 == 0 && d->K1 % 32 == 0 && a.vecA && classes <= 4 &&
        d->KH * d->KW <= 16) {
        FastTaps tt;
        bool ok = true;
        for (int c = 0; c < classes && ok; ++c) {
            const int py = d->transposed ? c / d->stride : 0, px = d->transposed ? c % d->stride : 0;
            int nt = 0;
            for (int ky = 0; ky < d->KH; ++ky)
                for (int kx = 0; kx < d->KW; ++kx) {
                    int dy, dx;
                    if (d->transposed) {
                        const int ny = py + d->pad - ky, nx = px + d->pad - kx;
                        if (((ny % d->stride) + d->stride) % d->stride || ((nx % d->stride) + d->stride) % d->stride) continue;
                        dy = ny / d->stride; dx = nx / d->stride;
                    } else { dy = ky - d->pad; dx = kx - d->pad; }
                    if (nt >= 16) { ok = false; break; }
                    tt.dy[c * 16 + nt] = dy; tt.dx[c * 16 + nt] = dx; tt.dpix[c * 16 + nt] = dy * d->IW + dx;
                    tt.tap[c * 16 + nt] = ky * d->KW + kx;
                    ++nt;
                }
            tt.ntap[c] = nt;
            if (nt == 0) ok = false;
        }
        if (ok) {
            static const int force = (int)mi_knob("MI_IGEMM_TILE", 0);   // 11 / 10 / 01 / 00 = bm64,bn64 (profiling)
            if (force) { bm64 = (force / 10) % 10 == 1; bn64 = force % 10 == 1; if (force == 100) { bm64 = false; bn64 = false; } }
            if (!bm64 && !bn64) launch_fast<128, 128>(a, tt, classes, in16, st);
            else if (!bm64 && bn64) launch_fast<128, 64>(a, tt, classes, in16, st);
            else if (bm64 && !bn64) launch_fast<64, 128>(a, tt, classes, in16, st);
            else launch_fast<64, 64>(a, tt, classes, in16, st);
            MI_LAUNCH_CHECK();
            return 0;
        }
    }
    MI_REQUIRE(!in16, "bf16 activations: the layer does not fit the ring kernel (mi_conv_igemm_bf16w_io_supported)");
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
    if (d->transposed && d->stride > 1) { classes = d->stride * d->stride; OHc = (OHc + d->stride - 1) / d->stride; OWc = (OWc + d->stride - 1) / d->stride; }
    int Mc = d->N * OHc * OWc;
    long tiles128 = (long)((Mc + 127) / 128) * ((d->Nc + 127) / 128) * classes;
    bool bn64 = d->Nc <= 64;
    bool bm64 = tiles128 < 384 || Mc <= 64;
    if (bn64 == false && bm64 && (long)((Mc + 63) / 64) * ((d->Nc + 127) / 128) * classes < 384) bn64 = true;
    *bm = bm64 ? 64 : 128; *bn = bn64 ? 64 : 128;
    return 0;
}
