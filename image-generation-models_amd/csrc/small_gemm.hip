// Exact-fp32 GEMM for the time-embedding MLP (ddpm.py:126-130,186-193): nn.Linear forward, its input gradient
// and its weight gradient with M = batch rows (128) -- a few hundred MFLOP spread over 8 launches per step.
// The generic implicit-GEMM kernel spends 20-30 us on each (two barriers and one exposed memory round trip
// per 32-channel step, a handful of workgroups); here one 256-thread workgroup owns a 32x32 output tile
// (2x2 per thread, fp32 FMA), the next K chunk's operands are in flight while the current one is
// multiplied, and (when the caller allows it: backward only, the forward stays bit-reproducible) long contractions
// are split over blockIdx.z and combined with fp32 atomics.
//   C[i][j] (+)= bias[j] + sum_k opA(i, k) * opB(k, j)
//   opA(i, k) = ta ? A[k*lda + i] : A[i*lda + k]        opB(k, j) = tb ? B[j*ldb + k] : B[k*ldb + j]
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SgArgs {
    const float* A; const float* B; const float* bias; float* C;
    int I, J, K, lda, ldb, ldc, accumulate, ksplit, cps;     // cps = 32-wide K chunks per z slice
};

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void small_gemm_kernel(const SgArgs a) {
    constexpr int P = 36;                                 // LDS pitch (floats): 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) float As[2][32 * P], Bs[2][32 * P];      // [k][i], [k][j]
    const int t = threadIdx.x;
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int nchunks = a.K / 32;
    const int c0 = blockIdx.z * a.cps, c1 = min(nchunks, c0 + a.cps);
    if (c0 >= c1) return;
    const int r = t >> 3, q = t & 7;                      // staging: row r (0..31), float4 q (0..7)

    // one float4 of each operand tile per thread; rows / columns past the extent are clamped (masked at the store to C)
    auto load_a = [&](int c) -> f32x4 {
        if constexpr (TA) return *reinterpret_cast<const f32x4*>(a.A + (size_t)(c * 32 + r) * a.lda + min(i0 + 4 * q, a.I - 4));
        else return *reinterpret_cast<const f32x4*>(a.A + (size_t)min(i0 + r, a.I - 1) * a.lda + c * 32 + 4 * q);
    };
    auto load_b = [&](int c) -> f32x4 {
        if constexpr (TB) return *reinterpret_cast<const f32x4*>(a.B + (size_t)min(j0 + r, a.J - 1) * a.ldb + c * 32 + 4 * q);
        else return *reinterpret_cast<const f32x4*>(a.B + (size_t)(c * 32 + r) * a.ldb + min(j0 + 4 * q, a.J - 4));
    };
    auto store_a = [&](int buf, f32x4 v) {
        if constexpr (TA) *reinterpret_cast<f32x4*>(&As[buf][r * P + 4 * q]) = v;                     // row = k, cols = i
        else { As[buf][(4 * q + 0) * P + r] = v.x; As[buf][(4 * q + 1) * P + r] = v.y; As[buf][(4 * q + 2) * P + r] = v.z; As[buf][(4 * q + 3) * P + r] = v.w; }
    };
    auto store_b = [&](int buf, f32x4 v) {
        if constexpr (TB) { Bs[buf][(4 * q + 0) * P + r] = v.x; Bs[buf][(4 * q + 1) * P + r] = v.y; Bs[buf][(4 * q + 2) * P + r] = v.z; Bs[buf][(4 * q + 3) * P + r] = v.w; }
        else *reinterpret_cast<f32x4*>(&Bs[buf][r * P + 4 * q]) = v;
    };

    const int ti = t >> 4, tj = t & 15;                   // outputs (2ti, 2ti+1) x (2tj, 2tj+1)
    float acc00 = 0.f, acc01 = 0.f, acc10 = 0.f, acc11 = 0.f;
    f32x4 ra = load_a(c0), rb = load_b(c0);
    store_a(0, ra); store_b(0, rb);
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
        const int buf = (c - c0) & 1;
        const int cn = min(c + 1, c1 - 1);                // past the end: a harmless re-read
        ra = load_a(cn); rb = load_b(cn);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float2 av = *reinterpret_cast<const float2*>(&As[buf][k * P + 2 * ti]);
            const float2 bv = *reinterpret_cast<const float2*>(&Bs[buf][k * P + 2 * tj]);
            acc00 += av.x * bv.x; acc01 += av.x * bv.y; acc10 += av.y * bv.x; acc11 += av.y * bv.y;
        }
        store_a(buf ^ 1, ra); store_b(buf ^ 1, rb);
        __syncthreads();
    }
    const float accs[2][2] = {{acc00, acc01}, {acc10, acc11}};
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const int i = i0 + 2 * ti + di, j = j0 + 2 * tj + dj;
            if (i >= a.I || j >= a.J) continue;
            float v = accs[di][dj];
            float* cp = a.C + (size_t)i * a.ldc + j;
            if (blockIdx.z == 0 && a.bias) v += a.bias[j];
            if (a.ksplit > 1) atomicAdd(cp, v);
            else *cp = a.accumulate ? *cp + v : v;
        }
}

}  // namespace

extern "C" int mi_small_gemm_supported(int ta, int tb, int I, int J, int K, int lda, int ldb) {
    if (I <= 0 || J <= 0 || K <= 0 || K % 32 || lda % 4 || ldb % 4) return 0;
    if (ta && (I % 4 || I < 4)) return 0;
    if (!tb && (J % 4 || J < 4)) return 0;
    return 1;
}

extern "C" int mi_small_gemm(int ta, int tb, int I, int J, int K, const float* A, int lda, const float* B, int ldb,
                             const float* bias, float* C, int ldc, int accumulate, int allow_split, void* stream) {
    MI_REQUIRE(A && B && C && mi_small_gemm_supported(ta, tb, I, J, K, lda, ldb) && (((uintptr_t)A | (uintptr_t)B) & 15) == 0,
               "needs K % 32 == 0, 16-byte aligned operands with ld % 4 == 0 (and I % 4 / J % 4 for the transposed forms)");
    SgArgs a;
    a.A = A; a.B = B; a.bias = bias; a.C = C; a.I = I; a.J = J; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.accumulate = accumulate;
    const int nchunks = K / 32;
    const long tiles = (long)((I + 31) / 32) * ((J + 31) / 32);
    int ks = 1;
    if (allow_split && (ldc == J || accumulate)) {         // split the contraction when there are few tiles (atomics need a dense / pre-filled C)
        while (tiles * ks < 256 && ks * 2 <= nchunks && ks < 32) ks *= 2;
    }
    a.cps = (nchunks + ks - 1) / ks;
    a.ksplit = (nchunks + a.cps - 1) / a.cps;
    hipStream_t st = (hipStream_t)stream;
    if (a.ksplit > 1 && !accumulate) {
        hipError_t e = mi_zero_async(C, (size_t)I * ldc * sizeof(float), st);
        if (e != hipSuccess) return mi_set_error((int)e, "mi_small_gemm: memset: %s", hipGetErrorString(e));
    }
    dim3 grid((I + 31) / 32, (J + 31) / 32, a.ksplit);
    if (ta) { if (tb) hipLaunchKernelGGL((small_gemm_kernel<true, true>), grid, dim3(256), 0, st, a);
              else    hipLaunchKernelGGL((small_gemm_kernel<true, false>), grid, dim3(256), 0, st, a); }
    else    { if (tb) hipLaunchKernelGGL((small_gemm_kernel<false, true>), grid, dim3(256), 0, st, a);
              else    hipLaunchKernelGGL((small_gemm_kernel<false, false>), grid, dim3(256), 0, st, a); }
    MI_LAUNCH_CHECK();
    return 0;
}
