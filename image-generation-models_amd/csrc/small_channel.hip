// The two 3-channel ends of the UNet: Conv2d(3, C, 3, padding=1) of downs.0.0.block1 (ddpm.py:116,208)
// and final_conv's Conv2d(C, 3, 1) (ddpm.py:236), forward / dgrad / wgrad.  With 3 (or 3x9 = 27)
// contraction elements an MFMA tile would be 90 % padding, so these are plain fp32 VALU kernels
// bound by the one large tensor they stream (the C-channel activation or its gradient).
#include "common.h"

namespace {

// y[px][co] = b[co] + sum_{tap,ci<Cin} x[px+tap][ci] * w[tap][ci][co];   x has pixel stride ldx (>= 4), Cin <= 4
__global__ __launch_bounds__(256) void conv3x3_small_cin_fwd(int N, int H, int W, int Cin, int Cout, const float* __restrict__ x,
                                                             int ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ y, int ldy) {
    extern __shared__ float wl[];                         // [9*Cin][Cout]
    for (int i = threadIdx.x; i < 9 * Cin * Cout; i += 256) wl[i] = w[i];
    __syncthreads();
    const int nq = Cout / 4;                              // channel quads
    const int q = threadIdx.x % nq, psub = threadIdx.x / nq, pp = 256 / nq;
    const int M = N * H * W;
    for (int m = blockIdx.x * pp + psub; m < M; m += gridDim.x * pp) {
        const int n = m / (H * W), rem = m - n * (H * W);
        const int yy = rem / W, xx = rem - yy * W;
        float4 acc = bias ? *reinterpret_cast<const float4*>(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const float* xp = x + (size_t)((n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * ldx;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float xv = ok ? xp[ci] : 0.f;
                    const float4 wv = *reinterpret_cast<const float4*>(&wl[((ky * 3 + kx) * Cin + ci) * Cout + 4 * q]);
                    acc.x += xv * wv.x; acc.y += xv * wv.y; acc.z += xv * wv.z; acc.w += xv * wv.w;
                }
            }
        *reinterpret_cast<float4*>(y + (size_t)m * ldy + 4 * q) = acc;
    }
}

// dW[tap][ci][co] += sum_px x[px+tap][ci] * dy[px][co]   (Cin <= 4): thread = co, 9*Cin accumulators
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_small_cin_wgrad(int N, int H, int W, int Cout, const float* __restrict__ x, int ldx,
                                                               const float* __restrict__ dy, int lddy, float* __restrict__ dW,
                                                               int px_per_block) {
    const int M = N * H * W;
    const int co = blockIdx.y * 256 + threadIdx.x;
    if (co >= Cout) return;
    float acc[9 * CIN];
#pragma unroll
    for (int i = 0; i < 9 * CIN; ++i) acc[i] = 0.f;
    const int mb = blockIdx.x * px_per_block, me = min(M, mb + px_per_block);
    for (int m = mb; m < me; ++m) {
        const int n = m / (H * W), rem = m - n * (H * W);
        const int yy = rem / W, xx = rem - yy * W;
        const float g = dy[(size_t)m * lddy + co];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;          // wave-uniform (m is per block)
                const float* xp = x + (size_t)((n * H + iy) * W + ix) * ldx;  // same address in every lane: broadcast
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) acc[(ky * 3 + kx) * CIN + ci] += xp[ci] * g;
            }
    }
#pragma unroll
    for (int i = 0; i < 9 * CIN; ++i) atomicAdd(dW + (size_t)i * Cout + co, acc[i]);
}

// ---- 1x1 conv to/from a few channels (Cs <= 4 "small" side, C large side), weights w[c][j] (c < C, j < Cs)
// forward: y[px][j] = b[j] + sum_c x[px][c] w[c][j]; one wave per pixel pair, shuffle reduction
__global__ __launch_bounds__(256) void conv1x1_small_cout_fwd(int M, int C, int Cs, const float* __restrict__ x, int ldx,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ y, int ldy) {
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int m = blockIdx.x * 4 + wv; m < M; m += gridDim.x * 4) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int c = 4 * l; c < C; c += 256) {
            const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)m * ldx + c);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* wr = w + (size_t)(c + k) * Cs;
                a0 += xs[k] * wr[0];
                if (Cs > 1) a1 += xs[k] * wr[1];
                if (Cs > 2) a2 += xs[k] * wr[2];
                if (Cs > 3) a3 += xs[k] * wr[3];
            }
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
        if (l < Cs) {
            float v = l == 0 ? a0 : l == 1 ? a1 : l == 2 ? a2 : a3;
            y[(size_t)m * ldy + l] = v + (bias ? bias[l] : 0.f);
        }
    }
}
// dgrad: dx[px][c] (+)= sum_j dy[px][j] w[c][j]
__global__ __launch_bounds__(256) void conv1x1_small_cout_dgrad(int M, int C, int Cs, const float* __restrict__ dy, int lddy,
                                                                const float* __restrict__ w, float* __restrict__ dx, int lddx, int acc) {
    const int nq = C / 4;
    const size_t tot = (size_t)M * nq;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
        const size_t m = i / nq; const int c = (int)(i % nq) * 4;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < Cs; ++j) g[j] = dy[m * lddy + j];
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float* wr = w + (size_t)(c + k) * Cs;
            float v = 0.f;
            for (int j = 0; j < Cs; ++j) v += g[j] * wr[j];
            o[k] = v;
        }
        float* p = dx + m * lddx + c;
        if (acc) { float4 prev = *reinterpret_cast<const float4*>(p); o[0] += prev.x; o[1] += prev.y; o[2] += prev.z; o[3] += prev.w; }
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
// wgrad: dW[c][j] += sum_px x[px][c] dy[px][j]; thread = c
__global__ __launch_bounds__(256) void conv1x1_small_cout_wgrad(int M, int C, int Cs, const float* __restrict__ x, int ldx,
                                                                const float* __restrict__ dy, int lddy, float* __restrict__ dW,
                                                                int px_per_block) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const int mb = blockIdx.x * px_per_block, me = min(M, mb + px_per_block);
    for (int m = mb; m < me; ++m) {
        const float xv = x[(size_t)m * ldx + c];
        for (int j = 0; j < Cs; ++j) a[j] += xv * dy[(size_t)m * lddy + j];
    }
    for (int j = 0; j < Cs; ++j) atomicAdd(dW + (size_t)c * Cs + j, a[j]);
}

}  // namespace

extern "C" int mi_conv3x3_small_cin_fwd(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w,
                                        const float* bias, float* y, int ldy, void* stream) {
    MI_REQUIRE(x && w && y && Cin >= 1 && Cin <= 4 && Cout % 4 == 0 && Cout <= 1024 && 256 % (Cout / 4) == 0 && ldy % 4 == 0,
               "needs Cin <= 4, Cout a multiple of 4 with Cout/4 dividing 256");
    const int pp = 256 / (Cout / 4);
    long blocks = ((long)N * H * W + pp - 1) / pp; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv3x3_small_cin_fwd, dim3((unsigned)blocks), dim3(256), (size_t)9 * Cin * Cout * 4, (hipStream_t)stream,
                       N, H, W, Cin, Cout, x, ldx, w, bias, y, ldy);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_conv3x3_small_cin_wgrad(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* dy,
                                          int lddy, float* dW, void* stream) {
    MI_REQUIRE(x && dy && dW && Cin >= 1 && Cin <= 4, "needs Cin <= 4");
    const int M = N * H * W, per = (M + 1023) / 1024;
    dim3 grid((M + per - 1) / per, (Cout + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    switch (Cin) {
        case 1: hipLaunchKernelGGL(conv3x3_small_cin_wgrad<1>, grid, dim3(256), 0, st, N, H, W, Cout, x, ldx, dy, lddy, dW, per); break;
        case 2: hipLaunchKernelGGL(conv3x3_small_cin_wgrad<2>, grid, dim3(256), 0, st, N, H, W, Cout, x, ldx, dy, lddy, dW, per); break;
        case 3: hipLaunchKernelGGL(conv3x3_small_cin_wgrad<3>, grid, dim3(256), 0, st, N, H, W, Cout, x, ldx, dy, lddy, dW, per); break;
        default: hipLaunchKernelGGL(conv3x3_small_cin_wgrad<4>, grid, dim3(256), 0, st, N, H, W, Cout, x, ldx, dy, lddy, dW, per); break;
    }
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_conv1x1_small_cout(int op, int M, int C, int Cs, const float* a, int lda, const float* b, int ldb,
                                     const float* w, const float* bias, float* out, int ldo, int accumulate, void* stream) {
    // op 0: forward  (a = x[M][C], out = y[M][Cs], bias)        op 1: dgrad (a = dy[M][Cs], out = dx[M][C])
    // op 2: wgrad    (a = x[M][C], b = dy[M][Cs], out = dW[C][Cs], accumulated atomically)
    MI_REQUIRE(a && out && (op == 2 || w) && Cs >= 1 && Cs <= 4 && C % 4 == 0 && M > 0, "needs Cs <= 4, C % 4 == 0");
    hipStream_t st = (hipStream_t)stream;
    if (op == 0) {
        int blocks = (M + 3) / 4; if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(conv1x1_small_cout_fwd, dim3(blocks), dim3(256), 0, st, M, C, Cs, a, lda, w, bias, out, ldo);
    } else if (op == 1) {
        long blocks = ((long)M * (C / 4) + 255) / 256; if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(conv1x1_small_cout_dgrad, dim3((unsigned)blocks), dim3(256), 0, st, M, C, Cs, a, lda, w, out, ldo, accumulate);
    } else if (op == 2) {
        MI_REQUIRE(b, "wgrad needs dy");
        const int per = (M + 1023) / 1024;
        dim3 grid((M + per - 1) / per, (C + 255) / 256);
        hipLaunchKernelGGL(conv1x1_small_cout_wgrad, grid, dim3(256), 0, st, M, C, Cs, a, lda, b, ldb, out, per);
    } else {
        return mi_set_error(-1, "mi_conv1x1_small_cout: op must be 0, 1 or 2");
    }
    MI_LAUNCH_CHECK();
    return 0;
}
