// Operators of the WGAN-GP path that the DDPM / VQ-VAE paths do not have:
//   * GroupNorm(1, C) -- "layer" norm over a whole sample (src/networks/basic.py:33-37, forced by wgan_gp.py:30-31) --
//     forward, backward and the backward OF the backward, which the gradient penalty needs: wgan_gp.py:83-96
//     differentiates the critic's input gradient again (create_graph=True);
//   * LeakyReLU(0.2) / Tanh and their derivatives, the per-sample interpolation x^ = e x + (1 - e) G(z) (wgan_gp.py:84-87)
//     and the penalty mean((||grad||_2 - 1)^2) with its gradient (wgan_gp.py:95-97).
// Tensors are dense NHWC: a sample is P = H*W pixels x C channels, C % 4 == 0 and 2048 % C == 0 for the norm kernels
// (every thread of a 512-thread workgroup then owns one fixed channel quad).
//
// Sample norm, with x^ = (x - mu) rstd, g^ = gamma * dy, n = P*C, means over the sample:
//   forward    y  = gamma x^ + beta
//   backward   dx = rstd (g^ - mean(g^) - x^ mean(g^ x^)),   dgamma_c += sum_p dy x^,   dbeta_c += sum_p dy
//   backward of dx's dependence on (dy, x, gamma), given the adjoint u of dx:
//       ub = mean(u), mu_ = mean(u x^), gb = mean(g^), mg = mean(g^ x^), A = sum(u g^) - n ub gb - n mu_ mg
//       q      = rstd (u - ub - x^ mu_)
//       adj dy = gamma * q,      dgamma_c += sum_p dy q
//       adj x  = -A x^ rstd^2 / n - rstd^2 [ mg (u - ub - x^ mu_) + mu_ (g^ - gb - x^ mg) ]
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SN_T = 512;

__device__ __forceinline__ float hsum(f32x4 v) { return v.x + v.y + v.z + v.w; }

// Sum over the workgroup of NV values per thread; every thread gets the results.  red: NV * 8 doubles.
// The sample statistics are accumulated in fp64: the second-order formulas subtract nearly equal sums (A below, and
// E[x^2] - mean^2), and an fp32 accumulation of 32k terms loses exactly the digits those differences keep.
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int NV> __device__ __forceinline__ void block_sums(double (&v)[NV], double* red) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum_d(v[i]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) red[i * 8 + (threadIdx.x >> 6)] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < SN_T / 64; ++w) s += red[i * 8 + w];
        v[i] = s;
    }
}

// per-channel sums: threads with equal (t % Q) own the same channel quad; combine them through LDS and add to global
__device__ __forceinline__ void channel_reduce(f32x4 acc, int Q, float* lds4 /* SN_T*4 floats */, float* __restrict__ out) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(lds4 + 4 * threadIdx.x) = acc;
    __syncthreads();
    if ((int)threadIdx.x < Q) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int j = threadIdx.x; j < SN_T; j += Q) s += *reinterpret_cast<const f32x4*>(lds4 + 4 * j);
        atomicAdd(out + 4 * threadIdx.x + 0, s.x); atomicAdd(out + 4 * threadIdx.x + 1, s.y);
        atomicAdd(out + 4 * threadIdx.x + 2, s.z); atomicAdd(out + 4 * threadIdx.x + 3, s.w);
    }
}

__global__ __launch_bounds__(SN_T) void sample_norm_fwd_kernel(int P, int C, const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y,
                                                              float* __restrict__ stats, float eps) {
    __shared__ double red[16];
    const int Q = C / 4, n4 = P * Q, t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * P * C;
    const f32x4* xs = reinterpret_cast<const f32x4*>(x + base);
    double s[2] = {0.0, 0.0};
    for (int e = t; e < n4; e += SN_T) { const f32x4 v = xs[e]; s[0] += (double)hsum(v); s[1] += (double)hsum(v * v); }
    block_sums<2>(s, red);
    const double nd = (double)P * C, meand = s[0] / nd, vard = fmax(s[1] / nd - meand * meand, 0.0);
    const float mean = (float)meand, rstd = (float)(1.0 / sqrt(vard + (double)eps));
    if (t == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = rstd; }
    const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[t % Q], b = reinterpret_cast<const f32x4*>(beta)[t % Q];
    f32x4* ys = reinterpret_cast<f32x4*>(y + base);
    for (int e = t; e < n4; e += SN_T) ys[e] = (xs[e] - mean) * rstd * g + b;
}

// dx = rstd (g^ - gb - x^ mg) [+ extra];  dgamma, dbeta accumulated.  dx may alias dy.
__global__ __launch_bounds__(SN_T) void sample_norm_bwd_kernel(int P, int C, const float* __restrict__ x, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* dy, const float* __restrict__ extra,
                                                              float* dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[16];
    __shared__ __attribute__((aligned(16))) float lds4[SN_T * 4];
    const int Q = C / 4, n4 = P * Q, t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * P * C;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    const f32x4* xs = reinterpret_cast<const f32x4*>(x + base);
    const f32x4* ds = reinterpret_cast<const f32x4*>(dy + base);
    const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[t % Q];
    double s[2] = {0.0, 0.0};
    f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
    for (int e = t; e < n4; e += SN_T) {
        const f32x4 xh = (xs[e] - mean) * rstd, d = ds[e], gh = g * d;
        s[0] += (double)hsum(gh); s[1] += (double)hsum(gh * xh);
        ag += d * xh; ab += d;
    }
    block_sums<2>(s, red);
    const double nd = (double)P * C;
    const float gb = (float)(s[0] / nd), mg = (float)(s[1] / nd);
    const f32x4* ex = extra ? reinterpret_cast<const f32x4*>(extra + base) : nullptr;
    f32x4* os = reinterpret_cast<f32x4*>(dx + base);
    for (int e = t; e < n4; e += SN_T) {
        const f32x4 xh = (xs[e] - mean) * rstd;
        f32x4 r = (g * ds[e] - gb - xh * mg) * rstd;
        if (ex) r += ex[e];
        os[e] = r;
    }
    if (dgamma) channel_reduce(ag, Q, lds4, dgamma);
    if (dbeta) channel_reduce(ab, Q, lds4, dbeta);
}

__global__ __launch_bounds__(SN_T) void sample_norm_bwd2_kernel(int P, int C, const float* __restrict__ x, const float* __restrict__ stats,
                                                               const float* __restrict__ gamma, const float* __restrict__ dy,
                                                               const float* __restrict__ u, float* __restrict__ adj_dy, float* __restrict__ adj_x,
                                                               float* __restrict__ dgamma) {
    __shared__ double red[40];
    __shared__ __attribute__((aligned(16))) float lds4[SN_T * 4];
    const int Q = C / 4, n4 = P * Q, t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * P * C;
    const float mean = stats[2 * blockIdx.x], rstd = stats[2 * blockIdx.x + 1];
    const f32x4* xs = reinterpret_cast<const f32x4*>(x + base);
    const f32x4* ds = reinterpret_cast<const f32x4*>(dy + base);
    const f32x4* us = reinterpret_cast<const f32x4*>(u + base);
    const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[t % Q];
    double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int e = t; e < n4; e += SN_T) {
        const f32x4 xh = (xs[e] - mean) * rstd, gh = g * ds[e], uv = us[e];
        s[0] += (double)hsum(gh); s[1] += (double)hsum(gh * xh); s[2] += (double)hsum(uv); s[3] += (double)hsum(uv * xh);
        s[4] += (double)hsum(uv * gh);
    }
    block_sums<5>(s, red);
    const double nd = (double)P * C, gbd = s[0] / nd, mgd = s[1] / nd, ubd = s[2] / nd, mud = s[3] / nd;
    const double Ad = s[4] - nd * ubd * gbd - nd * mud * mgd;
    const float gb = (float)gbd, mg = (float)mgd, ub = (float)ubd, mu_ = (float)mud;
    const float r2 = rstd * rstd, ca = (float)(-Ad * (double)r2 / nd);
    f32x4 ag = {0.f, 0.f, 0.f, 0.f};
    f32x4* ty = reinterpret_cast<f32x4*>(adj_dy + base);
    f32x4* rx = reinterpret_cast<f32x4*>(adj_x + base);
    for (int e = t; e < n4; e += SN_T) {
        const f32x4 xh = (xs[e] - mean) * rstd, d = ds[e], gh = g * d, uv = us[e];
        const f32x4 pu = uv - ub - xh * mu_, pg = gh - gb - xh * mg;
        const f32x4 q = pu * rstd;
        ty[e] = g * q;
        rx[e] = xh * ca - (pu * mg + pg * mu_) * r2;
        ag += d * q;
    }
    if (dgamma) channel_reduce(ag, Q, lds4, dgamma);
}

constexpr int TPB = 256;
inline int nblk(size_t n) { size_t b = (n + TPB - 1) / TPB; return (int)(b > 8192 ? 8192 : (b ? b : 1)); }

__global__ void leaky_fwd_kernel(size_t n4, const f32x4* __restrict__ x, f32x4* __restrict__ y, float slope) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = x[i];
        y[i] = f32x4{v.x > 0.f ? v.x : slope * v.x, v.y > 0.f ? v.y : slope * v.y, v.z > 0.f ? v.z : slope * v.z, v.w > 0.f ? v.w : slope * v.w};
    }
}
__global__ void leaky_bwd_kernel(size_t n4, const f32x4* __restrict__ y, const f32x4* dy, f32x4* dx, float slope) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 o = y[i], g = dy[i];
        dx[i] = f32x4{o.x > 0.f ? g.x : slope * g.x, o.y > 0.f ? g.y : slope * g.y, o.z > 0.f ? g.z : slope * g.z, o.w > 0.f ? g.w : slope * g.w};
    }
}
__global__ void tanh_fwd_kernel(size_t n4, const f32x4* __restrict__ x, f32x4* __restrict__ y) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = x[i];
        y[i] = f32x4{tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w)};
    }
}
__global__ void tanh_bwd_kernel(size_t n4, const f32x4* __restrict__ y, const f32x4* dy, f32x4* dx) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 o = y[i];
        dx[i] = dy[i] * (1.f - o * o);
    }
}
// out[s][i] = e[s] a[s][i] + (1 - e[s]) b[s][i]
__global__ void lerp_rows_kernel(int per4, size_t n4, const f32x4* __restrict__ a, const f32x4* __restrict__ b, const float* __restrict__ e,
                                 f32x4* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float w = e[i / per4];
        out[i] = a[i] * w + b[i] * (1.f - w);
    }
}
// one workgroup per sample: nrm = ||g_s||, penalty += (nrm - 1)^2 / N, u_s = g_s * scale * (2 / N) (nrm - 1) / nrm
__global__ __launch_bounds__(256) void gp_penalty_kernel(int N, int per4, const f32x4* __restrict__ g, float* __restrict__ penalty,
                                                         f32x4* __restrict__ u, float scale, const float* __restrict__ scale_dev) {
    __shared__ float red[8];
    const f32x4* gs = g + (size_t)blockIdx.x * per4;
    float s = 0.f;
    for (int e = threadIdx.x; e < per4; e += 256) { const f32x4 v = gs[e]; s += hsum(v * v); }
    s = block_sum_256(s, red);
    const float nrm = sqrtf(s);
    if (threadIdx.x == 0 && penalty) atomicAdd(penalty, (nrm - 1.f) * (nrm - 1.f) / (float)N);
    if (u) {
        const float c = scale * (scale_dev ? scale_dev[0] : 1.f) * 2.f / (float)N * (nrm - 1.f) / nrm;
        f32x4* us = u + (size_t)blockIdx.x * per4;
        for (int e = threadIdx.x; e < per4; e += 256) us[e] = gs[e] * c;
    }
}

}  // namespace

#define ST ((hipStream_t)stream)
#define AL16(p) ((((uintptr_t)(p)) & 15) == 0)

static bool sn_ok(int N, int P, int C) { return N > 0 && P > 0 && C >= 4 && C % 4 == 0 && (4 * SN_T) % C == 0; }

extern "C" int mi_sample_norm_supported(int N, int P, int C) { return sn_ok(N, P, C) ? 1 : 0; }

extern "C" int mi_sample_norm_fwd(int N, int P, int C, const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                  float eps, void* stream) {
    MI_REQUIRE(sn_ok(N, P, C) && x && gamma && beta && y && stats && AL16(x) && AL16(y) && AL16(gamma) && AL16(beta),
               "needs C % 4 == 0, 2048 % C == 0, 16-byte aligned dense tensors");
    hipLaunchKernelGGL(sample_norm_fwd_kernel, dim3(N), dim3(SN_T), 0, ST, P, C, x, gamma, beta, y, stats, eps);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_sample_norm_bwd(int N, int P, int C, const float* x, const float* stats, const float* gamma, const float* dy,
                                  const float* extra_dx, float* dx, float* dgamma, float* dbeta, void* stream) {
    MI_REQUIRE(sn_ok(N, P, C) && x && stats && gamma && dy && dx && AL16(x) && AL16(dy) && AL16(dx) && AL16(gamma) && AL16(extra_dx),
               "needs C % 4 == 0, 2048 % C == 0, 16-byte aligned dense tensors");
    hipLaunchKernelGGL(sample_norm_bwd_kernel, dim3(N), dim3(SN_T), 0, ST, P, C, x, stats, gamma, dy, extra_dx, dx, dgamma, dbeta);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_sample_norm_bwd2(int N, int P, int C, const float* x, const float* stats, const float* gamma, const float* dy,
                                   const float* u, float* adj_dy, float* adj_x, float* dgamma, void* stream) {
    MI_REQUIRE(sn_ok(N, P, C) && x && stats && gamma && dy && u && adj_dy && adj_x && AL16(x) && AL16(dy) && AL16(u) && AL16(adj_dy) &&
               AL16(adj_x) && AL16(gamma), "needs C % 4 == 0, 2048 % C == 0, 16-byte aligned dense tensors");
    hipLaunchKernelGGL(sample_norm_bwd2_kernel, dim3(N), dim3(SN_T), 0, ST, P, C, x, stats, gamma, dy, u, adj_dy, adj_x, dgamma);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_leaky_relu_fwd(size_t n, const float* x, float* y, float slope, void* stream) {
    MI_REQUIRE(n > 0 && n % 4 == 0 && x && y && AL16(x) && AL16(y), "n % 4 == 0, 16-byte aligned");
    hipLaunchKernelGGL(leaky_fwd_kernel, dim3(nblk(n / 4)), dim3(TPB), 0, ST, n / 4, (const f32x4*)x, (f32x4*)y, slope);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_leaky_relu_bwd(size_t n, const float* y, const float* dy, float* dx, float slope, void* stream) {
    MI_REQUIRE(n > 0 && n % 4 == 0 && y && dy && dx && AL16(y) && AL16(dy) && AL16(dx), "n % 4 == 0, 16-byte aligned");
    hipLaunchKernelGGL(leaky_bwd_kernel, dim3(nblk(n / 4)), dim3(TPB), 0, ST, n / 4, (const f32x4*)y, (const f32x4*)dy, (f32x4*)dx, slope);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_tanh_fwd(size_t n, const float* x, float* y, void* stream) {
    MI_REQUIRE(n > 0 && n % 4 == 0 && x && y && AL16(x) && AL16(y), "n % 4 == 0, 16-byte aligned");
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(nblk(n / 4)), dim3(TPB), 0, ST, n / 4, (const f32x4*)x, (f32x4*)y);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_tanh_bwd(size_t n, const float* y, const float* dy, float* dx, void* stream) {
    MI_REQUIRE(n > 0 && n % 4 == 0 && y && dy && dx && AL16(y) && AL16(dy) && AL16(dx), "n % 4 == 0, 16-byte aligned");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(nblk(n / 4)), dim3(TPB), 0, ST, n / 4, (const f32x4*)y, (const f32x4*)dy, (f32x4*)dx);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_lerp_rows(int N, size_t per, const float* a, const float* b, const float* e, float* out, void* stream) {
    MI_REQUIRE(N > 0 && per > 0 && per % 4 == 0 && a && b && e && out && AL16(a) && AL16(b) && AL16(out), "per % 4 == 0, 16-byte aligned");
    const size_t n4 = (size_t)N * per / 4;
    hipLaunchKernelGGL(lerp_rows_kernel, dim3(nblk(n4)), dim3(TPB), 0, ST, (int)(per / 4), n4, (const f32x4*)a, (const f32x4*)b, e, (f32x4*)out);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_gp_penalty(int N, size_t per, const float* g, float* penalty, float* u, float scale, const float* scale_dev, void* stream) {
    MI_REQUIRE(N > 0 && per > 0 && per % 4 == 0 && g && (penalty || u) && AL16(g) && AL16(u), "per % 4 == 0, 16-byte aligned");
    hipLaunchKernelGGL(gp_penalty_kernel, dim3(N), dim3(256), 0, ST, N, (int)(per / 4), (const f32x4*)g, penalty, (f32x4*)u, scale, scale_dev);
    MI_LAUNCH_CHECK();
    return 0;
}
