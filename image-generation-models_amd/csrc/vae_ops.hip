// Operators of the VAE path (BASELINE cfg 1; reference src/models/vae.py, src/networks/basic.py:147-204) that the other
// paths do not have:
//   * nn.BatchNorm2d in training and evaluation mode (the `norm_type="batch"` default of configs/networks/conv_mnist.yaml):
//     batch statistics over all N*H*W rows of a dense NHWC tensor per channel, running-statistics update, backward;
//   * the latent block: chunk(mu, log_sigma) -> z = mu + exp(log_sigma) * eps (vae.py:46-55) with the KL term of
//     src/utils/losses.py:30-32 and its gradient.
// Batch norm runs as three launches each way -- per-channel sums (one atomic pair per channel and workgroup, accumulated in
// fp64 so that E[x^2] - mean^2 keeps its digits), a [C]-thread finalize, and the element-wise apply -- all bound by the one
// activation tensor they stream.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// sums[c] += sum_rows a[r][c], sums[C + c] += sum_rows a[r][c] * b[r][c]  (b == a: sum of squares; b = xhat-free second operand)
// Thread = (channel quad, row lane); a workgroup walks `rows_per_block` rows.  XHAT: b is normalised on the fly from (mean, rstd).
template <bool XHAT>
__global__ __launch_bounds__(256) void bn_sums_kernel(int M, int C, const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      double* __restrict__ sums, int rows_per_block) {
    __shared__ double red[2][256][4];
    const int Q = C / 4, t = threadIdx.x;
    const int lanes = 256 / Q > 0 ? 256 / Q : 1;                 // row lanes per pass (Q <= 256)
    const int q = t % Q, rl = t / Q;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (rl < lanes) {
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {1.f, 1.f, 1.f, 1.f};
        if (XHAT) { mu = *reinterpret_cast<const f32x4*>(mean + 4 * q); rs = *reinterpret_cast<const f32x4*>(rstd + 4 * q); }
        const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
        for (int r = r0 + rl; r < r1; r += lanes) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a + (size_t)r * C + 4 * q);
            f32x4 bv = *reinterpret_cast<const f32x4*>(b + (size_t)r * C + 4 * q);
            if (XHAT) bv = (bv - mu) * rs;
#pragma unroll
            for (int j = 0; j < 4; ++j) { s0[j] += (double)av[j]; s1[j] += (double)av[j] * (double)bv[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][t][j] = s0[j]; red[1][t][j] = s1[j]; }
    __syncthreads();
    if (t < Q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double v0 = 0, v1 = 0;
            for (int k = t; k < lanes * Q; k += Q) { v0 += red[0][k][j]; v1 += red[1][k][j]; }
            atomicAdd(sums + 4 * t + j, v0);
            atomicAdd(sums + C + 4 * t + j, v1);
        }
    }
}

// mean / rstd from the sums; running statistics as torch does (momentum, unbiased variance); clears the sums for the next use
__global__ void bn_finalize_kernel(int M, int C, double* __restrict__ sums, float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[c] / M, var = fmax(sums[C + c] / M - m * m, 0.0);
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(M > 1 ? var * M / (M - 1) : var);
    sums[c] = 0.0; sums[C + c] = 0.0;
}
__global__ void bn_eval_stats_kernel(int C, const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float* __restrict__ mean, float* __restrict__ rstd, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = running_mean[c];
    rstd[c] = rsqrtf(running_var[c] + eps);
}

// y = gamma (x - mean) rstd + beta
__global__ void bn_apply_kernel(size_t n4, int Q, const f32x4* __restrict__ x, const f32x4* __restrict__ mean, const f32x4* __restrict__ rstd,
                                const f32x4* __restrict__ gamma, const f32x4* __restrict__ beta, f32x4* __restrict__ y) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q);
        y[i] = (x[i] - mean[q]) * rstd[q] * gamma[q] + beta[q];
    }
}
// dx = gamma rstd (dy - S0/M - xhat S1/M), S0 = sum dy, S1 = sum dy xhat;  thread 0 row also emits dgamma += S1, dbeta += S0
__global__ void bn_bwd_apply_kernel(size_t n4, int Q, int M, const f32x4* __restrict__ x, const f32x4* dy, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma, double* __restrict__ sums,
                                    f32x4* dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int C = 4 * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + 4 * q), rs = *reinterpret_cast<const f32x4*>(rstd + 4 * q);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * q);
        f32x4 s0, s1;
#pragma unroll
        for (int j = 0; j < 4; ++j) { s0[j] = (float)(sums[4 * q + j] / M); s1[j] = (float)(sums[C + 4 * q + j] / M); }
        const f32x4 xh = (x[i] - mu) * rs;
        dx[i] = g * rs * (dy[i] - s0 - xh * s1);
        if (i < (size_t)Q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (dbeta) dbeta[4 * q + j] += (float)sums[4 * q + j];
                if (dgamma) dgamma[4 * q + j] += (float)sums[C + 4 * q + j];
            }
        }
    }
}
__global__ void clear_kernel(int n, double* p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

// h = [mu | log_sigma] rows of 2L; z = mu + exp(log_sigma) eps; kld += mean_rows(-0.5 sum(1 + 2 ls - mu^2 - exp(2 ls)))
__global__ __launch_bounds__(256) void vae_latent_fwd_kernel(int N, int L, const float* __restrict__ h, int ldh, const float* __restrict__ eps,
                                                             float* __restrict__ z, float* __restrict__ kld) {
    __shared__ float red[8];
    float acc = 0.f;
    const int total = N * L;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int n = i / L, k = i - n * L;
        const float mu = h[(size_t)n * ldh + k], ls = h[(size_t)n * ldh + L + k];
        const float sg = expf(ls);
        z[i] = mu + sg * eps[i];
        acc += -0.5f * (1.f + 2.f * ls - mu * mu - sg * sg);
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0 && kld) atomicAdd(kld, acc / (float)N);
}
// dh_mu = dz + g_kld mu / N;  dh_ls = dz eps exp(ls) + g_kld (exp(2 ls) - 1) / N
__global__ void vae_latent_bwd_kernel(int N, int L, const float* __restrict__ h, int ldh, const float* __restrict__ eps,
                                      const float* __restrict__ dz, float g_kld, const float* __restrict__ g_dev, float* __restrict__ dh, int lddh) {
    const int total = N * L;
    const float gk = g_kld * (g_dev ? g_dev[0] : 1.f) / (float)N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i / L, k = i - n * L;
        const float mu = h[(size_t)n * ldh + k], ls = h[(size_t)n * ldh + L + k];
        const float sg = expf(ls), d = dz[i];
        dh[(size_t)n * lddh + k] = d + gk * mu;
        dh[(size_t)n * lddh + L + k] = d * eps[i] * sg + gk * (sg * sg - 1.f);
    }
}

inline int nblk(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b ? b : 1)); }

}  // namespace

#define ST ((hipStream_t)stream)
#define AL16(p) ((((uintptr_t)(p)) & 15) == 0)

static bool bn_ok(int M, int C) { return M > 0 && C >= 4 && C % 4 == 0 && C <= 1024 && 256 % (C / 4 > 256 ? 256 : C / 4) == 0; }

extern "C" int mi_batchnorm_workspace(int C) { return 2 * C * (int)sizeof(double); }

// training != 0: batch statistics (saved in mean / rstd for the backward), running statistics updated when non-null.
// training == 0: mean / rstd are filled from the running statistics.  ws: 2*C doubles, zero on entry (left zero on exit).
extern "C" int mi_batchnorm_fwd(int M, int C, const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                float* running_mean, float* running_var, float momentum, float eps, int training, void* ws, void* stream) {
    MI_REQUIRE(bn_ok(M, C) && x && gamma && beta && y && mean && rstd && AL16(x) && AL16(y) && AL16(gamma) && AL16(beta) && AL16(mean) && AL16(rstd),
               "needs C % 4 == 0, C/4 a power of two <= 256, 16-byte aligned dense tensors");
    if (training) {
        MI_REQUIRE(ws, "training mode needs the workspace");
        const int rpb = 64 * (C >= 256 ? 1 : 256 / C);
        hipLaunchKernelGGL(bn_sums_kernel<false>, dim3((M + rpb - 1) / rpb), dim3(256), 0, ST, M, C, x, x, nullptr, nullptr, (double*)ws, rpb);
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, ST, M, C, (double*)ws, mean, rstd, running_mean, running_var, momentum, eps);
    } else {
        MI_REQUIRE(running_mean && running_var, "evaluation mode needs the running statistics");
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 63) / 64), dim3(64), 0, ST, C, running_mean, running_var, mean, rstd, eps);
    }
    const size_t n4 = (size_t)M * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(nblk(n4)), dim3(256), 0, ST, n4, C / 4, (const f32x4*)x, (const f32x4*)mean, (const f32x4*)rstd,
                       (const f32x4*)gamma, (const f32x4*)beta, (f32x4*)y);
    MI_LAUNCH_CHECK();
    return 0;
}

// Backward of the training-mode forward.  dx may alias dy; dgamma / dbeta are accumulated (nullable).
extern "C" int mi_batchnorm_bwd(int M, int C, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dy,
                                float* dx, float* dgamma, float* dbeta, void* ws, void* stream) {
    MI_REQUIRE(bn_ok(M, C) && x && mean && rstd && gamma && dy && dx && ws && AL16(x) && AL16(dy) && AL16(dx) && AL16(gamma) && AL16(mean) && AL16(rstd),
               "needs C % 4 == 0, C/4 a power of two <= 256, 16-byte aligned dense tensors");
    const int rpb = 64 * (C >= 256 ? 1 : 256 / C);
    hipLaunchKernelGGL(bn_sums_kernel<true>, dim3((M + rpb - 1) / rpb), dim3(256), 0, ST, M, C, dy, x, mean, rstd, (double*)ws, rpb);
    const size_t n4 = (size_t)M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblk(n4)), dim3(256), 0, ST, n4, C / 4, M, (const f32x4*)x, (const f32x4*)dy, mean, rstd, gamma,
                       (double*)ws, (f32x4*)dx, dgamma, dbeta);
    hipLaunchKernelGGL(clear_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, ST, 2 * C, (double*)ws);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_vae_latent_fwd(int N, int L, const float* h, int ldh, const float* eps, float* z, float* kld, void* stream) {
    MI_REQUIRE(N > 0 && L > 0 && h && eps && z && ldh >= 2 * L, "bad argument");
    hipLaunchKernelGGL(vae_latent_fwd_kernel, dim3(nblk((size_t)N * L)), dim3(256), 0, ST, N, L, h, ldh, eps, z, kld);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_vae_latent_bwd(int N, int L, const float* h, int ldh, const float* eps, const float* dz, float g_kld, const float* g_dev,
                                 float* dh, int lddh, void* stream) {
    MI_REQUIRE(N > 0 && L > 0 && h && eps && dz && dh && ldh >= 2 * L && lddh >= 2 * L, "bad argument");
    hipLaunchKernelGGL(vae_latent_bwd_kernel, dim3(nblk((size_t)N * L)), dim3(256), 0, ST, N, L, h, ldh, eps, dz, g_kld, g_dev, dh, lddh);
    MI_LAUNCH_CHECK();
    return 0;
}
