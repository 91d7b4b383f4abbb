// Small element-wise kernels around the UNet: time embedding, Mish on vectors, layout changes,
// the diffusion q_sample / loss / posterior update, fused Adam.  All HBM- or launch-bound.
#include <math.h>
#include "common.h"

namespace {

constexpr int TPB = 256;
inline int nblocks(size_t n, int per = TPB, int cap = 8192) {
    size_t b = (n + per - 1) / per;
    return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

// SinusoidalPosEmb (ddpm.py:52-59): fp32 throughout, like the reference
__global__ void time_embed_kernel(int B, int dim, const int64_t* __restrict__ t, float* __restrict__ out, float step) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * dim) return;
    int b = idx / dim, j = idx % dim, half = dim / 2;
    int jj = j < half ? j : j - half;
    float f = expf((float)jj * -step);
    float ang = (float)t[b] * f;
    out[idx] = j < half ? sinf(ang) : cosf(ang);
}

__global__ void mish_fwd_kernel(size_t n, const float* __restrict__ x, float* __restrict__ y) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = mish_f(x[i]);
}
__global__ void mish_bwd_kernel(size_t n, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dx[i] = dy[i] * mish_grad_f(x[i]);
}

// ReLU on a flat vector, float4 body + scalar tail; y may alias x, dx may alias dy (VQ-VAE networks, src/networks/vqvae.py)
__global__ void relu_fwd_kernel(size_t n, const float* __restrict__ x, float* __restrict__ y) {
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t i = i0; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        reinterpret_cast<float4*>(y)[i] = v;
    }
    for (size_t i = 4 * n4 + i0; i < n; i += stride) y[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd_kernel(size_t n, const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, int accumulate) {
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (size_t i = i0; i < n4; i += stride) {
        const float4 o = reinterpret_cast<const float4*>(y)[i], g = reinterpret_cast<const float4*>(dy)[i];
        float4 r = {o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f};
        if (accumulate) { const float4 a = reinterpret_cast<const float4*>(dx)[i]; r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w; }
        reinterpret_cast<float4*>(dx)[i] = r;
    }
    for (size_t i = 4 * n4 + i0; i < n; i += stride) dx[i] = (accumulate ? dx[i] : 0.f) + (y[i] > 0.f ? dy[i] : 0.f);
}

__global__ void nchw_to_nhwc_kernel(int B, int C, int HW, const float* __restrict__ x, float* __restrict__ y, int ld) {
    size_t tot = (size_t)B * HW * ld;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        int c = i % ld; size_t bp = i / ld; int p = bp % HW; int b = bp / HW;
        y[i] = c < C ? x[((size_t)b * C + c) * HW + p] : 0.f;
    }
}
__global__ void nhwc_to_nchw_kernel(int B, int C, int HW, const float* __restrict__ x, int ld, float* __restrict__ y) {
    size_t tot = (size_t)B * C * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        int p = i % HW; size_t bc = i / HW; int c = bc % C; int b = bc / C;
        y[i] = x[((size_t)b * HW + p) * ld + c];
    }
}

// x_t = sqrt(ac[t]) x0 + sqrt(1-ac[t]) eps   (ddpm.py:441-444)
__global__ void q_sample_kernel(int B, int C, int HW, const float* __restrict__ x0, const float* __restrict__ noise,
                                const int64_t* __restrict__ t, const float* __restrict__ sa, const float* __restrict__ sb,
                                float* __restrict__ xt_nhwc, int ld, float* __restrict__ xt_nchw) {
    size_t tot = (size_t)B * HW * ld;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        int c = i % ld; size_t bp = i / ld; int p = bp % HW; int b = bp / HW;
        float v = 0.f;
        if (c < C) {
            size_t j = ((size_t)b * C + c) * HW + p;
            int64_t tb = t[b];
            v = sa[tb] * x0[j] + sb[tb] * noise[j];
            if (xt_nchw) xt_nchw[j] = v;
        }
        if (xt_nhwc) xt_nhwc[i] = v;
    }
}

// loss = mean |target - pred| or mean (target-pred)^2 (ddpm.py:453-456) + gradient wrt pred
__global__ __launch_bounds__(256) void eps_loss_kernel(int B, int C, int HW, const float* __restrict__ pred, int ld,
                                                        const float* __restrict__ target, int loss_type,
                                                        float* __restrict__ loss, float* __restrict__ dpred, float gscale) {
    __shared__ float red[8];
    // (32-bit index arithmetic: the three 64-bit divisions per element made this 6 MB pass a 15 us launch; the host checks B * HW * ld < 2^31)
    const unsigned tot = (unsigned)B * HW * ld;
    const float inv = 1.0f / ((float)B * (float)C * (float)HW);
    float acc = 0.f;
    // four elements per thread and iteration, their loads requested together: the launch was a chain of dependent round trips
    // (128 workgroups -- more would queue on the one atomic -- x 16 iterations of load -> compute -> store)
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < tot; i0 += 4 * stride) {
        float pv[4], tv[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = i0 + u * stride;
            const unsigned ic = min(i, tot - 1);
            const unsigned bp = ic / (unsigned)ld, c = ic - bp * ld, b = bp / (unsigned)HW, p = bp - b * HW;
            live[u] = i < tot && (int)c < C;
            pv[u] = pred[ic];
            tv[u] = target[((size_t)b * C + min(c, (unsigned)C - 1)) * HW + p];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = i0 + u * stride;
            float g = 0.f;
            if (live[u]) {
                const float d = pv[u] - tv[u];
                if (loss_type == 0) { acc += fabsf(d); g = (d > 0.f) ? 1.f : (d < 0.f ? -1.f : 0.f); }
                else { acc += d * d; g = 2.f * d; }
            }
            if (dpred && i < tot) dpred[i] = g * inv * gscale;
        }
    }
    float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(loss, s * inv);
}

// posterior step (ddpm.py:359-397)
// (xp may be x itself -- every element is read and written by the same thread -- so neither is `restrict`: the graph sampler updates its
//  image in place and takes the NHWC copy the next UNet forward reads from the same launch)
__global__ void p_sample_kernel(int B, int C, int HW, const float* x, const float* __restrict__ eps, int ld,
                                const float* __restrict__ z, const int64_t* __restrict__ t, const float* __restrict__ sr,
                                const float* __restrict__ srm1, const float* __restrict__ c1, const float* __restrict__ c2,
                                const float* __restrict__ lv, int clip, float* xp, float* __restrict__ xp_nhwc, int ldo) {
    size_t tot = (size_t)B * C * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        int p = i % HW; size_t bc = i / HW; int c = bc % C; int b = bc / C;
        int64_t tb = t[b];
        float xv = x[i];
        float e = eps[((size_t)b * HW + p) * ld + c];
        float x0 = sr[tb] * xv - srm1[tb] * e;
        if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        float mean = c1[tb] * x0 + c2[tb] * xv;
        float nz = (tb == 0) ? 0.f : 1.f;
        float o = mean + nz * expf(0.5f * lv[tb]) * z[i];
        xp[i] = o;
        if (xp_nhwc) xp_nhwc[((size_t)b * HW + p) * ldo + c] = o;
    }
}

// torch.optim.Adam single-tensor step over a flat buffer
__global__ void adam_kernel(size_t n4, size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float lr, float b1, float b2, float eps, float bc1, float bc2s, float gs) {
    const float step = lr / bc1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = G[j] * gs;
            M[j] = M[j] + (1.f - b1) * (gr - M[j]);
            V[j] = b2 * V[j] + (1.f - b2) * gr * gr;
            P[j] -= step * M[j] / (sqrtf(V[j]) / bc2s + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail
    size_t i = n4 * 4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) {
        float gr = g[i] * gs;
        float mj = m[i] + (1.f - b1) * (gr - m[i]);
        float vj = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mj; v[i] = vj;
        p[i] -= step * mj / (sqrtf(vj) / bc2s + eps);
    }
}

// The same update with the step count and learning rate read from device memory, so that a captured hipGraph of
// (forward, backward, optimizer step) replays correctly: state[0] = step count (the BITS of a uint32, incremented by
// adam_tick_kernel before the update), state[1] = learning rate (float, written by the host when a scheduler changes it).
// The bias corrections 1 - b^t are evaluated in double from the double-precision betas, exactly like the host does for
// adam_kernel (1 - 0.999^t cancels ~4 digits in fp32 for small t), so eager and replayed steps apply the same update.
__global__ void adam_tick_kernel(float* state) { reinterpret_cast<unsigned*>(state)[0] += 1u; }
__global__ void adam_dev_kernel(size_t n4, size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, const float* __restrict__ state, double b1d, double b2d, float eps, float gs) {
    const double t = (double)reinterpret_cast<const unsigned*>(state)[0];
    const float lr = state[1], b1 = (float)b1d, b2 = (float)b2d;
    const float bc1 = (float)(1.0 - pow(b1d, t)), bc2s = sqrtf((float)(1.0 - pow(b2d, t)));
    const float step = lr / bc1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = G[j] * gs;
            M[j] = M[j] + (1.f - b1) * (gr - M[j]);
            V[j] = b2 * V[j] + (1.f - b2) * gr * gr;
            P[j] -= step * M[j] / (sqrtf(V[j]) / bc2s + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    }
    size_t i = n4 * 4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) {
        float gr = g[i] * gs;
        float mj = m[i] + (1.f - b1) * (gr - m[i]);
        float vj = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mj; v[i] = vj;
        p[i] -= step * mj / (sqrtf(vj) / bc2s + eps);
    }
}

__global__ void axpby_kernel(size_t n, float a, const float* __restrict__ x, int acc, float* __restrict__ y) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = a * x[i] + (acc ? y[i] : 0.f);
}

__global__ void axpby2d_kernel(int M, int C, float a, const float* __restrict__ x, int ldx, int acc, float* __restrict__ y, int ldy) {
    size_t tot = (size_t)M * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        size_t m = i / C; int c = i % C;
        float* yp = y + m * ldy + c;
        *yp = a * x[m * ldx + c] + (acc ? *yp : 0.f);
    }
}

__global__ void scale_dev_kernel(int M, int C, float* __restrict__ x, int ld, const float* __restrict__ s) {
    const float a = *s;
    size_t tot = (size_t)M * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x)
        x[(i / C) * ld + (i % C)] *= a;
}

// out[b][:] = table[idx[b]][:]   (time-bias rows for the sampler: the time MLP depends on t only)
__global__ void gather_rows_kernel(int B, int C, const float* __restrict__ table, const int64_t* __restrict__ idx, float* __restrict__ out) {
    size_t tot = (size_t)B * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        size_t b = i / C; int c = i % C;
        out[i] = table[(size_t)idx[b] * C + c];
    }
}

// Device-resident data path (src/datamodules/base.py::DeviceBatchLoader): the whole uint8 dataset [N][H][W][C] lives in HBM; one
// launch gathers the batch's rows and applies the reference transform chain ToTensor -> RandomHorizontalFlip -> Normalize(0.5, 0.5)
// (reference src/datamodules/base.py:37-71) with the same fp32 operations torch performs: u/255, then (v-0.5)/0.5.  Output NCHW.
__global__ void u8_gather_norm_kernel(int B, int C, int H, int W, const uint8_t* __restrict__ data, const int64_t* __restrict__ idx,
                                      const uint8_t* __restrict__ flip, int normalize, float* __restrict__ out) {
    const size_t HW = (size_t)H * W, tot = (size_t)B * C * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / (C * HW), r = i % (C * HW);
        const int c = (int)(r / HW), p = (int)(r % HW), y = p / W, x = p % W;
        const int xs = (flip && flip[b]) ? W - 1 - x : x;
        float v = (float)data[((size_t)idx[b] * HW + (size_t)y * W + xs) * C + c] / 255.0f;
        if (normalize) v = (v - 0.5f) / 0.5f;
        out[i] = v;
    }
}

// fp32 [M][C] (row stride ldx) -> bf16 [M][C] (row stride ldy), round-to-nearest-even: the rounding every bf16-MFMA kernel applies
// when it stages an fp32 activation, applied once for all of its consumers
__global__ void f32_to_bf16_kernel(size_t M, int C4, const float* __restrict__ x, int ldx, uint16_t* __restrict__ y, int ldy) {
    MI_PRIO_UP();
    const size_t tot = M * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / C4; const int c = (int)(i % C4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + m * ldx + c);
        *reinterpret_cast<uint2*>(y + m * ldy + c) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
    }
}

// y = bf16(x) (CVT) and / or column sums of x (SUM) in ONE pass over x: a thread owns a channel quad and walks rows (256 / C4 rows
// per sweep of the workgroup, four independent float4 loads in flight), then one LDS combine per workgroup.  The sums leave either
// as 4 atomics per quad (ATOMIC: out[c] +=) or as this workgroup's row of a partial matrix (out[block][C], plain stores).
// Device-scope fp32 atomics on ONE address from all eight XCDs serialise at ~40 ns each (measured: 512 workgroups -> 20 us,
// 4096 -> 150 us, whatever the bytes), so a big tensor goes through partial rows and a second, small pass of the same kernel.
template <bool CVT, bool SUM, bool ATOMIC>
__global__ __launch_bounds__(256) void cvt_colsum_kernel(int M, int C4, const float* __restrict__ x, int ldx, uint16_t* __restrict__ y, int ldy,
                                                         float* __restrict__ out, int rows_per_block) {
    MI_PRIO_UP();
    __shared__ float4 red[256];
    const int nr = 256 / C4, tc = threadIdx.x % C4, tr = threadIdx.x / C4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tr < nr) {
        const int mb = blockIdx.x * rows_per_block, me = min(M, mb + rows_per_block);
        int m = mb + tr;
        for (; m + 3 * nr < me; m += 4 * nr) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(x + (size_t)(m + k * nr) * ldx + tc * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (CVT) *reinterpret_cast<uint2*>(y + (size_t)(m + k * nr) * ldy + tc * 4) =
                             make_uint2(pack_bf16(v[k].x, v[k].y), pack_bf16(v[k].z, v[k].w));
                if (SUM) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
            }
        }
        for (; m < me; m += nr) {
            const float4 v = *reinterpret_cast<const float4*>(x + (size_t)m * ldx + tc * 4);
            if (CVT) *reinterpret_cast<uint2*>(y + (size_t)m * ldy + tc * 4) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
            if (SUM) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        }
    }
    if (!SUM) return;
    red[threadIdx.x] = s;
    __syncthreads();
    if (tr == 0) {
        for (int r = 1; r < nr; ++r) { const float4 o = red[r * C4 + tc]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
        if (ATOMIC) {
            atomicAdd(out + tc * 4, s.x); atomicAdd(out + tc * 4 + 1, s.y); atomicAdd(out + tc * 4 + 2, s.z); atomicAdd(out + tc * 4 + 3, s.w);
        } else {
            *reinterpret_cast<float4*>(out + (size_t)blockIdx.x * (C4 * 4) + tc * 4) = s;
        }
    }
}


// dst[c] += sum_r src[r * ld + c] for up to MI_ROWSUM_MAX (src, dst) pairs in ONE launch: the second pass of every reduction of a
// backward pass that leaves per-workgroup partial rows (channel LayerNorm's dg / db, round 4).  A workgroup = 64 columns x 64 rows of
// one item (thread = column, row lane; 16 rows per thread, all loads issued before the first add), LDS combine of the four row lanes,
// one atomic per column: rows / 64 atomics per address instead of one per producing workgroup.
struct RowSumArgs { MiRowSum it[MI_ROWSUM_MAX]; int first[MI_ROWSUM_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void rowsum_batch_kernel(const RowSumArgs a) {
    MI_PRIO_UP();
    __shared__ float red[4][64];
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.first[i + 1]) ++i;
    const MiRowSum it = a.it[i];
    const int local = blockIdx.x - a.first[i], ctiles = (it.cols + 63) >> 6;
    const int c = (local % ctiles) * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, r0 = (local / ctiles) * 64;
    const int cc = min(c, it.cols - 1);
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = it.src[(size_t)min(r0 + rl + 4 * k, it.rows - 1) * it.ld + cc];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += (r0 + rl + 4 * k < it.rows) ? v[k] : 0.f;
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < it.cols) atomicAdd(it.dst + c, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

}  // namespace

#define ST ((hipStream_t)stream)

// One pass over an fp32 [M][C] tensor (row stride ldx): y_bf16 (optional) = bf16(x), colsum (optional)[c] += column sums.
// Backward uses it on a residual-stream gradient that a weight-gradient kernel wants as bf16 and a bias gradient wants summed.
// workspace (optional, mi_f32_to_bf16_colsum_workspace(M, C) bytes, 16-byte aligned): per-workgroup partial sums, so that the
// pass runs on a full grid; without it the sums are added atomically from at most 64 workgroups (slow for large tensors).
extern "C" size_t mi_f32_to_bf16_colsum_workspace(size_t M, int C) {
    if (C < 4 || C > 1024 || C % 4) return 0;
    const int nr = 256 / (C / 4);
    size_t rows = 16 * (size_t)nr;
    while ((M + rows - 1) / rows > 2048) rows *= 2;
    return ((M + rows - 1) / rows) * (size_t)C * sizeof(float);
}

extern "C" int mi_f32_to_bf16_colsum(size_t M, int C, const float* x, int ldx, void* y_bf16, int ldy, float* colsum, void* workspace,
                                     size_t ws_bytes, void* stream) {
    MI_REQUIRE(M > 0 && M < (1u << 31) && C >= 4 && C <= 1024 && C % 4 == 0 && ldx % 4 == 0 && x && (y_bf16 || colsum) &&
               (((uintptr_t)x) & 15) == 0, "bad argument (4 <= C <= 1024, C and ldx % 4 == 0, x 16-byte aligned)");
    MI_REQUIRE(!y_bf16 || (ldy % 4 == 0 && (((uintptr_t)y_bf16) & 7) == 0), "bf16 output: ldy % 4 == 0, 8-byte aligned");
    const int C4 = C / 4, nr = 256 / C4;
    uint16_t* y = (uint16_t*)y_bf16;
    auto plan = [&](size_t cap) { int rows = 16 * nr; while ((M + rows - 1) / rows > cap) rows *= 2; return rows; };
    if (!colsum) {
        const int rows = plan(4096);
        hipLaunchKernelGGL((cvt_colsum_kernel<true, false, false>), dim3((unsigned)((M + rows - 1) / rows)), dim3(256), 0, ST, (int)M, C4, x, ldx, y, ldy,
                           colsum, rows);
        MI_LAUNCH_CHECK();
        return 0;
    }
    const size_t need = mi_f32_to_bf16_colsum_workspace(M, C);
    const bool two_pass = workspace && (((uintptr_t)workspace) & 15) == 0 && ws_bytes >= need && M >= 4096;
    const int rows = plan(two_pass ? 2048 : 64);
    const unsigned nb = (unsigned)((M + rows - 1) / rows);
    float* part = two_pass ? (float*)workspace : colsum;
    if (two_pass) {
        if (y) hipLaunchKernelGGL((cvt_colsum_kernel<true, true, false>), dim3(nb), dim3(256), 0, ST, (int)M, C4, x, ldx, y, ldy, part, rows);
        else   hipLaunchKernelGGL((cvt_colsum_kernel<false, true, false>), dim3(nb), dim3(256), 0, ST, (int)M, C4, x, ldx, y, ldy, part, rows);
        int rows2 = nr; while ((nb + rows2 - 1) / rows2 > 16) rows2 *= 2;          // the partial rows: <= 16 workgroups add atomically
        hipLaunchKernelGGL((cvt_colsum_kernel<false, true, true>), dim3((nb + rows2 - 1) / rows2), dim3(256), 0, ST, (int)nb, C4, part, C, nullptr, 0,
                           colsum, rows2);
    } else {
        if (y) hipLaunchKernelGGL((cvt_colsum_kernel<true, true, true>), dim3(nb), dim3(256), 0, ST, (int)M, C4, x, ldx, y, ldy, part, rows);
        else   hipLaunchKernelGGL((cvt_colsum_kernel<false, true, true>), dim3(nb), dim3(256), 0, ST, (int)M, C4, x, ldx, y, ldy, part, rows);
    }
    MI_LAUNCH_CHECK();
    return 0;
}

// The first pass only: y_bf16 (optional) = bf16(x) and one row of column sums per workgroup into part[*rows][C] (plain stores, every row
// written; needs mi_f32_to_bf16_colsum_workspace(M, C) bytes) -- the caller adds the rows later, with those of the other reductions of
// the backward pass, in one mi_rowsum_batch launch instead of a second pass per tensor.  *rows = 0: M < 4096, nothing was done (use
// mi_f32_to_bf16_colsum, which adds such a tensor's sums atomically from <= 64 workgroups).
extern "C" int mi_f32_to_bf16_colsum_part(size_t M, int C, const float* x, int ldx, void* y_bf16, int ldy, float* part, size_t part_bytes,
                                          int* rows, void* stream) {
    MI_REQUIRE(M > 0 && M < (1u << 31) && C >= 4 && C <= 1024 && C % 4 == 0 && ldx % 4 == 0 && x && part && rows &&
               (((uintptr_t)x | (uintptr_t)part) & 15) == 0, "bad argument (4 <= C <= 1024, C and ldx % 4 == 0, x / part 16-byte aligned)");
    MI_REQUIRE(!y_bf16 || (ldy % 4 == 0 && (((uintptr_t)y_bf16) & 7) == 0), "bf16 output: ldy % 4 == 0, 8-byte aligned");
    *rows = 0;
    if (M < 4096) return 0;
    MI_REQUIRE(part_bytes >= mi_f32_to_bf16_colsum_workspace(M, C), "part too small (mi_f32_to_bf16_colsum_workspace)");
    const int C4 = C / 4, nr = 256 / C4;
    int rpb = 16 * nr; while ((M + rpb - 1) / rpb > 2048) rpb *= 2;
    const unsigned nb = (unsigned)((M + rpb - 1) / rpb);
    uint16_t* y = (uint16_t*)y_bf16;
    if (y) hipLaunchKernelGGL((cvt_colsum_kernel<true, true, false>), dim3(nb), dim3(256), 0, ST, (int)M, C4, x, ldx, y, ldy, part, rpb);
    else   hipLaunchKernelGGL((cvt_colsum_kernel<false, true, false>), dim3(nb), dim3(256), 0, ST, (int)M, C4, x, ldx, y, ldy, part, rpb);
    MI_LAUNCH_CHECK();
    *rows = (int)nb;
    return 0;
}

extern "C" int mi_rowsum_batch(int n, const MiRowSum* items, void* stream) {
    MI_REQUIRE(n > 0 && n <= MI_ROWSUM_MAX && items, "1 <= n <= MI_ROWSUM_MAX items");
    RowSumArgs a{};
    a.n = n;
    int tot = 0;
    for (int i = 0; i < n; ++i) {
        const MiRowSum& it = items[i];
        MI_REQUIRE(it.src && it.dst && it.rows > 0 && it.cols > 0 && it.ld >= it.cols, "bad item");
        a.it[i] = it; a.first[i] = tot;
        tot += ((it.cols + 63) / 64) * ((it.rows + 63) / 64);
    }
    a.first[n] = tot;
    hipLaunchKernelGGL(rowsum_batch_kernel, dim3(tot), dim3(256), 0, ST, a);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_f32_to_bf16(size_t M, int C, const float* x, int ldx, void* y_bf16, int ldy, void* stream) {
    MI_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && x && y_bf16, "bad argument (C, ld % 4 == 0)");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(nblocks(M * (C / 4))), dim3(TPB), 0, ST, M, C / 4, x, ldx, (uint16_t*)y_bf16, ldy);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_u8_gather_normalize(int B, int C, int H, int W, const uint8_t* data, const int64_t* idx, const uint8_t* flip,
                                      int normalize, float* out_nchw, void* stream) {
    MI_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && data && idx && out_nchw, "bad argument");
    hipLaunchKernelGGL(u8_gather_norm_kernel, dim3(nblocks((size_t)B * C * H * W)), dim3(TPB), 0, ST, B, C, H, W, data, idx, flip,
                       normalize, out_nchw);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_time_embed(int B, int dim, const int64_t* t, float* out, void* stream) {
    MI_REQUIRE(B > 0 && dim >= 4 && dim % 2 == 0 && t && out, "bad argument");
    float step = (float)(log(10000.0) / (double)(dim / 2 - 1));
    hipLaunchKernelGGL(time_embed_kernel, dim3(nblocks((size_t)B * dim)), dim3(TPB), 0, ST, B, dim, t, out, step);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_mish_fwd(size_t n, const float* x, float* y, void* stream) {
    MI_REQUIRE(n > 0 && x && y, "bad argument");
    hipLaunchKernelGGL(mish_fwd_kernel, dim3(nblocks(n)), dim3(TPB), 0, ST, n, x, y);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_mish_bwd(size_t n, const float* x, const float* dy, float* dx, void* stream) {
    MI_REQUIRE(n > 0 && x && dy && dx, "bad argument");
    hipLaunchKernelGGL(mish_bwd_kernel, dim3(nblocks(n)), dim3(TPB), 0, ST, n, x, dy, dx);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_relu_fwd(size_t n, const float* x, float* y, void* stream) {
    MI_REQUIRE(n > 0 && x && y && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, "bad argument");
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(nblocks((n + 3) / 4)), dim3(TPB), 0, ST, n, x, y);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_relu_bwd(size_t n, const float* y, const float* dy, float* dx, int accumulate, void* stream) {
    MI_REQUIRE(n > 0 && y && dy && dx && (((uintptr_t)y | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0, "bad argument");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(nblocks((n + 3) / 4)), dim3(TPB), 0, ST, n, y, dy, dx, accumulate);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_nchw_to_nhwc(int B, int C, int HW, const float* x, float* y, int ld, void* stream) {
    MI_REQUIRE(B > 0 && C > 0 && HW > 0 && ld >= C && x && y, "bad argument");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblocks((size_t)B * HW * ld)), dim3(TPB), 0, ST, B, C, HW, x, y, ld);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_nhwc_to_nchw(int B, int C, int HW, const float* x, int ld, float* y, void* stream) {
    MI_REQUIRE(B > 0 && C > 0 && HW > 0 && ld >= C && x && y, "bad argument");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblocks((size_t)B * C * HW)), dim3(TPB), 0, ST, B, C, HW, x, ld, y);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_q_sample(int B, int C, int HW, const float* x0, const float* noise, const int64_t* t,
                           const float* sqrt_ac, const float* sqrt_1mac, float* xt_nhwc, int ld, float* xt_nchw,
                           void* stream) {
    MI_REQUIRE(B > 0 && C > 0 && HW > 0 && ld >= C && x0 && noise && t && sqrt_ac && sqrt_1mac && (xt_nhwc || xt_nchw), "bad argument");
    hipLaunchKernelGGL(q_sample_kernel, dim3(nblocks((size_t)B * HW * ld)), dim3(TPB), 0, ST, B, C, HW, x0, noise, t,
                       sqrt_ac, sqrt_1mac, xt_nhwc, ld, xt_nchw);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_eps_loss(int B, int C, int HW, const float* pred, int ld, const float* target, int loss_type,
                           float* loss, float* dpred, float gscale, void* stream) {
    MI_REQUIRE(B > 0 && C > 0 && HW > 0 && ld >= C && pred && target && loss && (loss_type == 0 || loss_type == 1) && (size_t)B * HW * ld < (1ull << 31),
               "bad argument");
    // (one device-scope atomic per workgroup on ONE address: they serialise at ~16 ns each -- 1024 workgroups spent 16 of the
    //  kernel's 17 us queueing there; 128 workgroups stream the same bytes in ~3)
    hipLaunchKernelGGL(eps_loss_kernel, dim3(nblocks((size_t)B * HW * ld, 4 * TPB, 128)), dim3(TPB), 0, ST, B, C, HW, pred, ld,
                       target, loss_type, loss, dpred, gscale);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_p_sample_update(int B, int C, int HW, const float* x, const float* eps_hat, int ld, const float* z,
                                  const int64_t* t, const float* sqrt_recip_ac, const float* sqrt_recipm1_ac,
                                  const float* coef1, const float* coef2, const float* logvar, int clip,
                                  float* x_prev, float* x_prev_nhwc, int ldo, void* stream) {
    MI_REQUIRE(B > 0 && C > 0 && HW > 0 && x && eps_hat && z && t && x_prev, "bad argument");
    hipLaunchKernelGGL(p_sample_kernel, dim3(nblocks((size_t)B * C * HW)), dim3(TPB), 0, ST, B, C, HW, x, eps_hat, ld, z, t,
                       sqrt_recip_ac, sqrt_recipm1_ac, coef1, coef2, logvar, clip, x_prev, x_prev_nhwc, ldo);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_adam_step(size_t n, float* p, const float* g, float* m, float* v, float lr, float b1, float b2,
                            float eps, float bc1, float bc2, float gscale, void* stream) {
    MI_REQUIRE(n > 0 && p && g && m && v, "bad argument");
    MI_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "buffers must be 16-byte aligned");
    size_t n4 = n / 4;
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n4 ? n4 : 1, TPB, 4096)), dim3(TPB), 0, ST, n4, n, p, g, m, v, lr, b1, b2,
                       eps, bc1, sqrtf(bc2), gscale);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_adam_tick(float* state, void* stream) {
    MI_REQUIRE(state, "bad argument");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, ST, state);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_adam_step_dev(size_t n, float* p, const float* g, float* m, float* v, const float* state, double b1, double b2,
                                float eps, float gscale, void* stream) {
    MI_REQUIRE(n > 0 && p && g && m && v && state, "bad argument");
    MI_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "buffers must be 16-byte aligned");
    size_t n4 = n / 4;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(nblocks(n4 ? n4 : 1, TPB, 4096)), dim3(TPB), 0, ST, n4, n, p, g, m, v, state, b1, b2, eps, gscale);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_axpby(size_t n, float a, const float* x, int accumulate, float* y, void* stream) {
    MI_REQUIRE(n > 0 && x && y, "bad argument");
    hipLaunchKernelGGL(axpby_kernel, dim3(nblocks(n)), dim3(TPB), 0, ST, n, a, x, accumulate, y);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_axpby2d(int M, int C, float a, const float* x, int ldx, int accumulate, float* y, int ldy, void* stream) {
    MI_REQUIRE(M > 0 && C > 0 && x && y && ldx >= C && ldy >= C, "bad argument");
    hipLaunchKernelGGL(axpby2d_kernel, dim3(nblocks((size_t)M * C)), dim3(TPB), 0, ST, M, C, a, x, ldx, accumulate, y, ldy);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_scale_by_device_scalar(int M, int C, float* x, int ld, const float* scalar, void* stream) {
    MI_REQUIRE(M > 0 && C > 0 && x && scalar && ld >= C, "bad argument");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(nblocks((size_t)M * C)), dim3(TPB), 0, ST, M, C, x, ld, scalar);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_gather_rows(int B, int C, const float* table, const int64_t* idx, float* out, void* stream) {
    MI_REQUIRE(B > 0 && C > 0 && table && idx && out, "bad argument");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nblocks((size_t)B * C)), dim3(TPB), 0, ST, B, C, table, idx, out);
    MI_LAUNCH_CHECK();
    return 0;
}

// ---- measurement aid (tools/cu_hog.py): `blocks` workgroups that spin for `usec` microseconds of wall clock on `stream`, to see how a
// training step behaves while another stream's kernel (a collective, in production) holds part of the chip.  Not part of the ABI.
namespace {
// mode 0: ALU spin (worst case for co-resident waves); mode 1: streaming copy loop over `buf` (2 x bytes per workgroup), which is
// closer to what a collective's kernel does: mostly waiting on memory -- but at full tilt for the whole step, i.e. gigabytes per step
// where a gradient all-reduce moves ~0.2 GB in and ~0.2 GB out; mode 2: the same copy with the duty cycle of that all-reduce (it streams
// for 0.5 ms of every 5 ms and sleeps in between, the workgroups staying resident as a collective's do while they wait for peers)
__global__ __launch_bounds__(256) void spin_kernel(unsigned long long ticks, float* buf, size_t per_wg, int mode) {
    const unsigned long long t0 = wall_clock64();
    if (mode == 0) {
        float v = threadIdx.x;
        while (wall_clock64() - t0 < ticks) {
#pragma unroll
            for (int i = 0; i < 64; ++i) v = v * 1.0000001f + 0.5f;
        }
        if (v == 12345.678f) buf[0] = v;
        return;
    }
    float4* src = reinterpret_cast<float4*>(buf + (size_t)blockIdx.x * 2 * per_wg);
    float4* dst = src + per_wg / 4;
    if (mode == 2) {
        while (wall_clock64() - t0 < ticks) {
            const unsigned long long p0 = wall_clock64();
            for (size_t i = threadIdx.x; i < per_wg / 4 && wall_clock64() - p0 < 50000ull; i += 256) { float4 v = src[i]; v.x += 1.f; dst[i] = v; }
            while (wall_clock64() - p0 < 500000ull && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
        }
        return;
    }
    while (wall_clock64() - t0 < ticks)
        for (size_t i = threadIdx.x; i < per_wg / 4; i += 256) { float4 v = src[i]; v.x += 1.f; dst[i] = v; }
}
}  // namespace
// ---- measurement aid (bench.py: config.sclk_mhz_under_mfma): the shader clock the chip sustains under matrix-core load.  One workgroup
// per CU issues back-to-back bf16 MFMAs for `usec` microseconds of the 100 MHz wall clock; workgroup 0 reports the s_memtime (shader
// clock) and wall-clock ticks it saw across the loop.  Round 5 measured 1.57 - 1.87 GHz inside conv launches against a 2.4 GHz boost
// clock; a 10 % box-to-box swing of the bench number can be attributed (or not) with this one number.  out: [2] uint64.
namespace {
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long ticks, unsigned long long* out) {
    typedef __attribute__((ext_vector_type(8))) __bf16 b8;
    typedef __attribute__((ext_vector_type(16))) float f16v;
    b8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x & 7)); b[i] = (__bf16)(0.002f * (threadIdx.x & 3)); }
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long w0 = wall_clock64(), s0 = __builtin_amdgcn_s_memtime();
    while (wall_clock64() - w0 < ticks) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
    }
    const unsigned long long s1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v += c0[i] + c1[i] + c2[i] + c3[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = s1 - s0; out[1] = w1 - w0; }
    if (v == 12345.678f) out[2] = 1;         // (keeps the MFMAs; never true)
}
}  // namespace
extern "C" int mi_debug_clock_probe(int blocks, int usec, unsigned long long* out3, void* stream) {
    MI_REQUIRE(blocks > 0 && usec > 0 && out3, "bad argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long)usec * 100ull, out3);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_debug_spin(int blocks, int usec, float* buf, size_t per_wg_floats, int mode, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long)usec * 100ull, buf, per_wg_floats, mode);   // 100 MHz clock
    MI_LAUNCH_CHECK();
    return 0;
}
