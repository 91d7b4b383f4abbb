// VQ-VAE codebook step: nearest codebook row per latent vector,
// replacing torch.cdist + argmin + gather + the two MSE losses of VectorQuantizer.forward
// (reference src/models/vqvae.py:24-43) and their backward.
//
// Forward.  Squared distances in the form torch.cdist itself uses for more than 25 rows,
//     d2[m][k] = ||z_m||^2 + ||e_k||^2 - 2 z_m . e_k      (clamped at 0),
// with the inner products on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32).  One 256-thread
// workgroup owns 128 latent rows (32 when M is small); its z tile stays in LDS, the codebook streams through LDS 128
// rows at a time (it is K*D*4 bytes, L2 resident); every lane keeps a running (min, index) for the
// 16 rows x 1 code column it sees per MFMA tile, the 32 lanes of a row are combined by shuffles
// with the lowest index winning ties (torch.argmin).  The kernel then gathers the winning rows
// (quantised output) and emits one partial sum of ||z - q||^2 per workgroup (summed by the caller:
// both losses are that sum / (M*D), vqvae.py:38-39).  HBM traffic: z once in, q once out.
//
// Backward of g_vq * mean((sg(z) - q)^2) + g_commit * mean((z - sg(q))^2):
//     dz += g_commit * 2 (z - q) / (M D),    dE[idx[m]] += g_vq * 2 (q - z_m) / (M D)   (fp32 atomics).
#include "common.h"

#ifndef MI_VQ_ABL
#define MI_VQ_ABL 0                                          // timing ablations (tools only): 1 no argmin epilogue, 2 no staging, 4 no MFMA
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct VqArgs {
    const float* z; const float* E; int* idx; float* zq; float* partial;
    int M, D, K, ldz, ldq;
};

__device__ __forceinline__ int vq_tile_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Stage `rows` rows of D floats (row stride ld, rows clamped to `last`) into an LDS tile of pitch P: four 16-byte loads in
// flight per thread, unconditional (clamped) so that the compiler keeps them batched.
__device__ __forceinline__ void vq_stage(float* dst, int P, const float* src, size_t ld, int first, int last, int rows, int q4, int t) {
    const int tot = rows * q4;
    for (int e0 = t; e0 < tot; e0 += 1024) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + 256 * u, tot - 1), r = e / q4, c = (e - r * q4) * 4;
            v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)min(first + r, last) * ld + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u, r = e / q4, c = (e - r * q4) * 4;
            if (e < tot) *reinterpret_cast<f32x4*>(dst + r * P + c) = v[u];
        }
    }
}

__device__ __forceinline__ float vq_row_norm(const float* row, int q4) {
    float s = 0.f;
    for (int c = 0; c < q4; ++c) { const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c); s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    return s;
}

// SPLIT = false: 128 rows per workgroup, wave w owns rows 32w..32w+31 and walks every code tile.
// SPLIT = true : 32 rows per workgroup (small M: fills the chip), wave w takes code tile w of every 128-code chunk and the
//                four waves' (min, index) pairs meet in LDS.
// The contraction runs over D rounded up to 8 (zero columns in LDS): within each group of eight columns lane half h feeds
// columns 4h..4h+3 to four MFMA steps, so both operands are one ds_read_b128 per group (pitch D8+4: rows 16 bytes apart
// in bank space, conflict-free) and the next group's reads are issued under the current group's MFMAs.
// NLD = 16-byte loads per thread for one 128-code chunk (D <= 32 NLD: the next chunk rides in registers under the MFMAs).
template <bool SPLIT, int NLD>
__global__ __launch_bounds__(256) void vq_nearest_kernel(const VqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int RB = SPLIT ? 32 : 128;
    const int D8 = (a.D + 7) & ~7, P = D8 + 4, G = D8 / 8;
    float* Zs = lds;                                         // [RB][P]
    float* Es = Zs + RB * P;                                 // [128][P]
    float* zn = Es + 128 * P;                                // [RB]
    float* en = zn + RB;                                     // [128]
    int* win = reinterpret_cast<int*>(en + 128);             // [RB]
    float* red = en + 128 + RB;                              // [8]
    float* wb = red + 8;                                     // SPLIT: [4][32] best, [4][32] index
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int m0 = blockIdx.x * RB;
    const int q4 = a.D / 4;

    if (D8 != a.D) {                                         // zero the padding columns once (staging never touches them)
        for (int r = t; r < RB + 128; r += 256) *reinterpret_cast<f32x4*>(Zs + r * P + D8 - 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    }
    vq_stage(Zs, P, a.z, a.ldz, m0, a.M - 1, RB, q4, t);     // rows past M repeat the last row (never written back)
    __syncthreads();
    if (t < RB) zn[t] = vq_row_norm(Zs + t * P, q4);

    float best[16]; int bidx[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { best[r] = INFINITY; bidx[r] = 0; }
    const float* ap = Zs + ((SPLIT ? 0 : 32 * w) + (l & 31)) * P + 4 * (l >> 5);
    const int etot = 128 * q4;
    int er[NLD], ec4[NLD];                                   // this thread's (row, column) slots of a code chunk
#pragma unroll
    for (int u = 0; u < NLD; ++u) { const int e = min(t + 256 * u, etot - 1); er[u] = e / q4; ec4[u] = (e - er[u] * q4) * 4; }
    f32x4 pre[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) pre[u] = *reinterpret_cast<const f32x4*>(a.E + (size_t)min(er[u], a.K - 1) * a.D + ec4[u]);
    __syncthreads();                                         // zn visible
    float znr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) znr[r] = zn[(SPLIT ? 0 : 32 * w) + vq_tile_row(r, l)];
    for (int c0 = 0; c0 < a.K; c0 += 128) {
        __syncthreads();                                     // previous chunk fully consumed
#pragma unroll
        for (int u = 0; u < NLD; ++u)
            if ((!(MI_VQ_ABL & 2) || c0 == 0) && t + 256 * u < etot) *reinterpret_cast<f32x4*>(Es + er[u] * P + ec4[u]) = pre[u];
        __syncthreads();
        if ((!(MI_VQ_ABL & 2) || c0 == 0) && t < 128) en[t] = vq_row_norm(Es + t * P, q4);
        if (!(MI_VQ_ABL & 2)) {   // next chunk (clamped: the last pass re-reads rows it never uses)
            const int cn = c0 + 128;
#pragma unroll
            for (int u = 0; u < NLD; ++u) pre[u] = *reinterpret_cast<const f32x4*>(a.E + (size_t)min(cn + er[u], a.K - 1) * a.D + ec4[u]);
        }
        __syncthreads();
        const int ntile = min(4, (a.K - c0 + 31) / 32);
        for (int ct = SPLIT ? w : 0; ct < ntile; ct += SPLIT ? 4 : 1) {
            f32x16 acc, acc1;                                // two accumulators over alternate contraction steps:
#pragma unroll                                               // back-to-back MFMAs never wait on each other's result
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
            const float* bp = Es + (32 * ct + (l & 31)) * P + 4 * (l >> 5);
            f32x4 av = *reinterpret_cast<const f32x4*>(ap), bv = *reinterpret_cast<const f32x4*>(bp);
            for (int g = 0; g < ((MI_VQ_ABL & 4) ? 1 : G); ++g) {
                const int gn = min(g + 1, G - 1) * 8;
                const f32x4 an = *reinterpret_cast<const f32x4*>(ap + gn), bn = *reinterpret_cast<const f32x4*>(bp + gn);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc1, 0, 0, 0);
                av = an; bv = bn;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
            const int code = c0 + 32 * ct + (l & 31);
            const float ec = en[32 * ct + (l & 31)];
            if (MI_VQ_ABL & 1) { if (code < a.K && acc[0] + acc[5] + acc[10] + acc[15] < best[0]) { best[0] = acc[0]; bidx[0] = code; } }
            else if (code < a.K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d2 = fmaxf(znr[r] + ec - 2.f * acc[r], 0.f);
                    if (d2 < best[r]) { best[r] = d2; bidx[r] = code; }      // codes ascend per lane: first minimum kept
                }
            }
        }
    }
    // combine the 32 code columns of every row (lanes with equal l >> 5); lowest index wins ties
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float b = best[r]; int i = bidx[r];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const float ob = __shfl_xor(b, o, 64); const int oi = __shfl_xor(i, o, 64);
            if (ob < b || (ob == b && oi < i)) { b = ob; i = oi; }
        }
        if ((l & 31) == 0) {
            if constexpr (SPLIT) { wb[32 * w + vq_tile_row(r, l)] = b; reinterpret_cast<int*>(wb)[128 + 32 * w + vq_tile_row(r, l)] = i; }
            else win[32 * w + vq_tile_row(r, l)] = i;
        }
    }
    __syncthreads();
    if constexpr (SPLIT) {
        if (t < 32) {
            float b = wb[t]; int i = reinterpret_cast<int*>(wb)[128 + t];
#pragma unroll
            for (int v = 1; v < 4; ++v) {
                const float ob = wb[32 * v + t]; const int oi = reinterpret_cast<int*>(wb)[128 + 32 * v + t];
                if (ob < b || (ob == b && oi < i)) { b = ob; i = oi; }
            }
            win[t] = i;
        }
        __syncthreads();
    }
    if (t < RB && m0 + t < a.M) a.idx[m0 + t] = win[t];
    // quantised rows + this workgroup's share of sum ||z - q||^2
    float ssum = 0.f;
    for (int e = t; e < RB * q4; e += 256) {
        const int r = e / q4, c = (e - r * q4) * 4;
        if (m0 + r < a.M) {
            const f32x4 qv = *reinterpret_cast<const f32x4*>(a.E + (size_t)win[r] * a.D + c);
            *reinterpret_cast<f32x4*>(a.zq + (size_t)(m0 + r) * a.ldq + c) = qv;
            const f32x4 d = *reinterpret_cast<const f32x4*>(Zs + r * P + c) - qv;
            ssum += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        }
    }
    ssum = block_sum_256(ssum, red);
    if (t == 0) a.partial[blockIdx.x] = ssum;
}

// One float per lane: a wave's atomics land in D consecutive floats of one codebook row (one or two cache lines per
// instruction instead of eight with a float4 per lane -- the L2 atomic units serialise per line), four elements in flight.
__global__ __launch_bounds__(256) void vq_bwd_kernel(int M, int D, const float* __restrict__ z, int ldz, const float* __restrict__ E,
                                                     const int* __restrict__ idx, float gv, float gc, const float* __restrict__ gdev,
                                                     float* __restrict__ dz, int lddz, int accumulate, float* __restrict__ dE) {
    if (gdev) { gv *= gdev[0]; gc *= gdev[1]; }              // upstream loss gradients that live on the device (no host sync)
    const size_t total = (size_t)M * D, stride = (size_t)gridDim.x * 256;
    for (size_t e0 = blockIdx.x * (size_t)256 + threadIdx.x; e0 < total; e0 += 4 * stride) {
        float diff[4]; size_t qo[4]; int m[4], c[4]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t e = min(e0 + u * stride, total - 1);
            ok[u] = e0 + u * stride < total;
            m[u] = (int)(e / D); c[u] = (int)(e - (size_t)m[u] * D);
            qo[u] = (size_t)idx[m[u]] * D + c[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) diff[u] = z[(size_t)m[u] * ldz + c[u]] - E[qo[u]];
        if (dz) {
            float old[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) old[u] = accumulate ? dz[(size_t)m[u] * lddz + c[u]] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (ok[u]) dz[(size_t)m[u] * lddz + c[u]] = old[u] + gc * diff[u];
        }
        if (dE) {
#pragma unroll
            for (int u = 0; u < 4; ++u) if (ok[u]) atomicAdd(dE + qo[u], -gv * diff[u]);
        }
    }
}

// dst[idx[m]][0..D) += src[m][0..D): the gradient a gather passes back to the table it read.
__global__ __launch_bounds__(256) void vq_scatter_kernel(int M, int D, const float* __restrict__ src, int lds_, const int* __restrict__ idx,
                                                         float* __restrict__ dst) {
    const size_t total = (size_t)M * D;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int m = (int)(e / D), c = (int)(e - (size_t)m * D);
        atomicAdd(dst + (size_t)idx[m] * D + c, src[(size_t)m * lds_ + c]);
    }
}

}  // namespace

static bool vq_split(int M) {                                 // 32-row workgroups while 128-row ones would leave CUs idle
    static const int thr = (int)mi_knob("MI_VQ_SPLIT_BELOW", 384);
    return (M + 127) / 128 < thr;
}

extern "C" int mi_vq_partials(int M) { return vq_split(M) ? (M + 31) / 32 : (M + 127) / 128; }

extern "C" int mi_vq_nearest_fwd(int M, int D, int K, const float* z, int ldz, const float* codebook, int* idx, float* zq, int ldq,
                                 float* loss_partial, void* stream) {
    MI_REQUIRE(M > 0 && K > 0 && D >= 4 && D % 4 == 0 && D <= 128 && z && codebook && idx && zq && loss_partial,
               "needs D % 4 == 0, 4 <= D <= 128");
    MI_REQUIRE(ldz % 4 == 0 && ldq % 4 == 0 && (((uintptr_t)z | (uintptr_t)codebook | (uintptr_t)zq) & 15) == 0, "16-byte aligned rows");
    VqArgs a;
    a.z = z; a.E = codebook; a.idx = idx; a.zq = zq; a.partial = loss_partial; a.M = M; a.D = D; a.K = K; a.ldz = ldz; a.ldq = ldq;
    const bool split = vq_split(M);
    const int rb = split ? 32 : 128, P = ((D + 7) & ~7) + 4;
    const size_t lds = ((size_t)(rb + 128) * P + 2 * rb + 128 + 8 + 256) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define VQ_GO(SP, NLD, GRID) do { \
        static MiPerDevice once; \
        once.run([] { (void)hipFuncSetAttribute((const void*)vq_nearest_kernel<SP, NLD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); hipLaunchKernelGGL((vq_nearest_kernel<SP, NLD>), dim3(GRID), dim3(256), lds, st, a); } while (0)
    const int nld = D <= 32 ? 4 : D <= 64 ? 8 : 16;
    if (split) { if (nld == 4) VQ_GO(true, 4, (M + 31) / 32); else if (nld == 8) VQ_GO(true, 8, (M + 31) / 32); else VQ_GO(true, 16, (M + 31) / 32); }
    else       { if (nld == 4) VQ_GO(false, 4, (M + 127) / 128); else if (nld == 8) VQ_GO(false, 8, (M + 127) / 128); else VQ_GO(false, 16, (M + 127) / 128); }
#undef VQ_GO
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_vq_bwd(int M, int D, int K, const float* z, int ldz, const float* codebook, const int* idx, float g_vq, float g_commit,
                         const float* g_dev, float* dz, int lddz, int accumulate_dz, float* dcodebook, void* stream) {
    MI_REQUIRE(M > 0 && K > 0 && D >= 4 && D % 4 == 0 && z && codebook && idx && (dz || dcodebook), "bad argument");
    MI_REQUIRE(ldz % 4 == 0 && (!dz || lddz % 4 == 0), "ld % 4 == 0");
    const float sc = 2.0f / ((float)M * (float)D);
    long blocks = ((long)M * D + 1023) / 1024; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(vq_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, M, D, z, ldz, codebook, idx, g_vq * sc,
                       g_commit * sc, g_dev, dz, lddz, accumulate_dz, dcodebook);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_vq_scatter_rows(int M, int D, int K, const float* src, int ld, const int* idx, float* table, void* stream) {
    MI_REQUIRE(M > 0 && D > 0 && K > 0 && src && idx && table && ld >= D, "bad argument");
    long blocks = ((long)M * D + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(vq_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, M, D, src, ld, idx, table);
    MI_LAUNCH_CHECK();
    return 0;
}
