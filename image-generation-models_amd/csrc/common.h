// Shared device helpers for the gfx950 DDPM kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <atomic>
#include "../../include/mi_ddpm.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

int mi_set_error(int code, const char* fmt, ...);

#define MI_REQUIRE(cond, msg) \
    do { if (!(cond)) return mi_set_error(-1, "%s: %s", __func__, msg); } while (0)
#define MI_LAUNCH_CHECK() \
    do { hipError_t e_ = hipGetLastError(); \
         if (e_ != hipSuccess) return mi_set_error((int)e_, "%s: %s", __func__, hipGetErrorString(e_)); } while (0)

// One-time set-up PER DEVICE (hipFuncSetAttribute, symbol addresses): a function-local `static bool once = [] {...}()` does the work on
// whichever device is current at the first call only -- a process that later launches on another device (a model moved to cuda:1, a
// single-process multi-GPU host) would run without it.  run() is idempotent work, so two threads racing through it is harmless.
struct MiPerDevice {
    std::atomic<unsigned long long> done{0};
    template <class F> void run(F&& f) {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) { f(); done.fetch_or(bit, std::memory_order_release); }
    }
};

// Zero fill on a stream as a KERNEL (round 5).  hipMemsetAsync becomes a memset node when the step is captured into a hipGraph, and on this
// stack (ROCm 7.0) such a node was seen to take effect out of stream order in replays: the split-K GEMM of the time-MLP backward then added
// its slices onto what the PREVIOUS tenant of that memory had left there (tools/proto/graph_nan.py: gradients of 1e38 from the second replay
// on, one fresh process in ten).  A kernel node keeps its place.  bytes % 4 == 0, p 4-byte aligned.
template <int = 0> __global__ void mi_zero_fill_kernel(uint32_t* __restrict__ p, size_t n) {
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)p) & 15) == 0) {
        for (size_t i = i0; i < n4; i += stride) reinterpret_cast<uint4*>(p)[i] = make_uint4(0u, 0u, 0u, 0u);
        for (size_t i = 4 * n4 + i0; i < n; i += stride) p[i] = 0u;
    } else {
        for (size_t i = i0; i < n; i += stride) p[i] = 0u;
    }
}
inline hipError_t mi_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (!bytes) return hipSuccess;
#ifdef MI_ZERO_WITH_MEMSET        // A/B builds only: the memset-node form (tests/test_kernels_gpu.py::test_split_gemm_in_a_replayed_graph_starts_from_zero)
    return hipMemsetAsync(p, 0, bytes, st);
#endif
    const size_t n = bytes >> 2;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((mi_zero_fill_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), n);
    return hipGetLastError();
}

// two fp32 -> packed bf16x2 (round-to-nearest-even, one v_cvt_pk_bf16_f32)
// Profiling / experiment switches (tile forcing, XCD maps off, alternative plans): read from the environment only in builds with
// -DMI_EXPERIMENT (tools/), constants in the product library -- every one of them would otherwise be an untested configuration.
#ifdef MI_EXPERIMENT
inline long mi_knob(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
#else
inline constexpr long mi_knob(const char*, long dflt) { return dflt; }
#endif

// GroupNorm sums left by a conv's epilogue (sum, sum of squares per sample and 16-channel slab): 64-bit fixed point with 20 fraction
// bits, added with INTEGER atomics -- the totals do not depend on the order in which the workgroups arrive.  (fp32 atomics made the
// fused inference path differ from run to run: 1e-7 in the sums, a few bf16 rounding flips per layer, 7e-3 in the UNet's output.)
// A workgroup's partial sums are formed in a fixed order before they are converted.
// A non-finite partial sum must not turn into a finite statistic (__double2ll_rn maps NaN to 0 and saturates Inf): it raises the slot to
// MI_GSUM_POISON with an atomic max instead -- later (finite) adds move it by far less than 2^56, the reader sees |slot| >= 2^60 and
// returns NaN, and GroupNorm's output is NaN exactly as with the two-pass kernels.
constexpr double MI_GSUM_SCALE = 1048576.0;
constexpr long long MI_GSUM_POISON = 1LL << 62;
__device__ __forceinline__ void gsum_add(void* base, size_t idx, float v) {
    if (__builtin_isfinite(v))
        atomicAdd(reinterpret_cast<unsigned long long*>(base) + idx, (unsigned long long)__double2ll_rn((double)v * MI_GSUM_SCALE));
    else
        atomicMax(reinterpret_cast<long long*>(base) + idx, MI_GSUM_POISON);
}
__device__ __forceinline__ double gsum_get(const void* base, size_t idx) {
    const long long r = reinterpret_cast<const long long*>(base)[idx];
    if (r >= (1LL << 60) || r <= -(1LL << 60)) return __builtin_nan("");
    return (double)r * (1.0 / MI_GSUM_SCALE);
}

// Wave priority of the library's kernels.  A collective's kernel (RCCL, priority 0) that shares a CU with a tile of one of ours takes issue
// slots from that tile's waves; the tile then finishes late and -- one round of tiles per launch -- so does the whole launch
// (tools/cu_hog.py: 8 foreign workgroups cost the step +13..36 %).  With the compute waves at a higher priority the co-runner gets the
// slots they leave, not half of them.  Waves of one kernel all run at the same priority, so nothing changes when the chip is ours.
#ifndef MI_PRIO
#define MI_PRIO 2
#endif
#define MI_PRIO_UP() __builtin_amdgcn_s_setprio(MI_PRIO)

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, p);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for 256-thread blocks; every thread gets the result. red: >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Mish(x) = x*tanh(softplus(x)) with torch's softplus threshold 20 (ddpm.py:62-64).
// tanh(log(1+e)) = (e*e+2e)/(e*e+2e+2) with e = exp(x): one exp, one divide, no cancellation.
__device__ __forceinline__ float mish_f(float x) {
    if (x > 20.f) return x * tanhf(x);           // softplus(x) = x above the threshold
    float e = __expf(x);
    float w = e * (e + 2.f);
    return x * (w / (w + 2.f));
}
// The same two functions for tensors that are stored as bf16 anyway (8 mantissa bits): approximate reciprocal
// (v_rcp_f32, 1 ulp) instead of IEEE division and no branch -- about 40 % fewer VALU cycles per element, which is
// what bounds the GroupNorm kernels.  exp is clamped at 20: tanh(softplus(x)) == 1 in fp32 beyond that.
__device__ __forceinline__ float mish_fast_f(float x) {
    const float e = __expf(fminf(x, 20.f));
    const float w = e * (e + 2.f);
    return x * w * __builtin_amdgcn_rcpf(w + 2.f);
}
__device__ __forceinline__ float mish_grad_fast_f(float x) {
    const float e = __expf(fminf(x, 20.f));
    const float w = e * (e + 2.f);
    const float r = __builtin_amdgcn_rcpf((w + 2.f) * (1.f + e));       // one reciprocal for tanh and sigmoid
    const float th = w * (1.f + e) * r, sg = e * (w + 2.f) * r;
    return th + x * (1.f - th * th) * sg;
}
// d mish / dx = tanh(sp) + x * (1 - tanh(sp)^2) * sigmoid(x)   (sigmoid -> 1 above threshold)
__device__ __forceinline__ float mish_grad_f(float x) {
    if (x > 20.f) { float th = tanhf(x); return th + x * (1.f - th * th); }
    float e = __expf(x);
    float w = e * (e + 2.f);
    float th = w / (w + 2.f);
    float sg = e / (1.f + e);
    return th + x * (1.f - th * th) * sg;
}
