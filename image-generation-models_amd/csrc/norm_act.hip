// GroupNorm(8)+Mish(+time bias)(+residual) and channel LayerNorm, forward and backward.
// HBM-bound kernels: one workgroup owns one (sample, group) slice (GroupNorm) or one wave owns
// one pixel (LayerNorm); the slice is held in registers between the statistics pass and the
// apply pass so HBM sees one read and one write per element; reductions are wave shuffles.
// Replaces aten::native_group_norm(+backward), softplus/tanh/mul (Mish, ddpm.py:62-64), the
// broadcast add at ddpm.py:140, the residual add at ddpm.py:143 and the 7-op LayerNorm
// at ddpm.py:92-95.
#include <stdlib.h>
#include "common.h"

#ifndef MI_GN_UB0
#define MI_GN_UB0 4      // rows per load batch of the uncached GroupNorm backward loops
#endif
#ifndef MI_GN_PD
#define MI_GN_PD 2       // units whose raw loads are in flight ahead of the one being worked on (pipelined GroupNorm backward)
#endif
#ifndef MI_GN_ABL
#define MI_GN_ABL 0       // profiling builds only (bit 0: forward without Mish, bit 1: forward without the statistics' reductions)
#endif
#ifndef MI_GN_FULL16
#define MI_GN_FULL16 0    // A/B builds: 16-unit slices (level 0 of cfg 2) on the pipelined backward -- 4: at four waves per SIMD (spills), 3: at three
#endif
#ifndef MI_GN_WAVES
#define MI_GN_WAVES 4     // waves per SIMD the packed-cache GroupNorm backward is compiled for
#endif

namespace {

template <int VEC> struct V;
template <> struct V<4> {
    float v[4];
    __device__ static V load(const float* p) { float4 t = *reinterpret_cast<const float4*>(p); return V{{t.x, t.y, t.z, t.w}}; }
    __device__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct V<8> {
    float v[8];
    __device__ static V load(const float* p) {
        float4 t = *reinterpret_cast<const float4*>(p), u = *reinterpret_cast<const float4*>(p + 4);
        return V{{t.x, t.y, t.z, t.w, u.x, u.y, u.z, u.w}};
    }
    __device__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct V<1> {
    float v[1];
    __device__ static V load(const float* p) { return V{{*p}}; }
    __device__ void store(float* p) const { *p = v[0]; }
};

// element-type-generic access: T16 = tensor stored as bf16 (offsets count elements)
template <int VEC, bool T16> __device__ __forceinline__ V<VEC> vload(const void* base, size_t off) {
    if constexpr (!T16) return V<VEC>::load(reinterpret_cast<const float*>(base) + off);
    else if constexpr (VEC == 8) {          // 8 bf16 = one 16-byte load
        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + off);
        return V<8>{{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u),
                     __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)}};
    } else if constexpr (VEC == 4) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + off);
        return V<4>{{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)}};
    } else {
        return V<1>{{__uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[off] << 16)}};
    }
}
template <int VEC, bool T16> __device__ __forceinline__ void vstore(void* base, size_t off, const V<VEC>& v) {
    if constexpr (!T16) v.store(reinterpret_cast<float*>(base) + off);
    else if constexpr (VEC == 8)
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + off) =
            make_uint4(pack_bf16(v.v[0], v.v[1]), pack_bf16(v.v[2], v.v[3]), pack_bf16(v.v[4], v.v[5]), pack_bf16(v.v[6], v.v[7]));
    else if constexpr (VEC == 4)
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + off) = make_uint2(pack_bf16(v.v[0], v.v[1]), pack_bf16(v.v[2], v.v[3]));
    else reinterpret_cast<uint16_t*>(base)[off] = (uint16_t)(pack_bf16(v.v[0], 0.f) & 0xffff);
}

// ---- round 6: packed-fp32 arithmetic for the bf16-storage GroupNorm backward.  The kernel is VALU-bound, not HBM-bound (level 0, B = 128:
// ~27 full-rate instructions + 2 transcendentals per element = ~15 us of VALU under a 16 us memory floor, 27.5 us measured; dephasing the
// workgroups changed nothing): two channels per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) and the closed form
//     d mish / dz = e w / n^2,   e = exp(min(z, 20)),  n = (e + 2) e + 2,  w = ((e + 4) e + (4 z + 6)) e + 4 z + 4
// (one exp, one rcp, 11 packed instructions per PAIR; the tanh / sigmoid form above is ~17 per element).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x2 pk_fma_s(f32x2 a, f32x2 sb, f32x2 c) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(sb), "v"(c)); return d; }   // sb: wave-uniform pair
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 unpack_bf16x2(uint32_t v) { return f32x2{__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)}; }
__device__ __forceinline__ f32x2 mish_grad_pk(f32x2 z) {
    const f32x2 zc = {fminf(z.x, 20.f), fminf(z.y, 20.f)};
    const f32x2 k6l = {6.0f, 1.44269504f};                   // ONE scalar pair: low half = 6, high half = log2(e)
    f32x2 zl, ep, n, t1, e4, p1, t2, om, er, g, o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(zl) : "v"(zc), "s"(k6l));
    const f32x2 e = {__builtin_amdgcn_exp2f(zl.x), __builtin_amdgcn_exp2f(zl.y)};
    // (s_nop: a VALU instruction must not read a transcendental's result in the next issue slot, and hipcc does not look inside asm)
    asm("s_nop 0\n\tv_pk_add_f32 %0, %1, 2.0 op_sel_hi:[1,0]" : "=v"(ep) : "v"(e));
    asm("v_pk_fma_f32 %0, %1, %2, 2.0 op_sel_hi:[1,1,0]" : "=v"(n) : "v"(e), "v"(ep));
    const f32x2 r = {__builtin_amdgcn_rcpf(n.x), __builtin_amdgcn_rcpf(n.y)};
    asm("v_pk_fma_f32 %0, %1, 4.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(zc), "s"(k6l));      // 4 z + 6
    // (every statement that reads e or r either carries the s_nop or depends on one that does: hipcc is free to reorder these asm statements,
    //  and an unguarded reader scheduled straight behind the v_exp / v_rcp takes the register's OLD value -- found by the poisoned-allocator
    //  soak test, not by the numerics tests)
    asm("v_pk_add_f32 %0, %1, 2.0 op_sel_hi:[1,0]" : "=v"(e4) : "v"(ep));                                         // e + 4, through ep
    p1 = pk_fma(e4, e, t1);
    asm("v_pk_add_f32 %0, %1, -2.0 op_sel_hi:[1,0]" : "=v"(t2) : "v"(t1));                                       // 4 z + 4
    om = pk_fma(p1, e, t2);
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2" : "=v"(er) : "v"(e), "v"(r));
    g = pk_mul(er, om);
    o = pk_mul(g, r);
    return o;
}

struct GnArgs {
    const float* x; const float* gamma; const float* beta; const float* temb; const float* res;
    float* y; float* stats;
    uint16_t* y16; int ldy16;      // optional second copy of y rounded to bf16 (the conv / weight-gradient operand of the next layer)
    float* coef;                   // statistics-only mode: [3][N][C] = scale (rstd*gamma), shift (beta - mean*scale), time bias; no y
    int N, HW, C, G, Cg; float eps; int ldx, ldy, ldr, ldt;
    // backward
    const float* dout; float* dx; float* dgamma; float* dbeta; float* dtemb; float* dbias; int lddo, lddx;
    int xcd_map;
    int vec8_units;                // > 0: units per thread if a lane took 8 channels (see gn_vec8)
};


// thread layout inside a (n,g) slice: W = Cg/VEC channel units per pixel; unit u = t % W handles
// channels [u*VEC, u*VEC+VEC); pixel rows pr = t / W, step PP = 256 / W.
template <int VEC, int MAXU, int IO = 0>     // IO bit 0: x is bf16, bit 1: y is bf16
__global__ __launch_bounds__(256) void gn_mish_fwd_kernel(const GnArgs a) {
    MI_PRIO_UP();
    constexpr bool X16 = IO & 1, Y16 = IO & 2;
    __shared__ float red[8];
    // Workgroup -> (sample, group).  Consecutive workgroup ids go to different XCDs (id % 8), each with its own L2, while
    // the groups of one sample share cache lines (C/G channels are 32..128 bytes of a pixel row): keep a sample's
    // groups on ONE XCD -- ids xcd + 8*slot with slot = g + G*m are sample xcd + 8*m.
    int n = blockIdx.x / a.G, g = blockIdx.x % a.G;
    if (a.N % 8 == 0 && a.xcd_map) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        g = slot % a.G; n = xcd + 8 * (slot / a.G);
    }
    const int ng = n * a.G + g;
    const int W = a.Cg / VEC, PP = 256 / W;
    const int t = threadIdx.x, u = t % W, pr = t / W;
    const int c0 = g * a.Cg + u * VEC;
    const size_t xoff = (size_t)n * a.HW * a.ldx + c0;        // element offsets (x may be fp32 or bf16)
    const float cnt = (float)a.HW * (float)a.Cg;

    V<VEC> cache[MAXU > 0 ? MAXU : 1];
    float s = 0.f;
    // (round 5) everything the kernel reads is requested BEFORE the statistics: the small slices (8x8 / 16x16 levels: 7-12 us per launch
    // for 8-33 MB) are a chain of dependent round trips -- x, then (after two block reductions) gamma / beta / time bias, then the
    // residual rows inside the apply loop -- and two of the three need nothing that the statistics produce
    float ga0[VEC], be0[VEC], tb[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        ga0[j] = a.gamma[c0 + j]; be0[j] = a.beta[c0 + j];
        tb[j] = a.temb ? a.temb[(size_t)n * a.ldt + c0 + j] : 0.f;
    }
    constexpr bool RPRE = MAXU > 0 && MAXU <= 4;             // residual rows prefetched with the slice (16 more registers at most)
    V<VEC> rpre[RPRE ? MAXU : 1];
    if constexpr (MAXU > 0) {
#pragma unroll
        for (int k = 0; k < MAXU; ++k) cache[k] = vload<VEC, X16>(a.x, xoff + (size_t)min(pr + k * PP, a.HW - 1) * a.ldx);
        if constexpr (RPRE) {
            if (a.res && !a.coef) {
#pragma unroll
                for (int k = 0; k < MAXU; ++k) rpre[k] = V<VEC>::load(a.res + (size_t)n * a.HW * a.ldr + c0 + (size_t)min(pr + k * PP, a.HW - 1) * a.ldr);
            }
        }
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const float live = pr + k * PP < a.HW ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) s += cache[k].v[j] * live;
        }
    } else {
        for (int p = pr; p < a.HW; p += PP) {
            V<VEC> q = vload<VEC, X16>(a.x, xoff + (size_t)p * a.ldx);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s += q.v[j];
        }
    }
#if MI_GN_ABL & 2
    const float mean = s / cnt;
#else
    const float mean = block_sum_256(s, red) / cnt;
#endif
    float s2 = 0.f;
    if constexpr (MAXU > 0) {
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const float live = pr + k * PP < a.HW ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) { float dlt = (cache[k].v[j] - mean) * live; s2 += dlt * dlt; }
        }
    } else {
        for (int p = pr; p < a.HW; p += PP) {
            V<VEC> q = vload<VEC, X16>(a.x, xoff + (size_t)p * a.ldx);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { float dlt = q.v[j] - mean; s2 += dlt * dlt; }
        }
    }
#if MI_GN_ABL & 2
    const float var = s2 / cnt;
#else
    const float var = block_sum_256(s2, red + 4) / cnt;
#endif
    const float rstd = 1.0f / sqrtf(var + a.eps);
    if (t == 0 && a.stats) { a.stats[2 * ng] = mean; a.stats[2 * ng + 1] = rstd; }

    float ga[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        ga[j] = ga0[j] * rstd;
        be[j] = be0[j] - mean * ga[j];
    }
    if (a.coef) {
        // the apply step is folded into the consuming conv's staging (mi_conv3x3_gn_mish): hand it the per-(sample, channel)
        // coefficients instead of writing y
        if (pr == 0) {
            const size_t NC = (size_t)a.N * a.C, o = (size_t)n * a.C + c0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) { a.coef[o + j] = ga[j]; a.coef[NC + o + j] = be[j]; a.coef[2 * NC + o + j] = tb[j]; }
        }
        return;
    }
    const size_t yoff = (size_t)n * a.HW * a.ldy + c0;
    const float* rb = a.res ? a.res + (size_t)n * a.HW * a.ldr + c0 : nullptr;
    auto apply = [&](V<VEC> q, int p, int k = 0) {
        V<VEC> o;
#pragma unroll
#if MI_GN_ABL & 1          // profiling: the apply step without its Mish (what the VALU work costs beside the memory phases)
        for (int j = 0; j < VEC; ++j) o.v[j] = q.v[j] * ga[j] + be[j] + tb[j];
#else
        for (int j = 0; j < VEC; ++j) o.v[j] = (X16 ? mish_fast_f(q.v[j] * ga[j] + be[j]) : mish_f(q.v[j] * ga[j] + be[j])) + tb[j];
#endif
        if (rb) {
            V<VEC> r;
            if constexpr (RPRE) r = rpre[k]; else r = V<VEC>::load(rb + (size_t)p * a.ldr);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] += r.v[j];
        }
        vstore<VEC, Y16>(a.y, yoff + (size_t)p * a.ldy, o);
        if (a.y16) vstore<VEC, true>(a.y16, (size_t)n * a.HW * a.ldy16 + c0 + (size_t)p * a.ldy16, o);
    };
    if constexpr (MAXU > 0) {
#pragma unroll
        for (int k = 0; k < MAXU; ++k) { int p = pr + k * PP; if (p < a.HW) apply(cache[k], p, k); }
    } else {
        for (int p = pr; p < a.HW; p += PP) apply(vload<VEC, X16>(a.x, xoff + (size_t)p * a.ldx), p);
    }
}

// Backward of y = mish(xhat*gamma+beta) + temb + res wrt x (the conv output), gamma, beta, temb.
// FULL: the slice is exactly MAXU units per thread (HW == MAXU * 256 / (Cg / VEC)): no guards, and the packed-cache path runs as a
// straight-line software pipeline (see below).
template <int VEC, int MAXU, int IO = 0, bool FULL = false>     // IO bit 0: x is bf16, bit 1: dx is written as bf16, bit 2: dout is bf16
__global__ __launch_bounds__(256, (((IO & 1) && MAXU > 4 && VEC >= 4) ? (MAXU * VEC > 64 ? 2 : (FULL && MAXU * VEC == 64 && MI_GN_FULL16 == 3) ? 3 : MI_GN_WAVES) : 1)) void gn_mish_bwd_kernel(const GnArgs a) {
    MI_PRIO_UP();
    constexpr bool X16 = IO & 1, DX16 = IO & 2, DO16 = IO & 4;
    __shared__ float part[4][256 * VEC];
    __shared__ float chs[4][128];
    __shared__ float s12[2];
    // Workgroup -> (sample, group).  Consecutive workgroup ids go to different XCDs (id % 8), each with its own L2, while
    // the groups of one sample share cache lines (C/G channels are 32..128 bytes of a pixel row): keep a sample's
    // groups on ONE XCD -- ids xcd + 8*slot with slot = g + G*m are sample xcd + 8*m.
    int n = blockIdx.x / a.G, g = blockIdx.x % a.G;
    if (a.N % 8 == 0 && a.xcd_map) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        g = slot % a.G; n = xcd + 8 * (slot / a.G);
    }
    n = __builtin_amdgcn_readfirstlane(n); g = __builtin_amdgcn_readfirstlane(g);      // wave-uniform: slice bases live in SGPRs
    const int ng = n * a.G + g;
    const int W = a.Cg / VEC, PP = 256 / W;
    const int t = threadIdx.x, u = t % W, pr = t / W;
    const int c0 = g * a.Cg + u * VEC;
    const float mean = a.stats[2 * ng], rstd = a.stats[2 * ng + 1];
    const size_t xoff = (size_t)n * a.HW * a.ldx + c0, dooff = (size_t)n * a.HW * a.lddo + c0;
    const float cnt = (float)a.HW * (float)a.Cg;
    float ga[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { ga[j] = a.gamma[c0 + j]; be[j] = a.beta[c0 + j]; }

    // Cached slice between the two passes.  fp32 x: xhat and dz as fp32.  bf16 x (PKC): x as loaded (packed pairs, exact) and dz
    // rounded to bf16 -- a quarter of the registers, i.e. 4 instead of 2 waves per SIMD for this HBM-bound kernel; xhat is one
    // FMA to recompute, and dx leaves as bf16 (or feeds a bf16-mode consumer) anyway.
    constexpr bool PKC = X16 && MAXU > 4 && VEC >= 4;
    const int cu = u * VEC;                                                                                  // this lane's first channel inside the group
    const uint16_t* xg = reinterpret_cast<const uint16_t*>(a.x) + (size_t)n * a.HW * a.ldx + g * a.Cg;     // (sample, group) slice bases: wave-uniform
    const void* dg = DO16 ? (const void*)(reinterpret_cast<const uint16_t*>(a.dout) + (size_t)n * a.HW * a.lddo + g * a.Cg)
                          : (const void*)(a.dout + (size_t)n * a.HW * a.lddo + g * a.Cg);
    void* dxg = DX16 ? (void*)(reinterpret_cast<uint16_t*>(a.dx) + (size_t)n * a.HW * a.lddx + g * a.Cg)
                     : (void*)(a.dx + (size_t)n * a.HW * a.lddx + g * a.Cg);          // the small slices (MAXU <= 4) hold 32 registers of cache either way
    V<VEC> cx[(MAXU > 0 && !PKC) ? MAXU : 1], cd[(MAXU > 0 && !PKC) ? MAXU : 1];   // cached xhat and dz
    uint32_t px[PKC ? MAXU : 1][VEC >= 2 ? VEC / 2 : 1], pd[PKC ? MAXU : 1][VEC >= 2 ? VEC / 2 : 1];
    const float icnt = 1.0f / cnt;
    float sA[VEC], sD[VEC], sT[VEC], sB[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) sA[j] = sD[j] = sT[j] = sB[j] = 0.f;
    // bf16 x (round 6): two channels per instruction -- packed coefficients, packed accumulators, mish_grad_pk
    constexpr bool PK2 = X16 && VEC >= 2;
    constexpr int H2 = VEC >= 2 ? VEC / 2 : 1;
    f32x2 ga2[H2], be2[H2], aA[H2], aD[H2], aT[H2], aB[H2];
    const f32x2 rstd2 = {rstd, rstd};
    const float nmr = -mean * rstd;
    const f32x2 nmr2 = {nmr, nmr};
    if constexpr (PK2) {
#pragma unroll
        for (int j2 = 0; j2 < H2; ++j2) {
            ga2[j2] = f32x2{ga[2 * j2], ga[2 * j2 + 1]}; be2[j2] = f32x2{be[2 * j2], be[2 * j2 + 1]};
            aA[j2] = aD[j2] = aT[j2] = aB[j2] = f32x2{0.f, 0.f};
        }
    }
    // one pair of channels of one pixel: x as loaded (two bf16), dout as a float pair -> dz (returned), xhat in h2; the four sums advance
    auto pair1q = [&](f32x2 q2, f32x2 d2, int j2, f32x2& h2) -> f32x2 {
        h2 = pk_fma_s(q2, rstd2, nmr2);
        const f32x2 dz2 = pk_mul(d2, mish_grad_pk(pk_fma(h2, ga2[j2], be2[j2])));
        aA[j2] = pk_add(aA[j2], dz2); aD[j2] = pk_fma(dz2, h2, aD[j2]); aT[j2] = pk_add(aT[j2], d2); aB[j2] = pk_add(aB[j2], h2);
        return dz2;
    };
    auto pair1 = [&](uint32_t xr, f32x2 d2, int j2, f32x2& h2) -> f32x2 { return pair1q(unpack_bf16x2(xr), d2, j2, h2); };

    auto pass1 = [&](int p, V<VEC>& xh, V<VEC>& dz) {
        V<VEC> q = vload<VEC, X16>(a.x, xoff + (size_t)p * a.ldx);
        V<VEC> d = vload<VEC, DO16>(a.dout, dooff + (size_t)p * a.lddo);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float h = (q.v[j] - mean) * rstd;
            float z = h * ga[j] + be[j];
            float dzz = d.v[j] * (X16 ? mish_grad_fast_f(z) : mish_grad_f(z));
            xh.v[j] = h; dz.v[j] = dzz;
            sA[j] += dzz; sD[j] += dzz * h; sT[j] += d.v[j]; sB[j] += h;
        }
    };
    if constexpr (MAXU > 0 && MAXU <= 4) {
        // small slices: every load first, unconditionally (rows past the slice re-read its last row and are masked) -- a
        // guarded load is followed by s_waitcnt vmcnt(0), i.e. 2*MAXU serialized round trips.  (For MAXU = 16 the
        // same change costs more in registers / occupancy than it saves: measured 35 -> 46 us at level 0.)
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const size_t p = (size_t)min(pr + k * PP, a.HW - 1);
            cx[k] = vload<VEC, X16>(a.x, xoff + p * a.ldx);
            cd[k] = vload<VEC, DO16>(a.dout, dooff + p * a.lddo);
        }
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const float live = pr + k * PP < a.HW ? 1.f : 0.f;
            if constexpr (PK2) {
                // (a masked row: x = mean and dout = 0 give xhat = 0, dz = 0 -- nothing reaches the sums)
                const float qdead = pr + k * PP < a.HW ? 0.f : 1.f;
#pragma unroll
                for (int j2 = 0; j2 < H2; ++j2) {
                    const f32x2 q2 = {cx[k].v[2 * j2] * live + qdead * mean, cx[k].v[2 * j2 + 1] * live + qdead * mean};
                    f32x2 h2;
                    const f32x2 dz2 = pair1q(q2, f32x2{cd[k].v[2 * j2] * live, cd[k].v[2 * j2 + 1] * live}, j2, h2);
                    cx[k].v[2 * j2] = h2.x; cx[k].v[2 * j2 + 1] = h2.y; cd[k].v[2 * j2] = dz2.x; cd[k].v[2 * j2 + 1] = dz2.y;
                }
            } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float h = (cx[k].v[j] - mean) * rstd * live, dd = cd[k].v[j] * live;
                const float z = h * ga[j] + be[j];
                const float dzz = dd * (X16 ? mish_grad_fast_f(z) : mish_grad_f(z));
                cx[k].v[j] = h; cd[k].v[j] = dzz;
                sA[j] += dzz; sD[j] += dzz * h; sT[j] += dd; sB[j] += h;
            }
            }
        }
    } else if constexpr (PKC && FULL) {
        // Round 4.  The guarded form below is one HBM round trip per unit and thread and, for an 8-unit slice in the 16-unit
        // instantiation, eight dead branches.  With every unit live the loop is straight-line, so unit k + 2's raw loads are issued
        // before unit k is worked on (two units = 8-12 registers in flight, counted vmcnt waits), and a scheduling barrier after
        // every unit keeps the compiler from interleaving the units' exp / rcp chains (which is what spilled when all loads were
        // hoisted).  Measured (B = 128, bf16): 256 channels @16x16 (8 units) 18.5 -> 13.7 us.  The 16-unit form needs ~160 registers
        // against the 128 that four waves per SIMD leave: 70-230 bytes of scratch per lane, or its first units parked in LDS -- both
        // measured equal to the guarded form at level 0 (27.4-28.9 us), which therefore keeps the guarded one.
        constexpr int DQ = DO16 ? VEC / 2 : VEC;                   // dwords of dout per unit
        constexpr int PD = MI_GN_PD;                              // prefetch distance (units)
        uint32_t rx[PD + 1][VEC / 2], rd[PD + 1][DQ];
        // running 32-bit element offsets (one v_add per fetch; opaque so that the 16 units' addresses are not all precomputed)
        uint32_t xo = (uint32_t)(pr * a.ldx + cu), qo = (uint32_t)(pr * a.lddo + cu);
        const uint32_t xstep = (uint32_t)(PP * a.ldx), qstep = (uint32_t)(PP * a.lddo);
        auto fetch = [&](int slot) {
            const uint16_t* xp = xg + xo;
            if constexpr (VEC == 8) {
                const uint4 w = *reinterpret_cast<const uint4*>(xp);
                rx[slot][0] = w.x; rx[slot][1] = w.y; rx[slot][2] = w.z; rx[slot][3] = w.w;
            } else {
                const uint2 w = *reinterpret_cast<const uint2*>(xp);
                rx[slot][0] = w.x; rx[slot][1] = w.y;
            }
            if constexpr (DO16) {
                const uint16_t* dp = reinterpret_cast<const uint16_t*>(dg) + qo;
                if constexpr (VEC == 8) {
                    const uint4 w = *reinterpret_cast<const uint4*>(dp);
                    rd[slot][0] = w.x; rd[slot][1] = w.y; rd[slot][2] = w.z; rd[slot][3] = w.w;
                } else {
                    const uint2 w = *reinterpret_cast<const uint2*>(dp);
                    rd[slot][0] = w.x; rd[slot][1] = w.y;
                }
            } else {
                const float* dp = reinterpret_cast<const float*>(dg) + qo;
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {
                    const uint4 w = *reinterpret_cast<const uint4*>(dp + 4 * q);
                    rd[slot][4 * q] = w.x; rd[slot][4 * q + 1] = w.y; rd[slot][4 * q + 2] = w.z; rd[slot][4 * q + 3] = w.w;
                }
            }
            xo += xstep; qo += qstep;
            asm volatile("" : "+v"(xo), "+v"(qo));
        };
        static_assert(MAXU >= PD, "prefetch distance");
#pragma unroll
        for (int k = 0; k < PD; ++k) fetch(k);
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            if (k + PD < MAXU) fetch((k + PD) % (PD + 1));
            const int sl = k % (PD + 1);
#pragma unroll
            for (int j2 = 0; j2 < VEC / 2; ++j2) {
                px[k][j2] = rx[sl][j2];
                f32x2 d2, h2;
                if constexpr (DO16) d2 = unpack_bf16x2(rd[sl][j2]);
                else d2 = f32x2{__uint_as_float(rd[sl][DO16 ? 0 : 2 * j2]), __uint_as_float(rd[sl][DO16 ? 0 : 2 * j2 + 1])};
                const f32x2 dz2 = pair1(rx[sl][j2], d2, j2, h2);
                pd[k][j2] = pack_bf16(dz2.x, dz2.y);
            }
            // pin the unit's work here: instruction selection orders a basic block by data dependences only, and without these
            // the four accumulation chains of all units sink below the last unit (every h / dz kept alive: 700 bytes of spills)
#pragma unroll
            for (int j2 = 0; j2 < VEC / 2; ++j2) asm volatile("" : "+v"(aA[j2]), "+v"(aD[j2]), "+v"(aT[j2]), "+v"(aB[j2]));
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (PKC) {
        // (Issuing every load of the slice -- or batches of 2..8 units -- before the first use, as the small-slice path above does,
        //  was tried here: one basic block of 16 units makes the scheduler interleave their exp / rcp chains and the register
        //  allocator spills 500-800 bytes per lane at 3 or 4 waves per SIMD.  The guarded form keeps one unit per block; its load
        //  round trips are covered by the 16 waves per CU that the packed cache makes room for.)
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const int p = pr + k * PP;
            if (p < a.HW) {
                // uniform slice base + one 32-bit offset per access
                const uint16_t* xp = xg + (uint32_t)(p * a.ldx + cu);
                if constexpr (VEC == 8) {
                    const uint4 w = *reinterpret_cast<const uint4*>(xp);
                    px[k][0] = w.x; px[k][1] = w.y; px[k][2] = w.z; px[k][3] = w.w;
                } else {
                    const uint2 w = *reinterpret_cast<const uint2*>(xp);
                    px[k][0] = w.x; px[k][1] = w.y;
                }
                const V<VEC> d = vload<VEC, DO16>(dg, (size_t)(uint32_t)(p * a.lddo + cu));
#pragma unroll
                for (int j2 = 0; j2 < VEC / 2; ++j2) {
                    f32x2 h2;
                    const f32x2 dz2 = pair1(px[k][j2], f32x2{d.v[2 * j2], d.v[2 * j2 + 1]}, j2, h2);
                    pd[k][j2] = pack_bf16(dz2.x, dz2.y);
                }
            }
        }
    } else if constexpr (MAXU > 0) {
#pragma unroll
        for (int k = 0; k < MAXU; ++k) { int p = pr + k * PP; if (p < a.HW) pass1(p, cx[k], cd[k]); }
    } else {
        // slices too large for the register cache (64x64 images): four rows' loads are issued before the first use -- one guarded
        // load per iteration is one serialized HBM round trip per row (rows past the slice re-read its last row and are masked)
        constexpr int UB = MI_GN_UB0;
        for (int p0 = pr; p0 < a.HW; p0 += UB * PP) {
            V<VEC> q[UB], d[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const size_t p = (size_t)min(p0 + u * PP, a.HW - 1);
                q[u] = vload<VEC, X16>(a.x, xoff + p * a.ldx);
                d[u] = vload<VEC, DO16>(a.dout, dooff + p * a.lddo);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const float live = p0 + u * PP < a.HW ? 1.f : 0.f;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float h = (q[u].v[j] - mean) * rstd * live, dd = d[u].v[j] * live;
                    const float z = h * ga[j] + be[j];
                    const float dzz = dd * (X16 ? mish_grad_fast_f(z) : mish_grad_f(z));
                    sA[j] += dzz; sD[j] += dzz * h; sT[j] += dd; sB[j] += h;
                }
            }
        }
    }
    if constexpr (PKC || (PK2 && MAXU > 0 && MAXU <= 4)) {
#pragma unroll
        for (int j2 = 0; j2 < H2; ++j2) {
            sA[2 * j2] = aA[j2].x; sA[2 * j2 + 1] = aA[j2].y; sD[2 * j2] = aD[j2].x; sD[2 * j2 + 1] = aD[j2].y;
            sT[2 * j2] = aT[j2].x; sT[2 * j2 + 1] = aT[j2].y; sB[2 * j2] = aB[j2].x; sB[2 * j2 + 1] = aB[j2].y;
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        part[0][t * VEC + j] = sA[j]; part[1][t * VEC + j] = sD[j];
        part[2][t * VEC + j] = sT[j]; part[3][t * VEC + j] = sB[j];
    }
    __syncthreads();
    // per-channel totals: thread (k, c) for k < 4, c < Cg  (Cg <= 64 -> one pass; Cg == 128 -> two)
    for (int idx = t; idx < 4 * a.Cg; idx += 256) {
        int k = idx / a.Cg, c = idx % a.Cg;
        int uu = c / VEC, jj = c % VEC;
        float tot = 0.f;
        for (int r = 0; r < PP; ++r) tot += part[k][(r * W + uu) * VEC + jj];
        chs[k][c] = tot;
    }
    __syncthreads();
    if (t < 64) {
        float v1 = 0.f, v2 = 0.f;
        for (int c = t; c < a.Cg; c += 64) {
            float gm = a.gamma[g * a.Cg + c];
            v1 += gm * chs[0][c]; v2 += gm * chs[1][c];
        }
        v1 = wave_sum(v1); v2 = wave_sum(v2);
        if (t == 0) { s12[0] = v1; s12[1] = v2; }
    }
    __syncthreads();
    const float s1 = s12[0], s2 = s12[1];
    if (t < a.Cg) {
        int c = g * a.Cg + t;
        if (a.dbeta) atomicAdd(a.dbeta + c, chs[0][t]);
        if (a.dgamma) atomicAdd(a.dgamma + c, chs[1][t]);
        if (a.dtemb) a.dtemb[(size_t)n * a.ldt + c] = chs[2][t];
        if (a.dbias) {
            float gm = a.gamma[c];
            atomicAdd(a.dbias + c, rstd * (gm * chs[0][t] - ((float)a.HW * s1 + s2 * chs[3][t]) / cnt));
        }
    }
    const size_t dxoff = (size_t)n * a.HW * a.lddx + c0;
    auto pass2 = [&](int p, const V<VEC>& xh, const V<VEC>& dz) {
        V<VEC> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = rstd * (dz.v[j] * ga[j] - (s1 + xh.v[j] * s2) * icnt);
        vstore<VEC, DX16>(a.dx, dxoff + (size_t)p * a.lddx, o);
    };
    if constexpr (PKC) {
        const f32x2 s2v = {s2, s2}, s1v = {s1, s1}, nri2 = {-rstd * icnt, -rstd * icnt};
        f32x2 rga2[H2];
#pragma unroll
        for (int j2 = 0; j2 < H2; ++j2) rga2[j2] = f32x2{rstd * ga[2 * j2], rstd * ga[2 * j2 + 1]};
        uint32_t so = (uint32_t)(pr * a.lddx + cu);
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const int p = pr + k * PP;
            if (FULL || p < a.HW) {
                // dx = rstd (dz gamma - (s1 + xhat s2) / cnt) = dz (rstd gamma) + (xhat s2 + s1) (-rstd / cnt), two channels per instruction
                V<VEC> o;
#pragma unroll
                for (int j2 = 0; j2 < VEC / 2; ++j2) {
                    const f32x2 h2 = pk_fma_s(unpack_bf16x2(px[k][j2]), rstd2, nmr2);
                    const f32x2 u2 = pk_fma_s(h2, s2v, s1v);
                    const f32x2 o2 = pk_fma_s(u2, nri2, pk_mul(unpack_bf16x2(pd[k][j2]), rga2[j2]));
                    o.v[2 * j2] = o2.x; o.v[2 * j2 + 1] = o2.y;
                }
                if constexpr (FULL) {
                    vstore<VEC, DX16>(dxg, (size_t)so, o);
                    so += (uint32_t)(PP * a.lddx);
                    asm volatile("" : "+v"(so));
                } else {
                    vstore<VEC, DX16>(dxg, (size_t)(uint32_t)(p * a.lddx + cu), o);
                }
            }
        }
    } else if constexpr (MAXU > 0) {
#pragma unroll
        for (int k = 0; k < MAXU; ++k) { int p = pr + k * PP; if (p < a.HW) pass2(p, cx[k], cd[k]); }
    } else {
        constexpr int UB = MI_GN_UB0;
        for (int p0 = pr; p0 < a.HW; p0 += UB * PP) {
            V<VEC> q[UB], d[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const size_t p = (size_t)min(p0 + u * PP, a.HW - 1);
                q[u] = vload<VEC, X16>(a.x, xoff + p * a.ldx);
                d[u] = vload<VEC, DO16>(a.dout, dooff + p * a.lddo);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int p = p0 + u * PP;
                if (p >= a.HW) continue;
                V<VEC> xh, dz;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    xh.v[j] = (q[u].v[j] - mean) * rstd;
                    dz.v[j] = d[u].v[j] * (X16 ? mish_grad_fast_f(xh.v[j] * ga[j] + be[j]) : mish_grad_f(xh.v[j] * ga[j] + be[j]));
                }
                pass2(p, xh, dz);
            }
        }
    }
}

int gn_prepare(const MiGnDesc* d, GnArgs& a, int& vec, int& units) {
    if (!d || d->N <= 0 || d->HW <= 0 || d->C <= 0 || d->G <= 0 || d->C % d->G) return -1;
    a.N = d->N; a.HW = d->HW; a.C = d->C; a.G = d->G; a.Cg = d->C / d->G; a.eps = d->eps;
    static const int xcd_env = (int)mi_knob("MI_GN_XCD", 1);
    a.xcd_map = xcd_env;
    a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = d->ldr;
    int cg = a.Cg;
    if (cg & (cg - 1)) return -2;                 // power of two channel groups only
    if (cg > 128) return -3;
    vec = (cg % 4 == 0) ? 4 : 1;
    if (vec == 1 && cg > 2) return -2;
    int W = cg / vec, PP = 256 / W;
    units = (d->HW + PP - 1) / PP;                // V<VEC> units per thread
    // 8 channels (16 bytes of bf16) per lane: half the load / store instructions for the same bytes, 64-byte instead of 32-byte
    // pieces of a pixel row per pair of lanes.  Callers switch it on with gn_vec8() when every bf16 pointer / stride allows it.
    a.vec8_units = (cg % 8 == 0) ? (d->HW + 256 / (cg / 8) - 1) / (256 / (cg / 8)) : 0;
    return 0;
}

// ------------------------------ channel LayerNorm --------------------------------------------
// one wave per pixel; lanes stride the channel axis in float4; up to 1024 channels cached.
struct LnArgs {
    const float* x; const float* g; const float* b; float* y; const float* dy; float* dx; float* dg; float* db;
    int M, C, ldx, ldy, lddy, lddx, accumulate; float eps;
    float* part;      // backward: this workgroup's row [2 C] = (dg | db) partial sums instead of the 2 C atomics (mi_chan_layernorm_bwd_part)
};
constexpr int LN_MAXV = 4;

// sum over the LPX lanes that share a pixel
template <int LPX> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPX / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LPX lanes per pixel (32 when C <= 128: two pixels per wave, else 64), up to MAXV float4 per lane
template <bool Y16, int LPX, int MAXV>      // Y16: y is written as bf16 (it only feeds the to_qkv 1x1 conv)
__global__ __launch_bounds__(256) void chan_ln_fwd_kernel(const LnArgs a) {
    MI_PRIO_UP();
    constexpr int PPW = 64 / LPX;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lp = l % LPX;
    const int nq = a.C / 4;
    for (int m0 = (blockIdx.x * 4 + w) * PPW; m0 < a.M; m0 += gridDim.x * 4 * PPW) {
        const int m = min(m0 + l / LPX, a.M - 1);
        const bool live = m0 + l / LPX < a.M;
        const float* xp = a.x + (size_t)m * a.ldx;
        float4 c[MAXV];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            int q = lp + LPX * k;
            if (q < nq) { c[k] = *reinterpret_cast<const float4*>(xp + 4 * q); s += c[k].x + c[k].y + c[k].z + c[k].w; }
        }
        const float mean = group_sum<LPX>(s) / (float)a.C;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            if (lp + LPX * k < nq) {
                float d0 = c[k].x - mean, d1 = c[k].y - mean, d2 = c[k].z - mean, d3 = c[k].w - mean;
                s2 += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
        }
        const float var = group_sum<LPX>(s2) / (float)a.C;
        const float inv = 1.0f / (sqrtf(var) + a.eps);
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            int q = lp + LPX * k;
            if (q < nq && live) {
                float4 gg = *reinterpret_cast<const float4*>(a.g + 4 * q);
                float4 bb = *reinterpret_cast<const float4*>(a.b + 4 * q);
                V<4> o;
                o.v[0] = (c[k].x - mean) * inv * gg.x + bb.x; o.v[1] = (c[k].y - mean) * inv * gg.y + bb.y;
                o.v[2] = (c[k].z - mean) * inv * gg.z + bb.z; o.v[3] = (c[k].w - mean) * inv * gg.w + bb.w;
                vstore<4, Y16>(a.y, (size_t)m * a.ldy + 4 * q, o);
            }
        }
    }
}

// y = xc * inv * g + b,  inv = 1/(sqrt(var)+eps).  dx_i = inv*(dh_i - mean(dh)) - inv^2 * S * xc_i / (sigma*C),
// dh = dy*g, S = sum dh*xc.
template <bool DY16, int LPX, int MAXV>     // DY16: dy (the gradient of the LayerNorm output) is stored as bf16
__global__ __launch_bounds__(256) void chan_ln_bwd_kernel(const LnArgs a) {
    MI_PRIO_UP();
    constexpr int PPW = 64 / LPX;
    __shared__ float red[2][4][LPX * MAXV * 4];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lp = l % LPX;
    const int nq = a.C / 4;
    float4 ag[MAXV], ab[MAXV], gg[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        ag[k] = ab[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        int q = lp + LPX * k;
        gg[k] = (q < nq) ? *reinterpret_cast<const float4*>(a.g + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int m0 = (blockIdx.x * 4 + w) * PPW; m0 < a.M; m0 += gridDim.x * 4 * PPW) {
        const int m = min(m0 + l / LPX, a.M - 1);
        const float live = m0 + l / LPX < a.M ? 1.f : 0.f;
        const float* xp = a.x + (size_t)m * a.ldx;
        float4 c[MAXV], d[MAXV];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            int q = lp + LPX * k;
            if (q < nq) {
                c[k] = *reinterpret_cast<const float4*>(xp + 4 * q);
                const V<4> dv = vload<4, DY16>(a.dy, (size_t)m * a.lddy + 4 * q);
                d[k] = make_float4(dv.v[0] * live, dv.v[1] * live, dv.v[2] * live, dv.v[3] * live);
                s += c[k].x + c[k].y + c[k].z + c[k].w;
            }
        }
        const float mean = group_sum<LPX>(s) / (float)a.C;
        float s2 = 0.f, sh = 0.f, sS = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            if (lp + LPX * k < nq) {
                c[k].x -= mean; c[k].y -= mean; c[k].z -= mean; c[k].w -= mean;
                s2 += c[k].x * c[k].x + c[k].y * c[k].y + c[k].z * c[k].z + c[k].w * c[k].w;
                float h0 = d[k].x * gg[k].x, h1 = d[k].y * gg[k].y, h2 = d[k].z * gg[k].z, h3 = d[k].w * gg[k].w;
                sh += h0 + h1 + h2 + h3;
                sS += h0 * c[k].x + h1 * c[k].y + h2 * c[k].z + h3 * c[k].w;
            }
        }
        const float var = group_sum<LPX>(s2) / (float)a.C;
        const float mh = group_sum<LPX>(sh) / (float)a.C;
        const float S = group_sum<LPX>(sS);
        const float sigma = sqrtf(var);
        const float inv = 1.0f / (sigma + a.eps);
        const float k2 = sigma > 0.f ? inv * inv * S / (sigma * (float)a.C) : 0.f;
        float* op = a.dx + (size_t)m * a.lddx;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            int q = lp + LPX * k;
            if (q < nq) {
                float4 o;
                o.x = inv * (d[k].x * gg[k].x - mh) - k2 * c[k].x;
                o.y = inv * (d[k].y * gg[k].y - mh) - k2 * c[k].y;
                o.z = inv * (d[k].z * gg[k].z - mh) - k2 * c[k].z;
                o.w = inv * (d[k].w * gg[k].w - mh) - k2 * c[k].w;
                if (live != 0.f) {
                    if (a.accumulate) {
                        float4 prev = *reinterpret_cast<const float4*>(op + 4 * q);
                        o.x += prev.x; o.y += prev.y; o.z += prev.z; o.w += prev.w;
                    }
                    *reinterpret_cast<float4*>(op + 4 * q) = o;
                }
                ag[k].x += d[k].x * c[k].x * inv; ag[k].y += d[k].y * c[k].y * inv;
                ag[k].z += d[k].z * c[k].z * inv; ag[k].w += d[k].w * c[k].w * inv;
                ab[k].x += d[k].x; ab[k].y += d[k].y; ab[k].z += d[k].z; ab[k].w += d[k].w;
            }
        }
    }
    // combine the pixel halves of a wave, then the 4 waves' per-channel partials: one atomic per channel per block
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        if constexpr (PPW == 2) {
            ag[k].x += __shfl_xor(ag[k].x, 32, 64); ag[k].y += __shfl_xor(ag[k].y, 32, 64); ag[k].z += __shfl_xor(ag[k].z, 32, 64); ag[k].w += __shfl_xor(ag[k].w, 32, 64);
            ab[k].x += __shfl_xor(ab[k].x, 32, 64); ab[k].y += __shfl_xor(ab[k].y, 32, 64); ab[k].z += __shfl_xor(ab[k].z, 32, 64); ab[k].w += __shfl_xor(ab[k].w, 32, 64);
        }
        int q = lp + LPX * k;
        if (q < nq && l < LPX) {
            red[0][w][4 * q + 0] = ag[k].x; red[0][w][4 * q + 1] = ag[k].y; red[0][w][4 * q + 2] = ag[k].z; red[0][w][4 * q + 3] = ag[k].w;
            red[1][w][4 * q + 0] = ab[k].x; red[1][w][4 * q + 1] = ab[k].y; red[1][w][4 * q + 2] = ab[k].z; red[1][w][4 * q + 3] = ab[k].w;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += 256) {
        float v0 = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        float v1 = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        if (a.part) {
            // Round 4: the 2 C atomics of <= 512 workgroups land on the same 2 C addresses and device-scope atomics on one address
            // serialise (~20 ns each: ~10 us at the end of EVERY launch, measured by leaving them out: 57.5 -> 48.2 us at level 0,
            // 19.7 -> 8.5 us for 256 channels @8x8).  Each workgroup leaves its row; mi_rowsum_batch adds the rows of all the step's
            // LayerNorms in one launch.
            a.part[(size_t)blockIdx.x * 2 * a.C + c] = v0;
            a.part[(size_t)blockIdx.x * 2 * a.C + a.C + c] = v1;
        } else {
            if (a.dg) atomicAdd(a.dg + c, v0);
            if (a.db) atomicAdd(a.db + c, v1);
        }
    }
}

}  // namespace

// 8-channel lanes: bf16 tensors only (a lane's 8 channels are one 16-byte access), every stride a multiple of 8 elements, slices
// of 5..8 units per thread (the level-0 layers)
static bool gn_vec8(const GnArgs& a, int io_all16, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
    static const int on = (int)mi_knob("MI_GN_VEC8", 0);
    if (!on || !io_all16 || a.vec8_units <= 4 || a.vec8_units > 8) return false;      // measured on the 1024-pixel slices: backward 36.1 -> 33.0 us with fp32 caches, but no better than 4-channel lanes once the caches are packed (4 waves per SIMD): off by default
    for (int l : lds) if (l % 8) return false;
    for (const void* p : ptrs) if ((uintptr_t)p & 15) return false;
    return true;
}
// Round 4: slices the 4-channel lanes cannot cache (more than 16 units per thread: the 64 x 64 level of cfg 3, C / G = 8, 4096 pixels)
// take 8-channel lanes with a 16-unit packed cache instead of the uncached two-pass form (which reads x and dout twice)
static bool gn_wide16(const GnArgs& a, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
    if (a.vec8_units <= 8 || a.vec8_units > 16) return false;
    for (int l : lds) if (l % 8) return false;
    for (const void* p : ptrs) if ((uintptr_t)p & 15) return false;
    return true;
}
#define GN_DISPATCH_IO_(KERNEL, IOV, FWD)                                                                 \
    do {                                                                                            \
        dim3 grid(a.N * a.G), blk(256);                                                             \
        if (vec == 8) {                                                                             \
            if (a.vec8_units <= 4) hipLaunchKernelGGL((KERNEL<8, 4, IOV>), grid, blk, 0, st, a);    \
            else hipLaunchKernelGGL((KERNEL<8, 8, IOV>), grid, blk, 0, st, a);                      \
        } else if (vec == 4) {                                                                             \
            if (units <= 4) hipLaunchKernelGGL((KERNEL<4, 4, IOV>), grid, blk, 0, st, a);           \
            else if (units <= 16) hipLaunchKernelGGL((KERNEL<4, 16, IOV>), grid, blk, 0, st, a);    \
            else if (units <= 32 && FWD) hipLaunchKernelGGL((KERNEL<4, 32, IOV>), grid, blk, 0, st, a); \
            else hipLaunchKernelGGL((KERNEL<4, 0, IOV>), grid, blk, 0, st, a);                      \
        } else {                                                                                    \
            if (units <= 16) hipLaunchKernelGGL((KERNEL<1, 16, IOV>), grid, blk, 0, st, a);         \
            else hipLaunchKernelGGL((KERNEL<1, 0, IOV>), grid, blk, 0, st, a);                      \
        }                                                                                           \
    } while (0)
#define GN_DISPATCH_IO(KERNEL, IOV) GN_DISPATCH_IO_(KERNEL, IOV, false)
#define GN_DISPATCH_FWD_IO(KERNEL, IOV) GN_DISPATCH_IO_(KERNEL, IOV, true)   /* forward also caches 32-unit slices (64x64 images, C/G = 8) */
#define GN_DISPATCH(KERNEL) GN_DISPATCH_IO(KERNEL, 0)
/* backward, bf16 x: slices of exactly 8 units per thread take the pipelined instantiation (FULL) */
#define GN_DISPATCH_BWD16(IOV)                                                                                          \
    do {                                                                                                                \
        const int ppx = 256 / (a.Cg / 4);                                                                               \
        static const int full_on = (int)mi_knob("MI_GN_FULL", 1);                                                       \
        static const int wide_on = (int)mi_knob("MI_GN_WIDE16", 1);                                                     \
        if (full_on && vec == 4 && d->HW == 8 * ppx) hipLaunchKernelGGL((gn_mish_bwd_kernel<4, 8, IOV, true>), dim3(a.N * a.G), dim3(256), 0, st, a); \
        else if (MI_GN_FULL16 && full_on && vec == 4 && d->HW == 16 * ppx) hipLaunchKernelGGL((gn_mish_bwd_kernel<4, 16, IOV, true>), dim3(a.N * a.G), dim3(256), 0, st, a); \
        else if (wide_on && vec == 4 && units > 16 && gn_wide16(a, {d->ldx, lddo, lddx}, {x, dout, dx, gamma, beta}))    \
            hipLaunchKernelGGL((gn_mish_bwd_kernel<8, 16, IOV, false>), dim3(a.N * a.G), dim3(256), 0, st, a);           \
        else GN_DISPATCH_IO(gn_mish_bwd_kernel, IOV);                                                                   \
    } while (0)

extern "C" int mi_gn_mish_fwd(const MiGnDesc* d, const float* x, const float* gamma, const float* beta,
                              const float* temb, int ldt, const float* residual, float* y, float* stats,
                              void* stream) {
    MI_REQUIRE(x && gamma && beta && y, "null argument");
    GnArgs a{};
    int vec, units;
    int rc = gn_prepare(d, a, vec, units);
    MI_REQUIRE(rc == 0, "C/G must be a power of two <= 128 (and 1, 2 or a multiple of 4)");
    MI_REQUIRE(vec == 1 || (d->ldx % 4 == 0 && d->ldy % 4 == 0 && (!residual || d->ldr % 4 == 0)), "ld must be a multiple of 4");
    a.x = x; a.gamma = gamma; a.beta = beta; a.temb = temb; a.ldt = ldt; a.res = residual; a.y = y; a.stats = stats;
    hipStream_t st = (hipStream_t)stream;
    GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 0);
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_gn_mish_bwd(const MiGnDesc* d, const float* x, const float* stats, const float* gamma,
                              const float* beta, const float* dout, int lddo, float* dx, int lddx,
                              float* dgamma, float* dbeta, float* dtemb, int ldt, float* dbias, void* stream) {
    MI_REQUIRE(x && stats && gamma && beta && dout && dx, "null argument");
    GnArgs a{};
    int vec, units;
    int rc = gn_prepare(d, a, vec, units);
    MI_REQUIRE(rc == 0, "C/G must be a power of two <= 128 (and 1, 2 or a multiple of 4)");
    MI_REQUIRE(vec == 1 || (d->ldx % 4 == 0 && lddo % 4 == 0 && lddx % 4 == 0), "ld must be a multiple of 4");
    a.x = x; a.stats = const_cast<float*>(stats); a.gamma = gamma; a.beta = beta; a.dout = dout; a.lddo = lddo;
    a.dx = dx; a.lddx = lddx; a.dgamma = dgamma; a.dbeta = dbeta; a.dtemb = dtemb; a.ldt = ldt; a.dbias = dbias;
    hipStream_t st = (hipStream_t)stream;
    GN_DISPATCH(gn_mish_bwd_kernel);
    MI_LAUNCH_CHECK();
    return 0;
}

// ---- the same two kernels with bf16 storage of the block-internal tensors -------------------------
// fwd io: bit 0 = x (conv output) is bf16, bit 1 = y is written as bf16.  residual / temb / stats stay fp32.
extern "C" int mi_gn_mish_fwd_io(const MiGnDesc* d, const void* x, const float* gamma, const float* beta,
                                 const float* temb, int ldt, const float* residual, void* y, float* stats, int io,
                                 void* stream) {
    MI_REQUIRE(x && gamma && beta && y && !(io & ~3), "bad argument");
    GnArgs a{};
    int vec, units;
    int rc = gn_prepare(d, a, vec, units);
    MI_REQUIRE(rc == 0 && vec >= 4, "bf16 storage needs C/G to be a multiple of 4 (power of two <= 128)");
    MI_REQUIRE(d->ldx % 4 == 0 && d->ldy % 4 == 0 && (!residual || d->ldr % 4 == 0), "ld must be a multiple of 4");
    a.x = (const float*)x; a.gamma = gamma; a.beta = beta; a.temb = temb; a.ldt = ldt; a.res = residual; a.y = (float*)y; a.stats = stats;
    if (gn_vec8(a, io & 1, {d->ldx, d->ldy, residual ? d->ldr : 0, temb ? ldt : 0}, {x, y, residual, temb, gamma, beta})) vec = 8;
    hipStream_t st = (hipStream_t)stream;
    switch (io) {
        case 0: GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 0); break;
        case 1: GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 1); break;
        case 2: GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 2); break;
        default: GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 3); break;
    }
    MI_LAUNCH_CHECK();
    return 0;
}
// ... and additionally a bf16-rounded copy of y (y16, pixel stride ldy16 elements): the residual stream stays fp32 (y) while the
// next layer's MFMA operands (conv input, weight-gradient operand) are read from the copy -- the rounding the consumers would
// apply when staging is applied once, here, and they fetch half the bytes.
extern "C" int mi_gn_mish_fwd_dual(const MiGnDesc* d, const void* x, const float* gamma, const float* beta,
                                   const float* temb, int ldt, const float* residual, void* y, void* y16, int ldy16,
                                   float* stats, int io, void* stream) {
    MI_REQUIRE(x && gamma && beta && y && y16 && !(io & ~3), "bad argument");
    GnArgs a{};
    int vec, units;
    int rc = gn_prepare(d, a, vec, units);
    MI_REQUIRE(rc == 0 && vec >= 4, "the bf16 copy needs C/G to be a multiple of 4 (power of two <= 128)");
    MI_REQUIRE(d->ldx % 4 == 0 && d->ldy % 4 == 0 && ldy16 % 4 == 0 && (!residual || d->ldr % 4 == 0), "ld must be a multiple of 4");
    a.x = (const float*)x; a.gamma = gamma; a.beta = beta; a.temb = temb; a.ldt = ldt; a.res = residual; a.y = (float*)y; a.stats = stats;
    a.y16 = (uint16_t*)y16; a.ldy16 = ldy16;
    if (gn_vec8(a, io & 1, {d->ldx, d->ldy, ldy16, residual ? d->ldr : 0, temb ? ldt : 0}, {x, y, y16, residual, temb, gamma, beta})) vec = 8;
    hipStream_t st = (hipStream_t)stream;
    switch (io) {
        case 0: GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 0); break;
        case 1: GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 1); break;
        case 2: GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 2); break;
        default: GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 3); break;
    }
    MI_LAUNCH_CHECK();
    return 0;
}
// Statistics only: stats[n][g] = {mean, rstd} as above, and coef[3][N][C] = {rstd*gamma, beta - mean*rstd*gamma, temb} -- what
// mi_conv3x3_gn_mish needs to apply GroupNorm + Mish (+ time bias) while it stages its input.  One read of x, no y.
extern "C" int mi_gn_stats_coef(const MiGnDesc* d, const void* x, const float* gamma, const float* beta, const float* temb, int ldt,
                                float* stats, float* coef, int x_is_bf16, void* stream) {
    MI_REQUIRE(x && gamma && beta && coef, "null argument");
    GnArgs a{};
    int vec, units;
    int rc = gn_prepare(d, a, vec, units);
    MI_REQUIRE(rc == 0 && vec == 4 && d->ldx % 4 == 0, "C/G must be a multiple of 4 (power of two <= 128), ldx % 4 == 0");
    a.x = (const float*)x; a.gamma = gamma; a.beta = beta; a.temb = temb; a.ldt = ldt; a.stats = stats; a.coef = coef;
    a.y = coef; a.ldy = d->C;      // unused
    hipStream_t st = (hipStream_t)stream;
    if (x_is_bf16) GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 1); else GN_DISPATCH_FWD_IO(gn_mish_fwd_kernel, 0);
    MI_LAUNCH_CHECK();
    return 0;
}
// The same stats / coef from per-slab sums that the producing conv's epilogue accumulated (mi_conv3x3_bf16w_io_gnsums):
// sums [N][C / 16][2] = (sum, sum of squares) per sample and 16-channel slab.  One thread per (n, c); the group's slabs are
// combined in double (var = E[x^2] - mean^2).
__global__ __launch_bounds__(256) void gn_coef_from_sums_kernel(int N, int C, int G, int HW, float eps, const float* __restrict__ sums,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ temb, int ldt, float* __restrict__ stats,
                                                                float* __restrict__ coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C, Cg = C / G, g = c / Cg, nslab = Cg / 16;
    // (the arithmetic conv_pw's VAR 3 repeats per lane: integer slab sums, one reciprocal of the count -- bitwise the same coefficients)
    long long si = 0, qi = 0;
    bool poisoned = false;
    for (int k = 0; k < nslab; ++k) {
        const size_t p = ((size_t)n * (C / 16) + g * nslab + k) * 2;
        const long long r0 = reinterpret_cast<const long long*>(sums)[p], r1 = reinterpret_cast<const long long*>(sums)[p + 1];
        poisoned |= r0 >= (1LL << 60) || r0 <= -(1LL << 60) || r1 >= (1LL << 60) || r1 <= -(1LL << 60);
        si += r0; qi += r1;
    }
    const double icnt = 1.0 / (MI_GSUM_SCALE * (double)HW * (double)Cg);
    const double mean = poisoned ? __builtin_nan("") : (double)si * icnt;
    double var = (double)qi * icnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = 1.0f / sqrtf((float)var + eps), mf = (float)mean;
    if (stats && c == g * Cg) { stats[2 * (n * G + g)] = mf; stats[2 * (n * G + g) + 1] = rstd; }
    const float sc = gamma[c] * rstd;
    const size_t NC = (size_t)N * C;
    coef[i] = sc; coef[NC + i] = beta[c] - mf * sc; coef[2 * NC + i] = temb ? temb[(size_t)n * ldt + c] : 0.f;
}
extern "C" int mi_gn_coef_from_sums(int N, int C, int G, int HW, float eps, const float* sums, const float* gamma, const float* beta,
                                    const float* temb, int ldt, float* stats, float* coef, void* stream) {
    MI_REQUIRE(N > 0 && C > 0 && G > 0 && HW > 0 && C % G == 0 && (C / G) % 16 == 0 && sums && gamma && beta && coef,
               "bad argument (C / G must be a multiple of 16)");
    hipLaunchKernelGGL(gn_coef_from_sums_kernel, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, C, G, HW, eps, sums, gamma,
                       beta, temb, ldt, stats, coef);
    MI_LAUNCH_CHECK();
    return 0;
}
// ---- round 6: GroupNorm + Mish (+ time bias) (+ residual) as a STREAMING apply.  The statistics come from the (sum, sum of squares) the
//      producing conv's epilogue left per sample and 16-channel slab (mi_conv3x3_pw_gnsums; mi_gn_coef_from_sums' arithmetic, resolved per
//      workgroup), so nothing has to see a whole (sample, group) slice: a workgroup = a run of pixels of ONE sample x ALL channels, every
//      load of the run requested before anything is computed, no block reduction, no second phase; reads and writes of different
//      workgroups overlap by themselves.  x bf16 (the conv's output); y bf16, or fp32 with an optional bf16 copy (the residual stream).
//      Also writes stats [N][G][2] = {mean, rstd} for the backward pass (the workgroup of a sample's first run).
namespace {
struct GnApplyArgs {
    const uint16_t* x; const long long* sums; const float* gamma; const float* beta; const float* temb; const float* res;
    void* y; uint16_t* y16; float* stats;
    int N, HW, C, G, Cg, ldx, ldy, ldr, ldt, ldy16, chunks; float eps; double icnt;
};
// mish(z) + tb for a channel pair: z w / (w + 2), w = e (e + 2), e = exp(min(z, 20)) (mish_fast_f's arithmetic), two channels per instruction
__device__ __forceinline__ f32x2 mish_tb_pk(f32x2 z, f32x2 tb) {
    const f32x2 zc = {fminf(z.x, 20.f), fminf(z.y, 20.f)};
    const f32x2 kl = {1.44269504f, 1.44269504f};
    f32x2 zl, ep, w, n, t, o;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(zl) : "v"(zc), "s"(kl));
    const f32x2 e = {__builtin_amdgcn_exp2f(zl.x), __builtin_amdgcn_exp2f(zl.y)};
    // (the ONLY direct reader of e, and below of r, carries the s_nop; everything else depends on it: see mish_grad_pk)
    asm("s_nop 0\n\tv_pk_add_f32 %0, %1, 2.0 op_sel_hi:[1,0]" : "=v"(ep) : "v"(e));
    w = pk_mul(e, ep);
    asm("v_pk_add_f32 %0, %1, 2.0 op_sel_hi:[1,0]" : "=v"(n) : "v"(w));
    const f32x2 r = {__builtin_amdgcn_rcpf(n.x), __builtin_amdgcn_rcpf(n.y)};
    t = pk_mul(z, w);
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(t), "v"(r), "v"(tb));
    return o;
}
template <bool Y16, bool RES, int UNR>
__global__ __launch_bounds__(256) void gn_apply_sums_kernel(const GnApplyArgs a) {
    MI_PRIO_UP();
    __shared__ float s_stat[2 * 64];
    const int t = threadIdx.x;
    const int TPP = a.C >> 3, PP = 256 / TPP;                 // threads per pixel (8 channels each), pixels per pass
    const int n = blockIdx.x / a.chunks, chunk = blockIdx.x - n * a.chunks;
    const int u = t & (TPP - 1), pr = t / TPP, c0 = u * 8;
    const size_t pix0 = (size_t)n * a.HW + (size_t)chunk * (PP * UNR) + pr;
    // ---- every x (and residual) row of the run: requested before the statistics are even read
    uint4 xr[UNR];
    f32x4 rr[RES ? UNR : 1][2];
#pragma unroll
    for (int k = 0; k < UNR; ++k) xr[k] = *reinterpret_cast<const uint4*>(a.x + (pix0 + (size_t)k * PP) * a.ldx + c0);
    if constexpr (RES) {
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const float* rp = a.res + (pix0 + (size_t)k * PP) * a.ldr + c0;
            rr[k][0] = *reinterpret_cast<const f32x4*>(rp); rr[k][1] = *reinterpret_cast<const f32x4*>(rp + 4);
        }
    }
    // ---- the sample's statistics (G <= 64 groups): thread g combines its group's slabs -- integer sums, double, var = E[x^2] - mean^2
    if (t < a.G) {
        const int nslab = a.Cg >> 4;
        long long si = 0, qi = 0;
        bool poisoned = false;
        for (int k = 0; k < nslab; ++k) {
            const size_t p = ((size_t)n * (a.C >> 4) + (size_t)t * nslab + k) * 2;
            const long long r0 = a.sums[p], r1 = a.sums[p + 1];
            poisoned |= r0 >= (1LL << 60) || r0 <= -(1LL << 60) || r1 >= (1LL << 60) || r1 <= -(1LL << 60);
            si += r0; qi += r1;
        }
        const double mean = poisoned ? __builtin_nan("") : (double)si * a.icnt;
        double var = (double)qi * a.icnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = 1.0f / sqrtf((float)var + a.eps), mf = (float)mean;
        s_stat[2 * t] = mf; s_stat[2 * t + 1] = rstd;
        if (chunk == 0 && a.stats) { a.stats[2 * (n * a.G + t)] = mf; a.stats[2 * (n * a.G + t) + 1] = rstd; }
    }
    f32x2 ga2[4], sh2[4], tb2[4];
    {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.gamma + c0), g1 = *reinterpret_cast<const f32x4*>(a.gamma + c0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.beta + c0), b1 = *reinterpret_cast<const f32x4*>(a.beta + c0 + 4);
        f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
        if (a.temb) { const float* tp = a.temb + (size_t)n * a.ldt + c0; t0 = *reinterpret_cast<const f32x4*>(tp); t1 = *reinterpret_cast<const f32x4*>(tp + 4); }
        __syncthreads();
        const int g = c0 / a.Cg;                              // the thread's 8 channels lie in one group (Cg % 16 == 0)
        const float mean = s_stat[2 * g], rstd = s_stat[2 * g + 1];
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s0 = gv[2 * j] * rstd, s1 = gv[2 * j + 1] * rstd;
            ga2[j] = f32x2{s0, s1}; sh2[j] = f32x2{bv[2 * j] - mean * s0, bv[2 * j + 1] - mean * s1}; tb2[j] = f32x2{tv[2 * j], tv[2 * j + 1]};
        }
    }
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
        const uint32_t xv[4] = {xr[k].x, xr[k].y, xr[k].z, xr[k].w};
        f32x2 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = mish_tb_pk(pk_fma(unpack_bf16x2(xv[j]), ga2[j], sh2[j]), tb2[j]);
        if constexpr (RES) {
            o[0] = pk_add(o[0], f32x2{rr[k][0].x, rr[k][0].y}); o[1] = pk_add(o[1], f32x2{rr[k][0].z, rr[k][0].w});
            o[2] = pk_add(o[2], f32x2{rr[k][1].x, rr[k][1].y}); o[3] = pk_add(o[3], f32x2{rr[k][1].z, rr[k][1].w});
        }
        const size_t p = pix0 + (size_t)k * PP;
        const uint4 pk = make_uint4(pack_bf16(o[0].x, o[0].y), pack_bf16(o[1].x, o[1].y), pack_bf16(o[2].x, o[2].y), pack_bf16(o[3].x, o[3].y));
        if constexpr (Y16) {
            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.y) + p * a.ldy + c0) = pk;
        } else {
            float* yp = reinterpret_cast<float*>(a.y) + p * a.ldy + c0;
            *reinterpret_cast<f32x4*>(yp) = f32x4{o[0].x, o[0].y, o[1].x, o[1].y};
            *reinterpret_cast<f32x4*>(yp + 4) = f32x4{o[2].x, o[2].y, o[3].x, o[3].y};
            if (a.y16) *reinterpret_cast<uint4*>(a.y16 + p * a.ldy16 + c0) = pk;
        }
    }
}
}  // namespace
// x: bf16 [N][HW][C] (pixel stride d->ldx); sums: [N][C / 16][2] 64-bit fixed point (mi_conv3x3_pw_gnsums); y: bf16 (y_is_bf16) or fp32
// (pixel stride d->ldy), residual fp32 (optional, fp32 y only, stride d->ldr), y_bf16: optional bf16 copy of an fp32 y (stride ldy16),
// stats [N][G][2] (optional).  C / G a multiple of 16 and <= 64 groups, C / 8 a power of two <= 256, 16-byte aligned rows.
// Returns 1 when the shape is not taken (the caller keeps mi_gn_mish_fwd_io), 0 when launched.
extern "C" int mi_gn_mish_apply_sums(const MiGnDesc* d, const void* x, const void* sums, const float* gamma, const float* beta, const float* temb,
                                     int ldt, const float* residual, void* y, int y_is_bf16, void* y_bf16, int ldy16, float* stats, void* stream) {
    MI_REQUIRE(d && x && sums && gamma && beta && y, "null argument");
    MI_REQUIRE(!(y_is_bf16 && (residual || y_bf16)), "residual / bf16 copy: fp32 y only");
    const int C = d->C, G = d->G;
    if (C <= 0 || G <= 0 || G > 64 || C % G || (C / G) % 16 || C % 8 || C / 8 > 256 || ((C / 8) & (C / 8 - 1))) return 1;
    if (d->ldx % 8 || d->ldy % 8 || (residual && d->ldr % 4) || (y_bf16 && ldy16 % 8) || (temb && ldt % 4)) return 1;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)(temb ? temb : gamma) | (uintptr_t)(residual ? residual : gamma) |
          (uintptr_t)(y_bf16 ? y_bf16 : y)) & 15) || ((uintptr_t)sums & 7)) return 1;
    const int PP = 256 / (C / 8);
    int unr = residual ? 4 : 8;
    while (unr > 1 && d->HW % (PP * unr)) unr >>= 1;
    if (d->HW % (PP * unr)) return 1;
    GnApplyArgs a{};
    a.x = (const uint16_t*)x; a.sums = (const long long*)sums; a.gamma = gamma; a.beta = beta; a.temb = temb; a.res = residual; a.y = y;
    a.y16 = (uint16_t*)y_bf16; a.stats = stats; a.N = d->N; a.HW = d->HW; a.C = C; a.G = G; a.Cg = C / G; a.ldx = d->ldx; a.ldy = d->ldy;
    a.ldr = d->ldr; a.ldt = ldt; a.ldy16 = ldy16; a.eps = d->eps; a.icnt = 1.0 / (MI_GSUM_SCALE * (double)d->HW * (double)(C / G));
    a.chunks = d->HW / (PP * unr);
    const dim3 grid((unsigned)(d->N * a.chunks)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define GN_APPLY_GO(Y16, RES) do { switch (unr) { \
        case 8: hipLaunchKernelGGL((gn_apply_sums_kernel<Y16, RES, 8>), grid, blk, 0, st, a); break; \
        case 4: hipLaunchKernelGGL((gn_apply_sums_kernel<Y16, RES, 4>), grid, blk, 0, st, a); break; \
        case 2: hipLaunchKernelGGL((gn_apply_sums_kernel<Y16, RES, 2>), grid, blk, 0, st, a); break; \
        default: hipLaunchKernelGGL((gn_apply_sums_kernel<Y16, RES, 1>), grid, blk, 0, st, a); break; } } while (0)
    if (y_is_bf16) GN_APPLY_GO(true, false);
    else if (residual) GN_APPLY_GO(false, true);
    else GN_APPLY_GO(false, false);
#undef GN_APPLY_GO
    MI_LAUNCH_CHECK();
    return 0;
}
// bwd io: bit 0 = x is bf16, bit 1 = dx is written as bf16, bit 2 = dout is bf16.
extern "C" int mi_gn_mish_bwd_io(const MiGnDesc* d, const void* x, const float* stats, const float* gamma,
                                 const float* beta, const void* dout, int lddo, void* dx, int lddx,
                                 float* dgamma, float* dbeta, float* dtemb, int ldt, float* dbias, int io, void* stream) {
    MI_REQUIRE(x && stats && gamma && beta && dout && dx && !(io & ~7), "bad argument");
    GnArgs a{};
    int vec, units;
    int rc = gn_prepare(d, a, vec, units);
    MI_REQUIRE(rc == 0 && vec >= 4, "bf16 storage needs C/G to be a multiple of 4 (power of two <= 128)");
    MI_REQUIRE(d->ldx % 4 == 0 && lddo % 4 == 0 && lddx % 4 == 0, "ld must be a multiple of 4");
    a.x = (const float*)x; a.stats = const_cast<float*>(stats); a.gamma = gamma; a.beta = beta; a.dout = (const float*)dout; a.lddo = lddo;
    a.dx = (float*)dx; a.lddx = lddx; a.dgamma = dgamma; a.dbeta = dbeta; a.dtemb = dtemb; a.ldt = ldt; a.dbias = dbias;
    if (gn_vec8(a, io & 1, {d->ldx, lddo, lddx}, {x, dout, dx, gamma, beta})) vec = 8;
    hipStream_t st = (hipStream_t)stream;
    switch (io) {
        case 0: GN_DISPATCH_IO(gn_mish_bwd_kernel, 0); break;
        case 1: GN_DISPATCH_BWD16(1); break;
        case 2: GN_DISPATCH_IO(gn_mish_bwd_kernel, 2); break;
        case 3: GN_DISPATCH_BWD16(3); break;
        case 4: GN_DISPATCH_IO(gn_mish_bwd_kernel, 4); break;
        case 5: GN_DISPATCH_BWD16(5); break;
        case 6: GN_DISPATCH_IO(gn_mish_bwd_kernel, 6); break;
        default: GN_DISPATCH_BWD16(7); break;
    }
    MI_LAUNCH_CHECK();
    return 0;
}

static int ln_fwd_go(int M, int C, const float* x, int ldx, const float* g, const float* b, float eps, void* yv, int ldy, int y16, void* stream);
extern "C" int mi_chan_layernorm_fwd(int M, int C, const float* x, int ldx, const float* g, const float* b,
                                     float eps, float* y, int ldy, void* stream) {
    return ln_fwd_go(M, C, x, ldx, g, b, eps, y, ldy, 0, stream);
}
// y16 != 0: y is a bf16 tensor (ldy counts elements)
extern "C" int mi_chan_layernorm_fwd_io(int M, int C, const float* x, int ldx, const float* g, const float* b,
                                        float eps, void* y, int ldy, int y16, void* stream) {
    return ln_fwd_go(M, C, x, ldx, g, b, eps, y, ldy, y16, stream);
}
static int ln_fwd_go(int M, int C, const float* x, int ldx, const float* g, const float* b, float eps, void* yv, int ldy, int y16, void* stream) {
    float* y = (float*)yv;
    MI_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ldy % 4 == 0, "C must be a multiple of 4, <= 1024");
    MI_REQUIRE(x && g && b && y, "null argument");
    LnArgs a{};
    a.x = x; a.g = g; a.b = b; a.y = y; a.M = M; a.C = C; a.ldx = ldx; a.ldy = ldy; a.eps = eps;
    const bool half = C <= 128;                       // two pixels per wave
    int blocks = (M + (half ? 7 : 3)) / (half ? 8 : 4); if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (half) { if (y16) hipLaunchKernelGGL((chan_ln_fwd_kernel<true, 32, 1>), dim3(blocks), dim3(256), 0, st, a);
                else     hipLaunchKernelGGL((chan_ln_fwd_kernel<false, 32, 1>), dim3(blocks), dim3(256), 0, st, a); }
    else      { if (y16) hipLaunchKernelGGL((chan_ln_fwd_kernel<true, 64, 4>), dim3(blocks), dim3(256), 0, st, a);
                else     hipLaunchKernelGGL((chan_ln_fwd_kernel<false, 64, 4>), dim3(blocks), dim3(256), 0, st, a); }
    MI_LAUNCH_CHECK();
    return 0;
}

static int ln_bwd_go(int M, int C, const float* x, int ldx, const float* g, float eps, const void* dyv, int lddy, float* dx, int lddx,
                     int accumulate_dx, float* dg, float* db, int dy16, void* stream, float* part = nullptr);
static int ln_bwd_blocks(int M, int C) {
    const bool half = C <= 128;                       // two pixels per wave
    // every workgroup ends with 2*C atomics on the same C addresses: fewer, longer-running workgroups for wide layers
    static const int capenv = (int)mi_knob("MI_LN_BLOCKS", 0);
    const int cap = capenv ? capenv : 512;            // measured best of 256..2048 on the cfg-2 shapes
    const int blocks = (M + (half ? 7 : 3)) / (half ? 8 : 4);
    return blocks > cap ? cap : blocks;
}
extern "C" int mi_chan_layernorm_bwd(int M, int C, const float* x, int ldx, const float* g, float eps,
                                     const float* dy, int lddy, float* dx, int lddx, int accumulate_dx,
                                     float* dg, float* db, void* stream) {
    return ln_bwd_go(M, C, x, ldx, g, eps, dy, lddy, dx, lddx, accumulate_dx, dg, db, 0, stream);
}
// dy16 != 0: dy is a bf16 tensor (lddy counts elements)
extern "C" int mi_chan_layernorm_bwd_io(int M, int C, const float* x, int ldx, const float* g, float eps,
                                        const void* dy, int lddy, float* dx, int lddx, int accumulate_dx,
                                        float* dg, float* db, int dy16, void* stream) {
    return ln_bwd_go(M, C, x, ldx, g, eps, dy, lddy, dx, lddx, accumulate_dx, dg, db, dy16, stream);
}
// The parameter gradients as partial rows: part[r][2 C] = (dg | db) sums of workgroup r, r < mi_chan_layernorm_bwd_part_rows(M, C)
// (every row is written; nothing is added to dg / db here -- mi_rowsum_batch does that for all the LayerNorms of a step at once).
extern "C" int mi_chan_layernorm_bwd_part_rows(int M, int C) { return (M > 0 && C > 0) ? ln_bwd_blocks(M, C) : 0; }
extern "C" int mi_chan_layernorm_bwd_part(int M, int C, const float* x, int ldx, const float* g, float eps,
                                          const void* dy, int lddy, float* dx, int lddx, int accumulate_dx,
                                          float* part, int dy16, void* stream) {
    MI_REQUIRE(part, "null argument");
    return ln_bwd_go(M, C, x, ldx, g, eps, dy, lddy, dx, lddx, accumulate_dx, nullptr, nullptr, dy16, stream, part);
}
static int ln_bwd_go(int M, int C, const float* x, int ldx, const float* g, float eps, const void* dyv, int lddy, float* dx, int lddx,
                     int accumulate_dx, float* dg, float* db, int dy16, void* stream, float* part) {
    const float* dy = (const float*)dyv;
    MI_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "C must be a multiple of 4, <= 1024");
    MI_REQUIRE(x && g && dy && dx, "null argument");
    LnArgs a{};
    a.x = x; a.g = g; a.dy = dy; a.dx = dx; a.dg = dg; a.db = db; a.M = M; a.C = C; a.ldx = ldx; a.lddy = lddy;
    a.lddx = lddx; a.accumulate = accumulate_dx; a.eps = eps; a.part = part;
    const bool half = C <= 128;                       // two pixels per wave
    const int blocks = ln_bwd_blocks(M, C);
    hipStream_t st = (hipStream_t)stream;
    if (half) { if (dy16) hipLaunchKernelGGL((chan_ln_bwd_kernel<true, 32, 1>), dim3(blocks), dim3(256), 0, st, a);
                else      hipLaunchKernelGGL((chan_ln_bwd_kernel<false, 32, 1>), dim3(blocks), dim3(256), 0, st, a); }
    else      { if (dy16) hipLaunchKernelGGL((chan_ln_bwd_kernel<true, 64, 4>), dim3(blocks), dim3(256), 0, st, a);
                else      hipLaunchKernelGGL((chan_ln_bwd_kernel<false, 64, 4>), dim3(blocks), dim3(256), 0, st, a); }
    MI_LAUNCH_CHECK();
    return 0;
}
