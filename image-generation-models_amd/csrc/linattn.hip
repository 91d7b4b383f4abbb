// LinearAttention core (ddpm.py:157-165) on the exact-fp32 matrix cores
// (v_mfma_f32_32x32x2_f32): per (batch, head) with d = e = 32 channels and n = H*W pixels
//   P[p,d]   = softmax over p of k[p,d]
//   ctx[d,e] = sum_p P[p,d] v[p,e]          (32x32, K = n)
//   out[p,e] = sum_d ctx[d,e] q[p,d]        (n x 32, K = 32)
// qkv is the NHWC output of the to_qkv 1x1 conv: [b][p][q(h,d) | k(h,d) | v(h,e)].
// One 256-thread workgroup per (b, head); the four waves split the pixel axis.  These are the
// only matmuls in the path whose contraction is not a convolution (0.4 % of the FLOPs).  fp32-stored
// tensors: everything on the exact-fp32 MFMA.  bf16-stored tensors (bf16 mode; round 4): the softmax,
// its statistics and the contractions over pixels (ctx, dctx) stay exact fp32, the contractions over
// channels (out, dq, dv, dP) run on the bf16 MFMA with ctx / dctx rounded once per workgroup -- see
// tile_mm_b16.
#include "common.h"

namespace {

constexpr int DH = 32;           // dim_head, fixed by the reference (ddpm.py:147)
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct AttnArgs {
    const float* qkv; float* out; float* ctx; float* kstat;
    const float* dout; float* dqkv;
    int B, n, heads, ldq;        // ldq = 3*heads*32
    int S, per;                  // pixel slices per (batch, head) and pixels per slice (S == 1: the whole image in one workgroup)
    float* part;                 // S > 1: per-slice partial results (see the PH template parameter of the kernels)
    // FOLD (inference, S == 1): instead of out, the workgroup writes ITS 32 input columns of to_out's effective weights (see linattn_fwd_kernel)
    const uint16_t* wout; uint16_t* weff; int C;      // wout: to_out's bf16 weight rows [co][heads * 32]
};

// workgroup -> (batch, head, slice).  The heads (and slices) of one batch element share cache lines (32 channels = 64..128 bytes of a
// pixel row) and consecutive workgroup ids land on different XCDs, so ids xcd + 8*slot with slot = s + S*(h + heads*m) are
// batch element xcd + 8*m: everything of one batch element stays on one XCD / one L2.
__device__ __forceinline__ void attn_block(const AttnArgs& a, int& b, int& h, int& sl) {
    const int id = blockIdx.x;
    sl = id % a.S; h = (id / a.S) % a.heads; b = id / (a.S * a.heads);
    if (a.B % 8 == 0) {
        const int xcd = id & 7, slot = id >> 3;
        sl = slot % a.S; h = (slot / a.S) % a.heads; b = xcd + 8 * (slot / (a.S * a.heads));
    }
}

// element access for the activation tensors (qkv, out, dout, dqkv): fp32, or bf16 when T16 (offsets count elements)
template <bool T16> __device__ __forceinline__ float ldx(const float* base, size_t i) {
    if constexpr (T16) return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[i] << 16);
    else return base[i];
}
template <bool T16> __device__ __forceinline__ void stx(float* base, size_t i, float v) {
    if constexpr (T16) reinterpret_cast<uint16_t*>(base)[i] = (uint16_t)(pack_bf16(v, 0.f) & 0xffffu);
    else base[i] = v;
}
template <bool T16> __device__ __forceinline__ const float* offs(const float* base, size_t i) {
    if constexpr (T16) return reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(base) + i);
    else return base + i;
}
template <bool T16> __device__ __forceinline__ float* offs(float* base, size_t i) {
    if constexpr (T16) return reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(base) + i);
    else return base + i;
}

// acc[r] of a 32x32 MFMA tile <-> (row, col): row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
__device__ __forceinline__ int tile_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// four consecutive channels of pixel p (clamped to the last valid row by the caller) as fp32
template <bool T16> __device__ __forceinline__ f32x4v ld4(const float* base, size_t p, int ld, int c4) {
    if constexpr (T16) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + p * ld + c4);
        return f32x4v{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
    } else {
        const float4 u = *reinterpret_cast<const float4*>(base + p * ld + c4);
        return f32x4v{u.x, u.y, u.z, u.w};
    }
}

// A 32-pixel x 32-channel MFMA result (lane = channel column, 16 rows per lane) written as pixel rows: through a wave-private LDS
// tile, then 8 lanes per pixel row with 8 / 16-byte stores (the accumulator layout would store one 2-byte element per lane).
template <bool T16>
__device__ __forceinline__ void store_tile(float* dst, int ld, int p0, int pe, const f32x16& acc, float* tile) {
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[tile_row(r, l) * 33 + (l & 31)] = acc[r];
    const int c4 = (l & 7) * 4, r0 = l >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = r0 + 8 * j, p = p0 + row;
        const float* q = tile + row * 33 + c4;
        const float v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        if (p < pe) {
            if constexpr (T16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(dst) + (size_t)p * ld + c4) = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
            else *reinterpret_cast<float4*>(dst + (size_t)p * ld + c4) = make_float4(v0, v1, v2, v3);
        }
    }
}

// S[d][e] = sum_p f(A[p][d]) * Bm[p][e] over this block's pixels, waves split p, result in sm[32][33].
// A wave walks 32-pixel tiles: both operand tiles arrive with coalesced 8 / 16-byte loads (8 lanes per pixel row; the first version
// fetched one 2-byte element per lane and MFMA operand, 850 k load instructions per launch), are transformed in registers, parked
// in two wave-private LDS tiles ([32][33] floats) and read back as MFMA operands (one pixel row per lane, conflict-free).
template <bool EXP, bool T16>
__device__ __forceinline__ void reduce_outer(const float* A, const float* Bm, int ldA, int ldB, int pb, int n,
                                             const float* kmax, float* sm, float* wsum, float* scratch, float* scratch2) {
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int i = l & 31, kk = l >> 5;
    const int c4 = (l & 7) * 4, r0 = l >> 3;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4v mx = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
    if (EXP) mx = f32x4v{kmax[c4], kmax[c4 + 1], kmax[c4 + 2], kmax[c4 + 3]};
    float* tA = scratch + w * (32 * 33);
    float* tB = scratch2 + w * (32 * 33);
    for (int p0 = pb + 32 * w; p0 < n; p0 += 128) {
        f32x4v av[4], bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                     // eight unconditional loads in flight; rows past n re-read row n-1 and are zeroed
            const size_t p = (size_t)min(p0 + r0 + 8 * j, n - 1);
            av[j] = ld4<T16>(A, p, ldA, c4);
            bv[j] = ld4<T16>(Bm, p, ldB, c4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float live = p0 + r0 + 8 * j < n ? 1.f : 0.f;
            f32x4v x = av[j];
            if (EXP) {
                x = f32x4v{__expf(x.x - mx.x), __expf(x.y - mx.y), __expf(x.z - mx.z), __expf(x.w - mx.w)} * live;
                ss += x;
            } else {
                x *= live;
            }
            if constexpr (T16) {
                // bf16-stored tensors (round 4): both operands parked TRANSPOSED as bf16 -- [channel][pixel], pitch 40 -- so that the
                // fragment of the bf16 MFMA (8 consecutive pixels of one channel) is one ds_read_b128: 2 MFMAs per tile instead of
                // 16 exact-fp32 ones.  f(A) is rounded to bf16 here (its sums, the softmax denominators, are taken from the fp32 values)
                uint16_t* pa = reinterpret_cast<uint16_t*>(tA) + c4 * 40 + r0 + 8 * j;
                uint16_t* pb2 = reinterpret_cast<uint16_t*>(tB) + c4 * 40 + r0 + 8 * j;
                const uint32_t x01 = pack_bf16(x.x, x.y), x23 = pack_bf16(x.z, x.w), b01 = pack_bf16(bv[j].x, bv[j].y), b23 = pack_bf16(bv[j].z, bv[j].w);
                pa[0] = (uint16_t)x01; pa[40] = (uint16_t)(x01 >> 16); pa[80] = (uint16_t)x23; pa[120] = (uint16_t)(x23 >> 16);
                pb2[0] = (uint16_t)b01; pb2[40] = (uint16_t)(b01 >> 16); pb2[80] = (uint16_t)b23; pb2[120] = (uint16_t)(b23 >> 16);
            } else {
                float* pa = tA + (r0 + 8 * j) * 33 + c4;
                float* pb2 = tB + (r0 + 8 * j) * 33 + c4;
                pa[0] = x.x; pa[1] = x.y; pa[2] = x.z; pa[3] = x.w;
                pb2[0] = bv[j].x; pb2[1] = bv[j].y; pb2[2] = bv[j].z; pb2[3] = bv[j].w;
            }
        }
        // (wave-private LDS tiles: the wave's own ds_write -> ds_read ordering is enough)
        if constexpr (T16) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const uint16_t*>(tA) + i * 40 + 16 * s2 + 8 * kk);
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const uint16_t*>(tB) + i * 40 + 16 * s2 + 8 * kk);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int k = 2 * s2 + kk;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tA[k * 33 + i], tB[k * 33 + i], acc, 0, 0, 0);
            }
        }
    }
    float ssum = 0.f;
    if (EXP) {                                            // per-channel sums: lanes with the same l & 7 hold the same channel quad
#pragma unroll
        for (int o = 8; o <= 32; o <<= 1) {
            ss.x += __shfl_xor(ss.x, o, 64); ss.y += __shfl_xor(ss.y, o, 64); ss.z += __shfl_xor(ss.z, o, 64); ss.w += __shfl_xor(ss.w, o, 64);
        }
    }
    // combine the four waves through LDS
    float* mine = scratch + w * (32 * 33);
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[tile_row(r, l) * 33 + i] = acc[r];
    (void)ssum;
    if (EXP && l < 8) { wsum[w * 32 + c4] = ss.x; wsum[w * 32 + c4 + 1] = ss.y; wsum[w * 32 + c4 + 2] = ss.z; wsum[w * 32 + c4 + 3] = ss.w; }
    __syncthreads();
    for (int idx = t; idx < 32 * 32; idx += 256) {
        int d = idx >> 5, e = idx & 31;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) v += scratch[ww * (32 * 33) + d * 33 + e];
        sm[d * 33 + e] = v;
    }
    __syncthreads();
}

// T[p][j] = sum_k A[p][k] * Bs[k][j] for one 32-pixel tile, A from global (row stride ldA), Bs from LDS
// with element (k, j) at Bs[k*sk + j*sj].  The 32x32 A tile is fetched with coalesced 16-byte (bf16: 8-byte) loads --
// 8 lanes per pixel row -- into the wave-private LDS area `stage` ([32][33] floats), from where the MFMA operand
// column (one pixel row per lane) is read conflict-free; a row-per-lane global read would touch 32 lines per load.
template <bool T16>
__device__ __forceinline__ f32x16 tile_mm(const float* A, int ldA, int p0, int n, const float* Bs, int sk, int sj, float* stage) {
    const int l = threadIdx.x & 63, i = l & 31, kk = l >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // unconditional loads (rows past n re-read row n-1 and are zeroed by `live`): the four fetches overlap
    float v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (l >> 3) + 8 * j, c4 = (l & 7) * 4;
        const size_t p = (size_t)min(p0 + row, n - 1);
        if constexpr (T16) {
            const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(A) + p * ldA + c4);
            v[j][0] = __uint_as_float(u.x << 16); v[j][1] = __uint_as_float(u.x & 0xffff0000u);
            v[j][2] = __uint_as_float(u.y << 16); v[j][3] = __uint_as_float(u.y & 0xffff0000u);
        } else {
            const float4 u = *reinterpret_cast<const float4*>(A + p * ldA + c4);
            v[j][0] = u.x; v[j][1] = u.y; v[j][2] = u.z; v[j][3] = u.w;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (l >> 3) + 8 * j, c4 = (l & 7) * 4;
        const float live = p0 + row < n ? 1.f : 0.f;
        float* sp = stage + row * 33 + c4;
        sp[0] = v[j][0] * live; sp[1] = v[j][1] * live; sp[2] = v[j][2] * live; sp[3] = v[j][3] * live;
    }
    // (wave-private LDS region: the wave's own ds_write -> ds_read ordering is enough)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        int k = 2 * s + kk;
        float av = stage[i * 33 + k];
        float bv = Bs[k * sk + i * sj];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    return acc;
}

// bf16-stored tensors (round 4): the products whose contraction runs over CHANNELS -- out = q ctx, dq = dout ctx^T, dv = P dctx,
// dP = v dctx^T -- on v_mfma_f32_32x32x16_bf16.  The row operand of a 32-pixel tile is read straight from memory in MFMA-fragment
// order (lane (pixel i, k-half) = 16 contiguous bytes of the pixel's 64-byte head row: no LDS staging, no conversion), the 32 x 32
// matrix operand (ctx / dctx, fp32 in LDS) is rounded to bf16 once per workgroup into two fragment registers per lane.  2 MFMAs
// instead of 16 exact-fp32 ones per product and tile; the contractions over PIXELS (ctx, dctx: reduce_outer) take the bf16 MFMA too, from operands parked transposed.
// B(k, j) = Bs[k * sk + j * sj]: fragment s holds k = 16 s + 8 (lane >> 5) .. + 7 of column j = lane & 31
__device__ __forceinline__ void frag_from_lds(const float* Bs, int sk, int sj, bf16x8 (&f)[2]) {
    const int l = threadIdx.x & 63, jx = l & 31, kh = l >> 5;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            w[e] = pack_bf16(Bs[(16 * s2 + 8 * kh + 2 * e) * sk + jx * sj], Bs[(16 * s2 + 8 * kh + 2 * e + 1) * sk + jx * sj]);
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        f[s2] = __builtin_bit_cast(bf16x8, u32x4_{w[0], w[1], w[2], w[3]});
    }
}
// T[p][j] = sum_k A[p][k] B(k, j) for the 32-pixel tile at p0, A = bf16 rows in memory (rows past n - 1 repeat the last one: their
// output rows are never stored)
__device__ __forceinline__ f32x16 tile_mm_b16(const float* A, int ldA, int p0, int n, const bf16x8 (&f)[2]) {
    const int l = threadIdx.x & 63;
    const uint16_t* row = reinterpret_cast<const uint16_t*>(A) + (size_t)min(p0 + (l & 31), n - 1) * ldA + 8 * (l >> 5);
    const uint4 a0 = *reinterpret_cast<const uint4*>(row), a1 = *reinterpret_cast<const uint4*>(row + 16);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), f[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), f[1], acc, 0, 0, 0);
    return acc;
}
// ... with the row operand taken from a wave-private fp32 LDS tile [pixel][33] (P in the backward)
__device__ __forceinline__ f32x16 tile_mm_b16_lds(const float* At, const bf16x8 (&f)[2]) {
    const int l = threadIdx.x & 63, i = l & 31, kh = l >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf16(At[i * 33 + 16 * s2 + 8 * kh + 2 * e], At[i * 33 + 16 * s2 + 8 * kh + 2 * e + 1]);
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, u32x4_{w[0], w[1], w[2], w[3]}), f[s2], acc, 0, 0, 0);
    }
    return acc;
}

// PH = 0: the whole image of one (batch, head) in one workgroup (S == 1).  With S > 1 the pixel axis is cut into slices and the
// three dependent steps become three launches over (batch, head, slice):  1: column max of k over the slice -> part_max;
// 2: max over the slices, then exp / outer product over the slice -> part_ctx, part_sum;  3: slices summed in a fixed order
// (deterministic), normalised, written to ctx / kstat by slice 0, then out for the slice's pixels.  One image per workgroup keeps
// 4096 pixels behind 8 KB of loads in flight; sliced, every CU holds several workgroups and the loads of all of them.
// FOLD (round 6, inference, PH == 0): out = ctx^T q is linear in q and to_out is a 1x1 conv over out's channels, so
//   to_out(out)[co][p] = sum_{h,d} ( sum_e W_out[co][h*32 + e] ctx_h[d][e] ) q[h*32 + d][p]
// -- a 1x1 conv of q with PER-SAMPLE weights W_eff[b] = W_out blockdiag(ctx_h^T).  The workgroup (b, h) writes its 32 input columns of W_eff[b]
// (C x 32 bf16, in conv1x1_pw_kernel's fragment order [co / 32][k / 16][lane][8]) instead of running the third phase: q is not read here, out is
// never written; mi_conv1x1_pw_batched then reads q straight out of qkv.
template <bool T16, int PH, bool FOLD = false>     // T16: qkv and out are stored as bf16
__global__ __launch_bounds__(256) void linattn_fwd_kernel(const AttnArgs a) {
    static_assert(!FOLD || (T16 && PH == 0), "the fold: bf16 storage, one workgroup per (batch, head)");
    MI_PRIO_UP();
    __shared__ float scratch[4 * 32 * 33], scratch2[PH == 1 ? 1 : 4 * 32 * 33];
    __shared__ float ctx_s[32 * 33];
    __shared__ float kmax_s[32], ksum_s[32], wsum[4 * 32], pmax[32 * 32];
    int b, h, sl;
    attn_block(a, b, h, sl);
    const int bh = b * a.heads + h;
    const int pb = sl * a.per, pe = min(a.n, pb + a.per);
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int hid = a.heads * DH;
    const float* q = offs<T16>(a.qkv, (size_t)b * a.n * a.ldq + h * DH);
    const float* k = offs<T16>(q, hid);
    const float* v = offs<T16>(q, 2 * hid);
    const int BHS = a.B * a.heads * a.S;
    float* part_max = a.part;                         // [BH][S][32]
    float* part_sum = a.part + (size_t)BHS * 32;      // [BH][S][32]
    float* part_ctx = a.part + (size_t)BHS * 64;      // [BH][S][32*32]
    const size_t bhs = (size_t)bh * a.S + sl;

    if constexpr (PH == 0 || PH == 1) {               // column max of k over the slice's pixels
        // thread = (row t >> 3 of a 32-row sweep, channel quad t & 7): 8 / 16-byte loads, four sweeps in flight; rows past the end
        // repeat the last row (max-neutral)
        const int c4 = (t & 7) * 4, pr = t >> 3;
        f32x4v m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int p = pb + pr; p < pe; p += 128) {
            f32x4v v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld4<T16>(k, (size_t)min(p + 32 * u, pe - 1), a.ldq, c4);
#pragma unroll
            for (int u = 0; u < 4; ++u) m = f32x4v{fmaxf(m.x, v[u].x), fmaxf(m.y, v[u].y), fmaxf(m.z, v[u].z), fmaxf(m.w, v[u].w)};
        }
        pmax[pr * 32 + c4] = m.x; pmax[pr * 32 + c4 + 1] = m.y; pmax[pr * 32 + c4 + 2] = m.z; pmax[pr * 32 + c4 + 3] = m.w;
        __syncthreads();
        if (t < 32) {
            float mm = pmax[t];
#pragma unroll
            for (int r = 1; r < 32; ++r) mm = fmaxf(mm, pmax[r * 32 + t]);
            kmax_s[t] = mm;
            if constexpr (PH == 1) part_max[bhs * 32 + t] = mm;
        }
        if constexpr (PH == 1) return;
        __syncthreads();
    } else {                                          // max over the slices
        if (t < 32) {
            float mm = part_max[(size_t)bh * a.S * 32 + t];
            for (int s2 = 1; s2 < a.S; ++s2) mm = fmaxf(mm, part_max[((size_t)bh * a.S + s2) * 32 + t]);
            kmax_s[t] = mm;
        }
        __syncthreads();
    }
    if constexpr (PH == 0 || PH == 2) {
        reduce_outer<true, T16>(k, v, a.ldq, a.ldq, pb, pe, kmax_s, ctx_s, wsum, scratch, scratch2);
        if (t < 32) ksum_s[t] = wsum[t] + wsum[32 + t] + wsum[64 + t] + wsum[96 + t];
        __syncthreads();
        if constexpr (PH == 2) {
            for (int idx = t; idx < 32 * 32; idx += 256) part_ctx[bhs * 1024 + idx] = ctx_s[(idx >> 5) * 33 + (idx & 31)];
            if (t < 32) part_sum[bhs * 32 + t] = ksum_s[t];
            return;
        }
    } else {                                          // PH == 3: slices in a fixed order
        for (int idx = t; idx < 32 * 32; idx += 256) {
            float c = 0.f;
            for (int s2 = 0; s2 < a.S; ++s2) c += part_ctx[((size_t)bh * a.S + s2) * 1024 + idx];
            ctx_s[(idx >> 5) * 33 + (idx & 31)] = c;
        }
        if (t < 32) {
            float c = 0.f;
            for (int s2 = 0; s2 < a.S; ++s2) c += part_sum[((size_t)bh * a.S + s2) * 32 + t];
            ksum_s[t] = c;
        }
        __syncthreads();
    }
    float* ctx_g = FOLD ? nullptr : a.ctx + (size_t)bh * 32 * 32;
    for (int idx = t; idx < 32 * 32; idx += 256) {
        int d = idx >> 5, e = idx & 31;
        float c = ctx_s[d * 33 + e] / ksum_s[d];
        ctx_s[d * 33 + e] = c;
        if constexpr (!FOLD) { if (sl == 0) ctx_g[idx] = c; }
    }
    if constexpr (!FOLD) {
        if (t < 32 && sl == 0) {
            float* ks = a.kstat + (size_t)bh * 64;
            ks[2 * t] = kmax_s[t]; ks[2 * t + 1] = ksum_s[t];
        }
    }
    __syncthreads();
    if constexpr (FOLD) {
        // W_eff[co][h*32 + d] = sum_e W_out[co][h*32 + e] ctx[d][e]: 32-channel tiles of to_out's bf16 weight rows [co][hidden] times ctx^T on the bf16
        // MFMA (two per tile), through the wave's LDS tile into conv1x1_pw_kernel's fragment order: 16 bytes = 8 consecutive k of one co
        bf16x8 fT[2];
        frag_from_lds(ctx_s, 1, 33, fT);                      // B(k = e, j = d) = ctx[d][e]
        const int KQ = hid / 16;
        const float* wrow = reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(a.wout) + h * DH);
        uint16_t* wb = a.weff + (size_t)b * a.C * hid;
        float* tile = scratch + w * (32 * 33);
        for (int co0 = 32 * w; co0 < a.C; co0 += 128) {
            const f32x16 acc = tile_mm_b16(wrow, hid, co0, a.C, fT);
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[tile_row(r, l) * 33 + (l & 31)] = acc[r];
            // (wave-private LDS region: the wave's own ds_write -> ds_read ordering is enough)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int dg = (l >> 5) + 2 * i, cl = l & 31;
                const float* q8 = tile + cl * 33 + dg * 8;
                typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
                *reinterpret_cast<u32x4w*>(wb + ((size_t)((co0 >> 5) * KQ + 2 * h + (dg >> 1)) * 64 + cl + 32 * (dg & 1)) * 8) =
                    u32x4w{pack_bf16(q8[0], q8[1]), pack_bf16(q8[2], q8[3]), pack_bf16(q8[4], q8[5]), pack_bf16(q8[6], q8[7])};
            }
        }
        return;
    }
    // out[p][e] = sum_d q[p][d] ctx[d][e]
    float* o = offs<T16>(a.out, (size_t)b * a.n * hid + h * DH);
    bf16x8 fctx[2];
    if constexpr (T16) frag_from_lds(ctx_s, 33, 1, fctx);                // B(k = d, j = e) = ctx[d][e]
    for (int p0 = pb + 32 * w; p0 < pe; p0 += 128) {
        f32x16 acc;
        if constexpr (T16) acc = tile_mm_b16(q, a.ldq, p0, pe, fctx);
        else acc = tile_mm<T16>(q, a.ldq, p0, pe, ctx_s, 33, 1, scratch + w * (32 * 33));
        store_tile<T16>(o, hid, p0, pe, acc, scratch + w * (32 * 33));
    }
}

// Backward.  With P = softmax_p(k), ctx = P^T v (saved), r[d] = sum_e ctx[d,e] dctx[d,e]:
//   dctx = q^T dout ; dq = dout ctx^T ; dv = P dctx ; dP = v dctx^T ; dk = P * (dP - r)
// PH = 0: one workgroup per (batch, head).  S > 1:  1: dctx over the slice -> part;  2: slices summed in a fixed order, then the
// per-pixel gradients of the slice.
template <bool T16, int PH>     // T16: qkv, dout and dqkv are stored as bf16
__global__ __launch_bounds__(256) void linattn_bwd_kernel(const AttnArgs a) {
    MI_PRIO_UP();
    __shared__ float scratch[4 * 32 * 33];
    __shared__ float ctx_s[32 * 33], dctx_s[32 * 33];
    __shared__ float stage_s[4 * 32 * 33];
    __shared__ float kmax_s[32], kinv_s[32], r_s[32];
    int b, h, sl;
    attn_block(a, b, h, sl);
    const int bh = b * a.heads + h;
    const int pb = sl * a.per, pe = min(a.n, pb + a.per);
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int hid = a.heads * DH;
    const float* q = offs<T16>(a.qkv, (size_t)b * a.n * a.ldq + h * DH);
    const float* k = offs<T16>(q, hid);
    const float* v = offs<T16>(q, 2 * hid);
    const float* dout = offs<T16>(a.dout, (size_t)b * a.n * hid + h * DH);
    float* dq = offs<T16>(a.dqkv, (size_t)b * a.n * a.ldq + h * DH);
    float* dk = offs<T16>(dq, hid);
    float* dv = offs<T16>(dq, 2 * hid);
    const size_t bhs = (size_t)bh * a.S + sl;

    if constexpr (PH != 1) {
        const float* ctx_g = a.ctx + (size_t)bh * 32 * 32;
        for (int idx = t; idx < 32 * 32; idx += 256) ctx_s[(idx >> 5) * 33 + (idx & 31)] = ctx_g[idx];
        if (t < 32) {
            const float* ks = a.kstat + (size_t)bh * 64;
            kmax_s[t] = ks[2 * t]; kinv_s[t] = 1.0f / ks[2 * t + 1];
        }
    }
    if constexpr (PH == 0 || PH == 1) {
        reduce_outer<false, T16>(q, dout, a.ldq, hid, pb, pe, nullptr, dctx_s, nullptr, scratch, stage_s);   // syncs inside
        if constexpr (PH == 1) {
            for (int idx = t; idx < 32 * 32; idx += 256) a.part[bhs * 1024 + idx] = dctx_s[(idx >> 5) * 33 + (idx & 31)];
            return;
        }
    } else {
        for (int idx = t; idx < 32 * 32; idx += 256) {
            float c = 0.f;
            for (int s2 = 0; s2 < a.S; ++s2) c += a.part[((size_t)bh * a.S + s2) * 1024 + idx];
            dctx_s[(idx >> 5) * 33 + (idx & 31)] = c;
        }
        __syncthreads();
    }
    if (t < 32) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) s += ctx_s[t * 33 + e] * dctx_s[t * 33 + e];
        r_s[t] = s;
    }
    __syncthreads();

    float* pt = scratch + w * (32 * 33);        // this wave's P tile [pixel][d]
    float* stg = stage_s + w * (32 * 33);       // this wave's operand staging tile
    bf16x8 fctxT[2], fdctx[2], fdctxT[2];
    if constexpr (T16) {
        frag_from_lds(ctx_s, 1, 33, fctxT);                              // B(k = e, j = d) = ctx[d][e]
        frag_from_lds(dctx_s, 33, 1, fdctx);                             // B(k = d, j = e) = dctx[d][e]
        frag_from_lds(dctx_s, 1, 33, fdctxT);                            // B(k = e, j = d) = dctx[d][e]
    }
    for (int p0 = pb + 32 * w; p0 < pe; p0 += 128) {
        const int col = l & 31;
        // dq[p][d] = sum_e dout[p][e] ctx[d][e]      (B(k=e, j=d) = ctx_s[d*33+e])
        f32x16 acc;
        if constexpr (T16) acc = tile_mm_b16(dout, hid, p0, pe, fctxT);
        else acc = tile_mm<T16>(dout, hid, p0, pe, ctx_s, 1, 33, stg);
        store_tile<T16>(dq, a.ldq, p0, pe, acc, stg);
        // P tile into LDS (rows = pixels) so it can serve as the A operand of dv
        {
            const int c4 = (l & 7) * 4, r0 = l >> 3;
            f32x4v kv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) kv[j] = ld4<T16>(k, (size_t)min(p0 + r0 + 8 * j, pe - 1), a.ldq, c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float live = p0 + r0 + 8 * j < pe ? 1.f : 0.f;
                float* pp = pt + (r0 + 8 * j) * 33 + c4;
#pragma unroll
                for (int e = 0; e < 4; ++e) pp[e] = __expf(kv[j][e] - kmax_s[c4 + e]) * kinv_s[c4 + e] * live;
            }
        }
        // (wave-private LDS region: the wave's own ds_write -> ds_read ordering is enough)
        // dv[p][e] = sum_d P[p][d] dctx[d][e]
        {
            const int i = l & 31, kk = l >> 5;
            f32x16 a2;
            if constexpr (T16) a2 = tile_mm_b16_lds(pt, fdctx);
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) a2[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    int kd = 2 * s + kk;
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(pt[i * 33 + kd], dctx_s[kd * 33 + i], a2, 0, 0, 0);
                }
            }
            store_tile<T16>(dv, a.ldq, p0, pe, a2, stg);
        }
        // dP[p][d] = sum_e v[p][e] dctx[d][e]        (B(k=e, j=d) = dctx_s[d*33+e]) ; dk = P*(dP - r)
        if constexpr (T16) acc = tile_mm_b16(v, a.ldq, p0, pe, fdctxT);
        else acc = tile_mm<T16>(v, a.ldq, p0, pe, dctx_s, 1, 33, stg);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = pt[tile_row(r, l) * 33 + col] * (acc[r] - r_s[col]);
        store_tile<T16>(dk, a.ldq, p0, pe, acc, stg);
    }
}

// slices per (batch, head): only as many as it takes to give every CU a workgroup (measured on MI355X: with 512 (batch, head)
// pairs -- cfg 2 at B = 128 -- slicing further changes nothing, the kernels are bound by their VALU + fp32-MFMA issue, not by
// loads in flight: forward 0.128 -> 0.153 ms per step with 8 slices; with 256 pairs -- the sampler at B = 64 -- two slices cost 2 %
// of a denoise step), slices of at least 128 pixels (one 32-pixel tile per wave)
int attn_slices(int B, int n, int heads) {
    static const int force = (int)mi_knob("MI_ATTN_SLICES", 0);
    if (force > 0) return (n / force >= 32) ? force : 1;
    int S = 1;
    while (S < 32 && (long)B * heads * S < 256 && n / (2 * S) >= 128) S *= 2;
    return S;
}

}  // namespace

static void attn_plan(AttnArgs& a, int S, void* workspace) {
    a.S = S;
    a.per = ((a.n + S - 1) / S + 31) / 32 * 32;
    a.part = (float*)workspace;
}

// scratch bytes the sliced launches need (0: one workgroup per (batch, head), no scratch)
extern "C" size_t mi_linattn_workspace(int B, int n, int heads) {
    if (B <= 0 || n <= 0 || heads <= 0) return 0;
    const int S = attn_slices(B, n, heads);
    return S > 1 ? (size_t)B * heads * S * (1024 + 64) * sizeof(float) : 0;
}

static int linattn_fwd_go(int B, int n, int heads, const void* qkv, void* out, float* ctx, float* kstat, int b16, void* workspace,
                          size_t ws_bytes, void* stream) {
    MI_REQUIRE(B > 0 && n > 0 && heads > 0 && qkv && out && ctx && kstat, "bad argument");
    AttnArgs a{};
    a.qkv = (const float*)qkv; a.out = (float*)out; a.ctx = ctx; a.kstat = kstat; a.B = B; a.n = n; a.heads = heads; a.ldq = 3 * heads * DH;
    int S = attn_slices(B, n, heads);
    if (S > 1 && (!workspace || ws_bytes < mi_linattn_workspace(B, n, heads) || ((uintptr_t)workspace & 15))) S = 1;
    attn_plan(a, S, workspace);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B * heads * S);
    if (S == 1) {
        if (b16) hipLaunchKernelGGL((linattn_fwd_kernel<true, 0>), grid, dim3(256), 0, st, a);
        else     hipLaunchKernelGGL((linattn_fwd_kernel<false, 0>), grid, dim3(256), 0, st, a);
    } else if (b16) {
        hipLaunchKernelGGL((linattn_fwd_kernel<true, 1>), grid, dim3(256), 0, st, a);
        hipLaunchKernelGGL((linattn_fwd_kernel<true, 2>), grid, dim3(256), 0, st, a);
        hipLaunchKernelGGL((linattn_fwd_kernel<true, 3>), grid, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((linattn_fwd_kernel<false, 1>), grid, dim3(256), 0, st, a);
        hipLaunchKernelGGL((linattn_fwd_kernel<false, 2>), grid, dim3(256), 0, st, a);
        hipLaunchKernelGGL((linattn_fwd_kernel<false, 3>), grid, dim3(256), 0, st, a);
    }
    MI_LAUNCH_CHECK();
    return 0;
}
static int linattn_bwd_go(int B, int n, int heads, const void* qkv, const float* ctx, const float* kstat,
                          const void* dout, void* dqkv, int b16, void* workspace, size_t ws_bytes, void* stream) {
    MI_REQUIRE(B > 0 && n > 0 && heads > 0 && qkv && ctx && kstat && dout && dqkv, "bad argument");
    AttnArgs a{};
    a.qkv = (const float*)qkv; a.ctx = const_cast<float*>(ctx); a.kstat = const_cast<float*>(kstat); a.dout = (const float*)dout; a.dqkv = (float*)dqkv;
    a.B = B; a.n = n; a.heads = heads; a.ldq = 3 * heads * DH;
    int S = attn_slices(B, n, heads);
    if (S > 1 && (!workspace || ws_bytes < mi_linattn_workspace(B, n, heads) || ((uintptr_t)workspace & 15))) S = 1;
    attn_plan(a, S, workspace);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B * heads * S);
    if (S == 1) {
        if (b16) hipLaunchKernelGGL((linattn_bwd_kernel<true, 0>), grid, dim3(256), 0, st, a);
        else     hipLaunchKernelGGL((linattn_bwd_kernel<false, 0>), grid, dim3(256), 0, st, a);
    } else if (b16) {
        hipLaunchKernelGGL((linattn_bwd_kernel<true, 1>), grid, dim3(256), 0, st, a);
        hipLaunchKernelGGL((linattn_bwd_kernel<true, 2>), grid, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((linattn_bwd_kernel<false, 1>), grid, dim3(256), 0, st, a);
        hipLaunchKernelGGL((linattn_bwd_kernel<false, 2>), grid, dim3(256), 0, st, a);
    }
    MI_LAUNCH_CHECK();
    return 0;
}

// Inference (see linattn_fwd_kernel's FOLD): qkv bf16 [B][n][3 * heads * 32], w_out_bf16 = to_out's weights as bf16 rows [C][heads * 32] (the
// reference's Conv2d(hidden, dim, 1), ddpm.py:152; mi_pack_weights_bf16's plain forward copy wf) -> weff bf16 [B][C * heads * 32]: per-sample
// weights of the 1x1 conv mi_conv1x1_pw_batched runs on q, in fragment order.  C % 32 == 0.
extern "C" int mi_linattn_fold_fwd(int B, int n, int heads, const void* qkv_bf16, const void* w_out_bf16, int C, void* weff_bf16, void* stream) {
    MI_REQUIRE(B > 0 && n > 0 && heads > 0 && (heads * DH) % 16 == 0 && qkv_bf16 && w_out_bf16 && weff_bf16 && C > 0 && C % 32 == 0 &&
               (((uintptr_t)weff_bf16 | (uintptr_t)w_out_bf16) & 15) == 0, "bad argument (C % 32 == 0, 16-byte aligned weights)");
    AttnArgs a{};
    a.qkv = (const float*)qkv_bf16; a.B = B; a.n = n; a.heads = heads; a.ldq = 3 * heads * DH;
    a.wout = (const uint16_t*)w_out_bf16; a.weff = (uint16_t*)weff_bf16; a.C = C;
    attn_plan(a, 1, nullptr);
    hipLaunchKernelGGL((linattn_fwd_kernel<true, 0, true>), dim3(B * heads), dim3(256), 0, (hipStream_t)stream, a);
    MI_LAUNCH_CHECK();
    return 0;
}
extern "C" int mi_linattn_fwd(int B, int n, int heads, const float* qkv, float* out, float* ctx, float* kstat,
                              void* stream) {
    return linattn_fwd_go(B, n, heads, qkv, out, ctx, kstat, 0, nullptr, 0, stream);
}
extern "C" int mi_linattn_bwd(int B, int n, int heads, const float* qkv, const float* ctx, const float* kstat,
                              const float* dout, float* dqkv, void* stream) {
    return linattn_bwd_go(B, n, heads, qkv, ctx, kstat, dout, dqkv, 0, nullptr, 0, stream);
}
// bf16 storage of the attention-internal activations: b16 != 0 -> qkv, out (forward) and qkv, dout, dqkv (backward)
// are bf16 tensors; ctx / kstat and all arithmetic stay fp32.
extern "C" int mi_linattn_fwd_io(int B, int n, int heads, const void* qkv, void* out, float* ctx, float* kstat, int b16,
                                 void* stream) {
    return linattn_fwd_go(B, n, heads, qkv, out, ctx, kstat, b16, nullptr, 0, stream);
}
extern "C" int mi_linattn_bwd_io(int B, int n, int heads, const void* qkv, const float* ctx, const float* kstat,
                                 const void* dout, void* dqkv, int b16, void* stream) {
    return linattn_bwd_go(B, n, heads, qkv, ctx, kstat, dout, dqkv, b16, nullptr, 0, stream);
}
// ... with a scratch buffer (mi_linattn_workspace bytes, 16-byte aligned): large images are cut into pixel slices, one workgroup
// each, so that every CU holds several workgroups' loads in flight; partial results are combined in a fixed order (deterministic).
// A null / too small workspace runs one workgroup per (batch, head) as above.
extern "C" int mi_linattn_fwd_ws(int B, int n, int heads, const void* qkv, void* out, float* ctx, float* kstat, int b16,
                                 void* workspace, size_t ws_bytes, void* stream) {
    return linattn_fwd_go(B, n, heads, qkv, out, ctx, kstat, b16, workspace, ws_bytes, stream);
}
extern "C" int mi_linattn_bwd_ws(int B, int n, int heads, const void* qkv, const float* ctx, const float* kstat,
                                 const void* dout, void* dqkv, int b16, void* workspace, size_t ws_bytes, void* stream) {
    return linattn_bwd_go(B, n, heads, qkv, ctx, kstat, dout, dqkv, b16, workspace, ws_bytes, stream);
}
