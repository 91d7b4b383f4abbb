// Weight gradient of the 3x3 / stride 1 / pad 1 convolution on bf16 MFMA:
//     dW[ky][kx][ci][co] += sum_{n,y,x} X[n, y+ky-1, x+kx-1, ci] * dY[n, y, x, co]
// The contraction axis (pixels x images) is the strided axis of both NHWC operands.  Instead of
// transposing pixel runs (which a tap shift would misalign), the 8 consecutive k-elements of an
// MFMA operand are the SAME pixel of 8 CONSECUTIVE IMAGES: LDS holds Xs[ci][position][8 images]
// (16 bytes per slot), so a tap shift moves whole 16-byte slots and every fragment read is an
// aligned ds_read_b128.  One workgroup = one (ci-tile, co-tile, ky, k-slice); per chunk
// (8 images x 16 pixels) the X row tile (+1 halo column each side) and the dY tile are staged
// once (fp32 -> bf16 via v_cvt_pk_bf16_f32 across images) and feed the three kx taps, whose
// accumulators all live in registers.  k-slices are combined with fp32 atomics.
#include <stdlib.h>
#include "common.h"

#ifndef MI_W3_ABL
#define MI_W3_ABL 0      // profiling only: 1 no MFMA, 2 no global loads, 4 no LDS commit (cvt + ds_write), 8 no fragment reads
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// compile-time loop: f(integral_constant<int, I>) for I in [0, N)
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

struct W3Args {
    const float* P; const float* P2; const float* Q; float* dW;
    int N, H, W, Ci, Cj, I1, ldp, ldp2, ldq;
    int TH, TW, tw_sh;     // spatial tile (TH*TW == 16), TW = 1 << tw_sh
    int tiles_x, tiles;    // W/TW, tiles per image
    int total, cps, splits;
    int gx, gy;
    int xcd_map;         // workgroup id -> (k-slice, tile) such that the tiles of one k-slice share an XCD (one L2)
    float* dbias;        // optional: dbias[co] += sum over pixels of Q (the conv's bias gradient), fused into the dY staging
    float* ws;           // per-workgroup partial tiles in register order (null -> fp32 atomics into dW)
};

// IO bit 0: P (layer input) is stored as bf16, bit 1: Q (output gradient) is stored as bf16.
template <int NJ, int KS, int IO = 0>
__global__ __launch_bounds__(256, 1) void wgrad3x3_kernel(const W3Args a) {
    constexpr bool P16 = IO & 1, Q16 = IO & 2;
    constexpr int BI = 128, BJ = 32 * 2 * NJ;
    constexpr int YP = 17 * 8;                          // dY pitch per co row (bf16 elems): 16 slots + 1 pad
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];      // 2 x (BI*XP + BJ*YP) bf16

    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int wi = wv >> 1, wj = wv & 1;
    // 1-D grid, one workgroup per CU (the kernel runs one wave per SIMD): b -> (k-slice, ky, ci-tile, co-tile)
    const int gsz = a.gx * a.gy * KS;
    // The gsz workgroups of one k-slice (3 ky x tiles) read the same X / dY rows.  Consecutive workgroup ids go to
    // different XCDs, so with xcd_map ids xcd + 8*slot, slot = within + gsz*m, are k-slice xcd + 8*m: one L2 serves them.
    int split = blockIdx.x / gsz, within = blockIdx.x % gsz;
    if (a.xcd_map) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        within = slot % gsz; split = xcd + 8 * (slot / gsz);
        if (split >= a.splits) return;
    }
    const int ky = within % KS, txy = within / KS;
    const int ci0 = (txy % a.gx) * BI, co0 = (txy / a.gx) * BJ;
    constexpr int NUX = KS == 3 ? 3 : 2;                // X staging units per wave (= position groups of 8)
    const int TW2 = a.TW + (KS - 1);
    const int PX = a.TH * TW2;                          // X positions per chunk (18, 20 or 24; 16 for 1x1)
    const int XP = (PX | 1) * 8;                        // odd slot count -> conflict-free fragment reads; slot PX is a pad
    const int HW = a.H * a.W;

    f32x16 acc[KS][2][NJ];
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[k][i][j][r] = 0.f;

    const int hp_sub = l & 7, c4 = l >> 3;
    const int cbeg = split * a.cps, cend = min(a.total, cbeg + a.cps);     // non-empty by construction of `splits`

    // ---- staging assignment (fixed per thread): NUX X units and NJ dY units per wave;
    //      unit = (8 positions) x (32 channels), one 16-byte load per image per lane.  Everything that does not
    //      depend on the chunk is resolved here so that the loop issues loads without any control flow.
    const char* x_base[NUX]; int x_ld[NUX], x_r[NUX], x_xx[NUX], x_dst[NUX]; bool x_in[NUX];
#pragma unroll
    for (int k = 0; k < NUX; ++k) {
        const int u = wv + 4 * k;
        const int pg = u % NUX, cg = u / NUX;
        const int pos = pg * 8 + hp_sub;
        x_in[k] = pos < PX;
        x_r[k] = pos / TW2; x_xx[k] = pos - x_r[k] * TW2;
        int ch = min(ci0 + cg * 32 + c4 * 4, a.Ci - 4);               // rows past Ci are never written back
        const bool second = ch >= a.I1;
        x_ld[k] = second ? a.ldp2 : a.ldp;
        x_base[k] = reinterpret_cast<const char*>(second ? a.P2 : a.P) + (size_t)(second ? ch - a.I1 : ch) * (P16 ? 2 : 4);
        x_dst[k] = (cg * 32 + c4 * 4) * XP + (x_in[k] ? pos : PX) * 8;  // positions past the tile land in the pad slot
    }
    const char* y_base[NJ]; int y_pix[NJ], y_dst[NJ], y_ch[NJ];
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        const int u = wv + 4 * k;
        const int pg = u & 1, cg = u >> 1;
        const int pos = pg * 8 + hp_sub;
        y_pix[k] = (pos >> a.tw_sh) * a.W + (pos & (a.TW - 1));
        y_ch[k] = co0 + cg * 32 + c4 * 4;
        y_base[k] = reinterpret_cast<const char*>(a.Q) + (size_t)min(y_ch[k], a.Cj - 4) * (Q16 ? 2 : 4);
        y_dst[k] = (cg * 32 + c4 * 4) * YP + pos * 8;
    }
    // Pipeline: LDS is double-buffered by chunk.  While the 8 k-steps of chunk c run on the matrix
    // cores out of buffer c&1, the wave's NUX+NJ staging units of chunk c+1 (fetched during chunk c-1) are
    // converted and written to the other buffer, one unit per k-step, and the same registers are immediately
    // re-armed with the unit of chunk c+2: every global load has a full chunk (~100 MFMAs) to land.  One
    // barrier per chunk.  The loop is straight-line code (clamped addresses, padding as an AND mask at the
    // LDS store, last chunk peeled) so that the compiler can wait with exact vmcnt counts.
    constexpr int NU = NUX + NJ;
    f32x4 U[NU][8];                                    // one register set per staging unit: a whole chunk in flight
    uint32_t keep[NUX];                                // all-ones when the unit's position is inside the image
    const int BUF = BI * XP + BJ * YP;                 // bf16 elements per LDS buffer

    auto issue_unit = [&](auto jc, int c) {            // unit j of chunk c: global -> registers
        constexpr int j = decltype(jc)::value;
        const int g = c / a.tiles, tile = c - g * a.tiles;
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int y0 = ty * a.TH, x0 = tx * a.TW, n0 = g * 8;
        if constexpr (j < NUX) {
            const int iy = y0 + x_r[j] + ky - KS / 2, ix = x0 + x_xx[j] - KS / 2;
            const bool ok = x_in[j] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            keep[j] = ok ? ~0u : 0u;
            const size_t img = (size_t)HW * x_ld[j] * (P16 ? 2 : 4);
            const char* src = x_base[j] + ((size_t)n0 * HW + (ok ? iy * a.W + ix : 0)) * x_ld[j] * (P16 ? 2 : 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if constexpr (P16) {      // 4 bf16 channels = 8 bytes, carried in .x/.y
                    const u32x2 u = *reinterpret_cast<const u32x2*>(src + q * img);
                    U[j][q] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
                } else {
                    U[j][q] = *reinterpret_cast<const f32x4*>(src + q * img);
                }
            }
        } else {
            constexpr int k = j - NUX;
            const size_t img = (size_t)HW * a.ldq * (Q16 ? 2 : 4);
            const char* src = y_base[k] + ((size_t)n0 * HW + y0 * a.W + x0 + y_pix[k]) * a.ldq * (Q16 ? 2 : 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if constexpr (Q16) {
                    const u32x2 u = *reinterpret_cast<const u32x2*>(src + q * img);
                    U[j][q] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
                } else {
                    U[j][q] = *reinterpret_cast<const f32x4*>(src + q * img);
                }
            }
        }
    };
    // 8 images x 4 channels -> 4 rows of 8 bf16 (ANDed with the padding mask)
    auto put = [&](uint16_t* dst, int pitch, const f32x4 (&v)[8], uint32_t m) {
        *reinterpret_cast<u32x4*>(dst)             = u32x4{pack_bf16(v[0].x, v[1].x), pack_bf16(v[2].x, v[3].x), pack_bf16(v[4].x, v[5].x), pack_bf16(v[6].x, v[7].x)} & m;
        *reinterpret_cast<u32x4*>(dst + pitch)     = u32x4{pack_bf16(v[0].y, v[1].y), pack_bf16(v[2].y, v[3].y), pack_bf16(v[4].y, v[5].y), pack_bf16(v[6].y, v[7].y)} & m;
        *reinterpret_cast<u32x4*>(dst + 2 * pitch) = u32x4{pack_bf16(v[0].z, v[1].z), pack_bf16(v[2].z, v[3].z), pack_bf16(v[4].z, v[5].z), pack_bf16(v[6].z, v[7].z)} & m;
        *reinterpret_cast<u32x4*>(dst + 3 * pitch) = u32x4{pack_bf16(v[0].w, v[1].w), pack_bf16(v[2].w, v[3].w), pack_bf16(v[4].w, v[5].w), pack_bf16(v[6].w, v[7].w)} & m;
    };
    // bias gradient: the workgroups with ci-tile 0 and the centre ky see every dY element exactly once
    const bool do_bias = a.dbias != nullptr && ci0 == 0 && ky == KS / 2;
    f32x4 bsum[NJ];
#pragma unroll
    for (int k = 0; k < NJ; ++k) bsum[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bf16 source: v[q].x holds channels {0,1}, v[q].y channels {2,3} of image q; interleave images with v_perm_b32
    auto put16 = [&](uint16_t* dst, int pitch, const f32x4 (&v)[8], uint32_t m) {
        constexpr unsigned LO = 0x05040100u, HI = 0x07060302u;      // result = {a.half, b.half}, a in the upper 16 bits
        auto row = [&](int comp, unsigned sel) {
            auto pick = [&](const f32x4& f) { return __float_as_uint(comp ? f.y : f.x); };
            return u32x4{__builtin_amdgcn_perm(pick(v[1]), pick(v[0]), sel), __builtin_amdgcn_perm(pick(v[3]), pick(v[2]), sel),
                         __builtin_amdgcn_perm(pick(v[5]), pick(v[4]), sel), __builtin_amdgcn_perm(pick(v[7]), pick(v[6]), sel)} & m;
        };
        *reinterpret_cast<u32x4*>(dst)             = row(0, LO);
        *reinterpret_cast<u32x4*>(dst + pitch)     = row(0, HI);
        *reinterpret_cast<u32x4*>(dst + 2 * pitch) = row(1, LO);
        *reinterpret_cast<u32x4*>(dst + 3 * pitch) = row(1, HI);
    };
    auto commit_unit = [&](auto jc, int buf) {         // registers -> LDS buffer `buf`
        constexpr int j = decltype(jc)::value;
        uint16_t* base = lds + buf * BUF;
        if constexpr (j < NUX) {
            if constexpr (P16) put16(base + x_dst[j], XP, U[j], keep[j]); else put(base + x_dst[j], XP, U[j], keep[j]);
        } else {
            constexpr int k = j - NUX;
            if constexpr (Q16) put16(base + BI * XP + y_dst[k], YP, U[j], ~0u); else put(base + BI * XP + y_dst[k], YP, U[j], ~0u);
            if (do_bias) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if constexpr (Q16) {
                        const unsigned ux = __float_as_uint(U[j][q].x), uy = __float_as_uint(U[j][q].y);
                        bsum[k] += f32x4{__uint_as_float(ux << 16), __uint_as_float(ux & 0xffff0000u),
                                         __uint_as_float(uy << 16), __uint_as_float(uy & 0xffff0000u)};
                    } else {
                        bsum[k] += U[j][q];
                    }
                }
            }
        }
    };
    const int arow = (wi * 64 + (l & 31)) * XP, brow = (wj * (32 * NJ) + (l & 31)) * YP;
    auto mma_step = [&](int s, int buf) {
        const uint16_t* Xb = lds + buf * BUF;
        const uint16_t* Yb = Xb + BI * XP;
        const int p = 2 * s + (l >> 5);
        const int r_ = p >> a.tw_sh, xx = p & (a.TW - 1);
        bf16x8 bf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(&Yb[brow + j * 32 * YP + ((MI_W3_ABL & 8) ? 0 : p * 8)]);
        const int xa = (r_ * TW2 + xx) * 8;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            bf16x8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&Xb[arow + i * 32 * XP + ((MI_W3_ABL & 8) ? 0 : xa + kx * 8)]);
            if constexpr (!(MI_W3_ABL & 1)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[kx][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[kx][i][j], 0, 0, 0);
            } else { acc[kx][0][0][0] += (float)af[0][0] + (float)bf[0][0] + (float)af[1][0] + (float)bf[NJ - 1][0]; }
        }
    };

    // ---- prologue: chunk cbeg through registers into buffer cbeg&1, then chunk cbeg+1 into the registers
    static_for<0, NU>([&](auto jc) { issue_unit(jc, cbeg); });
    static_for<0, NU>([&](auto jc) { commit_unit(jc, cbeg & 1); });
    static_for<0, NU>([&](auto jc) { issue_unit(jc, min(cbeg + 1, cend - 1)); });
    __syncthreads();
    for (int c = cbeg; c + 1 < cend; ++c) {
        const int buf = c & 1;
        const int c2 = min(c + 2, cend - 1);           // past the end: a harmless re-read of the last chunk
        static_for<0, 8>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s < NU) {
                if constexpr (!(MI_W3_ABL & 4)) commit_unit(sc, buf ^ 1);
                if constexpr (!(MI_W3_ABL & 2)) issue_unit(sc, c2);
            }
            mma_step(s, buf);
        });
        __syncthreads();
    }
    static_for<0, 8>([&](auto sc) { mma_step(decltype(sc)::value, (cend - 1) & 1); });

    if (do_bias) {
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            f32x4 v = bsum[k];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {              // the 8 lanes hp_sub = 0..7 hold the same channel quad
                v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64);
                v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
            }
            if (hp_sub == 0 && y_ch[k] < a.Cj) {
                atomicAdd(a.dbias + y_ch[k], v.x); atomicAdd(a.dbias + y_ch[k] + 1, v.y);
                atomicAdd(a.dbias + y_ch[k] + 2, v.z); atomicAdd(a.dbias + y_ch[k] + 3, v.w);
            }
        }
    }
    if (a.ws) {
        // this workgroup's partial tile in register order: slot ((kx*2+i)*NJ+j)*4+rq holds, for thread t, the four
        // accumulator values of register quad rq -> every store instruction writes 1 KB contiguous per wave;
        // wgrad_reduce_kernel sums the k-slices and undoes the permutation
        float* tile = a.ws + (size_t)(split * gsz + within) * (KS * BI * BJ) + t * 4;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq)
                        *reinterpret_cast<f32x4*>(tile + (((kx * 2 + i) * NJ + j) * 4 + rq) * 1024) =
                            f32x4{acc[kx][i][j][4 * rq], acc[kx][i][j][4 * rq + 1], acc[kx][i][j][4 * rq + 2], acc[kx][i][j][4 * rq + 3]};
        return;
    }
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
        float* out = a.dW + (size_t)(ky * KS + kx) * a.Ci * a.Cj;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = ci0 + wi * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (row >= a.Ci) continue;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    int col = co0 + wj * (32 * NJ) + j * 32 + (l & 31);
                    if (col < a.Cj) atomicAdd(out + (size_t)row * a.Cj + col, acc[kx][i][j][r]);
                }
            }
    }
}

// dW[(ky*KS+kx)][ci][co] += sum over k-slices of the partial tiles written above (fixed order: deterministic).
// grid = (G thread-slices, register slots, tiles); a workgroup sums 256/G threads' float4 of one slot, its G
// thread groups each taking every G-th k-slice with coalesced 16-byte loads (G = 16 when there are many k-slices).
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, int Ci, int Cj,
                                                           int KS, int NJ, int gx, int gy, int splits) {
    constexpr int IB = 256 / G;
    __shared__ f32x4 red[G][IB];
    const int BJ = 64 * NJ;
    const int gsz = gx * gy * KS;
    const size_t tile_elems = (size_t)KS * 128 * BJ;
    const int it = threadIdx.x % IB, grp = threadIdx.x / IB;
    const int tt = blockIdx.x * IB + it, wv = tt >> 6, l = tt & 63;       // thread of the producing workgroup
    const int slot = blockIdx.y, within = blockIdx.z;
    const float* p = ws + (size_t)within * tile_elems + (size_t)slot * 1024 + tt * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    {                                                    // four independent 16-byte loads in flight per thread
        const size_t stride = (size_t)gsz * tile_elems;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
        int sp = grp;
        for (; sp + 3 * G < splits; sp += 4 * G) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (size_t)sp * stride);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + G) * stride);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + 2 * G) * stride);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + 3 * G) * stride);
            s += v0 + v2; s1 += v1 + v3;
        }
        for (; sp < splits; sp += G) s += *reinterpret_cast<const f32x4*>(p + (size_t)sp * stride);
        s += s1;
    }
    red[grp][it] = s;
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int g = 1; g < G; ++g) s += red[g][it];
    const int rq = slot & 3, j = (slot >> 2) % NJ, i = ((slot >> 2) / NJ) & 1, kx = (slot >> 2) / NJ / 2;
    const int ky = within % KS, txy = within / KS;
    const int ci = (txy % gx) * 128 + (wv >> 1) * 64 + i * 32 + 8 * rq + 4 * (l >> 5);
    const int co = (txy / gx) * BJ + (wv & 1) * (32 * NJ) + j * 32 + (l & 31);
    if (co >= Cj) return;
    float* out = dW + ((size_t)(ky * KS + kx) * Ci + ci) * Cj + co;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (ci + e < Ci) out[(size_t)e * Cj] += s[e];
}

}  // namespace

static bool w3_ok(const MiWgradDesc* d) {
    const bool k3 = d->KH == 3 && d->KW == 3 && d->pad == 1, k1 = d->KH == 1 && d->KW == 1 && d->pad == 0;
    if (!(k3 || k1) || d->stride != 1 || !d->gather_i || d->mode != 1) return false;
    if (d->GH != d->DH || d->GW != d->DW) return false;
    if (d->N % 8 || d->Ci % 32 || d->Cj % 32 || d->I1 % 32) return false;
    int W = d->DW, H = d->DH;
    if (W >= 16) return W % 16 == 0;
    if (W == 8) return H % 2 == 0;
    if (W == 4) return H % 4 == 0;
    return false;
}

extern "C" int mi_conv3x3_wgrad_supported(const MiWgradDesc* d) { return (d && w3_ok(d)) ? 1 : 0; }

static int w3_plan(const MiWgradDesc* d, W3Args& a, int& BJ, bool& wide);

// Which instantiation wgrad3x3_kernel<NJ, KS, IO> runs for a descriptor and how many k-slices (profiling attribution only)
extern "C" int mi_conv3x3_wgrad_tile(const MiWgradDesc* d, int* nj, int* splits) {
    MI_REQUIRE(d && nj && splits && w3_ok(d), "unsupported descriptor");
    W3Args a; int BJ; bool wide;
    w3_plan(d, a, BJ, wide);
    *nj = wide ? 2 : 1; *splits = a.splits;
    return 0;
}

extern "C" size_t mi_conv3x3_wgrad_workspace(const MiWgradDesc* d) {
    if (!d || !w3_ok(d)) return 0;
    W3Args a; int BJ; bool wide;
    w3_plan(d, a, BJ, wide);
    return (size_t)a.gx * a.gy * d->KH * a.splits * d->KH * 128 * BJ * sizeof(float);
}

extern "C" int mi_conv3x3_wgrad_bias(const MiWgradDesc* d, const float* P, const float* P2, const float* Q, float* dW,
                                     float* dbias, void* workspace, size_t ws_bytes, void* stream);
static int w3_dispatch(const MiWgradDesc* d, const float* P, const float* P2, const float* Q, float* dW,
                       float* dbias, void* workspace, size_t ws_bytes, int io, void* stream);

extern "C" int mi_conv3x3_wgrad_ws(const MiWgradDesc* d, const float* P, const float* P2, const float* Q, float* dW,
                                   void* workspace, size_t ws_bytes, void* stream) {
    return mi_conv3x3_wgrad_bias(d, P, P2, Q, dW, nullptr, workspace, ws_bytes, stream);
}

extern "C" int mi_conv3x3_wgrad(const MiWgradDesc* d, const float* P, const float* P2, const float* Q, float* dW,
                                void* stream) {
    return mi_conv3x3_wgrad_ws(d, P, P2, Q, dW, nullptr, 0, stream);
}

extern "C" int mi_conv3x3_wgrad_bias(const MiWgradDesc* d, const float* P, const float* P2, const float* Q, float* dW,
                                     float* dbias, void* workspace, size_t ws_bytes, void* stream) {
    return w3_dispatch(d, P, P2, Q, dW, dbias, workspace, ws_bytes, 0, stream);
}

// bf16 activation storage: io bit 0 = P (and P2) are bf16 tensors, bit 1 = Q is bf16 (strides count elements). 3x3 only.
extern "C" int mi_conv3x3_wgrad_io(const MiWgradDesc* d, const void* P, const void* P2, const void* Q, float* dW,
                                   float* dbias, void* workspace, size_t ws_bytes, int io, void* stream) {
    if (!d || (d->KH != 3 && d->KH != 1) || (io & ~3)) return mi_set_error(-1, "mi_conv3x3_wgrad_io: 3x3 or 1x1, io in 0..3");
    return w3_dispatch(d, (const float*)P, (const float*)P2, (const float*)Q, dW, dbias, workspace, ws_bytes, io, stream);
}

// Measurement aid (bench.py's per-kernel HIP-event timing): 0 = both kernels (normal), 1 = the contraction kernel only,
// 2 = the partial-tile reduce only.  Calling a launch once with 1 and once with 2 is equivalent to one normal call.
static int g_w3_phase = 0;
extern "C" int mi_debug_wgrad3x3_phase(int phase) {
    if (phase < 0 || phase > 2) return mi_set_error(-1, "mi_debug_wgrad3x3_phase: phase in 0..2");
    g_w3_phase = phase;
    return 0;
}

static int w3_dispatch(const MiWgradDesc* d, const float* P, const float* P2, const float* Q, float* dW,
                       float* dbias, void* workspace, size_t ws_bytes, int io, void* stream) {
    MI_REQUIRE(d && P && Q && dW, "null argument");
    MI_REQUIRE(w3_ok(d), "descriptor not supported by the 3x3 wgrad kernel (use mi_conv_wgrad)");
    MI_REQUIRE(d->I1 == d->Ci || P2, "two-source split without P2");
    MI_REQUIRE(d->ldp % 4 == 0 && d->ldq % 4 == 0 && (!P2 || d->ldp2 % 4 == 0) &&
               (((uintptr_t)P | (uintptr_t)Q | (uintptr_t)(P2 ? P2 : P)) & 15) == 0, "operands must be 16-byte aligned, ld % 4 == 0");
    W3Args a;
    a.P = P; a.P2 = P2 ? P2 : P; a.Q = Q; a.dW = dW; a.dbias = dbias;
    a.ldp = d->ldp; a.ldp2 = P2 ? d->ldp2 : d->ldp; a.ldq = d->ldq;
    int BJ; bool wide;
    w3_plan(d, a, BJ, wide);
    const int KS = d->KH;
    a.ws = nullptr;
    if (workspace && ws_bytes >= (size_t)a.gx * a.gy * KS * a.splits * KS * 128 * BJ * sizeof(float) && a.splits > 1)
        a.ws = (float*)workspace;
    dim3 grid((unsigned)(a.gx * a.gy * KS * (a.xcd_map ? (a.splits + 7) / 8 * 8 : a.splits)));
    hipStream_t st = (hipStream_t)stream;
    const int XP = ((a.TH * (a.TW + KS - 1)) | 1) * 8;
    const size_t lds = (size_t)(128 * XP + BJ * 17 * 8) * 2 * 2;      // double-buffered
    static MiPerDevice once;
    once.run([] {
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (g_w3_phase == 2) {
        MI_REQUIRE(a.ws, "phase 2 needs the split plan");
    } else if (KS == 3 && io) {
        static MiPerDevice once_io;
        once_io.run([] {
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
#define MI_W3_GO(IOV) do { if (wide) hipLaunchKernelGGL((wgrad3x3_kernel<2, 3, IOV>), grid, dim3(256), lds, st, a); \
                           else hipLaunchKernelGGL((wgrad3x3_kernel<1, 3, IOV>), grid, dim3(256), lds, st, a); } while (0)
        if (io == 1) MI_W3_GO(1); else if (io == 2) MI_W3_GO(2); else MI_W3_GO(3);
#undef MI_W3_GO
    } else if (KS == 3) {
        if (wide) hipLaunchKernelGGL((wgrad3x3_kernel<2, 3>), grid, dim3(256), lds, st, a);
        else      hipLaunchKernelGGL((wgrad3x3_kernel<1, 3>), grid, dim3(256), lds, st, a);
    } else if (io) {
        static MiPerDevice once_io1;
        once_io1.run([] {
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
#define MI_W1_GO(IOV) do { if (wide) hipLaunchKernelGGL((wgrad3x3_kernel<2, 1, IOV>), grid, dim3(256), lds, st, a); \
                           else hipLaunchKernelGGL((wgrad3x3_kernel<1, 1, IOV>), grid, dim3(256), lds, st, a); } while (0)
        if (io == 1) MI_W1_GO(1); else if (io == 2) MI_W1_GO(2); else MI_W1_GO(3);
#undef MI_W1_GO
    } else {
        if (wide) hipLaunchKernelGGL((wgrad3x3_kernel<2, 1>), grid, dim3(256), lds, st, a);
        else      hipLaunchKernelGGL((wgrad3x3_kernel<1, 1>), grid, dim3(256), lds, st, a);
    }
    if (a.ws && g_w3_phase != 1) {
        const dim3 rg(1, KS * 2 * (BJ / 64) * 4, a.gx * a.gy * KS);
        if (a.splits >= 128)
            hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(16, rg.y, rg.z), dim3(256), 0, st, a.ws, dW, d->Ci, d->Cj, KS, BJ / 64, a.gx, a.gy, a.splits);
        else
            hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(4, rg.y, rg.z), dim3(256), 0, st, a.ws, dW, d->Ci, d->Cj, KS, BJ / 64, a.gx, a.gy, a.splits);
    }
    MI_LAUNCH_CHECK();
    return 0;
}

static int w3_plan(const MiWgradDesc* d, W3Args& a, int& BJ, bool& wide) {
    a.N = d->N; a.H = d->DH; a.W = d->DW; a.Ci = d->Ci; a.Cj = d->Cj; a.I1 = d->I1;
    a.TW = a.W >= 16 ? 16 : a.W; a.TH = 16 / a.TW;
    a.tw_sh = a.TW == 16 ? 4 : (a.TW == 8 ? 3 : 2);
    a.tiles_x = a.W / a.TW; a.tiles = a.tiles_x * (a.H / a.TH);
    a.total = (a.N / 8) * a.tiles;
    static const int force_nj = (int)mi_knob("MI_W3_NJ", 0);
    wide = force_nj ? force_nj == 2 : (d->Cj % 128 == 0 || d->Cj > 256);
    {   // the double-buffered LDS image must fit 160 KB: 4x4 images (25 X slots per channel) only fit with 64-wide co tiles
        const int xp = ((a.TH * (a.TW + d->KH - 1)) | 1) * 8;
        if ((size_t)(128 * xp + 128 * 17 * 8) * 2 * 2 > 160 * 1024) wide = false;
    }
    BJ = wide ? 128 : 64;
    const int KS = d->KH;
    long base = (long)((d->Ci + 127) / 128) * ((d->Cj + BJ - 1) / BJ) * KS;
    // exactly one round of workgroups: the kernel holds ~480 registers per lane, i.e. one workgroup per CU
    static const long target = mi_knob("MI_W3_BLOCKS", 256);
    long splits = target / base;
    // a k-slice costs a prologue, a partial-tile store and a share of the reduce: keep >= MINC chunks per slice
    static const long minc = mi_knob("MI_W3_MINC", 1);
    if (splits > a.total / minc) splits = a.total / minc;
    if (splits < 1) splits = 1;
    a.cps = (int)((a.total + splits - 1) / splits);
    a.splits = (a.total + a.cps - 1) / a.cps;
    a.gx = (d->Ci + 127) / 128; a.gy = (d->Cj + BJ - 1) / BJ;
    // measured: helps the HBM-bound 1x1 gradients with many k-slices (77 -> 64 us for 128->384 at 32x32), neutral or
    // harmful for the 3x3 ones (not traffic-bound; rounding the k-slices up to a multiple of 8 idles workgroups)
    static const int xcd_env = (int)mi_knob("MI_W3_XCD", 1);
    a.xcd_map = xcd_env == 2 || (xcd_env == 1 && KS == 1 && a.splits >= 32);
    return 0;
}
