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

namespace {

struct W3Args {
    const float* P; const float* P2; const float* Q; float* dW;
    int N, H, W, Ci, Cj, I1, ldp, ldp2, ldq;
    int TH, TW;            // spatial tile (TH*TW == 16)
    int tiles_x, tiles;    // W/TW, tiles per image
    int total, cps, splits;
    int gx, gy;
    int xcd_map;
    float* dbias;        // optional: dbias[co] += sum over pixels of Q (the conv's bias gradient), fused into the dY staging
    float* ws;           // per-workgroup partial tiles [block][KS][128][BJ] (null -> fp32 atomics into dW)
    int ablate;          // profiling only: 1 = no MFMA, 2 = no global loads, 4 = no LDS commit, 8 = no fragment reads
};

// IO bit 0: P (layer input) is stored as bf16, bit 1: Q (output gradient) is stored as bf16.
template <int NJ, int KS, int IO = 0>
__global__ __launch_bounds__(256, 1) void wgrad3x3_kernel(const W3Args a) {
    constexpr bool P16 = IO & 1, Q16 = IO & 2;
    constexpr int BI = 128, BJ = 32 * 2 * NJ;
    constexpr int YP = 17 * 8;                          // dY pitch per co row (bf16 elems): 16 slots + 1 pad
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];      // BI*XP + BJ*YP bf16

    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int wi = wv >> 1, wj = wv & 1;
    // 1-D grid, one workgroup per CU (the kernel runs one wave per SIMD): b -> (k-slice, ky, ci-tile, co-tile)
    const int gsz = a.gx * a.gy * KS;
    int split, within;
    if (a.xcd_map) {        // workgroup b is dispatched to XCD b % 8 (observed; speed only): keep the gsz workgroups of
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;   // one k-slice on one XCD so its L2 serves the shared dY / X rows
        split = (slot / gsz) * 8 + xcd; within = slot % gsz;
        if (split >= a.splits) return;
    } else {
        split = blockIdx.x / gsz; within = blockIdx.x % gsz;
    }
    const int ky = within % KS, txy = within / KS;
    const int ci0 = (txy % a.gx) * BI, co0 = (txy / a.gx) * BJ;
    constexpr int NUX = KS == 3 ? 3 : 2;                // X staging units per wave (= position groups of 8)
    const int TW2 = a.TW + (KS - 1);
    const int PX = a.TH * TW2;                          // X positions per chunk (18, 20 or 24; 16 for 1x1)
    const int XP = (PX | 1) * 8;                        // odd slot count -> conflict-free fragment reads
    const int npg = (PX + 7) / 8;                       // position groups of 8
    uint16_t* Xs = lds;
    uint16_t* Ys = lds + BI * XP;
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
    const int cbeg = split * a.cps, cend = min(a.total, cbeg + a.cps);

    // ---- staging assignment (fixed per thread): 3 X units and NJ dY units per wave;
    //      unit = (8 positions) x (32 channels), one float4 per image per lane
    int x_pos[NUX], x_ch[NUX], x_dst[NUX];
#pragma unroll
    for (int k = 0; k < NUX; ++k) {
        const int u = wv + 4 * k;
        const int pg = u % NUX, cg = u / NUX;
        x_pos[k] = pg * 8 + hp_sub;
        x_ch[k] = ci0 + cg * 32 + c4 * 4;
        x_dst[k] = (cg * 32 + c4 * 4) * XP + x_pos[k] * 8;
    }
    int y_pos[NJ], y_ch[NJ], y_dst[NJ];
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        const int u = wv + 4 * k;
        const int pg = u & 1, cg = u >> 1;
        y_pos[k] = pg * 8 + hp_sub;
        y_ch[k] = co0 + cg * 32 + c4 * 4;
        y_dst[k] = (cg * 32 + c4 * 4) * YP + y_pos[k] * 8;
    }
    // Pipeline: LDS is double-buffered by chunk.  While the 8 k-steps of chunk c run on the matrix
    // cores out of buffer c&1, the wave's 3+NJ staging units of chunk c+1 are fetched (one unit per
    // k-step, three register sets in flight => two k-steps of MFMA time per load) and written to the
    // other buffer.  One barrier per chunk (96 MFMAs per wave at NJ = 2).
    constexpr int NU = NUX + NJ;
    float4 U[NU][8];                                   // one register set per staging unit: a whole chunk in flight
    const int BUF = BI * XP + BJ * YP;                 // bf16 elements per LDS buffer

    auto issue_unit = [&](float4 (&r)[8], int j, int c) {     // unit j of chunk c: global -> registers
        if (c >= cend || (a.ablate & 2)) return;
        const int g = c / a.tiles, tile = c - g * a.tiles;
        const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        const int y0 = ty * a.TH, x0 = tx * a.TW, n0 = g * 8;
        if (j < NUX) {
            const int r_ = x_pos[j] / TW2, xx = x_pos[j] - r_ * TW2;
            const int iy = y0 + r_ + ky - KS / 2, ix = x0 + xx - KS / 2;
            const bool ok = x_pos[j] < PX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && x_ch[j] < a.Ci;
            const float* src = a.P; int ld = a.ldp; int cc = x_ch[j];
            if (cc >= a.I1) { src = a.P2; ld = a.ldp2; cc -= a.I1; }
            const size_t eoff = ok ? ((size_t)n0 * HW + iy * a.W + ix) * ld + cc : (size_t)n0 * HW * ld;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float4 v;
                if constexpr (P16) {      // 4 bf16 channels = 8 bytes, carried in .x/.y
                    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(src) + eoff + (size_t)q * HW * ld);
                    v = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
                } else {
                    v = *reinterpret_cast<const float4*>(src + eoff + (size_t)q * HW * ld);
                }
                r[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int k = j - NUX;
            const int r_ = y_pos[k] / a.TW, xx = y_pos[k] - r_ * a.TW;
            const bool ok = y_ch[k] < a.Cj;
            const size_t eoff = ((size_t)n0 * HW + (y0 + r_) * a.W + x0 + xx) * a.ldq + (ok ? y_ch[k] : 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float4 v;
                if constexpr (Q16) {
                    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(a.Q) + eoff + (size_t)q * HW * a.ldq);
                    v = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
                } else {
                    v = *reinterpret_cast<const float4*>(a.Q + eoff + (size_t)q * HW * a.ldq);
                }
                r[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto put = [&](uint16_t* dst, int pitch, const float4 (&v)[8]) {   // 8 images x 4 channels -> 4 rows of 8 bf16
        *reinterpret_cast<uint4*>(dst)             = make_uint4(pack_bf16(v[0].x, v[1].x), pack_bf16(v[2].x, v[3].x), pack_bf16(v[4].x, v[5].x), pack_bf16(v[6].x, v[7].x));
        *reinterpret_cast<uint4*>(dst + pitch)     = make_uint4(pack_bf16(v[0].y, v[1].y), pack_bf16(v[2].y, v[3].y), pack_bf16(v[4].y, v[5].y), pack_bf16(v[6].y, v[7].y));
        *reinterpret_cast<uint4*>(dst + 2 * pitch) = make_uint4(pack_bf16(v[0].z, v[1].z), pack_bf16(v[2].z, v[3].z), pack_bf16(v[4].z, v[5].z), pack_bf16(v[6].z, v[7].z));
        *reinterpret_cast<uint4*>(dst + 3 * pitch) = make_uint4(pack_bf16(v[0].w, v[1].w), pack_bf16(v[2].w, v[3].w), pack_bf16(v[4].w, v[5].w), pack_bf16(v[6].w, v[7].w));
    };
    // bias gradient: the workgroups with ci-tile 0 and the centre ky see every dY element exactly once
    const bool do_bias = a.dbias != nullptr && ci0 == 0 && ky == KS / 2;
    float4 bsum[NJ];
#pragma unroll
    for (int k = 0; k < NJ; ++k) bsum[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // bf16 source: v[q].x holds channels {0,1}, v[q].y channels {2,3} of image q; interleave images with v_perm_b32
    auto put16 = [&](uint16_t* dst, int pitch, const float4 (&v)[8]) {
        constexpr unsigned LO = 0x05040100u, HI = 0x07060302u;      // result = {a.half, b.half}, a in the upper 16 bits
        auto row = [&](auto pick, unsigned sel) {
            return make_uint4(__builtin_amdgcn_perm(pick(v[1]), pick(v[0]), sel), __builtin_amdgcn_perm(pick(v[3]), pick(v[2]), sel),
                              __builtin_amdgcn_perm(pick(v[5]), pick(v[4]), sel), __builtin_amdgcn_perm(pick(v[7]), pick(v[6]), sel));
        };
        auto px = [](const float4& f) { return __float_as_uint(f.x); };
        auto py = [](const float4& f) { return __float_as_uint(f.y); };
        *reinterpret_cast<uint4*>(dst)             = row(px, LO);
        *reinterpret_cast<uint4*>(dst + pitch)     = row(px, HI);
        *reinterpret_cast<uint4*>(dst + 2 * pitch) = row(py, LO);
        *reinterpret_cast<uint4*>(dst + 3 * pitch) = row(py, HI);
    };
    auto commit_unit = [&](const float4 (&r)[8], int j, int buf) {     // registers -> LDS buffer `buf`
        if (a.ablate & 4) return;
        uint16_t* base = lds + buf * BUF;
        if (j < NUX) {
            if (x_pos[j] < PX) { if constexpr (P16) put16(base + x_dst[j], XP, r); else put(base + x_dst[j], XP, r); }
        } else {
            if constexpr (Q16) put16(base + BI * XP + y_dst[j - NUX], YP, r); else put(base + BI * XP + y_dst[j - NUX], YP, r);
            if (do_bias) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if constexpr (Q16) {
                        const unsigned ux = __float_as_uint(r[q].x), uy = __float_as_uint(r[q].y);
                        bsum[j - NUX].x += __uint_as_float(ux << 16); bsum[j - NUX].y += __uint_as_float(ux & 0xffff0000u);
                        bsum[j - NUX].z += __uint_as_float(uy << 16); bsum[j - NUX].w += __uint_as_float(uy & 0xffff0000u);
                    } else {
                        bsum[j - NUX].x += r[q].x; bsum[j - NUX].y += r[q].y; bsum[j - NUX].z += r[q].z; bsum[j - NUX].w += r[q].w;
                    }
                }
            }
        }
    };
    auto mma_step = [&](int s, int buf) {
        if (a.ablate & 1) return;
        const uint16_t* Xb = lds + buf * BUF;
        const uint16_t* Yb = Xb + BI * XP;
        const int arow = (wi * 64 + (l & 31)) * XP, brow = (wj * (32 * NJ) + (l & 31)) * YP;
        const int p = 2 * s + (l >> 5);
        const int r_ = p / a.TW, xx = p - r_ * a.TW;
        bf16x8 bf[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(&Yb[brow + j * 32 * YP + p * 8]);
        const int xa = (r_ * TW2 + xx) * 8;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            bf16x8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&Xb[arow + i * 32 * XP + xa + kx * 8]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[kx][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[kx][i][j], 0, 0, 0);
        }
    };

    // Pipeline: registers hold chunk c+1 (fetched during chunk c-1); at k-step s of chunk c, unit s is
    // converted and written to the other LDS buffer, and the same registers are immediately re-armed
    // with unit s of chunk c+2 -> every global load has a full chunk (8 k-steps, ~100 MFMAs) to land.
    if (cbeg < cend) {
#pragma unroll
        for (int j = 0; j < NU; ++j) { issue_unit(U[j], j, cbeg); commit_unit(U[j], j, cbeg & 1); }
#pragma unroll
        for (int j = 0; j < NU; ++j) issue_unit(U[j], j, cbeg + 1);
    }
    __syncthreads();
    for (int c = cbeg; c < cend; ++c) {
        const int buf = c & 1;
        const bool nxt = c + 1 < cend;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < NU) {
                if (nxt) commit_unit(U[s], s, buf ^ 1);
                issue_unit(U[s], s, c + 2);
            }
            mma_step(s, buf);
        }
        __syncthreads();
    }

    if (do_bias) {
#pragma unroll
        for (int k = 0; k < NJ; ++k) {
            float4 v = bsum[k];
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
        // plain coalesced stores of this workgroup's partial tile; wgrad_reduce_kernel sums the k-slices
        float* tile = a.ws + (size_t)blockIdx.x * (KS * BI * BJ);
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wi * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        tile[(size_t)(kx * BI + row) * BJ + wj * (32 * NJ) + j * 32 + (l & 31)] = acc[kx][i][j][r];
                }
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
// A workgroup owns 32 consecutive outputs; its 8 waves... (8 groups of 32 lanes) each take every 8th slice.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, int Ci, int Cj,
                                                           int KS, int BJ, int gx, int gy, int splits) {
    __shared__ float red[8][32];
    const size_t total = (size_t)KS * KS * Ci * Cj;
    const int gsz = gx * gy * KS;
    const size_t tile_elems = (size_t)KS * 128 * BJ;
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (size_t e0 = (size_t)blockIdx.x * 32; e0 < total; e0 += (size_t)gridDim.x * 32) {
        const size_t e = e0 + el;
        float s = 0.f;
        if (e < total) {
            const int co = (int)(e % Cj); size_t q = e / Cj;
            const int ci = (int)(q % Ci); const int tap = (int)(q / Ci);
            const int ky = tap / KS, kx = tap - ky * KS;
            const int tx = ci / 128, ty = co / BJ;
            const int within = (ty * gx + tx) * KS + ky;
            const float* p = ws + (size_t)within * tile_elems + (size_t)(kx * 128 + (ci - tx * 128)) * BJ + (co - ty * BJ);
            for (int sp = grp; sp < splits; sp += 8) s += p[(size_t)sp * gsz * tile_elems];
        }
        red[grp][el] = s;
        __syncthreads();
        if (grp == 0 && e < total) {
            float v = red[0][el];
#pragma unroll
            for (int g = 1; g < 8; ++g) v += red[g][el];
            dW[e] += v;
        }
        __syncthreads();
    }
}

// few k-slices, many outputs: one thread per output, slices summed serially
__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const float* __restrict__ ws, float* __restrict__ dW, int Ci, int Cj,
                                                                int KS, int BJ, int gx, int gy, int splits) {
    const size_t total = (size_t)KS * KS * Ci * Cj;
    const int gsz = gx * gy * KS;
    const size_t tile_elems = (size_t)KS * 128 * BJ;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int co = (int)(e % Cj); size_t q = e / Cj;
        const int ci = (int)(q % Ci); const int tap = (int)(q / Ci);
        const int ky = tap / KS, kx = tap - ky * KS;
        const int tx = ci / 128, ty = co / BJ;
        const int within = (ty * gx + tx) * KS + ky;
        const float* p = ws + (size_t)within * tile_elems + (size_t)(kx * 128 + (ci - tx * 128)) * BJ + (co - ty * BJ);
        float s = 0.f;
        for (int sp = 0; sp < splits; ++sp) s += p[(size_t)sp * gsz * tile_elems];
        dW[e] += s;
    }
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
    if (!d || d->KH != 3 || (io & ~3)) return mi_set_error(-1, "mi_conv3x3_wgrad_io: 3x3 only, io in 0..3");
    return w3_dispatch(d, (const float*)P, (const float*)P2, (const float*)Q, dW, dbias, workspace, ws_bytes, io, stream);
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
    if (workspace && ws_bytes >= (size_t)a.gx * a.gy * KS * a.splits * KS * 128 * BJ * sizeof(float) && a.splits > 1 && !a.xcd_map)
        a.ws = (float*)workspace;
    dim3 grid((unsigned)(a.gx * a.gy * KS * (a.xcd_map ? (a.splits + 7) / 8 * 8 : a.splits)));
    hipStream_t st = (hipStream_t)stream;
    const int XP = ((a.TH * (a.TW + KS - 1)) | 1) * 8;
    const size_t lds = (size_t)(128 * XP + BJ * 17 * 8) * 2 * 2;      // double-buffered
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)once;
    if (KS == 3 && io) {
        static bool once_io = [] {
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<2, 3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<1, 3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return true;
        }();
        (void)once_io;
#define MI_W3_GO(IOV) do { if (wide) hipLaunchKernelGGL((wgrad3x3_kernel<2, 3, IOV>), grid, dim3(256), lds, st, a); \
                           else hipLaunchKernelGGL((wgrad3x3_kernel<1, 3, IOV>), grid, dim3(256), lds, st, a); } while (0)
        if (io == 1) MI_W3_GO(1); else if (io == 2) MI_W3_GO(2); else MI_W3_GO(3);
#undef MI_W3_GO
    } else if (KS == 3) {
        if (wide) hipLaunchKernelGGL((wgrad3x3_kernel<2, 3>), grid, dim3(256), lds, st, a);
        else      hipLaunchKernelGGL((wgrad3x3_kernel<1, 3>), grid, dim3(256), lds, st, a);
    } else {
        if (wide) hipLaunchKernelGGL((wgrad3x3_kernel<2, 1>), grid, dim3(256), lds, st, a);
        else      hipLaunchKernelGGL((wgrad3x3_kernel<1, 1>), grid, dim3(256), lds, st, a);
    }
    if (a.ws) {
        const size_t total = (size_t)KS * KS * d->Ci * d->Cj;
        if (a.splits >= 16) {
            int blocks = (int)((total + 31) / 32); if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, a.ws, dW, d->Ci, d->Cj, KS, BJ, a.gx, a.gy, a.splits);
        } else {
            int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3(blocks), dim3(256), 0, st, a.ws, dW, d->Ci, d->Cj, KS, BJ, a.gx, a.gy, a.splits);
        }
    }
    MI_LAUNCH_CHECK();
    return 0;
}

static int w3_plan(const MiWgradDesc* d, W3Args& a, int& BJ, bool& wide) {
    a.N = d->N; a.H = d->DH; a.W = d->DW; a.Ci = d->Ci; a.Cj = d->Cj; a.I1 = d->I1;
    a.TW = a.W >= 16 ? 16 : a.W; a.TH = 16 / a.TW;
    a.tiles_x = a.W / a.TW; a.tiles = a.tiles_x * (a.H / a.TH);
    a.total = (a.N / 8) * a.tiles;
    static const int force_nj = [] { const char* e = getenv("MI_W3_NJ"); return e ? atoi(e) : 0; }();
    wide = force_nj ? force_nj == 2 : (d->Cj % 128 == 0 || d->Cj > 256);
    {   // the double-buffered LDS image must fit 160 KB: 4x4 images (25 X slots per channel) only fit with 64-wide co tiles
        const int xp = ((a.TH * (a.TW + d->KH - 1)) | 1) * 8;
        if ((size_t)(128 * xp + 128 * 17 * 8) * 2 * 2 > 160 * 1024) wide = false;
    }
    BJ = wide ? 128 : 64;
    const int KS = d->KH;
    long base = (long)((d->Ci + 127) / 128) * ((d->Cj + BJ - 1) / BJ) * KS;
    // exactly one round of workgroups: the kernel holds ~480 registers per lane, i.e. one workgroup per CU
    static const long target = [] { const char* e = getenv("MI_W3_BLOCKS"); return e ? atol(e) : 256L; }();
    long splits = target / base;
    if (splits > a.total) splits = a.total;
    if (splits < 1) splits = 1;
    a.cps = (int)((a.total + splits - 1) / splits);
    a.splits = (a.total + a.cps - 1) / a.cps;
    a.gx = (d->Ci + 127) / 128; a.gy = (d->Cj + BJ - 1) / BJ;
    static const int xcd_env = [] { const char* e = getenv("MI_W3_XCD"); return e ? atoi(e) : 0; }();
    a.xcd_map = xcd_env && a.gx * a.gy * KS <= 12;
    if (a.xcd_map) {                                    // splits a multiple of 8 so every XCD gets whole k-slices
        long s8 = (target / base) / 8 * 8; if (s8 < 8) s8 = 8;
        if (s8 > a.total) s8 = a.total;
        a.cps = (int)((a.total + s8 - 1) / s8);
        a.splits = (a.total + a.cps - 1) / a.cps;
    }
    static const int abl = [] { const char* e = getenv("MI_W3_ABLATE"); return e ? atoi(e) : 0; }();
    a.ablate = abl;
    return 0;
}
