// 3x3 / stride 1 / pad 1 convolution (forward and data gradient) for bf16-stored activations -- the LDS-frugal variant.
//     Y[p][co] (+)= bias[co] + res[p][co] + sum_{ky,kx,ci} X[p + (ky-1, kx-1)][ci] * W[ky][kx][co][ci]
// (Block's Conv2d(dim, dim_out, 3, padding=1), reference src/models/ddpm.py:116, and its input gradient.)
//
// conv3x3_halo.hip and conv_dma.hip both move 1 KB of LDS per MFMA (a wave owns 64 x 64 outputs and reads two activation and two
// weight fragments per four MFMAs); with the weight tiles also being WRITTEN into LDS once per 128-256 pixels, the LDS port -- not
// the matrix pipe -- is what bounds them at 30-37 % MFMA-busy.  This kernel cuts the LDS traffic to 0.4 KB per MFMA:
//   * a wave owns 128 pixels x 64 channels (eight MFMA tiles): four activation + two weight fragments per eight MFMAs;
//   * the activation fragments of the LEFT and RIGHT tap columns are never read from LDS: an activation fragment holds one pixel
//     per lane, so the fragment of column kx-1 / kx+1 is the centre column's fragment moved by one lane (v_mov_b32_dpp
//     wave_shr:1 / wave_shl:1) -- 32-pixel blocks are whole image rows, so the lanes a shift would fill from a neighbouring row are
//     exactly the lanes that read the zero padding, and one v_and per register supplies it.  Only the centre column is fetched
//     (once per tap ROW and 64-channel chunk, kept in 64 registers), no column halo is staged at all.
//   * everything is staged by LDS-DMA (global_load_lds_dwordx4; swizzled through the source address, see conv_dma.hip); a stage
//     is one tap (32 MFMAs per wave), weights ride in a ring of three slots two stages ahead, the next chunk's rows trickle in
//     over five stages, and every s_waitcnt vmcnt(N) is counted (the chunk's nine stages are one straight-line body);
//   * fragment reads run two (tap, k-step) units ahead of their MFMAs ACROSS the stage barriers (see conv_dma.hip).
// A workgroup = 4 waves (one per SIMD, 2 x 2) = 256 output pixels x 128 (NI = 2) or 64 (NI = 1) output channels.
#include "tr_common.h"

namespace {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) uint32_t g_zero_page2[64];     // 256 zero bytes: the rows above / below an image

struct ShiftConvArgs {
    const uint16_t* x; const uint16_t* x2; const uint16_t* w; const float* bias; const float* res; void* y;
    int N, H, W, K, Nc, K1, ldx, ldx2, ldy, ldr, accumulate, flip;
    int TH, TI, XP, tiles_per_img, xmap;
    int qmap, gx, gy;    // 1-D launch of gx pixel tiles x gy channel tiles: XCD = (pixel group, channel group) of a (8 / qmap) x qmap split
};

constexpr int SBM = 256, SCK = 64;
constexpr int MAXXP = 320;                      // tile pixels incl. the row above and below: 10 x 32, 18 x 16 (288), 4 images x 10 x 8
constexpr int SXBUF = MAXXP * 128;
constexpr int SPD = 2;

__device__ __forceinline__ bf16x8 lds_b128s(uint32_t addr) {
    typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
    return *(lds_bf16x8*)(uintptr_t)addr;
}

// fragment of the tap column to the left (DIR = 0: lane l takes lane l-1) / right (DIR = 1: lane l+1) of the centre column
template <int DIR> __device__ __forceinline__ bf16x8 shift_frag(const bf16x8& c, uint32_t mask) {
    const u32x4 v = __builtin_bit_cast(u32x4, c);
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        o[q] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[q], DIR == 0 ? 0x138 : 0x130, 0xf, 0xf, true) & mask;
    return __builtin_bit_cast(bf16x8, o);
}

// WM waves along the pixel axis (2: one wave per SIMD, 128 pixels per wave; 4: two waves per SIMD, 64 pixels per wave) x 2 along channels
template <int WM, int NI, bool OUT16>
__global__ __launch_bounds__(128 * WM, 1) void conv_shift_kernel(const ShiftConvArgs a) {
    constexpr int NW = 2 * WM, MI = 8 / WM;           // waves, 32-pixel blocks per wave
    constexpr int BN = 64 * NI;                       // output channels per workgroup (two waves of 32 * NI)
    constexpr int WTAP = BN * 128;                    // bytes of one tap's weight tile
    constexpr int WPER = BN / (8 * NW);               // weight DMA instructions per wave and tap
    constexpr int XPW = 40 / NW, XPS = XPW / 5;       // X DMA instructions per wave and chunk / per stage (first five stages)
    static_assert(WPER >= 1 && XPS >= 1, "every wave stages something");
    constexpr int XOFF = 0, WOFF = 2 * SXBUF;         // [X buffers][3 weight slots]
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wv >> 1, wn = wv & 1;             // wm in [0, WM)
    int bx = blockIdx.x;
    if (a.xmap) {        // an image's row tiles share rows: keep them on one XCD (ids xcd + 8*slot -> image xcd + 8*m)
        const int xcd = bx & 7, slot = bx >> 3;
        bx = (xcd + 8 * (slot / a.tiles_per_img)) * a.tiles_per_img + slot % a.tiles_per_img;
    }
    int by = blockIdx.y;
    if (a.qmap) {
        // Whole-image tiles (8x8 level): workgroup ids go round-robin over the 8 XCDs, each with its own L2.  With the channel tiles
        // in grid.y every XCD meets all of them and pulls the whole weight tensor: 54.8 MB for the 512 -> 512 layer (PMC) against
        // 21.5 MB algorithmic.  Here an XCD owns gx / P pixel tiles x gy / Q channel tiles: traffic Q * in + P * w + out, 44 MB at
        // (P, Q) = (4, 2); the channel tiles of one pixel tile are adjacent in time (same mapping as conv3x3_halo.hip).
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3, Q = a.qmap, P = 8 / Q;
        const int ppx = a.gx / P, cpq = a.gy / Q;
        bx = (xcd / Q) * ppx + slot / cpq; by = (xcd % Q) * cpq + slot % cpq;
    }
    const int m0 = bx * SBM, n0 = by * BN;
    const int TH2 = a.TH + 2;
    const int Mtot = a.N * a.H * a.W;
    const int nchunks = a.K / SCK;

    // ---- DMA pieces.  X: 40 instructions per chunk, 10 per wave, two per stage over a chunk's first five stages.  Instruction
    //      idx covers tile pixels 8*idx .. +7 (tile pixel hp = (image ti, row hy = y+1, column x)); lane -> pixel lane >> 3, stored
    //      chunk position lane & 7 holds channel chunk (lane & 7) ^ ((hp >> 1) & 7).
    int xpix[XPW], xcol[XPW];
    {
        int img0, y0;
        if (a.TI > 1) { img0 = bx * a.TI; y0 = 0; }
        else { img0 = bx / a.tiles_per_img; y0 = (bx % a.tiles_per_img) * a.TH; }
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int hp = 8 * (wv + NW * i) + (l >> 3);
            int v = -1;
            if (hp < a.XP) {
                const int row = hp / a.W, x = hp - row * a.W;
                const int ti = row / TH2, hy = row - ti * TH2;
                const int iy = y0 + hy - 1, img = img0 + ti;
                if (iy >= 0 && iy < a.H && img < a.N) v = (img * a.H + iy) * a.W + x;
            }
            xpix[i] = v;
            xcol[i] = ((l & 7) ^ ((hp >> 1) & 7)) * 8;
        }
    }
    int wrow[WPER], wcol[WPER];
#pragma unroll
    for (int i = 0; i < WPER; ++i) {
        const int n = 8 * (wv + NW * i) + (l >> 3);
        wrow[i] = min(n0 + n, a.Nc - 1);
        wcol[i] = ((l & 7) ^ ((n >> 1) & 7)) * 8;
    }
    const size_t tap_stride = (size_t)a.Nc * a.K;
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_page2);

    auto stage_x = [&](int ch, int i) {                  // piece i (0..9) of this wave of chunk ch's rows -> buffer ch & 1
        const int cc0 = min(ch, nchunks - 1) * SCK;
        const bool second = cc0 >= a.K1;
        const uint16_t* src = second ? a.x2 : a.x;
        const int ld = second ? a.ldx2 : a.ldx, cc = second ? cc0 - a.K1 : cc0;
        const uint16_t* p = xpix[i] >= 0 ? src + (size_t)xpix[i] * ld + cc + xcol[i] : zero + (l & 7) * 8;
        glds16(p, lds0 + XOFF + (ch & 1) * SXBUF + (wv + NW * i) * 1024);
    };
    // stage st = chunk st / 9, tap row (st % 9) / 3, tap column in the order centre, left, right
    auto stage_w = [&](int st) {
        const int stc = min(st, nchunks * 9 - 1);            // past the end: re-fetch the last tap (keeps the DMA counts static)
        const int ch = stc / 9, r9 = stc - ch * 9, ky = r9 / 3, j = r9 - ky * 3;
        const int tap = ky * 3 + (j == 0 ? 1 : (j == 1 ? 0 : 2));
        const uint16_t* base = a.w + (size_t)(a.flip ? 8 - tap : tap) * tap_stride + (size_t)ch * SCK;
#pragma unroll
        for (int i = 0; i < WPER; ++i)
            glds16(base + (size_t)wrow[i] * a.K + wcol[i], lds0 + WOFF + (st % 3) * WTAP + (wv + NW * i) * 1024);
    };

    // ---- fragment addressing.  Activations (MFMA "B" operand): lane -> pixel (l & 31) of the wave's i-th 32-pixel block (whole
    //      image rows), k-chunk 2*ks + (l >> 5); weights ("A"): lane -> output channel wn*32*NI + jn*32 + (l & 31).
    int hp0[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int r = wm * (32 * MI) + i * 32 + (l & 31);
        const int tx = r % a.W, q = r / a.W;
        const int ty = q % a.TH, ti = q / a.TH;
        hp0[i] = (ti * TH2 + ty) * a.W + tx;                 // tile pixel of tap row ky = 0 (one row up)
    }
    const int xin = (l & 31) % a.W;
    const uint32_t mask_l = xin == 0 ? 0u : ~0u, mask_r = xin == a.W - 1 ? 0u : ~0u;     // zero padding left / right of the row
    const int half16 = (l >> 5) * 16;
    int wb[NI], wsw[NI];
#pragma unroll
    for (int jn = 0; jn < NI; ++jn) {
        const int n = wn * (32 * NI) + jn * 32 + (l & 31);
        wb[jn] = n * 128; wsw[jn] = ((n >> 1) & 7) * 16;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
    bf16x8 XC[4][MI];                                        // centre-column fragments of the current tap row: [k-step][pixel block]
    bf16x8 FW[SPD + 1][NI];                                  // weight fragments, SPD units ahead
#pragma unroll
    for (int q = 0; q <= SPD; ++q)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int e = 0; e < 8; ++e) FW[q][jn][e] = (__bf16)0.f;          // the first SPD MFMA units add zero
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) XC[ks][i][e] = (__bf16)0.f;

    // ---- prologue: the first chunk's rows and the first two taps
#pragma unroll
    for (int i = 0; i < XPW; ++i) stage_x(0, i);
    stage_w(0);
    stage_w(1);

    static_assert(4 % 1 == 0 && 36 % (SPD + 1) == 0, "ring slots line up across chunks");
    for (int ch = 0; ch < nchunks; ++ch) {
        static_for<0, 9>([&](auto sc) {
            constexpr int sidx = decltype(sc)::value, ky = sidx / 3, j = sidx % 3;
            const int st = ch * 9 + sidx;
            // stage st's weights have landed (this wave's pieces: everything but what the previous stage's block requested), and
            // every LDS read this wave issued is complete ...
            constexpr int newer = WPER + ((sidx + 8) % 9 < 5 ? XPS : 0);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(newer) : "memory");
            __builtin_amdgcn_s_barrier();                     // ... for every wave: the slot of stage st-1 may be refilled
            asm volatile("" ::: "memory");
            if constexpr (sidx < 5) {
#pragma unroll
                for (int q = 0; q < XPS; ++q) stage_x(ch + 1, XPS * sidx + q);
            }
            stage_w(st + 2);
            const uint32_t xb = lds0 + XOFF + (ch & 1) * SXBUF, wbase = lds0 + WOFF + (st % 3) * WTAP;
            static_for<0, 4>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                constexpr int u = sidx * 4 + ks;                              // unit within the chunk body
                constexpr int um = (u + 36 - SPD) % 36;                       // the unit whose MFMAs run now (SPD behind)
                constexpr int mj = (um / 4) % 3, mks = um % 4, mslot = um % (SPD + 1), lslot = u % (SPD + 1);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    bf16x8 xf;
                    if constexpr (mj == 0) xf = XC[mks][i];
                    else if constexpr (mj == 1) xf = shift_frag<0>(XC[mks][i], mask_l);
                    else xf = shift_frag<1>(XC[mks][i], mask_r);
#pragma unroll
                    for (int jn = 0; jn < NI; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FW[mslot][jn], xf, acc[i][jn], 0, 0, 0);
                }
#pragma unroll
                for (int jn = 0; jn < NI; ++jn) FW[lslot][jn] = lds_b128s(wbase + wb[jn] + ((ks * 32 + half16) ^ wsw[jn]));
                if constexpr (j == 0) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int hp = hp0[i] + ky * a.W;
                        XC[ks][i] = lds_b128s(xb + hp * 128 + ((ks * 32 + half16) ^ (((hp >> 1) & 7) * 16)));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    static_for<0, SPD>([&](auto qc) {                        // the last SPD units: right tap column of the last row, k-steps 2, 3
        constexpr int um = 36 - SPD + decltype(qc)::value, mks = um % 4, mslot = um % (SPD + 1);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const bf16x8 xf = shift_frag<1>(XC[mks][i], mask_r);
#pragma unroll
            for (int jn = 0; jn < NI; ++jn)
                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FW[mslot][jn], xf, acc[i][jn], 0, 0, 0);
        }
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped re-fetches must not outlive the workgroup's LDS

    // ---- epilogue: lane = pixel (l & 31), register quad rq = channels 8*rq + 4*(l >> 5) .. +3
    f32x4 bq[NI][4];
#pragma unroll
    for (int jn = 0; jn < NI; ++jn)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int col = n0 + wn * (32 * NI) + jn * 32 + 8 * rq + 4 * (l >> 5);
            bq[jn][rq] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + min(col, a.Nc - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const size_t m = (size_t)m0 + wm * (32 * MI) + i * 32 + (l & 31);
        const size_t mc = min(m, (size_t)Mtot - 1);
        f32x4 v[NI][4];
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                v[jn][rq] = f32x4{acc[i][jn][4 * rq], acc[i][jn][4 * rq + 1], acc[i][jn][4 * rq + 2], acc[i][jn][4 * rq + 3]} + bq[jn][rq];
        if (a.res) {
#pragma unroll
            for (int jn = 0; jn < NI; ++jn)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int col = min(n0 + wn * (32 * NI) + jn * 32 + 8 * rq + 4 * (l >> 5), a.Nc - 4);
                    v[jn][rq] += *reinterpret_cast<const f32x4*>(a.res + mc * a.ldr + col);
                }
        }
        if (a.accumulate) {
#pragma unroll
            for (int jn = 0; jn < NI; ++jn)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int col = min(n0 + wn * (32 * NI) + jn * 32 + 8 * rq + 4 * (l >> 5), a.Nc - 4);
                    if constexpr (OUT16) {
                        const u32x2 o = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.y) + mc * a.ldy + col);
                        v[jn][rq] += f32x4{__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u),
                                           __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
                    } else {
                        v[jn][rq] += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.y) + mc * a.ldy + col);
                    }
                }
        }
        if (m >= (size_t)Mtot) continue;
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int col = n0 + wn * (32 * NI) + jn * 32 + 8 * rq + 4 * (l >> 5);
                if (col >= a.Nc) continue;
                if constexpr (OUT16)
                    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.y) + m * a.ldy + col) =
                        u32x2{pack_bf16(v[jn][rq].x, v[jn][rq].y), pack_bf16(v[jn][rq].z, v[jn][rq].w)};
                else
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + m * a.ldy + col) = v[jn][rq];
            }
    }
}

bool shift_geom(const MiConvDesc* d, int* TH, int* TI) {
    const int W = d->OW, H = d->OH;
    if (W != 8 && W != 16 && W != 32) return false;          // 32-pixel MFMA blocks must be whole image rows
    const int rows = SBM / W;
    if (rows <= H) { if (H % rows) return false; *TH = rows; *TI = 1; }
    else { if (rows % H) return false; *TH = H; *TI = rows / H; if ((long)d->N % *TI) return false; }
    return *TI * (*TH + 2) * W <= MAXXP;
}

bool shift_ok(const MiConvDesc* d, int* TH, int* TI) {
    if (d->KH != 3 || d->KW != 3 || d->pad != 1 || d->stride != 1 || d->mode != 1) return false;
    if (d->IH != d->OH || d->IW != d->OW) return false;
    if (d->K % 64 || d->K1 % 64 || d->Nc % 4 || d->ldx % 8 || (d->K1 != d->K && d->ldx2 % 8)) return false;
    if (((long)d->N * d->OH * d->OW) % SBM) return false;
    return shift_geom(d, TH, TI);
}

// 128-channel workgroups unless that leaves CUs idle (the 8x8 level at batch 128: 32 pixel tiles)
int shift_ni(const MiConvDesc* d) {
    static const int force = (int)mi_knob("MI_SHIFT_NI", 0);
    if (force == 1 || force == 2) return force;
    const long mt = (long)d->N * d->OH * d->OW / SBM;
    return mt * ((d->Nc + 127) / 128) >= 200 ? 2 : 1;
}

}  // namespace

extern "C" int mi_conv3x3_shift_supported(const MiConvDesc* d) {
    int th, ti;
    return (d && shift_ok(d, &th, &ti)) ? 1 : 0;
}
// profiling attribution: conv_shift_kernel<ni, out_bf16>
extern "C" int mi_conv3x3_shift_tile(const MiConvDesc* d, int* ni) {
    int th, ti;
    MI_REQUIRE(d && ni && shift_ok(d, &th, &ti), "descriptor not supported by the shift conv kernel");
    *ni = shift_ni(d);
    return 0;
}

// x / x2: bf16 tensors (pixel strides in elements, % 8 == 0); w: bf16 [ky][kx][Nc][K]; d->transposed = 1 -> flipped taps (data
// gradient); out_bf16: y is written as bf16 (else fp32).  bias / residual fp32, d->accumulate: y += result.
extern "C" int mi_conv3x3_shift(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16, const float* bias,
                                const float* residual, void* y, int out_bf16, void* stream) {
    MI_REQUIRE(d && x && w_nk_bf16 && y, "null argument");
    ShiftConvArgs a;
    MI_REQUIRE(shift_ok(d, &a.TH, &a.TI), "descriptor not supported by the shift conv kernel (use mi_conv3x3_bf16w_io)");
    MI_REQUIRE(d->K1 == d->K || x2, "two-source split without x2");
    MI_REQUIRE((((uintptr_t)x | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)w_nk_bf16) & 15) == 0, "operands must be 16-byte aligned");
    a.x = (const uint16_t*)x; a.x2 = (const uint16_t*)(x2 ? x2 : x); a.w = (const uint16_t*)w_nk_bf16; a.bias = bias; a.res = residual; a.y = y;
    a.N = d->N; a.H = d->OH; a.W = d->OW; a.K = d->K; a.Nc = d->Nc; a.K1 = d->K1; a.ldx = d->ldx; a.ldx2 = x2 ? d->ldx2 : d->ldx;
    a.ldy = d->ldy; a.ldr = d->ldr; a.accumulate = d->accumulate; a.flip = d->transposed ? 1 : 0;
    a.tiles_per_img = a.TI > 1 ? 1 : a.H / a.TH;
    a.XP = a.TI * (a.TH + 2) * a.W;
    a.xmap = a.TI == 1 && a.tiles_per_img > 1 && a.N % 8 == 0;
    const int ni = shift_ni(d);
    const int BN = 64 * ni;
    dim3 grid((unsigned)((long)d->N * d->OH * d->OW / SBM), (unsigned)((d->Nc + BN - 1) / BN));
    a.qmap = 0; a.gx = (int)grid.x; a.gy = (int)grid.y;
    static const int q_env = (int)mi_knob("MI_SHIFT_PQ", 1);
    if (q_env && !a.xmap && a.gy > 1 && a.gy % 2 == 0 && a.gx % 4 == 0) { a.qmap = 2; grid = dim3(grid.x * grid.y, 1, 1); }
    const size_t lds = (size_t)2 * SXBUF + (size_t)3 * BN * 128;
    static const int wm_env = (int)mi_knob("MI_SHIFT_WM", 4);
    hipStream_t st = (hipStream_t)stream;
#define MI_SHIFT_GO(WMV, NIV, O16) do { \
        static bool once_ = [] { (void)hipFuncSetAttribute((const void*)conv_shift_kernel<WMV, NIV, O16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return true; }(); \
        (void)once_; \
        hipLaunchKernelGGL((conv_shift_kernel<WMV, NIV, O16>), grid, dim3(128 * WMV), lds, st, a); } while (0)
    if (wm_env == 2) {
        if (ni == 2) { if (out_bf16) MI_SHIFT_GO(2, 2, true); else MI_SHIFT_GO(2, 2, false); }
        else { if (out_bf16) MI_SHIFT_GO(2, 1, true); else MI_SHIFT_GO(2, 1, false); }
    } else {
        if (ni == 2) { if (out_bf16) MI_SHIFT_GO(4, 2, true); else MI_SHIFT_GO(4, 2, false); }
        else { if (out_bf16) MI_SHIFT_GO(4, 1, true); else MI_SHIFT_GO(4, 1, false); }
    }
#undef MI_SHIFT_GO
    MI_LAUNCH_CHECK();
    return 0;
}
