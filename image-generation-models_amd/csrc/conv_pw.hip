// 3x3 / stride 1 / pad 1 convolution (forward and data gradient) for bf16-stored activations -- wave-PRIVATE weight streams.
//     Y[p][co] (+)= bias[co] + res[p][co] + sum_{ky,kx,ci} X[p + (ky-1, kx-1)][ci] * W[ky][kx][co][ci]
// (Block's Conv2d(dim, dim_out, 3, padding=1), reference src/models/ddpm.py:116, and its input gradient.)
//
// What bounded conv3x3_halo.hip / conv_dma.hip / conv_shift.hip: every tap (16 MFMAs per wave) ended in a
// workgroup barrier because the weight tile of a tap is shared by the waves of a workgroup, and a tile's prologue (its first
// activation rows arriving from HBM) and epilogue (its output stores) overlapped nothing: one workgroup per CU.  Here
//   * a workgroup = 4 waves = 128 output pixels (whole image rows) x 128 output channels; wave w owns ALL 128 pixels of the
//     channels [32w, 32w + 32): four 32x32 MFMA tiles, 64 accumulator registers;
//   * the weights a wave needs are needed by no other wave of the workgroup, so each wave streams its own: the bf16 copy is kept in
//     MFMA-FRAGMENT order ([tap][co / 32][ci / 16][lane][8], written by mi_pack_weights_bf16), a fragment is 1 KB contiguous in
//     memory, goes to the wave's private 8 KB ring in LDS with ONE global_load_lds_dwordx4 (lane-linear source and destination) and
//     is read back with one conflict-free ds_read_b128; the wave waits for its own DMA with a counted s_waitcnt vmcnt(N) -- no
//     barrier.  Seven fragments (28 MFMAs) are in flight per wave;
//   * the activation tile (TH + 2 rows x 64 channels, XOR-swizzled through the DMA source address) is shared and double-buffered per
//     64-channel chunk: ONE barrier per chunk (144 MFMAs per wave) instead of nine;
//   * only the centre tap column's activation fragments are read from LDS (once per tap row and 16-channel step, 4 reads); the left /
//     right columns are one-lane DPP shifts of them (conv_shift.hip): 7 LDS fragment reads per 12 MFMAs;
//   * 80 KB of LDS and <= 128 registers per lane... two workgroups per CU: one computes while the other waits for its first rows or
//     drains its stores, and every SIMD has two independent instruction streams.
#include "tr_common.h"

namespace {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) uint32_t g_zero_page3[64];     // 256 zero bytes: the rows above / below an image

struct PwArgs {
    const uint16_t* x; const uint16_t* x2; const uint16_t* w; const float* bias; const float* res; void* y;
    int N, H, W, K, Nc, K1, ldx, ldx2, ldy, ldr, accumulate, flip;
    int TH, TI, XP, tiles_per_img, xmap;
    int qmap, gx, gy;    // 1-D launch of gx pixel tiles x gy channel tiles: XCD = (pixel group, channel group) of a (8 / qmap) x qmap split
};

constexpr int PBM = 128;                        // output pixels per workgroup
constexpr int PCK = 64;                         // channels per chunk
constexpr int PXP = 192;                        // tile pixels incl. the row above and below: 6 x 32, 10 x 16, 2 images x 10 x 8
constexpr int PXBUF = PXP * 128;                // one chunk of the activation tile
constexpr int PXPW = PXP / 8 / 4;               // activation DMA instructions per wave and chunk
constexpr int PR = 8;                           // weight ring of a wave: fragments of 1 KB
constexpr int PWOFF = 2 * PXBUF;                // LDS: [2 activation buffers][4 waves x PR KB]
constexpr int PLDS = PWOFF + 4 * PR * 1024;     // 80 KB: two workgroups per CU
constexpr int PDD = 7;                          // a fragment's DMA is issued PDD units before the unit that multiplies it
constexpr int PRD = 2;                          // ... and read into registers PRD units before
static_assert(PWOFF % 8192 == 0 && PR == 8, "ring slots are addressed by toggling bit 12");

// LDS-DMA with a scalar base: 16 bytes per lane from sbase + voff to LDS byte address lds_dst (wave-uniform) + 16 * lane
__device__ __forceinline__ void glds16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    // the base IS wave-uniform; readfirstlane makes that provable where hipcc moved its computation to the vector ALU
    const uint64_t p = (uint64_t)(uintptr_t)sbase;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    const uint64_t q = ((uint64_t)hi << 32) | lo;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(q), "s"(dst) : "memory");
}

__device__ __forceinline__ bf16x8 lds_b128p(uint32_t addr) {
    typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
    return *(lds_bf16x8*)(uintptr_t)addr;
}

// fragment of the tap column to the left (DIR = 0: lane l takes lane l-1) / right (DIR = 1: lane l+1) of the centre column
// One v_and_b32_dpp per register: the DPP operand is the neighbour lane's value (0 beyond the wave's ends), the mask supplies the
// zero padding left / right of an image row.  (The builtin form, v_mov_b32_dpp + a select, is two instructions per register.)
// The shifted registers come straight from ds_read_b128, never from a VALU instruction (no VALU -> DPP wait states needed).
template <int DIR> __device__ __forceinline__ bf16x8 pw_shift(const bf16x8& c, uint32_t mask) {
    const u32x4 v = __builtin_bit_cast(u32x4, c);
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t r;
        if constexpr (DIR == 0)
            asm("v_and_b32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v[q]), "v"(mask));
        else
            asm("v_and_b32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v[q]), "v"(mask));
        o[q] = r;
    }
    // VALU write -> MFMA operand read needs two wait states; hipcc does not see the VALU instruction inside the statements above
    asm("s_nop 1" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]));
    return __builtin_bit_cast(bf16x8, o);
}

// Unit s (0..35) of a chunk body = (tap row ky, 16-channel step ks, tap column j in the order centre, left, right).
// Activation piece i of the NEXT chunk is requested in unit 3i + 1.
constexpr bool pw_is_x(int s) { return s >= 0 && s % 3 == 1 && s / 3 < PXPW; }
// DMA instructions issued after the request of the fragment that unit s reads (unit s + PRD's fragment), up to the start of unit s
// (a unit requests its fragment first, then its activation piece)
constexpr int pw_newer(int s) {
    int n = pw_is_x(s - (PDD - PRD)) ? 1 : 0;
    for (int q = s - (PDD - PRD - 1); q < s; ++q) n += 1 + (pw_is_x(q) ? 1 : 0);
    return n;
}

// ABL (profiling builds only, -DMI_PW_ABL_BUILD): 1 no fragment DMA in the main loop, 2 no activation DMA in the main loop, 4 no stores
template <bool OUT16, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_pw_kernel(const PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    int bx = blockIdx.x;
    if (a.xmap) {        // an image's row tiles share rows: keep them on one XCD (ids xcd + 8*slot -> image xcd + 8*m)
        const int xcd = bx & 7, slot = bx >> 3;
        bx = (xcd + 8 * (slot / a.tiles_per_img)) * a.tiles_per_img + slot % a.tiles_per_img;
    }
    int by = blockIdx.y;
    if (a.qmap) {        // XCD = (pixel group, channel group): an XCD's L2 holds gy / Q of the weight tiles (conv_shift.hip)
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3, Q = a.qmap, P = 8 / Q;
        const int ppx = a.gx / P, cpq = a.gy / Q;
        bx = (xcd / Q) * ppx + slot / cpq; by = (xcd % Q) * cpq + slot % cpq;
    }
    const int m0 = bx * PBM, n0 = by * 128;
    const int TH2 = a.TH + 2;
    const int lw = 31 - __builtin_clz(a.W), lth = 31 - __builtin_clz(a.TH);      // W and TH are powers of two
    const int nchunks = a.K / PCK;
    const int NB = a.Nc >> 5, KQ = a.K >> 4;
    const bool live = n0 + 32 * wv < a.Nc;                    // a ragged last channel tile: the wave computes a copy of the last block
    const int nb = min((n0 >> 5) + wv, NB - 1);

    // ---- activation DMA pieces: piece p = wv + 4i covers tile pixels 8p .. 8p+7 (tile pixel hp = (image ti, row hy = y+1, column x));
    //      lane -> pixel lane >> 3, stored 16-byte position lane & 7 holds channel chunk (lane & 7) ^ ((hp >> 1) & 7)
    int xpix[PXPW], xcol[PXPW];
    {
        int img0, y0;
        if (a.TI > 1) { img0 = bx * a.TI; y0 = 0; }
        else { img0 = bx / a.tiles_per_img; y0 = (bx % a.tiles_per_img) * a.TH; }
#pragma unroll
        for (int i = 0; i < PXPW; ++i) {
            const int hp = 8 * (wv + 4 * i) + (l >> 3);
            int v = -1;
            if (hp < a.XP) {
                const int row = hp >> lw, x = hp & (a.W - 1);
                const int ti = row >= TH2 ? 1 : 0, hy = row - ti * TH2;              // at most two images per tile
                const int iy = y0 + hy - 1, img = img0 + ti;
                if (iy >= 0 && iy < a.H && img < a.N) v = (img * a.H + iy) * a.W + x;
            }
            xpix[i] = v;
            xcol[i] = ((l & 7) ^ ((hp >> 1) & 7)) * 8;
        }
    }
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_page3);
    auto stage_x = [&](int ch, int i) {                      // piece i of this wave of chunk ch's rows -> buffer ch & 1
        const int cc0 = min(ch, nchunks - 1) * PCK;
        const bool second = cc0 >= a.K1;
        const uint16_t* src = second ? a.x2 : a.x;
        const int ld = second ? a.ldx2 : a.ldx, cc = second ? cc0 - a.K1 : cc0;
        const uint16_t* p = xpix[i] >= 0 ? src + (size_t)xpix[i] * ld + cc + xcol[i] : zero + (l & 7) * 8;
        glds16(p, lds0 + (ch & 1) * PXBUF + (wv + 4 * i) * 1024);
    };

    // ---- weight stream of this wave: fragment (tap, nb, kq) = 1 KB at ((tap * NB + nb) * KQ + kq) * 1024 bytes
    const uint8_t* wsrc = reinterpret_cast<const uint8_t*>(a.w) + (size_t)nb * KQ * 1024;
    const uint32_t tap_bytes = (uint32_t)a.Nc * a.K * 2;
    const uint32_t wl16 = l * 16;
    const uint32_t wring = lds0 + PWOFF + wv * (PR * 1024);
    // unit s of chunk ch (s may run past 35 into the next chunk): request its fragment into ring slot (36 ch + s) % 8
    auto stage_w = [&](int ch, auto sc) {
        constexpr int s0 = decltype(sc)::value, over = s0 >= 36 ? 1 : 0, s = s0 - 36 * over;
        constexpr int ky = s / 12, ks = (s / 3) % 4, j = s % 3, tap = ky * 3 + (j == 0 ? 1 : (j == 1 ? 0 : 2));
        const int chc = min(ch + over, nchunks - 1);         // past the end: re-fetch (keeps the DMA counts static)
        const uint32_t ph = ((ch + over) & 1) * 4096;
        const uint32_t off = (uint32_t)(a.flip ? 8 - tap : tap) * tap_bytes + (uint32_t)(chc * 4 + ks) * 1024;
        glds16s(wsrc + off, wl16, wring + (((s & 7) * 1024) ^ ph));
    };

    // ---- fragment addressing.  Activations (MFMA "B" operand): lane -> pixel (l & 31) of the wave's i-th 32-pixel block (whole
    //      image rows), 8-channel piece 2*ks + (l >> 5); weights ("A"): lane -> its own 16 bytes of the fragment.
    uint32_t xa[4][3];                                       // byte offset of (block i, tap row ky), 16-channel step 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 32 + (l & 31);
        const int tx = r & (a.W - 1), q = r >> lw;
        const int ty = q & (a.TH - 1), ti = q >> lth;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int hp = (ti * TH2 + ty + ky) * a.W + tx;
            xa[i][ky] = lds0 + hp * 128 + (((l >> 5) * 16) ^ (((hp >> 1) & 7) * 16));
        }
    }
    const int xin = l & 31 & (a.W - 1);
    const uint32_t mask_l = xin == 0 ? 0u : ~0u, mask_r = xin == a.W - 1 ? 0u : ~0u;     // zero padding left / right of the row
    const uint32_t wrd0 = wring + wl16;

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 XC[2][4];                                         // centre-column fragments of (tap row, 16-channel step): this group's, the next one's
    bf16x8 FW[PRD + 1];                                      // weight fragments, PRD units ahead

    // ---- prologue: the first chunk's rows, the first PDD fragments
#pragma unroll
    for (int i = 0; i < PXPW; ++i) stage_x(0, i);
    static_for<0, PDD>([&](auto sc) { stage_w(0, sc); });
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PDD - PRD) : "memory");     // the rows and fragments 0 .. PRD-1 have landed (this wave's)
    __builtin_amdgcn_s_barrier();                                          // ... every wave's
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) XC[0][i] = lds_b128p(xa[i][0]);
    static_for<0, PRD>([&](auto sc) { FW[decltype(sc)::value] = lds_b128p(wrd0 + decltype(sc)::value * 1024); });

    static_assert(36 % (PRD + 1) == 0, "fragment register slots line up across chunks");
    for (int ch = 0; ch < nchunks; ++ch) {
        // ring slot of unit s of this chunk: (36 ch + s) % 8 = (s & 7) with bit 2 toggled in odd chunks
        const uint32_t wsame = wrd0 + (ch & 1) * 4096, wflip = wrd0 + 4096 - (ch & 1) * 4096;
        const uint32_t xcur = (ch & 1) * PXBUF, xnxt = PXBUF - xcur;
        static_for<0, 36>([&](auto sc) {
            constexpr int s = decltype(sc)::value, g = s / 3, ky = g / 4, ks = g % 4, j = s % 3;
            // the fragment of unit s + PRD has landed ...
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(pw_newer(s)) : "memory");
            {
                constexpr int sr = s + PRD, over = sr >= 36 ? 1 : 0, srr = sr - 36 * over;
                FW[sr % (PRD + 1)] = lds_b128p((((srr & 4) != 0) != (over != 0) ? wflip : wsame) + (srr & 3) * 1024);
            }
            if constexpr (j == 0) {                          // the next group's centre-column fragments
                if constexpr (g == 11) {
                    // chunk boundary: every wave has read all it needs of this chunk's rows and has its pieces of the next chunk's
                    // (their requests are older than the fragment just waited for)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < 4; ++i) XC[0][i] = lds_b128p(xa[i][0] + xnxt);
                } else {
                    constexpr int gn = g + 1, kyn = gn / 4, ksn = gn % 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) XC[gn & 1][i] = lds_b128p((xa[i][kyn] ^ (ksn * 32)) + xcur);
                }
            }
            // ring slot of unit s - 1 is free (its fragment is in registers since the previous unit's MFMAs)
            if constexpr (!(ABL & 1)) stage_w(ch, std::integral_constant<int, s + PDD>{});
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (pw_is_x(s) && !(ABL & 2)) stage_x(ch + 1, s / 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x8 xf;
                if constexpr (j == 0) xf = XC[g & 1][i];
                else if constexpr (j == 1) xf = pw_shift<0>(XC[g & 1][i], mask_l);
                else xf = pw_shift<1>(XC[g & 1][i], mask_r);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FW[s % (PRD + 1)], xf, acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped re-fetches must not outlive the workgroup's LDS
    if constexpr ((ABL & 4) != 0) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) v += acc[i][r];
        if (v == 123.456f) reinterpret_cast<float*>(a.y)[t] = v;
        return;
    }

    // ---- epilogue.  A lane holds 4 consecutive channels of ONE pixel per register quad: stored from here a store instruction would
    //      write 16-byte pieces of 32 different rows (a [131072][128] bf16 tensor written that way takes 13 us against 7 us in whole
    //      rows, tools/proto/store_probe.hip).  The fp32 tile goes through LDS instead (the whole 80 KB is free now): [128 pixels][128
    //      channels] fp32, 16-byte chunk c of pixel p at position c ^ (p & 31) (conflict-free both ways), and leaves as whole rows --
    //      16 lanes x 8 channels = one pixel's 256 (bf16) / 512 (fp32) contiguous bytes; bias, residual and the accumulate operand are
    //      applied on the way out, read in the same row-contiguous pattern.
    __builtin_amdgcn_s_barrier();                            // every wave is done with the activation buffers, every DMA has landed
    asm volatile("" ::: "memory");
    if (live) {
        typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = i * 32 + (l & 31);
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int ck = 8 * wv + 2 * rq + (l >> 5);
                *(lds_f32x4*)(uintptr_t)(lds0 + p * 512 + ((ck ^ (p & 31)) << 4)) =
                    f32x4{acc[i][4 * rq], acc[i][4 * rq + 1], acc[i][4 * rq + 2], acc[i][4 * rq + 3]};
            }
        }
    }
    __syncthreads();
    {
        typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
        const int j = t & 15, col = n0 + 8 * j;              // this thread's 8 channels
        if (col >= a.Nc) return;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (a.bias) { b0 = *reinterpret_cast<const f32x4*>(a.bias + col); b1 = *reinterpret_cast<const f32x4*>(a.bias + col + 4); }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int p = it * 16 + (t >> 4);
            const size_t m = (size_t)m0 + p;
            f32x4 v0 = *(lds_f32x4*)(uintptr_t)(lds0 + p * 512 + (((2 * j) ^ (p & 31)) << 4)) + b0;
            f32x4 v1 = *(lds_f32x4*)(uintptr_t)(lds0 + p * 512 + (((2 * j + 1) ^ (p & 31)) << 4)) + b1;
            if (a.res) {
                v0 += *reinterpret_cast<const f32x4*>(a.res + m * a.ldr + col);
                v1 += *reinterpret_cast<const f32x4*>(a.res + m * a.ldr + col + 4);
            }
            if constexpr (OUT16) {
                uint16_t* yp = reinterpret_cast<uint16_t*>(a.y) + m * a.ldy + col;
                if (a.accumulate) {
                    const u32x4 o = *reinterpret_cast<const u32x4*>(yp);
                    v0 += f32x4{__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u),
                                __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
                    v1 += f32x4{__uint_as_float(o.z << 16), __uint_as_float(o.z & 0xffff0000u),
                                __uint_as_float(o.w << 16), __uint_as_float(o.w & 0xffff0000u)};
                }
                *reinterpret_cast<u32x4*>(yp) = u32x4{pack_bf16(v0.x, v0.y), pack_bf16(v0.z, v0.w), pack_bf16(v1.x, v1.y), pack_bf16(v1.z, v1.w)};
            } else {
                float* yp = reinterpret_cast<float*>(a.y) + m * a.ldy + col;
                if (a.accumulate) { v0 += *reinterpret_cast<const f32x4*>(yp); v1 += *reinterpret_cast<const f32x4*>(yp + 4); }
                *reinterpret_cast<f32x4*>(yp) = v0;
                *reinterpret_cast<f32x4*>(yp + 4) = v1;
            }
        }
    }
}

bool pw_geom(const MiConvDesc* d, int* TH, int* TI) {
    const int W = d->OW, H = d->OH;
    if (W != 8 && W != 16 && W != 32) return false;          // 32-pixel MFMA blocks must be whole image rows
    const int rows = PBM / W;
    if (rows <= H) { if (H % rows) return false; *TH = rows; *TI = 1; }
    else { if (rows % H) return false; *TH = H; *TI = rows / H; if ((long)d->N % *TI) return false; }
    if ((*TH & (*TH - 1)) || *TI > 2) return false;            // the kernel's index arithmetic: shifts, at most two images per tile
    return *TI * (*TH + 2) * W <= PXP;
}

bool pw_ok(const MiConvDesc* d, int* TH, int* TI) {
    if (d->KH != 3 || d->KW != 3 || d->pad != 1 || d->stride != 1 || d->mode != 1) return false;
    if (d->IH != d->OH || d->IW != d->OW) return false;
    if (d->K % 64 || d->K1 % 64 || d->Nc % 32 || d->ldx % 8 || (d->K1 != d->K && d->ldx2 % 8)) return false;
    if (((long)d->N * d->OH * d->OW) % PBM) return false;
    if ((long)d->Nc * d->K * 2 * 9 >= (1L << 31)) return false;      // 32-bit fragment offsets
    return pw_geom(d, TH, TI);
}

}  // namespace

extern "C" int mi_conv3x3_pw_supported(const MiConvDesc* d) {
    int th, ti;
    return (d && pw_ok(d, &th, &ti)) ? 1 : 0;
}

// x / x2: bf16 tensors (pixel strides in elements, % 8 == 0); w: bf16 weights in MFMA-fragment order [tap][Nc / 32][K / 16][64][8]
// (mi_pack_weights_bf16's wfq for the forward conv, wdq with d->transposed = 1 -> flipped taps for the data gradient);
// out_bf16: y is written as bf16 (else fp32).  bias / residual fp32, d->accumulate: y += result.
extern "C" int mi_conv3x3_pw(const MiConvDesc* d, const void* x, const void* x2, const void* w_frag_bf16, const float* bias,
                             const float* residual, void* y, int out_bf16, void* stream) {
    MI_REQUIRE(d && x && w_frag_bf16 && y, "null argument");
    PwArgs a;
    MI_REQUIRE(pw_ok(d, &a.TH, &a.TI), "descriptor not supported by the private-weight-stream conv kernel (use mi_conv3x3_bf16w_io)");
    MI_REQUIRE(d->K1 == d->K || x2, "two-source split without x2");
    MI_REQUIRE((((uintptr_t)x | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)w_frag_bf16) & 15) == 0, "operands must be 16-byte aligned");
    MI_REQUIRE(d->ldy % 4 == 0 && (!residual || d->ldr % 4 == 0), "output / residual pixel strides must be multiples of 4");
    a.x = (const uint16_t*)x; a.x2 = (const uint16_t*)(x2 ? x2 : x); a.w = (const uint16_t*)w_frag_bf16; a.bias = bias; a.res = residual; a.y = y;
    a.N = d->N; a.H = d->OH; a.W = d->OW; a.K = d->K; a.Nc = d->Nc; a.K1 = d->K1; a.ldx = d->ldx; a.ldx2 = x2 ? d->ldx2 : d->ldx;
    a.ldy = d->ldy; a.ldr = d->ldr; a.accumulate = d->accumulate; a.flip = d->transposed ? 1 : 0;
    a.tiles_per_img = a.TI > 1 ? 1 : a.H / a.TH;
    a.XP = a.TI * (a.TH + 2) * a.W;
    a.xmap = a.TI == 1 && a.tiles_per_img > 1 && a.N % 8 == 0;
    dim3 grid((unsigned)((long)d->N * d->OH * d->OW / PBM), (unsigned)((d->Nc + 127) / 128));
    a.qmap = 0; a.gx = (int)grid.x; a.gy = (int)grid.y;
    if (!a.xmap && a.gy > 1 && a.gy % 2 == 0 && a.gx % 4 == 0) { a.qmap = 2; grid = dim3(grid.x * grid.y, 1, 1); }
    hipStream_t st = (hipStream_t)stream;
    size_t lds = PLDS;
#ifdef MI_PW_ABL_BUILD
    static const int abl = [] { const char* e = getenv("MI_PW_ABL"); return e ? atoi(e) : 0; }();
    if (abl & 8) lds = 100 * 1024;            // one workgroup per CU
#define MI_PW_GO(O16, A) do { \
        static bool once_ = [] { (void)hipFuncSetAttribute((const void*)conv_pw_kernel<O16, A>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return true; }(); \
        (void)once_; \
        hipLaunchKernelGGL((conv_pw_kernel<O16, A>), grid, dim3(256), lds, st, a); } while (0)
    switch (abl & 7) {
        case 1: if (out_bf16) MI_PW_GO(true, 1); else MI_PW_GO(false, 1); break;
        case 2: if (out_bf16) MI_PW_GO(true, 2); else MI_PW_GO(false, 2); break;
        case 3: if (out_bf16) MI_PW_GO(true, 3); else MI_PW_GO(false, 3); break;
        case 4: if (out_bf16) MI_PW_GO(true, 4); else MI_PW_GO(false, 4); break;
        case 7: if (out_bf16) MI_PW_GO(true, 7); else MI_PW_GO(false, 7); break;
        default: if (out_bf16) MI_PW_GO(true, 0); else MI_PW_GO(false, 0); break;
    }
#undef MI_PW_GO
#else
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)conv_pw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
        (void)hipFuncSetAttribute((const void*)conv_pw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
        return true;
    }();
    (void)once;
    if (out_bf16) hipLaunchKernelGGL((conv_pw_kernel<true>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((conv_pw_kernel<false>), grid, dim3(256), lds, st, a);
#endif
    MI_LAUNCH_CHECK();
    return 0;
}
