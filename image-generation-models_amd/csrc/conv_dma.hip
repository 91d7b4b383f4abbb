// 3x3 / stride 1 / pad 1 convolution (forward and data gradient) for bf16-STORED activations, staged entirely by LDS-DMA:
//     Y[p][co] (+)= bias[co] + res[p][co] + sum_{ky,kx,ci} X[p + (ky-1, kx-1)][ci] * W[ky][kx][co][ci]
// (Block's Conv2d(dim, dim_out, 3, padding=1), reference src/models/ddpm.py:116, and its aten::convolution_backward input
// gradient: the same sum with the taps flipped and the [tap][ci][co] weight copy.)
//
// Compared with conv3x3_halo.hip (global -> registers -> v_cvt/v_and -> ds_write, one barrier per tap, 34 % MFMA-busy) nothing but
// MFMAs and LDS reads is left in the loop:
//   * the halo tile of a 64-channel chunk ([TH+2][W+2] pixels of one or two images x 128 bytes) and the weight tiles go from L2
//     into LDS by global_load_lds_dwordx4 -- no VGPRs, no VALU, no ds_write.  The LDS image is chosen through the per-lane SOURCE
//     address: a pixel's (or an output channel's) eight 16-byte k-chunks are stored at position chunk ^ ((row >> 1) & 7), which
//     makes the ds_read_b128 fragment reads of 16 consecutive rows conflict-free.  Out-of-image halo pixels are fetched from a
//     zero page, so padding costs nothing in the loop.
//   * a stage is one tap ROW (three taps of a 64-channel chunk = 48 MFMAs per wave) between barriers; the weights of the next
//     stage and the halo tile of the next chunk are in flight meanwhile (two weight slots, two halo buffers); fragments are
//     fetched two (tap, k-step) units ahead of their MFMAs through a ring of register sets, pinned with sched_barrier.
// A workgroup = 4 waves (one per SIMD) = 128 output pixels (whole image rows) x 128 output channels, a wave = 64 x 64.
// Weights are the MFMA "A" operand, so a lane ends up with 4 consecutive channels of one pixel per register quad (16-byte /
// 8-byte stores), exactly like the halo kernel's epilogue.
#include "tr_common.h"

namespace {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) uint32_t g_zero_page[64];      // 256 zero bytes: the source of out-of-image halo pixels

struct DmaConvArgs {
    const uint16_t* x; const uint16_t* x2; const uint16_t* w; const float* bias; const float* res; void* y;
    int N, H, W, K, Nc, K1, ldx, ldx2, ldy, ldr, accumulate, flip;
    int TH, TI, HP, tiles_per_img, xmap;
};

constexpr int BM = 128, BN = 128;
constexpr int MAXHP = 208;                       // halo pixels: 4 rows of 32 (6 x 34 = 204), 8 of 16 (180), two 8x8 images (200)
// CK = channels per chunk.  64: one workgroup per CU (148 KB of LDS).  32: half the LDS (77 KB), so TWO independent workgroups
// share a CU -- while one waits at its stage barrier, fetches its first tile or stores its outputs, the other one's waves keep the
// matrix pipe busy (the 8-wave single-workgroup kernels stall both waves of a SIMD at every barrier).
// TPS = taps per stage (between two barriers).  3: a tap row, weights double-buffered 2 x 3 taps.  1: one tap, weights 2 x 1 tap --
// with CK = 32 that is 42 KB of LDS, THREE workgroups per CU (3 waves per SIMD), fragments fetched one unit ahead.
template <int CK, int TPS = 3> struct DmaShape {
    static constexpr int ROWB = CK * 2;              // bytes of a pixel / an output channel of one tap in LDS
    static constexpr int NCH = ROWB / 16;            // 16-byte chunks per row
    static constexpr int PPI = 1024 / ROWB;          // rows one wave-wide DMA instruction covers
    static constexpr int SH = CK == 64 ? 1 : 2;      // chunk position = chunk ^ ((row >> SH) & (NCH - 1)): 16 consecutive rows, one chunk -> all banks
    static constexpr int XBUF = MAXHP * ROWB;        // one halo buffer
    static constexpr int WSTG = TPS * BN * ROWB;     // weights of one stage: TPS taps x 128 output channels x CK input channels
    static constexpr int XOFF = 0, WOFF = 2 * XBUF, PIXOFF = WOFF + 2 * WSTG;     // + int[MAXHP] source pixel of each halo pixel
    static constexpr int KSN = CK / 16;              // MFMA k-steps per tap
    static constexpr int NU = TPS * KSN;             // (tap, k-step) units per stage
    static constexpr int PD = TPS == 3 ? 2 : 1;      // (tap, k-step) units fetched ahead
    static constexpr int NST = 9 / TPS;              // stages per chunk
    static constexpr int WGS = CK == 64 ? 1 : (TPS == 3 ? 2 : 3);     // workgroups per CU
};

__device__ __forceinline__ bf16x8 lds_b128(uint32_t addr) {
    typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
    return *(lds_bf16x8*)(uintptr_t)addr;
}

template <int CK, bool OUT16, int TPS = 3>
__global__ __launch_bounds__(256, (DmaShape<CK, TPS>::WGS)) void conv_dma_kernel(const DmaConvArgs a) {
    using Sh = DmaShape<CK, TPS>;
    constexpr int PD = Sh::PD, NST = Sh::NST;
    constexpr int ROWB = Sh::ROWB, NCH = Sh::NCH, PPI = Sh::PPI, SH = Sh::SH, XBUF = Sh::XBUF, WSTG = Sh::WSTG;
    constexpr int XOFF = Sh::XOFF, WOFF = Sh::WOFF, PIXOFF = Sh::PIXOFF, KSN = Sh::KSN, NU = Sh::NU;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;
    int* pix = reinterpret_cast<int*>(lds_raw + PIXOFF);
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wv >> 1, wn = wv & 1;
    int bx = blockIdx.x;
    if (a.xmap) {        // an image's row tiles share halo rows: keep them on one XCD (ids xcd + 8*slot -> image xcd + 8*m)
        const int xcd = bx & 7, slot = bx >> 3;
        bx = (xcd + 8 * (slot / a.tiles_per_img)) * a.tiles_per_img + slot % a.tiles_per_img;
    }
    const int m0 = bx * BM, n0 = blockIdx.y * BN;
    const int W2 = a.W + 2, TH2 = a.TH + 2;
    const int Mtot = a.N * a.H * a.W;
    const int nchunks = a.K / CK;
    static_assert(CK == 64 || CK == 32, "chunk width");

    // ---- halo pixel -> source pixel (or -1)
    {
        int img0, y0;
        if (a.TI > 1) { img0 = bx * a.TI; y0 = 0; }
        else { img0 = bx / a.tiles_per_img; y0 = (bx % a.tiles_per_img) * a.TH; }
        for (int hp = t; hp < MAXHP; hp += 256) {
            int v = -1;
            if (hp < a.HP) {
                const int ti = hp / (TH2 * W2), rem = hp - ti * (TH2 * W2);
                const int hy = rem / W2, hx = rem - hy * W2;
                const int iy = y0 + hy - 1, ix = hx - 1, img = img0 + ti;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && img < a.N) v = (img * a.H + iy) * a.W + ix;
            }
            pix[hp] = v;
        }
    }
    __syncthreads();
    // DMA pieces of this lane.  X: instruction i of a wave covers halo pixels PPI*(wv + 4*i) .. +PPI-1, lane -> pixel lane / NCH,
    // stored chunk position lane % NCH holds channel chunk (lane % NCH) ^ swizzle(hp).
    constexpr int NXI = ((MAXHP + PPI - 1) / PPI + 3) / 4;
    int xpix[NXI], xcol[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int hp = PPI * (wv + 4 * i) + l / NCH;
        xpix[i] = hp < MAXHP ? pix[hp] : -1;
        xcol[i] = ((l % NCH) ^ ((hp >> SH) & (NCH - 1))) * 8;
    }
    // W: instruction i of a wave covers output channels PPI*(wv + 4*i) .. of one tap (BN / PPI instructions per tap)
    constexpr int NWI = BN / PPI / 4;
    int wrow[NWI], wcol[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int n = PPI * (wv + 4 * i) + l / NCH;
        wrow[i] = min(n0 + n, a.Nc - 1);
        wcol[i] = ((l % NCH) ^ ((n >> SH) & (NCH - 1))) * 8;
    }
    const size_t tap_stride = (size_t)a.Nc * a.K;
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_page);

    auto stage_x = [&](int ch) {                         // halo tile of chunk ch -> buffer ch & 1
        const int kc = ch * CK;
        const bool second = kc >= a.K1;
        const uint16_t* src = second ? a.x2 : a.x;
        const int ld = second ? a.ldx2 : a.ldx, cc = second ? kc - a.K1 : kc;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            if (PPI * (wv + 4 * i) < MAXHP) {            // wave-uniform: the last instruction slots of some waves lie past the buffer
                const uint16_t* p = xpix[i] >= 0 ? src + (size_t)xpix[i] * ld + cc + xcol[i] : zero + (l & 7) * 8;
                glds16(p, lds0 + XOFF + (ch & 1) * XBUF + (wv + 4 * i) * 1024);
            }
        }
    };
    auto stage_w = [&](int st) {                         // weights of stage st = (chunk st / NST, taps (st % NST) * TPS ..) -> slot st & 1
        const int ch = st / NST, t0 = (st - ch * NST) * TPS;
#pragma unroll
        for (int kx = 0; kx < TPS; ++kx) {
            const int tap = t0 + kx;
            const uint16_t* base = a.w + (size_t)(a.flip ? 8 - tap : tap) * tap_stride + (size_t)ch * CK;
#pragma unroll
            for (int i = 0; i < NWI; ++i)
                glds16(base + (size_t)wrow[i] * a.K + wcol[i], lds0 + WOFF + (st & 1) * WSTG + kx * (BN * ROWB) + (wv + 4 * i) * 1024);
        }
    };

    // ---- fragment addressing.  Activations (MFMA "B" operand): lane -> pixel row (l & 31) of the wave's i-th 32-pixel block,
    //      k-chunk 2*ks + (l >> 5); weights ("A"): lane -> output channel wn*64 + j*32 + (l & 31).
    int hp0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wm * 64 + i * 32 + (l & 31);
        const int tx = r % a.W, q = r / a.W;
        const int ty = q % a.TH, ti = q / a.TH;
        hp0[i] = (ti * TH2 + ty) * W2 + tx;
    }
    const int half16 = (l >> 5) * 16;
    int wb[2], wsw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = wn * 64 + j * 32 + (l & 31);
        wb[j] = n * ROWB; wsw[j] = ((n >> SH) & (NCH - 1)) * 16;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragment reads run PD (tap, k-step) units AHEAD of the MFMAs that consume them, and the barrier sits at the stage boundary
    // of the READ stream: when the reads of stage st begin, the MFMAs are still two units inside stage st-1, so the barrier + first
    // LDS latency are covered by MFMA work instead of stalling the matrix pipe at every stage.
    bf16x8 FX[PD + 1][2], FW[PD + 1][2];
#pragma unroll
    for (int q = 0; q <= PD; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { FX[q][i][e] = (__bf16)0.f; FW[q][i][e] = (__bf16)0.f; }      // the first two MFMA units add zero
        }
    stage_x(0);
    stage_w(0);
    const int nstages = nchunks * NST;
    static_assert(NU % (PD + 1) == 0, "ring slots must line up across stages");
    for (int st = 0; st < nstages; ++st) {
        // stage st's weights (and halo tile) have landed: this wave's pieces; every LDS read this wave issued is complete ...
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ... for every wave: stage st-1's slot may be refilled
        asm volatile("" ::: "memory");
        const int ch = st / NST, t0 = (st - ch * NST) * TPS;
        if (st + 1 < nstages) stage_w(st + 1);
        if (t0 == 0 && ch + 1 < nchunks) stage_x(ch + 1);
        const uint32_t xb = lds0 + XOFF + (ch & 1) * XBUF, wbase = lds0 + WOFF + (st & 1) * WSTG;
        // per tap column: the lane's two halo pixels, their byte offset and swizzle
        int xo[TPS][2], xs[TPS][2];
#pragma unroll
        for (int kx = 0; kx < TPS; ++kx)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int tap = t0 + kx;
                const int hp = hp0[i] + (tap / 3) * W2 + tap % 3;
                xo[kx][i] = hp * ROWB; xs[kx][i] = ((hp >> SH) & (NCH - 1)) * 16;
            }
        // NU units = (tap column kx, k-step ks); a unit = 2 + 2 fragment reads and 4 MFMAs
        static_for<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value, kx = u / KSN, ks = u % KSN;
            constexpr int ld_slot = u % (PD + 1), mm_slot = (u + NU - PD) % (PD + 1);      // the unit PD behind (of stage st-1 for u < PD)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FW[mm_slot][j], FX[mm_slot][i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) FX[ld_slot][i] = lds_b128(xb + xo[kx][i] + ((ks * 32 + half16) ^ xs[kx][i]));
#pragma unroll
            for (int j = 0; j < 2; ++j) FW[ld_slot][j] = lds_b128(wbase + kx * (BN * ROWB) + wb[j] + ((ks * 32 + half16) ^ wsw[j]));
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    static_for<0, PD>([&](auto qc) {                         // the last PD units
        constexpr int mm_slot = (NU - PD + decltype(qc)::value) % (PD + 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FW[mm_slot][j], FX[mm_slot][i], acc[i][j], 0, 0, 0);
    });

    // ---- epilogue: lane = pixel (l & 31), register quad rq = channels 8*rq + 4*(l >> 5) .. +3
    f32x4 bq[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int col = n0 + wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5);
            bq[j][rq] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + min(col, a.Nc - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const size_t m = (size_t)m0 + wm * 64 + i * 32 + (l & 31);
        const size_t mc = min(m, (size_t)Mtot - 1);
        f32x4 v[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                v[j][rq] = f32x4{acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]} + bq[j][rq];
        if (a.res) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int col = min(n0 + wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5), a.Nc - 4);
                    v[j][rq] += *reinterpret_cast<const f32x4*>(a.res + mc * a.ldr + col);
                }
        }
        if (a.accumulate) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int col = min(n0 + wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5), a.Nc - 4);
                    if constexpr (OUT16) {
                        const u32x2 o = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.y) + mc * a.ldy + col);
                        v[j][rq] += f32x4{__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u),
                                          __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
                    } else {
                        v[j][rq] += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.y) + mc * a.ldy + col);
                    }
                }
        }
        if (m >= (size_t)Mtot) continue;
#ifdef MI_DMA_ABL       // profiling only: no output stores (one guarded store keeps the accumulators live)
        if (v[0][0].x != 123.456f) continue;
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int col = n0 + wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5);
                if (col >= a.Nc) continue;
                if constexpr (OUT16)
                    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.y) + m * a.ldy + col) =
                        u32x2{pack_bf16(v[j][rq].x, v[j][rq].y), pack_bf16(v[j][rq].z, v[j][rq].w)};
                else
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + m * a.ldy + col) = v[j][rq];
            }
    }
}

int g_dma_ck = [] { const char* e = getenv("MI_CONV_DMA_CK"); const int v = e ? atoi(e) : 64; return (v == 32 || v == 33) ? v : 64; }();

bool dma_geom(const MiConvDesc* d, int* TH, int* TI) {
    const int W = d->OW, H = d->OH;
    if (BM % W) return false;
    const int rows = BM / W;
    if (rows <= H) { if (H % rows) return false; *TH = rows; *TI = 1; }
    else { if (rows % H) return false; *TH = H; *TI = rows / H; if ((long)d->N % *TI) return false; }
    return *TI * (*TH + 2) * (W + 2) <= MAXHP;
}

bool dma_ok(const MiConvDesc* d, int* TH, int* TI) {
    if (d->KH != 3 || d->KW != 3 || d->pad != 1 || d->stride != 1 || d->mode != 1) return false;
    if (d->IH != d->OH || d->IW != d->OW) return false;
    if (d->K % 64 || d->K1 % 64 || d->Nc % 4 || d->ldx % 8 || (d->K1 != d->K && d->ldx2 % 8)) return false;
    if (d->OW < 8 || d->OW > 32) return false;
    if (((long)d->N * d->OH * d->OW) % BM) return false;
    return dma_geom(d, TH, TI);
}

}  // namespace

extern "C" int mi_conv3x3_dma_supported(const MiConvDesc* d) {
    int th, ti;
    return (d && dma_ok(d, &th, &ti)) ? 1 : 0;
}

// x / x2: bf16 tensors (pixel strides in elements, % 8 == 0); w: bf16 [ky][kx][Nc][K]; d->transposed = 1 -> flipped taps (data
// gradient); out_bf16: y is written as bf16 (else fp32).  bias / residual fp32, d->accumulate: y += result.
extern "C" int mi_conv3x3_dma(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16, const float* bias,
                              const float* residual, void* y, int out_bf16, void* stream) {
    MI_REQUIRE(d && x && w_nk_bf16 && y, "null argument");
    DmaConvArgs a;
    MI_REQUIRE(dma_ok(d, &a.TH, &a.TI), "descriptor not supported by the LDS-DMA conv kernel (use mi_conv3x3_bf16w_io)");
    MI_REQUIRE(d->K1 == d->K || x2, "two-source split without x2");
    MI_REQUIRE((((uintptr_t)x | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)w_nk_bf16) & 15) == 0, "operands must be 16-byte aligned");
    a.x = (const uint16_t*)x; a.x2 = (const uint16_t*)(x2 ? x2 : x); a.w = (const uint16_t*)w_nk_bf16; a.bias = bias; a.res = residual; a.y = y;
    a.N = d->N; a.H = d->OH; a.W = d->OW; a.K = d->K; a.Nc = d->Nc; a.K1 = d->K1; a.ldx = d->ldx; a.ldx2 = x2 ? d->ldx2 : d->ldx;
    a.ldy = d->ldy; a.ldr = d->ldr; a.accumulate = d->accumulate; a.flip = d->transposed ? 1 : 0;
    a.tiles_per_img = a.TI > 1 ? 1 : a.H / a.TH;
    a.HP = a.TI * (a.TH + 2) * (a.W + 2);
    a.xmap = a.TI == 1 && a.tiles_per_img > 1 && a.N % 8 == 0;
    const dim3 grid((unsigned)((long)d->N * d->OH * d->OW / BM), (unsigned)((d->Nc + BN - 1) / BN));
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)conv_dma_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_dma_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_dma_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_dma_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_dma_kernel<32, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 53 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_dma_kernel<32, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 53 * 1024);
        return true;
    }();
    (void)once;
    hipStream_t st = (hipStream_t)stream;
    if (g_dma_ck == 33) {        // 32-channel chunks, one tap per stage: three workgroups per CU
        const size_t lds = (size_t)DmaShape<32, 1>::PIXOFF + MAXHP * 4;
        if (out_bf16) hipLaunchKernelGGL((conv_dma_kernel<32, true, 1>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((conv_dma_kernel<32, false, 1>), grid, dim3(256), lds, st, a);
    } else if (g_dma_ck == 32) {
        const size_t lds = (size_t)DmaShape<32>::PIXOFF + MAXHP * 4;
        if (out_bf16) hipLaunchKernelGGL((conv_dma_kernel<32, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((conv_dma_kernel<32, false>), grid, dim3(256), lds, st, a);
    } else {
        const size_t lds = (size_t)DmaShape<64>::PIXOFF + MAXHP * 4;
        if (out_bf16) hipLaunchKernelGGL((conv_dma_kernel<64, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((conv_dma_kernel<64, false>), grid, dim3(256), lds, st, a);
    }
    MI_LAUNCH_CHECK();
    return 0;
}

// 32: two workgroups per CU on 32-channel chunks; 64 (default): one workgroup per CU on 64-channel chunks
extern "C" int mi_debug_conv_dma_chunk(int ck) {
    if (ck != 32 && ck != 33 && ck != 64) return mi_set_error(-1, "mi_debug_conv_dma_chunk: 32, 33 (= 32 with one tap per stage) or 64");
    g_dma_ck = ck;
    return 0;
}
