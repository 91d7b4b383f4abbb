// 3x3 / stride 1 / pad 1 convolution (forward and data-gradient) on bf16 MFMA with an
// LDS-staged activation halo tile -- the hot kernel of the path (86 % of the UNet's FLOPs are
// Block's Conv2d(dim, dim_out, 3, padding=1), ddpm.py:116, plus its dgrad).
//
// A workgroup owns BM consecutive output pixels (whole image rows, so the tile plus a one-pixel
// halo is a small rectangle per image) x BN output channels.  Per chunk of CK input channels the
// halo rectangle is read from HBM/L2 ONCE (fp32 NHWC rows, coalesced float4), rounded to bf16
// and parked in LDS; all nine taps then read their shifted views of that tile as MFMA A
// operands, so activations are fetched 1x instead of 9x and converted 1x.  The weight tile of
// each tap streams through a double-buffered LDS slot from a bf16 [tap][n][k] copy of the
// weights (made once per optimizer step by mi_pack_weights_bf16) with 16-byte loads -- no
// conversion, no transposition in the loop.  v_mfma_f32_32x32x16_bf16, fp32 accumulate.
#include <stdlib.h>
#include "common.h"

namespace {

struct HaloArgs {
    const float* x; const float* x2; const uint16_t* w; const float* bias; const float* res; float* y;
    int N, H, W, K, Nc, K1, ldx, ldx2, ldy, ldr, accumulate, flip;
    int ksplit;          // K (channel-chunk) slices, gridDim.z; > 1 => results are combined with atomics
    int TH, TI;          // tile = TI images x TH rows x W columns (BM = TI*TH*W)
    int HP;              // halo pixels = TI*(TH+2)*(W+2)
    int tiles_per_img;   // H/TH when TI == 1
};

// waves are arranged (WAVES/2) along M x 2 along N; a wave owns (MI*32) pixels x 64 channels
template <int BM, int WAVES = (BM == 256 ? 8 : 4)> struct HaloCfg {
    static constexpr int NT = 64 * WAVES;                        // threads
    static constexpr int MI = BM / (16 * WAVES);                 // 32-row MFMA tiles per wave: 256/8 -> 2, 256/4 -> 4, 128/4 -> 2, 64/4 -> 1
    static constexpr int MAXHP = (BM == 256) ? 400 : (BM == 128 ? 288 : 160);
};

// Software pipeline, per workgroup.  "Stage" g = chunk*9 + tap.  While tap g runs on the matrix
// cores out of LDS (A halo buffer chunk&1, weight slot g&1), the registers receive stage g+2:
// the weight tile of tap g+2 and one ninth of the NEXT chunk's halo tile; at the end of tap g the
// registers of stage g+1 (fetched one full tap earlier) are written to the other weight slot /
// the other halo buffer.  Every global load therefore has two taps of MFMA time to land, there
// is one barrier per tap, and nothing is staged at chunk boundaries.  Taps are unrolled so the
// two register sets are static.
// IO bit 0: activations x / x2 are stored as bf16 (copied to LDS as they are); bit 1: y is written as bf16.
template <int BM, int CK, int KS, bool SK = false, int IO = 0, int WAVES = (BM == 256 ? 8 : 4)>
__global__ __launch_bounds__((HaloCfg<BM, WAVES>::NT)) void conv3x3_halo_kernel(const HaloArgs a) {
    constexpr bool IN16 = IO & 1, OUT16 = IO & 2;
    static_assert(!(SK && OUT16), "split-K accumulates with fp32 atomics");
    constexpr int BN = 128;
    constexpr int NTAP = KS * KS;                  // 9, or 1 for the 1x1 convolutions (plain GEMM, no halo)
    constexpr int NT = HaloCfg<BM, WAVES>::NT, MI = HaloCfg<BM, WAVES>::MI, NI = 2;
    constexpr int PITCH = CK + 8;                  // bf16 elements; 16-B aligned rows, conflict-free b128 reads
    constexpr int MAXHP = KS == 3 ? HaloCfg<BM>::MAXHP : BM;
    constexpr int Q = CK / 4;                      // float4 per halo pixel per chunk
    constexpr int A_IT = (MAXHP * Q + NT - 1) / NT;
    constexpr int A_SL = (A_IT + NTAP - 1) / NTAP; // float4 per thread per tap (NTAP slices cover a tile)
    constexpr int B_IT = BN * (CK / 8) / NT;       // 16-B loads per thread per tap
    constexpr int D = 4;                           // register ring depth: a load has D-1 taps of MFMA time to land

    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* As = lds;                                    // 2 x [MAXHP][PITCH]
    uint16_t* Bs = lds + 2 * MAXHP * PITCH;                // 2 x [BN][PITCH]
    int* pix = reinterpret_cast<int*>(Bs + 2 * BN * PITCH); // [MAXHP] source pixel index of each halo pixel, -1 = zero

    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int W2 = a.W + (KS - 1), TH2 = a.TH + (KS - 1);

    int img0, y0;
    if (a.TI > 1) { img0 = blockIdx.x * a.TI; y0 = 0; }
    else { img0 = blockIdx.x / a.tiles_per_img; y0 = (blockIdx.x % a.tiles_per_img) * a.TH; }
    for (int hp = t; hp < MAXHP; hp += NT) {
        int v = -1;
        if (KS == 1) {
            if (m0 + hp < a.N * a.H * a.W) v = m0 + hp;          // 1x1: the tile is BM consecutive pixels
        } else if (hp < a.HP) {
            int ti = hp / (TH2 * W2);
            int rem = hp - ti * (TH2 * W2);
            int hy = rem / W2, hx = rem - hy * W2;
            int iy = y0 + hy - 1, ix = hx - 1, img = img0 + ti;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && img < a.N) v = (img * a.H + iy) * a.W + ix;
        }
        pix[hp] = v;
    }
    __syncthreads();

    // ---- MFMA row -> halo pixel (before the tap shift)
    int a_row[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int r = wm * (MI * 32) + i * 32 + (l & 31);
        if (KS == 1) { a_row[i] = r * PITCH + (l >> 5) * 8; continue; }
        int tx = r % a.W, q = r / a.W;
        int ty = q % a.TH, ti = q / a.TH;
        a_row[i] = ((ti * TH2 + ty) * W2 + tx) * PITCH + (l >> 5) * 8;
    }
    const int b_row0 = (wn * 64 + (l & 31)) * PITCH + (l >> 5) * 8;
    // ---- staging assignments.  halo element e = t + NT*j: pixel e / Q, float4 e % Q
    const int a_c4 = t % Q, a_hp0 = t / Q;                 // + (NT/Q) pixels per j
    const int b_n = t / (CK / 8), b_k8 = t % (CK / 8);     // + NT/(CK/8) rows per i
    const size_t tap_stride = (size_t)a.Nc * a.K;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[D][A_SL];
    uint4 rb[D][B_IT];
    // this workgroup's slice of the channel chunks [ch0, ch0 + nchunks)
    const int allchunks = a.K / CK;
    const int per = (allchunks + a.ksplit - 1) / a.ksplit;
    const int ch0 = blockIdx.z * per;
    const int nchunks = max(0, min(allchunks, ch0 + per) - ch0);
    const int ntaps = nchunks * NTAP;

    // fetch slice `sl` (0..8) of chunk `ch`'s halo tile
    auto load_a = [&](float4 (&r)[A_SL], int ch, int sl) {
        if (ch >= nchunks) return;
        const int kc = (ch0 + ch) * CK;
        const float* src = a.x; int ld = a.ldx; int cc = kc;
        if (kc >= a.K1) { src = a.x2; ld = a.ldx2; cc = kc - a.K1; }
#pragma unroll
        for (int j = 0; j < A_SL; ++j) {
            const int hp = a_hp0 + (sl * A_SL + j) * (NT / Q);
            const int pv = hp < MAXHP ? pix[hp] : -1;
            float4 v;
            if constexpr (IN16) {     // 4 bf16 = 8 bytes, carried in .x/.y
                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(src) + (size_t)(pv >= 0 ? pv : 0) * ld + cc + a_c4 * 4);
                v = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
            } else {
                v = *reinterpret_cast<const float4*>(src + (size_t)(pv >= 0 ? pv : 0) * ld + cc + a_c4 * 4);
            }
            r[j] = pv >= 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_a = [&](int buf, const float4 (&r)[A_SL], int sl) {
#pragma unroll
        for (int j = 0; j < A_SL; ++j) {
            const int hp = a_hp0 + (sl * A_SL + j) * (NT / Q);
            if (hp < (KS == 1 ? MAXHP : a.HP))
                *reinterpret_cast<uint2*>(&As[buf * (MAXHP * PITCH) + hp * PITCH + a_c4 * 4]) =
                    IN16 ? make_uint2(__float_as_uint(r[j].x), __float_as_uint(r[j].y))
                         : make_uint2(pack_bf16(r[j].x, r[j].y), pack_bf16(r[j].z, r[j].w));
        }
    };
    auto load_b = [&](uint4 (&r)[B_IT], int g) {
        if (g >= ntaps) return;
        const int ch = g / NTAP, tap = g - ch * NTAP;
        const int wt = a.flip ? NTAP - 1 - tap : tap;
        const uint16_t* base = a.w + wt * tap_stride + (size_t)(ch0 + ch) * CK + b_k8 * 8;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = n0 + b_n + i * (NT / (CK / 8));
            uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)(n < a.Nc ? n : 0) * a.K);
            r[i] = n < a.Nc ? v : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_b = [&](int slot, const uint4 (&r)[B_IT]) {
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            *reinterpret_cast<uint4*>(&Bs[slot * (BN * PITCH) + (b_n + i * (NT / (CK / 8))) * PITCH + b_k8 * 8]) = r[i];
    };
    auto mma_tap = [&](int abuf, int slot, int tap) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const uint16_t* At = As + abuf * (MAXHP * PITCH) + (ky * W2 + kx) * PITCH;
        const uint16_t* Bt = Bs + slot * (BN * PITCH);
#pragma unroll
        for (int ks = 0; ks < CK / 16; ++ks) {
            bf16x8 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&At[a_row[i] + ks * 16]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(&Bt[b_row0 + j * 32 * PITCH + ks * 16]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    // weights as the MFMA "A" operand: D rows = output channels, D cols = pixels, so a lane
                    // ends up with 4 consecutive channels of one pixel per register quad -> 16-byte epilogue
                    // (the split-K variant keeps pixels as rows: its atomics then cover 128-byte row segments)
                    acc[i][j] = SK ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    // Stage bookkeeping (g = chunk*NTAP + tap, all ring indices static after unrolling):
    //   weights: stage g+D-1 is loaded at tap g into rb[(g+D-1) % D]; stage g+1 is stored at the end of tap g
    //            from rb[(g+1) % D] into slot (g+1) & 1;
    //   halo:    stage h = g+NTAP+D-2 (slice of the next chunks) is loaded at tap g into ra[h % D]; stage
    //            g+NTAP (slice `tap` of the next chunk) is stored at the end of tap g into the other halo buffer.
    // P = first stage of the chunk mod D (NTAP is odd, D = 4  =>  P = chunk & 3 and the halo buffer is P & 1).
    auto ring_load_b = [&](int r, int g) {
        if (r == 0) load_b(rb[0], g); else if (r == 1) load_b(rb[1], g); else if (r == 2) load_b(rb[2], g); else load_b(rb[3], g);
    };
    auto ring_store_b = [&](int r, int slot) {
        if (r == 0) store_b(slot, rb[0]); else if (r == 1) store_b(slot, rb[1]); else if (r == 2) store_b(slot, rb[2]); else store_b(slot, rb[3]);
    };
    auto ring_load_a = [&](int r, int h) {            // h = halo stage = chunk*NTAP + slice
        const int c = h / NTAP, sl = h - c * NTAP;
        if (r == 0) load_a(ra[0], c, sl); else if (r == 1) load_a(ra[1], c, sl); else if (r == 2) load_a(ra[2], c, sl); else load_a(ra[3], c, sl);
    };
    auto ring_store_a = [&](int r, int buf, int sl) {
        if (r == 0) store_a(buf, ra[0], sl); else if (r == 1) store_a(buf, ra[1], sl); else if (r == 2) store_a(buf, ra[2], sl); else store_a(buf, ra[3], sl);
    };
    auto chunk = [&](auto phase, int ch) {
        constexpr int P = decltype(phase)::value;
        const int g0 = ch * NTAP;
#pragma unroll
        for (int tp = 0; tp < NTAP; ++tp) {
            const int gm = (P + tp) % D;                       // g mod D, static
            ring_load_b((gm + D - 1) % D, g0 + tp + D - 1);
            ring_load_a((gm + NTAP + D - 2) % D, g0 + tp + NTAP + D - 2);
            mma_tap(P & 1, (P + tp) & 1, tp);
            ring_store_b((gm + 1) % D, (P + tp + 1) & 1);
            if (ch + 1 < nchunks) ring_store_a((gm + NTAP) % D, (P & 1) ^ 1, tp);
            __syncthreads();
        }
    };

    // ---- prologue: chunk 0's tile and weight stage 0 into LDS; stages 1..D-2 (weights) and the first D-2
    //      halo stages of chunk 1 into the ring (the steady state loads the (D-1)-th ahead at every tap)
    for (int sl = 0; sl < NTAP; ++sl) { load_a(ra[0], 0, sl); store_a(0, ra[0], sl); }
    load_b(rb[0], 0); store_b(0, rb[0]);
#pragma unroll
    for (int k = 1; k <= D - 2; ++k) ring_load_b(k % D, k);
#pragma unroll
    for (int k = 0; k < D - 2; ++k) ring_load_a((NTAP + k) % D, NTAP + k);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ch += 4) {
        chunk(std::integral_constant<int, 0>{}, ch);
        if (ch + 1 < nchunks) chunk(std::integral_constant<int, 1>{}, ch + 1);
        if (ch + 2 < nchunks) chunk(std::integral_constant<int, 2>{}, ch + 2);
        if (ch + 3 < nchunks) chunk(std::integral_constant<int, 3>{}, ch + 3);
    }

    // ---- epilogue: lane = pixel (l & 31), register quad rq = channels 8*rq + 4*(l >> 5) .. +3
    const int Mtot = a.N * a.H * a.W;
    const bool first = blockIdx.z == 0;                 // split-K: slice 0 carries bias and residual
    if constexpr (SK) {
        // rows = pixels, lanes 0..31 = 32 consecutive channels: one atomic instruction = two 128-byte rows
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t m = (size_t)m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (m >= (size_t)Mtot) continue;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int col = n0 + wn * 64 + j * 32 + (l & 31);
                    if (col >= a.Nc) continue;
                    float v = acc[i][j][r];
                    if (first) {
                        if (a.bias) v += a.bias[col];
                        if (a.res) v += a.res[m * a.ldr + col];
                    }
                    atomicAdd(a.y + m * a.ldy + col, v);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const size_t m = (size_t)m0 + wm * (MI * 32) + i * 32 + (l & 31);
        if (m >= (size_t)Mtot) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int col = n0 + wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5);
                if (col >= a.Nc) continue;              // Nc % 4 == 0: a quad is all in or all out
                float4 v = make_float4(acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]);
                if (a.bias && first) { float4 b = *reinterpret_cast<const float4*>(a.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                if (a.res && first) { float4 r = *reinterpret_cast<const float4*>(a.res + m * a.ldr + col); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                if constexpr (OUT16) {
                    uint16_t* yp = reinterpret_cast<uint16_t*>(a.y) + m * a.ldy + col;
                    if (a.accumulate) {
                        const uint2 o = *reinterpret_cast<const uint2*>(yp);
                        v.x += __uint_as_float(o.x << 16); v.y += __uint_as_float(o.x & 0xffff0000u);
                        v.z += __uint_as_float(o.y << 16); v.w += __uint_as_float(o.y & 0xffff0000u);
                    }
                    *reinterpret_cast<uint2*>(yp) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
                } else {
                    float* yp = a.y + m * a.ldy + col;
                    if (a.accumulate) { float4 o = *reinterpret_cast<const float4*>(yp); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *reinterpret_cast<float4*>(yp) = v;
                }
            }
    }
}

template <int BM, int CK, int KS = 3, bool SK = false, int IO = 0, int WAVES = (BM == 256 ? 8 : 4)>
void launch_halo(const HaloArgs& a, hipStream_t st) {
    constexpr int PITCH = CK + 8;
    constexpr int MAXHP = KS == 3 ? HaloCfg<BM>::MAXHP : BM;
    size_t lds = (size_t)(2 * MAXHP * PITCH + 2 * 128 * PITCH) * 2 + MAXHP * 4;
    dim3 grid((a.N * a.H * a.W + BM - 1) / BM, (a.Nc + 127) / 128, a.ksplit);
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<BM, CK, KS, SK, IO, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((conv3x3_halo_kernel<BM, CK, KS, SK, IO, WAVES>), grid, dim3(HaloCfg<BM, WAVES>::NT), lds, st, a);
}

// fp32 [tap][k][n] master weights -> bf16 Wd[tap][k][n] (same layout) and Wf[tap][n][k] (transposed per tap)
struct PackEntry { long long off; int taps, ci, co, tile0; };

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackEntry* __restrict__ ents, int nent,
                                                           const float* __restrict__ master, uint16_t* __restrict__ wd,
                                                           uint16_t* __restrict__ wf) {
    __shared__ float tile[32][33];
    int e = 0;
    while (e + 1 < nent && (int)blockIdx.x >= ents[e + 1].tile0) ++e;
    const PackEntry en = ents[e];
    int local = blockIdx.x - en.tile0;
    const int tci = (en.ci + 31) / 32, tco = (en.co + 31) / 32;
    int tap = local / (tci * tco);
    int rem = local - tap * (tci * tco);
    int bi = rem / tco, bj = rem - bi * tco;
    const float* src = master + en.off + (size_t)tap * en.ci * en.co;
    uint16_t* d1 = wd + en.off + (size_t)tap * en.ci * en.co;
    uint16_t* d2 = wf + en.off + (size_t)tap * en.ci * en.co;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        int i = bi * 32 + r, j = bj * 32 + tx;
        float v = (i < en.ci && j < en.co) ? src[(size_t)i * en.co + j] : 0.f;
        tile[r][tx] = v;
        if (i < en.ci && j < en.co) d1[(size_t)i * en.co + j] = (uint16_t)(pack_bf16(v, 0.f) & 0xffff);
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int j = bj * 32 + r, i = bi * 32 + tx;
        if (i < en.ci && j < en.co) d2[(size_t)j * en.ci + i] = (uint16_t)(pack_bf16(tile[tx][r], 0.f) & 0xffff);
    }
}

}  // namespace

// Tile geometry for BM output pixels: whole image rows (TH rows of one image) or TI whole images.
static bool halo_geom(const MiConvDesc* d, int BM, int* TH, int* TI) {
    int W = d->OW, H = d->OH;
    if (BM % W) return false;
    int rows = BM / W;
    if (rows <= H) { if (H % rows) return false; *TH = rows; *TI = 1; }
    else { if (rows % H) return false; *TH = H; *TI = rows / H; if ((long)d->N % *TI) return false; }
    int maxhp = BM == 256 ? 400 : (BM == 128 ? 288 : 160);
    return *TI * (*TH + 2) * (W + 2) <= maxhp;
}

// Can the halo kernel take this descriptor?  (3x3, stride 1, pad 1, full-width row tiles)
static bool halo_ok(const MiConvDesc* d, int* bm, int* ck) {
    const bool k3 = d->KH == 3 && d->KW == 3 && d->pad == 1, k1 = d->KH == 1 && d->KW == 1 && d->pad == 0;
    if (!(k3 || k1) || d->stride != 1 || d->mode != 1) return false;
    if (d->IH != d->OH || d->IW != d->OW) return false;
    if (d->K % 32 || d->K1 % 32 || d->Nc % 4) return false;
    if (k1) {                                  // 1x1: plain GEMM over M = N*H*W pixels, any geometry
        const long M = (long)d->N * d->OH * d->OW, nt = (d->Nc + 127) / 128;
        *bm = (M + 255) / 256 * nt >= 200 ? 256 : ((M + 127) / 128 * nt >= 400 ? 128 : 64);
        *ck = 32;                              // 64 would spill: the whole A tile rides in the 4-deep register ring
        return true;
    }
    if (d->OW < 4 || d->OW > 64) return false;
    static const int force = [] { const char* e = getenv("MI_HALO_BM"); return e ? atoi(e) : 0; }();
    const long M = (long)d->N * d->OH * d->OW, nt = (d->Nc + 127) / 128;
    int TH, TI, best = 0;
    const int cand[3] = {256, 128, 64};
    const long need[3] = {200, 400, 0};       // workgroups wanted: 256-pixel tiles run 1/CU (8 waves), smaller ones 2-3/CU
    for (int c = 0; c < 3; ++c) {
        int BM = cand[c];
        if (force && BM != force) continue;
        if (!halo_geom(d, BM, &TH, &TI)) continue;
        best = BM;                             // smallest legal tile so far (fallback)
        if ((M + BM - 1) / BM * nt >= need[c] || force) break;
    }
    if (!best) return false;
    *bm = best;
    static const int force_ck = [] { const char* e = getenv("MI_HALO_CK"); return e ? atoi(e) : 0; }();
    *ck = (best == 256 && d->K % 64 == 0 && d->K1 % 64 == 0) ? 64 : 32;   // CK=64 only where one workgroup/CU is the plan anyway
    if (force_ck == 32 || (force_ck == 64 && d->K % 64 == 0 && d->K1 % 64 == 0)) *ck = force_ck;
    return true;
}

static int halo_dispatch(const MiConvDesc* d, const float* x, const float* x2, const void* w_nk_bf16,
                         const float* bias, const float* residual, float* y, int io, void* stream);

extern "C" int mi_conv3x3_bf16w(const MiConvDesc* d, const float* x, const float* x2, const void* w_nk_bf16,
                                const float* bias, const float* residual, float* y, void* stream) {
    return halo_dispatch(d, x, x2, w_nk_bf16, bias, residual, y, 0, stream);
}

// Same kernel with bf16 activation storage: io bit 0 = x (and x2) are bf16 tensors, bit 1 = y is written as bf16
// (pixel strides then count bf16 elements).  3x3 only; the split-K variant cannot write bf16 (fp32 atomics), so
// with bit 1 set small-M layers use the 64/128-pixel tiles instead.
extern "C" int mi_conv3x3_bf16w_io(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16,
                                   const float* bias, const float* residual, void* y, int io, void* stream) {
    if (!d || d->KH != 3 || (io & ~3)) return mi_set_error(-1, "mi_conv3x3_bf16w_io: 3x3 only, io in 0..3");
    return halo_dispatch(d, (const float*)x, (const float*)x2, w_nk_bf16, bias, residual, (float*)y, io, stream);
}

static int halo_dispatch(const MiConvDesc* d, const float* x, const float* x2, const void* w_nk_bf16,
                         const float* bias, const float* residual, float* y, int io, void* stream) {
    MI_REQUIRE(d && x && w_nk_bf16 && y, "null argument");
    int BM, CK;
    MI_REQUIRE(halo_ok(d, &BM, &CK), "descriptor not supported by the halo kernel (use mi_conv_igemm)");
    MI_REQUIRE(d->K1 == d->K || x2, "two-source split without x2");
    MI_REQUIRE(d->ldx % 4 == 0 && (!x2 || d->ldx2 % 4 == 0) && (((uintptr_t)x | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)w_nk_bf16) & 15) == 0,
               "activations/weights must be 16-byte aligned with ld % 4 == 0");
    HaloArgs a;
    a.x = x; a.x2 = x2 ? x2 : x; a.w = (const uint16_t*)w_nk_bf16; a.bias = bias; a.res = residual; a.y = y;
    a.N = d->N; a.H = d->OH; a.W = d->OW; a.K = d->K; a.Nc = d->Nc; a.K1 = d->K1; a.ldx = d->ldx;
    a.ldx2 = x2 ? d->ldx2 : d->ldx; a.ldy = d->ldy; a.ldr = d->ldr; a.accumulate = d->accumulate;
    a.flip = d->transposed ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    // Small-M layers (8x8 levels): a 256-pixel x 64-channel-chunk tile with the K loop split over
    // 2-4 workgroups beats 64-pixel tiles (weights are re-read per M tile); slices are summed with
    // row-coalesced fp32 atomics.
    if (d->KH == 3 && !(io & 2) && BM < 256 && d->K % 64 == 0 && d->K1 % 64 == 0 && (d->accumulate || d->ldy == d->Nc)) {
        static const int allow = [] { const char* e = getenv("MI_HALO_SPLITK"); return e ? atoi(e) : 1; }();
        int th, ti;
        const long b256 = ((long)d->N * d->OH * d->OW + 255) / 256 * ((d->Nc + 127) / 128);
        const int chunks = d->K / 64;
        if (allow && chunks >= 8 && halo_geom(d, 256, &th, &ti) && b256 >= 32) {   // measured: loses for K < 512
            int ks = 2;
            while (b256 * ks < 200 && ks * 4 <= chunks && ks < 8) ks *= 2;             // >= 2 chunks per slice
            if (ks * 2 <= chunks) {
                a.TH = th; a.TI = ti; a.tiles_per_img = ti > 1 ? 1 : a.H / th; a.HP = ti * (th + 2) * (a.W + 2);
                a.ksplit = ks;
                if (!d->accumulate) {
                    hipError_t e = hipMemsetAsync(y, 0, (size_t)d->N * d->OH * d->OW * d->ldy * sizeof(float), st);
                    if (e != hipSuccess) return mi_set_error((int)e, "mi_conv3x3_bf16w: memset: %s", hipGetErrorString(e));
                }
                if (io & 1) launch_halo<256, 64, 3, true, 1>(a, st); else launch_halo<256, 64, 3, true>(a, st);
                MI_LAUNCH_CHECK();
                return 0;
            }
        }
    }
    // split-K when even the chosen tile leaves most CUs idle (8x8 levels, small batches): slices are
    // combined with fp32 atomics into a zeroed (or, for accumulate, the existing) output
    a.ksplit = 1;
    {
        static const int force_ks = [] { const char* e = getenv("MI_HALO_KSPLIT"); return e ? atoi(e) : 0; }();
        const long blocks = ((long)d->N * d->OH * d->OW + BM - 1) / BM * ((d->Nc + 127) / 128);
        const int chunks = d->K / CK;
        int ks = 1;
        const long want = BM == 256 ? 200 : 400;
        (void)want; (void)blocks;
        if (force_ks) ks = force_ks <= chunks ? force_ks : 1;
        if (ks > 1 && (d->accumulate || d->ldy == d->Nc)) {
            a.ksplit = ks;
            if (!d->accumulate) {
                hipError_t e = hipMemsetAsync(y, 0, (size_t)d->N * d->OH * d->OW * d->ldy * sizeof(float), st);
                if (e != hipSuccess) return mi_set_error((int)e, "mi_conv3x3_bf16w: memset: %s", hipGetErrorString(e));
            }
        }
    }
    if (d->KH == 1) {
        a.TH = 1; a.TI = 1; a.tiles_per_img = 1; a.HP = BM;
        if (BM == 256)      launch_halo<256, 32, 1>(a, st);
        else if (BM == 128) launch_halo<128, 32, 1>(a, st);
        else                launch_halo<64, 32, 1>(a, st);
        MI_LAUNCH_CHECK();
        return 0;
    }
    MI_REQUIRE(halo_geom(d, BM, &a.TH, &a.TI), "halo tile geometry");
    a.tiles_per_img = a.TI > 1 ? 1 : a.H / a.TH;
    a.HP = a.TI * (a.TH + 2) * (a.W + 2);
    // (a 4-wave variant with 128x64 wave tiles was measured 15 % slower than 8 waves of 64x64: thread-level
    //  parallelism matters more than LDS bytes per MFMA here)
#define MI_HALO_GO(IOV) \
    do { if (BM == 256) { if (CK == 64) launch_halo<256, 64, 3, false, IOV>(a, st); else launch_halo<256, 32, 3, false, IOV>(a, st); } \
         else if (BM == 128) launch_halo<128, 32, 3, false, IOV>(a, st); \
         else launch_halo<64, 32, 3, false, IOV>(a, st); } while (0)
    switch (io) { case 0: MI_HALO_GO(0); break; case 1: MI_HALO_GO(1); break; case 2: MI_HALO_GO(2); break; default: MI_HALO_GO(3); break; }
#undef MI_HALO_GO
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_conv3x3_bf16w_supported(const MiConvDesc* d) {
    int bm, ck;
    return (d && halo_ok(d, &bm, &ck)) ? 1 : 0;
}

extern "C" int mi_pack_weights_bf16(int nent, const void* entries_dev, int total_tiles, const float* master,
                                    void* wd_bf16, void* wf_bf16, void* stream) {
    MI_REQUIRE(nent > 0 && entries_dev && total_tiles > 0 && master && wd_bf16 && wf_bf16, "bad argument");
    hipLaunchKernelGGL(pack_weights_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const PackEntry*)entries_dev, nent, master, (uint16_t*)wd_bf16, (uint16_t*)wf_bf16);
    MI_LAUNCH_CHECK();
    return 0;
}
