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

#ifdef MI_HALO_TIMING
// profiling build only (make EXTRA=-DMI_HALO_TIMING): per-workgroup phase timestamps, 100 MHz wall clock
__device__ unsigned long long g_halo_ts[5 * 4096];
#define MI_TS(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 4096) g_halo_ts[(k) * 4096 + blockIdx.x] = wall_clock64(); } while (0)
extern "C" int mi_debug_halo_ts(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_halo_ts), sizeof(g_halo_ts));
}
#else
#define MI_TS(k) do {} while (0)
#endif
#ifdef MI_HALO_TAPTIME
// profiling build only (make EXTRA=-DMI_HALO_TAPTIME): per wave, shader-clock cycles spent between barriers (work) and from the
// arrival at a tap's barrier (own LDS operations drained + the other waves) to leaving it, summed over the main loop
__device__ unsigned long long g_halo_tap[4096 * 8 * 5];
extern "C" int mi_debug_halo_tap(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_halo_tap), sizeof(g_halo_tap));
}
#define MI_TAP_BARRIER() do { const unsigned long long b0_ = __builtin_amdgcn_s_memtime(); __syncthreads(); \
        const unsigned long long b1_ = __builtin_amdgcn_s_memtime(); tt_work += b0_ - tt_prev; tt_bar += b1_ - b0_; tt_prev = b1_; ++tt_n; } while (0)
#else
#define MI_TAP_BARRIER() __syncthreads()
#endif
#ifndef MI_HALO_STORE_FIRST
#define MI_HALO_STORE_FIRST 1
#endif
#ifndef MI_ABL
#define MI_ABL 0      // profiling only: 1 no tap barrier, 2 no global loads, 4 no LDS stores, 8 no MFMA in the main loop, 16 (PIPE) half the weight-fragment reads
#endif

namespace {

// compile-time loop: f(integral_constant<int, I>) for I in [0, N) -- ring slots must be static register names
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct HaloArgs {
    const float* x; const float* x2; const uint16_t* w; const float* bias; const float* res; float* y;
    int N, H, W, K, Nc, K1, ldx, ldx2, ldy, ldr, accumulate, flip;
    int ksplit;          // K (channel-chunk) slices, gridDim.z; > 1 => results are combined with atomics
    int TH, TI;          // tile = TI images x TH rows x W columns (BM = TI*TH*W)
    int HP;              // halo pixels = TI*(TH+2)*(W+2)
    int tiles_per_img;   // H/TH when TI == 1
    int xmap;            // XCD-aware tile order (3x3, several row tiles per image, N % 8 == 0)
    const float* coef;   // FUSE: [3][N][K] scale / shift / time bias of the GroupNorm + Mish applied to x while it is staged
    int skew;            // extra LDS elements per halo ROW (see HaloSkew); 0 = rows packed
    int ep_rows;         // epilogue through LDS: row-contiguous stores and residual / accumulate loads
    uint16_t* y16; int ldy16;   // optional second output: the same values rounded to bf16 (the copy the next Block's conv / the skip
                         // connection's consumer reads), written by this epilogue instead of a separate conversion pass over y
    int qmap, gx, gy;    // 1-D launch of gx pixel tiles x gy channel tiles, each XCD takes gx / (8 / qmap) pixel tiles x gy / qmap channel tiles
    const float* gn_sums; const float* gn_gamma; const float* gn_beta; const float* gn_temb;   // FUSE without a coef tensor: the
    int gn_cg, gn_ldt; float gn_eps; double gn_icnt;   // producing conv's per-slab sums [N][K / 16][2] + the affine / time-bias vectors
    float* gsum;         // optional [N][C / 16][2]: += (sum, sum of squares) of the STORED outputs per sample and 16-channel slab --
                         // the GroupNorm statistics of the next layer, taken from this conv's epilogue instead of a pass over y
};

// Bank conflicts of the 8- and 16-pixel-wide tiles.  A ds_read_b128 serves a half-wave (32 lanes) conflict-free when the 32
// 16-byte granules it touches are distinct mod 32 (measured with SQ_LDS_BANK_CONFLICT on W = 8 / 16 / 32 tiles).  A halo pixel
// is PITCH = CK + 8 elements = an odd number g of granules, so 32 consecutive pixels of one row are fine (W = 32); with W = 16 / 8
// the half-wave spans 2 / 4 image rows (W + 2 positions apart) and granules collide (W = 8: 32 % of SQ_LDS_IDX_ACTIVE were
// conflict cycles, W = 16: 27 %).  The rows' granule sets interleave exactly when consecutive rows are 16 (W = 16) resp. 8 (W = 8)
// granules apart mod 32, which a skew of SK granules per halo row gives: SK = 14 for CK = 64 ((W + 2) * 9 + 14 = 176 / 104),
// 22 for CK = 32 ((W + 2) * 5 + 22 = 112 / 72).  The skewed rows use the part of the halo buffer that these narrow tiles leave
// empty (HP < MAXHP); staging slots past the tile all write their zeros to the dump row.
template <int CK> struct HaloSkew { static constexpr int EL = (CK == 64 ? 14 : 22) * 8; };     // elements per halo row

// waves are arranged (WAVES/2) along M x 2 along N; a wave owns (MI*32) pixels x 64 channels
template <int BM, int WAVES = (BM == 256 ? 8 : 4)> struct HaloCfg {
    static constexpr int NT = 64 * WAVES;                        // threads
    static constexpr int MI = BM / (16 * WAVES);                 // 32-row MFMA tiles per wave: 256/8 -> 2, 256/4 -> 4, 128/4 -> 2, 64/4 -> 1
    static constexpr int MAXHP = (BM == 256) ? 400 : (BM == 128 ? 288 : 160);
    static constexpr int MAXHP3 = (BM == 256) ? 355 : MAXHP;      // with a third weight slot (PIPE) 400 halo pixels no longer fit 160 KB
};

// Software pipeline, per workgroup.  "Stage" g = chunk*9 + tap (3x3) or the 32-channel chunk index (1x1).
// While tap g runs on the matrix cores out of LDS (halo buffer chunk&1, weight slot g&1), global loads for
// later stages are in flight in a ring of R register slots: at tap g the weight tile of stage g+R is
// requested and, R-1 taps ahead of its LDS store, one ninth of the NEXT chunk's halo tile; at the end of
// tap g the registers of stage g+1 are written to the other weight slot / the other halo buffer.  A load
// has R-1 full taps of MFMA time to land and there is one barrier per tap.
// The loop body is straight-line code: taps are unrolled with TP % R == 0 so every ring index is static,
// the last group is a separate instantiation instead of a guarded one, padding is applied as an AND mask
// at the LDS store and out-of-range rows are clamped to valid addresses.  With no branch around a global
// load the compiler can count outstanding loads exactly (s_waitcnt vmcnt(N)); any conditional load makes
// it drain the whole ring (vmcnt(0)) at every tap, which is what bounded the first version of this kernel.
// IO bit 0: activations x / x2 are stored as bf16 (copied to LDS as they are); bit 1: y is written as bf16.
// FUSE (north_star's named kernel): x is the RAW output of the previous conv; GroupNorm-apply + Mish (+ time bias),
//     h = mish(x * scale[n][c] + shift[n][c]) + tb[n][c]      (reference ddpm.py:112-120,139-140),
// is applied in registers between the global load and the LDS store of the halo tile, so the normalised tensor never exists in HBM.
// The coefficients come from mi_gn_stats_coef; tiles must lie inside one image (TI == 1); zero padding stays zero (it is the
// padding of h, not of x: the AND mask is applied after the transform).
// PIPE: three weight slots instead of two -- the weights of stage g+2 are written during tap g, so everything tap g+1 reads was
// published by the barrier BEFORE tap g and the first MFMA operands of tap g+1 are fetched from LDS while tap g's last MFMAs
// run.  After the barrier the matrix pipe starts at once instead of waiting for an LDS round trip that all 8 waves of the
// workgroup (one workgroup per CU: nobody else to fill the gap) begin at the same moment.
// N64 (round 4): layers with at most 64 output channels (cfg 3's 64 x 64 level).  The wave grid is (WAVES / 2) x 2 with 64 channels per
// wave column: at Nc <= 64 the second column multiplied a clamped copy of the last weight row and threw the result away -- half of the
// workgroup's MFMAs.  Here all WAVES waves lie along M (one 32-pixel block each at BM = 256 instead of two) over the one 64-channel
// column: same tile, same halo buffers, every MFMA used.
template <int BM, int CK, int KS, bool SK = false, int IO = 0, int WAVES = (BM == 256 ? 8 : 4), bool FUSE = false, bool PIPE = false, bool N64 = false>
__global__ __launch_bounds__((HaloCfg<BM, WAVES>::NT)) void conv3x3_halo_kernel(const HaloArgs a) {
    static_assert(!N64 || (HaloCfg<BM, WAVES>::MI % 2 == 0 && KS == 3 && !SK), "N64: 3x3 tiles whose waves own two pixel blocks");
    MI_PRIO_UP();
    constexpr bool IN16 = IO & 1, OUT16 = IO & 2;
    static_assert(!PIPE || (KS == 3 && !SK && CK == 64), "PIPE: 3x3, 64-channel chunks");
    constexpr int WS = PIPE ? 3 : 2;               // weight slots
    static_assert(!FUSE || (KS == 3 && !SK), "fusion: 3x3 forward tiles only");
    static_assert(!(SK && OUT16), "split-K accumulates with fp32 atomics");
    static_assert(KS == 3 || CK == 32, "1x1: 32-channel stages");
    constexpr int BN = 128;
    constexpr int NTAP = KS * KS;                  // 9, or 1 for the 1x1 convolutions (plain GEMM, no halo)
    constexpr int TP = KS == 3 ? 9 : 4;            // taps per unrolled group: a chunk's 9 taps, or four 32-channel stages
    constexpr int R = KS == 3 ? 3 : 4;             // ring slots (TP % R == 0)
    constexpr int NSL = KS == 3 ? 9 : 1;           // slices a halo tile is fetched in
    constexpr int NT = HaloCfg<BM, WAVES>::NT, MI = N64 ? HaloCfg<BM, WAVES>::MI / 2 : HaloCfg<BM, WAVES>::MI, NI = 2;
    constexpr int PITCH = CK + 8;                  // bf16 elements; 16-B aligned rows, conflict-free b128 reads
    constexpr int MAXHP = KS == 3 ? (PIPE ? HaloCfg<BM>::MAXHP3 : HaloCfg<BM>::MAXHP) : BM;
    constexpr int ASZ = (MAXHP + 1) * PITCH;       // one halo buffer (+1 dump row for the staging slots past the tile)
    constexpr int Q = CK / 4;                      // float4 per halo pixel per chunk
    constexpr int A_IT = (MAXHP * Q + NT - 1) / NT;
    constexpr int A_SL = (A_IT + NSL - 1) / NSL;   // float4 per thread per slice
    constexpr int B_IT = BN * (CK / 8) / NT;       // 16-B loads per thread per tap
    static_assert(B_IT >= 1, "weight tile smaller than the workgroup");

    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* As = lds;                                    // 2 x [MAXHP + 1][PITCH]
    uint16_t* Bs = lds + 2 * ASZ;                          // WS x [BN][PITCH]
    int* pix = reinterpret_cast<int*>(Bs + WS * BN * PITCH); // [MAXHP] source pixel index of each halo pixel, -1 = zero

    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int wm = N64 ? wv : wv >> 1, wn = N64 ? 0 : wv & 1;
    // M tile of this workgroup.  Row tiles of one image share their halo rows and consecutive workgroup ids go to
    // different XCDs (separate L2s), so with xmap ids xcd + 8*slot, slot = tile_in_image + tiles_per_img*m, belong to
    // image xcd + 8*m: an image's tiles meet in one L2 and the halo rows come from HBM once.
    int bx = blockIdx.x;
    if (KS == 3 && a.xmap) {
        const int xcd = bx & 7, slot = bx >> 3;
        bx = (xcd + 8 * (slot / a.tiles_per_img)) * a.tiles_per_img + slot % a.tiles_per_img;
    }
    int by = blockIdx.y;
    if (a.qmap) {
        // Tiles of whole images (8x8 levels): ids go round-robin over the 8 XCDs, each with its own L2.  With the channel tiles in
        // grid.y every XCD met ALL of them, i.e. pulled the whole weight tensor (8 x 4.7 MB of the 54.7 MB a 512 -> 512 launch
        // moved).  Here XCD = (pixel group, channel group) of a P x Q = 8 split: traffic Q * in + P * w + out, 44 MB at (4, 2);
        // within an XCD the channel tiles of one pixel tile are adjacent in time (its input tile is still in L2).
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3, Q = a.qmap, P = 8 / Q;
        const int ppx = a.gx / P, cpq = a.gy / Q;
        bx = (xcd / Q) * ppx + slot / cpq; by = (xcd % Q) * cpq + slot % cpq;
    }
    const int m0 = bx * BM, n0 = by * BN;
    const int W2 = a.W + (KS - 1), TH2 = a.TH + (KS - 1);
    const int Mtot = a.N * a.H * a.W;

    // this workgroup's slice of the channel chunks [ch0, ch0 + nchunks)
    const int allchunks = a.K / CK;
    const int per = (allchunks + a.ksplit - 1) / a.ksplit;
    const int ch0 = blockIdx.z * per;
    const int nchunks = min(allchunks, ch0 + per) - ch0;
    if (nchunks <= 0) return;
    const int ngroups = KS == 3 ? nchunks : (nchunks + TP - 1) / TP;   // 1x1: stages past K multiply zeroed weights

    MI_TS(0);
    const int a_c4 = t % Q, a_hp0 = t / Q;                 // staging element e = t + NT*j: pixel e / Q, float4 e % Q
    int pofs[NSL][A_SL];                                   // source pixel of each of this thread's staging slots, < 0 = zero
    if constexpr (KS == 3) {
        int img0, y0;
        if (a.TI > 1) { img0 = bx * a.TI; y0 = 0; }
        else { img0 = bx / a.tiles_per_img; y0 = (bx % a.tiles_per_img) * a.TH; }
        for (int hp = t; hp < MAXHP; hp += NT) {
            int v = -1;
            if (hp < a.HP) {
                int ti = hp / (TH2 * W2);
                int rem = hp - ti * (TH2 * W2);
                int hy = rem / W2, hx = rem - hy * W2;
                int iy = y0 + hy - 1, ix = hx - 1, img = img0 + ti;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && img < a.N) v = (img * a.H + iy) * a.W + ix;
            }
            pix[hp] = v;
        }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int j = 0; j < A_SL; ++j) {
                const int hp = a_hp0 + (sl * A_SL + j) * (NT / Q);
                pofs[sl][j] = (sl * A_SL + j < A_IT && hp < MAXHP) ? pix[hp] : -1;
            }
    } else {
#pragma unroll
        for (int j = 0; j < A_SL; ++j) {                   // 1x1: the tile is BM consecutive pixels
            const int hp = a_hp0 + j * (NT / Q);
            pofs[0][j] = (hp < MAXHP && m0 + hp < Mtot) ? m0 + hp : -1;
        }
    }

    // LDS element offset of each of this thread's staging slots (slots past the tile go to the dump row)
    int aofs[NSL][A_SL];
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int j = 0; j < A_SL; ++j) {
            const int hp = a_hp0 + (sl * A_SL + j) * (NT / Q);
            aofs[sl][j] = (hp < (KS == 3 ? a.HP : MAXHP) ? hp * PITCH + (KS == 3 ? (hp / W2) * a.skew : 0) : MAXHP * PITCH) + a_c4 * 4;
        }

    // ---- MFMA row -> halo pixel (before the tap shift)
    int a_row[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int r = wm * (MI * 32) + i * 32 + (l & 31);
        if (KS == 1) { a_row[i] = r * PITCH + (l >> 5) * 8; continue; }
        int tx = r % a.W, q = r / a.W;
        int ty = q % a.TH, ti = q / a.TH;
        a_row[i] = ((ti * TH2 + ty) * W2 + tx) * PITCH + (ti * TH2 + ty) * a.skew + (l >> 5) * 8;
    }
    const int b_row0 = (wn * 64 + (l & 31)) * PITCH + (l >> 5) * 8;
    const int b_n = t / (CK / 8), b_k8 = t % (CK / 8);     // + NT/(CK/8) rows per i
    const size_t tap_stride = (size_t)a.Nc * a.K;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[R][A_SL];
    u32x4 rb[R][B_IT];
    // FUSE: coefficients of the chunk whose slices are being stored (this thread's channel quad), of image bx / tiles_per_img
    f32x4 cf_s = {1.f, 1.f, 1.f, 1.f}, cf_b = {0.f, 0.f, 0.f, 0.f}, cf_t = {0.f, 0.f, 0.f, 0.f};
    auto load_coef = [&](int ch) {
        if constexpr (FUSE) {
            const int n = bx / a.tiles_per_img, c0 = (ch0 + ch) * CK + a_c4 * 4;
            if (a.gn_sums) {
                // statistics straight from the sums the producing conv's epilogue left (no coefficient tensor, no extra launch):
                // the group's 16-channel slabs are combined in double, var = E[x^2] - mean^2
                const int g = c0 / a.gn_cg, nslab = a.gn_cg >> 4;
                const size_t p = ((size_t)n * (a.K >> 4) + (size_t)g * nslab) * 2;
                double sm = 0.0, sq = 0.0;
                for (int k = 0; k < nslab; ++k) { sm += gsum_get(a.gn_sums, p + 2 * k); sq += gsum_get(a.gn_sums, p + 2 * k + 1); }
                const double mean = sm * a.gn_icnt;                                      // no fp64 division in the loop
                const float var = (float)fmax(sq * a.gn_icnt - mean * mean, 0.0), rstd = __builtin_amdgcn_rsqf(var + a.gn_eps), mf = (float)mean;
                cf_s = *reinterpret_cast<const f32x4*>(a.gn_gamma + c0) * rstd;
                cf_b = *reinterpret_cast<const f32x4*>(a.gn_beta + c0) - cf_s * mf;
                cf_t = a.gn_temb ? *reinterpret_cast<const f32x4*>(a.gn_temb + (size_t)n * a.gn_ldt + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                const size_t NK = (size_t)a.N * a.K;
                const float* c = a.coef + (size_t)n * a.K + c0;
                cf_s = *reinterpret_cast<const f32x4*>(c); cf_b = *reinterpret_cast<const f32x4*>(c + NK); cf_t = *reinterpret_cast<const f32x4*>(c + 2 * NK);
            }
        }
    };

    // fetch slice `sl` of chunk `ch`'s halo tile (unconditional: padding slots read pixel 0 and are masked at the store)
    auto load_a = [&](f32x4 (&r)[A_SL], int ch, int sl) {
        const int kc = (ch0 + ch) * CK;
        const bool second = kc >= a.K1;
        const float* src = second ? a.x2 : a.x;
        const int ld = second ? a.ldx2 : a.ldx;
        const int cc = second ? kc - a.K1 : kc;
#pragma unroll
        for (int j = 0; j < A_SL; ++j) {
            if (KS == 3 && sl * A_SL + j >= A_IT) continue;      // slots past the largest tile (NSL * A_SL rounds A_IT up): no thread has one -- no
                                                                 // dummy load, no dump-row store, no registers (BM = 256: 5 of 18)
            const size_t e = (size_t)max(pofs[sl][j], 0) * ld + cc + a_c4 * 4;
            if constexpr (IN16) {     // 4 bf16 = 8 bytes, carried in .x/.y
                const u32x2 u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(src) + e);
                r[j] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
            } else {
                r[j] = *reinterpret_cast<const f32x4*>(src + e);
            }
        }
    };
    auto store_a = [&](int buf, const f32x4 (&r)[A_SL], int sl) {
#pragma unroll
        for (int j = 0; j < A_SL; ++j) {
            if (KS == 3 && sl * A_SL + j >= A_IT) continue;
            const uint32_t keep = ~(uint32_t)(pofs[sl][j] >> 31);
            u32x2 v;
            if constexpr (FUSE) {
                f32x4 x4;
                if constexpr (IN16) {
                    const uint32_t ux = __float_as_uint(r[j].x), uy = __float_as_uint(r[j].y);
                    x4 = f32x4{__uint_as_float(ux << 16), __uint_as_float(ux & 0xffff0000u), __uint_as_float(uy << 16), __uint_as_float(uy & 0xffff0000u)};
                } else {
                    x4 = r[j];
                }
                const f32x4 z = x4 * cf_s + cf_b;
                v = u32x2{pack_bf16(mish_fast_f(z.x) + cf_t.x, mish_fast_f(z.y) + cf_t.y), pack_bf16(mish_fast_f(z.z) + cf_t.z, mish_fast_f(z.w) + cf_t.w)};
            } else {
                v = IN16 ? u32x2{__float_as_uint(r[j].x), __float_as_uint(r[j].y)}
                         : u32x2{pack_bf16(r[j].x, r[j].y), pack_bf16(r[j].z, r[j].w)};
            }
            *reinterpret_cast<u32x2*>(&As[buf * ASZ + aofs[sl][j]]) = u32x2{v.x & keep, v.y & keep};
        }
    };
    // weight tile of chunk `ch`, tap `tap`; rows past Nc are clamped (their output columns are never written)
    auto load_b = [&](u32x4 (&r)[B_IT], int ch, int tap) {
        const int wt = a.flip ? NTAP - 1 - tap : tap;
        const uint16_t* base = a.w + wt * tap_stride + (size_t)(ch0 + ch) * CK + b_k8 * 8;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = min(n0 + b_n + i * (NT / (CK / 8)), a.Nc - 1);
            r[i] = *reinterpret_cast<const u32x4*>(base + (size_t)n * a.K);
        }
    };
    auto store_b = [&](int slot, const u32x4 (&r)[B_IT], uint32_t keep = ~0u) {
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            *reinterpret_cast<u32x4*>(&Bs[slot * (BN * PITCH) + (b_n + i * (NT / (CK / 8))) * PITCH + b_k8 * 8]) =
                KS == 1 ? r[i] & keep : r[i];
    };
    bf16x8 pa[2][MI], pb[2][NI];                           // PIPE: operand sets that live across taps
    // tap `tp` of a chunk with the operands of its first k-step already in set 0 (tp > 0: fetched during tap tp - 1; tp == 0: the
    // chunk's halo buffer was completed by the slice stored during the previous tap, so it is read here, after the barrier)
    auto mma_tap_pipe = [&](auto tpc, auto lastc, int abuf) {
        constexpr int tp = decltype(tpc)::value;
        auto opnd = [&](int set, int tap, int ks) {
            const int ky = tap / KS, kx = tap - ky * KS;
            const uint16_t* At = As + abuf * ASZ + (ky * W2 + kx) * PITCH + ky * a.skew;
            const uint16_t* Bt = Bs + (tap % WS) * (BN * PITCH);
#pragma unroll
            for (int i = 0; i < MI; ++i) pa[set][i] = *reinterpret_cast<const bf16x8*>(&At[a_row[i] + ks * 16]);
#pragma unroll
            for (int j = 0; j < ((MI_ABL & 16) ? 1 : NI); ++j) pb[set][j] = *reinterpret_cast<const bf16x8*>(&Bt[b_row0 + j * 32 * PITCH + ks * 16]);
            if constexpr ((MI_ABL & 16) != 0) pb[set][1] = pb[set][0];        // profiling only: one of the two weight fragments is not read
        };
        if constexpr (tp == 0) opnd(0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < CK / 16; ++ks) {
            if (ks + 1 < CK / 16) opnd((ks + 1) & 1, tp, ks + 1);
            else if constexpr (tp + 1 < TP) opnd(0, tp + 1, 0);       // next tap: slot (tp + 1) % 3 and the halo buffer are already published
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pb[ks & 1][j], pa[ks & 1][i], acc[i][j], 0, 0, 0);
        }
    };
    auto mma_tap = [&](int abuf, int slot, int tap) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const uint16_t* At = As + abuf * ASZ + (ky * W2 + kx) * PITCH + ky * a.skew;
        const uint16_t* Bt = Bs + slot * (BN * PITCH);
        bf16x8 af[2][MI], bf[2][NI];
        auto frags = [&](int set, int ks) {
#pragma unroll
            for (int i = 0; i < MI; ++i) af[set][i] = *reinterpret_cast<const bf16x8*>(&At[a_row[i] + ks * 16]);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[set][j] = *reinterpret_cast<const bf16x8*>(&Bt[b_row0 + j * 32 * PITCH + ks * 16]);
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < CK / 16; ++ks) {
            if (ks + 1 < CK / 16) frags((ks + 1) & 1, ks + 1);     // operands of the next k-step ride behind this one's MFMAs
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    // weights as the MFMA "A" operand: D rows = output channels, D cols = pixels, so a lane
                    // ends up with 4 consecutive channels of one pixel per register quad -> 16-byte epilogue
                    // (the split-K variant keeps pixels as rows: its atomics then cover 128-byte row segments)
                    acc[i][j] = SK ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks & 1][j], af[ks & 1][i], acc[i][j], 0, 0, 0);
        }
    };
    // One group = TP taps.  3x3: group = chunk `gi`; weights stage g+R is loaded at tap g into rb[g % R] and stage
    // g+1 stored at the end of tap g; halo slice (tp+R-1) % 9 of the next chunk(s) is loaded into ra[(tp+R-1) % R]
    // and slice tp of chunk gi+1 stored at the end of tap tp.  1x1: stage = 32-channel chunk gi*4 + tp, activations
    // and weights both ride R taps ahead.  LAST: nothing beyond this group is loaded or stored.
#ifdef MI_HALO_TAPTIME
    unsigned long long tt_prev = 0, tt_work = 0, tt_bar = 0, tt_n = 0, tt_st = 0;
#endif
    auto group = [&](auto lastc, int gi) {
        constexpr bool LAST = decltype(lastc)::value;
        if constexpr (FUSE && !LAST) load_coef(gi + 1);      // the slices stored during this group belong to chunk gi + 1
        static_for<0, TP>([&](auto tpc) {
            constexpr int tp = decltype(tpc)::value;
            if constexpr (PIPE) {
                // weights: stage g+4 is requested, stage g+2 goes to slot (tp + 2) % 3 (9 % 3 == 0: slot = tap % 3 in every chunk).
                // The LDS stores come FIRST: nothing reads their targets before the next barrier, their sources were requested two
                // taps ago, and issued here they complete under this tap's MFMAs -- at the end of the tap every wave would sit
                // through vmcnt -> ds_write -> lgkmcnt(0) in front of the barrier with the matrix pipe idle (ablation: 23 % of the
                // 8x8-level kernel, 14 % at level 0).
                if constexpr (MI_HALO_STORE_FIRST && !(MI_ABL & 4)) {
                    if constexpr (tp + 2 < TP || !LAST) store_b((tp + 2) % WS, rb[(tp + 2) % R]);
                    if constexpr (!LAST) store_a((gi & 1) ^ 1, ra[tp % R], tp);
#ifdef MI_HALO_TAPTIME
                    __builtin_amdgcn_sched_barrier(0);
                    tt_st += __builtin_amdgcn_s_memtime() - tt_prev;      // from leaving the barrier to the last LDS store issued
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
                if constexpr (tp + 4 < TP) load_b(rb[(tp + 1) % R], gi, tp + 4);
                else if constexpr (!LAST) load_b(rb[(tp + 1) % R], gi + 1, tp + 4 - TP);
                if constexpr (!LAST) {
                    constexpr int sl = (tp + R - 1) % TP;
                    load_a(ra[(tp + R - 1) % R], tp + R - 1 < TP ? gi + 1 : min(gi + 2, nchunks - 1), sl);
                }
                mma_tap_pipe(tpc, lastc, gi & 1);
                if constexpr (!MI_HALO_STORE_FIRST && !(MI_ABL & 4)) {
                if constexpr (tp + 2 < TP || !LAST) store_b((tp + 2) % WS, rb[(tp + 2) % R]);
                if constexpr (!LAST) store_a((gi & 1) ^ 1, ra[tp % R], tp);
                }
            } else if constexpr (KS == 3) {
                if constexpr (!(MI_ABL & 2)) {
                if constexpr (tp + R < TP) load_b(rb[tp % R], gi, tp + R);
                else if constexpr (!LAST) load_b(rb[tp % R], gi + 1, tp + R - TP);
                if constexpr (!LAST) {
                    constexpr int sl = (tp + R - 1) % TP;
                    load_a(ra[(tp + R - 1) % R], tp + R - 1 < TP ? gi + 1 : min(gi + 2, nchunks - 1), sl);
                }
                }
                if constexpr (!(MI_ABL & 8)) mma_tap(gi & 1, (gi + tp) & 1, tp);
                if constexpr (!(MI_ABL & 4)) {
                if constexpr (tp + 1 < TP || !LAST) store_b(((gi + tp) & 1) ^ 1, rb[(tp + 1) % R]);
                if constexpr (!LAST) store_a((gi & 1) ^ 1, ra[tp % R], tp);
                }
            } else {
                if constexpr (!LAST) {
                    const int sn = min(gi * TP + tp + R, nchunks - 1);
                    load_b(rb[tp % R], sn, 0);
                    load_a(ra[tp % R], sn, 0);
                }
                mma_tap(tp & 1, tp & 1, 0);
                if constexpr (tp + 1 < TP || !LAST) {
                    store_b((tp + 1) & 1, rb[(tp + 1) % R], gi * TP + tp + 1 < nchunks ? ~0u : 0u);
                    store_a((tp + 1) & 1, ra[(tp + 1) % R], 0);
                }
            }
            if constexpr (!(MI_ABL & 1)) MI_TAP_BARRIER();
        });
    };

    MI_TS(1);
    // ---- prologue: every load of stage 0 (the whole first halo tile) is issued at once, then the ring is
    //      primed in the order the steady state would have issued it, then stage 0 goes to LDS
    {
        f32x4 p0[NSL][A_SL];
        u32x4 b0[B_IT];
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) load_a(p0[sl], 0, sl);
        load_b(b0, 0, 0);
        if constexpr (PIPE) {
            const int c1 = min(1, nchunks - 1);
            u32x4 b1[B_IT];
            load_b(b1, 0, 1);
            load_b(rb[2], 0, 2); load_a(ra[0], c1, 0);
            load_b(rb[0], 0, 3); load_a(ra[1], c1, 1);
            store_b(1, b1);
        } else if constexpr (KS == 3) {
            const int c1 = min(1, nchunks - 1);
            load_b(rb[1], 0, 1); load_a(ra[0], c1, 0);
            load_b(rb[2], 0, 2); load_a(ra[1], c1, 1);
        } else {
            load_b(rb[1], min(1, nchunks - 1), 0); load_a(ra[1], min(1, nchunks - 1), 0);
            load_b(rb[2], min(2, nchunks - 1), 0); load_a(ra[2], min(2, nchunks - 1), 0);
            load_b(rb[3], min(3, nchunks - 1), 0); load_a(ra[3], min(3, nchunks - 1), 0);
        }
        load_coef(0);
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) store_a(0, p0[sl], sl);
        store_b(0, b0);
    }
    __syncthreads();
    MI_TS(2);
#ifdef MI_HALO_TAPTIME
    tt_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long tt_w0 = wall_clock64();
#endif
    for (int gi = 0; gi + 1 < ngroups; ++gi) group(std::false_type{}, gi);
    group(std::true_type{}, ngroups - 1);
#ifdef MI_HALO_TAPTIME
    if (l == 0 && blockIdx.x < 4096 && blockIdx.y == 0 && blockIdx.z == 0) {
        unsigned long long* g = g_halo_tap + ((size_t)blockIdx.x * 8 + wv) * 5;
        g[0] = tt_work; g[1] = tt_bar; g[2] = tt_n; g[3] = wall_clock64() - tt_w0; g[4] = tt_st;       // 100 MHz ticks over the same span
    }
#endif

    MI_TS(3);
    // ---- epilogue: lane = pixel (l & 31), register quad rq = channels 8*rq + 4*(l >> 5) .. +3
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
    // bias for this lane's 8 channel quads, fetched with one wait; the residual rows likewise per 32-pixel block
    f32x4 bq[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) bq[j][rq] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias && first) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int col = n0 + wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5);
                bq[j][rq] = *reinterpret_cast<const f32x4*>(a.bias + min(col, a.Nc - 4));
            }
    }
    if (a.ep_rows) {
        // Row-contiguous epilogue.  In the accumulator layout a lane owns 4 consecutive channels of ONE pixel, so a store (and a
        // residual / accumulate load) instruction touches 64 different rows with 8-16 bytes each: 13.1 us instead of 7.1 us for a
        // [131072][128] bf16 tensor (tools/proto/store_probe.hip), and nothing overlaps it because all workgroups of a round finish
        // together.  The wave's tile goes through a wave-private LDS area ([pixel][64 co] fp32, 272-byte pitch: an odd number of
        // 16-byte granules, so the 32 rows a half-wave writes are conflict-free) and comes back as lane = (pixel, 4-channel chunk):
        // an instruction then covers 4 whole 256-byte (fp32) / 128-byte (bf16) row segments.
        constexpr int EPP = 68;                                  // floats per staged pixel row
        __syncthreads();                                         // every wave is done with the halo / weight tiles
        float* ep = reinterpret_cast<float*>(lds) + wv * (MI * 32 * EPP);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq)
                    *reinterpret_cast<f32x4*>(ep + (i * 32 + (l & 31)) * EPP + j * 32 + 8 * rq + 4 * (l >> 5)) =
                        f32x4{acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]} + bq[j][rq];
        const int c4 = (l & 15) * 4, col = n0 + wn * 64 + c4, colc = min(col, a.Nc - 4);
        const bool use_res = a.res && first;
#pragma unroll
        for (int k0 = 0; k0 < MI * 8; k0 += 4) {                 // 4 pixels per pass, 4 passes per batch of loads
            f32x4 v[4], rv[4], ov[4];
            size_t mm[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = (k0 + q) * 4 + (l >> 4);
                mm[q] = (size_t)m0 + wm * (MI * 32) + p;
                const size_t mc = min(mm[q], (size_t)Mtot - 1);
                v[q] = *reinterpret_cast<const f32x4*>(ep + p * EPP + c4);
                rv[q] = use_res ? *reinterpret_cast<const f32x4*>(a.res + mc * a.ldr + colc) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.accumulate) {
                    if constexpr (OUT16) {
                        const u32x2 o = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.y) + mc * a.ldy + colc);
                        ov[q] = f32x4{__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u),
                                      __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
                    } else {
                        ov[q] = *reinterpret_cast<const f32x4*>(a.y + mc * a.ldy + colc);
                    }
                } else {
                    ov[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 o = (v[q] + rv[q]) + ov[q];
                if (mm[q] >= (size_t)Mtot || col >= a.Nc) continue;
                if constexpr (OUT16)
                    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.y) + mm[q] * a.ldy + col) = u32x2{pack_bf16(o.x, o.y), pack_bf16(o.z, o.w)};
                else
                    *reinterpret_cast<f32x4*>(a.y + mm[q] * a.ldy + col) = o;
            }
        }
        MI_TS(4);
        return;
    }
    float wps[NI * 2], wpq[NI * 2];             // a.gsum: this wave's sums over its MI pixel blocks
#pragma unroll
    for (int k = 0; k < NI * 2; ++k) wps[k] = wpq[k] = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const size_t m = (size_t)m0 + wm * (MI * 32) + i * 32 + (l & 31);
        const size_t mc = min(m, (size_t)Mtot - 1);
        f32x4 v[NI][4];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                v[j][rq] = f32x4{acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]} + bq[j][rq];
        if (a.res && first) {
            f32x4 rv[NI][4];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int col = n0 + wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5);
                    rv[j][rq] = *reinterpret_cast<const f32x4*>(a.res + mc * a.ldr + min(col, a.Nc - 4));
                }
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) v[j][rq] += rv[j][rq];
        }
        if (a.accumulate) {
            f32x4 ov[NI][4];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int col = min(n0 + wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5), a.Nc - 4);
                    if constexpr (OUT16) {
                        const u32x2 o = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(a.y) + mc * a.ldy + col);
                        ov[j][rq] = f32x4{__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u),
                                          __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
                    } else {
                        ov[j][rq] = *reinterpret_cast<const f32x4*>(a.y + mc * a.ldy + col);
                    }
                }
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) v[j][rq] += ov[j][rq];
        }
        if (a.gsum) {
            // per-lane partial sums of this 32-pixel block (one image: H*W % 32 == 0), four 16-channel slabs per wave column
            // block; the values are the ones the next kernel will read (rounded to bf16 when y is stored as bf16)
            float ps[NI * 2], pq[NI * 2];
#pragma unroll
            for (int k = 0; k < NI * 2; ++k) ps[k] = pq[k] = 0.f;
            const bool rowok = m < (size_t)Mtot;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int col = n0 + wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5);
                    f32x4 q = v[j][rq];
                    if constexpr (OUT16) {
                        const uint32_t p0 = pack_bf16(q.x, q.y), p1 = pack_bf16(q.z, q.w);
                        q = f32x4{__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xffff0000u), __uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
                    }
                    const float live = (rowok && col < a.Nc) ? 1.f : 0.f;
                    ps[j * 2 + (rq >> 1)] += live * ((q.x + q.y) + (q.z + q.w));
                    pq[j * 2 + (rq >> 1)] += live * ((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
                }
#pragma unroll
            for (int k = 0; k < NI * 2; ++k) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) { ps[k] += __shfl_xor(ps[k], off, 64); pq[k] += __shfl_xor(pq[k], off, 64); }
            }
            if (a.TI == 1) {                       // the whole tile is one image: combined per workgroup after the loop
#pragma unroll
                for (int k = 0; k < NI * 2; ++k) { wps[k] += ps[k]; wpq[k] += pq[k]; }
            } else if (l == 0) {
                const size_t mblk = (size_t)m0 + wm * (MI * 32) + i * 32;
                if (mblk < (size_t)Mtot) {
                    const int n = (int)(mblk / ((size_t)a.H * a.W));
#pragma unroll
                    for (int k = 0; k < NI * 2; ++k) {
                        const int col = n0 + wn * 64 + (k >> 1) * 32 + (k & 1) * 16;
                        if (col < a.Nc) {
                            const size_t g = ((size_t)n * (a.Nc / 16) + col / 16) * 2;
                            gsum_add(a.gsum, g, ps[k]); gsum_add(a.gsum, g + 1, pq[k]);
                        }
                    }
                }
            }
        }
        if (m >= (size_t)Mtot) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int col = n0 + wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5);
                if (col >= a.Nc) continue;              // Nc % 4 == 0: a quad is all in or all out
                if constexpr (OUT16)
                    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(a.y) + m * a.ldy + col) =
                        u32x2{pack_bf16(v[j][rq].x, v[j][rq].y), pack_bf16(v[j][rq].z, v[j][rq].w)};
                else
                    *reinterpret_cast<f32x4*>(a.y + m * a.ldy + col) = v[j][rq];
                if (!OUT16 && a.y16)
                    *reinterpret_cast<u32x2*>(a.y16 + m * a.ldy16 + col) = u32x2{pack_bf16(v[j][rq].x, v[j][rq].y), pack_bf16(v[j][rq].z, v[j][rq].w)};
            }
    }
    if (a.gsum && a.TI == 1) {
        // one atomic pair per (sample, 16-channel slab) and workgroup instead of one per wave and pixel block: device-scope fp32
        // atomics run at ~8 per ns chip-wide (131 k of them cost the level-0 conv 16 us), the combine through LDS costs nothing
        float* red = reinterpret_cast<float*>(lds);                         // [wave][4 slabs][2]
        __syncthreads();                                                    // every wave is done with the tiles
        if (l == 0) {
#pragma unroll
            for (int k = 0; k < NI * 2; ++k) { red[(wv * 4 + k) * 2] = wps[k]; red[(wv * 4 + k) * 2 + 1] = wpq[k]; }
        }
        __syncthreads();
        if (t < 16) {                                                       // thread = (wn, slab k, sum / square)
            const int wn2 = t >> 3, k = (t >> 1) & 3, q = t & 1;
            float tot = 0.f;
            if constexpr (N64) { if (wn2 == 0) for (int w2 = 0; w2 < WAVES; ++w2) tot += red[(w2 * 4 + k) * 2 + q]; }
            else
            for (int w2 = 0; w2 < WAVES / 2; ++w2) tot += red[((w2 * 2 + wn2) * 4 + k) * 2 + q];
            const int col = n0 + wn2 * 64 + (k >> 1) * 32 + (k & 1) * 16;
            if (col < a.Nc && (size_t)m0 < (size_t)Mtot)
                gsum_add(a.gsum, ((size_t)(m0 / (a.H * a.W)) * (a.Nc / 16) + col / 16) * 2 + q, tot);
        }
    }
    MI_TS(4);
}

template <int BM, int CK, int KS = 3, bool SK = false, int IO = 0, int WAVES = (BM == 256 ? 8 : 4), bool FUSE = false, bool PIPE = false, bool N64 = false>
void launch_halo(const HaloArgs& a_in, hipStream_t st) {
    constexpr int PITCH = CK + 8;
    constexpr int MAXHP = KS == 3 ? (PIPE ? HaloCfg<BM>::MAXHP3 : HaloCfg<BM>::MAXHP) : BM;
    size_t lds = (size_t)(2 * (MAXHP + 1) * PITCH + (PIPE ? 3 : 2) * 128 * PITCH) * 2 + MAXHP * 4;
    HaloArgs a = a_in;
    // off by default: measured neutral (level 0 56.7 -> 55.8 us, step 6.39 vs 6.41 ms) -- what the ablation charges to the stores is
    // their burst at the end of a round of workgroups, not the rows per instruction
    static const int ep_env = (int)mi_knob("MI_HALO_EPI", 0);
    a.ep_rows = (ep_env && !SK && !a.gsum && !a.y16) ? 1 : 0;
    if (a.ep_rows) {
        const size_t need = (size_t)WAVES * HaloCfg<BM, WAVES>::MI * 32 * 68 * sizeof(float);
        if (need > lds) lds = need;
    }
    static const int skew_env = (int)mi_knob("MI_HALO_SKEW", 1);
    a.skew = 0;
    if (skew_env && KS == 3 && (a.W == 8 || a.W == 16)) {     // the skewed tile must end before the dump row
        const int rows = a.TI * (a.TH + 2);
        if (a.HP * PITCH + rows * HaloSkew<CK>::EL <= MAXHP * PITCH) a.skew = HaloSkew<CK>::EL;
    }
    dim3 grid((a.N * a.H * a.W + BM - 1) / BM, (a.Nc + 127) / 128, a.ksplit);
    a.qmap = 0; a.gx = (int)grid.x; a.gy = (int)grid.y;
    static const int q_env = (int)mi_knob("MI_HALO_PQ", 1);
    if (q_env && KS == 3 && !SK && a.ksplit == 1 && !a.xmap && a.gy > 1 && a.TH == a.H) {   // whole-image tiles
        const int Q = (a.gy % 2 == 0) ? 2 : 1, P = 8 / Q;
        if (Q > 1 && a.gx % P == 0) { a.qmap = Q; grid = dim3(grid.x * grid.y, 1, 1); }
    }
    // 1x1 convs with several channel tiles (to_qkv: 3): (P, Q) = (8, 1) -- an XCD walks its pixel tiles with the channel tiles of one
    // pixel tile adjacent, so the input tile is read from HBM once instead of once per channel tile (level 0: 203 -> 137 MB)
    if (q_env && KS == 1 && !SK && a.ksplit == 1 && a.gy > 1 && a.gx % 8 == 0) { a.qmap = 1; grid = dim3(grid.x * grid.y, 1, 1); }
    static MiPerDevice once;
    once.run([] {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<BM, CK, KS, SK, IO, WAVES, FUSE, PIPE, N64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL((conv3x3_halo_kernel<BM, CK, KS, SK, IO, WAVES, FUSE, PIPE, N64>), grid, dim3(HaloCfg<BM, WAVES>::NT), lds, st, a);
}

// fp32 [tap][k][n] master weights -> bf16 Wd[tap][k][n] (same layout) and Wf[tap][n][k] (transposed per tap); entries with frag != 0
// (3x3 layers, ci % 64 == 0 and co % 64 == 0) also get both copies in MFMA-fragment order for conv_pw.hip:
//   Wfq[tap][co / 32][ci / 16][lane][8] = W[tap][ci = 16 kq + 8 (lane >> 5) + e][co = 32 nb + (lane & 31)]   (forward operand)
//   Wdq[tap][ci / 32][co / 16][lane][8] = W[tap][ci = 32 nb + (lane & 31)][co = 16 kq + 8 (lane >> 5) + e]   (data-gradient operand)
struct PackEntry { long long off; int taps, ci, co, tile0, frag, pad_; };

constexpr int PACK_TILE = 64;       // mi_pack_weights_tile(): entries count their tiles as taps * ceil(ci / 64) * ceil(co / 64)

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackEntry* __restrict__ ents, int nent,
                                                           const float* __restrict__ master, uint16_t* __restrict__ wd,
                                                           uint16_t* __restrict__ wf, uint16_t* __restrict__ wdq,
                                                           uint16_t* __restrict__ wfq) {
    __shared__ float tile[PACK_TILE][PACK_TILE + 1];
    int e = 0;                      // last entry whose first tile is <= blockIdx.x: binary search (a linear walk is ~60 dependent
    for (int hi = nent; hi - e > 1;) {      // scalar loads per workgroup and was most of this kernel's time)
        const int mid = (e + hi) >> 1;
        if ((int)blockIdx.x >= ents[mid].tile0) e = mid; else hi = mid;
    }
    const PackEntry en = ents[e];
    int local = blockIdx.x - en.tile0;
    const int tci = (en.ci + PACK_TILE - 1) / PACK_TILE, tco = (en.co + PACK_TILE - 1) / PACK_TILE;
    int tap = local / (tci * tco);
    int rem = local - tap * (tci * tco);
    int bi = rem / tco, bj = rem - bi * tco;
    const float* src = master + en.off + (size_t)tap * en.ci * en.co;
    uint16_t* d1 = wd + en.off + (size_t)tap * en.ci * en.co;
    uint16_t* d2 = wf + en.off + (size_t)tap * en.ci * en.co;
    if (((en.ci | en.co) & 3) == 0 && (en.off & 3) == 0) {
        // 64 x 64 tile, 16 threads per row: float4 in, 4 bf16 (8 bytes) out, both copies in 128-byte row segments
        const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = r0 + 16 * p, i = bi * PACK_TILE + r, j = bj * PACK_TILE + c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < en.ci && j < en.co) {
                v = *reinterpret_cast<const f32x4*>(src + (size_t)i * en.co + j);
                *reinterpret_cast<u32x2*>(d1 + (size_t)i * en.co + j) = u32x2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
            }
            tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = r0 + 16 * p, j = bj * PACK_TILE + r, i = bi * PACK_TILE + c4;      // output row = co, 4 consecutive ci
            if (i < en.ci && j < en.co)
                *reinterpret_cast<u32x2*>(d2 + (size_t)j * en.ci + i) =
                    u32x2{pack_bf16(tile[c4][r], tile[c4 + 1][r]), pack_bf16(tile[c4 + 2][r], tile[c4 + 3][r])};
        }
        if (en.frag && wdq) {
            // the tile = 2 x 4 fragments of each order; thread = (fragments f and f + 4, lane l): 8 values out of LDS and one 16-byte
            // store per fragment, 1 KB per wave and store
            const int l = threadIdx.x & 63;
            const size_t tapo = (size_t)en.off + (size_t)tap * en.ci * en.co;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int f = (threadIdx.x >> 6) + 4 * h;
                const int r32 = (f >> 2) * 32 + (l & 31), c8 = (f & 3) * 16 + 8 * (l >> 5);
                {   // data-gradient operand: rows = ci (n), columns = co (k)
                    const size_t fo = ((size_t)(bi * 2 + (f >> 2)) * (en.co / 16) + bj * 4 + (f & 3)) * 512 + l * 8;
                    *reinterpret_cast<u32x4*>(wdq + tapo + fo) =
                        u32x4{pack_bf16(tile[r32][c8], tile[r32][c8 + 1]), pack_bf16(tile[r32][c8 + 2], tile[r32][c8 + 3]),
                              pack_bf16(tile[r32][c8 + 4], tile[r32][c8 + 5]), pack_bf16(tile[r32][c8 + 6], tile[r32][c8 + 7])};
                }
                {   // forward operand: rows = co (n), columns = ci (k)
                    const size_t fo = ((size_t)(bj * 2 + (f >> 2)) * (en.ci / 16) + bi * 4 + (f & 3)) * 512 + l * 8;
                    *reinterpret_cast<u32x4*>(wfq + tapo + fo) =
                        u32x4{pack_bf16(tile[c8][r32], tile[c8 + 1][r32]), pack_bf16(tile[c8 + 2][r32], tile[c8 + 3][r32]),
                              pack_bf16(tile[c8 + 4][r32], tile[c8 + 5][r32]), pack_bf16(tile[c8 + 6][r32], tile[c8 + 7][r32])};
                }
            }
        }
        return;
    }
    // channel counts that are not multiples of 4 (the 3-channel ends): element by element
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < PACK_TILE; r += 4) {
        int i = bi * PACK_TILE + r, j = bj * PACK_TILE + tx;
        float v = (i < en.ci && j < en.co) ? src[(size_t)i * en.co + j] : 0.f;
        tile[r][tx] = v;
        if (i < en.ci && j < en.co) d1[(size_t)i * en.co + j] = (uint16_t)(pack_bf16(v, 0.f) & 0xffff);
    }
    __syncthreads();
    for (int r = ty; r < PACK_TILE; r += 4) {
        int j = bj * PACK_TILE + r, i = bi * PACK_TILE + tx;
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
    if (k1) {                                  // 1x1: plain GEMM over M = N*H*W pixels, any geometry; four 32-channel stages per group (K is padded with zero weights)
        const long M = (long)d->N * d->OH * d->OW, nt = (d->Nc + 127) / 128;
        *bm = (M + 255) / 256 * nt >= 200 ? 256 : ((M + 127) / 128 * nt >= 400 ? 128 : 64);
        static const int force1 = (int)mi_knob("MI_HALO_K1_BM", 0);     // experiment: 64 / 128 / 256
        if (force1 == 64 || force1 == 128 || (force1 == 256 && M >= 256)) *bm = force1;
        *ck = 32;                              // 64 would spill: the whole A tile rides in the 4-deep register ring
        return true;
    }
    if (d->OW < 4 || d->OW > 64) return false;
    static const int force = (int)mi_knob("MI_HALO_BM", 0);
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
    static const int force_ck = (int)mi_knob("MI_HALO_CK", 0);
    *ck = (best == 256 && d->K % 64 == 0 && d->K1 % 64 == 0) ? 64 : 32;   // CK=64 only where one workgroup/CU is the plan anyway
    if (force_ck == 32 || (force_ck == 64 && d->K % 64 == 0 && d->K1 % 64 == 0)) *ck = force_ck;
    return true;
}

static int halo_dispatch(const MiConvDesc* d, const float* x, const float* x2, const void* w_nk_bf16,
                         const float* bias, const float* residual, float* y, int io, void* stream, float* gsum = nullptr,
                         void* y16 = nullptr, int ldy16 = 0);

// k-slices of the split-K plan for this layer (0 = none).  Small-M layers (8x8 levels): a 256-pixel x
// 64-channel-chunk tile with the K loop split over 2-8 workgroups beats 64-pixel tiles (weights are re-read
// per M tile); slices are summed with row-coalesced fp32 atomics, so the output must be fp32.
static int halo_splitk(const MiConvDesc* d, int BM, int* th, int* ti) {
    if (d->KH != 3 || BM >= 256 || d->K % 64 || d->K1 % 64 || !(d->accumulate || d->ldy == d->Nc)) return 0;
    // off by default: since the staging ring runs with exact waits the 64-pixel tiles (2-3 workgroups per CU) are
    // faster than split-K for every cfg-2 / cfg-3 layer (measured 14.17k vs 13.95k images/s); MI_HALO_SPLITK=1 enables it
    static const int allow = (int)mi_knob("MI_HALO_SPLITK", 0);
    const long b256 = ((long)d->N * d->OH * d->OW + 255) / 256 * ((d->Nc + 127) / 128);
    const long b128 = ((long)d->N * d->OH * d->OW + 127) / 128 * ((d->Nc + 127) / 128);
    const int chunks = d->K / 64;
    if (allow == 2 && b128 >= 200) return 0;                                             // 2: only where the 128-pixel / 8-wave tiles do not apply
    if (!(allow && chunks >= 8 && halo_geom(d, 256, th, ti) && b256 >= 32)) return 0;   // measured: loses for K < 512
    int ks = 2;
    while (b256 * ks < 200 && ks * 4 <= chunks && ks < 8) ks *= 2;                      // >= 2 chunks per slice
    return ks * 2 <= chunks ? ks : 0;
}

// 8x8-level layers (too few pixels for 256-pixel tiles): 128-pixel tiles with 8 waves and 64-channel chunks -- the weight
// tile is shared by 8 waves and read once per 128 pixels, 8 MFMAs per wave between barriers -- when that still gives
// (almost) every CU a workgroup.  MI_HALO_W8 = minimum number of workgroups (0 = never).
static bool halo_w8(const MiConvDesc* d, int BM, int* th, int* ti) {
    static const int w8 = (int)mi_knob("MI_HALO_W8", 200);
    const long b128 = ((long)d->N * d->OH * d->OW + 127) / 128 * ((d->Nc + 127) / 128);
    return w8 && d->KH == 3 && BM == 64 && d->K % 64 == 0 && d->K1 % 64 == 0 && b128 >= w8 && halo_geom(d, 128, th, ti);
}
// 64-pixel tiles that cannot fill the chip twice anyway (<= 256 workgroups): 64-channel chunks, i.e. twice the MFMAs per
// barrier; the larger LDS footprint (one workgroup per CU) costs nothing then
// three weight slots + operands of the next tap fetched before the barrier (PIPE); MI_HALO_PIPE=0 restores the two-slot kernels
static bool halo_pipe() {
    static const int on = (int)mi_knob("MI_HALO_PIPE", 0);
    return on != 0;
}
static bool halo_wide64(const MiConvDesc* d, int BM) {
    static const int ck64 = (int)mi_knob("MI_HALO_CK64", 1);
    return ck64 && d->KH == 3 && BM == 64 && d->K % 64 == 0 && d->K1 % 64 == 0 &&
           ((long)d->N * d->OH * d->OW + 63) / 64 * ((d->Nc + 127) / 128) <= 256;
}

// 1 when mi_conv3x3_bf16w would take the split-K plan for this descriptor (fp32 output only): callers that can
// choose the output type use it to keep such layers in fp32.
extern "C" int mi_conv3x3_bf16w_uses_splitk(const MiConvDesc* d) {
    int bm, ck, th, ti;
    if (!d || !halo_ok(d, &bm, &ck)) return 0;
    return halo_splitk(d, bm, &th, &ti) ? 1 : 0;
}

extern "C" int mi_conv3x3_bf16w(const MiConvDesc* d, const float* x, const float* x2, const void* w_nk_bf16,
                                const float* bias, const float* residual, float* y, void* stream) {
    return halo_dispatch(d, x, x2, w_nk_bf16, bias, residual, y, 0, stream);
}

// Same kernel with bf16 activation storage: io bit 0 = x (and x2) are bf16 tensors, bit 1 = y is written as bf16
// (pixel strides then count bf16 elements).  3x3 only; the split-K variant cannot write bf16 (fp32 atomics), so
// with bit 1 set small-M layers use the 64/128-pixel tiles instead.
extern "C" int mi_conv3x3_bf16w_io(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16,
                                   const float* bias, const float* residual, void* y, int io, void* stream) {
    if (!d || (d->KH != 3 && d->KH != 1) || (io & ~3)) return mi_set_error(-1, "mi_conv3x3_bf16w_io: 3x3 or 1x1, io in 0..3");
    return halo_dispatch(d, (const float*)x, (const float*)x2, w_nk_bf16, bias, residual, (float*)y, io, stream);
}

// ... and the next layer's GroupNorm statistics from this conv's epilogue: gsum [N][Nc / 16][2] (zeroed by the caller) += (sum,
// sum of squares) of the values as stored, per sample and 16-channel slab; mi_gn_coef_from_sums turns them into the coefficients
// mi_conv3x3_gn_mish reads.  3x3, Nc % 16 == 0, H*W % 32 == 0, no split-K plan.
extern "C" int mi_conv3x3_bf16w_io_gnsums(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16, const float* bias,
                                          const float* residual, void* y, int io, float* gsum, void* stream) {
    if (!d || d->KH != 3 || (io & ~3) || !gsum || d->Nc % 16 || (d->OH * d->OW) % 32)
        return mi_set_error(-1, "mi_conv3x3_bf16w_io_gnsums: 3x3, io in 0..3, Nc %% 16 == 0, H*W %% 32 == 0");
    return halo_dispatch(d, (const float*)x, (const float*)x2, w_nk_bf16, bias, residual, (float*)y, io, stream, gsum);
}

// ... with a second, bf16 copy of an fp32 output written by the same epilogue (io bit 1 must be 0; 3x3 or 1x1; no split-K plan):
// the residual-stream tensors that the next Block's conv and its weight gradient read as bf16.
extern "C" int mi_conv3x3_bf16w_io_dual(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16, const float* bias,
                                        const float* residual, float* y, void* y_bf16, int ldy16, int io, void* stream) {
    if (!d || (d->KH != 3 && d->KH != 1) || (io & ~1) || !y_bf16 || ldy16 % 4 || ((uintptr_t)y_bf16 & 7))
        return mi_set_error(-1, "mi_conv3x3_bf16w_io_dual: 3x3 or 1x1, fp32 y (io 0 or 1), 8-byte aligned bf16 copy with ldy16 %% 4 == 0");
    return halo_dispatch(d, (const float*)x, (const float*)x2, w_nk_bf16, bias, residual, y, io, stream, nullptr, y_bf16, ldy16);
}

static int halo_dispatch(const MiConvDesc* d, const float* x, const float* x2, const void* w_nk_bf16,
                         const float* bias, const float* residual, float* y, int io, void* stream, float* gsum, void* y16, int ldy16) {
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
    a.gsum = gsum; a.coef = nullptr; a.gn_sums = nullptr; a.y16 = (uint16_t*)y16; a.ldy16 = ldy16;
    hipStream_t st = (hipStream_t)stream;
    // Small-M layers (8x8 levels): a 256-pixel x 64-channel-chunk tile with the K loop split over
    // 2-4 workgroups beats 64-pixel tiles (weights are re-read per M tile); slices are summed with
    // row-coalesced fp32 atomics.
    if (!(io & 2) && !gsum && !y16) {
        int th, ti;
        const int ks = halo_splitk(d, BM, &th, &ti);
        if (ks) {
            {
                a.TH = th; a.TI = ti; a.tiles_per_img = ti > 1 ? 1 : a.H / th; a.HP = ti * (th + 2) * (a.W + 2);
                a.ksplit = ks; a.xmap = 0;
                if (!d->accumulate) {
                    hipError_t e = mi_zero_async(y, (size_t)d->N * d->OH * d->OW * d->ldy * sizeof(float), st);
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
        static const int force_ks = (int)mi_knob("MI_HALO_KSPLIT", 0);
        const long blocks = ((long)d->N * d->OH * d->OW + BM - 1) / BM * ((d->Nc + 127) / 128);
        const int chunks = d->K / CK;
        int ks = 1;
        const long want = BM == 256 ? 200 : 400;
        (void)want; (void)blocks;
        if (force_ks) ks = force_ks <= chunks ? force_ks : 1;
        if (ks > 1 && (d->accumulate || d->ldy == d->Nc)) {
            a.ksplit = ks;
            if (!d->accumulate) {
                hipError_t e = mi_zero_async(y, (size_t)d->N * d->OH * d->OW * d->ldy * sizeof(float), st);
                if (e != hipSuccess) return mi_set_error((int)e, "mi_conv3x3_bf16w: memset: %s", hipGetErrorString(e));
            }
        }
    }
    if (d->KH == 1) {
        a.TH = 1; a.TI = 1; a.tiles_per_img = 1; a.HP = BM; a.xmap = 0;
#define MI_HALO1_GO(IOV) \
    do { if (BM == 256) launch_halo<256, 32, 1, false, IOV>(a, st); \
         else if (BM == 128) launch_halo<128, 32, 1, false, IOV>(a, st); \
         else launch_halo<64, 32, 1, false, IOV>(a, st); } while (0)
        switch (io) { case 0: MI_HALO1_GO(0); break; case 1: MI_HALO1_GO(1); break; case 2: MI_HALO1_GO(2); break; default: MI_HALO1_GO(3); break; }
#undef MI_HALO1_GO
        MI_LAUNCH_CHECK();
        return 0;
    }
    MI_REQUIRE(halo_geom(d, BM, &a.TH, &a.TI), "halo tile geometry");
    a.tiles_per_img = a.TI > 1 ? 1 : a.H / a.TH;
    static const int xmap_env = (int)mi_knob("MI_HALO_XCD", 1);
    a.xmap = xmap_env && a.TI == 1 && a.tiles_per_img > 1 && a.N % 8 == 0;
    a.HP = a.TI * (a.TH + 2) * (a.W + 2);
    // (a 4-wave variant with 128x64 wave tiles was measured 15 % slower than 8 waves of 64x64: thread-level
    //  parallelism matters more than LDS bytes per MFMA here)
    {
        int th8, ti8;
        if (halo_w8(d, BM, &th8, &ti8)) {
            a.TH = th8; a.TI = ti8; a.tiles_per_img = ti8 > 1 ? 1 : a.H / th8; a.HP = ti8 * (th8 + 2) * (a.W + 2);
            a.xmap = a.xmap && a.TI == 1 && a.tiles_per_img > 1;
            if (halo_pipe())
                switch (io) {
                    case 0: launch_halo<128, 64, 3, false, 0, 8, false, true>(a, st); break;
                    case 1: launch_halo<128, 64, 3, false, 1, 8, false, true>(a, st); break;
                    case 2: launch_halo<128, 64, 3, false, 2, 8, false, true>(a, st); break;
                    default: launch_halo<128, 64, 3, false, 3, 8, false, true>(a, st); break;
                }
            else
            switch (io) {
                case 0: launch_halo<128, 64, 3, false, 0, 8>(a, st); break;
                case 1: launch_halo<128, 64, 3, false, 1, 8>(a, st); break;
                case 2: launch_halo<128, 64, 3, false, 2, 8>(a, st); break;
                default: launch_halo<128, 64, 3, false, 3, 8>(a, st); break;
            }
            MI_LAUNCH_CHECK();
            return 0;
        }
    }
    const bool wide64 = halo_wide64(d, BM);
    static const int n64_on = (int)mi_knob("MI_HALO_N64", 1);
    const bool n64 = n64_on && BM == 256 && CK == 64 && d->Nc <= 64;
#define MI_HALO_GO(IOV) \
    do { if (BM == 256) { if (n64 && halo_pipe() && a.HP <= HaloCfg<256>::MAXHP3) launch_halo<256, 64, 3, false, IOV, 8, false, true, true>(a, st); \
                          else if (n64) launch_halo<256, 64, 3, false, IOV, 8, false, false, true>(a, st); \
                          else if (CK == 64 && halo_pipe() && a.HP <= HaloCfg<256>::MAXHP3) launch_halo<256, 64, 3, false, IOV, 8, false, true>(a, st); \
                          else if (CK == 64) launch_halo<256, 64, 3, false, IOV>(a, st); else launch_halo<256, 32, 3, false, IOV>(a, st); } \
         else if (BM == 128) launch_halo<128, 32, 3, false, IOV>(a, st); \
         else if (wide64) launch_halo<64, 64, 3, false, IOV>(a, st); \
         else launch_halo<64, 32, 3, false, IOV>(a, st); } while (0)
    switch (io) { case 0: MI_HALO_GO(0); break; case 1: MI_HALO_GO(1); break; case 2: MI_HALO_GO(2); break; default: MI_HALO_GO(3); break; }
#undef MI_HALO_GO
    MI_LAUNCH_CHECK();
    return 0;
}

// ---- north_star's named kernel: GroupNorm-apply + Mish (+ time bias) fused into the 3x3 conv's input staging -------------------
// Tile = the largest of 256 / 128 / 64 output pixels that lies inside ONE image and still fills the chip.
static bool fused_plan(const MiConvDesc* d, int* bm, int* ck, int* th) {
    if (d->KH != 3 || d->KW != 3 || d->pad != 1 || d->stride != 1 || d->mode != 1 || d->transposed) return false;
    if (d->IH != d->OH || d->IW != d->OW || d->K1 != d->K || d->K % 32 || d->Nc % 4 || d->OW < 4 || d->OW > 64) return false;
    const long M = (long)d->N * d->OH * d->OW, nt = (d->Nc + 127) / 128;
    const int cand[3] = {256, 128, 64};
    const long need[3] = {200, 400, 0};
    int best = 0, bth = 0;
    for (int c = 0; c < 3; ++c) {
        int TH, TI;
        if (!halo_geom(d, cand[c], &TH, &TI) || TI != 1) continue;
        best = cand[c]; bth = TH;
        if ((M + cand[c] - 1) / cand[c] * nt >= need[c]) break;
    }
    if (!best) return false;
    *bm = best; *th = bth;
    *ck = (d->K % 64 == 0 && (best == 256 || (best == 64 && (M + 63) / 64 * nt <= 256))) ? 64 : 32;
    return true;
}

extern "C" int mi_conv3x3_gn_mish_supported(const MiConvDesc* d) {
    int bm, ck, th;
    return (d && fused_plan(d, &bm, &ck, &th)) ? 1 : 0;
}

// profiling attribution: conv3x3_halo_kernel<bm, ck, 3, false, io, waves, true>
extern "C" int mi_conv3x3_gn_mish_tile(const MiConvDesc* d, int* bm, int* ck) {
    int th;
    MI_REQUIRE(d && bm && ck && fused_plan(d, bm, ck, &th), "descriptor not supported by the fused kernel");
    return 0;
}

// y = conv3x3( mish(x * scale + shift) + tb ) + bias, with x the raw output of the previous conv (fp32 or bf16, io bit 0) and
// coef = [3][N][K] from mi_gn_stats_coef.  io bit 1: y is written as bf16.  Only io 0 (fp32 -> fp32) and 3 (bf16 -> bf16) exist.
static int fused_go(const MiConvDesc* d, const void* x, const float* coef, const float* sums, const float* gamma, const float* beta,
                    const float* temb, int ldt, int G, float eps, const void* w_nk_bf16, const float* bias, void* y, int io, void* stream);

extern "C" int mi_conv3x3_gn_mish(const MiConvDesc* d, const void* x, const float* coef, const void* w_nk_bf16, const float* bias,
                                  void* y, int io, void* stream) {
    MI_REQUIRE(coef, "null coef");
    return fused_go(d, x, coef, nullptr, nullptr, nullptr, nullptr, 0, 1, 0.f, w_nk_bf16, bias, y, io, stream);
}
// The same kernel fed by the producing conv's epilogue sums (mi_conv3x3_bf16w_io_gnsums) instead of a coefficient tensor: GroupNorm
// (G groups, K / G % 16 == 0) statistics, affine and time bias (temb [N][ldt], optional) are resolved per channel chunk inside the
// kernel -- Block -> Block costs two launches: conv1 (+ sums) and this one.
extern "C" int mi_conv3x3_gn_mish_sums(const MiConvDesc* d, const void* x, const float* sums, const float* gamma, const float* beta,
                                       const float* temb, int ldt, int G, float eps, const void* w_nk_bf16, const float* bias, void* y,
                                       int io, void* stream) {
    MI_REQUIRE(d && sums && gamma && beta && G > 0 && d->K % G == 0 && (d->K / G) % 16 == 0 && (!temb || ldt % 4 == 0) &&
               (((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)(temb ? temb : gamma)) & 15) == 0,
               "bad argument (K / G must be a multiple of 16, 16-byte aligned vectors)");
    return fused_go(d, x, nullptr, sums, gamma, beta, temb, ldt, G, eps, w_nk_bf16, bias, y, io, stream);
}

static int fused_go(const MiConvDesc* d, const void* x, const float* coef, const float* sums, const float* gamma, const float* beta,
                    const float* temb, int ldt, int G, float eps, const void* w_nk_bf16, const float* bias, void* y, int io, void* stream) {
    MI_REQUIRE(d && x && (coef || sums) && w_nk_bf16 && y && (io == 0 || io == 3), "bad argument (io 0 or 3)");
    int BM, CK, TH;
    MI_REQUIRE(fused_plan(d, &BM, &CK, &TH), "descriptor not supported by the fused GroupNorm+Mish+Conv3x3 kernel");
    MI_REQUIRE(d->ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)w_nk_bf16 | (uintptr_t)(coef ? coef : sums)) & 7) == 0 && !d->accumulate, "alignment / accumulate");
    HaloArgs a;
    a.x = (const float*)x; a.x2 = a.x; a.w = (const uint16_t*)w_nk_bf16; a.bias = bias; a.res = nullptr; a.y = (float*)y;
    a.N = d->N; a.H = d->OH; a.W = d->OW; a.K = d->K; a.Nc = d->Nc; a.K1 = d->K; a.ldx = d->ldx; a.ldx2 = d->ldx; a.ldy = d->ldy; a.ldr = 0;
    a.accumulate = 0; a.flip = 0; a.ksplit = 1; a.coef = coef; a.gsum = nullptr; a.y16 = nullptr; a.ldy16 = 0;
    a.gn_sums = sums; a.gn_gamma = gamma; a.gn_beta = beta; a.gn_temb = temb; a.gn_ldt = ldt; a.gn_eps = eps;
    a.gn_cg = sums ? d->K / G : 16; a.gn_icnt = 1.0 / ((double)d->OH * (double)d->OW * (double)a.gn_cg);
    a.TH = TH; a.TI = 1; a.tiles_per_img = a.H / TH; a.HP = (TH + 2) * (a.W + 2);
    a.xmap = a.tiles_per_img > 1 && a.N % 8 == 0;
    hipStream_t st = (hipStream_t)stream;
#define MI_FUSE_GO(IOV) \
    do { if (BM == 256 && CK == 64) launch_halo<256, 64, 3, false, IOV, 8, true>(a, st); \
         else if (BM == 256) launch_halo<256, 32, 3, false, IOV, 8, true>(a, st); \
         else if (BM == 128) launch_halo<128, 32, 3, false, IOV, 4, true>(a, st); \
         else if (CK == 64) launch_halo<64, 64, 3, false, IOV, 4, true>(a, st); \
         else launch_halo<64, 32, 3, false, IOV, 4, true>(a, st); } while (0)
    if (io == 0) MI_FUSE_GO(0); else MI_FUSE_GO(3);
#undef MI_FUSE_GO
    MI_LAUNCH_CHECK();
    return 0;
}

// Which instantiation conv3x3_halo_kernel<BM, CK, KS, SK, IO, WAVES> mi_conv3x3_bf16w(_io) launches for a descriptor
// (profiling attribution only): *sk = 1 for the split-K plan.
extern "C" int mi_conv3x3_bf16w_tile(const MiConvDesc* d, int io, int* bm, int* ck, int* sk) {
    MI_REQUIRE(d && bm && ck && sk, "null argument");
    MI_REQUIRE(halo_ok(d, bm, ck), "descriptor not supported by the halo kernel");
    int th, ti;
    *sk = (!(io & 2) && halo_splitk(d, *bm, &th, &ti)) ? 1 : 0;
    if (*sk) { *bm = 256; *ck = 64; }
    else if (halo_w8(d, *bm, &th, &ti)) { *bm = 128; *ck = 64; }     // 8 waves (the only 128-pixel / 64-channel form)
    else if (halo_wide64(d, *bm)) *ck = 64;
    return 0;
}

extern "C" int mi_conv3x3_bf16w_supported(const MiConvDesc* d) {
    int bm, ck;
    return (d && halo_ok(d, &bm, &ck)) ? 1 : 0;
}

// tile edge the entries' tile0 fields are counted in: taps * ceil(ci / T) * ceil(co / T) tiles per entry
extern "C" int mi_pack_weights_tile(void) { return PACK_TILE; }

extern "C" int mi_pack_weights_bf16(int nent, const void* entries_dev, int total_tiles, const float* master,
                                    void* wd_bf16, void* wf_bf16, void* wdq_bf16, void* wfq_bf16, void* stream) {
    MI_REQUIRE(nent > 0 && entries_dev && total_tiles > 0 && master && wd_bf16 && wf_bf16, "bad argument");
    MI_REQUIRE((wdq_bf16 == nullptr) == (wfq_bf16 == nullptr), "fragment-order copies: both or neither");
    MI_REQUIRE((((uintptr_t)wdq_bf16 | (uintptr_t)wfq_bf16) & 15) == 0, "fragment-order copies must be 16-byte aligned");
    hipLaunchKernelGGL(pack_weights_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const PackEntry*)entries_dev, nent, master, (uint16_t*)wd_bf16, (uint16_t*)wf_bf16,
                       (uint16_t*)wdq_bf16, (uint16_t*)wfq_bf16);
    MI_LAUNCH_CHECK();
    return 0;
}
