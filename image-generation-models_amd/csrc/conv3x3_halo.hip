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
#include "common.h"

namespace {

struct HaloArgs {
    const float* x; const float* x2; const uint16_t* w; const float* bias; const float* res; float* y;
    int N, H, W, K, Nc, K1, ldx, ldx2, ldy, ldr, accumulate, flip;
    int TH, TI;          // tile = TI images x TH rows x W columns (BM = TI*TH*W)
    int HP;              // halo pixels = TI*(TH+2)*(W+2)
    int tiles_per_img;   // H/TH when TI == 1
};

template <int BM, int CK>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const HaloArgs a) {
    constexpr int BN = 128;
    constexpr int PITCH = CK + 8;                  // bf16 elements; 16-B aligned rows, conflict-free b128 reads
    constexpr int MAXHP = (BM == 128) ? 288 : 160;
    constexpr int A_IT = (MAXHP * (CK / 4) + 255) / 256;   // float4 loads per thread per chunk
    constexpr int B_IT = BN * (CK / 8) / 256;              // 16-B loads per thread per tap
    constexpr int MI = BM / 64, NI = 2;                    // waves 2(M) x 2(N); wave tile (BM/2) x 64

    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* As = lds;                                    // [MAXHP][PITCH]
    uint16_t* Bs = lds + MAXHP * PITCH;                    // 2 x [BN][PITCH]

    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int W2 = a.W + 2, TH2 = a.TH + 2;

    // tile origin
    int img0, y0;
    if (a.TI > 1) { img0 = blockIdx.x * a.TI; y0 = 0; }
    else { img0 = blockIdx.x / a.tiles_per_img; y0 = (blockIdx.x % a.tiles_per_img) * a.TH; }

    // ---- halo staging assignment: thread -> (halo pixel, float4 of the chunk), fixed for all chunks
    constexpr int Q = CK / 4;                              // float4 per halo pixel per chunk
    int a_off[A_IT];                                       // pixel offset in floats/ld units, -1 = zero fill
    int a_lds[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int idx = t + 256 * i;
        int hp = idx / Q, c4 = idx % Q;
        a_lds[i] = -1; a_off[i] = -1;
        if (hp < a.HP) {
            int ti = hp / (TH2 * W2);
            int rem = hp - ti * (TH2 * W2);
            int hy = rem / W2, hx = rem - hy * W2;
            int iy = y0 + hy - 1, ix = hx - 1, img = img0 + ti;
            a_lds[i] = hp * PITCH + c4 * 4;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && img < a.N)
                a_off[i] = (img * a.H + iy) * a.W + ix;
        }
    }
    // ---- MFMA row -> halo pixel (before the tap shift)
    int a_row[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int r = wm * (BM / 2) + i * 32 + (l & 31);
        int tx = r % a.W, q = r / a.W;
        int ty = q % a.TH, ti = q / a.TH;
        a_row[i] = ((ti * TH2 + ty) * W2 + tx) * PITCH + (l >> 5) * 8;
    }
    const int b_row0 = (wn * 64 + (l & 31)) * PITCH + (l >> 5) * 8;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_IT];
    uint4 rb[B_IT];

    auto load_a = [&](int kc) {
        const float* src = a.x; int ld = a.ldx; int cc = kc;
        if (kc >= a.K1) { src = a.x2; ld = a.ldx2; cc = kc - a.K1; }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int c4 = (t + 256 * i) % Q;
            // unconditional load from a clamped address, then select (no branch around the load)
            const bool ok = a_off[i] >= 0;
            float4 v = *reinterpret_cast<const float4*>(src + (size_t)(ok ? a_off[i] : 0) * ld + cc + c4 * 4);
            ra[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (a_lds[i] >= 0)
                *reinterpret_cast<uint2*>(&As[a_lds[i]]) = make_uint2(pack_bf16(ra[i].x, ra[i].y), pack_bf16(ra[i].z, ra[i].w));
    };
    auto load_b = [&](int tap, int kc) {
        const int wt = a.flip ? 8 - tap : tap;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int idx = t + 256 * i;
            int n = idx / (CK / 8), k8 = idx % (CK / 8);
            int ng = n0 + n;
            const bool ok = ng < a.Nc;
            uint4 v = *reinterpret_cast<const uint4*>(a.w + ((size_t)wt * a.Nc + (ok ? ng : 0)) * a.K + kc + k8 * 8);
            rb[i] = ok ? v : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int idx = t + 256 * i;
            int n = idx / (CK / 8), k8 = idx % (CK / 8);
            *reinterpret_cast<uint4*>(&Bs[buf * (BN * PITCH) + n * PITCH + k8 * 8]) = rb[i];
        }
    };

    const int nchunks = a.K / CK;
    load_a(0);
    load_b(0, 0);
    int buf = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int kc = ch * CK;
        __syncthreads();                       // everyone done reading the previous chunk's A tile / last B slot
        store_a();
        store_b(buf);
        __syncthreads();
        if (ch + 1 < nchunks) load_a(kc + CK); // next chunk's halo rides under the 9 taps of MFMAs
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // prefetch the next weight tile (next tap, or tap 0 of the next chunk)
            const bool last = (tap == 8);
            if (!last) load_b(tap + 1, kc);
            else if (ch + 1 < nchunks) load_b(0, kc + CK);
            const int ky = tap / 3, kx = tap - ky * 3;
            const int shift = (ky * W2 + kx) * PITCH;
            const uint16_t* Bt = Bs + buf * (BN * PITCH);
#pragma unroll
            for (int ks = 0; ks < CK / 16; ++ks) {
                bf16x8 af[MI], bf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&As[a_row[i] + shift + ks * 16]);
#pragma unroll
                for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(&Bt[b_row0 + j * 32 * PITCH + ks * 16]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            if (!last) {
                store_b(buf ^ 1);              // other slot: nobody reads it during this tap
                __syncthreads();
                buf ^= 1;
            }
        }
        buf ^= 1;                              // tap 0 of the next chunk goes to the other slot (stored above after sync)
    }

    // ---- epilogue
    const int Mtot = a.N * a.H * a.W;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            size_t m = (size_t)m0 + row;
            if (m >= (size_t)Mtot) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int col = n0 + wn * 64 + j * 32 + (l & 31);
                if (col >= a.Nc) continue;
                float v = acc[i][j][r];
                if (a.bias) v += a.bias[col];
                if (a.res) v += a.res[m * a.ldr + col];
                float* yp = a.y + m * a.ldy + col;
                if (a.accumulate) v += *yp;
                *yp = v;
            }
        }
}

template <int BM, int CK>
void launch_halo(const HaloArgs& a, hipStream_t st) {
    constexpr int PITCH = CK + 8;
    constexpr int MAXHP = (BM == 128) ? 288 : 160;
    size_t lds = (size_t)(MAXHP * PITCH + 2 * 128 * PITCH) * 2;
    dim3 grid((a.N * a.H * a.W + BM - 1) / BM, (a.Nc + 127) / 128);
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<BM, CK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
    }();
    (void)once;
    hipLaunchKernelGGL((conv3x3_halo_kernel<BM, CK>), grid, dim3(256), lds, st, a);
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

// Can the halo kernel take this descriptor?  (3x3, stride 1, pad 1, full-width row tiles)
static bool halo_ok(const MiConvDesc* d, int* bm, int* ck) {
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->mode != 1) return false;
    if (d->IH != d->OH || d->IW != d->OW) return false;
    if (d->K % 32 || d->K1 % 32 || d->Nc % 4) return false;
    int W = d->OW, H = d->OH;
    if (W < 4 || W > 64 || (128 % W)) return false;
    int BM = 128;
    // smaller tile when the grid would not fill the chip
    long tiles = ((long)d->N * H * W + 127) / 128 * ((d->Nc + 127) / 128);
    if (tiles < 320 && (64 % W) == 0) BM = 64;
    int rows = BM / W;                  // image rows per tile (may exceed H -> several images)
    if (rows <= H) { if (H % rows) return false; }
    else { if (rows % H) return false; if ((long)d->N % (rows / H)) return false; }
    int TH = rows <= H ? rows : H, TI = rows <= H ? 1 : rows / H;
    if (TI * (TH + 2) * (W + 2) > (BM == 128 ? 288 : 160)) {
        if (BM == 128) return false;
        BM = 128; rows = BM / W;                         // retry with the large tile
        if (rows <= H) { if (H % rows) return false; }
        else { if (rows % H) return false; if ((long)d->N % (rows / H)) return false; }
        TH = rows <= H ? rows : H; TI = rows <= H ? 1 : rows / H;
        if (TI * (TH + 2) * (W + 2) > 288) return false;
    }
    *bm = BM;
    *ck = (d->K % 64 == 0 && d->K1 % 64 == 0) ? 64 : 32;
    return true;
}

extern "C" int mi_conv3x3_bf16w(const MiConvDesc* d, const float* x, const float* x2, const void* w_nk_bf16,
                                const float* bias, const float* residual, float* y, void* stream) {
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
    int rows = BM / a.W;
    if (rows <= a.H) { a.TH = rows; a.TI = 1; a.tiles_per_img = a.H / rows; }
    else { a.TH = a.H; a.TI = rows / a.H; a.tiles_per_img = 1; }
    a.HP = a.TI * (a.TH + 2) * (a.W + 2);
    MI_REQUIRE(a.HP <= (BM == 128 ? 288 : 160), "halo tile too large");
    hipStream_t st = (hipStream_t)stream;
    if (BM == 128) { if (CK == 64) launch_halo<128, 64>(a, st); else launch_halo<128, 32>(a, st); }
    else           { if (CK == 64) launch_halo<64, 64>(a, st); else launch_halo<64, 32>(a, st); }
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
