// Weight gradients of the two stride-2 layers of the UNet for bf16-stored operands, on the machinery of wgrad_tr.hip (LDS-DMA
// staging, transposing LDS reads, all taps of a tile in one workgroup, several layers per launch):
//
//   Downsample = Conv2d(C, C, 3, stride 2, pad 1)          (reference src/models/ddpm.py:70)
//       dW[ky][kx][ci][co] += sum_{n,y,x} X[n, 2y+ky-1, 2x+kx-1, ci] * dY[n, y, x, co]
//   Upsample   = ConvTranspose2d(C, C, 4, stride 2, pad 1) (reference src/models/ddpm.py:79)
//       dW[ky][kx][ci][co] += sum_{n,y,x} X[n, y, x, ci] * dY[n, 2y+ky-1, 2x+kx-1, co]
//
// Both are "a big tensor B (2h x 2w) gathered at 2s + k - 1 against a small tensor S (h x w)"; only which of the two carries ci
// differs, and that is the order of the two MFMA operands (SWAP).  The stride disappears in the staging: LDS-DMA places 16 bytes
// per lane from any source address, so a big row is stored as its EVEN-column plane and its ODD-column plane, and every tap column
// becomes a unit-stride read of one plane:   kx = 0: odd plane, x-1   kx = 1: even, x   kx = 2: odd, x   kx = 3: even, x+1.
// The one pixel a shifted read takes from outside the row comes from a zero block (per-lane address select), so rows carry no
// padding.  Big rows are fetched exactly once: a step (64 small pixels = TR small rows) owns the 2 TR big rows 2 r0 + 1 ..
// 2 r0 + 2 TR (numbered across the whole batch, so an image's row 0 arrives with the step before it); it reads those, the last two
// rows of the previous step's slot, and a zero row above / below an image.
//   * 3x3: one workgroup = 64 (big channels) x 128 (small channels) x 9 taps, both planes staged (32 KB per step), a wave holds
//     32 x 32 x 9 = 144 accumulators, 36 MFMAs per step.
//   * 4x4: 16 taps would need 256 accumulators per wave, so a workgroup takes ONE column plane (8 taps, 128 accumulators, 16 KB of
//     big rows per step); the two planes are separate tiles.
// A step's DMA is requested one step ahead (ring of 3 big slots, 2 small slots).  k-slices, partial tiles in register order,
// fixed-order reduce and CU shares proportional to the work are as in wgrad_tr.hip.
#include "tr_common.h"

namespace {

struct S2Args {
    const uint16_t* B; const uint16_t* S;   // big [N][2h][2w][Cb] (pixel stride ldb), small [N][h][w][Cs] (pixel stride lds_)
    float* ws; float* dW;                   // dW [KS][KS][Ci][Cj]
    int w, h, KS, Cb, Cs, ldb, lds_, Ci, Cj;
    int rows2;                              // big rows in the batch: N * 2h
    int total, sps, splits;                 // steps of 64 small pixels, steps per k-slice, k-slices
    int gx, gy, ntiles;                     // big-channel tiles (64), small-channel tiles (128), tiles incl. the plane factor
    int wg0, tile0, xcd_map;
};
struct S2Batch { S2Args p[MAXP]; int n; };

// ---- compile-time schedule of a step.  A unit = one A fragment = (big row bi of the step, 16-pixel column group g, tap column);
//      it feeds the MFMA of every k-step j whose tap row lands on that big row.
template <int W, int KS> struct S2Shape {
    static constexpr int TR = 64 / W, JPR = W >= 16 ? W / 16 : 1, NBI = 2 * TR + 2, NKX = KS == 3 ? 3 : 2, MAXU = NBI * JPR * NKX;
};
template <int W, int KS> constexpr int s2_ky(int bi, int g, int j) {          // tap row of k-step j on unit row bi, or -1
    using Sh = S2Shape<W, KS>;
    if (W == 8) { const int ky = bi - 4 * j; return (ky >= 0 && ky < KS) ? ky : -1; }
    if (j % Sh::JPR != g) return -1;
    const int ky = bi - 2 * (j / Sh::JPR);
    return (ky >= 0 && ky < KS) ? ky : -1;
}
template <int W, int KS> constexpr bool s2_active(int u) {
    using Sh = S2Shape<W, KS>;
    const int bi = u / (Sh::JPR * Sh::NKX), g = (u / Sh::NKX) % Sh::JPR;
    for (int j = 0; j < 4; ++j) if (s2_ky<W, KS>(bi, g, j) >= 0) return true;
    return false;
}
template <int W, int KS> constexpr int s2_count() {
    int n = 0;
    for (int u = 0; u < S2Shape<W, KS>::MAXU; ++u) n += s2_active<W, KS>(u) ? 1 : 0;
    return n;
}
template <int W, int KS> constexpr int s2_nth(int k) {                        // k-th active unit
    for (int u = 0; u < S2Shape<W, KS>::MAXU; ++u)
        if (s2_active<W, KS>(u)) { if (k == 0) return u; --k; }
    return -1;
}

// W = small width (8, 16, 32); KS = 3 (conv, big = X, rows of the MFMA tile = big channels) / 4 (transposed conv, big = dY, SWAP)
template <int W, int KS>
__device__ __forceinline__ void wgrad_s2_body(const S2Args& a, const int wg, uint8_t* lds_raw) {
    using Sh = S2Shape<W, KS>;
    constexpr bool SWAP = KS == 4;
    constexpr int TR = Sh::TR, JPR = Sh::JPR, NKX = Sh::NKX;
    constexpr int NB = W / 8;                      // 8-pixel blocks of a plane row
    constexpr int NPL = KS == 3 ? 2 : 1;           // planes staged
    constexpr int PROWB = NB * 1024, BROWB = NPL * PROWB, SLOTB = 2 * TR * BROWB;
    constexpr int ZROW = 0, BRING = BROWB, SRING = BRING + 3 * SLOTB, SSTEP = 64 * 256;
    constexpr int NT = KS == 3 ? 9 : 8;            // taps (accumulator tiles) per wave
    constexpr int NBLK = 2 * TR * NPL * NB;        // big blocks per step (32 / 16)
    constexpr int NU = s2_count<W, KS>();
    constexpr int PD = 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;

    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wv >> 2, wj = wv & 3;             // 2 big-channel halves x 4 small-channel quarters
    // (k-slice, tile) in XCD rank order: a slice's tiles (they read the same rows) run on one XCD / one L2 -- see wgrad1x1_tr.hip
    int rank = wg;
    if (a.xcd_map) {
        const int Wg = a.ntiles * a.splits, x = (a.wg0 + wg) & 7;
        rank = (wg - ((x - a.wg0) & 7)) >> 3;
        for (int xx = 0; xx < x; ++xx) rank += (Wg - ((xx - a.wg0) & 7) + 7) >> 3;
    }
    const int split = rank / a.ntiles, tile = rank - split * a.ntiles;
    const int tb = tile % a.gx, tsm = (tile / a.gx) % a.gy, q = tile / (a.gx * a.gy);      // q: column plane of a 4x4 tile (0 even, 1 odd)
    const int cb0 = tb * 64, cs0 = tsm * 128;
    const int sb = split * a.sps, se = min(a.total, sb + a.sps);

    for (int i = t * 16; i < BRING; i += 512 * 16) *reinterpret_cast<u32x4*>(lds_raw + i) = u32x4{0u, 0u, 0u, 0u};

    // ---- DMA sources.  big block (8 plane pixels x 64 ch): lane -> half lane >> 5, pixel (lane >> 2) & 7, 8-channel chunk lane & 3
    //      small block (4 pixels x 128 ch): quarter lane >> 4, pixel (lane >> 2) & 3, chunk lane & 3
    const uint16_t* bsrc = a.B + (size_t)(2 * ((l >> 2) & 7)) * a.ldb + cb0 + (l >> 5) * 32 + (l & 3) * 8;
    const uint16_t* ssrc = a.S + (size_t)((l >> 2) & 3) * a.lds_ + min(cs0 + (l >> 4) * 32 + (l & 3) * 8, a.Cs - 8);
    const int last = a.total - 1, lastrow = a.rows2 - 1;
    auto big_block = [&](int row, int i_in_row, uint32_t dst_row) {          // one 8-pixel block of one plane of big row `row`
        const int pl = i_in_row / NB, b = i_in_row % NB;
        const int plane = KS == 3 ? pl : q;
        glds16(bsrc + ((size_t)min(max(row, 0), lastrow) * (2 * W) + 16 * b + plane) * a.ldb, dst_row + pl * PROWB + b * 1024);
    };
    auto stage = [&](int step, int bslot) {
        const int st = min(step, last);
#pragma unroll
        for (int k = 0; k < NBLK / 8; ++k) {
            const int i = wv + 8 * k;
            const int qrow = i / (NPL * NB);
            big_block(2 * st * TR + 1 + qrow, i % (NPL * NB), lds0 + BRING + bslot * SLOTB + qrow * BROWB);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = wv + 8 * k;
            glds16(ssrc + ((size_t)st * 64 + i * 4) * a.lds_, lds0 + SRING + (step & 1) * SSTEP + i * 1024);
        }
    };

    // ---- per-lane fragment offsets.  Transposing read: 16-lane group (l >> 4) & 1 = channels 0-15 / 16-31 of the wave's 32, piece row
    //      (l & 15) >> 2 = pixel within the 4-pixel block, piece column 4 * (l & 3) channels.  Plane pixel read by (group g, read r,
    //      shift s): 16 g + 8 half + 4 r + psub + s (W >= 16), or 4 r + psub + s of the half-wave's own row (W = 8).
    const int half = l >> 5, psub = (l & 15) >> 2;
    const int lane_b = ((l >> 4) & 1) * 32 + (l & 3) * 8;
    int fa[2][3];                                  // [read][shift + 1], bytes relative to the plane row, g = 0
    bool zl, zr;                                   // this lane's pixel is the one left of the row (shift -1, read 0, group 0) / right of it
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int idx = (W >= 16 ? 8 * half : 0) + 4 * r + psub + s - 1;
            fa[r][s] = (idx >> 3) * 1024 + (idx & 7) * 64 + wi * 512 + lane_b;
        }
    zl = (W >= 16 ? half == 0 : true) && psub == 0;
    zr = (W >= 16 ? half == 1 : true) && psub == 3;
    const uint32_t zaddr = lds0 + ZROW + wi * 512 + lane_b;
    const int fb = half * 2048 + psub * 64 + wj * 256 + lane_b;

    f32x16 acc[NT];
#pragma unroll
    for (int tp = 0; tp < NT; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

    __syncthreads();
    {   // the two rows the first step takes from its predecessor's slot: big rows 2 sb TR - 1 and 2 sb TR
        const int pslot = (sb + 2) % 3;
#pragma unroll
        for (int k = 0; k < (2 * NPL * NB + 7) / 8; ++k) {
            const int i = wv + 8 * k;
            if (i < 2 * NPL * NB) {
                const int qrow = i / (NPL * NB);
                big_block(2 * sb * TR - 1 + qrow, i % (NPL * NB), lds0 + BRING + pslot * SLOTB + (2 * TR - 2 + qrow) * BROWB);
            }
        }
    }
    int slot = sb % 3;
    stage(sb, slot);

    for (int s = sb; s < se; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // step s has landed ...
        __builtin_amdgcn_s_barrier();                         // ... for every wave, and every wave is done reading step s-1
        asm volatile("" ::: "memory");
        const int nslot = slot == 2 ? 0 : slot + 1, pslot = slot == 0 ? 2 : slot - 1;
        stage(s + 1, nslot);
        const int y0 = (s * TR) % a.h;
        // big rows of the step, bi = 0 .. 2 TR + 1  <->  image row 2 y0 - 1 + bi
        int RB[2 * TR + 2];
        RB[0] = y0 == 0 ? ZROW : BRING + pslot * SLOTB + (2 * TR - 2) * BROWB;
        RB[1] = BRING + pslot * SLOTB + (2 * TR - 1) * BROWB;
#pragma unroll
        for (int bi = 2; bi < 2 * TR + 2; ++bi) RB[bi] = BRING + slot * SLOTB + (bi - 2) * BROWB;
        if (y0 + TR == a.h) RB[2 * TR + 1] = ZROW;
        const uint32_t yb = lds0 + SRING + (s & 1) * SSTEP;
        bf16x8 bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = tr_pair(yb + fb + j * 4 * 1024, yb + fb + (j * 4 + 1) * 1024);
        auto load_unit = [&](auto kc) -> bf16x8 {
            constexpr int u = s2_nth<W, KS>(decltype(kc)::value);
            constexpr int bi = u / (JPR * NKX), g = (u / NKX) % JPR, kxi = u % NKX;
            // tap column -> plane slot in LDS and shift.  3x3: kx 0 odd/-1, 1 even/0, 2 odd/0.  4x4: plane q, kxi 0/1 -> odd: -1 / 0, even: 0 / +1
            uint32_t rb;
            if constexpr (W == 8) rb = lds0 + (half ? RB[bi + 2 <= 2 * TR + 1 ? bi + 2 : bi] : RB[bi]);
            else rb = lds0 + RB[bi] + g * 2048;
            if constexpr (KS == 3) {
                constexpr int pl = kxi == 1 ? 0 : 1, sh = kxi == 0 ? -1 : 0;
                uint32_t a0 = rb + pl * PROWB + fa[0][sh + 1];
                const uint32_t a1 = rb + pl * PROWB + fa[1][sh + 1];
                if constexpr (sh == -1 && g == 0) a0 = zl ? zaddr : a0;
                return tr_pair(a0, a1);
            } else {
                const int sh = q ? kxi - 1 : kxi;                             // wave-uniform
                uint32_t a0 = rb + (sh < 0 ? fa[0][0] : sh == 0 ? fa[0][1] : fa[0][2]);
                uint32_t a1 = rb + (sh < 0 ? fa[1][0] : sh == 0 ? fa[1][1] : fa[1][2]);
                if constexpr (g == 0) a0 = (sh < 0 && zl) ? zaddr : a0;
                if constexpr (g == JPR - 1) a1 = (sh > 0 && zr) ? zaddr : a1;
                return tr_pair(a0, a1);
            }
        };
        bf16x8 F[PD + 1];
        static_for<0, PD>([&](auto kc) { F[decltype(kc)::value] = load_unit(kc); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NU>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int u = s2_nth<W, KS>(k);
            constexpr int bi = u / (JPR * NKX), g = (u / NKX) % JPR, kxi = u % NKX;
            if constexpr (k + PD < NU) F[(k + PD) % (PD + 1)] = load_unit(std::integral_constant<int, k + PD>{});
            static_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int ky = s2_ky<W, KS>(bi, g, j);
                if constexpr (ky >= 0) {
                    constexpr int ta = KS == 3 ? ky * 3 + kxi : ky * 2 + kxi;
                    if constexpr (SWAP) acc[ta] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], F[k % (PD + 1)], acc[ta], 0, 0, 0);
                    else acc[ta] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[k % (PD + 1)], bfr[j], acc[ta], 0, 0, 0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        slot = nslot;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // nothing may still be writing LDS when the workgroup retires

    // accumulator tile: rows (A operand) x columns (B operand).  Rows carry ci, columns co in both cases.
    const int row0 = (SWAP ? cs0 + wj * 32 : cb0 + wi * 32) + 4 * (l >> 5);
    const int col = (SWAP ? cb0 + wi * 32 : cs0 + wj * 32) + (l & 31);
    auto tap_of = [&](int ta) { return KS == 3 ? ta : (ta >> 1) * 4 + (q ? ((ta & 1) ? 2 : 0) : ((ta & 1) ? 3 : 1)); };
    if (a.splits == 1) {
        const bool ok_col = SWAP ? true : col < a.Cs;
#pragma unroll
        for (int ta = 0; ta < NT; ++ta) {
            float* o = a.dW + ((size_t)tap_of(ta) * a.Ci + row0) * a.Cj + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                if (ok_col && (!SWAP || row0 + rr < a.Cs)) o[(size_t)rr * a.Cj] += acc[ta][r];
            }
        }
        return;
    }
    float* out = a.ws + (size_t)(split * a.ntiles + tile) * (NT * 4 * 2048) + t * 4;
#pragma unroll
    for (int ta = 0; ta < NT; ++ta)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            *reinterpret_cast<f32x4*>(out + (ta * 4 + rq) * 2048) =
                f32x4{acc[ta][4 * rq], acc[ta][4 * rq + 1], acc[ta][4 * rq + 2], acc[ta][4 * rq + 3]};
}

__global__ __launch_bounds__(512, 1) void wgrad_s2_tr_kernel(const S2Batch b) {
    MI_PRIO_UP();
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    int p = 0;
#pragma unroll
    for (int k = 1; k < MAXP; ++k)
        if (k < b.n && (int)blockIdx.x >= b.p[k].wg0) p = k;
    const S2Args& a = b.p[p];
    const int wg = blockIdx.x - a.wg0;
    if (a.KS == 3) {
        switch (a.w) {
            case 8:  wgrad_s2_body<8, 3>(a, wg, lds_raw); break;
            case 16: wgrad_s2_body<16, 3>(a, wg, lds_raw); break;
            default: wgrad_s2_body<32, 3>(a, wg, lds_raw); break;
        }
    } else {
        switch (a.w) {
            case 8:  wgrad_s2_body<8, 4>(a, wg, lds_raw); break;
            case 16: wgrad_s2_body<16, 4>(a, wg, lds_raw); break;
            default: wgrad_s2_body<32, 4>(a, wg, lds_raw); break;
        }
    }
}

// dW += sum over k-slices of the partial tiles (fixed order).  grid = (position groups, slots of the largest problem, tiles of all
// problems that have k-slices); a workgroup = G slice groups x (256 / G) float4 positions of one slot.
template <int G>
__global__ __launch_bounds__(256) void wgrad_s2_reduce_kernel(const S2Batch b) {
    MI_PRIO_UP();
    constexpr int IB = 256 / G;
    __shared__ f32x4 red[G][IB];
    int pi = 0;
#pragma unroll
    for (int k = 1; k < MAXP; ++k)
        if (k < b.n && b.p[k].splits > 1 && (int)blockIdx.z >= b.p[k].tile0) pi = k;
    const S2Args& a = b.p[pi];
    const bool swap = a.KS == 4;
    const int nslot = (a.KS == 3 ? 9 : 8) * 4;
    const int slot = blockIdx.y;
    if (slot >= nslot) return;
    const int it = threadIdx.x % IB, grp = threadIdx.x / IB;
    const int tt = blockIdx.x * IB + it;                       // thread of the producing workgroup
    const int tile = blockIdx.z - a.tile0;
    const size_t tile_fl = (size_t)nslot * 2048;
    const float* p = a.ws + (size_t)tile * tile_fl + (size_t)slot * 2048 + tt * 4;
    const size_t stride = (size_t)a.ntiles * tile_fl;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int sp = grp;
    for (; sp + 7 * G < a.splits; sp += 8 * G) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + k * G) * stride);
        s0 += v[0] + v[4]; s1 += v[1] + v[5]; s2 += v[2] + v[6]; s3 += v[3] + v[7];
    }
    for (; sp < a.splits; sp += G) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)sp * stride);
    f32x4 s = (s0 + s1) + (s2 + s3);
    red[grp][it] = s;
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int g = 1; g < G; ++g) s += red[g][it];
    const int wv = tt >> 6, l = tt & 63, wi = wv >> 2, wj = wv & 3;
    const int rq = slot & 3, ta = slot >> 2;
    const int tb = tile % a.gx, tsm = (tile / a.gx) % a.gy, q = tile / (a.gx * a.gy);
    const int cb0 = tb * 64, cs0 = tsm * 128;
    const int row = (swap ? cs0 + wj * 32 : cb0 + wi * 32) + 8 * rq + 4 * (l >> 5);
    const int col = (swap ? cb0 + wi * 32 : cs0 + wj * 32) + (l & 31);
    const int tap = a.KS == 3 ? ta : (ta >> 1) * 4 + (q ? ((ta & 1) ? 2 : 0) : ((ta & 1) ? 3 : 1));
    if (col >= a.Cj) return;
    float* out = a.dW + ((size_t)tap * a.Ci + row) * a.Cj + col;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (row + e < a.Ci) out[(size_t)e * a.Cj] += s[e];
}

bool s2_ok(const MiWgradDesc* d) {
    if (d->stride != 2 || d->pad != 1 || d->mode != 1 || d->KH != d->KW) return false;
    if (!((d->KH == 3 && d->gather_i) || (d->KH == 4 && !d->gather_i))) return false;
    if (d->GH != 2 * d->DH || d->GW != 2 * d->DW || d->I1 != d->Ci) return false;
    const int w = d->DW, h = d->DH;
    if (w != 8 && w != 16 && w != 32) return false;
    if (h % (64 / w)) return false;
    if (((long)d->N * h * w) % 64) return false;
    const int Cb = d->gather_i ? d->Ci : d->Cj, Cs = d->gather_i ? d->Cj : d->Ci;
    if (Cb % 64 || Cs % 32 || Cs < 32) return false;
    if (d->ldp % 8 || d->ldq % 8) return false;
    return true;
}

long s2_target() {
    static const long env_target = mi_knob("MI_WS2_BLOCKS", 256);
    return env_target;
}

void s2_plan(const MiWgradDesc* d, S2Args& a, long wgs) {
    a.w = d->DW; a.h = d->DH; a.KS = d->KH; a.Ci = d->Ci; a.Cj = d->Cj;
    a.Cb = d->gather_i ? d->Ci : d->Cj; a.Cs = d->gather_i ? d->Cj : d->Ci;
    a.ldb = d->gather_i ? d->ldp : d->ldq; a.lds_ = d->gather_i ? d->ldq : d->ldp;
    a.rows2 = d->N * 2 * d->DH;
    a.gx = a.Cb / 64; a.gy = (a.Cs + 127) / 128;
    a.ntiles = a.gx * a.gy * (a.KS == 4 ? 2 : 1);
    a.total = (int)((long)d->N * d->DH * d->DW / 64);
    long splits = wgs / a.ntiles;
    if (splits < 1) splits = 1;
    if (splits > a.total) splits = a.total;
    a.sps = (int)((a.total + splits - 1) / splits);
    a.splits = (a.total + a.sps - 1) / a.sps;
}

long s2_tiles(const MiWgradDesc* d) {
    const int Cb = d->gather_i ? d->Ci : d->Cj, Cs = d->gather_i ? d->Cj : d->Ci;
    return (long)(Cb / 64) * ((Cs + 127) / 128) * (d->KH == 4 ? 2 : 1);
}

void s2_shares(int n, const MiWgradDesc* d, long* wgs) {
    double tot = 0, fl[MAXP];
    for (int i = 0; i < n; ++i) { fl[i] = (double)d[i].N * d[i].DH * d[i].DW * d[i].Ci * d[i].Cj * d[i].KH * d[i].KW; tot += fl[i]; }
    const long target = s2_target();
    static const int greedy = (int)mi_knob("MI_WS2_BALANCE", 1);
    if (greedy) {
        long tiles[MAXP];
        for (int i = 0; i < n; ++i) tiles[i] = s2_tiles(&d[i]);
        balance_shares(n, fl, tiles, target, wgs);
        return;
    }
    for (int i = 0; i < n; ++i) {
        const long tiles = s2_tiles(&d[i]);
        long w = (long)(target * fl[i] / tot + 0.5);
        w = w / tiles * tiles;
        wgs[i] = w < tiles ? tiles : w;
    }
}

size_t s2_lds(const S2Args& a) {
    const size_t browb = (size_t)(a.KS == 3 ? 2 : 1) * (a.w / 8) * 1024;
    return browb + 3 * (2 * (64 / a.w)) * browb + 2 * 64 * 256;
}
size_t s2_ws_floats(const S2Args& a) { return a.splits > 1 ? (size_t)a.splits * a.ntiles * (a.KS == 3 ? 36 : 32) * 2048 : 0; }

int g_ws2_phase = 0;

}  // namespace

extern "C" int mi_conv_s2_wgrad_tr_supported(const MiWgradDesc* d) { return (d && s2_ok(d)) ? 1 : 0; }

extern "C" size_t mi_conv_s2_wgrad_tr_batch_workspace(int n, const MiWgradDesc* descs) {
    if (!descs || n < 1 || n > MAXP) return 0;
    long wgs[MAXP];
    for (int i = 0; i < n; ++i) if (!s2_ok(&descs[i])) return 0;
    s2_shares(n, descs, wgs);
    size_t fl = 0;
    for (int i = 0; i < n; ++i) { S2Args a; s2_plan(&descs[i], a, wgs[i]); fl += s2_ws_floats(a); }
    return fl * sizeof(float) + 256;
}

extern "C" int mi_debug_wgrad_s2_tr_phase(int phase) {
    if (phase < 0 || phase > 2) return mi_set_error(-1, "mi_debug_wgrad_s2_tr_phase: phase in 0..2");
    g_ws2_phase = phase;
    return 0;
}

extern "C" int mi_conv_s2_wgrad_tr_batch(int n, const MiWgradDesc* descs, const void* const* P, const void* const* Q,
                                         float* const* dW, void* workspace, size_t ws_bytes, void* stream) {
    MI_REQUIRE(n >= 1 && n <= MAXP && descs && P && Q && dW, "1..8 problems, non-null arrays");
    S2Batch b;
    b.n = n;
    long wgs[MAXP];
    for (int i = 0; i < n; ++i) {
        MI_REQUIRE(s2_ok(&descs[i]), "descriptor not supported by the stride-2 LDS-DMA weight-gradient kernel (use mi_conv_wgrad)");
        MI_REQUIRE(P[i] && Q[i] && dW[i], "null operand");
        MI_REQUIRE((((uintptr_t)P[i] | (uintptr_t)Q[i]) & 15) == 0, "operands must be 16-byte aligned");
    }
    s2_shares(n, descs, wgs);
    size_t off = 0, lds = 0;
    int wg = 0, tile = 0, max_splits = 1;
    for (int i = 0; i < n; ++i) {
        S2Args& a = b.p[i];
        s2_plan(&descs[i], a, wgs[i]);
        a.B = (const uint16_t*)(descs[i].gather_i ? P[i] : Q[i]);
        a.S = (const uint16_t*)(descs[i].gather_i ? Q[i] : P[i]);
        a.dW = dW[i];
        a.ws = (float*)workspace + off;
        off += s2_ws_floats(a);
        static const int xcd_env = (int)mi_knob("MI_WS2_XCD", 1);
        a.xcd_map = xcd_env && a.ntiles > 1 && a.splits > 1;
        a.wg0 = wg; wg += a.ntiles * a.splits;
        a.tile0 = tile; if (a.splits > 1) tile += a.ntiles;
        if (a.splits > max_splits) max_splits = a.splits;
        const size_t l = s2_lds(a);
        if (l > lds) lds = l;
    }
    MI_REQUIRE(off == 0 || (workspace && ((uintptr_t)workspace & 15) == 0 && ws_bytes >= off * sizeof(float)),
               "workspace too small (mi_conv_s2_wgrad_tr_batch_workspace)");
    hipStream_t st = (hipStream_t)stream;
    static MiPerDevice once;
    once.run([] {
        (void)hipFuncSetAttribute((const void*)wgrad_s2_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (g_ws2_phase != 2) hipLaunchKernelGGL(wgrad_s2_tr_kernel, dim3((unsigned)wg), dim3(512), lds, st, b);
    if (g_ws2_phase != 1 && tile > 0) {
        if (max_splits >= 64) hipLaunchKernelGGL(wgrad_s2_reduce_kernel<8>, dim3(16, 36, tile), dim3(256), 0, st, b);
        else hipLaunchKernelGGL(wgrad_s2_reduce_kernel<2>, dim3(4, 36, tile), dim3(256), 0, st, b);
    }
    MI_LAUNCH_CHECK();
    return 0;
}
