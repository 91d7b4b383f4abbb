// Weight gradient of the 3x3 / stride 1 / pad 1 convolution (Block's conv, reference src/models/ddpm.py:116) for bf16-stored
// operands, built on the two gfx950 LDS instructions that make the register staging of wgrad3x3.hip unnecessary:
//
//     dW[ky][kx][ci][co] += sum_{n,y,x} X[n, y+ky-1, x+kx-1, ci] * dY[n, y, x, co]
//
//   * LDS-DMA (global_load_lds_dwordx4): the NHWC rows of X and dY go from L2 straight into LDS, 1 KB per wave instruction, no
//     VGPRs, no VALU, no ds_write.  The destination is wave-uniform base + 16 * lane, so the LDS image is chosen through the
//     per-lane SOURCE address: X is kept as [row][8-pixel block][32-channel half][pixel][32 ch] and dY as
//     [4-pixel block][32-channel quarter][pixel][32 ch] -- consecutive pixels of one channel group are 64 bytes apart, which
//     spreads the four pixels of a transposing read over the four quarters of a bank row (conflict-free).
//   * ds_read_b64_tr_b16: the contraction axis (pixels) is the strided axis of both NHWC operands; the transposing read hands each
//     lane 4 consecutive PIXELS of its own channel (a 16-lane group reads a [4 pixels][16 channels] block, every lane giving the
//     address of its own 8-byte piece), so a tap shift (kx, ky) is nothing but a different address: all nine taps read the
//     same staged rows.
//
// One workgroup (8 waves, two per SIMD) owns a 64 (ci) x 128 (co) tile of ALL NINE taps for one slice of the pixel axis:
// a wave accumulates 32 ci x 32 co x 9 taps = 9 MFMA tiles (144 accumulator registers; 18 tiles per wave do not fit the 256
// accumulation registers and hipcc then shuttles them through v_accvgpr moves), so per 16-pixel k-step it issues at most 20
// transposing reads for 9 v_mfma_f32_32x32x16_bf16 (fewer: the row a tap (ky) of one output row reads is the row tap (ky-1) of the
// next output row reads, and the compiler keeps such fragments in registers).  The pixel axis is walked in steps of 64 pixels
// (TR = 64 / W image rows);
// X rows live in a ring of 4 steps (a step's halo rows are its neighbours' rows: every X row is fetched once), dY in a ring of 3.
// Step s+2 is requested while step s runs on the matrix cores; one s_waitcnt vmcnt(0) + s_barrier per step (36 MFMAs per wave).
// Image borders are a never-written zero row / zero pixel blocks in LDS.  Slices leave as partial tiles in register order (1 KB
// contiguous per wave and store) and wgrad_tr_reduce_kernel sums them in a fixed order (deterministic) into dW.
#include "tr_common.h"

#ifndef MI_WTR_PD
#define MI_WTR_PD 4
#endif
#ifndef MI_WTR_STAGE_AT
#define MI_WTR_STAGE_AT -1  // >= 0: the unit of a step behind whose MFMAs the requests of step s + 2 are issued instead of at the top of the step (round 5, measured: 3115 -> 3253 cycles per step, off)
#endif
#ifndef MI_WTR_VMCNT2
#define MI_WTR_VMCNT2 0   // 1: leave the newest dY pieces in flight across the step barrier (measured: 0.640 vs 0.635 ms per step, no gain)
#endif
#ifndef MI_WTR_ABL
#define MI_WTR_ABL 0     // profiling only: 1 no partial-tile stores, 2 no main loop
#endif

#ifdef MI_WTR_TIMING
// profiling build only (-DMI_WTR_TIMING): per-workgroup phase timestamps, 100 MHz wall clock (tools/wtr_timeline.py)
__device__ unsigned long long g_wtr_ts[6 * 1024];
__device__ unsigned long long g_wtr_cyc[1024];       // steps of the workgroup (slot 5 of g_wtr_ts: wait cycles << 40 | work cycles, wave 0)
#define MI_TS(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_wtr_ts[(k) * 1024 + blockIdx.x] = wall_clock64(); } while (0)
extern "C" int mi_debug_wtr_ts(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wtr_ts), sizeof(g_wtr_ts));
}
extern "C" int mi_debug_wtr_steps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wtr_cyc), sizeof(g_wtr_cyc));
}
#else
#define MI_TS(k) do {} while (0)
#endif

namespace {

// W = image width (8, 16, 32 or 64); a step is TR = 64 / W rows of one image (H % TR == 0).  wg = workgroup index within the problem.
template <int W>
__device__ __forceinline__ void wgrad_tr_body(const TrArgs& a, const int wg, uint8_t* lds_raw) {
    constexpr int TR = 64 / W;
    constexpr int NBLK = W / 8;                    // interior 8-pixel blocks of a row
    constexpr int ROWB = (NBLK + 2) * 1024;        // bytes of an X row in LDS: zero block | interior | zero block
    constexpr int NR = 4 * TR;                     // X ring: rows of 4 steps
    constexpr int XRING = ROWB;                    // byte offsets: [zero row][X ring][dY ring]
    constexpr int YRING = XRING + NR * ROWB;
    constexpr int YSTEP = 64 * 256;                // 64 pixels x 128 channels of bf16
    constexpr int JPR = W >= 16 ? W / 16 : 1;      // 16-pixel k-steps per image row (W = 8: a k-step is two rows)
    constexpr int NRI = (W == 8 ? 6 : TR - 1) + 3; // X rows a step's taps touch, counted from relative row -1
    constexpr int NU = NRI * JPR * 3;              // X fragments per step
    constexpr int PD = MI_WTR_PD;                  // fragments fetched ahead of the MFMAs that consume them
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;

    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wv >> 2, wj = wv & 3;             // 2 ci halves x 4 co quarters
    const int ntiles = a.gx * a.gy;
    // k-slice and tile of this workgroup.  The tiles of one k-slice read the same X / dY rows; consecutive workgroup ids go to
    // different XCDs (id % 8), each with its own L2 -- when the slices divide evenly, ids xcd + 8*slot with slot = tile + ntiles*m
    // belong to slice xcd + 8*m, so that one L2 serves all tiles of a slice (the problem starts at a multiple of 8: host).
    int split = wg / ntiles, tile = wg - split * ntiles;
    if (a.xcd_map == 1) {
        const int xcd = wg & 7, slot = wg >> 3;
        tile = slot % ntiles; split = xcd + 8 * (slot / ntiles);
    } else if (a.xcd_map == 2) {
        // fewer than 8 slices: g = 8 / splits XCDs per slice, each takes ntiles / g consecutive tiles (consecutive tiles share their
        // co tile, i.e. their dY rows) -- ids xcd + 8*slot -> slice xcd / g, tile (xcd % g) * (ntiles / g) + slot
        const int xcd = wg & 7, slot = wg >> 3, g = 8 / a.splits;
        split = xcd / g; tile = (xcd % g) * (ntiles / g) + slot;
    }
    const int ci0 = (tile % a.gx) * 64, co0 = (tile / a.gx) * 128;
    const int sb = split * a.sps, se = min(a.total, sb + a.sps);

    MI_TS(0);
    // ---- zero the X area once: the halo pixel blocks and the zero row are never written again
    for (int i = t * 16; i < YRING; i += 512 * 16) *reinterpret_cast<u32x4*>(lds_raw + i) = u32x4{0u, 0u, 0u, 0u};

    // ---- DMA sources.  X block (8 pixels x 64 ch): lane -> half h = lane >> 5, pixel (lane >> 2) & 7, 8-channel chunk lane & 3.
    //      dY block (4 pixels x 128 ch): quarter lane >> 4, pixel (lane >> 2) & 3, chunk lane & 3.
    const bool second = ci0 >= a.I1;
    const int ldx = second ? a.ldp2 : a.ldp;
    const uint16_t* xsrc = (second ? a.P2 : a.P) + (size_t)((l >> 2) & 7) * ldx +
                           min((second ? ci0 - a.I1 : ci0) + (l >> 5) * 32 + (l & 3) * 8, (second ? a.Ci - a.I1 : a.I1) - 8);
    const uint16_t* ysrc = a.Q + (size_t)((l >> 2) & 3) * a.ldq + min(co0 + (l >> 4) * 32 + (l & 3) * 8, a.Cj - 8);
    const int last = a.total - 1;
    auto stage = [&](int step) {                   // X rows and dY pixels of `step` -> ring slots (sources clamped to the batch)
        const size_t pix0 = (size_t)min(step, last) * 64;
        const int sm = step & 3;
        {
            const int i = wv;                      // X block i of the step (8 per step, one per wave): row i / NBLK, block i % NBLK
            const int rr = i / NBLK, b = i % NBLK;
            glds16(xsrc + (pix0 + i * 8) * ldx, lds0 + XRING + (sm * TR + rr) * ROWB + (b + 1) * 1024);
        }
        const int ys = step % 3;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = wv + 8 * k;              // dY block i (16 per step): pixels 4i .. 4i+3
            glds16(ysrc + (pix0 + i * 4) * a.ldq, lds0 + YRING + ys * YSTEP + i * 1024);
        }
    };

    // ---- per-lane fragment offsets (bytes).  Lane l of a transposing read: 16-lane group (l >> 4) & 1 = channels 0-15 / 16-31 of
    //      the wave's 32, piece row (l & 15) >> 2 = pixel within the 4-pixel block, piece column 4 * (l & 3) channels.
    //      Pixel of the step read by (k-step j, read r): p = 16j + 8*half + 4r + psub.  Everything that depends on j is a
    //      compile-time constant; the lane-dependent part of an X address is fa[r][kx] (pixel m = 4r + psub of an 8-pixel block
    //      shifted by kx - 1: the shift may cross into the neighbouring block) and, for W >= 16, 1 KB for the upper half-wave.
    const int half = l >> 5, psub = (l & 15) >> 2;
    const int lane_b = ((l >> 4) & 1) * 32 + (l & 3) * 8;
    int fa[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int mm = r * 4 + psub + 7 + kx;                    // physical pixel within/after the block that holds x = 8q
            fa[r][kx] = (mm >> 3) * 1024 + (mm & 7) * 64 + wi * 512 + lane_b + (W >= 16 ? half * 1024 : 0);
        }
    const int fb = half * 2048 + psub * 64 + wj * 256 + lane_b;      // dY: 4-pixel block 4j + 2*half + r, channel quarter wj

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

    __syncthreads();
    MI_TS(1);
    if (sb > 0) {                                  // the row above the slice's first step (X only matters; dY slot is overwritten later)
        const int step = sb - 1;
        const size_t pix0 = (size_t)step * 64;
        const int i = wv;
        glds16(xsrc + (pix0 + i * 8) * ldx, lds0 + XRING + ((step & 3) * TR + i / NBLK) * ROWB + (i % NBLK + 1) * 1024);
    }
    stage(sb);
    stage(sb + 1);
#ifdef MI_WTR_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MI_TS(2);
#endif

#ifdef MI_WTR_TIMING
    unsigned long long tw_wait = 0, tw_work = 0, tw_n = 0;
#endif
    for (int s = sb; s < ((MI_WTR_ABL & 2) ? sb + 1 : se); ++s) {
#ifdef MI_WTR_TIMING
        const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
        // everything this step reads has landed: X and dY of steps <= s, the X row of step s+1 (halo below).  (MI_WTR_VMCNT2: the two
        // dY pieces of step s+1 are the newest requests and could stay in flight -- loads complete in order.)
#if MI_WTR_VMCNT2
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if !(MI_WTR_ABL & 4)      // (profiling only, 4: no step barrier -- wrong results, the time a perfectly hidden hand-over would leave)
        __builtin_amdgcn_s_barrier();                         // ... for every wave, and every wave is done reading step s-1
#endif
        asm volatile("" ::: "memory");
#ifdef MI_WTR_TIMING
        const unsigned long long tq1 = __builtin_amdgcn_s_memtime();
#endif
        // (round 5 experiment, MI_WTR_STAGE_AT: these three requests behind the step's first MFMAs instead of in front of them -- slower)
        if constexpr (MI_WTR_STAGE_AT < 0) stage(s + 2);
        const int sm = s & 3, y0 = (s * TR) % a.H;
        // row bases of relative rows -1 .. TR (ring slot, or the zero row at the image border)
        int RB[TR + 2];
        RB[0] = y0 == 0 ? 0 : XRING + ((sm * TR + NR - 1) % NR) * ROWB;
#pragma unroll
        for (int rr = 0; rr < TR; ++rr) RB[rr + 1] = XRING + (sm * TR + rr) * ROWB;
        RB[TR + 1] = y0 + TR == a.H ? 0 : XRING + (((sm + 1) & 3) * TR) * ROWB;
        const uint32_t yb = lds0 + YRING + (s % 3) * YSTEP;
        // The step's 36 MFMAs per wave, ordered by the X fragment they consume: a fragment = (X row ri, 16-pixel column group,
        // tap column kx); it serves every output row whose tap ky lands on that X row (up to three MFMAs).  Fragments are
        // fetched PD units ahead of their MFMAs through a ring of PD + 1 register sets.
        bf16x8 bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = tr_pair(yb + fb + j * 4 * 1024, yb + fb + (j * 4 + 1) * 1024);
        bf16x8 abl_last = bfr[0];                              // (profiling only, MI_WTR_ABL & 8)
        auto load_unit = [&](auto uc) -> bf16x8 {
            constexpr int u = decltype(uc)::value;
            constexpr int ri = u / (JPR * 3), qidx = (u / 3) % JPR, kx = u % 3;
#if MI_WTR_ABL & 8       // only the centre tap column is read from LDS, the others reuse a fragment already in registers (wrong results:
            if constexpr (kx != 1) return abl_last;              //  the time a kernel that derived the shifted fragments for free would take)
#endif
            uint32_t rb;
            if constexpr (W == 8) rb = lds0 + (half ? RB[ri + 1] : RB[ri]);          // lanes 32-63 read the next row
            else rb = lds0 + RB[ri] + qidx * 2048;                                    // 16 pixels = two 8-pixel blocks
#if MI_WTR_ABL & 8
            abl_last = tr_pair(rb + fa[0][kx], rb + fa[1][kx]);
            return abl_last;
#else
            return tr_pair(rb + fa[0][kx], rb + fa[1][kx]);
#endif
        };
        bf16x8 F[PD + 1];
        static_for<0, PD>([&](auto uc) { F[decltype(uc)::value] = load_unit(uc); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int ri = u / (JPR * 3), qidx = (u / 3) % JPR, kx = u % 3;
            if constexpr (u + PD < NU) F[(u + PD) % (PD + 1)] = load_unit(std::integral_constant<int, u + PD>{});
            static_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int r0 = W == 8 ? 2 * j : j / JPR, jq = W == 8 ? 0 : j % JPR;
                constexpr int ky = ri - r0;
                if constexpr (jq == qidx && ky >= 0 && ky <= 2)
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[u % (PD + 1)], bfr[j], acc[ky * 3 + kx], 0, 0, 0);
            });
            if constexpr (MI_WTR_STAGE_AT >= 0 && u == (MI_WTR_STAGE_AT < NU ? MI_WTR_STAGE_AT : NU - 1)) stage(s + 2);
            // pin the software pipeline: hipcc otherwise sinks every fragment read to just before its MFMA (read, wait for it,
            // one MFMA, next read ...), which runs at half the MFMA rate; nothing may move across a unit boundary
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef MI_WTR_TIMING
        { const unsigned long long tq2 = __builtin_amdgcn_s_memtime(); tw_wait += tq1 - tq0; tw_work += tq2 - tq1; ++tw_n; }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // nothing may still be writing LDS when the workgroup retires
    MI_TS(3);
#ifdef MI_WTR_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 1024) { g_wtr_ts[5 * 1024 + blockIdx.x] = (tw_wait << 40) | ((tw_work & 0xffffffffffull)); g_wtr_cyc[blockIdx.x] = tw_n; }
#endif

#if MI_WTR_ABL & 1      // profiling only: no partial-tile stores (one store keeps the accumulators live)
    {
        float v = 0.f;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) v += acc[tp][r];
        if (v == 123.456f) a.ws[t] = v;
        return;
    }
#endif
    if (a.splits == 1) {
        // the only k-slice of its tile: add into dW (lanes 0-31 = 32 consecutive co: 128-byte row segments)
        const int ci = ci0 + wi * 32 + 4 * (l >> 5), co = co0 + wj * 32 + (l & 31);
        if (co < a.Cj) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                float* o = a.dW + ((size_t)tp * a.Ci + ci) * a.Cj + co;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * a.Cj] += acc[tp][r];
            }
        }
        return;
    }
    // ---- partial tile in register order: slot tap*4 + rq holds, for thread t (of 512), accumulator registers 4rq .. 4rq+3
    float* out = a.ws + (size_t)(split * ntiles + tile) * (36 * 2048) + t * 4;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            *reinterpret_cast<f32x4*>(out + (tp * 4 + rq) * 2048) =
                f32x4{acc[tp][4 * rq], acc[tp][4 * rq + 1], acc[tp][4 * rq + 2], acc[tp][4 * rq + 3]};
#ifdef MI_WTR_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MI_TS(4);
#endif
}

__global__ __launch_bounds__(512, 1) void wgrad_tr_kernel(const TrBatch b) {
    MI_PRIO_UP();
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    int p = 0;
#pragma unroll
    for (int q = 1; q < MAXP; ++q)
        if (q < b.n && (int)blockIdx.x >= b.p[q].wg0) p = q;
    const TrArgs& a = b.p[p];
    const int wg = blockIdx.x - a.wg0;
    switch (a.W) {
        case 8:  wgrad_tr_body<8>(a, wg, lds_raw); break;
        case 16: wgrad_tr_body<16>(a, wg, lds_raw); break;
        case 32: wgrad_tr_body<32>(a, wg, lds_raw); break;
        default: wgrad_tr_body<64>(a, wg, lds_raw); break;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same weight gradient in exact-fp32 mode (Unet.compute_mode = "fp32": fp32 X and dY, v_mfma_f32_32x32x2_f32 = an fp32 fma
// chain).  Same decomposition -- one workgroup (8 waves) = a 64 (ci) x 128 (co) tile of ALL NINE taps for one slice of the pixel
// axis, 144 accumulators per wave, 64-pixel steps, the same k-slices, partial tiles, reduce kernel and batching -- but no transposing
// reads are needed: the fp32 MFMA contracts over TWO pixels per instruction, lane half k supplies pixel 2 kp + k, so both operands
// are plain ds_read_b32 of a pixel-major tile (32 lanes = 32 consecutive channels = one 128-byte row segment, conflict-free):
//   X  [row][zero pixel | W pixels | zero pixel][64 ci] fp32 (256 bytes per pixel; a tap shift is a different address),
//   dY [pixel][128 co] fp32.
// Both arrive by LDS-DMA (1 KB = 4 X pixels or 2 dY pixels per wave instruction, plain row-major sources).  X rows live in a ring
// of 4 steps (the step after the current one supplies the halo row below), dY in a ring of 2; step s requests X of step s + 2 and dY
// of step s + 1: a step is 288 MFMAs of 64 cycles per wave (~9 us), far longer than any load.  150 KB of LDS, one workgroup per CU.
__device__ __forceinline__ float lds_f32(uint32_t addr) {
    typedef __attribute__((address_space(3))) float lds_float;
    return *(lds_float*)(uintptr_t)addr;
}

template <int W>
__device__ __forceinline__ void wgrad_tr32_body(const TrArgs& a, const int wg, uint8_t* lds_raw) {
    constexpr int TR = 64 / W;
    constexpr int ROWB = (W + 2) * 256;
    constexpr int NR = 4 * TR;
    constexpr int XRING = ROWB;                    // byte offsets: [zero row][X ring][dY ring]
    constexpr int YRING = XRING + NR * ROWB;
    constexpr int YSTEP = 64 * 512;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wv >> 2, wj = wv & 3;             // 2 ci halves x 4 co quarters
    const int ntiles = a.gx * a.gy;
    int split = wg / ntiles, tile = wg - split * ntiles;
    if (a.xcd_map == 1) {
        const int xcd = wg & 7, slot = wg >> 3;
        tile = slot % ntiles; split = xcd + 8 * (slot / ntiles);
    } else if (a.xcd_map == 2) {
        const int xcd = wg & 7, slot = wg >> 3, g = 8 / a.splits;
        split = xcd / g; tile = (xcd % g) * (ntiles / g) + slot;
    }
    const int ci0 = (tile % a.gx) * 64, co0 = (tile / a.gx) * 128;
    const int sb = split * a.sps, se = min(a.total, sb + a.sps);

    // zero the X area once: the zero row and the zero pixels left / right of every row are never written again
    for (int i = t * 16; i < YRING; i += 512 * 16) *reinterpret_cast<u32x4*>(lds_raw + i) = u32x4{0u, 0u, 0u, 0u};

    // DMA sources.  X piece (4 pixels x 64 ci): lane -> pixel lane >> 4, 4-channel chunk lane & 15; dY piece (2 pixels x 128 co):
    // pixel lane >> 5, chunk lane & 31
    const bool second = ci0 >= a.I1;
    const int ldx = second ? a.ldp2 : a.ldp;
    const float* xsrc = reinterpret_cast<const float*>(second ? a.P2 : a.P) + (size_t)(l >> 4) * ldx +
                        min((second ? ci0 - a.I1 : ci0) + (l & 15) * 4, (second ? a.Ci - a.I1 : a.I1) - 4);
    const float* ysrc = reinterpret_cast<const float*>(a.Q) + (size_t)(l >> 5) * a.ldq + min(co0 + (l & 31) * 4, a.Cj - 4);
    const int last = a.total - 1;
    auto stage_x = [&](int step) {
        const size_t pix0 = (size_t)min(step, last) * 64;
        const int sm = step & 3;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = wv + 8 * k;              // piece i of the step: pixels 4i .. 4i+3 (one row: W % 4 == 0)
            const int rr = (4 * i) / W, x = (4 * i) % W;
            glds16(xsrc + (pix0 + 4 * i) * ldx, lds0 + XRING + (sm * TR + rr) * ROWB + (x + 1) * 256);
        }
    };
    auto stage_y = [&](int step) {
        const size_t pix0 = (size_t)min(step, last) * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = wv + 8 * k;              // piece i: pixels 2i, 2i+1
            glds16(ysrc + (pix0 + 2 * i) * a.ldq, lds0 + YRING + (step & 1) * YSTEP + i * 1024);
        }
    };
    // lane parts of the fragment addresses: pixel 2 kp + (l >> 5), channel l & 31 of the wave's 32
    const uint32_t la = lds0 + (l >> 5) * 256 + (wi * 32 + (l & 31)) * 4;
    const uint32_t lb = lds0 + YRING + (l >> 5) * 512 + (wj * 32 + (l & 31)) * 4;

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

    __syncthreads();
    if (sb > 0) {                                  // the row above the slice's first step: the last row of step sb - 1
        const int step = sb - 1;
        const size_t pix0 = (size_t)step * 64;
        constexpr int PPR = W / 4;                 // pieces per row
        if (wv < PPR) {
            const int i = 16 - PPR + wv;
            glds16(xsrc + (pix0 + 4 * i) * ldx, lds0 + XRING + ((step & 3) * TR + TR - 1) * ROWB + ((4 * i) % W + 1) * 256);
        }
        if constexpr (PPR > 8) if (wv + 8 < PPR) {
            const int i = 16 - PPR + wv + 8;
            glds16(xsrc + (pix0 + 4 * i) * ldx, lds0 + XRING + ((step & 3) * TR + TR - 1) * ROWB + ((4 * i) % W + 1) * 256);
        }
    }
    stage_x(sb); stage_x(sb + 1); stage_y(sb);

    for (int s = sb; s < se; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // X of steps <= s + 1 and dY of step s: all requested a step ago
        __builtin_amdgcn_s_barrier();                         // ... for every wave, and every wave is done reading step s - 1
        asm volatile("" ::: "memory");
        stage_x(s + 2); stage_y(s + 1);
        const int sm = s & 3, y0 = (s * TR) % a.H;
        uint32_t RB[TR + 2];                                  // row bases of relative rows -1 .. TR (ring slot, or the zero row)
        RB[0] = la + (y0 == 0 ? 0 : XRING + ((sm * TR + NR - 1) % NR) * ROWB);
#pragma unroll
        for (int rr = 0; rr < TR; ++rr) RB[rr + 1] = la + XRING + (sm * TR + rr) * ROWB;
        RB[TR + 1] = la + (y0 + TR == a.H ? 0 : XRING + (((sm + 1) & 3) * TR) * ROWB);
        const uint32_t yb = lb + (s & 1) * YSTEP;
        // 32 k-pairs of two pixels; a pair's ten operands are read one pair ahead of its nine MFMAs
        float Bv[2], Av[2][9];
        auto load_pair = [&](auto kc) {
            constexpr int kp = decltype(kc)::value, r = (2 * kp) / W, x = (2 * kp) % W;
            Bv[kp & 1] = lds_f32(yb + 2 * kp * 512);
            static_for<0, 9>([&](auto tc) {
                constexpr int tp = decltype(tc)::value, ky = tp / 3, kx = tp % 3;
                Av[kp & 1][tp] = lds_f32(RB[r + ky] + (x + kx) * 256);
            });
        };
        load_pair(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 32>([&](auto kc) {
            constexpr int kp = decltype(kc)::value;
            if constexpr (kp + 1 < 32) load_pair(std::integral_constant<int, kp + 1>{});
            static_for<0, 9>([&](auto tc) {
                constexpr int tp = decltype(tc)::value;
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(Av[kp & 1][tp], Bv[kp & 1], acc[tp], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // nothing may still be writing LDS when the workgroup retires

    if (a.splits == 1) {
        const int ci = ci0 + wi * 32 + 4 * (l >> 5), co = co0 + wj * 32 + (l & 31);
        if (co < a.Cj) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                float* o = a.dW + ((size_t)tp * a.Ci + ci) * a.Cj + co;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * a.Cj] += acc[tp][r];
            }
        }
        return;
    }
    float* out = a.ws + (size_t)(split * ntiles + tile) * (36 * 2048) + t * 4;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            *reinterpret_cast<f32x4*>(out + (tp * 4 + rq) * 2048) =
                f32x4{acc[tp][4 * rq], acc[tp][4 * rq + 1], acc[tp][4 * rq + 2], acc[tp][4 * rq + 3]};
}

__global__ __launch_bounds__(512, 1) void wgrad_tr32_kernel(const TrBatch b) {
    MI_PRIO_UP();
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    int p = 0;
#pragma unroll
    for (int q = 1; q < MAXP; ++q)
        if (q < b.n && (int)blockIdx.x >= b.p[q].wg0) p = q;
    const TrArgs& a = b.p[p];
    const int wg = blockIdx.x - a.wg0;
    switch (a.W) {
        case 8:  wgrad_tr32_body<8>(a, wg, lds_raw); break;
        case 16: wgrad_tr32_body<16>(a, wg, lds_raw); break;
        case 32: wgrad_tr32_body<32>(a, wg, lds_raw); break;
        default: wgrad_tr32_body<64>(a, wg, lds_raw); break;
    }
}

// dW[tap][ci][co] += sum over k-slices of the partial tiles (fixed order).  grid = (2G position groups, 36 slots, tiles of all
// problems that have k-slices); a workgroup = G slice groups x (256 / G) float4 positions of one slot; each thread keeps 8
// independent 16-byte loads in flight.
template <int G>
__global__ __launch_bounds__(256) void wgrad_tr_reduce_kernel(const TrBatch b) {
    MI_PRIO_UP();
    constexpr int IB = 256 / G;
    __shared__ f32x4 red[G][IB];
    int pi = 0;
#pragma unroll
    for (int q = 1; q < MAXP; ++q)
        if (q < b.n && b.p[q].splits > 1 && (int)blockIdx.z >= b.p[q].tile0) pi = q;
    const TrArgs& a = b.p[pi];
    const int ntiles = a.gx * a.gy, splits = a.splits;
    const int it = threadIdx.x % IB, grp = threadIdx.x / IB;
    const int tt = blockIdx.x * IB + it;                       // thread of the producing workgroup
    const int slot = blockIdx.y, tile = blockIdx.z - a.tile0;
    const float* p = a.ws + (size_t)tile * (36 * 2048) + (size_t)slot * 2048 + tt * 4;
    const size_t stride = (size_t)ntiles * (36 * 2048);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int sp = grp;
    for (; sp + 7 * G < splits; sp += 8 * G) {
        f32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(p + (size_t)(sp + q * G) * stride);
        s0 += v[0] + v[4]; s1 += v[1] + v[5]; s2 += v[2] + v[6]; s3 += v[3] + v[7];
    }
    for (; sp < splits; sp += G) s0 += *reinterpret_cast<const f32x4*>(p + (size_t)sp * stride);
    f32x4 s = (s0 + s1) + (s2 + s3);
    red[grp][it] = s;
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int g = 1; g < G; ++g) s += red[g][it];
    const int wv = tt >> 6, l = tt & 63;
    const int rq = slot & 3, tap = slot >> 2;
    const int ci = (tile % a.gx) * 64 + (wv >> 2) * 32 + 8 * rq + 4 * (l >> 5);
    const int co = (tile / a.gx) * 128 + (wv & 3) * 32 + (l & 31);
    if (co >= a.Cj) return;
    float* out = a.dW + ((size_t)tap * a.Ci + ci) * a.Cj + co;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (ci + e < a.Ci) out[(size_t)e * a.Cj] += s[e];
}

bool tr_ok(const MiWgradDesc* d) {
    if (d->KH != 3 || d->KW != 3 || d->pad != 1 || d->stride != 1 || !d->gather_i || (d->mode != 1 && d->mode != 0)) return false;
    if (d->GH != d->DH || d->GW != d->DW) return false;
    const int W = d->DW, H = d->DH;
    if (W != 8 && W != 16 && W != 32 && W != 64) return false;
    if (H % (64 / W)) return false;
    if (((long)d->N * H * W) % 64) return false;
    if (d->Ci % 64 || d->I1 % 64 || d->Cj % 32 || d->Cj < 32) return false;
    const int lda = d->mode == 1 ? 8 : 4;          // 16-byte pieces: bf16 (mode 1) or fp32 (mode 0: exact-fp32 kernel) operands
    if (d->ldp % lda || d->ldq % lda || (d->I1 != d->Ci && d->ldp2 % lda)) return false;
    return true;
}

int g_wtr_blocks = 0;       // mi_debug_wgrad_tr_blocks: workgroups wanted per launch (0 = one per CU)

long tr_target() {
    static const long env_target = mi_knob("MI_WTR_BLOCKS", 256);
    return g_wtr_blocks > 0 ? g_wtr_blocks : env_target;
}

// k-slices of one problem that is given `wgs` workgroups
void tr_plan(const MiWgradDesc* d, TrArgs& a, long wgs) {
    a.W = d->DW; a.H = d->DH; a.Ci = d->Ci; a.Cj = d->Cj; a.I1 = d->I1;
    a.gx = d->Ci / 64; a.gy = (d->Cj + 127) / 128;
    a.total = (int)((long)d->N * d->DH * d->DW / 64);
    const int ntiles = a.gx * a.gy;
    long splits = wgs / ntiles;
    if (splits < 1) splits = 1;
    if (splits > a.total) splits = a.total;
    a.sps = (int)((a.total + splits - 1) / splits);
    a.splits = (a.total + a.sps - 1) / a.sps;
}

// workgroups per problem in proportion to the MFMA work (N*H*W*Ci*Cj rounded up to whole tiles), every problem at least its tiles
void tr_shares(int n, const MiWgradDesc* d, long* wgs) {
    double fl[MAXP]; long tiles[MAXP];
    for (int i = 0; i < n; ++i) {
        fl[i] = (double)d[i].N * d[i].DH * d[i].DW * d[i].Ci * ((d[i].Cj + 127) / 128 * 128);
        tiles[i] = (long)(d[i].Ci / 64) * ((d[i].Cj + 127) / 128);
    }
    static const int greedy = (int)mi_knob("MI_WTR_BALANCE", 1);
    const long target = tr_target();
    if (greedy) { balance_shares(n, fl, tiles, target, wgs); return; }
    double tot = 0;
    for (int i = 0; i < n; ++i) tot += fl[i];
    for (int i = 0; i < n; ++i) {
        long w = (long)(target * fl[i] / tot + 0.5);
        w = w / tiles[i] * tiles[i];                            // whole k-slices
        wgs[i] = w < tiles[i] ? tiles[i] : w;
    }
}

size_t tr_lds(int W) { return (size_t)((W / 8 + 2) * 1024) * (1 + 4 * (64 / W)) + 3 * 64 * 256; }
size_t tr32_lds(int W) { return (size_t)((W + 2) * 256) * (1 + 4 * (64 / W)) + 2 * 64 * 512; }

size_t tr_ws_floats(const TrArgs& a) { return a.splits > 1 ? (size_t)a.splits * a.gx * a.gy * 36 * 2048 : 0; }

}  // namespace

extern "C" int mi_conv3x3_wgrad_tr_supported(const MiWgradDesc* d) { return (d && tr_ok(d)) ? 1 : 0; }

// Scratch bytes for a batch of n problems (n = 1: a single layer on every CU)
extern "C" size_t mi_conv3x3_wgrad_tr_batch_workspace(int n, const MiWgradDesc* descs) {
    if (!descs || n < 1 || n > MAXP) return 0;
    long wgs[MAXP];
    for (int i = 0; i < n; ++i) if (!tr_ok(&descs[i])) return 0;
    tr_shares(n, descs, wgs);
    size_t fl = 0;
    for (int i = 0; i < n; ++i) { TrArgs a; tr_plan(&descs[i], a, wgs[i]); fl += tr_ws_floats(a); }
    return fl * sizeof(float) + 256;
}
extern "C" size_t mi_conv3x3_wgrad_tr_workspace(const MiWgradDesc* d) { return mi_conv3x3_wgrad_tr_batch_workspace(1, d); }

extern "C" int mi_conv3x3_wgrad_tr_splits(const MiWgradDesc* d) {
    if (!d || !tr_ok(d)) return 0;
    TrArgs a; tr_plan(d, a, tr_target());
    return a.splits;
}

// tests: plan for `blocks` workgroups instead of one per CU (odd slice boundaries); 0 restores the default
extern "C" int mi_debug_wgrad_tr_blocks(int blocks) { g_wtr_blocks = blocks > 0 ? blocks : 0; return 0; }

// phase: 0 = contraction + reduce, 1 = contraction only, 2 = reduce only (per-kernel event timing, as mi_debug_wgrad3x3_phase)
static int g_wtr_phase = 0;
extern "C" int mi_debug_wgrad_tr_phase(int phase) {
    if (phase < 0 || phase > 2) return mi_set_error(-1, "mi_debug_wgrad_tr_phase: phase in 0..2");
    g_wtr_phase = phase;
    return 0;
}

extern "C" int mi_conv3x3_wgrad_tr_batch(int n, const MiWgradDesc* descs, const void* const* P, const void* const* P2,
                                         const void* const* Q, float* const* dW, void* workspace, size_t ws_bytes, void* stream) {
    MI_REQUIRE(n >= 1 && n <= MAXP && descs && P && Q && dW, "1..8 problems, non-null arrays");
    TrBatch b;
    b.n = n;
    long wgs[MAXP];
    for (int i = 0; i < n; ++i) {
        MI_REQUIRE(tr_ok(&descs[i]), "descriptor not supported by the LDS-DMA weight-gradient kernel (use mi_conv3x3_wgrad_io)");
        MI_REQUIRE(descs[i].mode == descs[0].mode, "one numeric mode per batch");
        MI_REQUIRE(P[i] && Q[i] && dW[i], "null operand");
        MI_REQUIRE(descs[i].I1 == descs[i].Ci || (P2 && P2[i]), "two-source split without P2");
        MI_REQUIRE((((uintptr_t)P[i] | (uintptr_t)Q[i] | (uintptr_t)((P2 && P2[i]) ? P2[i] : P[i])) & 15) == 0, "operands must be 16-byte aligned");
    }
    tr_shares(n, descs, wgs);
    size_t off = 0, lds = 0;
    int wg = 0, tile = 0, max_splits = 1;
    for (int i = 0; i < n; ++i) {
        TrArgs& a = b.p[i];
        tr_plan(&descs[i], a, wgs[i]);
        a.P = (const uint16_t*)P[i]; a.P2 = (const uint16_t*)((P2 && P2[i]) ? P2[i] : P[i]); a.Q = (const uint16_t*)Q[i];
        a.dW = dW[i];
        a.ldp = descs[i].ldp; a.ldp2 = (P2 && P2[i]) ? descs[i].ldp2 : descs[i].ldp; a.ldq = descs[i].ldq;
        a.ws = (float*)workspace + off;
        off += tr_ws_floats(a);
        static const int xcd_env = (int)mi_knob("MI_WTR_XCD", 2);
        a.xcd_map = 0;
        if (xcd_env && a.gx * a.gy > 1 && wg % 8 == 0) {
            if (a.splits % 8 == 0) a.xcd_map = 1;
            else if (xcd_env > 1 && a.splits < 8 && 8 % a.splits == 0 && (a.gx * a.gy) % (8 / a.splits) == 0) a.xcd_map = 2;
        }
        a.wg0 = wg; wg += a.gx * a.gy * a.splits;
        a.tile0 = tile; if (a.splits > 1) tile += a.gx * a.gy;
        if (a.splits > max_splits) max_splits = a.splits;
        const size_t l = descs[i].mode == 0 ? tr32_lds(a.W) : tr_lds(a.W);
        if (l > lds) lds = l;
    }
    static const int dbg = (int)mi_knob("MI_WTR_DEBUG", 0);
    if (dbg) {
        fprintf(stderr, "[wgrad_tr] %d layers, %d workgroups\n", n, wg);
        for (int i = 0; i < n; ++i) {
            const TrArgs& a = b.p[i];
            fprintf(stderr, "   W%-2d Ci%-4d Cj%-4d tiles %2d x splits %3d (steps/slice %3d): %.2f GFLOP per workgroup\n", a.W, a.Ci, a.Cj, a.gx * a.gy,
                    a.splits, a.sps, 2.0 * 64 * a.sps * 64 * 128 * 9 / 1e9);
        }
    }
    MI_REQUIRE(off == 0 || (workspace && ((uintptr_t)workspace & 15) == 0 && ws_bytes >= off * sizeof(float)),
               "workspace too small (mi_conv3x3_wgrad_tr_batch_workspace)");
    hipStream_t st = (hipStream_t)stream;
    static MiPerDevice once;
    once.run([] {
        (void)hipFuncSetAttribute((const void*)wgrad_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)wgrad_tr32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (g_wtr_phase != 2) {
        if (descs[0].mode == 0) hipLaunchKernelGGL(wgrad_tr32_kernel, dim3((unsigned)wg), dim3(512), lds, st, b);
        else hipLaunchKernelGGL(wgrad_tr_kernel, dim3((unsigned)wg), dim3(512), lds, st, b);
    }
    if (g_wtr_phase != 1 && tile > 0) {
        if (max_splits >= 64) hipLaunchKernelGGL(wgrad_tr_reduce_kernel<8>, dim3(16, 36, tile), dim3(256), 0, st, b);
        else hipLaunchKernelGGL(wgrad_tr_reduce_kernel<2>, dim3(4, 36, tile), dim3(256), 0, st, b);
    }
    MI_LAUNCH_CHECK();
    return 0;
}

extern "C" int mi_conv3x3_wgrad_tr(const MiWgradDesc* d, const void* P, const void* P2, const void* Q, float* dW,
                                   void* workspace, size_t ws_bytes, void* stream) {
    return mi_conv3x3_wgrad_tr_batch(1, d, &P, &P2, &Q, &dW, workspace, ws_bytes, stream);
}
