// 3x3 / stride 1 / pad 1 convolution (forward and data gradient) for bf16-stored activations -- wave-PRIVATE weight streams.
//     Y[p][co] (+)= bias[co] + res[p][co] + sum_{ky,kx,ci} X[p + (ky-1, kx-1)][ci] * W[ky][kx][co][ci]
// (Block's Conv2d(dim, dim_out, 3, padding=1), reference src/models/ddpm.py:116, and its input gradient.)
//
// What bounded conv3x3_halo.hip and round 2's conv_dma / conv_shift experiments (since removed): every tap (16 MFMAs per wave) ended in a
// workgroup barrier because the weight tile of a tap is shared by the waves of a workgroup, and a tile's prologue (its first
// activation rows arriving from HBM) and epilogue (its output stores) overlapped nothing: one workgroup per CU.  Here
//   * a workgroup = 4 waves = 128 output pixels (whole image rows) x 128 output channels; wave w owns ALL 128 pixels of the
//     channels [32w, 32w + 32): four 32x32 MFMA tiles, 64 accumulator registers;
//   * the weights a wave needs are needed by no other wave of the workgroup, so they never touch LDS: the bf16 copy is kept in
//     MFMA-FRAGMENT order ([tap][co / 32][ci / 16][lane][8], written by mi_pack_weights_bf16), a fragment is 1 KB contiguous in
//     memory and each lane loads ITS 16 bytes of it straight into the registers the MFMA reads (global_load_dwordx4 with a scalar
//     base per tap, issued from asm and counted by hand).  The nine fragments of a 16-channel step arrive during the step before;
//   * the activation tile (TH + 2 rows x 64 channels, XOR-swizzled through the DMA source address) is shared and double-buffered per
//     64-channel chunk (LDS-DMA): ONE barrier per chunk (144 MFMAs per wave) instead of nine;
//   * a 32-pixel MFMA block is output row i of every 4-row band of the tile, so "tap row ky of output row i" is "halo row i + ky":
//     ONE activation fragment per halo row serves the three tap rows, and only the centre tap column is read from LDS -- the left /
//     right columns are one-lane DPP shifts of it (first tried in round 2's conv_shift kernel), made once per halo row: per 16-channel step 6 LDS fragment
//     reads and 12 shifts feed 36 MFMAs;
//   * 66 KB of LDS and <= 256 registers per lane: two workgroups per CU -- one computes while the other waits for its first rows or
//     drains its stores.
// What it measures (B = 128, bf16 in / out, tools/bench_pw.py; round 2's per-shape pick beside it): 128 -> 128 @32x32 38-39 us (52-54),
// 256 -> 256 @16x16 34 us (41-43), 512 -> 512 @8x8 33.5-34.7 us (43-45) = 1.11-1.15 PFLOP/s.  Ablations of this structure (profiling
// build, tools/abl_pw.sh): the DPP shifts are free (<= 4 %), waiting for loads is free (a build that never waits: +-1 %), what
// costs is ISSUING vector-memory instructions next to the MFMA stream -- without the fragment loads -19 % on the deep layers,
// without the activation DMA -16 %, without the output stores -7..12 %; the same time with one and with two waves per SIMD.
// Round 5 ("Round 5"; tools/pw_timeline.py with -DMI_PW_TIMING): the plain bf16 / fp32-input conv (VAR 0 / 1) runs a PINNED,
// software-pipelined main loop -- every MFMA is followed by at most one side operation and a sched_barrier; a row unit issues its MFMAs from
// fragments that are complete in registers, reads the centre fragment of the unit after next and shifts the next unit's (two-register halves, no
// wait states); fragment requests one per gap from the start of a step; the next chunk's pieces through registers in steps 0-1, written in steps 2-3;
// the barrier needs no vmcnt -- behind a prologue whose first requests leave after ~1 100 cycles (was 3 600: divisions by multiplication, scalar piece
// table), and every workgroup walks the contraction in its own rotation of the chunk order (L2 channel hot spots).  Level 0 43-46 -> 34-36 us,
// 256 -> 256 @16x16 36-39 -> 32-35, 512 -> 512 @8x8 32-35 -> 30-32, 1024 -> 256 @8x8 46 -> 38 (same boxes, round 4's loop beside it).
#include "tr_common.h"

#ifndef MI_PW_PABL
#define MI_PW_PABL 0     // profiling builds: 1 no fragment requests in the pinned loop, 2 no piece requests, 4 no step-opening wait, 8 no shifts, 16 no barrier
#endif
#ifndef MI_PW_SPIECE
#define MI_PW_SPIECE 1    // scalar piece addressing in the plain conv (0: stage_x's per-lane 64-bit addresses)
#endif
#ifndef MI_PW_SKEW
#define MI_PW_SKEW 0      // 16-cycle units of delay per wave index behind every barrier of the pinned loop
#endif
#ifndef MI_PW_RSTAGE
#define MI_PW_RSTAGE 1    // pinned loop: the next chunk's pieces through registers (global_load + ds_write_b128) instead of LDS-DMA
#endif
#ifndef MI_PW_ROT
#define MI_PW_ROT 1       // per-workgroup rotation of the chunk order
#endif
#ifndef MI_PW_WSTR
#define MI_PW_WSTR 2
#endif
#ifndef MI_PW_PIPE
#define MI_PW_PIPE 1      // 0: round 3 / 4's compiler-scheduled main loop for the plain conv too (A/B builds)
#endif
#ifndef MI_PW_GNSABL
#define MI_PW_GNSABL 0    // profiling builds of the GroupNorm-sums epilogue: 1 no reduction / atomics, 2 no per-element sums (bf16 tile path)
#endif
#ifndef MI_PW_GNSW
#define MI_PW_GNSW 0
#endif
#ifndef MI_PW_FXABL
#define MI_PW_FXABL 0     // profiling builds of the pinned fused loop: 1 no transform, 2 unpacked fp32 instructions
#endif
#ifndef MI_PW_FPIPE
#define MI_PW_FPIPE 3     // round 6: the fused GroupNorm + Mish variants (VAR 2 / 3) on the pinned loop -- bit 0 bf16 input, bit 1 fp32 input (0: round 4's loop, A/B builds)
#endif
#ifdef MI_PW_TIMING
// profiling build only (-DMI_PW_TIMING, tools/pw_timeline.py): shader-clock stamps (s_memtime) at every row unit of conv_pw_kernel's main
// loop, kept in registers (v_writelane: slot p holds the time of point p - 1) and written once at the end: [workgroup][wave][5][64]
// words -- sets 0..3 = step ks, lane = chunk * (BH + 1) + unit; set 4: 0 last point, 1 loop end, 2 kernel end, 3 kernel entry, 4 loop
// entry, 5 HW_ID, 6 XCC_ID, 7 / 8 the 100 MHz wall clock at entry / end.
__device__ uint32_t g_pw_ts[1024 * 4 * 5 * 64];
__device__ __forceinline__ uint32_t pw_wlane(uint32_t val, uint32_t lane, uint32_t v) {
    uint32_t keep;
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                 : "+v"(v), "=&s"(keep) : "s"(__builtin_amdgcn_readfirstlane(val)), "s"(__builtin_amdgcn_readfirstlane(lane)));
    return v;
}
extern "C" int mi_debug_pw_ts(uint32_t* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pw_ts), sizeof(g_pw_ts)); }
#endif

namespace {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) uint32_t g_zero_page3[64];     // 256 zero bytes: the rows above / below an image

struct PwArgs {
    const uint16_t* x; const uint16_t* x2; const uint16_t* w; const float* bias; const float* res; void* y;
    int N, H, W, K, Nc, K1, ldx, ldx2, ldy, ldr, accumulate, flip;
    int TH, TI, XP, tiles_per_img, xmap;
    int qmap, gx, gy;    // 1-D launch of gx pixel tiles x gy channel tiles: XCD = (pixel group, channel group) of a (8 / qmap) x qmap split
    // round 5: the prologue's integer divisions as multiplications (q = umulhi(x, magic), magic = ceil(2^32 / d), 0 for d = 1; exact for
    // x * d < 2^32) and the zero page's address as an argument -- 3 600 cycles passed between a wave's entry and its first request
    uint32_t tpi_magic, cpq_magic, nch_magic, cpg_magic; int ppx, cpq, lnsub;
    const void* zero;
    float* gsum;         // VAR 1: [N][Nc / 16][2] sum / sum of squares of the stored values per sample and 16-channel slab (+=)
    const float* coef;   // VAR 2: [3][N][K] scale, shift, time bias of the GroupNorm + Mish applied to x while it is staged
    // VAR 3: the same coefficients resolved in the kernel from the sums the producing conv's epilogue left (gsum layout), the affine
    // parameters and the time bias rows temb [N][ldt] (may be null)
    const float* sums; const float* gamma; const float* beta; const float* temb;
    int ldt, cpg, hw, ng; float eps;
    double icnt;         // VAR 3: 1 / (MI_GSUM_SCALE * hw * cpg)
    // split-K (gridDim.z = ksplit > 1; launches that would leave CUs without a workgroup): slice s = ksplit - 1 - blockIdx.z contracts chunks
    // [s, s + 1) * nchunks / ksplit.  Slices 1 .. ksplit - 1 are dispatched FIRST, leave their fp32 accumulators in sk_part (register order, per wave)
    // and raise the wave's flag; slice 0 (dispatched last: whatever it waits for is resident or done) adds them and runs the epilogue.
    int ksplit; float* sk_part; int* sk_flag;
};

// PT = output pixels per workgroup: 128 (four 32-pixel MFMA blocks per wave) or, for the layers whose 128-pixel tiles would leave
// CUs without a workgroup, 64 (two blocks: a fragment feeds two MFMAs instead of four, but the grid is twice as large)
constexpr int pw_xp(int pt) { return pt == 256 ? 320 : pt == 128 ? 192 : 128; }      // tile pixels incl. the rows above and below: 10 x 32, 18 x 16 | 6 x 32, 10 x 16, 2 images x 10 x 8 | 4 x 32, 6 x 16, 10 x 8

__device__ __forceinline__ int pw_fastdiv(int x, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)x, magic) : x; }
inline uint32_t pw_magic(int d) { return d > 1 ? (uint32_t)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d) : 0u; }

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

// 16 bytes per lane from sbase + voff + IMM straight into registers; asynchronous: the caller counts it (s_waitcnt vmcnt) and touches
// dst again only behind that wait
template <int IMM> __device__ __forceinline__ void gload16s(u32x4& dst, uint64_t sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
__device__ __forceinline__ void landed16(u32x4& r) { asm volatile("" : "+v"(r)); }

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
    // (one statement: hipcc pads every asm statement that precedes an MFMA with an s_nop of its own.  The trailing s_nop 1: a VALU
    //  write -> MFMA operand read needs two wait states and hipcc does not see the VALU instructions in here)
    if constexpr (DIR == 0)
        asm("v_and_b32_dpp %0, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_and_b32_dpp %1, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_and_b32_dpp %2, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_and_b32_dpp %3, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1"
            : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(mask));
    else
        asm("v_and_b32_dpp %0, %4, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_and_b32_dpp %1, %5, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_and_b32_dpp %2, %6, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_and_b32_dpp %3, %7, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1"
            : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(mask));
    return __builtin_bit_cast(bf16x8, o);
}

// the same shift for the pinned loop, where the result is consumed one row unit (>= 4 MFMAs) later: no wait states inside, and HALF = 0 / 1
// makes registers 0-1 / 2-3 only, so that a gap carries two DPP instructions instead of four (+ s_nop)
template <int DIR, int HALF> __device__ __forceinline__ void pw_shift_h(u32x4& o, const bf16x8& c, uint32_t mask) {
    const u32x4 v = __builtin_bit_cast(u32x4, c);
    uint32_t a0, a1;
    if constexpr (DIR == 0)
        asm("v_and_b32_dpp %0, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_and_b32_dpp %1, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "=&v"(a0), "=&v"(a1) : "v"(v[2 * HALF]), "v"(v[2 * HALF + 1]), "v"(mask));
    else
        asm("v_and_b32_dpp %0, %2, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_and_b32_dpp %1, %3, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "=&v"(a0), "=&v"(a1) : "v"(v[2 * HALF]), "v"(v[2 * HALF + 1]), "v"(mask));
    o[2 * HALF] = a0; o[2 * HALF + 1] = a1;
}

constexpr int pw_lds(int pt, bool raw = false) { return (pt == 256 ? 128 : pt == 128 ? (raw ? 72 : 64) : (raw ? 48 : 32)) * 1024 + 2048; }   // two activation buffers (24 / 16 KB each) under the epilogue's fp32 tile (+ the GroupNorm sums' 512 bytes); raw: + the fp32 staging area of the fp32-input variants (24 / 16 KB behind the tiles)

// VAR 0: the plain conv.  VAR 1: the epilogue also accumulates the GroupNorm sums of the NEXT layer (a.gsum).  VAR 2 / 3: the named
// fused kernel (2: coefficients given, 3: resolved here from the producer's sums) -- x is the RAW output of the previous conv and mish(x * scale[n][c] + shift[n][c]) + tb[n][c] (a.coef; GroupNorm-apply
// + Mish + time bias, reference src/models/ddpm.py:112-120,139-141) is applied ONCE per staged element: every wave transforms the
// pieces it requested itself as one block per chunk (bf16 input: in place in LDS; fp32 input: from the raw staging area, half a chunk
// at a time) before the chunk barrier publishes them; rows outside the image stay zero.  One image per tile (TI == 1).
// ABL (profiling builds only, -DMI_PW_ABL_BUILD): 1 no fragment DMA in the main loop, 2 no activation DMA in the main loop, 4 no stores,
// 16 no DPP shifts (every tap column multiplies the centre fragments), 32 loads issued but never waited for in the main loop
// IN32: x / x2 are fp32 tensors (the residual stream: the sampler's block1 convs, fp32 block storage).  Their raw pieces arrive by
// LDS-DMA in a staging area behind the two bf16 tiles, half a chunk at a time; each lane reads back its own 8 channels, rounds them to
// bf16 once and writes the 16 bytes of its slot of the tile -- the lane requests the channel chunk that belongs in ITS slot, as the
// DMA's source addresses do for bf16 input.  (Round 3 staged two pieces per step through registers with counted waits; only the
// prologue still does.)
// IN32 + VAR 2 / 3 (round 4; the named fused kernel for fp32-stored activations): the transform is applied to the fp32 values while
// they sit in those registers, between their load and the one rounding -- no LDS read-modify-write, no unpack.
// F32: the exact-fp32 mode of the same kernel (Unet.compute_mode = "fp32", the reference's default precision and the mode that carries
// the 1e-4 parity bar): x / x2 / y fp32, weights fp32 in fragment order [tap][co / 32][ci / 8][lane][4] (mi_pack_weights_f32frag),
// v_mfma_f32_32x32x2_f32 -- bit-equal to an fp32 fmaf chain.  Everything keeps its byte geometry: a 16-byte piece of a pixel row is 4
// fp32 channels instead of 8 bf16 ones, so a chunk is 32 channels, a step 8, a fragment is still 1 KB and feeds four MFMAs per block
// (k = 2 each: lane half h supplies channel 4h + j of the octet in step j).  Per 16 bytes loaded the fp32 MFMA runs 8x longer than
// the bf16 one, so this variant is bound by the matrix pipe, not by the issue of loads.
template <bool OUT16, int VAR = 0, int ABL = 0, int PT = 128, bool IN32 = false, bool F32 = false>
__global__ __launch_bounds__(256, PT == 256 ? 1 : 2) void conv_pw_kernel(const PwArgs a) {
    MI_PRIO_UP();
#ifdef MI_PW_TIMING
    constexpr bool TIMED = VAR == 0 && ABL == 0 && !IN32 && !F32;
    uint32_t tv[5] = {0u, 0u, 0u, 0u, 0u};
    uint64_t tprev = 0;
    if constexpr (TIMED) {
        tprev = __builtin_amdgcn_s_memtime();
        tv[4] = pw_wlane((uint32_t)wall_clock64(), 7, tv[4]);
        tv[4] = pw_wlane((uint32_t)tprev, 3, tv[4]);
    }
#define MI_PW_STAMP(SET, LANE) do { if constexpr (TIMED) { tv[SET] = pw_wlane((uint32_t)tprev, (LANE), tv[SET]); tprev = __builtin_amdgcn_s_memtime(); } } while (0)
#define MI_PW_NOW(LANE) do { if constexpr (TIMED) { tv[4] = pw_wlane((uint32_t)__builtin_amdgcn_s_memtime(), (LANE), tv[4]); } } while (0)
#else
#define MI_PW_STAMP(SET, LANE) do {} while (0)
#define MI_PW_NOW(LANE) do {} while (0)
#endif
    static_assert(PT != 256 || (VAR < 2 && !IN32 && !F32), "256-pixel tiles: the plain bf16 conv (with or without the GroupNorm sums)");
    constexpr bool FUSE = VAR >= 2, GNS = VAR == 1;
    // round 6: the fused variants on round 5's pinned loop.  The coefficients of ALL K channels of the tile's image (scale log2(e), shift log2(e),
    // time bias: 12 bytes per channel, built once in the prologue) wait in LDS behind the two activation buffers, a piece of the next chunk is
    // transformed in the registers it was loaded into (no LDS read-modify-write) and its work -- coefficient reads, two Mish pairs per half piece,
    // the ds_write_b128 -- rides in the MFMA gaps of steps 2 and 3 like the plain conv's register staging; the counted waits are the plain conv's.
    constexpr bool FPIPE = FUSE && !F32 && ABL == 0 && (MI_PW_PIPE != 0) && (((MI_PW_FPIPE) >> (IN32 ? 1 : 0)) & 1) != 0;
    static_assert(!IN32 || ABL == 0, "fp32 input: no ablation builds");
    static_assert(!F32 || (VAR == 0 && ABL == 0 && !IN32 && !OUT16), "exact-fp32 mode: the plain conv, fp32 in and out");
    constexpr int PCK = F32 ? 32 : 64;                       // channels per chunk (128 bytes of a pixel row)
    constexpr int ESZ = F32 ? 4 : 2, EPP = 16 / ESZ;         // element size, elements per 16-byte piece
    constexpr int BH = PT / 32;                              // rows per band = 32-pixel blocks per wave
    constexpr int PXBUF = pw_xp(PT) * 128;                   // one chunk of the activation tile
    constexpr int PXPW = pw_xp(PT) / 8 / 4;                  // activation DMA instructions per wave and chunk
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    int bx = blockIdx.x;
    if (a.xmap) {        // an image's row tiles share rows: keep them on one XCD (ids xcd + 8*slot -> image xcd + 8*m)
        const int xcd = bx & 7, slot = bx >> 3, q = pw_fastdiv(slot, a.tpi_magic);
        bx = (xcd + 8 * q) * a.tiles_per_img + (slot - q * a.tiles_per_img);
    }
    int by = blockIdx.y;
    if (a.qmap) {        // XCD = (pixel group, channel group) of a 4 x 2 split: an XCD's L2 holds half of the weight tiles
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3, q = pw_fastdiv(slot, a.cpq_magic);
        bx = (xcd >> 1) * a.ppx + q; by = (xcd & 1) * a.cpq + (slot - q * a.cpq);
    }
    const int m0 = bx * PT, n0 = by * 128;
    const int TH2 = a.TH + 2;
    const int lw = 31 - __builtin_clz(a.W);                   // W (and TH) are powers of two
    const int slice = a.ksplit > 1 ? a.ksplit - 1 - (int)blockIdx.z : 0;
    const int nchunks = a.K / PCK / a.ksplit, kbase = slice * nchunks;      // this workgroup's share of the contraction
    const int NB = a.Nc >> 5, KQ = a.K / (2 * EPP);              // fragments per (tap, 32-channel block): one per step
    const bool live = n0 + 32 * wv < a.Nc;                    // a ragged last channel tile: the wave computes a copy of the last block
    const int nb = min((n0 >> 5) + wv, NB - 1);
    // round 5: every workgroup walks the contraction in its own rotation of the chunk order (iteration ch works on chunk (ch + rot) mod
    // nchunks).  The workgroups of a channel tile otherwise request the SAME weight lines in the same microseconds -- 16 CUs of an XCD on a
    // handful of L2 channels -- and the per-chunk time stayed at ~6 000 cycles whatever was done about issue order, waits or staging.
    // (The sum of a pixel's products is formed in a different order per tile: deterministic, tile by tile.)
    const int rot = MI_PW_ROT ? bx - pw_fastdiv(bx, a.nch_magic) * nchunks : 0;
    auto cof = [&](int ch) -> int { const int c = min(ch, nchunks - 1) + rot; return kbase + (c >= nchunks ? c - nchunks : c); };

    // ---- weight stream of this wave: fragment (tap, nb, kq) = 1 KB at ((tap * NB + nb) * KQ + kq) * 1024 bytes.  The fragments are
    //      wave-private, so they never touch LDS: each lane loads ITS 16 bytes of a fragment straight into the registers the MFMA
    //      reads (global_load_dwordx4 with a scalar base per tap, issued from asm and counted by hand like the DMA).  The nine
    //      fragments of a 16-channel step are loaded during the step before (two register sets).
    const uint8_t* wsrc = reinterpret_cast<const uint8_t*>(a.w) + (size_t)nb * KQ * 1024;
    const uint32_t tap_bytes = (uint32_t)a.Nc * a.K * ESZ;
    const uint32_t wl16 = l * 16;
    uint64_t wtap[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const uint64_t q = (uint64_t)(uintptr_t)(wsrc + (size_t)(a.flip ? 8 - tp : tp) * tap_bytes);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)q), hi = __builtin_amdgcn_readfirstlane((uint32_t)(q >> 32));
        wtap[tp] = ((uint64_t)hi << 32) | lo;
    }
    u32x4 WB[2][9];
    // part `part` of NPART of the nine taps of step (ch, ks) (ks == 4: step 0 of chunk ch + 1; past the end: a re-fetch) -> set ks & 1.
    // Every part is requested BEFORE the chunk boundary's wait (the last row unit requests none): no register-destination load is
    // ever in flight across the loop's back edge or its exit, where hipcc may copy or reuse the destination registers.
    constexpr int NPART = BH >= 3 ? 3 : 2;
    auto load_w3 = [&](int ch, auto ksc, auto partc) {
        constexpr int ks0 = decltype(ksc)::value, over = ks0 >= 4 ? 1 : 0, ks = ks0 - 4 * over, part = decltype(partc)::value;
        constexpr int t0 = (9 * part + NPART - 1) / NPART, t1 = (9 * (part + 1) + NPART - 1) / NPART;
        const uint32_t voff = wl16 + (uint32_t)cof(ch + over) * 4096;
        static_for<t0, t1>([&](auto tc) {
            constexpr int tp = decltype(tc)::value;
            gload16s<ks * 1024>(WB[ks & 1][tp], wtap[tp], voff);
        });
    };

    // one tap of step (ch, ks) (round 5's pinned schedule: one request per MFMA gap)
    auto load_w1 = [&](int ch, auto ksc, auto tapc) {
        constexpr int ks0 = decltype(ksc)::value, over = ks0 >= 4 ? 1 : 0, ks = ks0 - 4 * over, tp = decltype(tapc)::value;
        const uint32_t voff = wl16 + (uint32_t)cof(ch + over) * 4096;
        gload16s<ks * 1024>(WB[ks & 1][tp], wtap[tp], voff);
    };

    // (round 5: the plain conv requests its first step's fragments before anything else is computed -- they depend on the tile's
    //  channel block only; the fused / fp32-input variants count their coefficient and row requests first and keep the old order)
    constexpr bool EARLYW = VAR < 2 && !IN32;
    if constexpr (EARLYW) {
        MI_PW_NOW(10);
        static_for<0, NPART>([&](auto pc) { load_w3(0, std::integral_constant<int, 0>{}, pc); });
        __builtin_amdgcn_sched_barrier(0);
    }


    // ---- activation DMA pieces: piece p = wv + 4i covers tile pixels 8p .. 8p+7 (tile pixel hp = (image ti, row hy = y+1, column x));
    //      lane -> pixel lane >> 3, stored 16-byte position lane & 7 holds channel chunk (lane & 7) ^ swz(hp), swz = (hp >> 1) & 7
    // ((hp >> 1) & 7 = (4 wv + (l >> 4)) & 7 for every piece of a wave: a lane always holds the same channel chunk)
    // W = 8 (round 4): a 16-lane group of a fragment read covers pixels of FOUR tile rows (two bands, or two bands x two images) with
    // only four different x >> 1, and (hp >> 1) & 7 = 4 (row & 1) + (x >> 1) is the same for all four rows: two lanes per 16-byte
    // slot, SQ_LDS_BANK_CONFLICT = 44 % of the LDS cycles of the 8x8-level layers.  There swz = ((hp >> 1) & 3) | ((row >> 1) & 1) << 2:
    // the rows a group reads are {r, r + 4, r + 10, r + 14} (128-pixel tiles, two images) or {r, r + 2, r + 4, r + 6} (64-pixel tiles),
    // and bit 1 of the row differs inside each pair that shares its columns -- 16 lanes, 16 slots.  For a wave's pieces
    // (row >> 1) & 1 = (hp >> 4) & 1 = (wv >> 1) & 1: still one channel chunk per lane.
    int xpix[PXPW];
    const bool w8 = a.W == 8;
    const int xcol = ((l & 7) ^ (w8 ? (((l >> 4) & 3) | (((wv >> 1) & 1) << 2)) : ((4 * wv + (l >> 4)) & 7))) * EPP;
    int img0;
    {
        int y0;
        if (a.TI > 1) { img0 = bx * a.TI; y0 = 0; }
        else { img0 = pw_fastdiv(bx, a.tpi_magic); y0 = (bx - img0 * a.tiles_per_img) * a.TH; }
#pragma unroll
        for (int i = 0; i < PXPW; ++i) {                     // (branch-free: bit operations on the conditions)
            const int hp = 8 * (wv + 4 * i) + (l >> 3);
            const int row = hp >> lw, x = hp & (a.W - 1);
            const int ti = row >= TH2 ? 1 : 0, hy = row - ti * TH2;                  // at most two images per tile
            const int iy = y0 + hy - 1, img = img0 + ti;
            const bool ok = (hp < a.XP) & (iy >= 0) & (iy < a.H) & (img < a.N);
            xpix[i] = ok ? (img * a.H + iy) * a.W + x : -1;
        }
    }
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(a.zero);
    auto stage_x = [&](int ch, int i) {                      // piece i of this wave of chunk ch's rows -> buffer ch & 1
        const int cc0 = cof(ch) * PCK;
        const bool second = cc0 >= a.K1;
        const uint16_t* src = second ? a.x2 : a.x;
        const int ld = second ? a.ldx2 : a.ldx, cc = second ? cc0 - a.K1 : cc0;
        // (both addresses are formed, then one is selected: a select with a multiply in one arm becomes a branch around it)
        int xp = xpix[i];
        asm volatile("" : "+v"(xp));                         // (opaque: the 64-bit row offsets are recomputed per chunk, not hoisted into six
        size_t off = (size_t)max(xp, 0) * ld + cc + xcol;    //  register pairs that end up in scratch)
        asm volatile("" : "+v"(off));
        const uint8_t* p = xp >= 0 ? reinterpret_cast<const uint8_t*>(src) + off * ESZ : reinterpret_cast<const uint8_t*>(zero) + (l & 7) * 16;
        glds16(p, lds0 + (ch & 1) * PXBUF + (wv + 4 * i) * 1024);
    };
    // ---- round 5, the plain conv: a piece = 8 pixels of ONE tile row (W >= 8), so everything about it but the lane's place in it is
    //      wave-uniform: source = scalar base (tensor + row + chunk) + a per-lane offset that is the same for every piece and chunk
    //      ((l >> 3) pixels + the lane's channel slot), i.e. no vector arithmetic per request at all (stage_x: a 64-bit multiply-add, a
    //      compare and two selects per piece).  A piece outside the image is not fetched from a zero page: its LDS rows are zeroed once
    //      (both buffers) and its request -- still issued, the counted waits assume it -- reads valid memory into a dump area.
    constexpr bool SPIECE = (((MI_PW_SPIECE != 0 || IN32) && VAR < 2) || FPIPE) && !F32 && ABL == 0;     // (fp32 input, fused: the pinned loop's register staging needs it)
    constexpr int XSZ = IN32 ? 4 : ESZ;                     // bytes per element of x / x2 in memory
    int prow[SPIECE ? PXPW : 1];                             // first pixel of the piece in the tensor, 0 when outside (scalar)
    uint32_t pvalid = 0;                                     // bit i: piece i lies in the image
    uint32_t lane_off1 = 0, lane_off2 = 0;
    constexpr uint32_t DUMP = 2 * PXBUF;                     // 1 KB behind the two tiles (the epilogue's tile starts over)
    if constexpr (SPIECE) {
        const int y0s = a.TI > 1 ? 0 : (bx - img0 * a.tiles_per_img) * a.TH;
#pragma unroll
        for (int i = 0; i < PXPW; ++i) {
            const int hp0 = 8 * (wv + 4 * i);
            const int row = hp0 >> lw, x0 = hp0 & (a.W - 1);
            const int ti = row >= TH2 ? 1 : 0, hy = row - ti * TH2;
            const int iy = y0s + hy - 1, img = img0 + ti;
            const bool ok = (hp0 < a.XP) & (iy >= 0) & (iy < a.H) & (img < a.N);
            prow[i] = ok ? (img * a.H + iy) * a.W + x0 : 0;
            pvalid |= ok ? (1u << i) : 0u;
        }
        lane_off1 = (uint32_t)(((l >> 3) * a.ldx + xcol) * XSZ); lane_off2 = (uint32_t)(((l >> 3) * a.ldx2 + xcol) * XSZ);
    }
    auto stage_s = [&](int ch, auto ic) {
        constexpr int i = decltype(ic)::value;
        const int cc0 = cof(ch) * PCK;
        const bool second = cc0 >= a.K1;
        const uint8_t* src = reinterpret_cast<const uint8_t*>(second ? a.x2 : a.x);
        const int ld = second ? a.ldx2 : a.ldx, cc = second ? cc0 - a.K1 : cc0;
        const uint8_t* sb = src + ((size_t)prow[i] * ld + cc) * ESZ;
        const uint32_t dst = ((pvalid >> i) & 1u) ? lds0 + (ch & 1) * PXBUF + (wv + 4 * i) * 1024 : lds0 + DUMP;
        glds16s(sb, second ? lane_off2 : lane_off1, dst);
    };
    if constexpr (EARLYW) {                                  // the first chunk's rows: on their way before the rest of the set-up
        if constexpr (SPIECE) {
            static_for<0, PXPW>([&](auto ic) { stage_s(0, ic); });
            typedef __attribute__((address_space(3))) u32x4 lds_u32x4z;
#pragma unroll
            for (int i = 0; i < PXPW; ++i)
                if (!((pvalid >> i) & 1u)) {                 // zero padding, once: nothing is ever written there again
                    *(lds_u32x4z*)(uintptr_t)(lds0 + (wv + 4 * i) * 1024 + l * 16) = u32x4{0u, 0u, 0u, 0u};
                    *(lds_u32x4z*)(uintptr_t)(lds0 + PXBUF + (wv + 4 * i) * 1024 + l * 16) = u32x4{0u, 0u, 0u, 0u};
                }
        } else {
#pragma unroll
            for (int i = 0; i < PXPW; ++i) stage_x(0, i);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // fp32 input: piece i of chunk ch -> the two registers of `dst` (8 channels of this lane's pixel), asynchronous
    auto load_x32 = [&](int ch, int i, u32x4* dst) {
        const int cc0 = cof(ch) * PCK;
        const bool second = cc0 >= a.K1;
        const float* src = reinterpret_cast<const float*>(second ? a.x2 : a.x);
        const int ld = second ? a.ldx2 : a.ldx, cc = second ? cc0 - a.K1 : cc0;
        int xp = xpix[i];
        asm volatile("" : "+v"(xp));
        size_t off = (size_t)max(xp, 0) * ld + cc + xcol;
        asm volatile("" : "+v"(off));
        const float* pf = xp >= 0 ? src + off : reinterpret_cast<const float*>(a.zero) + (l & 7) * 8;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                     : "=&v"(dst[0]), "=&v"(dst[1]) : "v"(pf) : "memory");
    };
    // ... rounded and written to the tile (buffer buf) once the wait that covers its loads has passed
    auto store_x32 = [&](int buf, int i, u32x4* r) {
        typedef __attribute__((address_space(3))) u32x4 lds_u32x4_;
        landed16(r[0]); landed16(r[1]);
        const u32x4 o = {pack_bf16(__uint_as_float(r[0].x), __uint_as_float(r[0].y)), pack_bf16(__uint_as_float(r[0].z), __uint_as_float(r[0].w)),
                         pack_bf16(__uint_as_float(r[1].x), __uint_as_float(r[1].y)), pack_bf16(__uint_as_float(r[1].z), __uint_as_float(r[1].w))};
        *(lds_u32x4_*)(uintptr_t)(lds0 + buf * PXBUF + (wv + 4 * i) * 1024 + l * 16) = o;
    };

    // ---- fused variants: the 3 x 8 coefficients of this lane's channel chunk of chunk ch.  Plain loads would make hipcc drain the DMA
    //      queue (vmcnt(0)) at their first use, so they are issued from one asm statement and counted by hand; their
    //      destination registers are touched again only behind coef_landed(), which sits after the counted wait that covers them.
    f32x4 cq[6];                                             // scale[8], shift[8], time bias[8] of the chunk being transformed
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // VAR 3 (round 4): the statistics of ALL groups of the image once per wave -- lane g < K / cpg loads the slabs of group g in the
    // prologue and computes (mean, rstd) there (mi_gn_coef_from_sums' arithmetic: integer slab sums, double, var = E[x^2] - mean^2);
    // a chunk's coefficients then need two ds_bpermute per lane instead of four more loads and the double arithmetic per lane and chunk
    // (the first form cost the sums variant 3-6 us per level-0 launch over the coefficient-tensor one).
    u32x4 sq[4];                                             // prologue only: (sum, sum of squares) of lane's group, 64-bit fixed point each
    float gmean = 0.f, grstd = 0.f;                          // lane g: group g of this image
    const int nslab = FUSE ? a.cpg >> 4 : 1;
    auto load_stats = [&]() {                                // counted like the coefficients (four loads, the oldest of the prologue)
        const int ng = a.ng, gq = min(l, ng - 1);
        const float* sp = a.sums + ((size_t)img0 * (a.K >> 4) + (size_t)gq * nslab) * 4;           // 16 bytes per slab
        const float* s1 = sp + (nslab > 1 ? 4 : 0); const float* s2 = sp + (nslab > 2 ? 8 : 0); const float* s3 = sp + (nslab > 2 ? 12 : 0);
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                     "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off"
                     : "=&v"(sq[0]), "=&v"(sq[1]), "=&v"(sq[2]), "=&v"(sq[3]) : "v"(sp), "v"(s1), "v"(s2), "v"(s3) : "memory");
    };
    auto stats_landed = [&]() {
        asm volatile("" : "+v"(sq[0]), "+v"(sq[1]), "+v"(sq[2]), "+v"(sq[3]) :: "memory");
        long long si = 0, qi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nslab) {
                si += (long long)(((unsigned long long)sq[k].y << 32) | sq[k].x);
                qi += (long long)(((unsigned long long)sq[k].w << 32) | sq[k].z);
            }
        const double mean = (double)si * a.icnt;
        double var = (double)qi * a.icnt - mean * mean;
        if (var < 0.0) var = 0.0;
        grstd = 1.0f / sqrtf((float)var + a.eps); gmean = (float)mean;
    };
    auto load_coef = [&](int ch) {
        const int c = cof(ch) * PCK + xcol;
        const float* p0; const float* p1; const float* p2;
        if constexpr (VAR == 2) {
            p0 = a.coef + (size_t)img0 * a.K + c;
            const size_t pl = (size_t)a.N * a.K;
            p1 = p0 + pl; p2 = p0 + 2 * pl;
        } else {             // gamma, beta, the time-bias row (the zero page when there is none)
            p0 = a.gamma + c; p1 = a.beta + c;
            p2 = a.temb ? a.temb + (size_t)img0 * a.ldt + c : reinterpret_cast<const float*>(a.zero);
        }
        asm volatile("global_load_dwordx4 %0, %6, off\n\tglobal_load_dwordx4 %1, %6, off offset:16\n\t"
                     "global_load_dwordx4 %2, %7, off\n\tglobal_load_dwordx4 %3, %7, off offset:16\n\t"
                     "global_load_dwordx4 %4, %8, off\n\tglobal_load_dwordx4 %5, %8, off offset:16"
                     : "=&v"(cq[0]), "=&v"(cq[1]), "=&v"(cq[2]), "=&v"(cq[3]), "=&v"(cq[4]), "=&v"(cq[5])
                     : "v"(p0), "v"(p1), "v"(p2) : "memory");
    };
    auto coef_landed = [&](int ch) {
        asm volatile("" : "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]), "+v"(cq[3]), "+v"(cq[4]), "+v"(cq[5]) :: "memory");
        if constexpr (VAR == 3) {
            const int grp = pw_fastdiv(cof(ch) * PCK + xcol, a.cpg_magic);      // the lane's 8 channels lie in one group
            const float mf = __shfl(gmean, grp, 64), rstd = __shfl(grstd, grp, 64);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sc = cq[h][e] * rstd;
                    cq[2 + h][e] = cq[2 + h][e] - mf * sc;
                    cq[h][e] = sc;
                }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) cq[h] *= 1.44269504f;    // scale, shift -> scale log2(e), shift log2(e): see mish_tb2
    };
    // two elements (one packed register) of a piece: channels 2q, 2q + 1 of the lane's chunk.
    // mish(u) + tb with u = x * scale + shift.  Round 4: written for the PACKED fp32 pipe (v_pk_fma_f32 / v_pk_add_f32: two elements per
    // instruction) and without the clamp: scale and shift carry log2(e) (coef_landed), so t = u log2(e) feeds v_exp_f32 directly;
    // with e = 2^t, q = e (e + 2) + 2 and r = 1 / q,  tanh(softplus(u)) = 1 - 2 r,  so  mish(u) + tb = t * (ln 2 - 2 ln 2 * r) + tb.
    // e = +inf (u > 88) gives r = 0 and the result u + tb, e = 0 gives exactly tb: no clamp, no NaN.  Per element pair: 4 packed
    // instructions + 4 transcendentals (before: ~12 + 4) -- the transform is what the fused kernel pays over the plain conv.
    auto mish_tb2 = [&](f32x2 x, f32x2 sc, f32x2 sh, f32x2 tb) -> f32x2 {
        // (explicit packed instructions: left to itself instruction selection splits most of the vector arithmetic back into scalar
        //  v_fma_f32.  A 32-bit constant is used for both halves with op_sel_hi 0; {-2 ln 2, ln 2} is ONE scalar pair read twice --
        //  low half as the factor, high half as the addend -- since an instruction may read one scalar operand only)
        f32x2 t, e, q, r, sm, o;
        const f32x2 kln = {-1.38629436f, 0.693147182f};
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(x), "v"(sc), "v"(sh));
        e = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        // (s_nop: on gfx940+ a VALU instruction must not read a transcendental's result in the very next issue slot, and hipcc does not
        //  look for that hazard inside asm statements)
        asm("s_nop 0\n\tv_pk_add_f32 %0, %1, 2.0 op_sel_hi:[1,0]" : "=v"(q) : "v"(e));
        asm("v_pk_fma_f32 %0, %1, %2, 2.0 op_sel_hi:[1,1,0]" : "=v"(q) : "v"(e), "0"(q));
        r = f32x2{__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(sm) : "v"(r), "s"(kln));
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(t), "v"(sm), "v"(tb));
        return o;
    };
    auto cq2 = [&](int k, auto ec) -> f32x2 {              // elements e, e + 1 (e even) of coefficient set k (0 scale, 1 shift, 2 time bias)
        constexpr int e = decltype(ec)::value;
        return f32x2{cq[2 * k + (e >> 2)][e & 3], cq[2 * k + (e >> 2)][(e & 3) + 1]};
    };
    auto tpart = [&](uint32_t v, auto qc, uint32_t vmask) -> uint32_t {
        constexpr int q = decltype(qc)::value;
        constexpr std::integral_constant<int, 2 * q> E{};
        const f32x2 x = {__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)};
        const f32x2 m = mish_tb2(x, cq2(0, E), cq2(1, E), cq2(2, E));
        return pack_bf16(m.x, m.y) & vmask;                  // rows above / below the image stay zero padding
    };
    auto piece_addr = [&](int buf, int i) -> uint32_t { return lds0 + buf * PXBUF + (wv + 4 * i) * 1024 + l * 16; };
    auto piece_mask = [&](int i) -> uint32_t {
        uint32_t m = xpix[i] >= 0 ? ~0u : 0u;
        asm volatile("" : "+v"(m));                          // a mask, not a branch around the arithmetic
        return m;
    };
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    // fp32 input: half h (elements 4h .. 4h + 3 of the lane's 8 channels) of a piece in registers -> two packed registers of `dst`
    auto xhalf = [&](const u32x4& r, auto hc, uint32_t vmask, u32x4& dst) {
        constexpr int h = decltype(hc)::value;
        constexpr std::integral_constant<int, 4 * h> E0{};
        constexpr std::integral_constant<int, 4 * h + 2> E1{};
        const f32x2 m0_ = mish_tb2(f32x2{__uint_as_float(r.x), __uint_as_float(r.y)}, cq2(0, E0), cq2(1, E0), cq2(2, E0));
        const f32x2 m1_ = mish_tb2(f32x2{__uint_as_float(r.z), __uint_as_float(r.w)}, cq2(0, E1), cq2(1, E1), cq2(2, E1));
        dst[2 * h] = pack_bf16(m0_.x, m0_.y) & vmask; dst[2 * h + 1] = pack_bf16(m1_.x, m1_.y) & vmask;
    };

    // fp32 input, fused (round 4): raw staging area behind the two bf16 buffers -- [wave][slot 0..2][2 KB]: a piece = 8 pixels x 64 fp32
    // channels = two DMA instructions; lane l fetches channels xcol + 4 j .. + 3 (j = 0, 1) of ITS pixel, so that the same lane reads
    // back its own 8 channels (conflict-free by construction), transforms them and writes the 16 bf16 bytes of its slot of the tile
    constexpr uint32_t RAW0 = 2 * PXBUF;
    constexpr int RHP = PXPW / 2;                            // pieces per half chunk and wave (= staging slots): 3 (128-pixel tiles) or 2
    auto stage_raw = [&](int ch, int i, int slot) {
        const int cc0 = cof(ch) * PCK;
        const bool second = cc0 >= a.K1;
        const float* src = reinterpret_cast<const float*>(second ? a.x2 : a.x);
        const int ld = second ? a.ldx2 : a.ldx, cc = second ? cc0 - a.K1 : cc0;
        int xp = xpix[i];
        asm volatile("" : "+v"(xp));
        size_t off = (size_t)max(xp, 0) * ld + cc + xcol;
        asm volatile("" : "+v"(off));
        const float* pf = xp >= 0 ? src + off : reinterpret_cast<const float*>(a.zero) + (l & 7) * 8;
        glds16(pf, lds0 + RAW0 + (wv * RHP + slot) * 2048);
        glds16(pf + 4, lds0 + RAW0 + (wv * RHP + slot) * 2048 + 1024);
    };
    auto raw_half = [&](int buf, int half) {                  // pieces RHP half .. of this wave: staging area -> transformed bf16 tile
        typedef __attribute__((address_space(3))) u32x4 lds_u32x4r;
        u32x4 rv[RHP][2];
#pragma unroll
        for (int i = 0; i < RHP; ++i) {
            rv[i][0] = *(lds_u32x4r*)(uintptr_t)(lds0 + RAW0 + (wv * RHP + i) * 2048 + l * 16);
            rv[i][1] = *(lds_u32x4r*)(uintptr_t)(lds0 + RAW0 + (wv * RHP + i) * 2048 + 1024 + l * 16);
        }
#pragma unroll
        for (int i = 0; i < RHP; ++i) {
            u32x4 o;
            if constexpr (FUSE) {
                const uint32_t vm = piece_mask(RHP * half + i);
                xhalf(rv[i][0], std::integral_constant<int, 0>{}, vm, o); xhalf(rv[i][1], std::integral_constant<int, 1>{}, vm, o);
            } else {                                         // the plain conv: one rounding (rows outside the image were read from the zero page)
                o = u32x4{pack_bf16(__uint_as_float(rv[i][0].x), __uint_as_float(rv[i][0].y)), pack_bf16(__uint_as_float(rv[i][0].z), __uint_as_float(rv[i][0].w)),
                          pack_bf16(__uint_as_float(rv[i][1].x), __uint_as_float(rv[i][1].y)), pack_bf16(__uint_as_float(rv[i][1].z), __uint_as_float(rv[i][1].w))};
            }
            *(lds_u32x4r*)(uintptr_t)piece_addr(buf, RHP * half + i) = o;
        }
    };

    // ---- fragment addressing.  A 32-pixel MFMA block = output row i (0..3) of every 4-row band of the tile: lane q = l & 31 ->
    //      column x = q % W of band (q / W) % (TH / 4) of image q / W / (TH / 4) (W = 32: one row of one image; 16: rows i, i + 4;
    //      8: rows i, i + 4 of two images).  With that, the block of tap row ky of output row i is "halo row r = i + ky of every
    //      band" -- ONE activation fragment X_r serves the three tap rows (output rows r, r - 1, r - 2), and its two column shifts
    //      are made once: per 16-channel step 6 LDS fragment reads and 12 DPP shifts feed 36 MFMAs (before: 12 reads, 24 shifts).
    //      Lane -> 8-channel piece 2 ks + (l >> 5) of its pixel; weights: lane -> its own 16 bytes of the fragment.
    uint32_t xr[BH + 2];                                     // byte address of X_r, 16-channel step 0, buffer 0
    int ep_p0;                                               // epilogue: tile pixel of (output row 0, this lane)
    {
        const int q = l & 31, x = q & (a.W - 1), rest = q >> lw, nsub = 1 << a.lnsub;            // nsub = TH / BH
        const int sub = rest & (nsub - 1), ti = rest >> a.lnsub;
#pragma unroll
        for (int r = 0; r < BH + 2; ++r) {
            const int hp = (ti * TH2 + r + BH * sub) * a.W + x;
            const int swz = w8 ? (((hp >> 1) & 3) | (((hp >> 4) & 1) << 2)) : ((hp >> 1) & 7);
            xr[r] = lds0 + hp * 128 + (((l >> 5) * 16) ^ (swz * 16));
        }
        ep_p0 = (ti * a.TH + BH * sub) * a.W + x;
    }
    const int xin = l & 31 & (a.W - 1);
    const uint32_t mask_l = xin == 0 ? 0u : ~0u, mask_r = xin == a.W - 1 ? 0u : ~0u;     // zero padding left / right of the row

    f32x16 acc[BH];
#pragma unroll
    for (int i = 0; i < BH; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 XA, XB, XP, XQ;                                   // centre fragments: rows 0 and BH + 1 of a step; odd / even rows 1 .. BH

    // bf16 output: the bias of this lane's four channel quads, requested before anything else and used only by the epilogue (there a
    // load would put an HBM round trip in front of the tile's way out)
    // (round 5, pinned loop: the 128 bias values of the tile wait in LDS instead of 16 registers per lane)
    constexpr bool PIPE_ = (MI_PW_PIPE != 0) && (VAR < 2 || FPIPE) && !F32 && ABL == 0;
    constexpr uint32_t BIASL = 2 * PXBUF + 1024;             // 512 bytes behind the dump area
    constexpr uint32_t COEFL = 2 * PXBUF + 2048;             // FPIPE: the coefficient table, [K / 4][scale x 4, shift x 4, time bias x 4] = 12 K bytes
    f32x4 bias_q[(OUT16 && !FUSE) ? 4 : 1];
    if constexpr (OUT16 && (!FUSE || FPIPE) && PIPE_) {
        typedef __attribute__((address_space(3))) f32x4 lds_f32x4b;
        if (t < 32) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (a.bias && n0 + 4 * t < a.Nc) bv = *reinterpret_cast<const f32x4*>(a.bias + n0 + 4 * t);
            *(lds_f32x4b*)(uintptr_t)(lds0 + BIASL + 16 * t) = bv;
        }
    } else
    if constexpr (OUT16 && !FUSE) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            bias_q[rq] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias && live) bias_q[rq] = *reinterpret_cast<const f32x4*>(a.bias + n0 + 4 * (8 * wv + 2 * rq + (l >> 5)));
        }
    }
    // ---- prologue: the first chunk's rows, the first step's fragments
    if constexpr (!EARLYW) MI_PW_NOW(10);
    // ---- fused variants on the pinned loop: table operands (oldest requests), the first step's fragments, chunk 0's pieces through registers
    constexpr int RSNF = IN32 ? 2 : 1;
    auto fx_read = [&](int ch, auto hc, f32x4 (&cf)[3]) {      // coefficients of half h (4 channels) of this lane's slot of chunk ch
        constexpr int h = decltype(hc)::value;
        typedef __attribute__((address_space(3))) f32x4 lds_f32x4c;
        const uint32_t ad = lds0 + COEFL + (uint32_t)(cof(ch) * (PCK / 4) + (xcol >> 2) + h) * 48u;
        cf[0] = *(lds_f32x4c*)(uintptr_t)ad; cf[1] = *(lds_f32x4c*)(uintptr_t)(ad + 16); cf[2] = *(lds_f32x4c*)(uintptr_t)(ad + 32);
    };
    auto fx_apply = [&](u32x4 (&r)[RSNF], auto hc, const f32x4 (&cf)[3], u32x4& o) {   // half h of a piece in registers -> two packed registers of o (o may be r[0]: in place)
        constexpr int h = decltype(hc)::value;
        f32x2 x0, x1;
        if constexpr (IN32) {
            x0 = f32x2{__uint_as_float(r[h].x), __uint_as_float(r[h].y)}; x1 = f32x2{__uint_as_float(r[h].z), __uint_as_float(r[h].w)};
        } else {
            const uint32_t v0 = r[0][2 * h], v1 = r[0][2 * h + 1];
            x0 = f32x2{__uint_as_float(v0 << 16), __uint_as_float(v0 & 0xffff0000u)}; x1 = f32x2{__uint_as_float(v1 << 16), __uint_as_float(v1 & 0xffff0000u)};
        }
#if MI_PW_FXABL == 1      // profiling build: no transform (wrong results; the floor of the fused loop)
        o[2 * h] = pack_bf16(x0.x + cf[0].x, x0.y + cf[1].x); o[2 * h + 1] = pack_bf16(x1.x + cf[2].x, x1.y);
#elif MI_PW_FXABL == 2    // plain (unpacked) fp32 instructions, one asm statement each so that nothing is SLP-packed
        auto m1 = [&](float x, float sc, float sh, float tb) -> float {
            float t_, e_, q_, r_, s_, o_;
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t_) : "v"(x), "v"(sc), "v"(sh));
            asm("v_exp_f32 %0, %1" : "=v"(e_) : "v"(t_));
            asm("s_nop 0\n\tv_add_f32 %0, 2.0, %1" : "=v"(q_) : "v"(e_));
            asm("v_fma_f32 %0, %1, %2, 2.0" : "=v"(q_) : "v"(e_), "0"(q_));
            asm("v_rcp_f32 %0, %1" : "=v"(r_) : "v"(q_));
            asm("s_nop 0\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(s_) : "v"(r_), "s"(-1.38629436f), "v"(0.693147182f));
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(o_) : "v"(t_), "v"(s_), "v"(tb));
            return o_;
        };
        o[2 * h] = pack_bf16(m1(x0.x, cf[0].x, cf[1].x, cf[2].x), m1(x0.y, cf[0].y, cf[1].y, cf[2].y));
        o[2 * h + 1] = pack_bf16(m1(x1.x, cf[0].z, cf[1].z, cf[2].z), m1(x1.y, cf[0].w, cf[1].w, cf[2].w));
#else
        const f32x2 m0_ = mish_tb2(x0, f32x2{cf[0].x, cf[0].y}, f32x2{cf[1].x, cf[1].y}, f32x2{cf[2].x, cf[2].y});
        const f32x2 m1_ = mish_tb2(x1, f32x2{cf[0].z, cf[0].w}, f32x2{cf[1].z, cf[1].w}, f32x2{cf[2].z, cf[2].w});
        o[2 * h] = pack_bf16(m0_.x, m0_.y); o[2 * h + 1] = pack_bf16(m1_.x, m1_.y);
#endif
    };
    if constexpr (FPIPE) {
        // issue order = landing order: the statistics (VAR 3), this lane's coefficients of chunk 0 (registers: chunk 0 is transformed before
        // the table is published), the table's operands, the first step's fragments, chunk 0's pieces
        if constexpr (VAR == 3) load_stats();
        load_coef(0);
        u32x4 tq[3];
        const int c4 = min(4 * t, a.K - 4);
        {
            const float* p0; const float* p1; const float* p2;
            if constexpr (VAR == 2) {
                p0 = a.coef + (size_t)img0 * a.K + c4;
                const size_t pl = (size_t)a.N * a.K;
                p1 = p0 + pl; p2 = p0 + 2 * pl;
            } else {
                p0 = a.gamma + c4; p1 = a.beta + c4;
                p2 = a.temb ? a.temb + (size_t)img0 * a.ldt + c4 : reinterpret_cast<const float*>(a.zero);
            }
            asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %4, off\n\tglobal_load_dwordx4 %2, %5, off"
                         : "=&v"(tq[0]), "=&v"(tq[1]), "=&v"(tq[2]) : "v"(p0), "v"(p1), "v"(p2) : "memory");
        }
        static_for<0, NPART>([&](auto pc) { load_w3(0, std::integral_constant<int, 0>{}, pc); });
        u32x4 pr[PXPW][RSNF];
        static_for<0, PXPW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const uint64_t sb = (uint64_t)(uintptr_t)(reinterpret_cast<const uint8_t*>(a.x) + ((size_t)prow[i] * a.ldx + cof(0) * PCK) * XSZ);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
            gload16s<0>(pr[i][0], ((uint64_t)hi << 32) | lo, lane_off1);
            if constexpr (IN32) gload16s<16>(pr[i][1], ((uint64_t)hi << 32) | lo, lane_off1);
        });
        // rows outside the image: zero in both buffers, once (their requests land in the dump area)
        {
            typedef __attribute__((address_space(3))) u32x4 lds_u32x4z;
#pragma unroll
            for (int i = 0; i < PXPW; ++i)
                if (!((pvalid >> i) & 1u)) {
                    *(lds_u32x4z*)(uintptr_t)(lds0 + (wv + 4 * i) * 1024 + l * 16) = u32x4{0u, 0u, 0u, 0u};
                    *(lds_u32x4z*)(uintptr_t)(lds0 + PXBUF + (wv + 4 * i) * 1024 + l * 16) = u32x4{0u, 0u, 0u, 0u};
                }
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(9 + PXPW * RSNF) : "memory");      // statistics, coefficients, table operands
        if constexpr (VAR == 3) stats_landed();
        asm volatile("" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]) :: "memory");
        {
            f32x4 sc = __builtin_bit_cast(f32x4, tq[0]), sh = __builtin_bit_cast(f32x4, tq[1]);
            if constexpr (VAR == 3) {
                const int grp = pw_fastdiv(c4, a.cpg_magic);                         // a quad lies in one group (cpg % 16 == 0)
                const float mf = __shfl(gmean, grp, 64), rstd = __shfl(grstd, grp, 64);
                sc *= rstd; sh -= mf * sc;
            }
            sc *= 1.44269504f; sh *= 1.44269504f;                                     // mish_tb2 feeds v_exp_f32 directly
            typedef __attribute__((address_space(3))) f32x4 lds_f32x4c;
            if (4 * t < a.K) {
                const uint32_t ad = lds0 + COEFL + (uint32_t)t * 48u;
                *(lds_f32x4c*)(uintptr_t)ad = sc; *(lds_f32x4c*)(uintptr_t)(ad + 16) = sh; *(lds_f32x4c*)(uintptr_t)(ad + 32) = __builtin_bit_cast(f32x4, tq[2]);
            }
        }
        coef_landed(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // chunk 0's pieces (and the first step's fragments)
        static_for<0, PXPW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            landed16(pr[i][0]);
            u32x4 o;
            if constexpr (IN32) {
                landed16(pr[i][1]);
                xhalf(pr[i][0], std::integral_constant<int, 0>{}, ~0u, o); xhalf(pr[i][1], std::integral_constant<int, 1>{}, ~0u, o);
            } else {
                static_for<0, 4>([&](auto qc) { o[decltype(qc)::value] = tpart(pr[i][0][decltype(qc)::value], qc, ~0u); });
            }
            const uint32_t off = ((pvalid >> i) & 1u) ? (uint32_t)(4 * i * 1024) : (uint32_t)(DUMP - wv * 1024);
            *(lds_u32x4*)(uintptr_t)(lds0 + wv * 1024 + l * 16 + off) = o;
        });
    }
    if constexpr (VAR == 3 && !FPIPE) load_stats();
    if constexpr (FUSE && !FPIPE) load_coef(0);
    if constexpr (FPIPE) {
    } else if constexpr (IN32) {
        u32x4 pr[PXPW][2];
#pragma unroll
        for (int i = 0; i < PXPW; ++i) load_x32(0, i, pr[i]);
        static_for<0, NPART>([&](auto pc) { load_w3(0, std::integral_constant<int, 0>{}, pc); });
        if constexpr (VAR == 3) {                            // the statistics are the oldest requests: their arithmetic runs under the rows' flight
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(6 + 2 * PXPW + 9) : "memory");
            stats_landed();
        }
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");     // the rows of chunk 0 (and, older, the coefficients)
        if constexpr (FUSE) {
            coef_landed(0);
#pragma unroll
            for (int i = 0; i < PXPW; ++i) {
                landed16(pr[i][0]); landed16(pr[i][1]);
                const uint32_t vm = piece_mask(i);
                u32x4 o;
                xhalf(pr[i][0], std::integral_constant<int, 0>{}, vm, o); xhalf(pr[i][1], std::integral_constant<int, 1>{}, vm, o);
                *(lds_u32x4*)(uintptr_t)piece_addr(0, i) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PXPW; ++i) store_x32(0, i, pr[i]);
            if constexpr (SPIECE) {                          // pinned loop: pieces outside the image are never written again (buffer 0 just got its zeros)
                typedef __attribute__((address_space(3))) u32x4 lds_u32x4z;
#pragma unroll
                for (int i = 0; i < PXPW; ++i)
                    if (!((pvalid >> i) & 1u)) *(lds_u32x4z*)(uintptr_t)(lds0 + PXBUF + (wv + 4 * i) * 1024 + l * 16) = u32x4{0u, 0u, 0u, 0u};
            }
        }
    } else {
        if constexpr (!EARLYW) {
#pragma unroll
            for (int i = 0; i < PXPW; ++i) stage_x(0, i);
            static_for<0, NPART>([&](auto pc) { load_w3(0, std::integral_constant<int, 0>{}, pc); });
        }
    }
    // bf16 input, fused: this wave's six pieces of a chunk, rewritten in place in LDS in ONE go -- three pieces (twelve independent
    // exp -> rcp chains per lane) in flight at a time.  Round 4: the main loop does this at the chunk boundary instead of spreading
    // the parts over the units of the steps (two element pairs per unit, "under" the MFMAs): a wave issues in order, so the two
    // dependent transcendental chains of a unit stalled the MFMAs queued behind them -- the interleaved form cost 4.3 us per chunk of
    // a level-0 launch, the same work as a block 1.5 us (the other workgroup of the CU owns the matrix pipe meanwhile), and the block
    // can be skipped for the last chunk, which has no successor (a third of the transforms at K = 128).
    constexpr int HP = PXPW / 2;                             // pieces per half chunk and wave: 3 (128-pixel tiles) or 2
    auto transform_chunk = [&](int buf) {
        static_for<0, 2>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            u32x4 pv[HP];
#pragma unroll
            for (int i = 0; i < HP; ++i) pv[i] = __builtin_bit_cast(u32x4, lds_b128p(piece_addr(buf, HP * h + i)));
#pragma unroll
            for (int i = 0; i < HP; ++i) {
                const uint32_t vm = piece_mask(HP * h + i);
                u32x4 o;
                static_for<0, 4>([&](auto qc) { o[decltype(qc)::value] = tpart(pv[i][decltype(qc)::value], qc, vm); });
                *(lds_u32x4*)(uintptr_t)piece_addr(buf, HP * h + i) = o;
            }
        });
    };
    if constexpr (FUSE && !IN32 && !FPIPE) {
        if constexpr (VAR == 3) {                            // the statistics are the oldest requests: their arithmetic runs under the rows' flight
            asm volatile("s_waitcnt vmcnt(%0)" :: "i"(6 + PXPW + 9) : "memory");
            stats_landed();
        }
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");     // coefficients and rows of chunk 0
        coef_landed(0);
        transform_chunk(0);
    }
    // the rows and the fragments have landed (this wave's; the fused variant's rewritten pieces are in LDS) ...
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    MI_PW_NOW(11);
    __builtin_amdgcn_s_barrier();                                          // ... every wave's
    asm volatile("" ::: "memory");
    MI_PW_NOW(12);
    // ---- round 5: the PINNED schedule of the plain bf16 conv (VAR 0 / 1).  The unit timeline of the loop below (tools/pw_timeline.py,
    //      one wave per SIMD, 512 -> 512 @8x8) read 6 450 cycles per chunk against 4 608 of MFMAs: hipcc sinks each LDS read next to its
    //      first use and clusters the three fragment requests of a unit between two MFMAs (every unit +50..150 cycles), and the chunk
    //      boundary's vmcnt(0) waited out the fragment requests issued one unit earlier (+600).  Here every MFMA is followed by at most
    //      one side operation and a sched_barrier, as a software pipeline over the row units:
    //        unit u issues its MFMAs from fragments that are complete in registers (centre, left, right), reads the centre fragment
    //        of unit u + 2 (first gap) and shifts unit u + 1's (gaps 1, 2; the last unit of a step prepares the next step's unit 0);
    //        the next step's nine fragment requests go out one per gap (every other gap from four blocks per wave up) from the start
    //        of the step, the next chunk's activation pieces behind them in steps 0 and 1 only -- so step 3's opening wait covers them
    //        and the barrier (before unit BH - 1 of step 3, the first to read the other buffer) needs no vmcnt at all; the fragments
    //        of the next chunk's first step are waited for at the very end of step 3, >= 18 gaps after their request.
    constexpr bool PIPE = PIPE_;
    if constexpr (PIPE) {
        constexpr int CN = (BH % 3 == 1) ? 4 : 3;            // centre fragment registers in rotation (unit u: C[u % CN])
        constexpr int WSTR = BH >= 4 ? MI_PW_WSTR : 1;        // a fragment request every WSTR gaps
        constexpr int PG0 = 9 * WSTR;                         // first gap with an activation piece request
        constexpr int PPS = (PXPW + 1) / 2;                   // pieces requested per step (steps 0 and 1)
        // the four waves leave a barrier in the same cycle and would issue every request in the same gap: wave w starts w * 16 cycles late
        // (one dwordx4 request occupies the CU's address path for about 16 cycles)
        auto pw_skew = [&]() { if constexpr (MI_PW_SKEW > 0) for (int i_ = 0; i_ < wv * MI_PW_SKEW; ++i_) asm volatile("s_nop 15"); };
        pw_skew();
        // the next chunk's pieces through registers (an LDS-DMA request cost the issuing wave 50-100 cycles in the unit timeline, a plain
        // request + ds_write_b128 a fraction of that): batch 0 = the first PPS pieces, requested in step 0 and written in step 2 (in-order
        // returns: they have landed once step 2's fragments have), batch 1 requested in step 1 and written in step 3 before the barrier
        constexpr bool RST = (MI_PW_RSTAGE != 0 || IN32 || FPIPE) && SPIECE;
        constexpr int RSN = IN32 ? 2 : 1;                      // registers sets per piece: fp32 input = 32 bytes per lane
        u32x4 RS[2][RST ? PPS : 1][RSN];
        const uint32_t wbase = lds0 + wv * 1024 + l * 16;
        auto rs_load = [&](int ch, auto bc, auto jc) {
            constexpr int b = decltype(bc)::value, j = decltype(jc)::value, i = PPS * b + j;
            const int cc0 = cof(ch) * PCK;
            const bool second = cc0 >= a.K1;
            const uint8_t* src = reinterpret_cast<const uint8_t*>(second ? a.x2 : a.x);
            const int ld = second ? a.ldx2 : a.ldx, cc = second ? cc0 - a.K1 : cc0;
            const uint64_t sb = (uint64_t)(uintptr_t)(src + ((size_t)prow[i < PXPW ? i : 0] * ld + cc) * XSZ);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
            gload16s<0>(RS[b][j][0], ((uint64_t)hi << 32) | lo, second ? lane_off2 : lane_off1);
            if constexpr (IN32) gload16s<16>(RS[b][j][1], ((uint64_t)hi << 32) | lo, second ? lane_off2 : lane_off1);
        };
        auto rs_store = [&](int ch, auto bc, auto jc) {
            constexpr int b = decltype(bc)::value, j = decltype(jc)::value, i = PPS * b + j;
            typedef __attribute__((address_space(3))) u32x4 lds_u32x4s;
            landed16(RS[b][j][0]);
            const uint32_t off = ((pvalid >> i) & 1u) ? (uint32_t)((ch & 1) * PXBUF + 4 * i * 1024) : (uint32_t)(DUMP - wv * 1024);
            if constexpr (IN32) {                            // one rounding, as mi_f32_to_bf16 does
                landed16(RS[b][j][1]);
                const u32x4 r0 = RS[b][j][0], r1 = RS[b][j][1];
                *(lds_u32x4s*)(uintptr_t)(wbase + off) =
                    u32x4{pack_bf16(__uint_as_float(r0.x), __uint_as_float(r0.y)), pack_bf16(__uint_as_float(r0.z), __uint_as_float(r0.w)),
                          pack_bf16(__uint_as_float(r1.x), __uint_as_float(r1.y)), pack_bf16(__uint_as_float(r1.z), __uint_as_float(r1.w))};
            } else
                *(lds_u32x4s*)(uintptr_t)(wbase + off) = RS[b][j][0];
        };
        bf16x8 Ac, Bc, C[CN];
        f32x4 CF[2][3];                                        // fused: the coefficients of the half pieces being transformed
        u32x4 Al, Bl, Ar, Br, L[2], R[2];                      // shifted fragments (registers written two at a time: pw_shift_h)
        // A shift is four DPP instructions; from three blocks per wave up it is issued as two halves in two consecutive gaps (a gap then
        // carries at most two DPP instructions + one memory instruction), and never needs wait states: its consumer is a unit away.
        constexpr bool HALVES = BH >= 3;
        auto sh = [&](u32x4& dst, const bf16x8& src, auto dirc, auto slotc) {         // slot 0 / 1: lower / upper register pair
            constexpr int dir = decltype(dirc)::value, slot = decltype(slotc)::value;
            if constexpr (MI_PW_PABL & 8) { dst = __builtin_bit_cast(u32x4, src); }
            else if constexpr (HALVES) pw_shift_h<dir, slot>(dst, src, dir == 0 ? mask_l : mask_r);
            else if constexpr (slot == 0) { pw_shift_h<dir, 0>(dst, src, dir == 0 ? mask_l : mask_r); pw_shift_h<dir, 1>(dst, src, dir == 0 ? mask_l : mask_r); }
        };
        constexpr std::integral_constant<int, 0> I0{}; constexpr std::integral_constant<int, 1> I1{};
        // gap (within a unit) of shift part q = 0 .. 3 (left lower, left upper, right lower, right upper); -1: folded into its lower half
        auto sgap = [](int q) constexpr -> int { return HALVES ? 1 + q : (q == 0 ? 1 : q == 2 ? 2 : -1); };
        Ac = lds_b128p(xr[0]); Bc = lds_b128p(xr[BH + 1]); C[1 % CN] = lds_b128p(xr[1]);
        pw_shift_h<0, 0>(Al, Ac, mask_l); pw_shift_h<0, 1>(Al, Ac, mask_l); pw_shift_h<0, 0>(Bl, Bc, mask_l); pw_shift_h<0, 1>(Bl, Bc, mask_l);
        pw_shift_h<1, 0>(Ar, Ac, mask_r); pw_shift_h<1, 1>(Ar, Ac, mask_r); pw_shift_h<1, 0>(Br, Bc, mask_r); pw_shift_h<1, 1>(Br, Bc, mask_r);
        __builtin_amdgcn_sched_barrier(0);
        MI_PW_STAMP(4, 3);
        for (int ch = 0; ch < nchunks; ++ch) {
            const uint32_t xcur = (ch & 1) * PXBUF, xnxt = PXBUF - xcur;
            static_for<0, 4>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, cur = ks & 1;
                constexpr uint32_t kx32 = ks * 32, kxn = ((ks + 1) & 3) * 32;
                const uint32_t bufn = ks == 3 ? xnxt : xcur;                      // where the next step's rows are
                constexpr int prevp = (ks == 1 || ks == 2) ? ((PXPW - PPS * (ks - 1)) < PPS ? (PXPW - PPS * (ks - 1)) : PPS) : 0;
                if constexpr (ks > 0 && !(MI_PW_PABL & 4)) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(prevp * (RST ? RSN : 1)) : "memory");
                static_for<0, 9>([&](auto tc) { landed16(WB[cur][decltype(tc)::value]); });
                auto mm = [&](auto ic, auto tapc, const bf16x8& xf) {
                    constexpr int i = decltype(ic)::value, tp = decltype(tapc)::value;
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WB[cur][tp]), xf, acc[i], 0, 0, 0);
                };
                auto mmu = [&](auto ic, auto tapc, const u32x4& xf) { mm(ic, tapc, __builtin_bit_cast(bf16x8, xf)); };
                // what rides in gap g of the step besides the unit's own preparation: fragment requests, then piece requests
                auto gside = [&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if constexpr (g % WSTR == 0 && g / WSTR < 9 && !(MI_PW_PABL & 1)) load_w1(ch, std::integral_constant<int, ks + 1>{}, std::integral_constant<int, g / WSTR>{});
                    if constexpr (!(MI_PW_PABL & 2) && ks < 2 && g >= PG0 && (g - PG0) % 2 == 0 && (g - PG0) / 2 < PPS && PPS * ks + (g - PG0) / 2 < PXPW)
                    {
                        if constexpr (RST) rs_load(ch + 1, std::integral_constant<int, ks>{}, std::integral_constant<int, (g - PG0) / 2>{});
                        else if constexpr (SPIECE) stage_s(ch + 1, std::integral_constant<int, (PPS * ks + (g - PG0) / 2) < PXPW ? (PPS * ks + (g - PG0) / 2) : 0>{});
                        else stage_x(ch + 1, PPS * ks + (g - PG0) / 2);
                    }
                    // (register staging: the batch requested two steps ago goes to LDS in the odd gaps 1, 3, ... of steps 2 and 3)
                    if constexpr (FPIPE) {
                        // fused: the batch requested two steps ago (PPS pieces) is transformed IN its registers, half pieces at a time so that a
                        // half's coefficients (three ds_read_b128, the same for every piece: a lane keeps its channel slot) are read once per batch:
                        // gap 1: coefficients of half 0; gaps 2 .. 1 + PPS: half 0 of piece 0 .. PPS - 1; then half 1 the same way; then the PPS
                        // writes -- 2 PPS + 2 + PPS gaps from gap 1 (four blocks per wave: gaps 1 .. 11, the barrier of step 3 sits before gap 21;
                        // two blocks per wave: 18 gaps per step, barrier before gap 6 -- both halves' coefficients in gap 1, a piece per gap, the
                        // writes with the second piece).  The last chunk has no successor: its (clamped) requests were issued, nothing is transformed.
                        if constexpr (ks >= 2 && g >= 1) {
                            constexpr int b = ks - 2, q = g - 1;
                            constexpr int NPB = (PXPW - PPS * b) < PPS ? (PXPW - PPS * b) : PPS;     // pieces of this batch
                            auto fx_write = [&](auto jc) {
                                constexpr int j = decltype(jc)::value, i = PPS * b + j;
                                typedef __attribute__((address_space(3))) u32x4 lds_u32x4s;
                                const uint32_t off = ((pvalid >> i) & 1u) ? (uint32_t)(((ch + 1) & 1) * PXBUF + 4 * i * 1024) : (uint32_t)(DUMP - wv * 1024);
                                *(lds_u32x4s*)(uintptr_t)(wbase + off) = RS[b][j][0];
                            };
                            if (ch + 1 < nchunks) {
                                if constexpr (BH >= 4) {
                                    if constexpr (q == 0) { static_for<0, NPB>([&](auto jc) { landed16(RS[b][decltype(jc)::value][0]); if constexpr (IN32) landed16(RS[b][decltype(jc)::value][1]); }); fx_read(ch + 1, I0, CF[0]); }
                                    if constexpr (q >= 1 && q <= NPB) fx_apply(RS[b][q - 1], I0, CF[0], RS[b][q - 1][0]);
                                    if constexpr (q == NPB) fx_read(ch + 1, I1, CF[1]);
                                    if constexpr (q >= NPB + 1 && q <= 2 * NPB) fx_apply(RS[b][q - NPB - 1], I1, CF[1], RS[b][q - NPB - 1][0]);
                                    if constexpr (q >= 2 * NPB + 1 && q <= 3 * NPB) fx_write(std::integral_constant<int, q - 2 * NPB - 1>{});
                                } else {
                                    if constexpr (q == 0) { static_for<0, NPB>([&](auto jc) { landed16(RS[b][decltype(jc)::value][0]); if constexpr (IN32) landed16(RS[b][decltype(jc)::value][1]); }); fx_read(ch + 1, I0, CF[0]); fx_read(ch + 1, I1, CF[1]); }
                                    if constexpr (q >= 1 && q <= NPB) {
                                        fx_apply(RS[b][q - 1], I0, CF[0], RS[b][q - 1][0]); fx_apply(RS[b][q - 1], I1, CF[1], RS[b][q - 1][0]);
                                        fx_write(std::integral_constant<int, q - 1>{});
                                    }
                                }
                            }
                        }
                    } else
                    if constexpr (RST && ks >= 2 && (g & 1) && g / 2 < PPS && PPS * (ks - 2) + g / 2 < PXPW)
                        rs_store(ch + 1, std::integral_constant<int, ks - 2>{}, std::integral_constant<int, g / 2>{});
                };
                // ---- unit 0: rows 0 (output row 0, tap row 0) and BH + 1 (output row BH - 1, tap row 2); it reads row 2 and shifts row 1
                {
                    MI_PW_STAMP(ks, ch * (BH + 1));
                    static_for<0, 6>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        if constexpr (j == 0) mm(std::integral_constant<int, 0>{}, std::integral_constant<int, 0 * 3 + 1>{}, Ac);
                        if constexpr (j == 1) mm(std::integral_constant<int, BH - 1>{}, std::integral_constant<int, 2 * 3 + 1>{}, Bc);
                        if constexpr (j == 2) mmu(std::integral_constant<int, 0>{}, std::integral_constant<int, 0 * 3 + 0>{}, Al);
                        if constexpr (j == 3) mmu(std::integral_constant<int, BH - 1>{}, std::integral_constant<int, 2 * 3 + 0>{}, Bl);
                        if constexpr (j == 4) mmu(std::integral_constant<int, 0>{}, std::integral_constant<int, 0 * 3 + 2>{}, Ar);
                        if constexpr (j == 5) mmu(std::integral_constant<int, BH - 1>{}, std::integral_constant<int, 2 * 3 + 2>{}, Br);
                        if constexpr (j == 0) C[2 % CN] = lds_b128p((xr[2] ^ kx32) + xcur);
                        if constexpr (j == sgap(0)) sh(L[1], C[1 % CN], I0, I0);
                        if constexpr (j == sgap(1)) sh(L[1], C[1 % CN], I0, I1);
                        if constexpr (j == sgap(2)) sh(R[1], C[1 % CN], I1, I0);
                        if constexpr (j == sgap(3)) sh(R[1], C[1 % CN], I1, I1);
                        gside(jc);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
                // ---- units 1 .. BH: row u feeds output rows u - ky (tap row ky)
                static_for<1, BH + 1>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    constexpr int ky0 = u == BH ? 1 : 0, ky1 = u == 1 ? 1 : 2, n = ky1 - ky0 + 1;      // valid tap rows ky0 .. ky1
                    constexpr int g0 = u == 1 ? 6 : 12 + 9 * (u - 2);                               // the unit's first gap of the step
                    MI_PW_STAMP(ks, ch * (BH + 1) + u);
                    if constexpr (ks == 3 && u == BH - 1) {
                        // chunk boundary: every wave has read what it needs of this chunk's rows (the last read, row BH, was issued
                        // >= 6 gaps ago) and its pieces of the next chunk landed before step 3 began
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if constexpr (!(MI_PW_PABL & 16)) __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                        pw_skew();
                    }
                    static_for<0, 3 * n>([&](auto jc) {
                        constexpr int j = decltype(jc)::value, kxi = j / n, ky = ky0 + j % n, kx = kxi == 0 ? 1 : kxi == 1 ? 0 : 2;
                        if constexpr (kx == 1) mm(std::integral_constant<int, u - ky>{}, std::integral_constant<int, ky * 3 + kx>{}, C[u % CN]);
                        else mmu(std::integral_constant<int, u - ky>{}, std::integral_constant<int, ky * 3 + kx>{}, kx == 0 ? L[u & 1] : R[u & 1]);
                        if constexpr (j == 0) {
                            if constexpr (u + 2 <= BH) C[(u + 2) % CN] = lds_b128p((xr[u + 2 <= BH ? u + 2 : 0] ^ kx32) + xcur);
                            else if constexpr (u == BH - 1) { Ac = lds_b128p((xr[0] ^ kxn) + bufn); Bc = lds_b128p((xr[BH + 1] ^ kxn) + bufn); }
                            else C[1 % CN] = lds_b128p((xr[1] ^ kxn) + bufn);
                        }
                        // the shifts of unit u + 1's row; the last unit prepares the next step's unit 0 (four shifts: from three blocks per
                        // wave up the two left ones ride in the spare gaps of unit BH - 1, whose first gap read those rows)
                        constexpr bool SPLIT0 = BH >= 3;
                        if constexpr (u < BH) {
                            if constexpr (j == sgap(0)) sh(L[(u + 1) & 1], C[(u + 1) % CN], I0, I0);
                            if constexpr (j == sgap(1)) sh(L[(u + 1) & 1], C[(u + 1) % CN], I0, I1);
                            if constexpr (j == sgap(2)) sh(R[(u + 1) & 1], C[(u + 1) % CN], I1, I0);
                            if constexpr (j == sgap(3)) sh(R[(u + 1) & 1], C[(u + 1) % CN], I1, I1);
                            if constexpr (SPLIT0 && u == BH - 1) {
                                if constexpr (j == 5) sh(Al, Ac, I0, I0);
                                if constexpr (j == 6) sh(Al, Ac, I0, I1);
                                if constexpr (j == 7) sh(Bl, Bc, I0, I0);
                                if constexpr (j == 8) sh(Bl, Bc, I0, I1);
                            }
                        } else if constexpr (SPLIT0) {
                            if constexpr (j == 1) sh(Ar, Ac, I1, I0);
                            if constexpr (j == 2) sh(Ar, Ac, I1, I1);
                            if constexpr (j == 3) sh(Br, Bc, I1, I0);
                            if constexpr (j == 4) sh(Br, Bc, I1, I1);
                        } else {
                            if constexpr (j == 1) sh(Al, Ac, I0, I0);
                            if constexpr (j == 2) sh(Bl, Bc, I0, I0);
                            if constexpr (j == 3) sh(Ar, Ac, I1, I0);
                            if constexpr (j == 4) sh(Br, Bc, I1, I0);
                        }
                        gside(std::integral_constant<int, g0 + j>{});
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
                // the fragments of the next chunk's first step: no register-destination load crosses the loop's back edge (sixth rule)
                if constexpr (ks == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            });
        }
        asm volatile("" :: "v"(Ac), "v"(Bc), "v"(Al), "v"(Bl), "v"(Ar), "v"(Br));
    } else {
    XA = lds_b128p(xr[0]); XB = lds_b128p(xr[BH + 1]);

    // One chunk = four 16-channel steps of BH + 1 row units (rows {0, BH + 1}, then 1 .. BH: 6, 6, 9, 9, 6 MFMAs at BH = 4, 6, 6, 6 at
    // BH = 2; consecutive MFMAs go to different accumulators).  A unit reads the next unit's centre fragment(s); units 0-2 request the
    // next step's fragments (three taps each; units 0-1 at BH = 2), unit min(3, BH) of the first steps two activation pieces of the next chunk; a step
    // starts once everything the previous one requested before those pieces has landed.  Chunk boundary (before the last unit of
    // step 3, whose MFMAs then cover the first reads from the other buffer): every wave has read all it needs of this chunk's rows
    // and has its pieces of the next chunk's.
    // Fused variants, bf16 input: coefficients and all pieces of the next chunk are requested in step 0 and transformed as one block at
    // the chunk boundary (transform_chunk).  fp32 input (fused or not): the raw pieces of the next chunk arrive in the staging area half a
    // chunk at a time and are converted / transformed by raw_half at the end of step 1 and at the chunk boundary.  The last chunk has no
    // successor: its (clamped) requests are still issued -- the counted waits assume them -- but nothing is transformed.
    constexpr int XU = BH < 3 ? BH : 3;                      // the unit that requests activation pieces
    MI_PW_STAMP(4, 3);                                       // (rewrites slot 3 with the entry time; the new stamp = loop entry)
    for (int ch = 0; ch < nchunks; ++ch) {
        const uint32_t xcur = (ch & 1) * PXBUF, xnxt = PXBUF - xcur;
        static_for<0, 4>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value, cur = ks & 1, kx32 = ks * 32;
            // pieces the previous step requested behind its fragments
            constexpr int PPS = PT == 256 ? 3 : 2;             // activation pieces a step requests (unit XU)
            constexpr int prevp = ks == 0 ? 0 : (PXPW - PPS * (ks - 1) >= PPS ? PPS : (PXPW - PPS * (ks - 1) > 0 ? PXPW - PPS * (ks - 1) : 0));
            // (fused: step 0 requested the coefficients, its fragment parts and pieces, the last piece(s) (unit 2: two DMAs) behind
            //  everything step 1 needs; fp32 input: the second half chunk's six DMAs are requested at the end of step 1, behind step
            //  2's fragments)
            if constexpr (ks > 0 && !(ABL & 32))
                asm volatile("s_waitcnt vmcnt(%0)" :: "i"((FUSE || IN32) ? ((ks == 1 ? 2 : 0) + ((IN32 && ks == 2) ? 2 * (PXPW / 2) : 0)) : (ABL & 2) ? 0 : prevp) : "memory");
            static_for<0, 9>([&](auto tc) { landed16(WB[cur][decltype(tc)::value]); });
            auto mm = [&](auto ic, auto tapc, const bf16x8& xf) {
                constexpr int i = decltype(ic)::value, tp = decltype(tapc)::value;
                if constexpr (F32) {
                    const f32x4 wv4 = __builtin_bit_cast(f32x4, WB[cur][tp]), xv4 = __builtin_bit_cast(f32x4, xf);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv4[j], xv4[j], acc[i], 0, 0, 0);
                } else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WB[cur][tp]), xf, acc[i], 0, 0, 0);
            };
#define MI_MM(I, KY, KX, XF) mm(std::integral_constant<int, I>{}, std::integral_constant<int, (KY) * 3 + (KX)>{}, XF)
            auto sh_l = [&](const bf16x8& c) { if constexpr (ABL & 16) return c; else return pw_shift<0>(c, mask_l); };
            auto sh_r = [&](const bf16x8& c) { if constexpr (ABL & 16) return c; else return pw_shift<1>(c, mask_r); };
            // what a unit issues besides its MFMAs
            auto issue = [&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if constexpr (IN32 && FUSE && u == 0 && ks == 0) load_coef(ch + 1);      // (the oldest requests of the step)
                if constexpr (u < NPART && !(ABL & 1)) load_w3(ch, std::integral_constant<int, ks + 1>{}, uc);
                if constexpr (IN32) {
                    // the raw fp32 pieces of the next chunk go to the staging area by DMA, half a chunk at a time: the first half's pieces in
                    // step 0 (one per unit, behind the unit's fragments), the second half's at the end of step 1 (raw_half's caller).
                    // (Round 4: the plain fp32-input conv too -- it staged two pieces per step through registers with counted waits.)
                    if constexpr (ks == 0 && u < RHP) stage_raw(ch + 1, u, u);
                }
                if constexpr (FUSE && !IN32) {
                    // Round 4: ALL pieces of the next chunk are requested in step 0 (two per unit, behind the unit's fragments; a DMA
                    // needs no registers), so a piece has a whole step to arrive before it is read back and transformed -- requested
                    // one unit before the step that transforms it, every step began by waiting out an HBM round trip.
                    if constexpr (ks == 0 && u == 0) load_coef(ch + 1);
                    if constexpr (ks == 0 && 2 * u < PXPW) { stage_x(ch + 1, 2 * u); stage_x(ch + 1, 2 * u + 1); }
                } else if constexpr (u == XU && !IN32 && PPS * ks < PXPW && !(ABL & 2)) {
                    static_for<0, PPS>([&](auto pc) { constexpr int pi = PPS * ks + decltype(pc)::value; if constexpr (pi < PXPW) stage_x(ch + 1, pi); });
                }
            };
            // ---- unit 0: rows 0 (output row 0, tap row 0) and BH + 1 (output row BH - 1, tap row 2)
            {
                constexpr std::integral_constant<int, 0> U{};
                MI_PW_STAMP(ks, ch * (BH + 1));
                XP = lds_b128p((xr[1] ^ kx32) + xcur);
                issue(U);
                MI_MM(0, 0, 1, XA); MI_MM(BH - 1, 2, 1, XB);
                { const bf16x8 la = sh_l(XA), lb = sh_l(XB); MI_MM(0, 0, 0, la); MI_MM(BH - 1, 2, 0, lb); }
                { const bf16x8 ra = sh_r(XA), rb = sh_r(XB); MI_MM(0, 0, 2, ra); MI_MM(BH - 1, 2, 2, rb); }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- units 1 .. BH: row r feeds output rows r - ky (tap row ky); the last one reads rows 0 and BH + 1 of the next step
            static_for<1, BH + 1>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                MI_PW_STAMP(ks, ch * (BH + 1) + r);
                auto& xc = [&]() -> bf16x8& { if constexpr (r & 1) return XP; else return XQ; }();
                if constexpr (r < BH) {
                    auto& xn = [&]() -> bf16x8& { if constexpr (r & 1) return XQ; else return XP; }();
                    xn = lds_b128p((xr[r + 1] ^ kx32) + xcur);
                } else if constexpr (ks == 3) {
                    if constexpr (FUSE && !IN32) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (requested in step 0)
                        if (ch + 1 < nchunks) { coef_landed(ch + 1); transform_chunk((ch + 1) & 1); }
                    }
                    if constexpr (IN32) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the second half chunk: requested two steps ago)
                        if (ch + 1 < nchunks) raw_half((ch + 1) & 1, 1);
                    }
                    if constexpr (ABL & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    XA = lds_b128p(xr[0] + xnxt); XB = lds_b128p(xr[BH + 1] + xnxt);
                } else {
                    if constexpr (IN32 && ks == 1) {
                        // first half chunk (requested in step 0; younger: this step's nine fragment requests), then the second half's
                        // requests -- the staging slots are free once raw_half has read them
                        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                        if (ch + 1 < nchunks) { if constexpr (FUSE) coef_landed(ch + 1); raw_half((ch + 1) & 1, 0); }
                        static_for<0, RHP>([&](auto ic) { stage_raw(ch + 1, RHP + decltype(ic)::value, decltype(ic)::value); });
                    }
                    XA = lds_b128p((xr[0] ^ (kx32 + 32)) + xcur); XB = lds_b128p((xr[BH + 1] ^ (kx32 + 32)) + xcur);
                }
                issue(rc);
                // (ky, i = r - ky) with 0 <= i < BH, centre column first
                auto taps = [&](auto kxc, const bf16x8& xf) {
                    constexpr int kx = decltype(kxc)::value;
                    static_for<0, 3>([&](auto kyc) {
                        constexpr int ky = decltype(kyc)::value, i = r - ky;
                        if constexpr (i >= 0 && i < BH) MI_MM(i, ky, kx, xf);
                    });
                };
                taps(std::integral_constant<int, 1>{}, xc);
                { const bf16x8 lf = sh_l(xc); taps(std::integral_constant<int, 0>{}, lf); }
                { const bf16x8 rf = sh_r(xc); taps(std::integral_constant<int, 2>{}, rf); }
                __builtin_amdgcn_sched_barrier(0);
            });
#undef MI_MM
        });
    }
    }   // !PIPE
    MI_PW_STAMP(4, 0);                                       // slot 0: the last point's time; new stamp = loop end
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the clamped re-fetches must not outlive the workgroup's LDS ...
    static_for<0, 9>([&](auto tc) { landed16(WB[0][decltype(tc)::value]); landed16(WB[1][decltype(tc)::value]); });   // ... nor their registers
    if (a.ksplit > 1) {
        // one exchange per WAVE: wave w of slice s > 0 writes its accumulators (register quad q of block i = 1 KB: 16 bytes per lane) and raises
        // flag [s - 1][tile][w]; wave w of slice 0 waits for it, adds, and lowers it for the next launch.  The two workgroups may sit on different XCDs,
        // whose L2s do not see each other's lines: every access of the exchange carries sc1 (agent-coherent: written through / read past the L2), so
        // that no cache-wide write-back or invalidate is needed -- an agent-scope fence per wave (buffer_wbl2 / buffer_inv sc1) threw the weight
        // lines of every workgroup of the XCD out of its L2: 8x8 x 512 -> 512 at B = 64 took 34 us split against 23 us unsplit.
        const unsigned tile = blockIdx.y * gridDim.x + blockIdx.x, ntiles = gridDim.x * gridDim.y;
        if (!a.sk_flag) { if (slice > 0) return; }
        else if (slice > 0) {
            if (live) {
                float* part = a.sk_part + ((size_t)(slice - 1) * ntiles + tile) * (PT * 128) + (size_t)wv * (PT * 32) + l * 4;
#pragma unroll
                for (int i = 0; i < BH; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(part + (i * 4 + q) * 256), "v"(v) : "memory");
                    }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (l == 0) __hip_atomic_store(a.sk_flag + ((size_t)(slice - 1) * ntiles + tile) * 4 + wv, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        if (live && a.sk_flag) {
            for (int sl = 1; sl < a.ksplit; ++sl) {
                int* flag = a.sk_flag + ((size_t)(sl - 1) * ntiles + tile) * 4 + wv;
                while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(4);
                const float* part = a.sk_part + ((size_t)(sl - 1) * ntiles + tile) * (PT * 128) + (size_t)wv * (PT * 32) + l * 4;
                u32x4 pv[BH * 4];
#pragma unroll
                for (int k = 0; k < BH * 4; ++k) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(pv[k]) : "v"(part + k * 256) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < BH * 4; ++k) {
                    landed16(pv[k]);
                    const int i = k >> 2, q = k & 3;
                    acc[i][4 * q] += __uint_as_float(pv[k].x); acc[i][4 * q + 1] += __uint_as_float(pv[k].y);
                    acc[i][4 * q + 2] += __uint_as_float(pv[k].z); acc[i][4 * q + 3] += __uint_as_float(pv[k].w);
                }
                if (l == 0) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if constexpr ((ABL & 4) != 0) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) v += acc[i][r];
        if (v == 123.456f) reinterpret_cast<float*>(a.y)[t] = v;
        return;
    }

#ifdef MI_PW_TIMING
    auto ts_out = [&]() {
        if constexpr (TIMED) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the tile's stores have left
            tv[4] = pw_wlane((uint32_t)tprev, 1, tv[4]);
            tv[4] = pw_wlane((uint32_t)__builtin_amdgcn_s_memtime(), 2, tv[4]);
            tv[4] = pw_wlane((uint32_t)__builtin_amdgcn_s_getreg(63492), 5, tv[4]);
            tv[4] = pw_wlane((uint32_t)__builtin_amdgcn_s_getreg(63508), 6, tv[4]);
            tv[4] = pw_wlane((uint32_t)wall_clock64(), 8, tv[4]);
            const unsigned blk = blockIdx.y * gridDim.x + blockIdx.x;
            if (blk < 1024) {
#pragma unroll
                for (int k = 0; k < 5; ++k) g_pw_ts[((blk * 4 + wv) * 5 + k) * 64 + l] = tv[k];
            }
        }
    };
#else
    auto ts_out = [&]() {};
#endif
    // ---- epilogue.  A lane holds 4 consecutive channels of ONE pixel per register quad: stored from here a store instruction would
    //      write 16-byte pieces of 32 different rows (a [131072][128] bf16 tensor written that way takes 13 us against 7 us in whole
    //      rows, tools/proto/store_probe.hip).  The fp32 tile goes through LDS instead (the whole 80 KB is free now): [128 pixels][128
    //      channels] fp32, 16-byte chunk c of pixel p at position c ^ (p & 31) (conflict-free both ways), and leaves as whole rows --
    //      16 lanes x 8 channels = one pixel's 256 (bf16) / 512 (fp32) contiguous bytes; bias, residual and the accumulate operand are
    //      applied on the way out, read in the same row-contiguous pattern.
    auto pw_gns_out = [&](float (&gs)[2][2]) {
#if MI_PW_GNSABL == 1      // profiling build: the per-element sums only (no reduction, no atomics)
        if (gs[0][0] + gs[0][1] + gs[1][0] + gs[1][1] == 123.456f) a.gsum[t] = 1.f;
        return;
#endif
        // a 16-channel slab = two neighbouring lanes (j = 2q, 2q + 1) x the 16 row groups t >> 4 (4 per wave): lanes, then waves
        // through the LDS behind the tile, then ONE atomic pair per slab, image and workgroup (rows 0-63 / 64-127 are the two images
        // of a TI == 2 tile; with TI == 1 both halves belong to image img0)
        float r[4] = {gs[0][0], gs[0][1], gs[1][0], gs[1][1]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r[k] += __shfl_xor(r[k], 1, 64); r[k] += __shfl_xor(r[k], 16, 64); r[k] += __shfl_xor(r[k], 32, 64);
        }
#if MI_PW_GNSW == 1        // A/B build: every wave adds its own partial sums (no LDS, no barrier; four times the atomics)
        if ((l & 0x31) == 0) {
            const int slab = (n0 >> 4) + (l >> 1);
            if (slab * 16 < a.Nc) {
                if (a.TI > 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) gsum_add(a.gsum, ((size_t)(img0 + (k >> 1)) * (a.Nc >> 4) + slab) * 2 + (k & 1), r[k]);
                } else {
                    gsum_add(a.gsum, ((size_t)img0 * (a.Nc >> 4) + slab) * 2, r[0] + r[2]);
                    gsum_add(a.gsum, ((size_t)img0 * (a.Nc >> 4) + slab) * 2 + 1, r[1] + r[3]);
                }
            }
        }
        return;
#endif
        float* red = reinterpret_cast<float*>(lds_raw + PT * 512);        // behind the tile: [4 waves][8 slabs][4]
        if ((l & 0x31) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[(wv * 8 + (l >> 1)) * 4 + k] = r[k];
        }
        // (round 6: NOT __syncthreads() -- its vmcnt(0) made every wave wait out the tile's global stores, 2 - 3 us of a level-0 launch;
        //  only the four LDS writes above have to be visible)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t < 32) {
            const int q = t >> 2, k = t & 3;                 // slab, (image half, sum / sum of squares)
            const float v = red[(0 * 8 + q) * 4 + k] + red[(1 * 8 + q) * 4 + k] + red[(2 * 8 + q) * 4 + k] + red[(3 * 8 + q) * 4 + k];
            const int slab = (n0 >> 4) + q;
            if (slab * 16 < a.Nc) {
                const int img = a.TI > 1 ? img0 + (k >> 1) : img0;
                gsum_add(a.gsum, ((size_t)img * (a.Nc >> 4) + slab) * 2 + (k & 1), v);
            }
        }
        };
    __builtin_amdgcn_s_barrier();                            // every wave is done with the activation buffers, every DMA has landed
    asm volatile("" ::: "memory");
    MI_PW_NOW(13);
    // Round 4: a bf16 output with nothing to add on the way out (no residual, no accumulate -- every forward Block conv and the data
    // gradients into block-internal tensors) takes its bias in registers and crosses LDS as bf16: half the tile (32 KB), one
    // ds_read_b128 per thread and pixel, no arithmetic between the read and the store.  8-byte slot c (4 channels) of pixel p at
    // c ^ ((p & 15) << 1): an even XOR keeps a thread's two slots an aligned pair; rows p and p + 16 share banks (2-way on the writes only).
    const bool tile16 = OUT16 && (!FUSE || FPIPE) && !a.res && !a.accumulate;
    if (tile16) {
        if constexpr (OUT16) {
            typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
            typedef __attribute__((address_space(3))) u32x4 lds_u32x4e;
            if (live) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int ck = 8 * wv + 2 * rq + (l >> 5);
                    f32x4 bq;
                    if constexpr (PIPE_) { typedef __attribute__((address_space(3))) f32x4 lds_f32x4b; bq = *(lds_f32x4b*)(uintptr_t)(lds0 + BIASL + 16 * ck); }
                    else bq = bias_q[rq];
#pragma unroll
                    for (int i = 0; i < BH; ++i) {
                        const int p = ep_p0 + i * a.W;
                        *(lds_u32x2*)(uintptr_t)(lds0 + p * 256 + ((ck ^ ((p & 15) << 1)) << 3)) =
                            u32x2{pack_bf16(acc[i][4 * rq] + bq.x, acc[i][4 * rq + 1] + bq.y), pack_bf16(acc[i][4 * rq + 2] + bq.z, acc[i][4 * rq + 3] + bq.w)};
                    }
                }
            }
            __syncthreads();
            MI_PW_NOW(14);
            const int j = t & 15, col = n0 + 8 * j;
            float gs[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
            if (col < a.Nc) {
#pragma unroll
                for (int it = 0; it < PT / 16; ++it) {
                    const int p = it * 16 + (t >> 4);
                    const u32x4 o = *(lds_u32x4e*)(uintptr_t)(lds0 + p * 256 + (((2 * j) ^ ((p & 15) << 1)) << 3));
                    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(a.y) + ((size_t)m0 + p) * a.ldy + col) = o;
                    if constexpr (GNS && MI_PW_GNSABL != 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float e0 = __uint_as_float(o[q] << 16), e1 = __uint_as_float(o[q] & 0xffff0000u);
                            gs[PT >= 256 ? 0 : it >> 2][0] += e0 + e1; gs[PT >= 256 ? 0 : it >> 2][1] += e0 * e0 + e1 * e1;
                        }
                    }
                }
            }
            if constexpr (GNS) pw_gns_out(gs);
        }
        MI_PW_NOW(15);
        ts_out();
        return;
    }
    if (live) {
        typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
#pragma unroll
        for (int i = 0; i < BH; ++i) {
            const int p = ep_p0 + i * a.W;               // output row i of this lane's band
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int ck = 8 * wv + 2 * rq + (l >> 5);
                *(lds_f32x4*)(uintptr_t)(lds0 + p * 512 + ((ck ^ (p & 31)) << 4)) =
                    f32x4{acc[i][4 * rq], acc[i][4 * rq + 1], acc[i][4 * rq + 2], acc[i][4 * rq + 3]};
            }
        }
    }
    __syncthreads();
    MI_PW_NOW(14);
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    const int j = t & 15, col = n0 + 8 * j;                  // this thread's 8 channels
    float gs[2][2] = {{0.f, 0.f}, {0.f, 0.f}};               // VAR 1: [image of the tile][sum, sum of squares]
    if (col < a.Nc) {
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (a.bias) { b0 = *reinterpret_cast<const f32x4*>(a.bias + col); b1 = *reinterpret_cast<const f32x4*>(a.bias + col + 4); }
#pragma unroll
        for (int it = 0; it < PT / 16; ++it) {
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
                const u32x4 o = u32x4{pack_bf16(v0.x, v0.y), pack_bf16(v0.z, v0.w), pack_bf16(v1.x, v1.y), pack_bf16(v1.z, v1.w)};
                *reinterpret_cast<u32x4*>(yp) = o;
                if constexpr (GNS) {                         // statistics of what the next layer will read: the rounded values
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float e0 = __uint_as_float(o[q] << 16), e1 = __uint_as_float(o[q] & 0xffff0000u);
                        gs[PT >= 256 ? 0 : it >> 2][0] += e0 + e1; gs[PT >= 256 ? 0 : it >> 2][1] += e0 * e0 + e1 * e1;
                    }
                }
            } else {
                float* yp = reinterpret_cast<float*>(a.y) + m * a.ldy + col;
                if (a.accumulate) { v0 += *reinterpret_cast<const f32x4*>(yp); v1 += *reinterpret_cast<const f32x4*>(yp + 4); }
                *reinterpret_cast<f32x4*>(yp) = v0;
                *reinterpret_cast<f32x4*>(yp + 4) = v1;
                if constexpr (GNS) {
                    gs[PT >= 256 ? 0 : it >> 2][0] += (v0.x + v0.y) + (v0.z + v0.w) + (v1.x + v1.y) + (v1.z + v1.w);
                    gs[PT >= 256 ? 0 : it >> 2][1] += (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w) + (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w);
                }
            }
        }
    }
    if constexpr (GNS) pw_gns_out(gs);
    MI_PW_NOW(15);
    ts_out();
}

bool pw_geom(const MiConvDesc* d, int pt, int* TH, int* TI) {
    const int W = d->OW, H = d->OH;
    if (W != 8 && W != 16 && W != 32) return false;          // 32-pixel MFMA blocks must be whole image rows
    const int rows = pt / W, bh = pt / 32;
    if (rows <= H) { if (H % rows) return false; *TH = rows; *TI = 1; }
    else { if (rows % H) return false; *TH = H; *TI = rows / H; if ((long)d->N % *TI) return false; }
    if ((*TH & (*TH - 1)) || *TH % bh || *TI > 2) return false; // the kernel's index arithmetic: shifts, bands of bh rows, at most two images per tile
    if (pt == 256 && *TI > 1) return false;                     // (the 256-pixel tile's GroupNorm-sums epilogue attributes every row to ONE image)
    return *TI * (*TH + 2) * W <= pw_xp(pt);
}

bool pw_ok(const MiConvDesc* d, int pt, int* TH, int* TI, bool in32 = false, bool f32 = false) {
    if (d->KH != 3 || d->KW != 3 || d->pad != 1 || d->stride != 1 || d->mode != (f32 ? 0 : 1)) return false;
    if (d->IH != d->OH || d->IW != d->OW) return false;
    const int lda = (in32 || f32) ? 4 : 8;                   // 16-byte pieces of a pixel row
    const int ck = f32 ? 32 : 64;                            // channels per chunk
    // (K % 64, Nc % 64: the layers mi_pack_weights_bf16 / _f32frag write fragment-order copies for)
    if (d->K % 64 || d->K1 % ck || d->Nc % 64 || d->ldx % lda || (d->K1 != d->K && d->ldx2 % lda)) return false;
    if (((long)d->N * d->OH * d->OW) % pt) return false;
    if ((long)d->Nc * d->K * (f32 ? 4 : 2) * 9 >= (1L << 31)) return false;      // 32-bit fragment offsets
    return pw_geom(d, pt, TH, TI);
}
// 64-pixel tiles where 128-pixel ones would leave CUs without a workgroup (and the geometry allows them)
int g_pw_force_tile = 0;                 // tests: 0 = the rule below, 64 / 128 / 256 = that tile (or unsupported)
int g_pw_auto256 = 0, g_pw_min256 = 256;   // (measured slower than two 128-pixel workgroups per CU on every cfg-2 shape: off)  // the automatic pick takes 256-pixel tiles when they give at least g_pw_min256 workgroups
constexpr bool pw_fpipe(bool in32) { return (MI_PW_PIPE != 0) && (((MI_PW_FPIPE) >> (in32 ? 1 : 0)) & 1) != 0; }   // conv_pw_kernel's FPIPE
// split-K workspace of conv_pw_kernel, per device: [flags: 64 KB of ints, zero when no launch is in flight][fp32 partial tiles].  The caller owns the
// memory (zeroed once, alive as long as any captured graph may replay); launches that use it must be ordered against each other (one stream).
std::atomic<void*> g_pw_sk_ws[64];
std::atomic<size_t> g_pw_sk_bytes[64];
int g_pw_sk_mode = 1;                    // 0 off, 1 the rule below, 2 / 4: that split wherever the geometry allows (tests)
constexpr size_t PW_SK_FLAG_BYTES = 64 * 1024;
// Split-K: launches that leave CUs without a workgroup (the sampler at B = 64, cfg 3 at B = 32: the 16x16 / 8x8 levels).  Their 64-pixel tiles are
// bound by the weight bytes a CU gets from L2 (two MFMAs per fragment); 128-pixel tiles whose contraction is cut in 2 or 4 move half of those bytes
// for the same number of workgroups, and the fp32 partial tiles cross L2 once.  -> the split (1: none) and the tile it runs on
int pw_split(const MiConvDesc* d, int var, bool in32, int* pt, int* TH, int* TI) {
    if (!g_pw_sk_mode) return 1;
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!g_pw_sk_ws[dev_ & 63].load(std::memory_order_acquire)) return 1;
    int cand = g_pw_force_tile;
    if (cand == 256) return 1;
    if (!cand) cand = (pw_ok(d, 128, TH, TI, in32) && (var < 2 || *TI == 1)) ? 128 : 64;
    if (!pw_ok(d, cand, TH, TI, in32) || (var >= 2 && *TI != 1)) return 1;
    const long tiles = ((long)d->N * d->OH * d->OW / cand) * ((d->Nc + 127) / 128);
    const int nch = d->K / 64;
    // the rule (tools/bench_splitk.py, B = 64 / 32, microseconds per launch): a split pays where the contraction is long and the tiles are few --
    // 8x8 x 1024 -> 512: 45.6 -> 38.1 (two slices of 128 tiles), 39.1 -> 26.8 (four slices of 64 tiles).  8x8 x 512 -> 512: 23.4 -> 22.6 at 128
    // tiles, 17.0 -> 18.2 at 64; every 16x16 layer loses (256 -> 256: 18.4 -> 22.1): the exchange costs ~4 us and without it (wrong results,
    // -DMI_PW_SK_ABL) the 512 -> 512 layer would only reach 18.7 -- those launches are not short of workgroups, they are short of work per weight byte.
    int ks = 1;
    if (g_pw_sk_mode >= 2) ks = g_pw_sk_mode;
    else if (tiles < 200 && nch >= 16) ks = tiles * 2 >= 256 ? 2 : 4;
    while (ks > 1 && (nch % ks || nch / ks < 2)) ks >>= 1;
    if (ks <= 1) return 1;
    const size_t need = PW_SK_FLAG_BYTES + (size_t)(ks - 1) * tiles * cand * 128 * 4;
    if ((size_t)(ks - 1) * tiles * 4 * sizeof(int) > PW_SK_FLAG_BYTES || need > g_pw_sk_bytes[dev_ & 63].load(std::memory_order_acquire)) return 1;
    *pt = cand;
    return ks;
}
int pw_pick_tile(const MiConvDesc* d, int var, int* TH, int* TI, bool in32 = false, bool f32 = false, int* ksplit = nullptr) {
    if (ksplit) *ksplit = 1;
    if (f32 && var != 0) return 0;
    if (var >= 2 && pw_fpipe(in32) && d->K > 1024) return 0;   // the coefficient table of the pinned fused loop: 12 K bytes of LDS
    if (!f32 && ksplit) {
        int pts = 0;
        const int ks = pw_split(d, var, in32, &pts, TH, TI);
        if (ks > 1) { *ksplit = ks; return pts; }
    }
    if (g_pw_force_tile == 64) return pw_ok(d, 64, TH, TI, in32, f32) ? 64 : 0;
    if (g_pw_force_tile == 128) return pw_ok(d, 128, TH, TI, in32, f32) ? 128 : 0;
    if (g_pw_force_tile == 256) return (var < 2 && !in32 && !f32 && pw_ok(d, 256, TH, TI)) ? 256 : 0;
    const long t128 = ((long)d->N * d->OH * d->OW / 128) * ((d->Nc + 127) / 128);
    // 256-pixel tiles (one workgroup per CU, a weight fragment feeds eight MFMAs instead of four) where they still fill the chip
    if (g_pw_auto256 && var < 2 && !in32 && !f32 && t128 >= 2 * g_pw_min256 && pw_ok(d, 256, TH, TI)) return 256;
    if (t128 < 200 && pw_ok(d, 64, TH, TI, in32, f32) && (var < 2 || *TI == 1)) return 64;
    return pw_ok(d, 128, TH, TI, in32, f32) ? 128 : 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// 1x1 convolution on the same machinery (to_qkv, to_out + residual, res_conv -- also over the skip concat's two sources -- and their
// data gradients: reference src/models/ddpm.py:134,151-152).  These are streaming kernels (64-250 FLOP per byte): a workgroup =
// PXT (128 or 64) pixels x 128 output channels, wave w owns all pixels of channels [32w, 32w + 32).  Per 128-channel chunk the
// PXT x 128-channel activation tile (two swizzled 64-channel halves) arrives by LDS-DMA into one of two buffers while the previous
// chunk is multiplied, and each wave loads the 8 weight fragments of its 32 channels straight into registers (two sets); one
// barrier per chunk (8 x PXT / 32 MFMAs per wave).  Then the row-contiguous epilogue through LDS (bias, fp32 residual, accumulate;
// optional bf16 copy of an fp32 output: to_out's epilogue writes the residual-stream tensor AND the copy its consumers read).
// 64 KB of LDS: two workgroups per CU.
// LDS of a launch: two chunk tiles (PXT x 256 bytes each) under the epilogue's fp32 tile (PXT x 512 bytes).  (Round 5: the 64-pixel tiles asked
// for the 128-pixel tiles' 64 KB -- two workgroups per CU whatever their registers allowed; with 32 KB the 126-168-register instantiations
// run three or four.)
constexpr size_t pw1_lds(int pxt) { return (size_t)pxt * 512; }
struct Pw1Args {
    const uint16_t* x; const uint16_t* x2; const uint16_t* w; const float* bias; const float* res; void* y; uint16_t* y16;
    int M, K, K1, Nc, ldx, ldx2, ldy, ldr, ldy16, accumulate, gx, gy;
    int wbatch, hw;      // per-sample weights (mi_conv1x1_pw_batched): the fragments of the tile's sample start wbatch elements x (m0 / hw) behind w
};

// F32: exact-fp32 mode (see conv_pw_kernel): fp32 x / y, fp32 fragment-order weights, v_mfma_f32_32x32x2_f32; a chunk is 64 channels.
// IN32 (bf16 MFMA mode): x / x2 are fp32 tensors (the residual-stream gradient of to_out / res_conv's data gradients, res_conv's input in
// inference): a chunk's pieces are loaded into registers during the previous chunk, rounded to bf16 once and written to the lane's slot
// of the tile after that chunk's MFMAs (see conv_pw_kernel's IN32).
// NLOOP (round 5; bf16 in / bf16 out, K = 128, nothing added on the way out: to_qkv at the 128-channel level): ONE workgroup per pixel tile
// walks all channel tiles.  The activation tile is staged once (the 2-D grid had every channel tile's workgroup fetch it again from L2 and
// sit through its own load -> MFMA -> store chain: 3.9 TB/s of algorithmic traffic at level 0 against 6.1 for to_out), the next channel
// tile's fragments are requested under the current tile's MFMAs, and the epilogue stages through the second (otherwise idle) buffer.
template <bool OUT16, bool DUAL, int PXT, bool F32 = false, bool IN32 = false, bool NLOOP = false>
__global__ __launch_bounds__(256, 2) void conv1x1_pw_kernel(const Pw1Args a) {
    MI_PRIO_UP();
    static_assert(!F32 || (!OUT16 && !DUAL), "exact-fp32 mode writes fp32");
    static_assert(!IN32 || !F32, "fp32 input of the bf16 MFMA mode");
    static_assert(!NLOOP || (OUT16 && !DUAL && !F32 && !IN32 && PXT == 128), "channel-tile loop: the bf16 -> bf16 single-chunk form");
    constexpr int ESZ = F32 ? 4 : 2, EPP = 16 / ESZ, CKC = 16 * EPP;       // element size, elements per 16-byte piece, channels per chunk
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;
    constexpr int NBLK = PXT / 32;                   // 32-pixel MFMA blocks per wave
    constexpr int XH = PXT * 128;                    // one 64-channel half of a chunk's tile (bytes)
    constexpr int XB = 2 * XH;                       // one chunk's tile
    constexpr int NPC = 2 * PXT / 8 / 4;             // activation pieces (8 pixels x 64 channels = 1 KB) per wave and chunk
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    // the channel tiles of one pixel tile are adjacent in time on one XCD (ids xcd + 8*slot): the tile is read from HBM once
    int bx = blockIdx.x, by = blockIdx.y;
    if (!NLOOP && a.gy > 1 && a.gx % 8 == 0) {
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        bx = xcd * (a.gx >> 3) + slot / a.gy; by = slot % a.gy;
    }
    if constexpr (NLOOP) by = 0;
    const int m0 = bx * PXT;
    int n0 = by * 128;
    const int NB = a.Nc >> 5, KQ = a.K / (2 * EPP), nchunks = a.K / CKC;
    bool live = n0 + 32 * wv < a.Nc;
    int nb = min((n0 >> 5) + wv, NB - 1);

    // activation pieces of chunk ch -> buffer ch & 1: piece p = wv + 4i = pixels 8 (p % (PXT / 8)) .. +7 of half p / (PXT / 8); lane ->
    // pixel l >> 3, stored 16-byte position l & 7 holds channel chunk (l & 7) ^ ((pixel >> 1) & 7)
    auto stage_x = [&](int ch) {
        const int c0 = min(ch, nchunks - 1) * CKC;
        const bool second = c0 >= a.K1;
        const uint8_t* src = reinterpret_cast<const uint8_t*>(second ? a.x2 : a.x);
        const int ld = second ? a.ldx2 : a.ldx, cc = second ? c0 - a.K1 : c0;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int p = wv + 4 * i, half = p / (PXT / 8), px = 8 * (p % (PXT / 8)) + (l >> 3);
            const int col = cc + half * (CKC / 2) + ((l & 7) ^ ((px >> 1) & 7)) * EPP;
            glds16(src + ((size_t)(m0 + px) * ld + col) * ESZ, lds0 + (ch & 1) * XB + half * XH + (p % (PXT / 8)) * 1024);
        }
    };
    // fp32 input: the same pieces through registers (8 channels = 32 bytes per lane and piece)
    u32x4 XR[IN32 ? NPC : 1][2];
    auto load_x32 = [&](int ch) {
        const int c0 = min(ch, nchunks - 1) * CKC;
        const bool second = c0 >= a.K1;
        const float* src = reinterpret_cast<const float*>(second ? a.x2 : a.x);
        const int ld = second ? a.ldx2 : a.ldx, cc = second ? c0 - a.K1 : c0;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int p = wv + 4 * i, half = p / (PXT / 8), px = 8 * (p % (PXT / 8)) + (l >> 3);
            const float* pf = src + (size_t)(m0 + px) * ld + cc + half * 64 + ((l & 7) ^ ((px >> 1) & 7)) * 8;
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                         : "=&v"(XR[i][0]), "=&v"(XR[i][1]) : "v"(pf) : "memory");
        }
    };
    auto store_x32 = [&](int ch) {                   // ... rounded and written to buffer ch & 1 (behind the wait that covers the loads)
        typedef __attribute__((address_space(3))) u32x4 lds_u32x4_;
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int p = wv + 4 * i, half = p / (PXT / 8);
            landed16(XR[i][0]); landed16(XR[i][1]);
            const u32x4 o = {pack_bf16(__uint_as_float(XR[i][0].x), __uint_as_float(XR[i][0].y)), pack_bf16(__uint_as_float(XR[i][0].z), __uint_as_float(XR[i][0].w)),
                             pack_bf16(__uint_as_float(XR[i][1].x), __uint_as_float(XR[i][1].y)), pack_bf16(__uint_as_float(XR[i][1].z), __uint_as_float(XR[i][1].w))};
            *(lds_u32x4_*)(uintptr_t)(lds0 + (ch & 1) * XB + half * XH + (p % (PXT / 8)) * 1024 + l * 16) = o;
        }
    };
    // weight fragments (nb, kq = 8 ch .. 8 ch + 7): 8 KB contiguous
    uint64_t wsrc;
    {
        const size_t wb = a.wbatch ? (size_t)(m0 / a.hw) * a.wbatch * ESZ : 0;
        const uint64_t q = (uint64_t)(uintptr_t)(reinterpret_cast<const uint8_t*>(a.w) + wb + (size_t)nb * KQ * 1024);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)q), hi = __builtin_amdgcn_readfirstlane((uint32_t)(q >> 32));
        wsrc = ((uint64_t)hi << 32) | lo;
    }
    const uint32_t wl16 = l * 16;
    u32x4 WB[2][8];
    auto load_w = [&](int ch, auto setc) {
        constexpr int set = decltype(setc)::value;
        const uint32_t voff = wl16 + (uint32_t)min(ch, nchunks - 1) * 8192;
        static_for<0, 4>([&](auto uc) { gload16s<decltype(uc)::value * 1024>(WB[set][decltype(uc)::value], wsrc, voff); });
        static_for<0, 4>([&](auto uc) { gload16s<decltype(uc)::value * 1024>(WB[set][4 + decltype(uc)::value], wsrc, voff + 4096); });
    };
    uint32_t xa[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const int px = i * 32 + (l & 31);
        xa[i] = lds0 + px * 128 + (((l >> 5) * 16) ^ (((px >> 1) & 7) * 16));
    }
    f32x16 acc[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    f32x4 bias_q[OUT16 ? 4 : 1];                     // bf16 output: the bias of this lane's channel quads, requested first (see the epilogue)
    if constexpr (OUT16) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            bias_q[rq] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias && live) bias_q[rq] = *reinterpret_cast<const f32x4*>(a.bias + n0 + 4 * (8 * wv + 2 * rq + (l >> 5)));
        }
    }
    if constexpr (NLOOP) {
        typedef __attribute__((address_space(3))) u32x2 lds_u32x2n;
        typedef __attribute__((address_space(3))) u32x4 lds_u32x4n;
        // fragments of channel tile jt -> register set: base of this wave's 32 channels, one 8 KB chunk
        auto load_wj = [&](int jt, auto setc) {
            constexpr int set = decltype(setc)::value;
            const int nbj = min(4 * jt + wv, NB - 1);
            const uint64_t q = (uint64_t)(uintptr_t)(reinterpret_cast<const uint8_t*>(a.w) + (size_t)nbj * KQ * 1024);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)q), hi = __builtin_amdgcn_readfirstlane((uint32_t)(q >> 32));
            const uint64_t wj = ((uint64_t)hi << 32) | lo;
            static_for<0, 4>([&](auto uc) { gload16s<decltype(uc)::value * 1024>(WB[set][decltype(uc)::value], wj, wl16); });
            static_for<0, 4>([&](auto uc) { gload16s<decltype(uc)::value * 1024>(WB[set][4 + decltype(uc)::value], wj, wl16 + 4096); });
        };
        stage_x(0);
        load_wj(0, std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_for<0, 8>([&](auto uc) { landed16(WB[0][decltype(uc)::value]); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // the tile is every wave's
        asm volatile("" ::: "memory");
        auto cotile = [&](int jt, auto setc) {
            constexpr int set = decltype(setc)::value;
            n0 = jt * 128; live = n0 + 32 * wv < a.Nc;
            if (jt + 1 < a.gy) load_wj(jt + 1, std::integral_constant<int, set ^ 1>{});
            bf16x8 XC[2][NBLK];
#pragma unroll
            for (int i = 0; i < NBLK; ++i) XC[0][i] = lds_b128p(xa[i]);
            static_for<0, 8>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if constexpr (u + 1 < 8) {
                    constexpr int un = u + 1;
#pragma unroll
                    for (int i = 0; i < NBLK; ++i) XC[un & 1][i] = lds_b128p((xa[i] ^ ((un & 3) * 32)) + (un >> 2) * XH);
                }
#pragma unroll
                for (int i = 0; i < NBLK; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WB[set][u]), XC[u & 1][i], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            // this channel tile out through the second buffer (the first one keeps the activation tile); the stores of the previous
            // channel tile were read out of it before the barrier that closed that epilogue
            if (live) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int ck = 8 * wv + 2 * rq + (l >> 5);
                    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) bq = *reinterpret_cast<const f32x4*>(a.bias + n0 + 4 * ck);
#pragma unroll
                    for (int i = 0; i < NBLK; ++i) {
                        const int p = i * 32 + (l & 31);
                        *(lds_u32x2n*)(uintptr_t)(lds0 + XB + p * 256 + ((ck ^ ((p & 15) << 1)) << 3)) =
                            u32x2{pack_bf16(acc[i][4 * rq] + bq.x, acc[i][4 * rq + 1] + bq.y), pack_bf16(acc[i][4 * rq + 2] + bq.z, acc[i][4 * rq + 3] + bq.w)};
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NBLK; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            // the next channel tile's fragments (requested 32 MFMAs ago) are waited for HERE, before this tile's stores are issued: vmcnt
            // counts stores too, and a wait behind them would put a write round trip between two channel tiles
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            static_for<0, 8>([&](auto uc) { landed16(WB[set ^ 1][decltype(uc)::value]); });
            __syncthreads();
            {
                const int j = t & 15, col = n0 + 8 * j;
                if (col < a.Nc) {
#pragma unroll
                    for (int it = 0; it < PXT / 16; ++it) {
                        const int p = it * 16 + (t >> 4);
                        const u32x4 o = *(lds_u32x4n*)(uintptr_t)(lds0 + XB + p * 256 + (((2 * j) ^ ((p & 15) << 1)) << 3));
                        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(a.y) + ((size_t)m0 + p) * a.ldy + col) = o;
                    }
                }
            }
            __syncthreads();                                   // the staging area is free again (every lane holds what it read)
        };
        for (int jt = 0; jt < a.gy; jt += 2) {
            cotile(jt, std::integral_constant<int, 0>{});
            if (jt + 1 < a.gy) cotile(jt + 1, std::integral_constant<int, 1>{});
        }
        return;
    }
    if constexpr (IN32) load_x32(0); else stage_x(0);
    load_w(0, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<0, 8>([&](auto uc) { landed16(WB[0][decltype(uc)::value]); });
    if constexpr (IN32) store_x32(0);
    auto chunk = [&](int ch, auto setc) {
        constexpr int set = decltype(setc)::value;
        // this chunk's tile and fragments have landed (this wave's requests were waited for at the end of the previous chunk -- no
        // register-destination load is in flight across the loop's back edge, DESIGN.md "the sixth rule"); the barrier makes it every
        // wave's pieces, and the other buffer and register set are free
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ch + 1 < nchunks) {
            if constexpr (IN32) load_x32(ch + 1); else stage_x(ch + 1);
            load_w(ch + 1, std::integral_constant<int, set ^ 1>{});
        }
        const uint32_t xb = (ch & 1) * XB;
        bf16x8 XC[2][NBLK];
#pragma unroll
        for (int i = 0; i < NBLK; ++i) XC[0][i] = lds_b128p(xa[i] + xb);
        static_for<0, 8>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (u + 1 < 8) {
                constexpr int un = u + 1;
#pragma unroll
                for (int i = 0; i < NBLK; ++i) XC[un & 1][i] = lds_b128p((xa[i] ^ ((un & 3) * 32)) + (un >> 2) * XH + xb);
            }
#pragma unroll
            for (int i = 0; i < NBLK; ++i) {
                if constexpr (F32) {
                    const f32x4 wv4 = __builtin_bit_cast(f32x4, WB[set][u]), xv4 = __builtin_bit_cast(f32x4, XC[u & 1][i]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv4[j], xv4[j], acc[i], 0, 0, 0);
                } else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WB[set][u]), XC[u & 1][i], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the next chunk's requests (issued 32+ MFMAs ago)
        static_for<0, 8>([&](auto uc) { landed16(WB[set ^ 1][decltype(uc)::value]); });
        if constexpr (IN32) { if (ch + 1 < nchunks) store_x32(ch + 1); }     // into the buffer every wave stopped reading a chunk ago
    };
    for (int ch = 0; ch < nchunks; ch += 2) {
        chunk(ch, std::integral_constant<int, 0>{});
        if (ch + 1 < nchunks) chunk(ch + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: as conv_pw_kernel's (fp32 tile through LDS, whole rows out; a bf16 output with nothing to add on the way out
    //      crosses LDS as bf16, the bias taken in registers -- to_qkv and the other bf16-output 1x1 convs are all prologue and epilogue)
    __syncthreads();
    if constexpr (OUT16) {
        if (!a.res && !a.accumulate) {
            typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
            typedef __attribute__((address_space(3))) u32x4 lds_u32x4e;
            if (live) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int ck = 8 * wv + 2 * rq + (l >> 5);
                    const f32x4 bq = bias_q[rq];
#pragma unroll
                    for (int i = 0; i < NBLK; ++i) {
                        const int p = i * 32 + (l & 31);
                        *(lds_u32x2*)(uintptr_t)(lds0 + p * 256 + ((ck ^ ((p & 15) << 1)) << 3)) =
                            u32x2{pack_bf16(acc[i][4 * rq] + bq.x, acc[i][4 * rq + 1] + bq.y), pack_bf16(acc[i][4 * rq + 2] + bq.z, acc[i][4 * rq + 3] + bq.w)};
                    }
                }
            }
            __syncthreads();
            const int j = t & 15, col = n0 + 8 * j;
            if (col >= a.Nc) return;
#pragma unroll
            for (int it = 0; it < PXT / 16; ++it) {
                const int p = it * 16 + (t >> 4);
                const u32x4 o = *(lds_u32x4e*)(uintptr_t)(lds0 + p * 256 + (((2 * j) ^ ((p & 15) << 1)) << 3));
                *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(a.y) + ((size_t)m0 + p) * a.ldy + col) = o;
            }
            return;
        }
    }
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    if (live) {
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
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
    const int j = t & 15, col = n0 + 8 * j;
    if (col >= a.Nc) return;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (a.bias) { b0 = *reinterpret_cast<const f32x4*>(a.bias + col); b1 = *reinterpret_cast<const f32x4*>(a.bias + col + 4); }
#pragma unroll
    for (int it = 0; it < PXT / 16; ++it) {
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
                v0 += f32x4{__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u), __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
                v1 += f32x4{__uint_as_float(o.z << 16), __uint_as_float(o.z & 0xffff0000u), __uint_as_float(o.w << 16), __uint_as_float(o.w & 0xffff0000u)};
            }
            *reinterpret_cast<u32x4*>(yp) = u32x4{pack_bf16(v0.x, v0.y), pack_bf16(v0.z, v0.w), pack_bf16(v1.x, v1.y), pack_bf16(v1.z, v1.w)};
        } else {
            float* yp = reinterpret_cast<float*>(a.y) + m * a.ldy + col;
            if (a.accumulate) { v0 += *reinterpret_cast<const f32x4*>(yp); v1 += *reinterpret_cast<const f32x4*>(yp + 4); }
            *reinterpret_cast<f32x4*>(yp) = v0;
            *reinterpret_cast<f32x4*>(yp + 4) = v1;
            if constexpr (DUAL)
                *reinterpret_cast<u32x4*>(a.y16 + m * a.ldy16 + col) =
                    u32x4{pack_bf16(v0.x, v0.y), pack_bf16(v0.z, v0.w), pack_bf16(v1.x, v1.y), pack_bf16(v1.z, v1.w)};
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Inference: channel LayerNorm applied while to_qkv's input is staged (PreNorm(LinearAttention / Attention), reference
// src/models/ddpm.py:85-106,151: y = (x - mean) / (sqrt(var) + eps) * g + b over the channels of a pixel, then the bias-free 1x1 conv).  The
// normalised tensor is never written: one workgroup per PXT-pixel tile reads the fp32 residual stream once (every lane its pixels' channel octets --
// NPB pixel blocks x NCH 128-channel chunks x two halves, all in registers), forms the two-pass mean / variance per pixel (a pixel = the 8 lanes of a
// 16-byte row of LDS positions), normalises, rounds to bf16 once and writes ITS slots of the [chunk][half][pixel][128 B] tile -- conv1x1_pw_kernel's
// layout and swizzle; then walks the channel tiles and chunks as conv1x1_pw_kernel's NLOOP form does: the wave's 8 fragments of the next
// (channel tile, chunk) unit are requested under the current unit's MFMAs, a channel tile leaves through a staging area behind the tile.
// Training (a.ln_out): to_qkv's weight gradient reads the normalised tensor -- the lane that rounds a piece for the tile also stores it.
struct LnPw1Args {
    const float* x; const uint16_t* w; const float* g; const float* b; const float* bias; uint16_t* y;
    int M, K, Nc, ldx, ldy, gx, gy; float eps;
    uint16_t* ln_out; int ldl;      // training: the normalised tensor as well (bf16 [M][ldl]): to_qkv's weight gradient reads it
};
constexpr size_t lnpw1_lds(int pxt, int nch) { return (size_t)nch * pxt * 256 + (size_t)pxt * 256 + (size_t)nch * 1024; }
template <int PXT, int NCH, bool G2 = false>
__global__ __launch_bounds__(256, 2) void ln_conv1x1_pw_kernel(const LnPw1Args a) {
    MI_PRIO_UP();
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    typedef __attribute__((address_space(3))) u32x2 lds_u32x2n;
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4n;
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4n;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;
    constexpr int NBLK = PXT / 32;                   // 32-pixel MFMA blocks per wave
    constexpr int NPB = PXT / 32;                    // 8-pixel blocks a wave stages (block wv + 4 j)
    constexpr int XH = PXT * 128, XB = 2 * XH;       // one 64-channel half / one 128-channel chunk of the tile
    constexpr int STG = NCH * XB;                    // the channel tile's way out: PXT x 256 bytes
    constexpr int GB = STG + PXT * 256;              // g | b: 2 x K floats
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    // G2: a workgroup per (pixel tile, channel tile) -- every channel tile normalises the pixel tile again (the small levels: their launches are
    // chains of latencies, not bytes); the channel tiles of a pixel tile are adjacent ids on one XCD: the tile comes from HBM once
    int bx = blockIdx.x, jt0 = 0, jt1 = a.gy;
    if constexpr (G2) {
        if (a.gx % 8 == 0) { const int id = blockIdx.x, xcd = id & 7, slot = id >> 3; bx = xcd * (a.gx >> 3) + slot / a.gy; jt0 = slot % a.gy; }
        else { bx = blockIdx.x / a.gy; jt0 = blockIdx.x % a.gy; }
        jt1 = jt0 + 1;
    }
    const int m0 = bx * PXT;
    const int NB = a.Nc >> 5, KQ = a.K / 16;

    // ---- the tile: raw values -> registers
    const int cidx = (l & 7) ^ ((4 * wv + (l >> 4)) & 7);          // the channel octet of a half that belongs in this lane's slot (the same for every block)
    f32x4 XR[NPB][NCH][2][2];
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        const float* px = a.x + (size_t)(m0 + 8 * (wv + 4 * j) + (l >> 3)) * a.ldx + cidx * 8;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                XR[j][ch][h][0] = *reinterpret_cast<const f32x4*>(px + ch * 128 + h * 64);
                XR[j][ch][h][1] = *reinterpret_cast<const f32x4*>(px + ch * 128 + h * 64 + 4);
            }
    }
    // the affine parameters -> LDS (K <= 512: one float4 per thread and vector is enough for two of them)
    for (int q = t; q < a.K / 4; q += 256) {
        *(lds_f32x4n*)(uintptr_t)(lds0 + GB + q * 16) = *reinterpret_cast<const f32x4*>(a.g + 4 * q);
        *(lds_f32x4n*)(uintptr_t)(lds0 + GB + a.K * 4 + q * 16) = *reinterpret_cast<const f32x4*>(a.b + 4 * q);
    }
    // the first unit's fragments
    const uint32_t wl16 = l * 16;
    u32x4 WB[2][8];
    auto load_wu = [&](int jt, int ch, auto setc) {
        constexpr int set = decltype(setc)::value;
        const int nbj = min(4 * jt + wv, NB - 1);
        const uint64_t q = (uint64_t)(uintptr_t)(reinterpret_cast<const uint8_t*>(a.w) + ((size_t)nbj * KQ + 8 * ch) * 1024);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)q), hi = __builtin_amdgcn_readfirstlane((uint32_t)(q >> 32));
        const uint64_t wj = ((uint64_t)hi << 32) | lo;
        static_for<0, 4>([&](auto uc) { gload16s<decltype(uc)::value * 1024>(WB[set][decltype(uc)::value], wj, wl16); });
        static_for<0, 4>([&](auto uc) { gload16s<decltype(uc)::value * 1024>(WB[set][4 + decltype(uc)::value], wj, wl16 + 4096); });
    };
    load_wu(jt0, 0, std::integral_constant<int, 0>{});
    __syncthreads();                                  // g | b are in LDS (the waits of the plain loads above are hipcc's)

    const float invc = 1.0f / (float)a.K;
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        float s = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int v = 0; v < 2; ++v) s += (XR[j][ch][h][v].x + XR[j][ch][h][v].y) + (XR[j][ch][h][v].z + XR[j][ch][h][v].w);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float mean = s * invc;
        float s2 = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const f32x4 d = XR[j][ch][h][v] - mean;
                    s2 += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
        s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
        const float inv = 1.0f / (sqrtf(s2 * invc) + a.eps);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t gq = lds0 + GB + (ch * 128 + h * 64 + cidx * 8) * 4;
                const f32x4 g0 = *(lds_f32x4n*)(uintptr_t)gq, g1 = *(lds_f32x4n*)(uintptr_t)(gq + 16);
                const f32x4 b0 = *(lds_f32x4n*)(uintptr_t)(gq + a.K * 4), b1 = *(lds_f32x4n*)(uintptr_t)(gq + a.K * 4 + 16);
                const f32x4 o0 = (XR[j][ch][h][0] - mean) * inv * g0 + b0, o1 = (XR[j][ch][h][1] - mean) * inv * g1 + b1;
                const u32x4 o = {pack_bf16(o0.x, o0.y), pack_bf16(o0.z, o0.w), pack_bf16(o1.x, o1.y), pack_bf16(o1.z, o1.w)};
                *(lds_u32x4n*)(uintptr_t)(lds0 + ch * XB + h * XH + (wv + 4 * j) * 1024 + l * 16) = o;
                // (the 8 lanes of a pixel write the 128 bytes of its 64-channel half: whole lines)
                if (a.ln_out && jt0 == 0)
                    *reinterpret_cast<u32x4*>(a.ln_out + (size_t)(m0 + 8 * (wv + 4 * j) + (l >> 3)) * a.ldl + ch * 128 + h * 64 + cidx * 8) = o;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<0, 8>([&](auto uc) { landed16(WB[0][decltype(uc)::value]); });
    __syncthreads();                                  // the tile is every wave's

    uint32_t xa[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const int px = i * 32 + (l & 31);
        xa[i] = lds0 + px * 128 + (((l >> 5) * 16) ^ (((px >> 1) & 7) * 16));
    }
    f32x16 acc[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // unit (jt, ch): the fragments in set (JP * NCH + ch) & 1 (JP = jt & 1), the next unit's requested into the other set
    auto unit = [&](int jt, auto chc, auto setc) {
        constexpr int ch = decltype(chc)::value, set = decltype(setc)::value;
        const bool last_unit = ch == NCH - 1 && jt + 1 >= jt1;
        if (!last_unit) {
            if constexpr (ch + 1 < NCH) load_wu(jt, ch + 1, std::integral_constant<int, set ^ 1>{});
            else load_wu(jt + 1, 0, std::integral_constant<int, set ^ 1>{});
        }
        bf16x8 XC[2][NBLK];
#pragma unroll
        for (int i = 0; i < NBLK; ++i) XC[0][i] = lds_b128p(xa[i] + ch * XB);
        static_for<0, 8>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (u + 1 < 8) {
                constexpr int un = u + 1;
#pragma unroll
                for (int i = 0; i < NBLK; ++i) XC[un & 1][i] = lds_b128p((xa[i] ^ ((un & 3) * 32)) + (un >> 2) * XH + ch * XB);
            }
#pragma unroll
            for (int i = 0; i < NBLK; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WB[set][u]), XC[u & 1][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (ch == NCH - 1) {
            const int n0 = jt * 128;
            if (n0 + 32 * wv < a.Nc) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int ck = 8 * wv + 2 * rq + (l >> 5);
                    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) bq = *reinterpret_cast<const f32x4*>(a.bias + n0 + 4 * ck);
#pragma unroll
                    for (int i = 0; i < NBLK; ++i) {
                        const int p = i * 32 + (l & 31);
                        *(lds_u32x2n*)(uintptr_t)(lds0 + STG + p * 256 + ((ck ^ ((p & 15) << 1)) << 3)) =
                            u32x2{pack_bf16(acc[i][4 * rq] + bq.x, acc[i][4 * rq + 1] + bq.y), pack_bf16(acc[i][4 * rq + 2] + bq.z, acc[i][4 * rq + 3] + bq.w)};
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NBLK; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        // the next unit's fragments are waited for before this channel tile's stores are issued (vmcnt counts stores too)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_for<0, 8>([&](auto uc) { landed16(WB[set ^ 1][decltype(uc)::value]); });
        if constexpr (ch == NCH - 1) {
            const int n0 = jt * 128;
            __syncthreads();
            {
                const int j = t & 15, col = n0 + 8 * j;
                if (col < a.Nc) {
#pragma unroll
                    for (int it = 0; it < PXT / 16; ++it) {
                        const int p = it * 16 + (t >> 4);
                        const u32x4 o = *(lds_u32x4n*)(uintptr_t)(lds0 + STG + p * 256 + (((2 * j) ^ ((p & 15) << 1)) << 3));
                        *reinterpret_cast<u32x4*>(a.y + ((size_t)m0 + p) * a.ldy + col) = o;
                    }
                }
            }
            __syncthreads();                                   // the staging area is free again
        }
    };
    auto cotile = [&](int jt, auto jpc) {
        constexpr int JP = decltype(jpc)::value;
        static_for<0, NCH>([&](auto chc) {
            constexpr int ch = decltype(chc)::value;
            unit(jt, chc, std::integral_constant<int, (JP * NCH + ch) & 1>{});
        });
    };
    for (int jt = jt0; jt < jt1; jt += 2) {
        cotile(jt, std::integral_constant<int, 0>{});
        if (jt + 1 < jt1) cotile(jt + 1, std::integral_constant<int, 1>{});
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Tap-gather GEMM on the same machinery (round 4): the stride-2 Downsample conv, the ConvTranspose2d(4, 2, 1) Upsample and their data
// gradients (reference src/models/ddpm.py:67-82), i.e. every conv that is "a few taps, each a strided view of the input":
//     Y[out(m)][co] (+)= bias[co] + sum_{taps t of the class} sum_ci X[n, oy * si + dy_t, ox * si + dx_t][ci] * W[wt_t][co][ci]
// over the pixels m = (n, oy, ox) of a class grid [N][GH][GW]; out(m) = (n, oy * so + ao, ox * so + bo) on the produced tensor.
// A strided conv is ONE class with si = stride, so = 1 and k x k taps; a transposed conv (or a strided conv's data gradient) is
// stride^2 parity classes (blockIdx.z) with si = 1, so = stride and only the taps whose parity matches the class (1, 2, 2, 4 of the 3x3
// kernel, 4 each of the 4x4 one).  Structure = conv1x1_pw_kernel: a chunk is two 64-channel halves, each half = (tap, 64-channel
// slice) -- the LDS-DMA gathers a half's pixel rows at the tap's offset (a DMA piece is a pixel's 128 bytes wherever it lies, so the
// stride costs nothing; out-of-image taps read the zero page), weights are the wave's private fragment stream in mi_pack_weights_bf16's
// fragment order, one barrier per chunk.  Replaces round 1's igemm_fast_kernel (register-staged ring, 8-16 MFMAs per barrier:
// 200-380 TFLOP/s) for these layers.
struct GtClass { unsigned long long dyq, dxq, wtq; int ntap, ao, bo, pad_; };     // 4-bit fields per tap: dy + 8, dx + 8, weight tap
struct GtArgs {
    const uint16_t* x; const uint16_t* w; const float* bias; const float* res; void* y; uint16_t* y16; int ldy16;
    int M, K, Nc, ldx, ldy, ldr, accumulate, gx, gy;
    int IH, IW, lgGW, lgGHW, si, so, OH, OW;
    GtClass cls[4];
};

// F32: exact-fp32 mode (see conv_pw_kernel): fp32 x / y, fp32 fragment-order weights (mi_pack_weights_f32frag), v_mfma_f32_32x32x2_f32; the
// byte geometry is the same -- a half is 32 channels, a fragment 8.
template <bool OUT16, int PXT, bool F32 = false>
__global__ __launch_bounds__(256, 2) void conv_gt_kernel(const GtArgs a) {
    MI_PRIO_UP();
    static_assert(!F32 || !OUT16, "exact-fp32 mode writes fp32");
    constexpr int ESZ = F32 ? 4 : 2, EPP = 16 / ESZ, HCH = 8 * EPP;      // element size, elements per 16-byte piece, channels per half
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_raw;
    constexpr int NBLK = PXT / 32;                   // 32-pixel MFMA blocks per wave = pixels a lane stages per half
    constexpr int XH = PXT * 128;                    // one 64-channel half of a chunk's tile (bytes)
    constexpr int XB = 2 * XH;
    const int t = threadIdx.x, l = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.gy > 1 && a.gx % 8 == 0) {                 // the channel tiles of one pixel tile adjacent in time on one XCD
        const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
        bx = xcd * (a.gx >> 3) + slot / a.gy; by = slot % a.gy;
    }
    const GtClass cl = a.cls[blockIdx.z];
    const int m0 = bx * PXT, n0 = by * 128;
    const int NB = a.Nc >> 5, KQ = a.K / (2 * EPP), kc64 = a.K / HCH;
    const int nhc = cl.ntap * kc64, nchunks = (nhc + 1) >> 1;
    const bool live = n0 + 32 * wv < a.Nc;
    const int nb = min((n0 >> 5) + wv, NB - 1);

    // the pixels this lane stages (one per 8-pixel block wv + 4 j of the tile, the same for both halves): image base, oy * si, ox * si
    int pbase[NBLK], iy0[NBLK], ix0[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        const int m = m0 + 8 * (wv + 4 * j) + (l >> 3);
        const int n = m >> a.lgGHW, r = m & ((1 << a.lgGHW) - 1);
        pbase[j] = n * a.IH * a.IW; iy0[j] = (r >> a.lgGW) * a.si; ix0[j] = (r & ((1 << a.lgGW) - 1)) * a.si;
    }
    const uint8_t* zero = reinterpret_cast<const uint8_t*>(g_zero_page3) + (l & 7) * 16;
    auto stage_x = [&](int ch) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int hc = min(2 * ch + h, 2 * nchunks - 1);
            const bool pad_half = hc >= nhc;                                      // odd number of halves: the last one multiplies zeros
            const int hcc = min(hc, nhc - 1), tp = hcc / kc64, c0 = (hcc - tp * kc64) * HCH;
            const int dy = (int)((cl.dyq >> (4 * tp)) & 15) - 8, dx = (int)((cl.dxq >> (4 * tp)) & 15) - 8;
#pragma unroll
            for (int j = 0; j < NBLK; ++j) {
                const int px = 8 * (wv + 4 * j) + (l >> 3);
                const int iy = iy0[j] + dy, ix = ix0[j] + dx;
                const bool ok = !pad_half && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
                size_t off = (size_t)(pbase[j] + iy * a.IW + ix) * a.ldx + c0 + ((l & 7) ^ ((px >> 1) & 7)) * EPP;
                asm volatile("" : "+v"(off));
                const uint8_t* src = ok ? reinterpret_cast<const uint8_t*>(a.x) + off * ESZ : zero;
                glds16(src, lds0 + (ch & 1) * XB + h * XH + (wv + 4 * j) * 1024);
            }
        }
    };
    // weight fragments of half-chunk (tap wt, channels c0 ..): (wt * NB + nb) * KQ + c0 / 16 .. + 3, 4 KB contiguous
    const uint8_t* wbase = reinterpret_cast<const uint8_t*>(a.w) + (size_t)nb * KQ * 1024;
    const uint32_t tap_bytes = (uint32_t)NB * KQ * 1024;
    const uint32_t wl16 = l * 16;
    u32x4 WB[2][8];
    auto load_w = [&](int ch, auto setc) {
        constexpr int set = decltype(setc)::value;
        static_for<0, 2>([&](auto hcst) {
            constexpr int h = decltype(hcst)::value;
            const int hcc = min(2 * min(ch, nchunks - 1) + h, nhc - 1), tp = hcc / kc64, c0 = (hcc - tp * kc64) * HCH;
            const int wt = (int)((cl.wtq >> (4 * tp)) & 15);
            const uint64_t q = (uint64_t)(uintptr_t)(wbase + (size_t)wt * tap_bytes + (size_t)(c0 / (2 * EPP)) * 1024);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)q), hi = __builtin_amdgcn_readfirstlane((uint32_t)(q >> 32));
            const uint64_t sb = ((uint64_t)hi << 32) | lo;
            static_for<0, 4>([&](auto uc) { gload16s<decltype(uc)::value * 1024>(WB[set][4 * h + decltype(uc)::value], sb, wl16); });
        });
    };
    uint32_t xa[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
        const int px = i * 32 + (l & 31);
        xa[i] = lds0 + px * 128 + (((l >> 5) * 16) ^ (((px >> 1) & 7) * 16));
    }
    f32x16 acc[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    stage_x(0);
    load_w(0, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<0, 8>([&](auto uc) { landed16(WB[0][decltype(uc)::value]); });
    auto chunk = [&](int ch, auto setc) {
        constexpr int set = decltype(setc)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ch + 1 < nchunks) {
            stage_x(ch + 1);
            load_w(ch + 1, std::integral_constant<int, set ^ 1>{});
        }
        const uint32_t xb = (ch & 1) * XB;
        bf16x8 XC[2][NBLK];
#pragma unroll
        for (int i = 0; i < NBLK; ++i) XC[0][i] = lds_b128p(xa[i] + xb);
        static_for<0, 8>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if constexpr (u + 1 < 8) {
                constexpr int un = u + 1;
#pragma unroll
                for (int i = 0; i < NBLK; ++i) XC[un & 1][i] = lds_b128p((xa[i] ^ ((un & 3) * 32)) + (un >> 2) * XH + xb);
            }
#pragma unroll
            for (int i = 0; i < NBLK; ++i) {
                if constexpr (F32) {
                    const f32x4 wv4 = __builtin_bit_cast(f32x4, WB[set][u]), xv4 = __builtin_bit_cast(f32x4, XC[u & 1][i]);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv4[jj], xv4[jj], acc[i], 0, 0, 0);
                } else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WB[set][u]), XC[u & 1][i], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the next chunk's requests (no register-destination load across the back edge)
        static_for<0, 8>([&](auto uc) { landed16(WB[set ^ 1][decltype(uc)::value]); });
    };
    for (int ch = 0; ch < nchunks; ch += 2) {
        chunk(ch, std::integral_constant<int, 0>{});
        if (ch + 1 < nchunks) chunk(ch + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: fp32 tile through LDS, whole channel rows out to the pixels' places on the produced tensor
    __syncthreads();
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    if (live) {
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
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
    const int j = t & 15, col = n0 + 8 * j;
    if (col >= a.Nc) return;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (a.bias) { b0 = *reinterpret_cast<const f32x4*>(a.bias + col); b1 = *reinterpret_cast<const f32x4*>(a.bias + col + 4); }
#pragma unroll
    for (int it = 0; it < PXT / 16; ++it) {
        const int p = it * 16 + (t >> 4);
        const int mm = m0 + p, n = mm >> a.lgGHW, r = mm & ((1 << a.lgGHW) - 1);
        const size_t m = ((size_t)n * a.OH + (r >> a.lgGW) * a.so + cl.ao) * a.OW + (r & ((1 << a.lgGW) - 1)) * a.so + cl.bo;
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
                v0 += f32x4{__uint_as_float(o.x << 16), __uint_as_float(o.x & 0xffff0000u), __uint_as_float(o.y << 16), __uint_as_float(o.y & 0xffff0000u)};
                v1 += f32x4{__uint_as_float(o.z << 16), __uint_as_float(o.z & 0xffff0000u), __uint_as_float(o.w << 16), __uint_as_float(o.w & 0xffff0000u)};
            }
            *reinterpret_cast<u32x4*>(yp) = u32x4{pack_bf16(v0.x, v0.y), pack_bf16(v0.z, v0.w), pack_bf16(v1.x, v1.y), pack_bf16(v1.z, v1.w)};
        } else {
            float* yp = reinterpret_cast<float*>(a.y) + m * a.ldy + col;
            if (a.accumulate) { v0 += *reinterpret_cast<const f32x4*>(yp); v1 += *reinterpret_cast<const f32x4*>(yp + 4); }
            *reinterpret_cast<f32x4*>(yp) = v0;
            *reinterpret_cast<f32x4*>(yp + 4) = v1;
            if (a.y16)         // the bf16 copy the next Block's conv and its weight gradient read, from the same epilogue (wave-uniform)
                *reinterpret_cast<u32x4*>(a.y16 + m * a.ldy16 + col) =
                    u32x4{pack_bf16(v0.x, v0.y), pack_bf16(v0.z, v0.w), pack_bf16(v1.x, v1.y), pack_bf16(v1.z, v1.w)};
        }
    }
}

bool pw1_ok(const MiConvDesc* d, bool f32 = false, bool in32 = false) {
    if (in32) {
        if (d->KH != 1 || d->KW != 1 || d->pad != 0 || d->stride != 1 || d->mode != 1 || d->IH != d->OH || d->IW != d->OW) return false;
        if (d->K % 128 || d->K1 % 128 || d->K1 <= 0 || d->K1 > d->K || d->Nc % 64 || d->ldx % 4 || (d->K1 != d->K && d->ldx2 % 4)) return false;
        return ((long)d->N * d->OH * d->OW) % 128 == 0;
    }
    if (f32) {
        if (d->KH != 1 || d->KW != 1 || d->pad != 0 || d->stride != 1 || d->mode != 0 || d->IH != d->OH || d->IW != d->OW) return false;
        if (d->K % 64 || d->K1 % 64 || d->K1 <= 0 || d->K1 > d->K || d->Nc % 64 || d->ldx % 4 || (d->K1 != d->K && d->ldx2 % 4)) return false;
        return ((long)d->N * d->OH * d->OW) % 128 == 0;
    }
    if (d->KH != 1 || d->KW != 1 || d->pad != 0 || d->stride != 1 || d->mode != 1) return false;
    if (d->IH != d->OH || d->IW != d->OW) return false;
    if (d->K % 128 || d->K1 % 128 || d->K1 <= 0 || d->K1 > d->K || d->Nc % 64 || d->ldx % 8 || (d->K1 != d->K && d->ldx2 % 8)) return false;
    return ((long)d->N * d->OH * d->OW) % 128 == 0;
}

struct PwGn { const float* sums; const float* gamma; const float* beta; const float* temb; int ldt, G; float eps; };

static int pw_launch(const char* who, const MiConvDesc* d, const void* x, const void* x2, const void* w_frag_bf16, const float* bias,
                     const float* residual, void* y, int out_bf16, int var, float* gsum, const float* coef, void* stream,
                     const PwGn* gn = nullptr, bool in32 = false, bool f32 = false) {
    PwArgs a{};
    if (!d || !x || !w_frag_bf16 || !y) return mi_set_error(-1, "%s: null argument", who);
    const int pt = pw_pick_tile(d, var, &a.TH, &a.TI, in32, f32, &a.ksplit);
    if (!pt) return mi_set_error(-1, "%s: descriptor not supported by the private-weight-stream conv kernel", who);
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (a.ksplit > 1) {
        uint8_t* ws = reinterpret_cast<uint8_t*>(g_pw_sk_ws[dev_ & 63].load(std::memory_order_acquire));
        a.sk_flag = reinterpret_cast<int*>(ws); a.sk_part = reinterpret_cast<float*>(ws + PW_SK_FLAG_BYTES);
#ifdef MI_PW_SK_ABL
        a.sk_flag = nullptr;                     // profiling build: no exchange (wrong results)
#endif
    }
    if (d->K1 != d->K && !x2) return mi_set_error(-1, "%s: two-source split without x2", who);
    if ((((uintptr_t)x | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)w_frag_bf16) & 15) != 0) return mi_set_error(-1, "%s: operands must be 16-byte aligned", who);
    if (d->ldy % 8 || (residual && d->ldr % 4)) return mi_set_error(-1, "%s: output pixel stride must be a multiple of 8, the residual's of 4", who);
    if (var >= 2 && (a.TI != 1 || d->K1 != d->K))
        return mi_set_error(-1, "%s: the fused variants need one image per tile and one source", who);
    if (var == 2 && (!coef || ((uintptr_t)coef & 15)))
        return mi_set_error(-1, "%s: the fused variant needs a 16-byte aligned coefficient tensor", who);
    if (var == 3) {
        if (!gn || !gn->sums || !gn->gamma || !gn->beta || gn->G <= 0 || d->K % gn->G || (d->K / gn->G) % 16 || d->K / gn->G > 64)
            return mi_set_error(-1, "%s: sums, gamma, beta; K / G in {16, 32, 64}", who);
        if ((((uintptr_t)gn->gamma | (uintptr_t)gn->beta | (uintptr_t)gn->sums) & 7) || ((uintptr_t)gn->sums & 15) || (gn->temb && (((uintptr_t)gn->temb & 15) || gn->ldt % 4)))
            return mi_set_error(-1, "%s: misaligned GroupNorm operands", who);
        a.sums = gn->sums; a.gamma = gn->gamma; a.beta = gn->beta; a.temb = gn->temb; a.ldt = gn->ldt; a.cpg = d->K / gn->G;
        a.cpg_magic = pw_magic(a.cpg); a.ng = gn->G; a.hw = d->OH * d->OW; a.eps = gn->eps; a.icnt = 1.0 / (MI_GSUM_SCALE * (double)a.hw * (double)a.cpg);
    }
    if (var == 1 && (!gsum || d->Nc % 16)) return mi_set_error(-1, "%s: GroupNorm sums need Nc %% 16 == 0 and a sum buffer", who);
    a.x = (const uint16_t*)x; a.x2 = (const uint16_t*)(x2 ? x2 : x); a.w = (const uint16_t*)w_frag_bf16; a.bias = bias; a.res = residual; a.y = y;
    a.N = d->N; a.H = d->OH; a.W = d->OW; a.K = d->K; a.Nc = d->Nc; a.K1 = d->K1; a.ldx = d->ldx; a.ldx2 = x2 ? d->ldx2 : d->ldx;
    a.ldy = d->ldy; a.ldr = d->ldr; a.accumulate = d->accumulate; a.flip = d->transposed ? 1 : 0;
    a.gsum = gsum; a.coef = coef;
    a.tiles_per_img = a.TI > 1 ? 1 : a.H / a.TH;
    a.XP = a.TI * (a.TH + 2) * a.W;
    a.xmap = a.TI == 1 && a.tiles_per_img > 1 && a.N % 8 == 0;
    // (the zero page's address is per DEVICE and a failed lookup is not remembered)
    static std::atomic<const void*> zero_pages[64];
    const void* zero_page = zero_pages[dev_ & 63].load(std::memory_order_acquire);
    if (!zero_page) {
        void* p_ = nullptr;
        if (hipGetSymbolAddress(&p_, HIP_SYMBOL(g_zero_page3)) != hipSuccess || !p_) return mi_set_error(-1, "%s: zero page address", who);
        zero_pages[dev_ & 63].store(p_, std::memory_order_release);
        zero_page = p_;
    }
    a.zero = zero_page; a.tpi_magic = pw_magic(a.tiles_per_img); a.nch_magic = pw_magic(d->K / (f32 ? 32 : 64) / a.ksplit);
    { int ns = a.TH / (pt / 32), ln = 0; while ((1 << ln) < ns) ++ln; a.lnsub = ln; }
    dim3 grid((unsigned)((long)d->N * d->OH * d->OW / pt), (unsigned)((d->Nc + 127) / 128));
    a.qmap = 0; a.gx = (int)grid.x; a.gy = (int)grid.y;
    if (!a.xmap && a.gy > 1 && a.gy % 2 == 0 && a.gx % 4 == 0) { a.qmap = 2; grid = dim3(grid.x * grid.y, 1, 1); }
    a.ppx = a.gx / 4; a.cpq = a.gy / 2; a.cpq_magic = pw_magic(a.cpq);
    grid.z = (unsigned)a.ksplit;
    hipStream_t st = (hipStream_t)stream;
    size_t lds = pw_lds(pt, in32 && !(var >= 2 && pw_fpipe(true)));
    if (var >= 2 && pw_fpipe(in32)) {                        // the coefficient table sits behind the two activation buffers + 2 KB
        const size_t need = (size_t)2 * pw_xp(pt) * 128 + 2048 + (size_t)12 * d->K;
        if (need > lds) lds = need;
    }
#define MI_PW_GO_T(O16, V, A, T) do { \
        static MiPerDevice once_; \
        once_.run([] { (void)hipFuncSetAttribute((const void*)conv_pw_kernel<O16, V, A, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        hipLaunchKernelGGL((conv_pw_kernel<O16, V, A, T>), grid, dim3(256), lds, st, a); } while (0)
#define MI_PW_GO(O16, V, A) MI_PW_GO_T(O16, V, A, 128)
#ifdef MI_PW_ABL_BUILD
    static const int abl = [] { const char* e = getenv("MI_PW_ABL"); return e ? atoi(e) : 0; }();
    if (abl & 8) lds = 100 * 1024;            // one workgroup per CU
    if (var == 0 && pt == 64 && (abl & 0x37)) {       // debugging the 64-pixel tiles
        switch (abl & 0x37) {
            case 1: MI_PW_GO_T(false, 0, 1, 64); break;
            case 2: MI_PW_GO_T(false, 0, 2, 64); break;
            case 3: MI_PW_GO_T(false, 0, 3, 64); break;
            case 4: MI_PW_GO_T(false, 0, 4, 64); break;
            default: MI_PW_GO_T(false, 0, 7, 64); break;
        }
        hipError_t e_ = hipGetLastError();
        return e_ == hipSuccess ? 0 : mi_set_error((int)e_, "%s: %s", who, hipGetErrorString(e_));
    }
    if (var == 0 && pt == 128 && (abl & 0x37)) {
#define MI_PW_ABL_CASE(V) case V: if (out_bf16) MI_PW_GO(true, 0, V); else MI_PW_GO(false, 0, V); break;
        switch (abl & 0x37) {
            MI_PW_ABL_CASE(1) MI_PW_ABL_CASE(2) MI_PW_ABL_CASE(3) MI_PW_ABL_CASE(4) MI_PW_ABL_CASE(7)
            MI_PW_ABL_CASE(16) MI_PW_ABL_CASE(17) MI_PW_ABL_CASE(19) MI_PW_ABL_CASE(23) MI_PW_ABL_CASE(32) MI_PW_ABL_CASE(36)
            default: return mi_set_error(-1, "MI_PW_ABL: combination not built");
        }
#undef MI_PW_ABL_CASE
        hipError_t e_ = hipGetLastError();
        return e_ == hipSuccess ? 0 : mi_set_error((int)e_, "%s: %s", who, hipGetErrorString(e_));
    }
#endif
#define MI_PW_GO_X(O16, V, T) do { \
        static MiPerDevice once_; \
        once_.run([] { (void)hipFuncSetAttribute((const void*)conv_pw_kernel<O16, V, 0, T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        hipLaunchKernelGGL((conv_pw_kernel<O16, V, 0, T, true>), grid, dim3(256), lds, st, a); } while (0)
    if (f32) {
#define MI_PW_GO_F(T) do { \
        static MiPerDevice once_; \
        once_.run([] { (void)hipFuncSetAttribute((const void*)conv_pw_kernel<false, 0, 0, T, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        hipLaunchKernelGGL((conv_pw_kernel<false, 0, 0, T, false, true>), grid, dim3(256), lds, st, a); } while (0)
        if (out_bf16) return mi_set_error(-1, "%s: the exact-fp32 kernel writes fp32", who);
        if (pt == 64) MI_PW_GO_F(64); else MI_PW_GO_F(128);
#undef MI_PW_GO_F
    } else if (in32) {
        if (pt == 64) switch (var) {
            case 1: if (out_bf16) MI_PW_GO_X(true, 1, 64); else MI_PW_GO_X(false, 1, 64); break;
            case 2: if (out_bf16) MI_PW_GO_X(true, 2, 64); else MI_PW_GO_X(false, 2, 64); break;
            case 3: if (out_bf16) MI_PW_GO_X(true, 3, 64); else MI_PW_GO_X(false, 3, 64); break;
            default: if (out_bf16) MI_PW_GO_X(true, 0, 64); else MI_PW_GO_X(false, 0, 64); break;
        } else switch (var) {
            case 1: if (out_bf16) MI_PW_GO_X(true, 1, 128); else MI_PW_GO_X(false, 1, 128); break;
            case 2: if (out_bf16) MI_PW_GO_X(true, 2, 128); else MI_PW_GO_X(false, 2, 128); break;
            case 3: if (out_bf16) MI_PW_GO_X(true, 3, 128); else MI_PW_GO_X(false, 3, 128); break;
            default: if (out_bf16) MI_PW_GO_X(true, 0, 128); else MI_PW_GO_X(false, 0, 128); break;
        }
    } else if (pt == 256) switch (var) {
        case 1: if (out_bf16) MI_PW_GO_T(true, 1, 0, 256); else MI_PW_GO_T(false, 1, 0, 256); break;
        default: if (out_bf16) MI_PW_GO_T(true, 0, 0, 256); else MI_PW_GO_T(false, 0, 0, 256); break;
    } else if (pt == 64) switch (var) {
        case 1: if (out_bf16) MI_PW_GO_T(true, 1, 0, 64); else MI_PW_GO_T(false, 1, 0, 64); break;
        case 2: if (out_bf16) MI_PW_GO_T(true, 2, 0, 64); else MI_PW_GO_T(false, 2, 0, 64); break;
        case 3: if (out_bf16) MI_PW_GO_T(true, 3, 0, 64); else MI_PW_GO_T(false, 3, 0, 64); break;
        default: if (out_bf16) MI_PW_GO_T(true, 0, 0, 64); else MI_PW_GO_T(false, 0, 0, 64); break;
    } else switch (var) {
        case 1: if (out_bf16) MI_PW_GO(true, 1, 0); else MI_PW_GO(false, 1, 0); break;
        case 2: if (out_bf16) MI_PW_GO(true, 2, 0); else MI_PW_GO(false, 2, 0); break;
        case 3: if (out_bf16) MI_PW_GO(true, 3, 0); else MI_PW_GO(false, 3, 0); break;
        default: if (out_bf16) MI_PW_GO(true, 0, 0); else MI_PW_GO(false, 0, 0); break;
    }
#undef MI_PW_GO
#undef MI_PW_GO_T
#undef MI_PW_GO_X
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : mi_set_error((int)e_, "%s: %s", who, hipGetErrorString(e_));
}

}  // namespace

// split-K workspace (see pw_launch): ws = device memory the caller zeroed once and keeps alive, bytes >= 64 KB of flags + the partial tiles of the
// largest split launch (a launch that does not fit runs unsplit); null / 0 switches the split launches off for the calling thread's device.
extern "C" int mi_conv_pw_set_splitk_workspace(void* ws, size_t bytes) {
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (ws && (bytes < 2 * PW_SK_FLAG_BYTES || ((uintptr_t)ws & 255))) return mi_set_error(-1, "mi_conv_pw_set_splitk_workspace: 256-byte aligned, >= 128 KB");
    g_pw_sk_bytes[dev_ & 63].store(ws ? bytes : 0, std::memory_order_release);
    g_pw_sk_ws[dev_ & 63].store(ws, std::memory_order_release);
    return 0;
}
// tests / A-B: 0 no split launches, 1 the rule (default), 2 / 4 that split wherever the geometry allows; returns the previous mode
extern "C" int mi_debug_conv_pw_splitk(int mode) {
    const int was = g_pw_sk_mode;
    if (mode == 0 || mode == 1 || mode == 2 || mode == 4) g_pw_sk_mode = mode;
    return was;
}
// the split a launch of this descriptor would use now (1: none) + 16 x the tile it would run on -- profiling attribution
extern "C" int mi_conv3x3_pw_splitk(const MiConvDesc* d, int var, int in32) {
    int th, ti, ks = 1;
    const int pt = d ? pw_pick_tile(d, var, &th, &ti, in32 != 0, false, &ks) : 0;
    return pt ? (pt << 4) | ks : 1;
}

extern "C" int mi_conv3x3_pw_supported(const MiConvDesc* d) {
    int th, ti;
    return (d && pw_pick_tile(d, 0, &th, &ti)) ? 1 : 0;
}
// test switch: force the pixel tile (64 / 128), 0 = automatic
extern "C" int mi_debug_conv_pw_tile(int pt) {
    if (pt != 0 && pt != 64 && pt != 128 && pt != 256) return mi_set_error(-1, "mi_debug_conv_pw_tile: 0, 64, 128 or 256");
    g_pw_force_tile = pt;
    return 0;
}
// A/B switch: the automatic pick takes 256-pixel tiles from min_workgroups workgroups up (0: never)
extern "C" int mi_debug_conv_pw_auto256(int min_workgroups) {
    if (min_workgroups < 0) return mi_set_error(-1, "mi_debug_conv_pw_auto256: min_workgroups >= 0");
    g_pw_auto256 = min_workgroups > 0; g_pw_min256 = min_workgroups > 0 ? min_workgroups : 256;
    return 0;
}
// pixels per workgroup the launch would use (256, 128 or 64; 0: not supported) -- profiling attribution and the host layer's pick
extern "C" int mi_conv3x3_pw_tile(const MiConvDesc* d) {
    int th, ti;
    return d ? pw_pick_tile(d, 0, &th, &ti) : 0;
}
// the fused GroupNorm-apply + Mish + conv variant: tiles inside one image, one source
extern "C" int mi_conv3x3_pw_gn_mish_supported(const MiConvDesc* d) {
    int th, ti;
    return (d && pw_pick_tile(d, 2, &th, &ti) && ti == 1 && d->K1 == d->K) ? 1 : 0;
}
// pixels per workgroup of the fused variants (128 or 64; 0: not supported)
extern "C" int mi_conv3x3_pw_gn_mish_tile(const MiConvDesc* d) {
    int th, ti;
    const int pt = d ? pw_pick_tile(d, 2, &th, &ti) : 0;
    return (pt && ti == 1 && d->K1 == d->K) ? pt : 0;
}

// x / x2: bf16 tensors (pixel strides in elements, % 8 == 0); w: bf16 weights in MFMA-fragment order [tap][Nc / 32][K / 16][64][8]
// (mi_pack_weights_bf16's wfq for the forward conv, wdq with d->transposed = 1 -> flipped taps for the data gradient);
// out_bf16: y is written as bf16 (else fp32).  bias / residual fp32, d->accumulate: y += result.
extern "C" int mi_conv3x3_pw(const MiConvDesc* d, const void* x, const void* x2, const void* w_frag_bf16, const float* bias,
                             const float* residual, void* y, int out_bf16, void* stream) {
    return pw_launch(__func__, d, x, x2, w_frag_bf16, bias, residual, y, out_bf16, 0, nullptr, nullptr, stream);
}
// ... that also adds the sum and the sum of squares of the values it stores, per sample and 16-channel slab, into gsum [N][Nc/16][2]
// (zeroed by the caller): the statistics of the GroupNorm that follows (mi_gn_coef_from_sums), without a pass over y
extern "C" int mi_conv3x3_pw_gnsums(const MiConvDesc* d, const void* x, const void* x2, const void* w_frag_bf16, const float* bias,
                                    const float* residual, void* y, int out_bf16, float* gsum, void* stream) {
    return pw_launch(__func__, d, x, x2, w_frag_bf16, bias, residual, y, out_bf16, 1, gsum, nullptr, stream);
}
// The same two entry points for fp32 x / x2 (pixel strides in floats, % 4 == 0): the residual-stream tensors as they are, rounded to
// bf16 once while they are staged (the numbers mi_f32_to_bf16 + mi_conv3x3_pw give).  gsum may be null (no sums).
extern "C" int mi_conv3x3_pw_x32_supported(const MiConvDesc* d) {
    int th, ti;
    return (d && pw_pick_tile(d, 0, &th, &ti, true)) ? 1 : 0;
}
extern "C" int mi_conv3x3_pw_x32_tile(const MiConvDesc* d) {
    int th, ti;
    return d ? pw_pick_tile(d, 0, &th, &ti, true) : 0;
}
extern "C" int mi_conv3x3_pw_x32(const MiConvDesc* d, const float* x, const float* x2, const void* w_frag_bf16, const float* bias,
                                 const float* residual, void* y, int out_bf16, float* gsum, void* stream) {
    return pw_launch(__func__, d, x, x2, w_frag_bf16, bias, residual, y, out_bf16, gsum ? 1 : 0, gsum, nullptr, stream, nullptr, true);
}
// ---- exact-fp32 mode (d->mode = 0): x / x2 / y fp32 (pixel strides in floats, % 4 == 0), w_frag_f32 = the layer's slice of
//      mi_pack_weights_f32frag's wfq32 (forward) or wdq32 (d->transposed = 1: data gradient)
extern "C" int mi_conv3x3_pw_f32_tile(const MiConvDesc* d) {
    int th, ti;
    return d ? pw_pick_tile(d, 0, &th, &ti, false, true) : 0;
}
extern "C" int mi_conv3x3_pw_f32(const MiConvDesc* d, const float* x, const float* x2, const float* w_frag_f32, const float* bias,
                                 const float* residual, float* y, void* stream) {
    return pw_launch(__func__, d, x, x2, w_frag_f32, bias, residual, y, 0, 0, nullptr, nullptr, stream, nullptr, false, true);
}

namespace {
// fp32 fragment-order copies of the conv weights the table flags (3x3 and 1x1 layers with ci % 64 == 0 and co % 64 == 0), same offsets
// as the master buffer [tap][ci][co]; one workgroup per 64 x 64 tile (the tile numbering of mi_pack_weights_bf16's table):
//   wfq32[tap][co / 32][ci / 8][lane][4] = W[tap][ci = 8 ko + 4 (lane >> 5) + j][co = 32 nb + (lane & 31)]   (forward operand)
//   wdq32[tap][ci / 32][co / 8][lane][4] = W[tap][ci = 32 nb + (lane & 31)][co = 8 ko + 4 (lane >> 5) + j]   (data-gradient operand)
struct PackEntryF { long long off; int taps, ci, co, tile0, frag, pad_; };
__global__ __launch_bounds__(256) void pack_f32frag_kernel(const PackEntryF* __restrict__ ents, int nent, const float* __restrict__ master,
                                                           float* __restrict__ wdq, float* __restrict__ wfq) {
    __shared__ float tile[64][65];
    int e = 0;
    for (int hi = nent; hi - e > 1;) {
        const int mid = (e + hi) >> 1;
        if ((int)blockIdx.x >= ents[mid].tile0) e = mid; else hi = mid;
    }
    const PackEntryF en = ents[e];
    if (!en.frag) return;
    const int local = blockIdx.x - en.tile0, tci = en.ci / 64, tco = en.co / 64;
    const int tap = local / (tci * tco), rem = local - tap * (tci * tco), bi = rem / tco, bj = rem - bi * tco;
    const size_t tapo = (size_t)en.off + (size_t)tap * en.ci * en.co;
    const float* src = master + tapo;
    const int c4 = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + 16 * p;
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)(bi * 64 + r) * en.co + bj * 64 + c4);
        tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
    }
    __syncthreads();
    const int l = threadIdx.x & 63;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int f = (threadIdx.x >> 6) + 4 * h;            // 16 fragments per order: (32-row block f >> 3, octet f & 7)
        const int r32 = (f >> 3) * 32 + (l & 31), c4o = (f & 7) * 8 + 4 * (l >> 5);
        *reinterpret_cast<f32x4*>(wdq + tapo + ((size_t)(bi * 2 + (f >> 3)) * (en.co / 8) + bj * 8 + (f & 7)) * 256 + l * 4) =
            f32x4{tile[r32][c4o], tile[r32][c4o + 1], tile[r32][c4o + 2], tile[r32][c4o + 3]};
        *reinterpret_cast<f32x4*>(wfq + tapo + ((size_t)(bj * 2 + (f >> 3)) * (en.ci / 8) + bi * 8 + (f & 7)) * 256 + l * 4) =
            f32x4{tile[c4o][r32], tile[c4o + 1][r32], tile[c4o + 2][r32], tile[c4o + 3][r32]};
    }
}
}  // namespace

// entries_dev / total_tiles: mi_pack_weights_bf16's table; wdq32 / wfq32: fp32 buffers of the master buffer's size (16-byte aligned)
extern "C" int mi_pack_weights_f32frag(int nent, const void* entries_dev, int total_tiles, const float* master, float* wdq32, float* wfq32,
                                       void* stream) {
    MI_REQUIRE(nent > 0 && entries_dev && total_tiles > 0 && master && wdq32 && wfq32, "bad argument");
    MI_REQUIRE((((uintptr_t)wdq32 | (uintptr_t)wfq32 | (uintptr_t)master) & 15) == 0, "buffers must be 16-byte aligned");
    hipLaunchKernelGGL(pack_f32frag_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const PackEntryF*)entries_dev, nent, master,
                       wdq32, wfq32);
    MI_LAUNCH_CHECK();
    return 0;
}

// BASELINE.json's named kernel on this structure: y = conv3x3(mish(x * scale + shift) + tb) + bias, x the RAW bf16 output of the
// previous conv, coef [3][N][K] = scale, shift, tb (mi_gn_coef_from_sums / mi_gn_stats_coef)
extern "C" int mi_conv3x3_pw_gn_mish(const MiConvDesc* d, const void* x, const float* coef, const void* w_frag_bf16, const float* bias,
                                     void* y, int out_bf16, void* stream) {
    return pw_launch(__func__, d, x, nullptr, w_frag_bf16, bias, nullptr, y, out_bf16, 2, nullptr, coef, stream);
}
// ... with the coefficients resolved in the kernel from the sums the producing conv's epilogue left (mi_conv3x3_pw_gnsums or
// mi_conv3x3_bf16w_io_gnsums: sums [N][K / 16][2]), gamma / beta [K] and the block's time-bias rows temb [N][ldt] (optional): Block ->
// Block is two launches, no statistics pass and no coefficient tensor.  K / G in {16, 32, 64}.
extern "C" int mi_conv3x3_pw_gn_mish_sums(const MiConvDesc* d, const void* x, const float* sums, const float* gamma, const float* beta,
                                          const float* temb, int ldt, int G, float eps, const void* w_frag_bf16, const float* bias,
                                          void* y, int out_bf16, void* stream) {
    const PwGn gn{sums, gamma, beta, temb, ldt, G, eps};
    return pw_launch(__func__, d, x, nullptr, w_frag_bf16, bias, nullptr, y, out_bf16, 3, nullptr, nullptr, stream, &gn);
}

// The named fused kernel for fp32-STORED activations (round 4): x is the raw fp32 output of the previous conv (pixel stride in floats,
// % 4 == 0), the transform is applied to the fp32 values in registers between their load and the one rounding to bf16; y fp32 or bf16.
extern "C" int mi_conv3x3_pw_x32_gn_mish_supported(const MiConvDesc* d) {
    int th, ti;
    return (d && pw_pick_tile(d, 2, &th, &ti, true) && ti == 1 && d->K1 == d->K) ? 1 : 0;
}
extern "C" int mi_conv3x3_pw_x32_gn_mish_tile(const MiConvDesc* d) {
    int th, ti;
    const int pt = d ? pw_pick_tile(d, 2, &th, &ti, true) : 0;
    return (pt && ti == 1 && d->K1 == d->K) ? pt : 0;
}
extern "C" int mi_conv3x3_pw_x32_gn_mish(const MiConvDesc* d, const float* x, const float* coef, const void* w_frag_bf16, const float* bias,
                                         void* y, int out_bf16, void* stream) {
    return pw_launch(__func__, d, x, nullptr, w_frag_bf16, bias, nullptr, y, out_bf16, 2, nullptr, coef, stream, nullptr, true);
}
extern "C" int mi_conv3x3_pw_x32_gn_mish_sums(const MiConvDesc* d, const float* x, const float* sums, const float* gamma, const float* beta,
                                              const float* temb, int ldt, int G, float eps, const void* w_frag_bf16, const float* bias,
                                              void* y, int out_bf16, void* stream) {
    const PwGn gn{sums, gamma, beta, temb, ldt, G, eps};
    return pw_launch(__func__, d, x, nullptr, w_frag_bf16, bias, nullptr, y, out_bf16, 3, nullptr, nullptr, stream, &gn, true);
}

// ---- 1x1 convs with K % 128 == 0 (see conv1x1_pw_kernel): w_frag_bf16 = the layer's slice of wfq (d->transposed = 0) or wdq (data
//      gradient); x2: second source of a two-source layer (channels K1 .. K - 1, d->K1 % 128 == 0), else null
extern "C" int mi_conv1x1_pw_supported(const MiConvDesc* d) { return (d && pw1_ok(d)) ? 1 : 0; }
// fp32 x / x2 in the bf16 MFMA mode (pixel strides in floats, % 4 == 0): rounded to bf16 once while staged
extern "C" int mi_conv1x1_pw_x32_supported(const MiConvDesc* d) { return (d && pw1_ok(d, false, true)) ? 1 : 0; }
extern "C" int mi_conv1x1_pw_x32(const MiConvDesc* d, const float* x, const float* x2, const void* w_frag_bf16, const float* bias,
                                 const float* residual, void* y, int out_bf16, void* stream) {
    MI_REQUIRE(d && x && w_frag_bf16 && y, "null argument");
    MI_REQUIRE(pw1_ok(d, false, true), "descriptor not supported (1x1, bf16 mode, K and K1 % 128 == 0, Nc % 64 == 0, N*H*W % 128 == 0)");
    MI_REQUIRE(d->K1 == d->K || x2, "two-source split without x2");
    MI_REQUIRE((((uintptr_t)x | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)w_frag_bf16) & 15) == 0, "operands must be 16-byte aligned");
    MI_REQUIRE(d->ldy % 8 == 0 && (!residual || d->ldr % 4 == 0), "pixel strides: y % 8, residual % 4");
    Pw1Args a{};
    a.x = (const uint16_t*)x; a.x2 = (const uint16_t*)(x2 ? x2 : x); a.w = (const uint16_t*)w_frag_bf16; a.bias = bias; a.res = residual;
    a.y = y; a.y16 = nullptr;
    a.M = d->N * d->OH * d->OW; a.K = d->K; a.K1 = d->K1; a.Nc = d->Nc; a.ldx = d->ldx; a.ldx2 = x2 ? d->ldx2 : d->ldx;
    a.ldy = d->ldy; a.ldr = d->ldr; a.ldy16 = 0; a.accumulate = d->accumulate;
    a.gy = (d->Nc + 127) / 128;
    a.gx = a.M / 64;                 // always 64-pixel tiles: the fp32 pieces of a 128-pixel tile would not fit the register budget
    dim3 grid((unsigned)a.gx, (unsigned)a.gy);
    if (a.gy > 1 && a.gx % 8 == 0) grid = dim3((unsigned)(a.gx * a.gy), 1, 1);
    hipStream_t st = (hipStream_t)stream;
#define MI_PW1X_GO(O16, PX) do { \
        static MiPerDevice once_; \
        once_.run([] { (void)hipFuncSetAttribute((const void*)conv1x1_pw_kernel<O16, false, PX, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); }); \
        hipLaunchKernelGGL((conv1x1_pw_kernel<O16, false, PX, false, true>), grid, dim3(256), pw1_lds(PX), st, a); } while (0)
    if (out_bf16) MI_PW1X_GO(true, 64); else MI_PW1X_GO(false, 64);
#undef MI_PW1X_GO
    MI_LAUNCH_CHECK();
    return 0;
}
// exact-fp32 mode (d->mode = 0): x / x2 / y fp32, w_frag_f32 from mi_pack_weights_f32frag; K % 64 == 0, K1 % 64 == 0
extern "C" int mi_conv1x1_pw_f32_supported(const MiConvDesc* d) { return (d && pw1_ok(d, true)) ? 1 : 0; }
extern "C" int mi_conv1x1_pw_f32(const MiConvDesc* d, const float* x, const float* x2, const float* w_frag_f32, const float* bias,
                                 const float* residual, float* y, void* stream) {
    MI_REQUIRE(d && x && w_frag_f32 && y, "null argument");
    MI_REQUIRE(pw1_ok(d, true), "descriptor not supported (1x1, fp32 mode, K and K1 % 64 == 0, Nc % 64 == 0, N*H*W % 128 == 0)");
    MI_REQUIRE(d->K1 == d->K || x2, "two-source split without x2");
    MI_REQUIRE((((uintptr_t)x | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)w_frag_f32) & 15) == 0, "operands must be 16-byte aligned");
    MI_REQUIRE(d->ldy % 8 == 0 && (!residual || d->ldr % 4 == 0), "pixel strides: y % 8, residual % 4");
    Pw1Args a{};
    a.x = (const uint16_t*)x; a.x2 = (const uint16_t*)(x2 ? x2 : x); a.w = (const uint16_t*)w_frag_f32; a.bias = bias; a.res = residual;
    a.y = y; a.y16 = nullptr;
    a.M = d->N * d->OH * d->OW; a.K = d->K; a.K1 = d->K1; a.Nc = d->Nc; a.ldx = d->ldx; a.ldx2 = x2 ? d->ldx2 : d->ldx;
    a.ldy = d->ldy; a.ldr = d->ldr; a.ldy16 = 0; a.accumulate = d->accumulate;
    a.gy = (d->Nc + 127) / 128;
    const bool small = (long)(a.M / 128) * a.gy < 256;
    a.gx = a.M / (small ? 64 : 128);
    dim3 grid((unsigned)a.gx, (unsigned)a.gy);
    if (a.gy > 1 && a.gx % 8 == 0) grid = dim3((unsigned)(a.gx * a.gy), 1, 1);
    hipStream_t st = (hipStream_t)stream;
    static MiPerDevice once_;
    once_.run([] {
        (void)hipFuncSetAttribute((const void*)conv1x1_pw_kernel<false, false, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute((const void*)conv1x1_pw_kernel<false, false, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); });
    if (small) hipLaunchKernelGGL((conv1x1_pw_kernel<false, false, 64, true>), grid, dim3(256), pw1_lds(64), st, a);
    else hipLaunchKernelGGL((conv1x1_pw_kernel<false, false, 128, true>), grid, dim3(256), 64 * 1024, st, a);
    MI_LAUNCH_CHECK();
    return 0;
}
// y fp32 or bf16 (out_bf16); y_bf16 (optional, fp32 y only): the bf16 copy of y written by the same epilogue, pixel stride ldy16
static int g_pw1_nloop = 1;           // tests / A-B: 0 = the 2-D grid for every layer, 2 = the loop form from two channel tiles up whatever the grid
static int g_pw1_nloop_min = 1024;
extern "C" int mi_debug_conv1x1_pw_nloop(int on) {
    const int was = g_pw1_nloop;
    if (on >= 0) { g_pw1_nloop = on > 2 ? 1 : on; g_pw1_nloop_min = on == 2 ? 0 : 1024; }
    return was;
}
static int pw1_go(const MiConvDesc* d, const void* x, const void* x2, const void* w_frag_bf16, int wbatch, const float* bias, const float* residual,
                  void* y, int out_bf16, void* y_bf16, int ldy16, void* stream);
extern "C" int mi_conv1x1_pw(const MiConvDesc* d, const void* x, const void* x2, const void* w_frag_bf16, const float* bias, const float* residual,
                             void* y, int out_bf16, void* y_bf16, int ldy16, void* stream) {
    return pw1_go(d, x, x2, w_frag_bf16, 0, bias, residual, y, out_bf16, y_bf16, ldy16, stream);
}
// ... with PER-SAMPLE weights: sample n of the batch (OH * OW pixels) multiplies with the fragments at w_frag_bf16 + n * wbatch elements (wbatch % 8 == 0;
// a pixel tile must lie inside one sample: OH * OW % 64 == 0).  The LinearAttention fold: x = qkv (ldx = 3 * hidden, K = hidden: the q channels),
// weights from mi_linattn_fold_fwd.
extern "C" int mi_conv1x1_pw_batched(const MiConvDesc* d, const void* x, const void* w_frag_bf16, int wbatch, const float* bias, const float* residual,
                                     void* y, int out_bf16, void* y_bf16, int ldy16, void* stream) {
    MI_REQUIRE(d && wbatch > 0 && wbatch % 8 == 0 && (d->OH * d->OW) % 64 == 0 && d->K1 == d->K, "per-sample weights: wbatch % 8 == 0, OH * OW % 64 == 0, one source");
    return pw1_go(d, x, nullptr, w_frag_bf16, wbatch, bias, residual, y, out_bf16, y_bf16, ldy16, stream);
}
static int pw1_go(const MiConvDesc* d, const void* x, const void* x2, const void* w_frag_bf16, int wbatch, const float* bias, const float* residual,
                  void* y, int out_bf16, void* y_bf16, int ldy16, void* stream) {
    MI_REQUIRE(d && x && w_frag_bf16 && y, "null argument");
    MI_REQUIRE(pw1_ok(d), "descriptor not supported (1x1, K and K1 % 128 == 0, Nc % 64 == 0, N*H*W % 128 == 0)");
    MI_REQUIRE(d->K1 == d->K || x2, "two-source split without x2");
    MI_REQUIRE((((uintptr_t)x | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)w_frag_bf16) & 15) == 0, "operands must be 16-byte aligned");
    MI_REQUIRE(d->ldy % 8 == 0 && (!residual || d->ldr % 4 == 0) && (!y_bf16 || (ldy16 % 8 == 0 && !out_bf16 && !d->accumulate)),
               "pixel strides: y % 8, residual % 4, bf16 copy % 8 (fp32 y, no accumulate)");
    Pw1Args a{};
    a.x = (const uint16_t*)x; a.x2 = (const uint16_t*)(x2 ? x2 : x); a.w = (const uint16_t*)w_frag_bf16; a.bias = bias; a.res = residual;
    a.y = y; a.y16 = (uint16_t*)y_bf16;
    a.M = d->N * d->OH * d->OW; a.K = d->K; a.K1 = d->K1; a.Nc = d->Nc; a.ldx = d->ldx; a.ldx2 = x2 ? d->ldx2 : d->ldx;
    a.ldy = d->ldy; a.ldr = d->ldr; a.ldy16 = ldy16; a.accumulate = d->accumulate;
    a.gy = (d->Nc + 127) / 128;
    a.wbatch = wbatch; a.hw = d->OH * d->OW;
    // 64-pixel tiles when 128-pixel ones would leave CUs without a workgroup (or would straddle two samples' weights)
    const bool small = (long)(a.M / 128) * a.gy < 256 || (wbatch && a.hw % 128 != 0);
    a.gx = a.M / (small ? 64 : 128);
    dim3 grid((unsigned)a.gx, (unsigned)a.gy);
    if (a.gy > 1 && a.gx % 8 == 0) grid = dim3((unsigned)(a.gx * a.gy), 1, 1);
    constexpr size_t lds = 64 * 1024;
    hipStream_t st = (hipStream_t)stream;
#define MI_PW1_GO(O16, DU, PX) do { \
        static MiPerDevice once_; \
        once_.run([] { (void)hipFuncSetAttribute((const void*)conv1x1_pw_kernel<O16, DU, PX>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); }); \
        hipLaunchKernelGGL((conv1x1_pw_kernel<O16, DU, PX>), grid, dim3(256), pw1_lds(PX), st, a); } while (0)
#define MI_PW1_PICK(O16, DU) do { if (small) MI_PW1_GO(O16, DU, 64); else MI_PW1_GO(O16, DU, 128); } while (0)
    // to_qkv at the 128-channel level: one workgroup per pixel tile walks the channel tiles (conv1x1_pw_kernel's NLOOP)
    // (from 1 024 pixel tiles up: [128,32,32,128] -> 384 goes 34.3 -> 29.3 us = 3.9 -> 4.6 TB/s of algorithmic traffic; at the sampler's B = 64 the
    //  512 workgroups of the loop form are one round of two per CU and time the same as the 1 536 of the 2-D grid or slightly worse)
    if (out_bf16 && !residual && !d->accumulate && !small && !wbatch && d->K == 128 && d->K1 == d->K && a.gy > 1 && g_pw1_nloop && a.gx >= g_pw1_nloop_min) {
        static MiPerDevice once_;
        once_.run([] { (void)hipFuncSetAttribute((const void*)conv1x1_pw_kernel<true, false, 128, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); });
        hipLaunchKernelGGL((conv1x1_pw_kernel<true, false, 128, false, false, true>), dim3((unsigned)a.gx), dim3(256), pw1_lds(128), st, a);
        MI_LAUNCH_CHECK();
        return 0;
    }
    if (out_bf16) MI_PW1_PICK(true, false);
    else if (y_bf16) MI_PW1_PICK(false, true);
    else MI_PW1_PICK(false, false);
#undef MI_PW1_PICK
#undef MI_PW1_GO
    MI_LAUNCH_CHECK();
    return 0;
}

// ---- inference: channel LayerNorm + the bias-free 1x1 conv behind it in one launch (see ln_conv1x1_pw_kernel).  x: the fp32 residual stream
//      [M][ldx], g / b: the LayerNorm's affine parameters [K], w_frag_bf16: the conv's slice of wfq, y: bf16 [M][ldy]
static bool lnpw1_ok(const MiConvDesc* d) {
    if (!d || d->KH != 1 || d->KW != 1 || d->pad != 0 || d->stride != 1 || d->mode != 1 || d->IH != d->OH || d->IW != d->OW) return false;
    if ((d->K != 128 && d->K != 256 && d->K != 512) || d->K1 != d->K || d->Nc % 64 || d->ldx % 4 || d->ldy % 8 || d->accumulate) return false;
    return ((long)d->N * d->OH * d->OW) % 128 == 0;
}
extern "C" int mi_ln_conv1x1_pw_supported(const MiConvDesc* d) { return lnpw1_ok(d) ? 1 : 0; }
static int lnpw1_launch(const MiConvDesc* d, const float* x, const float* ln_g, const float* ln_b, float eps, const void* w_frag_bf16,
                        const float* bias, void* y_bf16, void* ln_bf16, int ldl, void* stream);
extern "C" int mi_ln_conv1x1_pw(const MiConvDesc* d, const float* x, const float* ln_g, const float* ln_b, float eps, const void* w_frag_bf16,
                                const float* bias, void* y_bf16, void* stream) {
    return lnpw1_launch(d, x, ln_g, ln_b, eps, w_frag_bf16, bias, y_bf16, nullptr, 0, stream);
}
// ... that also writes the normalised tensor (bf16 [M][ldl], ldl % 8 == 0): training -- to_qkv's weight gradient reads it, the LayerNorm launch is gone
extern "C" int mi_ln_conv1x1_pw_dual(const MiConvDesc* d, const float* x, const float* ln_g, const float* ln_b, float eps, const void* w_frag_bf16,
                                     const float* bias, void* y_bf16, void* ln_bf16, int ldl, void* stream) {
    MI_REQUIRE(ln_bf16 && ldl % 8 == 0 && (((uintptr_t)ln_bf16) & 15) == 0, "the normalised tensor: 16-byte aligned, ldl % 8 == 0");
    return lnpw1_launch(d, x, ln_g, ln_b, eps, w_frag_bf16, bias, y_bf16, ln_bf16, ldl, stream);
}
static int lnpw1_launch(const MiConvDesc* d, const float* x, const float* ln_g, const float* ln_b, float eps, const void* w_frag_bf16,
                        const float* bias, void* y_bf16, void* ln_bf16, int ldl, void* stream) {
    MI_REQUIRE(d && x && ln_g && ln_b && w_frag_bf16 && y_bf16, "null argument");
    MI_REQUIRE(lnpw1_ok(d), "descriptor not supported (1x1, bf16 mode, one source, K = 128 / 256 / 512, Nc % 64 == 0, N*H*W % 128 == 0, ldx % 4, ldy % 8)");
    MI_REQUIRE((((uintptr_t)x | (uintptr_t)ln_g | (uintptr_t)ln_b | (uintptr_t)w_frag_bf16 | (uintptr_t)y_bf16) & 15) == 0, "operands must be 16-byte aligned");
    LnPw1Args a{};
    a.x = x; a.w = (const uint16_t*)w_frag_bf16; a.g = ln_g; a.b = ln_b; a.bias = bias; a.y = (uint16_t*)y_bf16;
    a.M = d->N * d->OH * d->OW; a.K = d->K; a.Nc = d->Nc; a.ldx = d->ldx; a.ldy = d->ldy; a.gy = (d->Nc + 127) / 128; a.eps = eps;
    a.ln_out = (uint16_t*)ln_bf16; a.ldl = ldl;
    hipStream_t st = (hipStream_t)stream;
#define MI_LNPW1_GO(PX, NC, G) do { \
        static MiPerDevice once_; \
        once_.run([] { (void)hipFuncSetAttribute((const void*)ln_conv1x1_pw_kernel<PX, NC, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lnpw1_lds(PX, NC)); }); \
        a.gx = a.M / PX; \
        hipLaunchKernelGGL((ln_conv1x1_pw_kernel<PX, NC, G>), dim3((unsigned)(a.gx * (G ? a.gy : 1))), dim3(256), lnpw1_lds(PX, NC), st, a); } while (0)
    // fewer than two workgroups per CU: one workgroup per (pixel tile, channel tile) -- sampler, B = 64, in the replayed step: 16x16 x 256 ch 12.0 -> 9.8 us,
    // 8x8 x 512 ch 13.5 -> 9.3 us; the 32x32 level (512 tiles of 128 pixels) stays on the channel-tile loop (19.9 us; 23.6 as a 2-D grid)
#define MI_LNPW1_PICK(PX, NC) do { if (a.M / PX < 512) MI_LNPW1_GO(PX, NC, true); else MI_LNPW1_GO(PX, NC, false); } while (0)
    if (d->K == 128) { if (a.M / 128 >= 512) MI_LNPW1_GO(128, 1, false); else MI_LNPW1_PICK(64, 1); }
    else if (d->K == 256) MI_LNPW1_PICK(64, 2);
    else MI_LNPW1_PICK(64, 4);
#undef MI_LNPW1_PICK
#undef MI_LNPW1_GO
    MI_LAUNCH_CHECK();
    return 0;
}

static int ilog2_exact(int v) { int lg = 0; while ((1 << lg) < v) ++lg; return (1 << lg) == v ? lg : -1; }

// descriptor -> classes; false when the kernel does not take the layer
static bool gt_plan(const MiConvDesc* d, GtArgs& a, int* ncls, int* pxt) {
    if (!d || (d->mode != 1 && d->mode != 0) || d->K1 != d->K || d->K % 64 || d->Nc % 64 || d->ldx % (d->mode == 1 ? 8 : 4)) return false;
    const int k = d->KH, s = d->stride, p = d->pad;
    if (d->KW != k || k * k > 16 || k < 1 || (s != 1 && s != 2) || p < 0 || p > 7) return false;
    if ((long)d->Nc * d->K * (d->mode == 1 ? 2 : 4) * k * k >= (1L << 31)) return false;
    int GH, GW;
    *ncls = 0;
    if (!d->transposed) {
        if (d->OH != (d->IH + 2 * p - k) / s + 1 || d->OW != (d->IW + 2 * p - k) / s + 1) return false;
        GH = d->OH; GW = d->OW; a.si = s; a.so = 1;
        GtClass c{};
        for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx) {
                const int tp = c.ntap++;
                c.dyq |= (unsigned long long)(ky - p + 8) << (4 * tp); c.dxq |= (unsigned long long)(kx - p + 8) << (4 * tp);
                c.wtq |= (unsigned long long)(ky * k + kx) << (4 * tp);
            }
        a.cls[(*ncls)++] = c;
    } else {
        // produced pixel oy = s u + cy gathers iy = (oy + p - ky) / s where that is divisible: ky = (cy + p) mod s, + s, ...
        if (d->OH != s * d->IH || d->OW != s * d->IW) return false;      // (the layers of this path: the produced tensor is s x the gathered one)
        GH = d->OH / s; GW = d->OW / s; a.si = 1; a.so = s;
        for (int cy = 0; cy < s; ++cy)
            for (int cx = 0; cx < s; ++cx) {
                GtClass c{};
                c.ao = cy; c.bo = cx;
                for (int ky = (cy + p) % s; ky < k; ky += s)
                    for (int kx = (cx + p) % s; kx < k; kx += s) {
                        const int dy = (cy + p - ky) / s, dx = (cx + p - kx) / s;       // exact: the numerators are multiples of s
                        if (dy < -8 || dy > 7 || dx < -8 || dx > 7) return false;
                        const int tp = c.ntap++;
                        c.dyq |= (unsigned long long)(dy + 8) << (4 * tp); c.dxq |= (unsigned long long)(dx + 8) << (4 * tp);
                        c.wtq |= (unsigned long long)(ky * k + kx) << (4 * tp);
                    }
                if (c.ntap == 0) return false;
                a.cls[(*ncls)++] = c;
            }
    }
    a.lgGW = ilog2_exact(GW); a.lgGHW = ilog2_exact(GH * GW);
    if (a.lgGW < 0 || a.lgGHW < 0) return false;
    a.M = d->N * GH * GW; a.K = d->K; a.Nc = d->Nc; a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW;
    if (a.M % 64) return false;
    a.gy = (d->Nc + 127) / 128;
    const bool small = a.M % 128 != 0 || (long)(a.M / 128) * a.gy * *ncls < 256;
    *pxt = small ? 64 : 128;
    a.gx = a.M / *pxt;
    return true;
}
extern "C" int mi_conv_gt_supported(const MiConvDesc* d) {
    GtArgs a{};
    int ncls, pxt;
    return gt_plan(d, a, &ncls, &pxt) ? 1 : 0;
}
// x: bf16 [N][IH][IW][K] (pixel stride ldx elements); w_frag: the layer's slice of mi_pack_weights_bf16's wfq (contraction over the master
// layout's ci: the forward convs) or wdq (over co: the data gradients); y fp32 or bf16 (out_bf16), bias / residual fp32.
// d->mode = 0 (exact-fp32 mode): x and y fp32 (ldx % 4 == 0), w_frag = the slice of mi_pack_weights_f32frag's wfq32 / wdq32.
static int gt_launch(const MiConvDesc* d, const void* x, const void* w_frag, const float* bias, const float* residual,
                     void* y, int out_bf16, void* y_bf16, int ldy16, void* stream);
extern "C" int mi_conv_gt(const MiConvDesc* d, const void* x, const void* w_frag, const float* bias, const float* residual,
                          void* y, int out_bf16, void* stream) {
    return gt_launch(d, x, w_frag, bias, residual, y, out_bf16, nullptr, 0, stream);
}
// ... with a bf16 copy of the fp32 output written by the same epilogue (pixel stride ldy16 elements, % 8 == 0; no accumulate): the
// operand of the next Block's conv and of its weight gradient, without a conversion launch
extern "C" int mi_conv_gt_dual(const MiConvDesc* d, const void* x, const void* w_frag, const float* bias, const float* residual,
                               float* y, void* y_bf16, int ldy16, void* stream) {
    MI_REQUIRE(y_bf16 && ldy16 % 8 == 0 && (((uintptr_t)y_bf16) & 15) == 0 && d && !d->accumulate && d->mode == 1,
               "bf16 copy: bf16 mode, 16-byte aligned, ldy16 % 8 == 0, no accumulate");
    return gt_launch(d, x, w_frag, bias, residual, y, 0, y_bf16, ldy16, stream);
}
static int gt_launch(const MiConvDesc* d, const void* x, const void* w_frag, const float* bias, const float* residual,
                     void* y, int out_bf16, void* y_bf16, int ldy16, void* stream) {
    MI_REQUIRE(d && x && w_frag && y, "null argument");
    GtArgs a{};
    int ncls, pxt;
    MI_REQUIRE(gt_plan(d, a, &ncls, &pxt), "descriptor not supported by the tap-gather kernel (one source, K and Nc % 64 == 0, "
               "k * k <= 16 taps, stride 1 / 2, power-of-two grids, N*GH*GW % 64 == 0)");
    MI_REQUIRE((((uintptr_t)x | (uintptr_t)w_frag) & 15) == 0, "operands must be 16-byte aligned");
    MI_REQUIRE(d->ldy % 8 == 0 && (!residual || d->ldr % 4 == 0), "pixel strides: y % 8, residual % 4");
    MI_REQUIRE(d->mode == 1 || !out_bf16, "the exact-fp32 kernel writes fp32");
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w_frag; a.bias = bias; a.res = residual; a.y = y; a.y16 = (uint16_t*)y_bf16; a.ldy16 = ldy16;
    a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = d->ldr; a.accumulate = d->accumulate;
    dim3 grid((unsigned)a.gx, (unsigned)a.gy, (unsigned)ncls);
    if (a.gy > 1 && a.gx % 8 == 0) grid = dim3((unsigned)(a.gx * a.gy), 1, (unsigned)ncls);
    hipStream_t st = (hipStream_t)stream;
#define MI_GT_GO(O16, PX, F) do { \
        static MiPerDevice once_; \
        once_.run([] { (void)hipFuncSetAttribute((const void*)conv_gt_kernel<O16, PX, F>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); }); \
        hipLaunchKernelGGL((conv_gt_kernel<O16, PX, F>), grid, dim3(256), (size_t)(PX == 128 ? 64 : 32) * 1024, st, a); } while (0)
    if (d->mode == 0) { if (pxt == 128) MI_GT_GO(false, 128, true); else MI_GT_GO(false, 64, true); }
    else if (pxt == 128) { if (out_bf16) MI_GT_GO(true, 128, false); else MI_GT_GO(false, 128, false); }
    else { if (out_bf16) MI_GT_GO(true, 64, false); else MI_GT_GO(false, 64, false); }
#undef MI_GT_GO
    MI_LAUNCH_CHECK();
    return 0;
}
// pixels per workgroup and number of classes (launch attribution)
extern "C" int mi_conv_gt_tile(const MiConvDesc* d, int* pxt, int* ncls) {
    GtArgs a{};
    return gt_plan(d, a, ncls, pxt) ? 0 : -1;
}

