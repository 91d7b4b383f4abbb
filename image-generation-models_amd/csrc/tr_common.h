// Shared pieces of the LDS-DMA / transposing-read weight-gradient kernels (wgrad_tr.hip: 3x3, wgrad1x1_tr.hip: 1x1).
#pragma once
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

struct TrArgs {
    const uint16_t* P; const uint16_t* P2; const uint16_t* Q;
    float* ws;          // this problem's partial tiles (splits > 1)
    float* dW;          // [3][3][Ci][Cj], accumulated into directly when splits == 1
    int W, H, Ci, Cj, I1, ldp, ldp2, ldq;
    int total;          // steps of 64 pixels over the whole batch: N*H*W / 64
    int sps, splits;    // steps per k-slice, k-slices
    int gx, gy;         // ci tiles (64), co tiles (128)
    int wg0;            // first workgroup of this problem in the launch
    int tile0;          // first tile of this problem in the reduce launch
    int xcd_map;        // workgroup -> (k-slice, tile) such that the tiles of a k-slice share an XCD
};
constexpr int MAXP = 8;
// One launch = up to MAXP independent weight-gradient problems, each on its own share of the workgroups.  A layer's partial-tile
// volume is (its workgroups) x 288 KB: eight layers side by side on 32 CUs each write an eighth of what one layer on 256 CUs
// writes (75 MB) for the same MFMA work, and layers with >= 32 tiles need no k-slices at all (they add into dW directly).
struct TrBatch { TrArgs p[MAXP]; int n; };

// Workgroups per problem of a batched launch: every problem gets whole k-slices (a multiple of its tile count), at most `target`
// workgroups in total (one round of the chip), and the launch lasts as long as its slowest workgroup -- so slices are handed out
// greedily to the problem whose workgroups currently carry the most work each.  (Shares proportional to the work, rounded down to
// whole slices, left a 24-tile layer with 48 of the 70 workgroups it was due: the whole launch ran 1.46x longer.)
inline void balance_shares(int n, const double* work, const long* tiles, long target, long* wgs) {
    long total = 0;
    for (int i = 0; i < n; ++i) { wgs[i] = tiles[i]; total += tiles[i]; }
    for (;;) {
        int best = -1; double worst = 0;
        for (int i = 0; i < n; ++i) {
            const double per = work[i] / (double)wgs[i];
            if (per > worst && total + tiles[i] <= target) { worst = per; best = i; }
        }
        if (best < 0) break;
        // stop when the most loaded problem cannot be relieved: further slices for the others would not shorten the launch
        double top = 0;
        for (int i = 0; i < n; ++i) top = work[i] / (double)wgs[i] > top ? work[i] / (double)wgs[i] : top;
        if (worst < top) break;
        wgs[best] += tiles[best]; total += tiles[best];
    }
}

// LDS-DMA: 16 bytes per lane from each lane's own global address to LDS byte address lds_dst (wave-uniform) + 16 * lane.
// M0 carries the LDS base and is compiler-reserved: it is saved, set and restored inside the one statement.  The compiler does
// not know this is a load: completion is waited for with explicit s_waitcnt vmcnt(N) below.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ bf16x8 tr_pair(uint32_t a0, uint32_t a1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a1);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

}  // namespace
