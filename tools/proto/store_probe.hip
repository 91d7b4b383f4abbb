// Store-pattern probe (run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/proto/store_probe.hip -o /tmp/store_probe && /tmp/store_probe)
// A conv epilogue writes a [pixels][128] bf16 tensor; each wave owns 64 pixels x 64 channels.  Pattern A (what the MFMA layout gives
// when the weights are the A operand): lane = pixel, 8 bytes (4 channels) per store, 16 stores per wave tile -- every store
// instruction touches 64 different 256-byte rows.  Pattern B (after a transpose through LDS): lane = (pixel l / 8, 16-byte chunk
// l % 8), 8 stores per wave tile, a store instruction covers 8 whole 128-byte row segments.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PAT> __global__ __launch_bounds__(256) void probe(uint16_t* y, int ldy, int spin) {
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const size_t m0 = (size_t)blockIdx.x * 128 + wm * 64;
    float acc = (float)t;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;          // stand-in for the main loop
    const uint32_t v = __float_as_uint(acc);
    if (PAT == 0) {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int rq = 0; rq < 4; ++rq) {
                    const size_t m = m0 + i * 32 + (l & 31);
                    const int col = wn * 64 + j * 32 + 8 * rq + 4 * (l >> 5);
                    *reinterpret_cast<u32x2*>(y + m * ldy + col) = u32x2{v, v + 1};
                }
    } else {
        for (int k = 0; k < 8; ++k) {
            const size_t m = m0 + k * 8 + (l >> 3);
            const int col = wn * 64 + (l & 7) * 8;
            *reinterpret_cast<u32x4*>(y + m * ldy + col) = u32x4{v, v + 1, v + 2, v + 3};
        }
    }
}

int main() {
    const int M = 128 * 32 * 32, C = 128;
    uint16_t* y; hipMalloc(&y, (size_t)M * C * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : {0, 2000, 8000}) {
        for (int pat = 0; pat < 2; ++pat) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < 10; ++it) {
                    if (pat == 0) hipLaunchKernelGGL(probe<0>, dim3(M / 128), dim3(256), 0, 0, y, C, spin);
                    else hipLaunchKernelGGL(probe<1>, dim3(M / 128), dim3(256), 0, 0, y, C, spin);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("spin %5d pattern %c: %.2f us per launch (%.0f GB/s)\n", spin, pat ? 'B' : 'A', best * 100.f, (double)M * C * 2 / (best * 1e-4) / 1e9);
        }
    }
    return 0;
}
