// Hardware probe for the next round's staging rewrite (DESIGN.md "Next-round plan"): pins down, on the GPU itself,
//   (1) the lane mapping of ds_read_b64_tr_b16 (gfx950 LDS transpose read) for a [4][16] bf16 block per 16-lane group,
//   (2) the destination rule of global_load_lds_dwordx4 (LDS-DMA): wave-uniform base + lane * 16.
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_probe tools/proto/lds_probe.hip && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void tr_probe(uint16_t* out, int row_stride_elems) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int l = threadIdx.x;
    // four [4][16] blocks (one per 16-lane group), rows `row_stride_elems` apart; element value = 1000*group + 16*row + col
    for (int i = l; i < 4 * 4 * row_stride_elems; i += 64) lds[i] = 0xffff;
    __syncthreads();
    for (int i = l; i < 4 * 64; i += 64) {
        const int g = i / 64, r = (i % 64) / 16, c = i % 16;
        lds[g * 4 * row_stride_elems + r * row_stride_elems + c] = (uint16_t)(1000 * g + 16 * r + c);
    }
    __syncthreads();
    const int g = l >> 4, i = l & 15;
    // hypothesis: lane i of a group supplies the address of ITS 8-byte piece (row i>>2, columns 4*(i&3)..+3) and receives column i
    const uint32_t addr = (uint32_t)(uintptr_t)(lds + g * 4 * row_stride_elems + (i >> 2) * row_stride_elems + (i & 3) * 4);
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = (uint16_t)(v.x & 0xffff); out[l * 4 + 1] = (uint16_t)(v.x >> 16);
    out[l * 4 + 2] = (uint16_t)(v.y & 0xffff); out[l * 4 + 3] = (uint16_t)(v.y >> 16);
}

__global__ void dma_probe(const uint32_t* __restrict__ src, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t ldsw[];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) ldsw[i] = 0xdeadbeefu;
    __syncthreads();
    // every lane fetches 16 bytes from ITS OWN global address (a permutation: lane l reads chunk (l * 7) % 64)
    const uint32_t* gp = src + ((l * 7) % 64) * 4;
    __builtin_amdgcn_global_load_lds(gp, ldsw + 128, 16, 0, 0);          // LDS base = word 128 (wave-uniform), + lane * 16 bytes
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 1024; i += 64) out[i] = ldsw[i];
}

int main() {
    for (int stride : {16, 24, 72}) {
        uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 4 * 4 * stride * 2 + 64, 0, d, stride);
        std::vector<uint16_t> h(256); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (h[l * 4 + j] != 1000 * (l >> 4) + 16 * j + (l & 15)) ++bad;
        printf("ds_read_b64_tr_b16, row stride %2d elems: hypothesis 'lane i gives piece (row i>>2, cols 4(i&3)..), gets column i' -> %s (%d mismatches)\n",
               stride, bad ? "WRONG" : "confirmed", bad);
        if (bad) for (int l = 0; l < 20; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        hipFree(d);
    }
    uint32_t *s, *o; hipMalloc(&s, 1024); hipMalloc(&o, 4096);
    std::vector<uint32_t> hs(256); for (int i = 0; i < 256; ++i) hs[i] = i;
    hipMemcpy(s, hs.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 4096, 0, s, o);
    std::vector<uint32_t> ho(1024); hipMemcpy(ho.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int w = 0; w < 4; ++w) if (ho[128 + l * 4 + w] != (uint32_t)(((l * 7) % 64) * 4 + w)) ++bad;
    int stray = 0; for (int i = 0; i < 1024; ++i) if ((i < 128 || i >= 384) && ho[i] != 0xdeadbeefu) ++stray;
    printf("global_load_lds_dwordx4: 'LDS dest = uniform base + lane*16, source per lane' -> %s (%d mismatches, %d stray writes)\n", bad || stray ? "WRONG" : "confirmed", bad, stray);
    if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d slot: %u %u %u %u\n", l, ho[128 + l * 4], ho[128 + l * 4 + 1], ho[128 + l * 4 + 2], ho[128 + l * 4 + 3]);
    return 0;
}
