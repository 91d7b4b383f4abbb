// Hardware probe: direction of v_mov_b32_dpp wave_shr:1 / wave_shl:1 on gfx950 (bound_ctrl: lanes without a source read 0).
// Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/dpp_probe tools/proto/dpp_probe.hip && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int v = 100 + threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);        // wave_shr:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);   // wave_shl:1
}
int main() {
    int* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("wave_shr:1 lane0 %d lane1 %d lane31 %d lane32 %d lane63 %d  (lane l reads lane l-1 -> 0 100 130 131 162)\n", h[0], h[1], h[31], h[32], h[63]);
    printf("wave_shl:1 lane0 %d lane1 %d lane31 %d lane32 %d lane63 %d  (lane l reads lane l+1 -> 101 102 132 133 0)\n", h[64], h[65], h[95], h[96], h[127]);
    return 0;
}
