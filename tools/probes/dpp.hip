// Probe: which source lane does __builtin_amdgcn_update_dpp(row_shl / row_shr : n) read, and what do lanes past the row end get?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dpp.hip -o tools/probes/dpp_probe && tools/probes/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    const int v = 100 + l;
    out[l] = __builtin_amdgcn_update_dpp(-1, v, 0x101, 0xf, 0xf, true);         // row_shl:1, bound_ctrl
    out[64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x103, 0xf, 0xf, true);    // row_shl:3
    out[128 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    out[192 + l] = __builtin_amdgcn_update_dpp(-7, v, 0x101, 0xf, 0xf, false);  // row_shl:1, no bound_ctrl (keeps old)
}
int main() {
    int* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"row_shl:1 bc", "row_shl:3 bc", "row_shr:1 bc", "row_shl:1 nobc"};
    for (int t = 0; t < 4; ++t) { printf("%-15s", names[t]); for (int l = 0; l < 20; ++l) printf(" %d", h[64 * t + l]); printf("\n"); }
    return 0;
}
