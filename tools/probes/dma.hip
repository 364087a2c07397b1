#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint32_t* out, const uint32_t* src, int nbytes) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xAAAAAAAAu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const int l = threadIdx.x;
    // per-lane source offset: reversed lanes; odd lanes out of range
    unsigned voff = (unsigned)((63 - l) * 16);
    if (l & 1) voff = 0xffffffffu;
    if (l == 2) voff = (unsigned)nbytes - 8;   // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    uint32_t *d, *s; hipMalloc(&d, 4096); hipMalloc(&s, 4096);
    uint32_t hs[1024]; for (int i = 0; i < 1024; ++i) hs[i] = i;
    hipMemcpy(s, hs, 4096, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, s, 1024);
    uint32_t h[1024]; hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    for (int l = 0; l < 12; ++l) printf("lds slot %2d: %08x %08x %08x %08x\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    printf("slot 63: %08x %08x %08x %08x ; slot 64: %08x\n", h[252], h[253], h[254], h[255], h[256]);
    return 0;
}
