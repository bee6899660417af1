// Probe of ds_read_b64_tr_b16 (gfx950): which (source lane, element) lands in (lane, element)?  Every lane reads the 4 halfwords at
// LDS element offset 4 * lane; the value stored at element i is i, so out[lane][e] = 4 * source_lane + source_element.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + threadIdx.x * 4));
    for (int e = 0; e < 4; e++) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; e++) printf("  (L%2d,e%d)", h[l * 4 + e] / 4, h[l * 4 + e] % 4);
        printf("\n");
    }
    return 0;
}
