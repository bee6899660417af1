// Does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?  (hip_ext.h says the flag is not supported on GFX9xx.)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long cycles, int* out) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int* d; hipMalloc(&d, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long cyc = 10000;      // wall_clock64: 100 MHz -> 100 us
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a, st);
            for (int i = 0; i < 50; i++) {
                hipLaunchKernelGGL(spin, dim3(32), dim3(64), 0, st, cyc, d);
                if (mode == 0) hipLaunchKernelGGL(spin, dim3(32), dim3(64), 0, st, cyc, d + 64);
                else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d + 64);
                else { hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d + 64);
                       hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d + 128); }
            }
            hipEventRecord(b, st); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("mode %d (%s): %.1f us per group of %d kernels of ~100 us\n", mode, mode == 0 ? "two in-order launches" : (mode == 1 ? "second launch any-order" : "second + third any-order"),
                   ms * 1000 / 50, mode == 2 ? 3 : 2);
        }
    }
    return 0;
}
