// What do hipEventRecord / hipStreamWaitEvent packets cost on the stream they are enqueued on (gfx950)?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/event_cost.hip -o /tmp/event_cost && /tmp/event_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(long cycles, int* out) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && out) out[blockIdx.x] = 1;
}
int main() {
    hipStream_t st, other; hipStreamCreateWithFlags(&st, hipStreamNonBlocking); hipStreamCreateWithFlags(&other, hipStreamNonBlocking);
    int* d; hipMalloc(&d, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int N = 400;
    std::vector<hipEvent_t> ev(N), evd(N);
    for (int i = 0; i < N; i++) { hipEventCreate(&ev[i]); hipEventCreateWithFlags(&evd[i], hipEventDisableTiming); }
    const char* names[] = {"kernels only", "+ hipEventRecord (timing event) after each", "+ hipEventRecord (hipEventDisableTiming) after each",
                           "+ record, and ANOTHER stream waits for it (fork)", "+ wait on an event the other stream recorded long ago (join of finished work)",
                           "+ fork a 10 us kernel to the other stream and join it right away"};
    for (int mode = 0; mode < 6; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 4) { for (int i = 0; i < N; i++) hipEventRecord(evd[i], other); hipStreamSynchronize(other); }
            hipEventRecord(a, st);
            for (int i = 0; i < N; i++) {
                hipLaunchKernelGGL(spin, dim3(32), dim3(64), 0, st, 1000L, d);      // ~10 us
                if (mode == 1) hipEventRecord(ev[i], st);
                if (mode == 2) hipEventRecord(evd[i], st);
                if (mode == 3) { hipEventRecord(evd[i], st); hipStreamWaitEvent(other, evd[i], 0); }
                if (mode == 4) hipStreamWaitEvent(st, evd[i], 0);
                if (mode == 5) { hipEventRecord(evd[i], st); hipStreamWaitEvent(other, evd[i], 0); hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, other, 1000L, d + 512);
                                 hipEventRecord(ev[i], other); hipStreamWaitEvent(st, ev[i], 0); }
            }
            hipEventRecord(b, st); hipEventSynchronize(b); hipStreamSynchronize(other);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%-78s %.2f us per iteration\n", names[mode], ms * 1000 / N);
        }
    }
    return 0;
}
