// What does synchronising the workgroups of ONE launch cost on gfx950, next to the dependent-launch boundary it would replace?  
//   hipcc --offload-arch=gfx950 -O2 tools/probes/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
// Measured, all on one stream, 256-thread workgroups:
//   (A) boundary: a chain of dependent launches, each workgroup re-writing a small / a 64 KB slab                          -> us per launch
//   (B) grid barrier inside one persistent launch, G workgroups: every workgroup publishes a 512-byte record (write-through `sc1` stores), arrives on ONE
//       monotonic counter (agent-scope relaxed atomic), one lane polls it (relaxed `sc1` load + s_sleep), then every workgroup re-reads G records and CHECKS them
//       (a stale record = a visibility bug of the protocol, counted)                                                           -> us per barrier
//   (C) the same with plain stores + __threadfence() on both sides (release before the arrive, acquire after the poll)
//   (D) XCD-hierarchical arrive: 8 per-XCD counters, the last arriver of an XCD arrives on the top counter, the last of those releases 8 generation words
//   (E) BatchNorm-style two-pass reduction over a tensor: (E1) reduce kernel -> fold kernel -> apply kernel (three launches, what the step does today),
//       (E2) reduce kernel whose LAST-ARRIVING workgroup folds (ticket) -> apply kernel, (E3) ONE launch: reduce, grid barrier, every workgroup folds the
//       records it needs, apply from registers.                                                                               -> us per BatchNorm backward
// All spins are bounded (a timeout sets an error word and lets the workgroup leave): the probe cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT
typedef unsigned long long u64;

__device__ __forceinline__ bool wait_ge(unsigned* ctr, unsigned target, unsigned* err) {
    for (unsigned spins = 0; spins < (1u << 22); spins++) {
        if (__hip_atomic_load(ctr, RLX, AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(err, 1u, RLX, AGENT);
    return false;
}

// ---- (A) ----
__global__ __launch_bounds__(256) void k_boundary(float* buf, int floats_per_wg) {
    float* p = buf + (long)blockIdx.x * floats_per_wg;
    for (int i = threadIdx.x; i < floats_per_wg; i += 256) p[i] = p[i] * 1.0001f + 1.f;
}

// ---- (B, C, D) ----
// rec[parity of the iteration][wg][64] u64: {iteration tag << 32 | wg} (two sets: a workgroup that is one barrier ahead writes the OTHER set); mode 0: sc1 stores / loads, 1: plain + __threadfence, 2: XCD-hierarchical arrive with sc1 records
__global__ __launch_bounds__(256) void k_barrier_loop(u64* rec, unsigned* ctr /* [0] top, [16 * (1 + x)] per-XCD, [16 * (9 + x)] generation */, unsigned* err, unsigned* stale, int iters, int mode) {
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    const int xcd = wg & 7, per_xcd = (G + 7 - xcd) / 8;
    unsigned bad = 0;
    for (int it = 0; it < iters; it++) {
        const u64 tag = ((u64)(it + 1) << 32) | (unsigned)wg;
        u64* rs = rec + (long)(it & 1) * 2048 * 64;
        if (tid < 64) {
            if (mode == 1) rs[(long)wg * 64 + tid] = tag + tid;
            else __hip_atomic_store(rs + (long)wg * 64 + tid, tag + tid, RLX, AGENT);
        }
        if (mode == 1) __threadfence();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have left (inline asm: invisible to the compiler's wait-count pass)
        __syncthreads();
        if (tid == 0) {
            bool ok;
            if (mode == 2) {
                const unsigned t = __hip_atomic_fetch_add(ctr + 16 * (1 + xcd), 1u, RLX, AGENT);
                if (t == (unsigned)(it + 1) * per_xcd - 1) {
                    const unsigned t2 = __hip_atomic_fetch_add(ctr, 1u, RLX, AGENT);
                    if (t2 == (unsigned)(it + 1) * (G < 8 ? G : 8) - 1)
                        for (int x = 0; x < 8; x++) __hip_atomic_store(ctr + 16 * (9 + x), (unsigned)(it + 1), RLX, AGENT);
                }
                ok = wait_ge(ctr + 16 * (9 + xcd), (unsigned)(it + 1), err);
            } else {
                __hip_atomic_fetch_add(ctr, 1u, RLX, AGENT);
                ok = wait_ge(ctr, (unsigned)(it + 1) * G, err);
            }
            (void)ok;
        }
        __syncthreads();
        if (mode == 1) __threadfence();
        // every workgroup checks one word of every other workgroup's record (thread t: workgroups t, t + 256, ...)
        for (int w = tid; w < G; w += 256) {
            const int word = (wg + w) & 63;
            const u64 v = mode == 1 ? ((volatile u64*)rs)[(long)w * 64 + word] : __hip_atomic_load(rs + (long)w * 64 + word, RLX, AGENT);
            if (v != ((((u64)(it + 1)) << 32) | (unsigned)w) + word) bad++;
        }
        if (__hip_atomic_load(err, RLX, AGENT)) break;
        __syncthreads();      // (this record set is next written two barriers from now: every reader has passed the barrier in between)
    }
    if (bad) atomicAdd(stale, bad);
}

// ---- (E) BatchNorm-backward-like: x, g (N pixels x C channels, NHWC) -> s1[c] = sum g, s2[c] = sum g * x ; out = g - s1 / N - x * s2 / N ----
struct BnArgs { const float* x; const float* g; float* out; double* part; double* sums; unsigned* ctr; unsigned* err; long npix; int C; int ppb; unsigned epoch; };
__device__ __forceinline__ void block_partials(const BnArgs& a, double* sh, float4* xv, float4* gv, int NI, bool keep) {
    const int C4 = a.C / 4, PT = 256 / C4, tid = threadIdx.x, pt = tid / C4, j = tid % C4;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long q0 = (long)blockIdx.x * a.ppb;
    for (int i = 0; i < NI; i++) {
        const long q = q0 + pt + (long)i * PT;
        float4 x = make_float4(0, 0, 0, 0), g = x;
        if (q < q0 + a.ppb && q < a.npix) { x = *(const float4*)(a.x + q * a.C + 4 * j); g = *(const float4*)(a.g + q * a.C + 4 * j); }
        if (keep) { xv[i] = x; gv[i] = g; }
        s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w;
        s[4] += (double)g.x * x.x; s[5] += (double)g.y * x.y; s[6] += (double)g.z * x.z; s[7] += (double)g.w * x.w;
    }
    for (int e = 0; e < 8; e++) sh[tid * 8 + e] = s[e];
    __syncthreads();
    for (int half = PT >> 1; half >= 1; half >>= 1) {
        if (pt < half) for (int e = 0; e < 8; e++) sh[tid * 8 + e] += sh[((pt + half) * C4 + j) * 8 + e];
        __syncthreads();
    }
}
template <int NI>
__global__ __launch_bounds__(256) void k_bn_reduce(BnArgs a, int ticket_fold) {
    __shared__ double sh[256 * 8];
    __shared__ int last;
    block_partials(a, sh, nullptr, nullptr, NI, false);
    const int C4 = a.C / 4, tid = threadIdx.x;
    if (tid < C4) for (int e = 0; e < 8; e++) {
        double* p = a.part + ((long)blockIdx.x * a.C * 2) + (e < 4 ? 4 * tid + e : a.C + 4 * tid + e - 4);
        if (ticket_fold) __hip_atomic_store((u64*)p, (u64)__double_as_longlong(sh[tid * 8 + e]), RLX, AGENT); else *p = sh[tid * 8 + e];
    }
    if (!ticket_fold) return;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(a.ctr, 1u, RLX, AGENT) == a.epoch * gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    for (int i = tid; i < 2 * a.C; i += 256) {
        double s = 0;
        for (int b = 0; b < (int)gridDim.x; b++) s += __longlong_as_double((long long)__hip_atomic_load((u64*)(a.part + (long)b * 2 * a.C + i), RLX, AGENT));
        a.sums[i] = s;
    }
}
__global__ __launch_bounds__(256) void k_bn_fold(BnArgs a, int nb) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    double s = 0;
    if (i < 2 * a.C) for (int b = lane; b < nb; b += 64) s += a.part[(long)b * 2 * a.C + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (i < 2 * a.C && lane == 0) a.sums[i] = s;
}
__global__ __launch_bounds__(256) void k_bn_apply(BnArgs a) {
    const int C4 = a.C / 4;
    const double inv = 1.0 / (double)a.npix;
    for (long it = blockIdx.x * 256L + threadIdx.x; it < a.npix * C4; it += (long)gridDim.x * 256) {
        const long q = it / C4; const int c = (int)(it - q * C4) * 4;
        const float4 x = *(const float4*)(a.x + q * a.C + c), g = *(const float4*)(a.g + q * a.C + c);
        float4 o;
        o.x = g.x - (float)(a.sums[c] * inv) - x.x * (float)(a.sums[a.C + c] * inv); o.y = g.y - (float)(a.sums[c + 1] * inv) - x.y * (float)(a.sums[a.C + c + 1] * inv);
        o.z = g.z - (float)(a.sums[c + 2] * inv) - x.z * (float)(a.sums[a.C + c + 2] * inv); o.w = g.w - (float)(a.sums[c + 3] * inv) - x.w * (float)(a.sums[a.C + c + 3] * inv);
        *(float4*)(a.out + q * a.C + c) = o;
    }
}
// one launch: partial sums, grid barrier, every workgroup folds all records (fixed order) for its own channel quads, apply from registers
template <int NI>
__global__ __launch_bounds__(256) void k_bn_coop(BnArgs a) {
    __shared__ double sh[256 * 8];
    float4 xv[NI], gv[NI];
    block_partials(a, sh, xv, gv, NI, true);
    const int C4 = a.C / 4, PT = 256 / C4, tid = threadIdx.x, pt = tid / C4, j = tid % C4, G = gridDim.x;
    if (tid < C4) for (int e = 0; e < 8; e++)
        __hip_atomic_store((u64*)(a.part + ((long)blockIdx.x * a.C * 2) + (e < 4 ? 4 * tid + e : a.C + 4 * tid + e - 4)), (u64)__double_as_longlong(sh[tid * 8 + e]), RLX, AGENT);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (tid == 0) { __hip_atomic_fetch_add(a.ctr, 1u, RLX, AGENT); wait_ge(a.ctr, a.epoch * G, a.err); }
    __syncthreads();
    // fold: thread (pt, j) sums records pt, pt + PT, ... for its 8 values, then the PT rows are combined through LDS in a fixed tree
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = pt; b < G; b += PT)
        for (int e = 0; e < 8; e++) s[e] += __longlong_as_double((long long)__hip_atomic_load((u64*)(a.part + (long)b * 2 * a.C + (e < 4 ? 4 * j + e : a.C + 4 * j + e - 4)), RLX, AGENT));
    for (int e = 0; e < 8; e++) sh[tid * 8 + e] = s[e];
    __syncthreads();
    for (int half = PT >> 1; half >= 1; half >>= 1) {
        if (pt < half) for (int e = 0; e < 8; e++) sh[tid * 8 + e] += sh[((pt + half) * C4 + j) * 8 + e];
        __syncthreads();
    }
    const double inv = 1.0 / (double)a.npix;
    float m1[4], m2[4];
    for (int e = 0; e < 4; e++) { m1[e] = (float)(sh[j * 8 + e] * inv); m2[e] = (float)(sh[j * 8 + 4 + e] * inv); }
    const long q0 = (long)blockIdx.x * a.ppb;
    for (int i = 0; i < NI; i++) {
        const long q = q0 + pt + (long)i * PT;
        if (q < q0 + a.ppb && q < a.npix) {
            float4 o;
            o.x = gv[i].x - m1[0] - xv[i].x * m2[0]; o.y = gv[i].y - m1[1] - xv[i].y * m2[1]; o.z = gv[i].z - m1[2] - xv[i].z * m2[2]; o.w = gv[i].w - m1[3] - xv[i].w * m2[3];
            *(float4*)(a.out + q * a.C + 4 * j) = o;
        }
    }
}

static float elapsed_us(hipEvent_t a, hipEvent_t b, int n) { float ms; hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b); return ms * 1000.f / n; }

int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* buf; hipMalloc(&buf, 512L * 65536 * 4);
    hipMemset(buf, 0, 512L * 65536 * 4);
    printf("(A) dependent-launch boundary, 256 workgroups x 256 threads, us per launch\n");
    for (int fl : {64, 1024, 16384}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, st);
            for (int i = 0; i < 400; i++) hipLaunchKernelGGL(k_boundary, dim3(256), dim3(256), 0, st, buf, fl);
            hipEventRecord(e1, st);
            printf("    %6d B re-written per workgroup: %.2f us\n", fl * 4, elapsed_us(e0, e1, 400));
        }
    }
    u64* rec; unsigned *ctr, *err, *stale;
    hipMalloc(&rec, 2 * 2048L * 64 * 8); hipMalloc(&ctr, 4096); hipMalloc(&err, 64); hipMalloc(&stale, 64);
    const char* mname[3] = {"(B) sc1 records, one counter", "(C) plain records + __threadfence() both sides", "(D) sc1 records, XCD-hierarchical arrive"};
    for (int mode = 0; mode < 3; mode++)
        for (int G : {128, 256, 512}) {
            hipMemsetAsync(rec, 0, 2 * 2048L * 64 * 8, st); hipMemsetAsync(ctr, 0, 4096, st); hipMemsetAsync(err, 0, 64, st); hipMemsetAsync(stale, 0, 64, st);
            const int iters = 300;
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(k_barrier_loop, dim3(G), dim3(256), 0, st, rec, ctr, err, stale, iters, mode);
            hipEventRecord(e1, st);
            const float us = elapsed_us(e0, e1, iters);
            unsigned he = 0, hs = 0; hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost);
            printf("%s, G = %3d: %.2f us per barrier (incl. publish + re-read of G records)   timeouts %u   stale words %u\n", mname[mode], G, us, he, hs);
        }
    printf("(E) BatchNorm-backward-shaped reduction + apply, us per call in a dependent chain\n");
    double *part, *sums; hipMalloc(&part, 1024L * 2 * 512 * 8); hipMalloc(&sums, 2 * 512 * 8);
    struct Shape { long npix; int C; } shapes[] = {{8L * 128 * 128, 16}, {8L * 64 * 64, 32}, {8L * 32 * 32, 64}, {8L * 32 * 32, 128}, {8L * 64 * 64, 128}};
    for (auto& sp : shapes) {
        const long el = sp.npix * sp.C;
        float *x, *g, *out; hipMalloc(&x, el * 4); hipMalloc(&g, el * 4); hipMalloc(&out, el * 4);
        std::vector<float> h(el); for (long i = 0; i < el; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
        hipMemcpy(x, h.data(), el * 4, hipMemcpyHostToDevice); hipMemcpy(g, h.data(), el * 4, hipMemcpyHostToDevice);
        const int C4 = sp.C / 4, PT = 256 / C4;
        constexpr int NI = 8;
        int ppb = PT * NI;                                 // pixels per workgroup when every thread keeps NI pixel quads
        int G = (int)((sp.npix + ppb - 1) / ppb);
        const bool coop_ok = G <= 512;
        BnArgs a{x, g, out, part, sums, ctr, err, sp.npix, sp.C, ppb, 0};
        const int N = 200;
        float t3, t2 = 0, t1 = 0;
        hipMemsetAsync(ctr, 0, 4096, st);
        hipEventRecord(e0, st);
        for (int i = 0; i < N; i++) {
            hipLaunchKernelGGL(k_bn_reduce<NI>, dim3(G), dim3(256), 0, st, a, 0);
            hipLaunchKernelGGL(k_bn_fold, dim3((2 * sp.C + 3) / 4), dim3(256), 0, st, a, G);
            hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)((sp.npix * C4 + 255) / 256)), dim3(256), 0, st, a);
        }
        hipEventRecord(e1, st); t3 = elapsed_us(e0, e1, N);
        std::vector<float> ref(el), got(el);
        hipMemcpy(ref.data(), out, el * 4, hipMemcpyDeviceToHost);
        hipMemsetAsync(ctr, 0, 4096, st);
        hipEventRecord(e0, st);
        for (int i = 0; i < N; i++) {
            a.epoch = i + 1;
            hipLaunchKernelGGL(k_bn_reduce<NI>, dim3(G), dim3(256), 0, st, a, 1);
            hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)((sp.npix * C4 + 255) / 256)), dim3(256), 0, st, a);
        }
        hipEventRecord(e1, st); t2 = elapsed_us(e0, e1, N);
        hipMemcpy(got.data(), out, el * 4, hipMemcpyDeviceToHost);
        long bad2 = 0; for (long i = 0; i < el; i++) if (fabsf(got[i] - ref[i]) > 1e-5f) bad2++;
        long bad1 = -1;
        if (coop_ok) {
            hipMemsetAsync(ctr, 0, 4096, st); hipMemsetAsync(err, 0, 64, st); hipMemsetAsync(out, 0, el * 4, st);
            hipEventRecord(e0, st);
            for (int i = 0; i < N; i++) { a.epoch = i + 1; hipLaunchKernelGGL(k_bn_coop<NI>, dim3(G), dim3(256), 0, st, a); }
            hipEventRecord(e1, st); t1 = elapsed_us(e0, e1, N);
            hipMemcpy(got.data(), out, el * 4, hipMemcpyDeviceToHost);
            bad1 = 0; for (long i = 0; i < el; i++) if (fabsf(got[i] - ref[i]) > 1e-5f) bad1++;
        }
        unsigned he = 0; hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost);
        printf("    %7ld px x %3d ch (%5.1f MB / tensor, %3d workgroups): 3 launches %.1f us | ticket fold + apply %.1f us (mismatching words %ld) | one cooperative launch %.1f us (mismatching %ld, timeouts %u)\n",
               sp.npix, sp.C, el * 4 / 1e6, G, t3, t2, bad2, t1, bad1, he);
        hipFree(x); hipFree(g); hipFree(out);
    }
    return 0;
}
