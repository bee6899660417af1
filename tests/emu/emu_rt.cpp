// TEST INFRASTRUCTURE ONLY -- runtime of the functional HIP simulator (see hip/hip_runtime.h in this dir).
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <mutex>
#include <condition_variable>

extern "C" void emu_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

namespace emu {
thread_local Fiber* t_cur = nullptr;
thread_local uint3 t_bid;
thread_local dim3 t_bdim, t_gdim;

static constexpr size_t STACK = 256 * 1024;
static constexpr int MAXT = 1024;

struct WaveState { uint64_t slot[64]; float A[64], B[64]; float big[64 * 16]; int arrived; unsigned gen; int nlanes;
                   uint32_t dppv[64][8]; unsigned long dppseq[64]; };      // DPP mailboxes (8 posts deep: a lane that alternates partners inside its quad runs up to two exchanges ahead of a partner that has not read yet), usable under divergence
struct BlockState { int nthreads; int arrived; unsigned gen; WaveState waves[MAXT / 64]; };
static thread_local BlockState t_blk;
static thread_local void* t_sched_sp;
static thread_local const std::function<void()>* t_body;
static thread_local unsigned long t_progress;
static thread_local char* t_stacks = nullptr;
static thread_local Fiber* t_fibers = nullptr;

static inline void yield() { Fiber* f = t_cur; emu_switch(&f->sp, t_sched_sp); }

static void fiber_main() {
    (*t_body)();
    Fiber* f = t_cur;
    f->done = true;
    t_progress++;
    emu_switch(&f->sp, t_sched_sp);
    abort();
}

void syncthreads() {
    BlockState& b = t_blk;
    unsigned g = b.gen;
    if (++b.arrived == b.nthreads) { b.arrived = 0; b.gen++; t_progress++; }
    else while (b.gen == g) yield();
}
void wave_sync() {
    WaveState& w = t_blk.waves[t_cur->lin >> 6];
    unsigned g = w.gen;
    if (++w.arrived == w.nlanes) { w.arrived = 0; w.gen++; t_progress++; }
    else while (w.gen == g) yield();
}
uint64_t wave_exchange(uint64_t v, int src_lane) {
    WaveState& w = t_blk.waves[t_cur->lin >> 6];
    w.slot[t_cur->lin & 63] = v;
    wave_sync();
    uint64_t r = (src_lane >= 0 && src_lane < w.nlanes) ? w.slot[src_lane] : v;
    wave_sync();
    return r;
}
// v_mov_b32_dpp quad_perm: every lane posts its value and reads the post of `src_lane` with the same sequence number.  NOT a wave-wide collective: lanes outside the
// quad may have left the code path (the hardware's EXEC mask); the lanes that exchange must execute the same sequence of calls.
uint32_t dpp_exchange(uint32_t v, int src_lane) {
    WaveState& w = t_blk.waves[t_cur->lin >> 6];
    const int me = t_cur->lin & 63;
    const unsigned long s = w.dppseq[me];
    w.dppv[me][s & 7] = v;
    w.dppseq[me] = s + 1;
    t_progress++;
    if (src_lane == me || src_lane < 0 || src_lane >= w.nlanes) return v;
    while (w.dppseq[src_lane] <= s) yield();
    return w.dppv[src_lane][s & 7];
}
void wave_gather2(float a, float b, float* A64, float* B64) {
    WaveState& w = t_blk.waves[t_cur->lin >> 6];
    int l = t_cur->lin & 63;
    w.A[l] = a; w.B[l] = b;
    wave_sync();
    memcpy(A64, w.A, sizeof(float) * 64); memcpy(B64, w.B, sizeof(float) * 64);
    wave_sync();
}

void wave_gather_n(const float* mine, int n, float* all) {
    WaveState& w = t_blk.waves[t_cur->lin >> 6];
    int l = t_cur->lin & 63;
    for (int i = 0; i < n; i++) w.big[l * n + i] = mine[i];
    wave_sync();
    memcpy(all, w.big, sizeof(float) * 64 * n);
    wave_sync();
}
//  32x32x16 bf16: A[i=l&31][k=(l>>5)*8+j], B[k=(l>>5)*8+j][j'=l&31], j<8; D as 32x32x2 (cdna_hip_programming.md section 3)
f32x16 mfma_f32_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c, int, int, int) {
    float mine[16], all[64 * 16];
    for (int j = 0; j < 8; j++) { mine[j] = (float)a[j]; mine[8 + j] = (float)b[j]; }
    wave_gather_n(mine, 16, all);
    int l = t_cur->lin & 63, col = l & 31;
    for (int r = 0; r < 16; r++) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; k++) acc = fmaf(all[(row + 32 * (k >> 3)) * 16 + (k & 7)], all[(col + 32 * (k >> 3)) * 16 + 8 + (k & 7)], acc);
        c[r] = acc;
    }
    return c;
}
// ds_read_b64_tr_b16: every lane reads the 4 halfwords at its own address; within each 16-lane group the 16 x 4 block is transposed:
// lane j, element i  <-  lane (4 i + (j >> 2)), element (j & 3)     (probe output on gfx950)
s16x4_t ds_read_tr16_b64(const void* p) {
    uint64_t mine; memcpy(&mine, p, 8);
    int l = t_cur->lin & 63, base = l & ~15, j = l & 15;
    s16x4_t r;
    for (int i = 0; i < 4; i++) {
        uint64_t u = wave_exchange(mine, base + 4 * i + (j >> 2));
        r[i] = (short)((u >> (16 * (j & 3))) & 0xFFFF);
    }
    return r;
}
f32x16 mfma_f32_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c, int, int, int) {      // same fragment maps as the bf16 form
    float mine[16], all[64 * 16];
    for (int j = 0; j < 8; j++) { mine[j] = (float)a[j]; mine[8 + j] = (float)b[j]; }
    wave_gather_n(mine, 16, all);
    int l = t_cur->lin & 63, col = l & 31;
    for (int r = 0; r < 16; r++) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; k++) acc = fmaf(all[(row + 32 * (k >> 3)) * 16 + (k & 7)], all[(col + 32 * (k >> 3)) * 16 + 8 + (k & 7)], acc);
        c[r] = acc;
    }
    return c;
}

// 16x16x32 (gfx950): lane l holds A[i = l & 15][k = 8 (l >> 4) .. + 7] and B[k = 8 (l >> 4) .. + 7][j = l & 15]; D: col = l & 15, row = 4 (l >> 4) + r
// (/opt/skills/guides/cdna_hip_programming.md section 3; checked on the MI355X by the conv_head kernel tests against torch)
f32x4 mfma_f32_16x16x32_f16(f16x8 a, f16x8 b, f32x4 c, int, int, int) {
    float mine[16], all[64 * 16];
    for (int j = 0; j < 8; j++) { mine[j] = (float)a[j]; mine[8 + j] = (float)b[j]; }
    wave_gather_n(mine, 16, all);
    int l = t_cur->lin & 63, col = l & 15;
    for (int r = 0; r < 4; r++) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; k++) acc = fmaf(all[(row + 16 * (k >> 3)) * 16 + (k & 7)], all[(col + 16 * (k >> 3)) * 16 + 8 + (k & 7)], acc);
        c[r] = acc;
    }
    return c;
}

f32x4 mfma_f32_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c, int, int, int) {      // same fragment maps as the f16 form
    float mine[16], all[64 * 16];
    for (int j = 0; j < 8; j++) { mine[j] = (float)a[j]; mine[8 + j] = (float)b[j]; }
    wave_gather_n(mine, 16, all);
    int l = t_cur->lin & 63, col = l & 15;
    for (int r = 0; r < 4; r++) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; k++) acc = fmaf(all[(row + 16 * (k >> 3)) * 16 + (k & 7)], all[(col + 16 * (k >> 3)) * 16 + 8 + (k & 7)], acc);
        c[r] = acc;
    }
    return c;
}

// Fragment maps per /opt/skills/guides/cdna_hip_programming.md section 3:
//  32x32x2: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
f32x16 mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int) {
    float A[64], B[64];
    wave_gather2(a, b, A, B);
    int l = t_cur->lin & 63, col = l & 31;
    for (int r = 0; r < 16; r++) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; k++) acc = fmaf(A[row + 32 * k], B[col + 32 * k], acc);
        c[r] = acc;
    }
    return c;
}
//  16x16x4: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+r
f32x4 mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
    float A[64], B[64];
    wave_gather2(a, b, A, B);
    int l = t_cur->lin & 63, col = l & 15;
    for (int r = 0; r < 4; r++) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; k++) acc = fmaf(A[row + 16 * k], B[col + 16 * k], acc);
        c[r] = acc;
    }
    return c;
}

static void run_block(unsigned bx, unsigned by, unsigned bz, dim3 grid, dim3 block, const std::function<void()>& body) {
    int n = block.x * block.y * block.z;
    if (n > MAXT) { fprintf(stderr, "emu: block too large %d\n", n); abort(); }
    if (!t_stacks) {
        t_stacks = (char*)mmap(nullptr, STACK * MAXT, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (t_stacks == (char*)MAP_FAILED) { perror("mmap"); abort(); }
        t_fibers = new Fiber[MAXT];
    }
    t_bid = uint3{bx, by, bz}; t_bdim = block; t_gdim = grid; t_body = &body;
    BlockState& b = t_blk;
    b.nthreads = n; b.arrived = 0; b.gen = 0;
    int nw = (n + 63) / 64;
    for (int w = 0; w < nw; w++) { b.waves[w].arrived = 0; b.waves[w].gen = 0; b.waves[w].nlanes = std::min(64, n - 64 * w);
                                   for (int l = 0; l < 64; l++) b.waves[w].dppseq[l] = 0; }
    for (int i = 0; i < n; i++) {
        Fiber& f = t_fibers[i];
        f.lin = i; f.tid.x = i % block.x; f.tid.y = (i / block.x) % block.y; f.tid.z = i / (block.x * block.y); f.done = false;
        uintptr_t top = ((uintptr_t)(t_stacks + STACK * (size_t)(i + 1))) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                 // fake return address of fiber_main (keeps rsp%16==8 at entry)
        *--sp = (void*)&fiber_main;      // popped by emu_switch's ret
        for (int r = 0; r < 6; r++) *--sp = nullptr;
        f.sp = sp;
    }
    int remaining = n; int stalls = 0;
    while (remaining) {
        unsigned long p0 = t_progress;
        for (int i = 0; i < n; i++) {
            Fiber& f = t_fibers[i];
            if (f.done) continue;
            t_cur = &f;
            emu_switch(&t_sched_sp, f.sp);
            if (f.done) remaining--;
        }
        if (t_progress == p0) { if (++stalls > 2) { fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier?\n", bx, by, bz); abort(); } }
        else stalls = 0;
    }
    t_cur = nullptr;
}

static int nworkers() {
    static int n = [] { const char* e = getenv("EMU_THREADS"); int v = e ? atoi(e) : (int)std::thread::hardware_concurrency(); return std::max(1, std::min(v, 64)); }();
    return n;
}

// Persistent worker pool: workers keep their fiber stacks (thread_local) across launches.
struct Pool {
    std::mutex m; std::condition_variable cv_job, cv_done;
    std::vector<std::thread> th;
    const std::function<void()>* job = nullptr; unsigned long job_id = 0; int pending = 0;
    Pool() {
        int n = nworkers();
        for (int i = 0; i < n; i++) th.emplace_back([this] {
            unsigned long seen = 0;
            for (;;) {
                const std::function<void()>* j;
                { std::unique_lock<std::mutex> lk(m); cv_job.wait(lk, [&] { return job_id != seen; }); seen = job_id; j = job; }
                (*j)();
                { std::lock_guard<std::mutex> lk(m); if (--pending == 0) cv_done.notify_all(); }
            }
        });
        for (auto& t : th) t.detach();
    }
    void run_all(const std::function<void()>& f) {
        std::unique_lock<std::mutex> lk(m);
        job = &f; pending = (int)th.size(); job_id++;
        cv_job.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

void run(dim3 grid, dim3 block, const std::function<void()>& body) {
    size_t total = (size_t)grid.x * grid.y * grid.z;
    if (total == 0) return;
    static Pool* pool = new Pool();
    static std::mutex launch_mutex;
    std::lock_guard<std::mutex> lg(launch_mutex);
    std::atomic<size_t> next{0};
    std::function<void()> work = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= total) break;
            unsigned bx = i % grid.x, by = (i / grid.x) % grid.y, bz = i / ((size_t)grid.x * grid.y);
            run_block(bx, by, bz, grid, block, body);
        }
    };
    pool->run_all(work);
}
}  // namespace emu

struct emu_event_t { std::chrono::steady_clock::time_point t; };
hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256); return *p ? 0 : 1; }
hipError_t hipFree(void* p) { free(p); return 0; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) { *st = nullptr; return 0; }
hipError_t hipStreamCreateWithPriority(hipStream_t* st, unsigned, int) { *st = nullptr; return 0; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return 0; }
hipError_t hipStreamDestroy(hipStream_t) { return 0; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
hipError_t hipDeviceSynchronize() { return 0; }
hipError_t hipGetLastError() { return 0; }
hipError_t hipPeekAtLastError() { return 0; }
const char* hipGetErrorString(hipError_t e) { return e ? "emu error" : "ok"; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event_t; return 0; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return 0; }
hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return 0; }
