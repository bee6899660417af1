// TEST INFRASTRUCTURE ONLY -- functional simulator of the HIP execution model for gfx950 kernels.
//
// This header shadows <hip/hip_runtime.h> when the kernel sources under
// playablevideogeneration_amd/csrc/ are compiled for the HOST with clang++ (tests/emu/build_emu.py).
// It lets the kernel index math, LDS tiling, wave-64 collectives and the MFMA fragment layouts be
// checked against the CPU oracle in a container that has no GPU.  It is never part of the shipped
// library: libcaddy_hip.so is built by hipcc from the same, unmodified sources and the product loader
// (playablevideogeneration_amd/_lib.py) only ever loads that file.
//
// Model: one OS worker thread executes one workgroup at a time; the workgroup's threads are fibers on
// that worker, scheduled round-robin; __syncthreads()/wave collectives yield until all participants
// arrive.  __shared__ maps to static thread_local storage (= one copy per worker = per running block).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <tuple>
#include <utility>
#include <functional>
#include <algorithm>

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { unsigned x, y, z; };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) double2 { double x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
typedef struct emu_stream_t* hipStream_t;
typedef struct emu_event_t* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

namespace emu {
struct Fiber {
    uint3 tid;
    unsigned lin;      // linear thread id in block
    void* sp;          // saved stack pointer
    bool done;
};
extern thread_local Fiber* t_cur;
extern thread_local uint3 t_bid;
extern thread_local dim3 t_bdim, t_gdim;

void syncthreads();
void wave_sync();
uint64_t wave_exchange(uint64_t v, int src_lane);           // returns value deposited by src_lane
void wave_gather2(float a, float b, float* A64, float* B64); // all lanes' (a,b) -> arrays
uint32_t dpp_exchange(uint32_t v, int src_lane);            // pairwise mailbox exchange (v_mov_b32_dpp): legal under divergence
void run(dim3 grid, dim3 block, const std::function<void()>& body);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
f32x16 mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int);
f32x4 mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int);
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
f32x16 mfma_f32_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c, int, int, int);
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
f32x16 mfma_f32_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c, int, int, int);
f32x4 mfma_f32_16x16x32_f16(f16x8 a, f16x8 b, f32x4 c, int, int, int);
f32x4 mfma_f32_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c, int, int, int);
void wave_gather_n(const float* mine, int n, float* all);   // all[lane*n + i]
typedef short s16x4_t __attribute__((ext_vector_type(4)));
s16x4_t ds_read_tr16_b64(const void* p);                    // gfx950 ds_read_b64_tr_b16 (lane map measured on the MI355X: tools/probes/tr_probe.hip)

template <typename... KArgs, typename... Args>
inline void launch(void (*k)(KArgs...), dim3 g, dim3 b, size_t, hipStream_t, Args&&... args) {
    std::tuple<KArgs...> tup(std::forward<Args>(args)...);
    run(g, b, [&]() { std::apply(k, tup); });
}
}  // namespace emu

#define threadIdx (emu::t_cur->tid)
#define blockIdx (emu::t_bid)
#define blockDim (emu::t_bdim)
#define gridDim (emu::t_gdim)
#define warpSize 64

#define hipLaunchKernelGGL(k, g, b, sh, st, ...) emu::launch(k, g, b, sh, st, __VA_ARGS__)
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu::mfma_f32_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu::mfma_f32_16x16x4f32
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 emu::mfma_f32_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 emu::mfma_f32_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 emu::mfma_f32_16x16x32_f16
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu::mfma_f32_16x16x32_bf16
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu::ds_read_tr16_b64((const void*)(p))
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)      /* scheduling hint only */
#define __builtin_amdgcn_sched_group_barrier(mask, size, id) ((void)0)      /* scheduling hint only */
static inline void __threadfence() {}      /* workgroups run one after the other on the simulator */

static inline void __syncthreads() { emu::syncthreads(); }

template <typename T>
static inline T __shfl(T v, int lane, int width = 64) {
    static_assert(sizeof(T) <= 8, "shfl type");
    uint64_t u = 0; memcpy(&u, &v, sizeof(T));
    int self = emu::t_cur->lin & 63;
    int base = self & ~(width - 1);
    uint64_t r = emu::wave_exchange(u, base + (lane & (width - 1)));
    T o; memcpy(&o, &r, sizeof(T)); return o;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    int self = emu::t_cur->lin & 63;
    return __shfl(v, (self ^ mask) & (width - 1), width);
}
template <typename T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int self = emu::t_cur->lin & 63;
    int l = (self & (width - 1)) + (int)d;
    if (l >= width) l = self & (width - 1);
    return __shfl(v, l, width);
}

static inline float atomicAdd(float* p, float v) {
    uint32_t* ip = (uint32_t*)p; uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw; float f;
    do { memcpy(&f, &old, 4); f += v; memcpy(&nw, &f, 4); } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4); return f;
}
static inline double atomicAdd(double* p, double v) {
    uint64_t* ip = (uint64_t*)p; uint64_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw; double f;
    do { memcpy(&f, &old, 8); f += v; memcpy(&nw, &f, 8); } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 8); return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

// v_mov_b32_dpp, quad_perm controls only (ctrl < 0x100): lane l reads lane (l & ~3) | ((ctrl >> 2 (l & 3)) & 3) of its quad
static inline int __builtin_amdgcn_mov_dpp(int v, int ctrl, int /*row_mask*/, int /*bank_mask*/, bool /*bound_ctrl*/) {
    const int self = emu::t_cur->lin & 63;
    return (int)emu::dpp_exchange((uint32_t)v, (self & ~3) | ((ctrl >> (2 * (self & 3))) & 3));
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }      /* only ever applied to wave-uniform values */
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c)); }      /* v_med3_f32 (a NaN drops out) */
using std::max;
#define __expf expf
#define __logf logf
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __ldg(const float* p) { return *p; }

// ---- host API subset used by the driver -------------------------------------------------------------------
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipStreamSynchronize(hipStream_t st);
#define hipStreamNonBlocking 1
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* st, unsigned flags, int priority);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned flags);
hipError_t hipDeviceSynchronize();
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
// graph API: the simulator has no stream capture -- hipStreamBeginCapture fails and the driver keeps its eager roll-out path
typedef struct emu_graph_t* hipGraph_t;
typedef struct emu_graph_exec_t* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
