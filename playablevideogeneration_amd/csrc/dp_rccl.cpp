// Data-parallel collectives of the CADDY step issued from C, straight into RCCL (`caddy_allreduce_grads(ctx, ncclComm_t, stream)`; 8e).
//
// The reference's only parallelism is nn.DataParallel (train.py:67-68): weights re-broadcast and 20 outputs gathered to GPU 0 every step, losses on GPU 0.  Here every
// rank (one process per GPU) owns its shard and three things are summed over the ranks with ncclAllReduce over xGMI:
//   (1) the flat fp32 gradient buffer: the dynamics / rendering ranges (91 % of 39.4 MB at BAIR-main) behind the side stream as soon as the time loop's backward is
//       done, the remaining ranges on the main stream when the backward returns (caddy_allreduce_grads);
//   (2) the K x K joint matrix of the mutual-information loss and (3) the centroid-EMA sums: a few dozen floats, stream-ordered inside the forward / loss kernels.
// RCCL is resolved at run time (dlopen of the librccl the process already carries -- PyTorch-ROCm ships one -- or the ROCm one): the library has no link-time dependency
// on it, loads without it (CPU build check, host simulator) and reports an error when a communicator is requested and no RCCL can be found.  The Python-side
// `torch.distributed` hooks (engine.enable_data_parallel) remain for gloo (CPU tests) and as a fallback.
//
// Stream discipline: ONE COMMUNICATOR PER STREAM.  `comm` carries everything issued on the context's stream (the small reductions inside the forward / loss kernels and
// the rest of the gradients), `comm2` the gradient buckets issued on the side stream during the backward.  A single communicator used from two streams is only correct
// if RCCL serialises the two streams' operations identically on every rank; with one communicator per stream each communicator sees one totally ordered sequence of
// collectives, and the host issues the calls of both in program order -- the same on every rank.  ncclCommInitRank blocks until every rank has called it: it runs in a
// helper thread with a time limit (CADDY_DP_INIT_TIMEOUT_S, default 180) so that a rank that never arrives becomes an error message, not a silent hang.
#include "net.h"
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

namespace {
struct UniqueId { char b[128]; };      // rccl.h: ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed to ncclCommInitRank BY VALUE
struct Rccl {
    void* h = nullptr; bool tried = false;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
typedef decltype(Rccl::CommInitRank) init_fn;
bool load_rccl() {
    Rccl& r = g_rccl;
    if (r.tried) return r.AllReduce != nullptr;
    r.tried = true;
    const char* names[] = {getenv("CADDY_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { if (!n) continue; r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) return false;
    r.GetUniqueId = (int (*)(void*))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (init_fn)dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (int (*)(void*))dlsym(r.h, "ncclCommDestroy");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.h, "ncclAllReduce");
    r.GetErrorString = (const char* (*)(int))dlsym(r.h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) { r.AllReduce = nullptr; return false; }
    return true;
}
constexpr int kNcclFloat32 = 7, kNcclSum = 0;      // rccl.h: ncclFloat32, ncclSum
int nccl_fail(const char* what, int rc) {
    set_error(std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error ") + " (" + std::to_string(rc) + ")");
    return -1;
}
// the small in-forward / in-loss reductions: same signature as the Python hook (head.h allreduce_hook_t), user = ctx
void small_allreduce(float* ptr, int count, void* user) {
    caddy_ctx* c = (caddy_ctx*)user;
    if (!c->comm) return;
    int rc = g_rccl.AllReduce(ptr, ptr, (size_t)count, kNcclFloat32, kNcclSum, c->comm, c->stream);
    if (rc != 0) { c->fail = true; nccl_fail("ncclAllReduce (small)", rc); }
}
// gradient buckets that become final during the backward (caddy_grads_ready_hook signature, user = ctx): the side stream's own communicator
void bucket_allreduce(float* grads, long offset, long count, void* stream, void* user) {
    caddy_ctx* c = (caddy_ctx*)user;
    if (!c->comm2) return;
    int rc = g_rccl.AllReduce(grads + offset, grads + offset, (size_t)count, kNcclFloat32, kNcclSum, c->comm2, (hipStream_t)stream);
    if (rc != 0) { c->fail = true; nccl_fail("ncclAllReduce (bucket)", rc); return; }
    c->comm_buckets.push_back({offset, count});
    c->comm_bucket_stream = (hipStream_t)stream;
}
// ncclCommInitRank with a time limit: the call blocks until EVERY rank has made it.  It runs in a helper thread that is joined when it returns; only when the limit passes is it
// detached -- it stays blocked, the caller gets an error and the process is expected to exit: there is no way to cancel a rendezvous that another rank never joins.
struct InitJob { std::mutex m; std::condition_variable cv; bool done = false; int rc = 0; void* comm = nullptr; };
int comm_init_guarded(void** out, int world_size, const UniqueId& id, int rank, const char* which) {
    int device = 0;
    hipGetDevice(&device);
    auto job = std::make_shared<InitJob>();
    const init_fn fn = g_rccl.CommInitRank;
    std::thread helper([job, fn, world_size, id, rank, device]() {
        hipSetDevice(device);      // (the helper thread must target the caller's GPU)
        void* comm = nullptr;
        const int rc = fn(&comm, world_size, id, rank);
        std::lock_guard<std::mutex> g(job->m);
        job->rc = rc; job->comm = comm; job->done = true;
        job->cv.notify_all();
    });
    const char* e = getenv("CADDY_DP_INIT_TIMEOUT_S");
    const long limit = e && atol(e) > 0 ? atol(e) : 180;
    std::unique_lock<std::mutex> lk(job->m);
    if (!job->cv.wait_for(lk, std::chrono::seconds(limit), [&] { return job->done; })) {
        helper.detach();      // still blocked in the rendezvous: nothing can cancel it, the process is expected to exit
        set_error(std::string("ncclCommInitRank (") + which + ") did not return within " + std::to_string(limit) + " s on rank " + std::to_string(rank) + " of " +
                  std::to_string(world_size) + ": another rank never reached caddy_dp_init (CADDY_DP_INIT_TIMEOUT_S)");
        return -3;
    }
    lk.unlock();
    helper.join();            // (returned: no thread outlives the call)
    if (job->rc != 0) return nccl_fail("ncclCommInitRank", job->rc);
    *out = job->comm;
    return 0;
}
}  // namespace

extern "C" {
int caddy_dp_available(void) { return load_rccl() ? 1 : 0; }
// TWO unique ids (2 x 128 bytes): one communicator per stream (see the header comment of this file)
int caddy_dp_unique_id(char* out256) {
    if (!load_rccl()) { set_error("no RCCL library found (librccl.so)"); return -1; }
    for (int i = 0; i < 2; i++) {
        char id[128]; memset(id, 0, sizeof(id));
        int rc = g_rccl.GetUniqueId(id);
        if (rc != 0) return nccl_fail("ncclGetUniqueId", rc);
        memcpy(out256 + 128 * i, id, 128);
    }
    return 0;
}
int caddy_dp_init(caddy_ctx* c, const char* id256, int world_size, int rank, int overlap) {
    if (!load_rccl()) { set_error("no RCCL library found (librccl.so)"); return -1; }
    if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    if (c->comm2) { g_rccl.CommDestroy(c->comm2); c->comm2 = nullptr; }
    UniqueId id; memcpy(id.b, id256, 128);
    void* comm = nullptr;
    int rc = comm_init_guarded(&comm, world_size, id, rank, "stream communicator");
    if (rc != 0) return rc;
    c->comm = comm; c->comm_world = world_size;
    if (overlap) {      // the gradient buckets' own communicator (side stream)
        memcpy(id.b, id256 + 128, 128);
        void* comm2 = nullptr;
        rc = comm_init_guarded(&comm2, world_size, id, rank, "bucket communicator");
        if (rc != 0) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; return rc; }
        c->comm2 = comm2;
    }
    c->hook = small_allreduce; c->hook_user = c; c->world = world_size > 1 ? world_size : 1;
    if (overlap) { c->grads_hook = bucket_allreduce; c->grads_user = c; } else { c->grads_hook = nullptr; c->grads_user = nullptr; }
    return 0;
}
int caddy_dp_shutdown(caddy_ctx* c) {
    if (c->comm || c->comm2) { hipStreamSynchronize(c->stream); if (c->side) hipStreamSynchronize(c->side); }
    if (c->comm2) { g_rccl.CommDestroy(c->comm2); c->comm2 = nullptr; }
    if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    c->hook = nullptr; c->grads_hook = nullptr; c->world = 1;
    return 0;
}
// after caddy_loss_backward: all-reduce what the buckets did not cover (E, A, state_to_hidden_state: ~9 % of the bytes) on the ctx stream and make that stream wait
// for the buckets in flight on the side stream.  The caller then runs caddy_adam_step with grad_scale = 1 / world_size.
int caddy_allreduce_grads(caddy_ctx* c) {
    if (!c->comm) { set_error("caddy_allreduce_grads: no communicator (caddy_dp_init)"); return -2; }
    std::vector<std::pair<long, long>> b = c->comm_buckets;
    std::sort(b.begin(), b.end());
    long pos = 0;
    auto reduce = [&](long lo, long hi) {
        if (hi <= lo) return 0;
        return g_rccl.AllReduce(c->G + lo, c->G + lo, (size_t)(hi - lo), kNcclFloat32, kNcclSum, c->comm, c->stream);
    };
    for (auto& r : b) { int rc = reduce(pos, r.first); if (rc) return nccl_fail("ncclAllReduce (rest)", rc); if (r.first + r.second > pos) pos = r.first + r.second; }
    { int rc = reduce(pos, c->n_train); if (rc) return nccl_fail("ncclAllReduce (rest)", rc); }
    if (!b.empty() && c->comm_bucket_stream && c->comm_bucket_stream != c->stream) {
        hipEvent_t e = c->sev();
        hipEventRecord(e, c->comm_bucket_stream);
        hipStreamWaitEvent(c->stream, e, 0);
    }
    c->comm_buckets.clear();
    return 0;
}
long caddy_dp_bucket_floats(caddy_ctx* c) { long n = 0; for (auto& r : c->comm_buckets) n += r.second; return n; }      // floats already handed to RCCL during the last backward (tests)
}
