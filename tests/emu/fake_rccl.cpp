// TEST ONLY: a stand-in for librccl.so that lets csrc/dp_rccl.cpp (the library's NATIVE data-parallel path: two communicators, bucket bookkeeping, caddy_allreduce_grads gap
// arithmetic, init watchdog) meet more than one rank on a machine without GPUs.  Loaded through CADDY_RCCL_LIB by the simulator build of the library (tests/emu); never by the
// product.  It implements the five entry points dp_rccl.cpp resolves -- ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce (fp32 sum), ncclGetErrorString -- for N
// PROCESSES of one host over a POSIX shared-memory segment named by the unique id.  The simulator executes stream work synchronously, so a collective is executed when it is
// enqueued: every rank copies its operand into its slot, all ranks meet, each rank adds the slots IN RANK ORDER (identical bits everywhere), all ranks meet again.
//
// What it checks beyond the arithmetic -- the failure modes of a real communicator that the gloo hook path cannot show:
//   * every communicator sees ONE totally ordered sequence of collectives: each call carries a sequence number and its element count; ranks that disagree (a collective issued
//     in a different order, or with a different size, on some rank) get ncclInvalidUsage instead of silently reducing unrelated buffers;
//   * a rank that never arrives is a time-out (FAKE_RCCL_TIMEOUT_S, default 60 s) -> ncclRemoteError, not a hang of the test suite;
//   * ncclCommInitRank blocks until every rank of the communicator has called it, like the real one (the watchdog of caddy_dp_init is testable).
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace {
constexpr int kMaxRanks = 16;
constexpr size_t kSlotFloats = 4u << 20;      // 16 MB per rank and pass (larger operands go through in passes); pages are only touched when used
enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4, ncclInvalidUsage = 5, ncclRemoteError = 6 };

struct Header {
    std::atomic<uint64_t> arrive;             // monotonic arrival counter of the barrier
    std::atomic<int> joined, left;
    std::atomic<int> poisoned;                // a rank detected a protocol violation: everybody fails from here on
    uint64_t seq[kMaxRanks];
    uint64_t count[kMaxRanks];
};
constexpr size_t kHeaderBytes = 4096;          // the slots start one page in
static_assert(sizeof(Header) <= kHeaderBytes, "header fits one page");

struct Comm {
    Header* h; float* slots; size_t bytes; int world, rank; uint64_t barriers, seq; char name[128];
};

double timeout_s() { const char* e = getenv("FAKE_RCCL_TIMEOUT_S"); return e && atof(e) > 0 ? atof(e) : 60.0; }

// all ranks make the same number of barrier calls on a communicator: call number b is complete when b * world arrivals have been counted
int barrier(Comm* c) {
    c->barriers++;
    c->h->arrive.fetch_add(1, std::memory_order_acq_rel);
    const uint64_t target = c->barriers * (uint64_t)c->world;
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (c->h->arrive.load(std::memory_order_acquire) < target) {
        if (c->h->poisoned.load(std::memory_order_acquire)) return ncclInvalidUsage;
        if (++spins > 2000) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return ncclRemoteError;
        } else std::this_thread::yield();
    }
    return ncclSuccess;
}
}  // namespace

extern "C" {
struct ncclUniqueId { char b[128]; };

const char* ncclGetErrorString(int rc) {
    switch (rc) {
        case ncclSuccess: return "no error";
        case ncclSystemError: return "fake RCCL: shared-memory segment could not be created / mapped";
        case ncclInvalidArgument: return "fake RCCL: invalid argument (only fp32 sum, <= 16 ranks)";
        case ncclInvalidUsage: return "fake RCCL: the ranks of a communicator issued different collectives (order or element count differs)";
        case ncclRemoteError: return "fake RCCL: a rank did not arrive in time (FAKE_RCCL_TIMEOUT_S)";
    }
    return "fake RCCL: unknown error";
}

int ncclGetUniqueId(ncclUniqueId* id) {
    static std::atomic<unsigned> n{0};
    memset(id->b, 0, sizeof(id->b));
    snprintf(id->b, sizeof(id->b), "/caddy_fake_rccl_%d_%u_%llx", (int)getpid(), n.fetch_add(1),
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

int ncclCommInitRank(void** out, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world || id.b[0] != '/') return ncclInvalidArgument;
    id.b[sizeof(id.b) - 1] = 0;
    const size_t bytes = kHeaderBytes + (size_t)world * kSlotFloats * sizeof(float);
    int fd = shm_open(id.b, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); return ncclSystemError; }      // (same size from every rank; new pages read as zero: the header starts zeroed)
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    Comm* c = new Comm{(Header*)p, (float*)((char*)p + kHeaderBytes), bytes, world, rank, 0, 0, {0}};
    snprintf(c->name, sizeof(c->name), "%s", id.b);
    c->h->joined.fetch_add(1);
    int rc = barrier(c);                      // blocks until every rank has called ncclCommInitRank (like the real rendezvous)
    if (rank == 0 || rc != ncclSuccess) shm_unlink(c->name);      // every rank holds its mapping now: drop the name so that nothing outlives the processes
    if (rc != ncclSuccess) { munmap(p, bytes); delete c; return rc; }
    *out = c;
    return ncclSuccess;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclInvalidArgument;
    c->h->left.fetch_add(1);
    munmap((void*)c->h, c->bytes);
    delete c;
    return ncclSuccess;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, void* /*stream: the simulator runs stream work at enqueue time*/) {
    Comm* c = (Comm*)comm;
    if (!c || dtype != 7 /* ncclFloat32 */ || op != 0 /* ncclSum */) return ncclInvalidArgument;
    c->seq++;
    c->h->seq[c->rank] = c->seq;
    c->h->count[c->rank] = count;
    const float* s = (const float*)send;
    float* r = (float*)recv;
    for (size_t done = 0; done < count || done == 0; done += kSlotFloats) {
        const size_t n = count - done < kSlotFloats ? count - done : kSlotFloats;
        memcpy(c->slots + (size_t)c->rank * kSlotFloats, s + done, n * sizeof(float));
        int rc = barrier(c);
        if (rc != ncclSuccess) return rc;
        for (int k = 0; k < c->world; k++)
            if (c->h->seq[k] != c->seq || c->h->count[k] != count) { c->h->poisoned.store(1); return ncclInvalidUsage; }
        for (size_t i = 0; i < n; i++) {      // rank order: bit-identical sums on every rank
            float v = c->slots[i];
            for (int k = 1; k < c->world; k++) v += c->slots[(size_t)k * kSlotFloats + i];
            r[done + i] = v;
        }
        rc = barrier(c);                      // nobody overwrites a slot before everybody has read it
        if (rc != ncclSuccess) return rc;
        if (count == 0) break;
    }
    return ncclSuccess;
}
}
