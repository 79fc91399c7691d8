// The one collective of the path: partial counts of the ranks -> ONE all-reduce of a single uint64 over
// RCCL / xGMI.  It replaces the OpenMP reduction(+:total) of gms/algorithms/set_based/triangle_count/parallel/total.h:12 and
// k_clique_count/k_clique_count_set_based.h:25, and the `#pragma omp atomic BK_CLIQUE_COUNTER++` of
// maximal_clique_enum/sequential/tomita.h:76-77, across processes (one process per GPU).
//
// librccl is loaded on first use (dlopen), so libgmsx.so itself has no link-time dependency on it: single-GPU users and the
// CPU-side loader tests never touch it.  The message is 8 bytes — latency-bound, ring/tree choice and link bandwidth are
// irrelevant; the call is issued on the library's stream right behind the counting kernels.
#include "device_graph.hpp"

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

namespace {

// the slice of <rccl/rccl.h> this file needs (ABI-stable NCCL types)
typedef struct ncclComm *ncclComm_t;
struct ncclUniqueId { char internal[128]; };
static_assert(sizeof(ncclUniqueId) == GMSX_COMM_ID_BYTES, "GMSX_COMM_ID_BYTES must be NCCL_UNIQUE_ID_BYTES");
constexpr int kNcclSuccess = 0, kNcclSum = 0, kNcclUint64 = 5;

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.handle) break;
        }
        if (!x.handle) return x;
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.handle, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.handle, "ncclCommInitRank"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.handle, "ncclAllReduce"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.AllReduce && x.CommDestroy;
        return x;
    }();
    return r;
}

double comm_timeout_s() {
    const char *e = std::getenv("GMSX_COMM_TIMEOUT_S");
    const double v = e ? std::atof(e) : 180.0;
    return v > 0 ? v : 180.0;
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// ncclCommInitRank on a helper thread, so that the caller's wait can be bounded.  (The non-blocking creation NCCL offers for this —
// ncclCommInitRankConfig with blocking = 0, polled by ncclCommGetAsyncError — was tried first: on RCCL 2.27.7 the call itself does not return
// while a peer is missing, tools/probes/comm_timeout_probe.py.)  A job whose wait ran out stays behind with its parked thread: the caller is
// expected to leave the process (gmsx.h), and neither is touched again.
struct InitJob {
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    int result = -1, device = 0, nranks = 1, rank = 0;
    ncclUniqueId id;
    ncclComm_t comm = nullptr;
};

}  // namespace

struct gmsx_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    unsigned long long *buf = nullptr;  // device: the value being reduced
    unsigned long long *pin = nullptr;  // PINNED host words owned by the communicator: [0] the partial on its way in, [1] the sum on its way out.  The caller's
                                        // `value` is pageable (a stack word, a ctypes object): a copy queued to it behind ncclAllReduce would block the host
                                        // inside hipMemcpyAsync when a peer is dead — before the bounded poll is reached — or, if asynchronous, still be
                                        // pending into freed memory after GMSX_ERR_TIMEOUT.  With pinned words both copies are plain queue
                                        // entries, and a late DMA lands in memory that lives as long as the communicator.
    hipEvent_t done = nullptr;          // recorded behind every reduction: its wait is a bounded poll
};

using namespace gmsx;

extern "C" {

int gmsx_comm_unique_id(void *id) {
    return gmsx::guard([&]() -> int {
        if (!id) return GMSX_ERR_INVALID;
        if (!rccl().ok) return GMSX_ERR_COMM;
        ncclUniqueId u;
        if (rccl().GetUniqueId(&u) != kNcclSuccess) return GMSX_ERR_COMM;
        std::memcpy(id, &u, sizeof(u));
        return GMSX_OK;
    });
}

int gmsx_comm_init(int rank, int nranks, const void *id, gmsx_comm **out) {
    return gmsx::guard([&]() -> int {
        if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return GMSX_ERR_INVALID;
        if (int rc = ensure_init()) return rc;  // the communicator lives on the device this process is bound to (gmsx_init)
        if (!rccl().ok) return GMSX_ERR_COMM;
        gmsx_comm *c = new (std::nothrow) gmsx_comm;
        if (!c) return GMSX_ERR_NOMEM;
        c->rank = rank;
        c->nranks = nranks;
        if (hipMalloc(reinterpret_cast<void **>(&c->buf), 16) != hipSuccess) {
            (void)hipGetLastError();
            delete c;
            return GMSX_ERR_DEVICE_MEM;
        }
        if (hipHostMalloc(reinterpret_cast<void **>(&c->pin), 16, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(c->buf);
            delete c;
            return GMSX_ERR_NOMEM;
        }
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof(u));
        int rc = GMSX_OK;
        {
            auto job = std::make_shared<InitJob>();
            job->device = ctx().device;
            job->nranks = nranks;
            job->rank = rank;
            job->id = u;
            std::thread([job] {
                (void)hipSetDevice(job->device);
                ncclComm_t comm = nullptr;
                const int r = rccl().CommInitRank(&comm, job->nranks, job->id, job->rank);
                std::lock_guard<std::mutex> lock(job->m);
                job->comm = comm;
                job->result = r;
                job->done = true;
                job->cv.notify_all();
            }).detach();
            std::unique_lock<std::mutex> lock(job->m);
            if (!job->cv.wait_for(lock, std::chrono::duration<double>(comm_timeout_s()), [&] { return job->done; })) rc = GMSX_ERR_TIMEOUT;
            else if (job->result != kNcclSuccess) rc = GMSX_ERR_COMM;
            else c->comm = job->comm;
        }
        if (rc != GMSX_OK) {
            (void)hipFree(c->buf);
            (void)hipHostFree(c->pin);
            delete c;
            return rc;
        }
        *out = c;
        return GMSX_OK;
    });
}

int gmsx_comm_allreduce_u64(gmsx_comm *c, uint64_t *value) {
    return gmsx::guard([&]() -> int {
        if (!c || !value) return GMSX_ERR_INVALID;
        hipStream_t s = ctx().stream;
        c->pin[0] = *value;
        GMSX_HIP(hipMemcpyAsync(c->buf, &c->pin[0], sizeof(uint64_t), hipMemcpyHostToDevice, s));
        if (rccl().AllReduce(c->buf, c->buf, 1, kNcclUint64, kNcclSum, c->comm, s) != kNcclSuccess) return GMSX_ERR_COMM;
        GMSX_HIP(hipMemcpyAsync(&c->pin[1], c->buf, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        // the wait for the result is bounded too: a peer that died between init and the reduction must not park this rank
        if (!c->done) GMSX_HIP(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
        hipEvent_t done = c->done;
        GMSX_HIP(hipEventRecord(done, s));
        const double t0 = now_s(), limit = comm_timeout_s();
        for (unsigned spins = 0;; ++spins) {
            const hipError_t q = hipEventQuery(done);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); return GMSX_ERR_KERNEL; }
            if (spins < 20000) continue;  // an 8-byte reduction is through in tens of microseconds: no sleep on the timed path
            if (now_s() - t0 > limit) return GMSX_ERR_TIMEOUT;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        *value = c->pin[1];  // only now: the caller's word is written by the host, after the reduction is known to be through
        return GMSX_OK;
    });
}

int gmsx_comm_rank(const gmsx_comm *c) { return c ? c->rank : GMSX_ERR_INVALID; }
int gmsx_comm_size(const gmsx_comm *c) { return c ? c->nranks : GMSX_ERR_INVALID; }

int gmsx_comm_finalize(gmsx_comm *c) {
    return gmsx::guard([&]() -> int {
        if (!c) return GMSX_OK;
        int rc = GMSX_OK;
        if (c->comm && rccl().ok && rccl().CommDestroy(c->comm) != kNcclSuccess) rc = GMSX_ERR_COMM;
        (void)hipFree(c->buf);
        (void)hipHostFree(c->pin);
        if (c->done) (void)hipEventDestroy(c->done);
        delete c;
        return rc;
    });
}

}  // extern "C"
