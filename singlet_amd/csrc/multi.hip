// Native multi-GPU path: cells sharded over the ranks of a TEAM, the per-iteration exchange done by
// the library itself over RCCL (xGMI), no host-side hook.
//
// The reference has no distribution; its nearest relative is the column-chunk list with a running
// column offset (src/singlet.cpp:384-402).  Here rank r owns a contiguous block of
// cells: A[:, blk_r], its transpose, H[:, blk_r]; W, its Gram and d are replicated.
//
// One ALS iteration on a team:
//   every rank   h = predict(A_r, w)                                   local, src/singlet.cpp:650
//                partials of the UNSCALED h:  B' = h A_r^T (k x m),  G' = h h^T (k x k),  s = rowsums(h) (k)
//   exchange 1   ONE grouped collective:  reduce-scatter of B' by gene blocks  +  all-reduce of [G' | s]
//                (unscaled partials commute with scale(h, d): with D = diag(s + 1e-15) the reference's
//                operands are D^-1 B', D^-1 G' D^-1 -- same sums, rounding order differs by ~1 ulp)
//   every rank   h /= D;  its gene block:  b = D^-1 B',  G = D^-1 G' D^-1 + 1e-15 I,  nnls -> w[:, block]   :654
//   exchange 2   all-gather of the w blocks (reduce-scatter + all-gather move the bytes of one all-reduce,
//                and the m solves are dealt out instead of replicated)
//   every rank   scale(w, d); tol = cor(w, w_prev)                      replicated, bit-identical   :655-659
//
// Two ways to form a team:
//   * sgl_multi_create: ONE process drives all devices (the R host: no launcher), communicators from
//     ncclCommInitAll, collectives issued for all ranks inside ncclGroupStart / ncclGroupEnd;
//   * sgl_comm_init_rank: one process per GPU (bench.py under torch.distributed.run), communicator from
//     ncclCommInitRank with an id the host broadcast.
// RCCL is bound at run time (dlopen): single-GPU users need no RCCL.  Ranks that share ONE device inside
// one process cannot form an RCCL communicator (duplicate GPU); they exchange through a HIP kernel
// that sums in rank order ("loopback") -- that is how the team logic is tested on a 1-GPU box.
#include "sgl_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ---------------------------------------------------------------- RCCL binding --
struct RcclApi {
    void* handle = nullptr;
    char path[512] = "";
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional: a build without it can only destroy
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi* rccl_api() {
    // bound once, whichever host thread comes first (the replica sweep runs one thread per device)
    static RcclApi api;
    static char why[256] = "not found";
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("SGL_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) { snprintf(api.path, sizeof(api.path), "%s", n); break; }
            const char* e = dlerror();   // one call: it clears the message
            if (e) snprintf(why, sizeof(why), "%s", e);
        }
        if (api.handle) {
            bool ok = true;
            auto bind = [&](const char* sym) {
                void* f = dlsym(api.handle, sym);
                if (!f) { ok = false; snprintf(why, sizeof(why), "symbol %s missing in %s", sym, api.path); }
                return f;
            };
            api.GetUniqueId = (decltype(api.GetUniqueId))bind("ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))bind("ncclCommInitRank");
            api.CommInitAll = (decltype(api.CommInitAll))bind("ncclCommInitAll");
            api.CommDestroy = (decltype(api.CommDestroy))bind("ncclCommDestroy");
            api.CommAbort = (decltype(api.CommAbort))dlsym(api.handle, "ncclCommAbort");
            api.CommCount = (decltype(api.CommCount))bind("ncclCommCount");
            api.AllReduce = (decltype(api.AllReduce))bind("ncclAllReduce");
            api.ReduceScatter = (decltype(api.ReduceScatter))bind("ncclReduceScatter");
            api.AllGather = (decltype(api.AllGather))bind("ncclAllGather");
            api.GroupStart = (decltype(api.GroupStart))bind("ncclGroupStart");
            api.GroupEnd = (decltype(api.GroupEnd))bind("ncclGroupEnd");
            api.GetErrorString = (decltype(api.GetErrorString))bind("ncclGetErrorString");
            if (!ok) { dlclose(api.handle); api.handle = nullptr; }
        }
    });
    if (!api.handle) { sgl_set_error("RCCL (librccl.so.1) could not be loaded: %s", why); return nullptr; }
    return &api;
}

#define NCCLCHK(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t r__ = (expr);                                                                 \
        if (r__ != ncclSuccess) {                                                                  \
            sgl_set_error("%s failed: %s (%s:%d)", #expr, rccl_api()->GetErrorString(r__), __FILE__, __LINE__); \
            return SGL_ECOMM;                                                                      \
        }                                                                                          \
    } while (0)

// ----------------------------------------------------------------------- team --
#define SGL_TEAM_MAX 16

// One host thread per local rank of a single-process team (sgl_multi).  A team iteration is ~50 kernel launches
// per rank; issued from ONE thread for 8 devices the last rank's first kernel would start ~1 ms after the first
// rank's, and every collective waits for the slowest -- at the 8-GPU shard size (4.5 ms per iteration) that is the
// difference between 6 x and 5 x.  The workers are persistent (one fork-join per library call), each binds its
// device once, and in RCCL mode they never wait for one another on the host: every rank issues its own collectives
// on its own communicator (the documented one-thread-per-device use of communicators from ncclCommInitAll).
// SGL_MULTI_SERIAL=1 keeps the single-threaded drive (grouped collectives) for A/B runs and as a fallback.
struct TeamPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done, cv_bar;
    std::function<int(int)> job;
    uint64_t gen = 0;
    int pending = 0;
    bool stop = false;
    std::vector<int> rc;
    std::vector<std::string> err;
    // host barrier of the workers (loopback exchange only); `failed` releases everybody when a rank has given up
    int bar_count = 0;
    uint64_t bar_gen = 0;
    std::atomic<bool> failed{false};
    // RCCL drive: exchange steps enqueued by any worker during the running call.  A rank that fails while a peer has a
    // collective in flight (or is about to enqueue one) would leave that peer blocked for ever in its next host wait --
    // the kernels of a collective only return when every rank has joined -- so the failing worker aborts the team's
    // communicators (team_abort): `enq` and `failed` are read / written in the order that closes the window (team_exchange).
    std::atomic<int> enq{0};
};

struct sgl_team {
    TeamPool* pool = nullptr;      // nullptr: the calling thread drives every local rank in turn
    int nranks = 1;
    std::vector<sgl_ctx*> local;   // the ranks this process drives
    std::vector<int> rank;         // team rank of local[i]
    std::vector<ncclComm_t> comm;  // per local rank; empty in loopback mode
    // comm[i] is used by rank i's worker thread only -- except when a failure makes another thread abort it: the owner holds
    // comm_mu[i] while it enqueues on it (microseconds: RCCL's launch does not wait for the peers), the aborting thread takes it
    std::vector<std::unique_ptr<std::timed_mutex>> comm_mu;
    std::atomic<bool> broken{false};   // communicators aborted after a failure: every later call on the team is refused
    bool loopback = false;
    std::vector<hipEvent_t> ev;    // loopback: stream ordering
    hipEvent_t done = nullptr;
    bool owns_ctx = false;         // sgl_multi_create made the contexts (sgl_multi_destroy frees them)
    int32_t nrow = 0;
    int64_t ncells_total = 0;
    std::vector<int64_t> cell_lo;  // single-process form: cell block boundaries (nranks + 1)
};
struct sgl_multi : sgl_team {};

int sgl_team_size(const sgl_ctx* c) { return (c && c->team) ? c->team->nranks : 1; }

void sgl_team_detach(sgl_ctx* c) {
    sgl_team* T = c->team;
    if (!T) return;
    for (size_t i = 0; i < T->local.size(); ++i)
        if (T->local[i] == c) {
            // (a broken team whose librccl cannot abort keeps its communicators: destroying one would wait for the very
            //  collectives that never complete -- leaked on purpose, see team_abort)
            if (!T->loopback && i < T->comm.size() && T->comm[i] && rccl_api() && !(T->broken.load() && !rccl_api()->CommAbort))
                (void)rccl_api()->CommDestroy(T->comm[i]);
            T->local.erase(T->local.begin() + i);
            T->rank.erase(T->rank.begin() + i);
            if (i < T->comm.size()) T->comm.erase(T->comm.begin() + i);
            if (i < T->comm_mu.size()) T->comm_mu.erase(T->comm_mu.begin() + i);
            break;
        }
    c->team = nullptr;
    if (T->local.empty() && !T->owns_ctx) {  // a team made by sgl_comm_init_rank dies with its context
        for (auto e : T->ev) (void)hipEventDestroy(e);
        if (T->done) (void)hipEventDestroy(T->done);
        delete T;
    }
}

// A rank has failed (or a call has timed out) while collectives may be in flight: abort every local communicator, which
// makes the kernels of the pending collectives return on all devices, so that the peers' host waits (stream
// synchronisation, implicit synchronisation inside hipMalloc / hipFree ...) come back and their next exchange step is
// refused (`broken`).  Once per team; the team cannot be used afterwards, only destroyed.
static void team_abort(sgl_team* T) {
    bool expected = false;
    if (!T->broken.compare_exchange_strong(expected, true)) return;
    if (T->loopback || T->comm.empty()) return;
    RcclApi* R = rccl_api();
    if (!R) return;
    // A librccl without ncclCommAbort can only destroy, and ncclCommDestroy WAITS for the collectives in flight -- the very ones
    // this abort is meant to break: then nothing is torn down here; the team is marked broken (every later call is refused with
    // SGL_ECOMM) and the communicators stay as they are until sgl_multi_destroy (round-5 advice).
    if (!R->CommAbort) return;
    for (size_t i = 0; i < T->comm.size(); ++i) {
        std::unique_lock<std::timed_mutex> lk;
        if (i < T->comm_mu.size() && T->comm_mu[i]) {
            lk = std::unique_lock<std::timed_mutex>(*T->comm_mu[i], std::defer_lock);
            // the owner holds its mutex only while it enqueues (microseconds); one that is stuck INSIDE an enqueue keeps its
            // communicator -- aborting it under the owner's feet would race with the enqueue -- and the abort of its peers'
            // communicators is what releases it
            if (!lk.try_lock_for(std::chrono::seconds(2))) continue;
        }
        if (T->comm[i]) {
            (void)R->CommAbort(T->comm[i]);
            T->comm[i] = nullptr;
        }
    }
}

static int team_refused(const sgl_team* T) {
    if (T->broken.load()) { sgl_set_error("team: the communicators were aborted after a failure on one rank; destroy the team"); return SGL_ECOMM; }
    return SGL_OK;
}

// ------------------------------------------------------------ loopback kernels --
struct PtrPack {
    void* p[SGL_TEAM_MAX];
};
template <typename T_>
__global__ void lb_allreduce_kernel(PtrPack bufs, int n, int64_t count) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        T_ s = static_cast<T_*>(bufs.p[0])[e];
        for (int r = 1; r < n; ++r) s += static_cast<T_*>(bufs.p[r])[e];
        for (int r = 0; r < n; ++r) static_cast<T_*>(bufs.p[r])[e] = s;
    }
}
// block r (cnt elements at offset r * cnt) of rank r's buffer = sum over the ranks of their block r
__global__ void lb_reduce_scatter_kernel(PtrPack bufs, int n, int64_t cnt) {
    const int64_t total = cnt * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / cnt);
        double s = static_cast<double*>(bufs.p[0])[e];
        for (int q = 1; q < n; ++q) s += static_cast<double*>(bufs.p[q])[e];
        static_cast<double*>(bufs.p[r])[e] = s;
    }
}
__global__ void lb_allgather_kernel(PtrPack bufs, int n, int64_t cnt) {
    const int64_t total = cnt * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / cnt);
        const double v = static_cast<double*>(bufs.p[r])[e];
        for (int q = 0; q < n; ++q)
            if (q != r) static_cast<double*>(bufs.p[q])[e] = v;
    }
}

// loopback: make stream 0 wait for all ranks' streams, run `launch` there, make the others wait for it
template <typename F>
static int lb_run(sgl_team* T, F&& launch) {
    const int n = (int)T->local.size();
    for (int i = 0; i < n; ++i) HIPCHK(hipEventRecord(T->ev[i], T->local[i]->stream));
    for (int i = 1; i < n; ++i) HIPCHK(hipStreamWaitEvent(T->local[0]->stream, T->ev[i], 0));
    launch(T->local[0]->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(T->done, T->local[0]->stream));
    for (int i = 1; i < n; ++i) HIPCHK(hipStreamWaitEvent(T->local[i]->stream, T->done, 0));
    return SGL_OK;
}

static unsigned lb_blocks(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 2048)); }

// ------------------------------------------------------------ worker threads --
static void pool_worker(sgl_team* T, int i) {
    TeamPool* P = T->pool;
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(P->mu);
            P->cv_go.wait(lk, [&] { return P->stop || P->gen != seen; });
            if (P->stop) return;
            seen = P->gen;
        }
        int rc = SGL_OK;
        if (hipSetDevice(T->local[i]->device) != hipSuccess) { sgl_set_error("team worker %d: hipSetDevice failed", i); rc = SGL_EHIP; }
        if (rc == SGL_OK) rc = P->job(i);   // set before the generation moved, untouched until every worker is back
        {
            std::lock_guard<std::mutex> lk(P->mu);
            P->rc[i] = rc;
            if (rc != SGL_OK) {
                P->err[i] = sgl_last_error();
                P->failed = true;          // releases ranks waiting in team_barrier; no worker enqueues an exchange step any more
                P->cv_bar.notify_all();
            }
        }
        // RCCL drive: a peer that has enqueued (or was just about to enqueue) an exchange step of this call waits on the
        // device for this rank, which will never join: abort the communicators so that its host wait returns
        if (rc != SGL_OK && !T->loopback && P->enq.load() > 0) team_abort(T);
        {
            std::lock_guard<std::mutex> lk(P->mu);
            if (--P->pending == 0) P->cv_done.notify_one();
        }
    }
}

static void pool_stop(sgl_team* T) {
    TeamPool* P = T->pool;
    if (!P) return;
    {
        std::lock_guard<std::mutex> lk(P->mu);
        P->stop = true;
    }
    P->cv_go.notify_all();
    for (auto& t : P->th) t.join();
    delete P;
    T->pool = nullptr;
}

static int pool_start(sgl_team* T) {
    const int n = (int)T->local.size();
    if (n < 2 || getenv("SGL_MULTI_SERIAL")) return SGL_OK;
    TeamPool* P = new (std::nothrow) TeamPool();
    if (!P) { sgl_set_error("out of host memory"); return SGL_ENOMEM; }
    P->rc.assign(n, SGL_OK);
    P->err.assign(n, std::string());
    T->pool = P;
    try {
        for (int i = 0; i < n; ++i) P->th.emplace_back(pool_worker, T, i);
    } catch (...) {
        pool_stop(T);
        sgl_set_error("team: could not start the worker threads");
        return SGL_ENOMEM;
    }
    return SGL_OK;
}

// fn(i) for every local rank: on the rank's worker thread (all at once) or, without a pool, one after the other on
// the calling thread.  Only for work whose ranks do not wait for one another on the HOST.
template <typename F>
static int team_parallel(sgl_team* T, F&& fn) {
    const int n = (int)T->local.size();
    TeamPool* P = T->pool;
    SGLCHK(team_refused(T));
    if (!P) {
        for (int i = 0; i < n; ++i) {
            HIPCHK(hipSetDevice(T->local[i]->device));
            SGLCHK(fn(i));
        }
        return SGL_OK;
    }
    bool timed_out = false;
    {
        std::unique_lock<std::mutex> lk(P->mu);
        P->job = [&fn](int i) -> int { return fn(i); };
        P->failed = false;
        P->enq = 0;
        P->bar_count = 0;
        std::fill(P->rc.begin(), P->rc.end(), (int)SGL_OK);
        P->pending = n;
        ++P->gen;
        P->cv_go.notify_all();
        // Watchdog: a library call on a team is seconds at most (an iteration is milliseconds, the set-up of a fit seconds).
        // When the workers are not back after SGL_TEAM_TIMEOUT_S (default 600 s, 0 = wait for ever) something is blocked for
        // good -- a rank that never joined a collective -- and the call gives up instead of hanging its host: the ranks
        // waiting at the host barrier are released, the communicators aborted (their kernels return), and the call reports
        // SGL_ECOMM once the workers are back.
        const char* te = getenv("SGL_TEAM_TIMEOUT_S");
        const double tmo = te ? atof(te) : 600.0;
        if (tmo > 0 && !P->cv_done.wait_for(lk, std::chrono::duration<double>(tmo), [&] { return P->pending == 0; })) {
            timed_out = true;
            P->failed = true;
            P->cv_bar.notify_all();
            lk.unlock();
            team_abort(T);
            lk.lock();
        }
        // (this second wait has no bound on purpose: the job closes over the caller's stack, so the call cannot return while a
        //  worker may still run it; after the abort above every device-side wait of the workers has ended)
        P->cv_done.wait(lk, [&] { return P->pending == 0; });
        P->job = nullptr;
    }
    if (timed_out) {
        sgl_set_error("team: the call did not come back within SGL_TEAM_TIMEOUT_S; ranks released, communicators aborted");
        return SGL_ECOMM;
    }
    for (int i = 0; i < n; ++i)
        if (P->rc[i] != SGL_OK) {
            sgl_set_error("%s", P->err[i].c_str());
            return P->rc[i];
        }
    return SGL_OK;
}

// host barrier of the worker threads (the loopback exchange is one kernel launched by rank 0's thread once every
// rank has enqueued its part); returns SGL_ECOMM when another rank has failed instead of waiting for it for ever
static int team_barrier(sgl_team* T) {
    TeamPool* P = T->pool;
    if (!P) return SGL_OK;
    std::unique_lock<std::mutex> lk(P->mu);
    if (!P->failed) {
        const uint64_t g = P->bar_gen;
        if (++P->bar_count == (int)T->local.size()) {
            P->bar_count = 0;
            ++P->bar_gen;
            P->cv_bar.notify_all();
            return SGL_OK;
        }
        P->cv_bar.wait(lk, [&] { return P->bar_gen != g || P->failed; });
        if (P->bar_gen != g) return SGL_OK;
    }
    sgl_set_error("team: another rank failed");
    return SGL_ECOMM;
}

// ------------------------------------------------------------------ collectives --
// Each takes one buffer per LOCAL rank (index i = T->local[i]) and issues the call for the local ranks [i0, i1);
// all are in place.  Loopback: one kernel for the whole team, [i0, i1) must be all local ranks.
static PtrPack pack(sgl_team* T, void* const* bufs) {
    PtrPack P;
    for (int i = 0; i < SGL_TEAM_MAX; ++i) P.p[i] = nullptr;
    for (size_t i = 0; i < T->local.size(); ++i) P.p[T->rank[i]] = bufs[i];
    return P;
}

// sum of `count` doubles (or int64) over the ranks
static int team_allreduce(sgl_team* T, int i0, int i1, void* const* bufs, int64_t count, bool i64 = false) {
    if (count <= 0) return SGL_OK;
    if (T->loopback) {
        const PtrPack P = pack(T, bufs);
        const int n = T->nranks;
        return lb_run(T, [&](hipStream_t s) {
            if (i64) lb_allreduce_kernel<long long><<<dim3(lb_blocks(count)), dim3(256), 0, s>>>(P, n, count);
            else lb_allreduce_kernel<double><<<dim3(lb_blocks(count)), dim3(256), 0, s>>>(P, n, count);
        });
    }
    RcclApi* R = rccl_api();
    if (!R) return SGL_ECOMM;
    for (int i = i0; i < i1; ++i) {
        HIPCHK(hipSetDevice(T->local[i]->device));
        NCCLCHK(R->AllReduce(bufs[i], bufs[i], (size_t)count, i64 ? ncclInt64 : ncclDouble, ncclSum, T->comm[i], T->local[i]->stream));
    }
    return SGL_OK;
}

// block r (cnt doubles at offset r * cnt) of rank r's buffer = sum over the ranks of their block r
static int team_reduce_scatter(sgl_team* T, int i0, int i1, void* const* bufs, int64_t cnt) {
    if (cnt <= 0) return SGL_OK;
    if (T->loopback) {
        const PtrPack P = pack(T, bufs);
        const int n = T->nranks;
        return lb_run(T, [&](hipStream_t s) { lb_reduce_scatter_kernel<<<dim3(lb_blocks(cnt * n)), dim3(256), 0, s>>>(P, n, cnt); });
    }
    RcclApi* R = rccl_api();
    if (!R) return SGL_ECOMM;
    for (int i = i0; i < i1; ++i) {
        HIPCHK(hipSetDevice(T->local[i]->device));
        double* b = static_cast<double*>(bufs[i]);
        NCCLCHK(R->ReduceScatter(b, b + (size_t)T->rank[i] * cnt, (size_t)cnt, ncclDouble, ncclSum, T->comm[i], T->local[i]->stream));
    }
    return SGL_OK;
}

static int team_allgather(sgl_team* T, int i0, int i1, void* const* bufs, int64_t cnt) {
    if (cnt <= 0) return SGL_OK;
    if (T->loopback) {
        const PtrPack P = pack(T, bufs);
        const int n = T->nranks;
        return lb_run(T, [&](hipStream_t s) { lb_allgather_kernel<<<dim3(lb_blocks(cnt * n)), dim3(256), 0, s>>>(P, n, cnt); });
    }
    RcclApi* R = rccl_api();
    if (!R) return SGL_ECOMM;
    for (int i = i0; i < i1; ++i) {
        HIPCHK(hipSetDevice(T->local[i]->device));
        double* b = static_cast<double*>(bufs[i]);
        NCCLCHK(R->AllGather(b + (size_t)T->rank[i] * cnt, b, (size_t)cnt, ncclDouble, T->comm[i], T->local[i]->stream));
    }
    return SGL_OK;
}

static int group_begin(sgl_team* T) {
    if (T->loopback) return SGL_OK;
    RcclApi* R = rccl_api();
    if (!R) return SGL_ECOMM;
    NCCLCHK(R->GroupStart());
    return SGL_OK;
}
static int group_end(sgl_team* T) {
    if (T->loopback) return SGL_OK;
    NCCLCHK(rccl_api()->GroupEnd());
    return SGL_OK;
}

// One exchange step = a few collectives issued as ONE RCCL group, timed as SGL_PH_COMM.
//   who < 0  : the calling thread issues it for every local rank (single-threaded drive, one process per GPU);
//   who = i  : worker thread i issues rank i's part.  RCCL: its own group on its own communicator, no host
//              rendez-vous (the kernels meet on the devices).  Loopback: the workers meet at a host barrier, rank 0's
//              thread launches the summing kernel for the team, a second barrier lets the others go on.
struct Xfer {
    int kind;                        // 0 all-reduce, 1 reduce-scatter, 2 all-gather
    const std::vector<void*>* bufs;  // one buffer per local rank
    int64_t cnt;
    bool i64;
};
static int xfer_issue(sgl_team* T, int i0, int i1, const Xfer& x) {
    switch (x.kind) {
        case 0: return team_allreduce(T, i0, i1, x.bufs->data(), x.cnt, x.i64);
        case 1: return team_reduce_scatter(T, i0, i1, x.bufs->data(), x.cnt);
        default: return team_allgather(T, i0, i1, x.bufs->data(), x.cnt);
    }
}
static int team_exchange(sgl_team* T, int who, std::initializer_list<Xfer> ops) {
    const int nl = (int)T->local.size();
    SGLCHK(team_refused(T));
    if (who < 0) {
        std::vector<PhaseEvent> pe(nl);
        for (int i = 0; i < nl; ++i) {
            HIPCHK(hipSetDevice(T->local[i]->device));
            SGLCHK(sgl_phase_begin(T->local[i], SGL_PH_COMM, &pe[i]));
        }
        int rc = group_begin(T);
        if (rc == SGL_OK) {
            for (const Xfer& x : ops)
                if (rc == SGL_OK) rc = xfer_issue(T, 0, nl, x);
            const int rc_end = group_end(T);
            if (rc == SGL_OK) rc = rc_end;
        }
        for (int i = 0; i < nl; ++i) {   // the phases end on every exit
            if (hipSetDevice(T->local[i]->device) != hipSuccess && rc == SGL_OK) { sgl_set_error("team: hipSetDevice failed"); rc = SGL_EHIP; }
            const int rc_pe = sgl_phase_end(T->local[i], &pe[i]);
            if (rc == SGL_OK) rc = rc_pe;
        }
        return rc;
    }
    sgl_ctx* c = T->local[who];
    HIPCHK(hipSetDevice(c->device));
    {   // test hook (tests/test_gpu_native_team.py): SGL_TEAM_TEST_STALL="rank:seconds" -- that rank's worker sleeps in front of
        // every exchange step, as a rank blocked on the host would; read once per process
        static const struct Stall { int rank = -1; double sec = 0; Stall() { if (const char* e = getenv("SGL_TEAM_TEST_STALL")) (void)sscanf(e, "%d:%lf", &rank, &sec); } } stall;
        if (stall.rank == T->rank[who] && stall.sec > 0) std::this_thread::sleep_for(std::chrono::duration<double>(stall.sec));
    }
    PhaseEvent pe;
    SGLCHK(sgl_phase_begin(c, SGL_PH_COMM, &pe));
    // (one exit: whatever fails below, the phase's two events go back to the context -- round-5 advice)
    int rc = SGL_OK;
    if (T->loopback) {
        rc = team_barrier(T);
        if (rc == SGL_OK && who == 0) {
            for (const Xfer& x : ops)
                if (rc == SGL_OK) rc = xfer_issue(T, 0, nl, x);
            if (rc == SGL_OK && hipSetDevice(c->device) != hipSuccess) { sgl_set_error("team: hipSetDevice failed"); rc = SGL_EHIP; }
        }
        if (rc == SGL_OK) rc = team_barrier(T);
    } else {
        // enq first, failed second (the failing worker writes failed first, reads enq second): either this rank sees the
        // failure and enqueues nothing, or the failing rank sees the pending step and aborts the communicators
        TeamPool* P = T->pool;
        if (P) {
            P->enq.fetch_add(1);
            if (P->failed.load()) { sgl_set_error("team: another rank failed"); rc = SGL_ECOMM; }
        }
        if (rc == SGL_OK) {
            std::unique_lock<std::timed_mutex> lk;
            if ((size_t)who < T->comm_mu.size() && T->comm_mu[who]) lk = std::unique_lock<std::timed_mutex>(*T->comm_mu[who]);
            rc = team_refused(T);
            if (rc == SGL_OK) rc = group_begin(T);
            if (rc == SGL_OK) {
                for (const Xfer& x : ops)
                    if (rc == SGL_OK) rc = xfer_issue(T, who, who + 1, x);
                const int rc_end = group_end(T);   // a group that was begun is always ended
                if (rc == SGL_OK) rc = rc_end;
            }
        }
    }
    const int rc_pe = sgl_phase_end(c, &pe);
    return rc != SGL_OK ? rc : rc_pe;
}

// ---------------------------------------------------------------- team set-up --
static int team_events(sgl_team* T) {
    if (!T->loopback) return SGL_OK;
    HIPCHK(hipSetDevice(T->local[0]->device));
    T->ev.resize(T->local.size());
    for (auto& e : T->ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&T->done, hipEventDisableTiming));
    return SGL_OK;
}

// per-gene non-zero counts over all ranks (which W columns predict() skips, src/singlet.cpp:340)
static int team_gene_counts(sgl_team* T) {
    if (T->nranks <= 1) return SGL_OK;
    const int nl = (int)T->local.size();
    std::vector<void*> bufs(nl);
    auto prep = [&](int i) -> int {
        sgl_ctx* c = T->local[i];
        HIPCHK(hipSetDevice(c->device));
        const int64_t m = c->A.nrow;
        if (!c->col_nnz_At_global) {
            hipError_t e = sgl_pool_malloc((void**)&c->col_nnz_At_global, sizeof(int64_t) * (size_t)std::max<int64_t>(m, 1));
            if (e != hipSuccess) { (void)hipGetLastError(); sgl_set_error("team: out of device memory"); return SGL_ENOMEM; }
        }
        HIPCHK(hipMemcpyAsync(c->col_nnz_At_global, c->col_nnz_At, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToDevice, c->stream));
        bufs[i] = c->col_nnz_At_global;
        return SGL_OK;
    };
    if (T->pool) {
        SGLCHK(team_parallel(T, [&](int i) -> int {
            SGLCHK(prep(i));
            return team_exchange(T, i, {{0, &bufs, (int64_t)T->nrow, true}});
        }));
    } else {
        for (int i = 0; i < nl; ++i) SGLCHK(prep(i));
        SGLCHK(team_exchange(T, -1, {{0, &bufs, (int64_t)T->nrow, true}}));
    }
    for (auto c : T->local) c->gene_nnz_global = true;
    return SGL_OK;
}

#define TEAM_GUARD(T)                                                                                 \
    do {                                                                                              \
        if ((T) == nullptr || (T)->local.empty()) { sgl_set_error("null or empty team"); return SGL_EINVAL; } \
        SGLCHK(team_refused(T));                                                                      \
    } while (0)

// W is replicated, so tol = cor(w, w_prev) must come out bit-identical on every local rank: checked on every
// iteration (a few compares) -- a team whose replicas have drifted apart must not go on silently.
static int team_tols_agree(const std::vector<double>& tols) {
    for (size_t i = 1; i < tols.size(); ++i)
        if (memcmp(&tols[i], &tols[0], sizeof(double)) != 0) {
            sgl_set_error("team: rank %zu computed tol %.17g, rank 0 %.17g -- the replicated W differs between ranks", i, tols[i], tols[0]);
            return SGL_ECOMM;
        }
    return SGL_OK;
}

// -------------------------------------------------------------- the iteration --
// One ALS iteration over the local ranks of the team (header comment).  Every enqueue of every rank
// is asynchronous; the only host wait is the final read of tol.  With worker threads (TeamPool) every rank runs
// its whole sequence -- local part, exchange 1, its gene block, exchange 2, scale + cor, read tol -- on its own
// thread; without, the calling thread walks the ranks phase by phase and groups the collectives.
static int team_iterate(sgl_team* T, double L1_w, double L1_h, double L2_w, double L2_h, double* tol_out) {
    const int nl = (int)T->local.size();
    const int N = T->nranks;
    for (auto c : T->local) {
        if (c->k == 0) { sgl_set_error("team: no fit initialised"); return SGL_ESTATE; }
        // c_linked_nmf on a team (src/singlet.cpp:1059-1086): link_h holds the columns of this rank's cells and is applied
        // inside the local H-update (sgl_step_h); link_w multiplies the rank's gene block of the summed right-hand sides
    }
    const int k = T->local[0]->k;
    const int64_t m = T->nrow;
    const int64_t mb = N > 1 ? (m + N - 1) / N : m;       // genes per rank block
    const int64_t mpad = mb * N;
    std::vector<void*> red(nl), tail(nl), wbuf(nl);
    std::vector<double> tols(nl, 0.0);
    auto local_part = [&](int i) -> int {   // h = predict(A_r, w); partials of the unscaled h
        sgl_ctx* c = T->local[i];
        HIPCHK(hipSetDevice(c->device));
        SGLCHK(sgl_step_begin(c));
        SGLCHK(sgl_step_h(c, L1_h, L2_h));
        double* Bw = c->red;
        double* Gh = c->red + (size_t)k * mpad;
        double* sh = Gh + (size_t)k * k;
        PhaseEvent pe;
        SGLCHK(sgl_phase_begin(c, SGL_PH_SCALE, &pe));
        SGLCHK(k_rowsum(c, c->H, k, c->A.ncol, sh));
        SGLCHK(sgl_phase_end(c, &pe));
        SGLCHK(sgl_phase_begin(c, SGL_PH_RHS_W, &pe));
        if (c->use_tiled && c->TAt.roff) SGLCHK(k_acc_tiled_all(c->stream, c->TAt, c->H, Bw, k));
        else SGLCHK(k_acc(c->stream, c->At, c->H, k, Bw, 0, 1, 0, 0, 0));
        SGLCHK(sgl_phase_end(c, &pe));
        SGLCHK(sgl_phase_begin(c, SGL_PH_GRAM, &pe));
        SGLCHK(k_gram(c, c->H, k, c->A.ncol, Gh, 0.0));
        SGLCHK(sgl_phase_end(c, &pe));
        red[i] = Bw;
        tail[i] = Gh;
        wbuf[i] = c->W;
        return SGL_OK;
    };
    // exchange 1: one grouped collective
    auto exchange1 = [&](int who) -> int {
        return team_exchange(T, who, {{1, &red, (int64_t)k * mb, false}, {0, &tail, (int64_t)k * k + k, false}});
    };
    auto gene_block = [&](int i) -> int {
        sgl_ctx* c = T->local[i];
        HIPCHK(hipSetDevice(c->device));
        const int r = T->rank[i];
        double* Gh = c->red + (size_t)k * mpad;
        double* sh = Gh + (size_t)k * k;
        PhaseEvent pe;
        SGLCHK(sgl_phase_begin(c, SGL_PH_SCALE, &pe));
        // d = rowsums + 1e-15; h /= d  (scale(h, d), src/singlet.cpp:219-225, with the GLOBAL row sums)
        HIPCHK(hipMemcpyAsync(c->d, sh, sizeof(double) * k, hipMemcpyDeviceToDevice, c->stream));
        SGLCHK(k_scale_apply(c->stream, c->H, k, c->A.ncol, c->d, 1));
        SGLCHK(sgl_phase_end(c, &pe));
        const int64_t g0 = (int64_t)r * mb;
        const int64_t ng = std::max<int64_t>(0, std::min<int64_t>(mb, m - g0));
        SGLCHK(sgl_phase_begin(c, SGL_PH_GRAM, &pe));
        HIPCHK(hipMemcpyAsync(c->G, Gh, sizeof(double) * k * k, hipMemcpyDeviceToDevice, c->stream));
        SGLCHK(k_gram_rescale(c->stream, c->G, k, c->d, 1e-15));   // AAt of the scaled h (+1e-15, l.204)
        SGLCHK(sgl_phase_end(c, &pe));
        SGLCHK(sgl_phase_begin(c, SGL_PH_NNLS_W, &pe));
        if (ng > 0) {
            double* Bblk = c->red + (size_t)g0 * k;
            SGLCHK(k_scale_apply(c->stream, Bblk, k, ng, c->d, 0));
            if (c->link_w) SGLCHK(k_link_mul(c->stream, Bblk, c->link_w + (size_t)g0 * c->link_w_rows, k, c->link_w_rows, ng));  // predict_link l.429-430
            const int64_t* gene_nnz = (N > 1) ? c->col_nnz_At_global : c->col_nnz_At;
            if (N > 1 && !c->gene_nnz_global) { sgl_set_error("team: global gene counts missing"); return SGL_ESTATE; }
            // (the dense front-end solves every column, src/singlet.cpp:370-381: no skip list)
            SGLCHK(sgl_nnls_shared(c, c->G, Bblk, c->W + (size_t)g0 * k, c->solve_empty ? nullptr : gene_nnz + g0, ng, L1_w, L2_w, c->sweep_counters + 1));
        }
        SGLCHK(sgl_phase_end(c, &pe));
        return SGL_OK;
    };
    auto exchange2 = [&](int who) -> int {   // all-gather of the solved w blocks
        if (N <= 1) return SGL_OK;
        return team_exchange(T, who, {{2, &wbuf, (int64_t)k * mb, false}});
    };
    auto tail_part = [&](int i) -> int {
        HIPCHK(hipSetDevice(T->local[i]->device));
        return sgl_scale_w_enqueue(T->local[i]);
    };
    auto fetch = [&](int i) -> int {
        HIPCHK(hipSetDevice(T->local[i]->device));
        return sgl_scale_w_fetch(T->local[i], &tols[i]);
    };
    if (T->pool) {
        SGLCHK(team_parallel(T, [&](int i) -> int {
            SGLCHK(local_part(i));
            SGLCHK(exchange1(i));
            SGLCHK(gene_block(i));
            SGLCHK(exchange2(i));
            SGLCHK(tail_part(i));
            return fetch(i);
        }));
    } else {
        for (int i = 0; i < nl; ++i) SGLCHK(local_part(i));
        SGLCHK(exchange1(-1));
        for (int i = 0; i < nl; ++i) SGLCHK(gene_block(i));
        SGLCHK(exchange2(-1));
        for (int i = 0; i < nl; ++i) SGLCHK(tail_part(i));
        for (int i = 0; i < nl; ++i) SGLCHK(fetch(i));
    }
    SGLCHK(team_tols_agree(tols));
    if (tol_out) *tol_out = tols[0];   // w is replicated bit for bit: every rank computes the same value
    return SGL_OK;
}

static int team_nmf_run(sgl_team* T, double tol, int32_t maxit, double L1_w, double L1_h, double L2_w, double L2_h,
                        int32_t* n_iter, double* tol_trace, const sgl_callbacks* cb) {
    double tol_ = 1.0;
    int it = 0;
    for (; it < maxit && tol_ > tol; ++it) {   // src/singlet.cpp:647
        SGLCHK(team_iterate(T, L1_w, L1_h, L2_w, L2_h, &tol_));
        if (tol_trace) tol_trace[it] = tol_;
        if (cb && cb->log) cb->log(cb->user, it + 1, tol_, NAN);
        if (cb && cb->poll && cb->poll(cb->user)) { sgl_set_error("interrupted"); return SGL_EINTR; }
    }
    for (auto c : T->local) {
        HIPCHK(hipSetDevice(c->device));
        SGLCHK(sgl_fetch_sweeps(c));
    }
    if (n_iter) *n_iter = it;
    return SGL_OK;
}

// ---------------------------------------------------- the masked (ARD) loop --
// c_ard_nmf_base (src/singlet.cpp:1090-1152) over a team.  The H-update is local (w replicated, hash on
// global cell indices, :485).  For the W-update every gene needs sums over ALL cells: its right-hand side
// b_g and its Gram downdate S_g = sum over the cells masked for g of h_c h_c^T (:458-463 with the
// offset of :590).  Each rank forms the partials of its cells for every gene; ONE grouped collective
// reduce-scatters [b (k x genes) | S (k x k x genes)] by gene blocks and all-reduces the k x k Gram of
// h; each rank then solves its block of genes against a_g = (h h^T + 1e-15 I) - (S_g + 1e-15 I), and
// the w blocks are all-gathered.  k x k x genes doubles is 600 MB at k = 50 -- moved as a
// reduce-scatter it costs each rank (N - 1) / N of that once, not the 2 x of an all-reduce.
// scale(h, d) keeps the reference's order here (row sums all-reduced first): k doubles.
static int team_mask_workspace(sgl_team* T) {
    const int N = T->nranks;
    for (auto c : T->local) {
        HIPCHK(hipSetDevice(c->device));
        SGLCHK(sgl_mask_workspace(c));
        if (N > 1 && !c->Sbuf) {
            const int64_t mb = (T->nrow + N - 1) / N;
            hipError_t e = sgl_pool_malloc((void**)&c->Sbuf, sizeof(double) * (size_t)mb * N * c->k * c->k);
            if (e != hipSuccess) { (void)hipGetLastError(); sgl_set_error("team: out of device memory for the per-gene downdates"); return SGL_ENOMEM; }
            // what the reduce-scatter moves: the lower triangles (SGL_TEAM_FULL_S=1: the full blocks, rounds 2 - 5 -- A/B, tests)
            if (!getenv("SGL_TEAM_FULL_S")) {
                e = sgl_pool_malloc((void**)&c->Stri, sizeof(double) * (size_t)mb * N * c->k * (c->k + 1) / 2);
                if (e != hipSuccess) { (void)hipGetLastError(); sgl_set_error("team: out of device memory for the per-gene downdates"); return SGL_ENOMEM; }
            }
        }
    }
    return SGL_OK;
}

static int team_ard_iterate(sgl_team* T, double L1, double L2, uint64_t seed, uint64_t inv_density, double* tol_out) {
    const int nl = (int)T->local.size();
    const int N = T->nranks;
    const int k = T->local[0]->k;
    const int64_t m = T->nrow;
    const int64_t mb = N > 1 ? (m + N - 1) / N : m;
    const int64_t mpad = mb * N;
    std::vector<void*> dv(nl), red(nl), sb(nl), tail(nl), wbuf(nl);
    std::vector<double> tols(nl, 0.0);
    auto h_update = [&](int i) -> int {   // H-update: local
        sgl_ctx* c = T->local[i];
        HIPCHK(hipSetDevice(c->device));
        SGLCHK(sgl_step_begin(c));
        SGLCHK(sgl_predict_mask_dev(c, c->A, c->col_nnz_A, c->W, c->H, c->B, seed, inv_density, L1, L2, 0, SGL_PH_RHS_H,
                                    SGL_PH_NNLS_H, c->sweep_counters + 0));
        PhaseEvent pe;
        SGLCHK(sgl_phase_begin(c, SGL_PH_SCALE, &pe));
        SGLCHK(k_rowsum(c, c->H, k, c->A.ncol, c->d));
        SGLCHK(sgl_phase_end(c, &pe));
        dv[i] = c->d;
        return SGL_OK;
    };
    auto exchange_d = [&](int who) -> int {
        if (N <= 1) return SGL_OK;
        return team_exchange(T, who, {{0, &dv, (int64_t)k, false}});
    };
    auto w_partials = [&](int i) -> int {   // scale(h, d); partials of the W-update
        sgl_ctx* c = T->local[i];
        HIPCHK(hipSetDevice(c->device));
        PhaseEvent pe;
        SGLCHK(sgl_phase_begin(c, SGL_PH_SCALE, &pe));
        SGLCHK(k_scale_apply(c->stream, c->H, k, c->A.ncol, c->d, 1));
        SGLCHK(sgl_phase_end(c, &pe));
        if (N == 1) {
            const int64_t* gene_nnz = c->col_nnz_At;
            return sgl_predict_mask_dev(c, c->At, gene_nnz, c->H, c->W, c->red, seed, inv_density, L1, L2, 1, SGL_PH_RHS_W,
                                        SGL_PH_NNLS_W, c->sweep_counters + 1);
        }
        double* Bw = c->red;
        double* Gh = c->red + (size_t)k * mpad;
        SGLCHK(sgl_phase_begin(c, SGL_PH_GRAM, &pe));
        SGLCHK(k_gram(c, c->H, k, c->A.ncol, Gh, 0.0));
        SGLCHK(sgl_phase_end(c, &pe));
        SGLCHK(sgl_phase_begin(c, SGL_PH_RHS_W, &pe));
        SGLCHK(sgl_masked_rhs(c, 1, c->H, Bw, seed, inv_density));   // hash: draw(cell = row + cell_offset, gene = column)
        SGLCHK(sgl_phase_end(c, &pe));
        SGLCHK(sgl_phase_begin(c, SGL_PH_MASK, &pe));
        HIPCHK(hipMemsetAsync(c->Sbuf, 0, sizeof(double) * (size_t)mpad * k * k, c->stream));
        if (!c->gene_nnz_global) { sgl_set_error("team: global gene counts missing"); return SGL_ESTATE; }
        const DevMaskList* ML = nullptr;   // this shard's part of every gene's mask as lists (built on the fit's first pass)
        if (k <= 128 && !getenv("SGL_MASK_NO_LIST")) {
            SGLCHK(sgl_mask_list_select(c, 1, m, c->At.nrow, seed, inv_density, 1, 0, c->cell_offset));
            if (c->ML[1].mask_t == 1) ML = &c->ML[1];
        }
        SGLCHK(k_mask_gram_cols(c->stream, 0, m, c->At.nrow, c->col_nnz_At_global, c->H, nullptr, k, seed, inv_density, 1, 0,
                                c->cell_offset, c->Sbuf, ML));
        SGLCHK(sgl_phase_end(c, &pe));
        red[i] = Bw;
        if (c->Stri) {   // S_g is symmetric: its lower triangle travels, k (k + 1) / 2 doubles per gene instead of k^2
            SGLCHK(sgl_phase_begin(c, SGL_PH_MASK, &pe));
            SGLCHK(k_tri_pack(c->stream, c->Sbuf, k, mpad, c->Stri));
            SGLCHK(sgl_phase_end(c, &pe));
        }
        sb[i] = c->Stri ? c->Stri : c->Sbuf;
        tail[i] = Gh;
        wbuf[i] = c->W;
        return SGL_OK;
    };
    const bool tri = T->local[0]->Stri != nullptr;
    const int64_t s_unit = tri ? (int64_t)k * (k + 1) / 2 : (int64_t)k * k;   // doubles per gene in the downdate exchange
    auto exchange_w = [&](int who) -> int {
        if (N <= 1) return SGL_OK;
        return team_exchange(T, who, {{1, &red, (int64_t)k * mb, false}, {1, &sb, s_unit * mb, false}, {0, &tail, (int64_t)k * k, false}});
    };
    auto gene_block = [&](int i) -> int {   // every rank: its block of genes
        if (N <= 1) return SGL_OK;
        sgl_ctx* c = T->local[i];
        HIPCHK(hipSetDevice(c->device));
        const int r = T->rank[i];
        const int64_t g0 = (int64_t)r * mb;
        const int64_t ng = std::max<int64_t>(0, std::min<int64_t>(mb, m - g0));
        double* Gh = c->red + (size_t)k * mpad;
        PhaseEvent pe;
        SGLCHK(sgl_phase_begin(c, SGL_PH_GRAM, &pe));
        HIPCHK(hipMemcpyAsync(c->G, Gh, sizeof(double) * k * k, hipMemcpyDeviceToDevice, c->stream));
        SGLCHK(k_gram_add_diag(c->stream, c->G, k, 1e-15));
        SGLCHK(sgl_phase_end(c, &pe));
        for (int64_t q0 = 0; q0 < ng; q0 += c->gcols_chunk) {
            const int64_t nq = std::min<int64_t>(c->gcols_chunk, ng - q0);
            SGLCHK(sgl_phase_begin(c, SGL_PH_MASK, &pe));
            if (c->Stri) SGLCHK(k_mask_gram_finalize_tri(c->stream, c->G, c->Stri + (size_t)(g0 + q0) * s_unit, k, nq, c->Gcols));
            else SGLCHK(k_mask_gram_finalize(c->stream, c->G, c->Sbuf + (size_t)(g0 + q0) * k * k, k, nq, c->Gcols));
            SGLCHK(sgl_phase_end(c, &pe));
            SGLCHK(sgl_phase_begin(c, SGL_PH_NNLS_W, &pe));
            SGLCHK(k_nnls_percol(c->stream, c->Gcols, (int64_t)k * k, c->red + (size_t)(g0 + q0) * k, c->W + (size_t)(g0 + q0) * k,
                               c->col_nnz_At_global + g0 + q0, k, nq, L1, L2, c->sweep_counters + 1));
            SGLCHK(sgl_phase_end(c, &pe));
        }
        return SGL_OK;
    };
    auto exchange_gather = [&](int who) -> int {
        if (N <= 1) return SGL_OK;
        return team_exchange(T, who, {{2, &wbuf, (int64_t)k * mb, false}});
    };
    auto tail_part = [&](int i) -> int {
        HIPCHK(hipSetDevice(T->local[i]->device));
        return sgl_scale_w_enqueue(T->local[i]);
    };
    auto fetch = [&](int i) -> int {
        HIPCHK(hipSetDevice(T->local[i]->device));
        return sgl_scale_w_fetch(T->local[i], &tols[i]);
    };
    if (T->pool) {
        SGLCHK(team_parallel(T, [&](int i) -> int {
            SGLCHK(h_update(i));
            SGLCHK(exchange_d(i));
            SGLCHK(w_partials(i));
            SGLCHK(exchange_w(i));
            SGLCHK(gene_block(i));
            SGLCHK(exchange_gather(i));
            SGLCHK(tail_part(i));
            return fetch(i);
        }));
    } else {
        for (int i = 0; i < nl; ++i) SGLCHK(h_update(i));
        SGLCHK(exchange_d(-1));
        for (int i = 0; i < nl; ++i) SGLCHK(w_partials(i));
        SGLCHK(exchange_w(-1));
        for (int i = 0; i < nl; ++i) SGLCHK(gene_block(i));
        SGLCHK(exchange_gather(-1));
        for (int i = 0; i < nl; ++i) SGLCHK(tail_part(i));
        for (int i = 0; i < nl; ++i) SGLCHK(fetch(i));
    }
    SGLCHK(team_tols_agree(tols));
    if (tol_out) *tol_out = tols[0];
    return SGL_OK;
}

// mse_test (src/singlet.cpp:536-568) over the team: local sums of the per-cell losses, one double all-reduced
static int team_mse_test(sgl_team* T, uint64_t seed, uint64_t inv_density, double* out) {
    const int nl = (int)T->local.size();
    std::vector<void*> sc(nl);
    auto local_loss = [&](int i) -> int {
        sgl_ctx* c = T->local[i];
        HIPCHK(hipSetDevice(c->device));
        SGLCHK(sgl_mse_test_enqueue(c, seed, inv_density));
        sc[i] = c->scalars + 1;
        return SGL_OK;
    };
    auto exchange = [&](int who) -> int {
        if (T->nranks <= 1) return SGL_OK;
        return team_exchange(T, who, {{0, &sc, (int64_t)1, false}});
    };
    if (T->pool) {
        SGLCHK(team_parallel(T, [&](int i) -> int {
            SGLCHK(local_loss(i));
            return exchange(i);
        }));
    } else {
        for (int i = 0; i < nl; ++i) SGLCHK(local_loss(i));
        SGLCHK(exchange(-1));
    }
    sgl_ctx* c0 = T->local[0];
    HIPCHK(hipSetDevice(c0->device));
    HIPCHK(hipMemcpyAsync(c0->pinned + 1, c0->scalars + 1, sizeof(double), hipMemcpyDeviceToHost, c0->stream));
    HIPCHK(hipStreamSynchronize(c0->stream));
    *out = c0->pinned[1] / (double)c0->ncells_total;   // losses.sum() / h.cols(), l.567
    return SGL_OK;
}

static int team_ard_run(sgl_team* T, double tol, int32_t maxit, double L1, double L2, uint64_t seed, uint64_t inv_density,
                        double overfit_threshold, int32_t trace_test_mse, double* test_mse, int32_t* iter, double* tol_out,
                        double* score_overfit, int32_t* n_trace, int32_t* n_iter, const sgl_callbacks* cb) {
    if (trace_test_mse <= 0 || inv_density == 0 || !test_mse || !iter || !tol_out || !score_overfit || !n_trace) {
        sgl_set_error("ard_run: bad arguments"); return SGL_EINVAL;
    }
    for (auto c : T->local) {
        if (c->k == 0) { sgl_set_error("team: no fit initialised"); return SGL_ESTATE; }
        if (c->k > SGL_MASK_MAX_K) { sgl_set_error("c_ard_nmf: rank %d above the masked path's limit of %d", c->k, SGL_MASK_MAX_K); return SGL_EINVAL; }
    }
    if (T->nranks > 1) {
        bool have = true;
        for (auto c : T->local) have = have && c->gene_nnz_global;
        if (!have) SGLCHK(team_gene_counts(T));
    }
    SGLCHK(team_mask_workspace(T));
    double tol_ = 1.0;
    int nt = 0, it = 0;
    auto push_trace = [&](int iter_now) -> int {   // l.1112-1121 / 1130-1141
        double err = 0.0;
        SGLCHK(team_mse_test(T, seed, inv_density, &err));
        test_mse[nt] = err;
        iter[nt] = iter_now;
        tol_out[nt] = tol_;
        double min_err = test_mse[0];
        for (int t = 1; t <= nt; ++t) min_err = std::min(min_err, test_mse[t]);
        score_overfit[nt] = (err - min_err) / (err + min_err);
        ++nt;
        return SGL_OK;
    };
    for (; it < maxit && tol_ > tol; ++it) {
        SGLCHK(team_ard_iterate(T, L1, L2, seed, inv_density, &tol_));
        if (it % trace_test_mse == 0) {
            SGLCHK(push_trace(it));
            if (cb && cb->log) cb->log(cb->user, it + 1, tol_, score_overfit[nt - 1]);
            if (score_overfit[nt - 1] > overfit_threshold) break;
        } else if (cb && cb->log) {
            cb->log(cb->user, it + 1, tol_, NAN);
        }
        if (cb && cb->poll && cb->poll(cb->user)) { sgl_set_error("interrupted"); return SGL_EINTR; }
    }
    if (it % trace_test_mse != 0) SGLCHK(push_trace(it));
    for (auto c : T->local) {
        HIPCHK(hipSetDevice(c->device));
        SGLCHK(sgl_fetch_sweeps(c));
    }
    *n_trace = nt;
    if (n_iter) *n_iter = it;
    return SGL_OK;
}

int sgl_ard_run_team(sgl_ctx* c, double tol, int32_t maxit, double L1, double L2, uint64_t seed, uint64_t inv_density,
                     double overfit_threshold, int32_t trace_test_mse, double* test_mse, int32_t* iter, double* tol_out,
                     double* score_overfit, int32_t* n_trace, int32_t* n_iter, const sgl_callbacks* cb) {
    sgl_team* T = c->team;
    if (!T || T->local.size() != 1) { sgl_set_error("sgl_ard_run: this context is driven by its sgl_multi; call sgl_multi_ard_run"); return SGL_ESTATE; }
    T->nrow = c->A.nrow;
    return team_ard_run(T, tol, maxit, L1, L2, seed, inv_density, overfit_threshold, trace_test_mse, test_mse, iter, tol_out,
                        score_overfit, n_trace, n_iter, cb);
}

// ------------------------------------------------- one process per GPU (ABI) --
extern "C" int sgl_comm_unique_id(void* id128) {
    if (!id128) { sgl_set_error("sgl_comm_unique_id: NULL buffer"); return SGL_EINVAL; }
    RcclApi* R = rccl_api();
    if (!R) return SGL_ECOMM;
    ncclUniqueId id;
    NCCLCHK(R->GetUniqueId(&id));
    static_assert(sizeof(id) == SGL_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return SGL_OK;
}

extern "C" int sgl_comm_init_rank(sgl_ctx* c, int nranks, int rank, const void* id128) {
    if (!c) { sgl_set_error("null context"); return SGL_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (nranks < 1 || nranks > SGL_TEAM_MAX || rank < 0 || rank >= nranks || !id128) { sgl_set_error("sgl_comm_init_rank: bad arguments"); return SGL_EINVAL; }
    if (c->team || c->allreduce) { sgl_set_error("sgl_comm_init_rank: the context already has a team or an all-reduce hook"); return SGL_ESTATE; }
    if (c->k != 0) { sgl_set_error("sgl_comm_init_rank: call before sgl_fit_init (buffer sizes depend on the team)"); return SGL_ESTATE; }
    RcclApi* R = rccl_api();
    if (!R) return SGL_ECOMM;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    NCCLCHK(R->CommInitRank(&comm, nranks, id, rank));
    sgl_team* T = new (std::nothrow) sgl_team();
    if (!T) { (void)R->CommDestroy(comm); sgl_set_error("out of host memory"); return SGL_ENOMEM; }
    T->nranks = nranks;
    T->local.push_back(c);
    T->rank.push_back(rank);
    T->comm.push_back(comm);
    c->team = T;
    c->team_rank = rank;
    c->gene_nnz_global = false;
    return SGL_OK;
}

// RCCL can be bound in this process (no collective call: ranks agree on this BEFORE sgl_comm_init_rank, which blocks
// until every rank has joined); path_out (may be NULL) receives the library name that was opened
extern "C" int sgl_comm_available(char* path_out, int path_len) {
    RcclApi* R = rccl_api();
    if (path_out && path_len > 0) snprintf(path_out, (size_t)path_len, "%s", R ? R->path : "");
    return R ? SGL_OK : SGL_ECOMM;
}

// what the library's own communicator says: ranks it spans (ncclCommCount; 1 without a team, the team size in
// loopback mode), whether it is an RCCL communicator, and the RCCL library bound
extern "C" int sgl_comm_info(sgl_ctx* c, int32_t* nranks, int32_t* is_rccl, char* path_out, int path_len) {
    if (!c) { sgl_set_error("null context"); return SGL_EINVAL; }
    if (nranks) *nranks = 1;
    if (is_rccl) *is_rccl = 0;
    if (path_out && path_len > 0) path_out[0] = 0;
    sgl_team* T = c->team;
    if (!T) return SGL_OK;
    if (nranks) *nranks = T->nranks;
    if (T->loopback || T->comm.empty()) return SGL_OK;
    RcclApi* R = rccl_api();
    if (!R) return SGL_ECOMM;
    for (size_t i = 0; i < T->local.size(); ++i)
        if (T->local[i] == c && i < T->comm.size() && T->comm[i]) {
            int cnt = 0;
            NCCLCHK(R->CommCount(T->comm[i], &cnt));
            if (nranks) *nranks = cnt;
            if (is_rccl) *is_rccl = 1;
            if (path_out && path_len > 0) snprintf(path_out, (size_t)path_len, "%s", R->path);
        }
    return SGL_OK;
}

// One ALS iteration on a context, whatever its exchange: none (one shard), the all-reduce hook (two
// all-reduces, the reference's operation order) or a native team (above).
extern "C" int sgl_nmf_iterate(sgl_ctx* c, double L1_w, double L1_h, double L2_w, double L2_h, double* tol) {
    if (!c) { sgl_set_error("null context"); return SGL_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (c->team) {
        sgl_team* T = c->team;
        if (T->local.size() != 1) { sgl_set_error("sgl_nmf_iterate: this context is driven by its sgl_multi; call sgl_multi_iterate"); return SGL_ESTATE; }
        T->nrow = c->A.nrow;
        if (T->nranks > 1 && !c->gene_nnz_global) SGLCHK(team_gene_counts(T));
        return team_iterate(T, L1_w, L1_h, L2_w, L2_h, tol);
    }
    SGLCHK(sgl_step_begin(c));
    SGLCHK(sgl_step_h(c, L1_h, L2_h));
    SGLCHK(sgl_step_scale_h(c));
    SGLCHK(sgl_step_w(c, L1_w, L2_w));
    return sgl_step_scale_w(c, tol);
}

// ---------------------------------------------- one process, all devices (ABI) --
extern "C" int sgl_multi_create(int ndev, const int* devices, sgl_multi** out) {
    if (!out) { sgl_set_error("sgl_multi_create: out is NULL"); return SGL_EINVAL; }
    *out = nullptr;
    if (ndev < 1 || ndev > SGL_TEAM_MAX) { sgl_set_error("sgl_multi_create: ndev=%d out of range (1..%d)", ndev, SGL_TEAM_MAX); return SGL_EINVAL; }
    std::vector<int> dev(ndev);
    for (int i = 0; i < ndev; ++i) dev[i] = devices ? devices[i] : i;
    bool all_same = true, distinct = true;
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j) {
            if (dev[i] == dev[j]) distinct = false;
            else all_same = false;
        }
    if (ndev > 1 && !distinct && !all_same) { sgl_set_error("sgl_multi_create: devices must be all distinct (RCCL) or all the same (loopback)"); return SGL_EINVAL; }
    sgl_multi* M = new (std::nothrow) sgl_multi();
    if (!M) { sgl_set_error("out of host memory"); return SGL_ENOMEM; }
    M->nranks = ndev;
    M->owns_ctx = true;
    M->loopback = ndev > 1 && all_same;
    int rc = SGL_OK;
    for (int i = 0; i < ndev && rc == SGL_OK; ++i) {
        sgl_ctx* c = nullptr;
        rc = sgl_create(dev[i], &c);
        if (rc == SGL_OK) {
            c->team = M;
            c->team_rank = i;
            M->local.push_back(c);
            M->rank.push_back(i);
        }
    }
    if (rc == SGL_OK && !M->loopback) {
        RcclApi* R = rccl_api();
        if (!R) rc = SGL_ECOMM;
        else {
            M->comm.assign(ndev, nullptr);
            for (int i = 0; i < ndev; ++i) M->comm_mu.emplace_back(new std::timed_mutex());
            ncclResult_t r = R->CommInitAll(M->comm.data(), ndev, dev.data());
            if (r != ncclSuccess) { sgl_set_error("ncclCommInitAll failed: %s", R->GetErrorString(r)); M->comm.clear(); rc = SGL_ECOMM; }
        }
    }
    if (rc == SGL_OK) rc = team_events(M);
    if (rc == SGL_OK) rc = pool_start(M);
    if (rc != SGL_OK) { sgl_multi_destroy(M); return rc; }
    *out = M;
    return SGL_OK;
}

extern "C" int sgl_multi_destroy(sgl_multi* M) {
    if (!M) return SGL_OK;
    pool_stop(M);
    std::vector<sgl_ctx*> ctxs = M->local;
    for (auto c : ctxs) sgl_destroy(c);   // detaches (and destroys the rank's communicator)
    for (auto e : M->ev) (void)hipEventDestroy(e);
    if (M->done) (void)hipEventDestroy(M->done);
    delete M;
    return SGL_OK;
}

extern "C" int sgl_multi_size(const sgl_multi* M) { return M ? M->nranks : 0; }

extern "C" int sgl_multi_ctx(sgl_multi* M, int rank, sgl_ctx** out) {
    TEAM_GUARD(M);
    if (!out || rank < 0 || rank >= (int)M->local.size()) { sgl_set_error("sgl_multi_ctx: bad rank"); return SGL_EINVAL; }
    *out = M->local[rank];
    return SGL_OK;
}

// Contiguous cell blocks with (nearly) equal non-zero counts, every block at least one cell: lo[0] = 0,
// lo[n] = ncol stay fixed and only the interior boundaries are clamped into [lo[r-1] + 1, ncol - (n - r)]
// (a heavy last cell used to push lo[n] past ncol: round-2 advice).  Host only; exported for the tests.
extern "C" int sgl_split_cells_by_nnz(const int32_t* p, int32_t ncol, int n, int64_t* lo) {
    if (!p || !lo || n < 1 || ncol < n) { sgl_set_error("sgl_split_cells_by_nnz: fewer cells (%d) than ranks (%d)", ncol, n); return SGL_EINVAL; }
    const int64_t total = p[ncol];
    lo[0] = 0;
    for (int r = 1; r < n; ++r) {
        const int64_t target = total * r / n;
        int64_t c = std::lower_bound(p, p + ncol + 1, (int32_t)std::min<int64_t>(target, INT32_MAX)) - p;
        c = std::max<int64_t>(c, lo[r - 1] + 1);
        c = std::min<int64_t>(c, (int64_t)ncol - (n - r));
        lo[r] = c;
    }
    lo[n] = ncol;
    return SGL_OK;
}

extern "C" int sgl_multi_upload_csc(sgl_multi* M, const double* Ax, const int32_t* Ai, const int32_t* Ap, int32_t nrow, int32_t ncol) {
    TEAM_GUARD(M);
    if (!Ax || !Ai || !Ap || nrow <= 0 || ncol <= 0) { sgl_set_error("sgl_multi_upload_csc: missing slot or empty matrix"); return SGL_EINVAL; }
    const int n = M->nranks;
    if (ncol < n) { sgl_set_error("sgl_multi_upload_csc: fewer cells (%d) than devices (%d)", ncol, n); return SGL_EINVAL; }
    M->cell_lo.assign(n + 1, 0);
    SGLCHK(sgl_split_cells_by_nnz(Ap, ncol, n, M->cell_lo.data()));
    M->nrow = nrow;
    M->ncells_total = ncol;
    std::vector<int32_t> p;
    for (int r = 0; r < n; ++r) {
        const int64_t lo = M->cell_lo[r], hi = M->cell_lo[r + 1];
        p.resize((size_t)(hi - lo) + 1);
        for (int64_t q = lo; q <= hi; ++q) p[(size_t)(q - lo)] = Ap[q] - Ap[lo];
        // transposes are built on the device (the host's t(A) describes the whole matrix, not a block)
        SGLCHK(sgl_upload_csc(M->local[r], Ax + Ap[lo], Ai + Ap[lo], p.data(), nullptr, nullptr, nullptr, nrow, (int32_t)(hi - lo), lo, ncol));
    }
    return SGL_OK;
}

extern "C" int sgl_multi_synth_csc(sgl_multi* M, uint64_t S, uint64_t inv_density, const double* levels16, int32_t ngenes,
                                   int64_t ncells_total) {
    TEAM_GUARD(M);
    const int n = M->nranks;
    if (ncells_total < n) { sgl_set_error("sgl_multi_synth_csc: fewer cells than devices"); return SGL_EINVAL; }
    M->cell_lo.assign(n + 1, 0);
    const int64_t base = ncells_total / n, rem = ncells_total % n;
    for (int r = 0; r < n; ++r) M->cell_lo[r + 1] = M->cell_lo[r] + base + (r < rem ? 1 : 0);
    M->nrow = ngenes;
    M->ncells_total = ncells_total;
    return team_parallel(M, [&](int r) -> int {
        return sgl_synth_csc(M->local[r], S, inv_density, levels16, ngenes, M->cell_lo[r], (int32_t)(M->cell_lo[r + 1] - M->cell_lo[r]), ncells_total);
    });
}

extern "C" int sgl_multi_fit_init(sgl_multi* M, int32_t k, const double* w_init, uint64_t synth_seed) {
    TEAM_GUARD(M);
    if (M->cell_lo.empty()) { sgl_set_error("sgl_multi_fit_init: no matrix resident"); return SGL_ESTATE; }
    SGLCHK(team_parallel(M, [&](int r) -> int { return sgl_fit_init(M->local[r], k, w_init, synth_seed); }));
    for (auto c : M->local) c->gene_nnz_global = false;
    return team_gene_counts(M);
}

// link matrices of c_linked_nmf for the whole matrix (column-major rows x cols as R holds them): link_h's columns are
// dealt out to the ranks with their cells, link_w goes to every rank; a link whose column count does not match its
// side is ignored, as in the reference (src/singlet.cpp:1059-1065)
extern "C" int sgl_multi_set_links(sgl_multi* M, const double* link_h, int32_t link_h_rows, int32_t link_h_cols, const double* link_w,
                                   int32_t link_w_rows, int32_t link_w_cols) {
    TEAM_GUARD(M);
    if (M->cell_lo.empty()) { sgl_set_error("sgl_multi_set_links: no matrix resident"); return SGL_ESTATE; }
    const bool use_h = link_h && link_h_rows > 0 && (int64_t)link_h_cols == M->ncells_total;
    for (int r = 0; r < M->nranks; ++r) {
        const int64_t lo = M->cell_lo[r], nloc = M->cell_lo[r + 1] - lo;
        SGLCHK(sgl_set_links(M->local[r], use_h ? link_h + (size_t)lo * link_h_rows : nullptr, link_h_rows, use_h ? (int32_t)nloc : 0, link_w,
                             link_w_rows, link_w_cols));
    }
    return SGL_OK;
}

extern "C" int sgl_multi_iterate(sgl_multi* M, double L1_w, double L1_h, double L2_w, double L2_h, double* tol) {
    TEAM_GUARD(M);
    return team_iterate(M, L1_w, L1_h, L2_w, L2_h, tol);
}

extern "C" int sgl_multi_nmf_run(sgl_multi* M, double tol, int32_t maxit, double L1_w, double L1_h, double L2_w, double L2_h,
                                 int32_t* n_iter, double* tol_trace, const sgl_callbacks* cb) {
    TEAM_GUARD(M);
    return team_nmf_run(M, tol, maxit, L1_w, L1_h, L2_w, L2_h, n_iter, tol_trace, cb);
}

extern "C" int sgl_multi_ard_run(sgl_multi* M, double tol, int32_t maxit, double L1, double L2, uint64_t seed, uint64_t inv_density,
                                 double overfit_threshold, int32_t trace_test_mse, double* test_mse, int32_t* iter,
                                 double* tol_out, double* score_overfit, int32_t* n_trace, int32_t* n_iter,
                                 const sgl_callbacks* cb) {
    TEAM_GUARD(M);
    return team_ard_run(M, tol, maxit, L1, L2, seed, inv_density, overfit_threshold, trace_test_mse, test_mse, iter, tol_out,
                        score_overfit, n_trace, n_iter, cb);
}

extern "C" int sgl_multi_get_factors(sgl_multi* M, double* w, double* d, double* h) {
    TEAM_GUARD(M);
    if (M->cell_lo.empty()) { sgl_set_error("sgl_multi_get_factors: no matrix resident"); return SGL_ESTATE; }
    const int k = M->local[0]->k;
    SGLCHK(sgl_get_factors(M->local[0], w, d, nullptr));
    if (h)
        for (int r = 0; r < M->nranks; ++r) SGLCHK(sgl_get_factors(M->local[r], nullptr, nullptr, h + (size_t)M->cell_lo[r] * k));
    return SGL_OK;
}

// c_nmf on all devices of this process: what sgl_c_nmf runs when SINGLET_NGPU asks for more than one
int sgl_c_nmf_multi(int ndev, const double* Ax, const int32_t* Ai, const int32_t* Ap, int32_t nrow, int32_t ncol, double tol,
                    uint16_t maxit, double L1_w, double L1_h, double L2_w, double L2_h, const double* w_init, int32_t k,
                    double* w_out, double* d_out, double* h_out, int32_t* n_iter, double* tol_trace, const sgl_callbacks* cb) {
    sgl_multi* M = nullptr;
    SGLCHK(sgl_multi_create(ndev, nullptr, &M));
    int rc = sgl_multi_upload_csc(M, Ax, Ai, Ap, nrow, ncol);
    if (rc == SGL_OK) rc = sgl_multi_fit_init(M, k, w_init, 0);
    if (rc == SGL_OK) rc = sgl_multi_nmf_run(M, tol, maxit, L1_w, L1_h, L2_w, L2_h, n_iter, tol_trace, cb);
    if (rc == SGL_OK) rc = sgl_multi_get_factors(M, w_out, d_out, h_out);
    sgl_multi_destroy(M);
    return rc;
}

int sgl_c_ard_nmf_multi(int ndev, const double* Ax, const int32_t* Ai, const int32_t* Ap, int32_t nrow, int32_t ncol, double tol,
                        uint16_t maxit, double L1, double L2, const double* w_init, int32_t k, uint64_t seed, uint64_t inv_density,
                        double overfit_threshold, uint16_t trace_test_mse, double* w_out, double* d_out, double* h_out,
                        double* test_mse, int32_t* iter, double* tol_out, double* score_overfit, int32_t* n_trace,
                        const sgl_callbacks* cb) {
    sgl_multi* M = nullptr;
    SGLCHK(sgl_multi_create(ndev, nullptr, &M));
    int32_t nit = 0;
    int rc = sgl_multi_upload_csc(M, Ax, Ai, Ap, nrow, ncol);
    if (rc == SGL_OK) rc = sgl_multi_fit_init(M, k, w_init, 0);
    if (rc == SGL_OK) rc = sgl_multi_ard_run(M, tol, maxit, L1, L2, seed, inv_density, overfit_threshold, trace_test_mse, test_mse,
                                             iter, tol_out, score_overfit, n_trace, &nit, cb);
    if (rc == SGL_OK) rc = sgl_multi_get_factors(M, w_out, d_out, h_out);
    sgl_multi_destroy(M);
    return rc;
}
