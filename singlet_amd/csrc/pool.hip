// Device-memory pool of the library (round 6).
//
// Why.  A one-shot call (sgl_c_nmf behind R's run_nmf) creates a context, allocates ~85 GB at config 3 -- the matrix slots, the
// transpose, two entry streams of 22 GB, the factors -- and frees all of it when it returns.  The NEXT call's hipMalloc of the
// 22 GB streams then takes ~3 s (measured with SGL_TRACE_SETUP=1: "roff / x buffers reserved 2948 / 3258 / 3049 / 3262 ms" in
// four of five calls at config 3, against 0.5 ms in a process that has freed nothing; profiles/r6_one_shot_trace.txt) -- the
// driver hands back memory it is still scrubbing -- i.e. as long as the 99 iterations of the fit itself.  R's ard_nmf /
// cross_validate_nmf make tens of such calls (R/ard_nmf.R:95-160).  The same allocator stalls (0.1 - 4 s per block,
// erratically) were met inside the config-5 grid in round 2, where the streams were made to outlive the fit; this is the general fix.
//
// What.  Blocks of at least 64 MB are not given back to the driver when the library frees them: they wait in a per-device list
// and serve the next request they fit (size >= the request, at most 1.25 x + 64 MB of it).  Everything else passes through.  A
// failed hipMalloc -- of a pooled or a small block -- releases every cached block and tries once more, and sgl_pool_mem_info
// counts the cached bytes as free, so the budgets derived from the free memory (mask lists, the chunk of per-column Grams) do
// not shrink because of the pool.  A block is cached only after the device has gone idle (hipFree synchronises too): the next
// owner may live on another stream.  SGL_POOL=0 switches the pool off (every free goes to the driver); sgl_cache_release()
// empties it.  Memory cached here stays reserved by the process -- other processes see it as used.
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct Blk {
    void* p;
    size_t bytes;
    int dev;
};
std::mutex g_mu;
std::unordered_map<void*, Blk> g_live;   // pooled blocks in use
std::vector<Blk> g_cache;                // pooled blocks waiting for their next owner (oldest first)
constexpr size_t POOL_MIN = (size_t)64 << 20;

bool pool_on() {
    static const bool on = [] { const char* e = getenv("SGL_POOL"); return !(e && atoi(e) == 0); }();
    return on;
}
size_t pool_cap_bytes(size_t total) {   // cached bytes per device above which the oldest blocks go back to the driver
    static const double gb = [] { const char* e = getenv("SGL_POOL_MAX_GB"); return e ? atof(e) : -1.0; }();
    return gb >= 0.0 ? (size_t)(gb * 1073741824.0) : total / 10 * 7;
}
size_t cached_on(int dev) {
    size_t s = 0;
    for (const Blk& b : g_cache)
        if (b.dev == dev) s += b.bytes;
    return s;
}
// every cached block (dev < 0) or those of one device: back to the driver; returns true if anything was freed
bool drop_cached(int dev) {
    std::vector<Blk> out;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t q = 0; q < g_cache.size();)
            if (dev < 0 || g_cache[q].dev == dev) { out.push_back(g_cache[q]); g_cache.erase(g_cache.begin() + (long)q); }
            else ++q;
    }
    for (const Blk& b : out) (void)hipFree(b.p);
    return !out.empty();
}
}  // namespace

hipError_t sgl_pool_malloc_raw(void** p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (!pool_on() || bytes < POOL_MIN) {
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess && drop_cached(dev)) { (void)hipGetLastError(); e = hipMalloc(p, bytes); }
        return e;
    }
    const size_t need = (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        long best = -1;
        for (size_t q = 0; q < g_cache.size(); ++q) {
            const Blk& b = g_cache[q];
            if (b.dev != dev || b.bytes < need || b.bytes > need + need / 4 + POOL_MIN) continue;
            if (best < 0 || b.bytes < g_cache[(size_t)best].bytes) best = (long)q;
        }
        if (best >= 0) {
            const Blk b = g_cache[(size_t)best];
            g_cache.erase(g_cache.begin() + best);
            g_live[b.p] = b;
            *p = b.p;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess && drop_cached(dev)) { (void)hipGetLastError(); e = hipMalloc(p, need); }
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[*p] = Blk{*p, need, dev};
    return hipSuccess;
}

hipError_t sgl_pool_free(void* p) {
    if (!p) return hipSuccess;
    Blk b{};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) b.p = nullptr;
        else { b = it->second; g_live.erase(it); }
    }
    if (!b.p || !pool_on()) return hipFree(p);
    // hipFree waits for the device before it releases memory; the block's next owner may run on another stream: same rule
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (have_cur && cur != b.dev) (void)hipSetDevice(b.dev);
    const hipError_t es = hipDeviceSynchronize();
    if (have_cur && cur != b.dev) (void)hipSetDevice(cur);
    if (es != hipSuccess) { (void)hipGetLastError(); return hipFree(p); }
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = 0; }
    std::vector<Blk> evict;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_cache.push_back(b);
        const size_t cap = pool_cap_bytes(total_b);
        while (total_b > 0 && cached_on(b.dev) > cap) {
            size_t q = 0;
            while (q < g_cache.size() && g_cache[q].dev != b.dev) ++q;
            if (q >= g_cache.size()) break;
            evict.push_back(g_cache[q]);
            g_cache.erase(g_cache.begin() + (long)q);
        }
    }
    for (const Blk& v : evict) (void)hipFree(v.p);
    return hipSuccess;
}

// hipMemGetInfo with the current device's cached blocks counted as free (they are released on demand)
hipError_t sgl_pool_mem_info(size_t* free_b, size_t* total_b) {
    const hipError_t e = hipMemGetInfo(free_b, total_b);
    if (e != hipSuccess) return e;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }
    std::lock_guard<std::mutex> lk(g_mu);
    *free_b += cached_on(dev);
    return hipSuccess;
}

void sgl_pool_release(void) { (void)drop_cached(-1); }

// bytes cached on the current device (tests)
size_t sgl_pool_cached_bytes(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lk(g_mu);
    return cached_on(dev);
}
