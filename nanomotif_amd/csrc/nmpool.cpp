// nmpool — a block cache behind nm_set_device_allocator for processes that bring no pool of their own (the command line), and the
// file parsers' pinned host buffers (nmres.h, at the end).
//
// Why: memory another process used before is SCRUBBED by the driver when it is handed out again, at 7-30 GB/s — the first part of
// the pre-filters waited 190 ms for its 1.5 GB of state planes (NM_INGEST_TIMING at 1 Gbp, profiles/r5/cli_1gbp.json) right after the
// parser had given back 8 GB it had already paid that price for.  hipFree also synchronises the whole device.  With the cache a
// large block that is freed stays with the process and serves the next request it fits; small blocks go to hipMalloc / hipFree as
// before.  The library waits for its own streams before it frees, so a cached block has no work in flight.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstdint>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/nmscan.h"
#include "nmres.h"

int nm_set_error(int code, const char *fmt, ...);

namespace {

struct BlockCache {
    struct Block { size_t size; void *ptr; int device; bool settled; };   // settled: the device has been synchronised since the block came back
    std::mutex mu;
    struct Live { size_t size; int device; };
    std::unordered_map<void *, Live> live;               // blocks handed out by this cache (large ones only): their true size and device
    std::vector<Block> idle;                             // cached blocks, unordered (there are a handful)
    size_t idle_bytes = 0, max_idle = 0;
    uint64_t hits = 0, misses = 0, released = 0;
    static constexpr size_t MIN_BLOCK = 32u << 20;       // smaller blocks are not worth keeping
    static constexpr size_t ROUND = 2u << 20;

    // hipMalloc; when it fails while blocks lie idle, they go back to the driver first (small and large requests alike: the idle
    // blocks are invisible to every other user of hipMalloc in the process)
    hipError_t malloc_retry(void **p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            bool any;
            {
                std::lock_guard<std::mutex> lk(mu);
                any = !idle.empty();
            }
            if (any) {
                trim(0);
                e = hipMalloc(p, bytes);
                if (e != hipSuccess) (void)hipGetLastError();
            }
        }
        return e;
    }

    int alloc(void **out, size_t bytes) {
        *out = nullptr;
        if (bytes < MIN_BLOCK) return malloc_retry(out, bytes) == hipSuccess ? 0 : -4;
        int device = 0;
        if (hipGetDevice(&device) != hipSuccess) return -4;
        const size_t want = (bytes + ROUND - 1) / ROUND * ROUND;
        {
            std::unique_lock<std::mutex> lk(mu);
            // best fit ON THIS DEVICE: the smallest idle block that holds the request and is not more than eight times its size
            int best = -1;
            for (int i = 0; i < (int)idle.size(); ++i)
                if (idle[i].device == device && idle[i].size >= want && idle[i].size <= 8 * want && (best < 0 || idle[i].size < idle[best].size)) best = i;
            if (best >= 0) {
                const Block blk = idle[best];
                idle.erase(idle.begin() + best);
                idle_bytes -= blk.size;
                live[blk.ptr] = Live{blk.size, blk.device};
                hits += 1;
                if (!blk.settled) {
                    // hipFree would have synchronised the device before the memory could be handed out again; a cached block is reused at
                    // once, so the cache does it here — once for everything that came back since the last time (the library waits for its
                    // own streams before it frees: this costs a few microseconds and guards against a caller that does not)
                    for (Block &o : idle)
                        if (o.device == device) o.settled = true;
                    lk.unlock();
                    if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
                }
                *out = blk.ptr;
                return 0;
            }
        }
        void *p = nullptr;
        if (malloc_retry(&p, want) != hipSuccess) return -4;
        std::lock_guard<std::mutex> lk(mu);
        live[p] = Live{want, device};
        misses += 1;
        *out = p;
        return 0;
    }

    int release(void *p) {
        if (!p) return 0;
        size_t size = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = live.find(p);
            if (it != live.end()) {
                size = it->second.size;
                const int device = it->second.device;
                live.erase(it);
                if (idle_bytes + size <= max_idle) {
                    idle.push_back(Block{size, p, device, false});
                    idle_bytes += size;
                    return 0;
                }
            }
        }
        if (size) {
            std::lock_guard<std::mutex> lk(mu);
            released += 1;
        }
        return hipFree(p) == hipSuccess ? 0 : -1;        // a small block, or the cache is full
    }

    void trim(size_t keep_bytes) {                        // frees idle blocks, largest first, until at most keep_bytes stay
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            std::sort(idle.begin(), idle.end(), [](const Block &x, const Block &y) { return x.size < y.size; });
            while (!idle.empty() && idle_bytes > keep_bytes) {
                drop.push_back(idle.back().ptr);
                idle_bytes -= idle.back().size;
                idle.pop_back();
            }
        }
        for (void *p : drop) (void)hipFree(p);
    }
};

BlockCache g_cache;
bool g_installed = false;

int cache_alloc(void *, void **ptr, size_t bytes) { return g_cache.alloc(ptr, bytes); }
int cache_free(void *, void *ptr) { return g_cache.release(ptr); }

// ---- pinned host buffers of the file parsers (nmres.h)
struct PinnedCache {
    struct Buf { void *p; size_t size; int device; };   // device: the current device of the thread that pinned it (the mapping is made for that one)
    std::mutex mu;
    std::vector<Buf> idle;
    std::unordered_map<void *, Buf> out;                 // buffers handed out: their true size and device
    size_t idle_bytes = 0;
};
PinnedCache g_pinned;

}  // namespace

namespace nmres {

hipError_t pinned_take(void **p, size_t bytes) {
    *p = nullptr;
    const size_t want = (bytes + ((1u << 20) - 1)) & ~(size_t)((1u << 20) - 1);      // whole MB: the parsers' sizes differ by a few KB
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); device = 0; }
    {
        std::lock_guard<std::mutex> lk(g_pinned.mu);
        int best = -1;
        for (int i = 0; i < (int)g_pinned.idle.size(); ++i)
            if (g_pinned.idle[i].device == device && g_pinned.idle[i].size >= want && g_pinned.idle[i].size <= 2 * want &&
                (best < 0 || g_pinned.idle[i].size < g_pinned.idle[best].size))
                best = i;
        if (best >= 0) {
            const PinnedCache::Buf b = g_pinned.idle[best];
            g_pinned.idle.erase(g_pinned.idle.begin() + best);
            g_pinned.idle_bytes -= b.size;
            g_pinned.out[b.p] = b;
            *p = b.p;
            return hipSuccess;
        }
    }
    void *q = nullptr;
    hipError_t e = hipHostMalloc(&q, want, hipHostMallocDefault);
    if (e != hipSuccess) {                                // pinned memory is short: give the idle buffers back and try once more
        (void)hipGetLastError();
        pinned_trim();
        e = hipHostMalloc(&q, want, hipHostMallocDefault);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(g_pinned.mu);
    g_pinned.out[q] = PinnedCache::Buf{q, want, device};
    *p = q;
    return hipSuccess;
}

void pinned_give(void *p) {
    if (!p) return;
    static const bool keep = getenv("NM_NO_PINNED_CACHE") == nullptr;        // (A/B: every buffer is freed at once)
    if (keep) {
        std::lock_guard<std::mutex> lk(g_pinned.mu);
        auto it = g_pinned.out.find(p);
        if (it != g_pinned.out.end()) {
            const PinnedCache::Buf b = it->second;
            g_pinned.out.erase(it);
            if (g_pinned.idle_bytes + b.size <= PINNED_KEEP_BYTES) {
                g_pinned.idle.push_back(b);
                g_pinned.idle_bytes += b.size;
                return;
            }
        }
    } else {
        std::lock_guard<std::mutex> lk(g_pinned.mu);
        g_pinned.out.erase(p);
    }
    (void)hipHostFree(p);
}

void pinned_trim() {
    std::vector<PinnedCache::Buf> drop;
    {
        std::lock_guard<std::mutex> lk(g_pinned.mu);
        drop.swap(g_pinned.idle);
        g_pinned.idle_bytes = 0;
    }
    for (const PinnedCache::Buf &b : drop) (void)hipHostFree(b.p);
}

}  // namespace nmres

extern "C" {

int nm_block_cache(int enable, uint64_t max_idle_bytes, uint64_t stats[4]) {
    if (stats) {
        std::lock_guard<std::mutex> lk(g_cache.mu);
        stats[0] = g_cache.hits;
        stats[1] = g_cache.misses;
        stats[2] = g_cache.idle_bytes;
        stats[3] = g_cache.live.size();
    }
    if (enable < 0) return NM_OK;                        // statistics only
    if (enable) {
        if (!g_installed) {
            const int rc = nm_set_device_allocator(cache_alloc, cache_free, nullptr);
            if (rc) return rc;
            g_installed = true;
        }
        // never more than a quarter of the device: what lies idle here is invisible to the other users of hipMalloc in the process
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) max_idle_bytes = std::min<uint64_t>(max_idle_bytes, total_b / 4);
        else (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(g_cache.mu);
        g_cache.max_idle = (size_t)max_idle_bytes;
        return NM_OK;
    }
    if (g_installed) {
        {
            std::lock_guard<std::mutex> lk(g_cache.mu);
            if (!g_cache.live.empty())
                return nm_set_error(NM_ESTATE, "nm_block_cache(0): %zu blocks of the cache are still in use", g_cache.live.size());
        }
        const int rc = nm_set_device_allocator(nullptr, nullptr, nullptr);
        if (rc) return rc;
        g_installed = false;
    }
    g_cache.trim(0);
    nmres::pinned_trim();
    return NM_OK;
}

}  // extern "C"
