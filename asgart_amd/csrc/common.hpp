// common.hpp -- shared host/device definitions of libasgart_hip (gfx950 only).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <cstdio>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/asgart_hip.h"

namespace asgart {

void set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            ::asgart::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                                __LINE__);                                                  \
            return e_ == hipErrorOutOfMemory ? ASGART_E_OOM : ASGART_E_HIP;                 \
        }                                                                                   \
    } while (0)

// hipStreamSynchronize for the waits outside the search calls' own watchdog (index preparation, suffix sort, list builds,
// post-processing): polls, and gives up after ASGART_BUILD_WATCHDOG_S seconds (default 900; 0 = wait forever) with
// hipErrorLaunchTimeOut, which HIP_TRY turns into an error that names the file and line of the wait -- a device that stops
// answering during a build becomes an error with a place instead of a process that hangs.
inline hipError_t stream_sync(hipStream_t s) {
    static const double limit = [] {
        const char *e = getenv("ASGART_BUILD_WATCHDOG_S");
        return e ? atof(e) : 900.0;
    }();
    if (limit <= 0.0) return hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
        const hipError_t q = hipStreamQuery(s);
        if (q != hipErrorNotReady) return q;
        (void)hipGetLastError();
        if (spins > 256) std::this_thread::sleep_for(std::chrono::microseconds(spins > 4096 ? 200 : 20));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return hipErrorLaunchTimeOut;
    }
}

// A device-to-host copy into PAGEABLE memory blocks inside the copy call until everything queued on the stream before it
// has run -- the polled wait behind it then finds an idle stream and its limit never applies.  Builders read their
// counters back through this: the stream is drained by the polled wait FIRST, the copy then has nothing to wait for.
inline hipError_t read_back(void *dst, const void *src, size_t bytes, hipStream_t s) {
    hipError_t e = stream_sync(s);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    return stream_sync(s);
}

// Runs fn() on the process's teardown thread and waits for it at most limit_s seconds (<= 0: runs it inline, however long
// it takes).  For runtime calls that cannot be polled (hipFree, hipHostFree, hipStreamDestroy: each synchronises with the
// device inside the runtime).  false: fn did not return in time -- the thread is left behind in it (a fresh one serves the
// next call) and whatever fn captured must stay alive; the caller reports and gives up on the rest of its work.
// ONE long-lived thread, not one per call: every host thread that makes HIP calls costs the runtime some device memory
// that it does not give back when the thread ends (measured: 70 create/destroy cycles with five helper threads each
// lowered the free device memory by 16 MB).
struct TeardownWorker {
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::function<void()>> q;
    bool stuck = false;
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return !q.empty(); });
            std::function<void()> j = std::move(q.front());
            q.erase(q.begin());
            lk.unlock();
            j();
            lk.lock();
        }
    }
    static std::shared_ptr<TeardownWorker> current() {
        static std::mutex gm;
        static std::shared_ptr<TeardownWorker> w;
        std::lock_guard<std::mutex> lk(gm);
        bool fresh = !w;
        if (w) {
            std::lock_guard<std::mutex> lk2(w->m);
            fresh = w->stuck;
        }
        if (fresh) {
            w = std::make_shared<TeardownWorker>();
            std::shared_ptr<TeardownWorker> keep = w;  // (the thread keeps its worker alive)
            std::thread([keep]() { keep->loop(); }).detach();
        }
        return w;
    }
};
// fn() on the worker, waited for at most limit_s seconds COUNTED FROM WHEN IT STARTS TO RUN: the queue is shared by every
// index and device of the process, and a slow but healthy job of another caller ahead of this one (a hipFree of tens of
// GB) must not make this caller give up on -- and leak -- a healthy index.  While the job waits in the queue the caller
// waits with it, up to ten limits in all (the job ahead has a caller of its own who bounds it).  The worker is marked
// stuck only when the caller's OWN job overran.
template <class F>
inline bool bounded_call(double limit_s, F fn) {
    if (limit_s <= 0.0) {
        fn();
        return true;
    }
    struct State {
        std::mutex m;
        std::condition_variable cv;
        bool done = false, started = false;
        std::chrono::steady_clock::time_point t_start;
    };
    auto st = std::make_shared<State>();
    std::shared_ptr<TeardownWorker> w = TeardownWorker::current();
    {
        std::lock_guard<std::mutex> lk(w->m);
        w->q.emplace_back([st, fn]() mutable {
            {
                std::lock_guard<std::mutex> lk2(st->m);
                st->started = true;
                st->t_start = std::chrono::steady_clock::now();
            }
            st->cv.notify_all();
            fn();
            {
                std::lock_guard<std::mutex> lk2(st->m);
                st->done = true;
            }
            st->cv.notify_all();
        });
    }
    w->cv.notify_all();
    const auto limit = std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(limit_s));
    const auto t_queued = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(st->m);
    bool overran = false;
    while (!st->done) {
        if (st->started) {
            if (!st->cv.wait_until(lk, st->t_start + limit, [&] { return st->done; })) {
                overran = true;
                break;
            }
        } else {
            st->cv.wait_for(lk, std::chrono::milliseconds(50), [&] { return st->done || st->started; });
            if (!st->started && std::chrono::steady_clock::now() - t_queued > 10 * limit) break;  // (never got its turn)
        }
    }
    const bool ok = st->done;
    lk.unlock();
    if (overran) {
        std::lock_guard<std::mutex> lk2(w->m);
        w->stuck = true;
    }
    return ok;
}
// fn() on the worker, not waited for at all (housekeeping that must not hold a caller: giving cached blocks back)
template <class F>
inline void background_call(F fn) {
    std::shared_ptr<TeardownWorker> w = TeardownWorker::current();
    {
        std::lock_guard<std::mutex> lk(w->m);
        w->q.emplace_back(fn);
    }
    w->cv.notify_all();
}

#define RC_TRY(expr)            \
    do {                        \
        int32_t rc_ = (expr);   \
        if (rc_ != 0) return rc_; \
    } while (0)

// Grow-only device buffer (workspace reuse across calls: no hipMalloc in the
// steady state).
// ASGART_TRACE_ALLOC=1 (diagnostics): every device allocation / release of 1 GiB or more with its duration
inline bool trace_alloc() {
    static const bool on = getenv("ASGART_TRACE_ALLOC") != nullptr;
    return on;
}
struct AllocTimer {
    const char *what;
    size_t bytes;
    std::chrono::steady_clock::time_point t0;
    AllocTimer(const char *w, size_t b) : what(w), bytes(b), t0(std::chrono::steady_clock::now()) {}
    ~AllocTimer() {
        if (trace_alloc() && bytes >= ((size_t)1 << 30))
            fprintf(stderr, "[asgart] %s %6.1f GiB: %8.1f ms\n", what, (double)bytes / (double)((size_t)1 << 30),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// Block cache behind DevBuf.  hipMalloc / hipFree of tens of GB are sub-millisecond by themselves on this runtime, but
// the first large hipMalloc after a lot of memory went back to the driver takes SECONDS (tools/ubench_malloc.hip:
// 1.5 - 5.5 s for 26 GB after 26 - 150 GB were freed) unless it can take over a block of the same size.  The index
// build frees ~150 GB of sorter scratch and allocates ~60 GB right after: released blocks of 256 MiB and more are
// therefore kept (per device, up to kCacheCap bytes: what a build takes back, not all it releases) and handed to the
// next request they fit with little waste.  The cache never outlives its purpose: it is emptied when an allocation
// fails, at the end of asgart_index_prepare (the sorter's scratch is of no use to the search calls), when the last
// index of the device is destroyed (live counts below) and by asgart_trim_cache -- other allocators on the GPU (the
// host application's tensors, RCCL buffers, other processes) cannot see what sits here.
struct BlockCache {
    // (the cap holds ALL the scratch of a GRCh38-sized suffix sort, 127 GB: what a build releases beyond the cap goes back to
    // the driver at once, and on most boxes of the pool the NEXT hipMalloc then pays 20-30 ms for every GiB that was freed --
    // ASGART_TRACE_ALLOC=1: "hipMalloc 6.4 GiB: 2403 ms" behind 82 GiB of frees)
    static constexpr size_t kCacheMin = (size_t)256 << 20, kCacheCap = (size_t)160 << 30;
    struct Dev {
        std::mutex mu;
        std::vector<std::pair<void *, size_t>> blocks;
        size_t bytes = 0;
        int live = 0;  // indexes alive on this device
    };
    static Dev &dev() {
        static Dev d[16];
        int id = 0;
        (void)hipGetDevice(&id);
        return d[(id >= 0 && id < 16) ? id : 0];
    }
    // a cached block of at least `bytes` that wastes at most a quarter (waste_pct = 25); nullptr: none
    static void *take(size_t bytes, size_t *cap, unsigned waste_pct = 25) {
        if (bytes < kCacheMin) return nullptr;
        Dev &d = dev();
        std::lock_guard<std::mutex> lk(d.mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < d.blocks.size(); ++i)
            if (d.blocks[i].second >= bytes && d.blocks[i].second <= bytes + bytes / 100 * waste_pct &&
                (best == (size_t)-1 || d.blocks[i].second < d.blocks[best].second))
                best = i;
        if (best == (size_t)-1) return nullptr;
        void *p = d.blocks[best].first;
        *cap = d.blocks[best].second;
        d.bytes -= *cap;
        d.blocks.erase(d.blocks.begin() + (ptrdiff_t)best);
        return p;
    }
    static bool give(void *p, size_t cap) {  // false: not kept (the caller frees it)
        if (cap < kCacheMin) return false;
        Dev &d = dev();
        std::lock_guard<std::mutex> lk(d.mu);
        if (d.bytes + cap > kCacheCap) return false;
        d.blocks.emplace_back(p, cap);
        d.bytes += cap;
        return true;
    }
    // gives cached blocks back to the device, largest first, until at most `keep` bytes are held
    static void trim(size_t keep = 0) {
        Dev &d = dev();
        std::vector<std::pair<void *, size_t>> out;
        {
            std::lock_guard<std::mutex> lk(d.mu);
            while (d.bytes > keep && !d.blocks.empty()) {
                size_t big = 0;
                for (size_t i = 1; i < d.blocks.size(); ++i)
                    if (d.blocks[i].second > d.blocks[big].second) big = i;
                out.push_back(d.blocks[big]);
                d.bytes -= d.blocks[big].second;
                d.blocks.erase(d.blocks.begin() + (ptrdiff_t)big);
            }
        }
        for (auto &b : out) {
            AllocTimer tm("hipFree  ", b.second);
            (void)hipFree(b.first);
        }
    }
    static size_t held() {
        Dev &d = dev();
        std::lock_guard<std::mutex> lk(d.mu);
        return d.bytes;
    }
    // indexes alive on the current device: the last one to go empties the cache
    static void index_born() {
        Dev &d = dev();
        std::lock_guard<std::mutex> lk(d.mu);
        ++d.live;
    }
    static void index_gone() {
        Dev &d = dev();
        bool last;
        {
            std::lock_guard<std::mutex> lk(d.mu);
            last = --d.live <= 0;
            if (last) d.live = 0;
        }
        if (last) trim();
    }
    static int live_indexes() {
        Dev &d = dev();
        std::lock_guard<std::mutex> lk(d.mu);
        return d.live;
    }
};

// Test hook (option test_fail_alloc = n): the (n+1)-th device allocation of the process from now on fails as if the
// device were full, once; -1 = off.  Exercises the out-of-memory paths (what is released, what degrades).
inline std::atomic<long long> &fail_alloc_countdown() {
    static std::atomic<long long> c{-1};
    return c;
}
inline bool test_alloc_fails() {
    std::atomic<long long> &c = fail_alloc_countdown();
    if (c.load(std::memory_order_relaxed) < 0) return false;
    return c.fetch_sub(1) == 0;
}

// hipMalloc / hipFree for the long-lived arrays of an index, through the same cache (sizes remembered per pointer)
inline std::mutex &raw_mu() { static std::mutex m; return m; }
inline std::vector<std::pair<void *, size_t>> &raw_sizes() { static std::vector<std::pair<void *, size_t>> v; return v; }
inline hipError_t dev_malloc(void **out, size_t bytes) {
    if (test_alloc_fails()) return hipErrorOutOfMemory;
    size_t cap = 0;
    void *p = BlockCache::take(bytes, &cap);
    hipError_t e = hipSuccess;
    if (!p) {
        cap = bytes;
        {
            AllocTimer tm("hipMalloc", bytes);
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            BlockCache::trim();
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) return e;
    }
    {
        std::lock_guard<std::mutex> lk(raw_mu());
        raw_sizes().emplace_back(p, cap);
    }
    *out = p;
    return hipSuccess;
}
inline void dev_free(void *p) {
    if (!p) return;
    size_t cap = 0;
    {
        std::lock_guard<std::mutex> lk(raw_mu());
        auto &v = raw_sizes();
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].first == p) {
                cap = v[i].second;
                v.erase(v.begin() + (ptrdiff_t)i);
                break;
            }
    }
    if (cap && BlockCache::give(p, cap)) return;
    AllocTimer tm("hipFree  ", cap);
    (void)hipFree(p);
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool view = false;  // a piece of another buffer (Workspace::arena): nothing to free; outgrown, it gets memory of its own
    int32_t reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        release();
        if (test_alloc_fails()) {
            set_error("hipMalloc(%zu bytes) failed: injected by option test_fail_alloc", bytes);
            return ASGART_E_OOM;
        }
        size_t want = bytes + bytes / 8 + 256;
        if ((p = BlockCache::take(want, &cap)) != nullptr) return 0;
        hipError_t e;
        {
            AllocTimer tm("hipMalloc", want);
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) {  // give the cached blocks back, ask for no slack
            (void)hipGetLastError();
            BlockCache::trim();
            e = hipMalloc(&p, bytes + 256);
            want = bytes + 256;
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
            return ASGART_E_OOM;
        }
        cap = want;
        return 0;
    }
    void release() {
        if (p && !view && !BlockCache::give(p, cap)) {
            AllocTimer tm("hipFree  ", cap);
            (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
        view = false;
    }
    template <class T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
};

// ---- base codes ----------------------------------------------------------
// 3-bit codes that preserve the byte order of the reference's alphabet
// ('$' 0x24 < 'A' < 'C' < 'G' < 'N' < 'T'), so that integer comparison of
// packed k-mers == bytewise comparison of the k-mers (reference
// src/searcher.rs:150 `a.cmp(b)`).
//   '$' (or past the end of the text) = 0, A=1, C=2, G=3, N=4, T=5
__host__ __device__ inline uint32_t base_code(uint8_t c) {
    // h = (c>>1)&7 : A->0 C->1 T->2 G->3 N->7
    const uint32_t lut = (1u << 0) | (2u << 4) | (5u << 8) | (3u << 12) | (4u << 28);
    return c < 0x40 ? 0u : ((lut >> (4 * ((c >> 1) & 7))) & 7u);
}
// complement on codes (reference src/utils.rs:1-17): A<->T, C<->G, N->N
__host__ __device__ inline uint32_t comp_code(uint32_t code) {
    const uint32_t lut = (0u) | (5u << 4) | (3u << 8) | (2u << 12) | (4u << 16) | (1u << 20);
    return (lut >> (4 * code)) & 7u;
}
// code -> 2-bit digit for the ACGT-only prefix table; 4 = not representable
__host__ __device__ inline uint32_t acgt_digit(uint32_t code) {
    const uint32_t lut = (4u) | (0u << 4) | (1u << 8) | (2u << 12) | (4u << 16) | (3u << 20);
    return (lut >> (4 * code)) & 7u;
}

constexpr int kMaxKey = 21;        // bases in one key word: 21 * 3 bits = 63
constexpr int kMaxK = 128;         // probe sizes above 21: the first 21 bases are the key word, the next 21 a second
                                   // word packed on demand, the rest is compared base by base through the text
                                   // (search_dev.hpp: tail_key, tail_cmp)
constexpr int kCacheLen = 8;       // reference src/searcher.rs:15
constexpr int kCacheEntries = 390625;  // 5^8
constexpr int kSmallInterval = 32;  // suffix-array intervals up to this size are handled by one thread
constexpr int kRankMin = 256;       // intervals above this size are counted by bisection when the index has position-sorted lists
constexpr uint32_t kRawUnknown = 0xFFFFFFFFu;  // p_raw of a probe the position filter answered: its interval was never looked up
constexpr uint32_t kSkipN = 0xFFFFFFFFu;     // probe skipped: first base 'N'
constexpr uint32_t kSkipCard = 0xFFFFFFFEu;  // probe skipped: > max_cardinality
constexpr uint32_t kPending = 0xFFFFFFFDu;   // large interval, counted by the wave kernel
constexpr uint32_t kPendingRank = 0xFFFFFFFCu;  // ... by bisection of its position-sorted list (rank_count_kernel); both
                                                // marks are gone when the probe search is over

inline bool valid_text_byte(uint8_t c) {
    return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N' || c == '$';
}

}  // namespace asgart
