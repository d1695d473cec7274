// zl_host.h -- host-side plumbing shared by the two halves of the C-ABI: grow-only device and
// pinned-host buffers, device probing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <pthread.h>

#include "zl_format.h"

namespace {

// Development builds (-DZL_DEV_SWITCHES: `python -m zstdlite_amd.build --variant dev -DZL_DEV_SWITCHES`) read a few
// environment switches; the product build reads none.  ZL_DEBUG=1 prints every failing HIP call to stderr.
static inline bool zl_hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
#ifdef ZL_DEV_SWITCHES
    static const bool dbg = getenv("ZL_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[zl_hip] %s failed: %s\n", what, hipGetErrorString(e));
#else
    (void)what;
#endif
    return false;
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + (n >> 3) + 256;
        if (!zl_hip_ok(hipMalloc(&p, want), "hipMalloc")) { p = nullptr; return false; }
        cap = want;
        // ZL_POISON_MEM=1 in the environment fills every new device buffer with a pattern: a kernel that
        // reads scratch it has not written then fails reproducibly instead of depending on what the
        // allocation held before (debug aid)
#ifdef ZL_DEV_SWITCHES
        static const bool poison = getenv("ZL_POISON_MEM") != nullptr;
        if (poison) { (void)hipMemset(p, 0xEE, want); (void)hipDeviceSynchronize(); }  // (the contexts' streams do not wait for the null stream)
#endif
        return true;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool ensure(size_t n) {
        if (n <= cap) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + (n >> 3) + 256;
        if (!zl_hip_ok(hipHostMalloc(&p, want, hipHostMallocDefault), "hipHostMalloc")) { p = nullptr; return false; }
        cap = want;
        return true;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

// growable host byte buffer that does not zero what it grows by (a std::vector would: 64 MiB per streaming job)
struct ByteBuf {
    u8* p = nullptr;
    size_t n = 0, cap = 0;
    ByteBuf() = default;
    ByteBuf(const ByteBuf&) = delete;
    ByteBuf& operator=(const ByteBuf&) = delete;
    ~ByteBuf() { free(p); }
    bool reserve(size_t want) {
        if (want <= cap) return true;
        size_t c = cap ? cap : 4096;
        while (c < want) c += c >> 1;
        u8* q = (u8*)realloc(p, c);
        if (!q) return false;
        p = q;
        cap = c;
        return true;
    }
    u8* grow(size_t k) {  // k uninitialised bytes at the end
        if (!reserve(n + k)) return nullptr;
        u8* at = p + n;
        n += k;
        return at;
    }
    bool append(const void* src, size_t k);  // (defined behind zl_memcpy_mt)
    void drop_front(size_t k) {
        if (k >= n) { n = 0; return; }
        memmove(p, p + k, n - k);
        n -= k;
    }
    void clear() { n = 0; }
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
};

// Two copy streams that use the link in BOTH directions at once.  Measured on this platform (round 2):
// a stream is tied to a copy engine by its first copy of a MiB or more -- the engine that is idle at that moment -- and keeps it;
// an upload stream and a download stream whose first large copies followed each other share ONE engine and take turns for good
// (1 GiB up + 0.65 GiB down: 31 ms instead of 21).  So the pair is bound on purpose: first copies from pinned memory, overlapping
// in time; then checked by timing (together against one after the other).  Returns true when the pair overlaps.
// pin / dev: scratch of at least 2 * bytes each (pinned host, device).
static inline bool zl_bind_copy_pair(hipStream_t up, hipStream_t down, void* pin, void* dev, size_t bytes) {
    u8* hp = (u8*)pin;
    u8* dp = (u8*)dev;
    auto both = [&]() {
        (void)hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, up);
        (void)hipMemcpyAsync(hp + bytes, dp + bytes, bytes, hipMemcpyDeviceToHost, down);
        (void)hipStreamSynchronize(up);
        (void)hipStreamSynchronize(down);
    };
    {   // the binding copies: a stream keeps the engine its first copy took, and that copy takes the first engine that is idle when
        // it is issued -- so the second stream's first copy is issued while the first stream's is certainly under way (issued
        // right behind it, it often found the engine still idle: the pair then took turns for good; round 4)
        (void)hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, up);
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        do { clock_gettime(CLOCK_MONOTONIC, &t1); } while ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) < 100000);
        (void)hipMemcpyAsync(hp + bytes, dp + bytes, bytes, hipMemcpyDeviceToHost, down);
        (void)hipStreamSynchronize(up);
        (void)hipStreamSynchronize(down);
    }
    auto now = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    double t_up = 1e9, t_dn = 1e9, t_both = 1e9;
    for (int k = 0; k < 2; k++) {
        double t0 = now();
        (void)hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, up); (void)hipStreamSynchronize(up);
        double t1 = now();
        (void)hipMemcpyAsync(hp + bytes, dp + bytes, bytes, hipMemcpyDeviceToHost, down); (void)hipStreamSynchronize(down);
        double t2 = now();
        both();
        double t3 = now();
        t_up = t1 - t0 < t_up ? t1 - t0 : t_up; t_dn = t2 - t1 < t_dn ? t2 - t1 : t_dn; t_both = t3 - t2 < t_both ? t3 - t2 : t_both;
    }
#ifdef ZL_DEV_SWITCHES
    if (getenv("ZL_DEBUG_SLAB")) fprintf(stderr, "[zl bind] %zu MiB: up %.3f ms, down %.3f ms, both %.3f ms\n", bytes >> 20, t_up, t_dn, t_both);
#endif
    // (together in less than three quarters of one after the other -- and neither direction on one of the slow engines)
    return t_both < 0.75 * (t_up + t_dn) && t_dn < 1.3 * t_up && t_up < 1.3 * t_dn;
}

}  // namespace

// A few host threads that stay around between calls, for the per-object loops of the batch entries (pack, header parse,
// unpack: 100k memcpys of a few hundred bytes).  Starting seven threads costs 0.2-0.5 ms, which is what such a loop takes
// on one thread; waking sleeping ones costs a tenth of that.  One user at a time (a second caller runs its loop itself);
// the workers are never joined (the object is leaked on purpose: they sleep on it until the process ends -- which is why the
// library is linked with -z nodelete: unloading it would unmap the code they sleep in); a forked child has none of them and
// runs its loops on the calling thread.
struct __attribute__((visibility("hidden"))) ZlPool {
    static const size_t kWorkers = 15;  // (seven until round 6: pack + unpack of a 100 000-object call 1.5 -> ms; a host with fewer cores uses as many as it has)
    std::mutex m, user;
    std::condition_variable cv_work, cv_done;
    void (*fn)(void*, size_t) = nullptr;
    void* ctx = nullptr;
    size_t gen = 0, want = 0, done = 0;
    void worker(size_t k) {
        size_t seen = 0;
        std::unique_lock<std::mutex> l(m);
        for (;;) {
            cv_work.wait(l, [&] { return gen != seen; });
            seen = gen;
            if (k >= want) continue;
            void (*f)(void*, size_t) = fn;
            void* c = ctx;
            l.unlock();
            f(c, k);
            l.lock();
            if (++done == want) cv_done.notify_one();
        }
    }
    // f(ctx, k) for k in [0, parts): part 0 on the caller, the others on the workers
    void run(void (*f)(void*, size_t), void* c, size_t parts) {
        {
            std::lock_guard<std::mutex> l(m);
            fn = f; ctx = c; want = parts - 1; done = 0; gen++;
        }
        cv_work.notify_all();
        f(c, parts - 1);
        std::unique_lock<std::mutex> l(m);
        cv_done.wait(l, [&] { return done == want; });
    }
};
// ONE pool per process: the struct and the two variables have external linkage (inline, C++17), so the translation units that
// include this header share them (until round 4 they sat inside the anonymous namespace below: a pool per API file, 14 parked threads).
__attribute__((visibility("hidden"))) inline ZlPool* g_pool = nullptr;
__attribute__((visibility("hidden"))) inline std::once_flag g_pool_once;
__attribute__((visibility("hidden"))) inline ZlPool* zl_pool() {
    std::call_once(g_pool_once, []() {
        pthread_atfork(nullptr, nullptr, []() { g_pool = nullptr; });  // (the child has none of the workers: it runs its loops itself)
        ZlPool* p = new (std::nothrow) ZlPool();
        if (!p) return;
        for (size_t k = 0; k < ZlPool::kWorkers; k++) {
            try { std::thread(&ZlPool::worker, p, k).detach(); } catch (...) { return; }  // (no threads to be had: stay serial)
        }
        g_pool = p;
    });
    return g_pool;
}

namespace {

// f(begin, end) over [0, n) on up to 16 host threads (the calling one among them); ranges of at least `grain` items, so
// small batches stay on the caller.
template <class F>
static inline void zl_parallel_for(size_t n, size_t grain, F f, size_t max_threads = ZlPool::kWorkers + 1) {
    const size_t hw = std::thread::hardware_concurrency();
    const size_t t = std::min<size_t>(std::min<size_t>(std::min<size_t>(ZlPool::kWorkers + 1, max_threads), hw ? hw : 1), grain ? n / grain : 1);
    ZlPool* pool = t > 1 ? zl_pool() : nullptr;
    if (!pool || !pool->user.try_lock()) { f((size_t)0, n); return; }
    struct Ctx { F* f; size_t n, per; } cx{&f, n, (n + t - 1) / t};
    pool->run([](void* v, size_t k) {
        Ctx* c = (Ctx*)v;
        const size_t b = k * c->per, e = std::min(c->n, b + c->per);
        if (b < e) (*c->f)(b, e);
    }, &cx, t);
    pool->user.unlock();
}

// The batch entries (zl_compress_batch, zl_decompress_batch) pack and unpack their objects on the pool: in pieces, a piece's copy issued as
// soon as it is packed; objects per task of a piece.
#ifndef ZL_BATCH_PIECES
#define ZL_BATCH_PIECES 4
#endif
#ifndef ZL_BATCH_GRAIN
#define ZL_BATCH_GRAIN 1024
#endif

// memcpy of a large block on the pool's threads: one core copies at 10 - 15 GB/s, which is what the streaming entries' job
// buffers ran at (64 MiB in and 40 MiB out per job, 8 ms of an 11 ms job); small blocks stay on the caller.
static inline void zl_memcpy_mt(void* dst, const void* src, size_t n) {
    if (n < ((size_t)2 << 20)) { if (n) memcpy(dst, src, n); return; }
    // (eight threads, as rounds 1-5: the streaming entries copy pieces of a few MiB, and waking sixteen threads for each of them cost the pledged stream a tenth of its rate)
    zl_parallel_for(n, (size_t)512 << 10, [&](size_t b, size_t e) { memcpy((u8*)dst + b, (const u8*)src + b, e - b); }, 8);
}

inline bool ByteBuf::append(const void* src, size_t k) {
    if (k == 0) return true;  // (nothing to add to a buffer that may not have been allocated yet)
    u8* at = grow(k);
    if (!at) return false;
    zl_memcpy_mt(at, src, k);
    return true;
}

static int g_device_state = 0;  // 0 unknown, 1 ok, -1 unavailable
static int g_num_cu = 256;
static bool device_ok() {
    if (g_device_state == 0) {
        int n = 0;
        if (!zl_hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount") || n <= 0) {
            g_device_state = -1;
        } else {
            hipDeviceProp_t prop;
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cu = prop.multiProcessorCount;
            g_device_state = 1;
        }
    }
    return g_device_state == 1;
}

static const size_t kGpuUnavailable = ZL_ERR(ZL_E_MEMORY);


}  // namespace
