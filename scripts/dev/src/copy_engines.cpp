// What makes an upload and a download on two HIP streams run at the same time on this machine?  (zl_bind_copy_pair, zl_host.h)
// build: hipcc -O2 -o zstdlite_amd/build/variants/copy_engines scripts/dev/src/copy_engines.cpp ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static void spin_us(double us) { const double t0 = now(); while ((now() - t0) * 1e3 < us) {} }
static char *hp, *dp;
static const size_t B = (size_t)32 << 20;
struct R { double up, dn, both; };
static R measure(hipStream_t up, hipStream_t down) {
    R r{1e9, 1e9, 1e9};
    for (int k = 0; k < 3; k++) {
        double t0 = now();
        hipMemcpyAsync(dp, hp, B, hipMemcpyHostToDevice, up); hipStreamSynchronize(up);
        double t1 = now();
        hipMemcpyAsync(hp + B, dp + B, B, hipMemcpyDeviceToHost, down); hipStreamSynchronize(down);
        double t2 = now();
        hipMemcpyAsync(dp, hp, B, hipMemcpyHostToDevice, up);
        hipMemcpyAsync(hp + B, dp + B, B, hipMemcpyDeviceToHost, down);
        hipStreamSynchronize(up); hipStreamSynchronize(down);
        double t3 = now();
        if (t1 - t0 < r.up) r.up = t1 - t0;
        if (t2 - t1 < r.dn) r.dn = t2 - t1;
        if (t3 - t2 < r.both) r.both = t3 - t2;
    }
    return r;
}
static void show(const char* what, R r) { printf("%-58s up %.2f dn %.2f both %.2f ms  -> %s\n", what, r.up, r.dn, r.both, r.both < 0.75 * (r.up + r.dn) ? "TWO ENGINES" : "taking turns"); fflush(stdout); }
int main(int argc, char** argv) {
    hipHostMalloc((void**)&hp, 2 * B); hipMalloc((void**)&dp, 2 * B);
    for (size_t i = 0; i < 2 * B; i += 4096) hp[i] = 1;
    const int extra = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<hipStream_t> dummies(extra);
    for (auto& s : dummies) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int trial = 0; trial < 4; trial++) {
        hipStream_t a, b;
        hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
        show("A fresh pair, first copies back to back", measure(a, b));
        hipStreamDestroy(a); hipStreamDestroy(b);
    }
    for (int trial = 0; trial < 4; trial++) {
        hipStream_t a, b;
        hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
        hipMemcpyAsync(dp, hp, B, hipMemcpyHostToDevice, a); spin_us(150); hipMemcpyAsync(hp + B, dp + B, B, hipMemcpyDeviceToHost, b);
        hipStreamSynchronize(a); hipStreamSynchronize(b);
        show("B first copies 150 us apart (second while first runs)", measure(a, b));
        hipStreamDestroy(a); hipStreamDestroy(b);
    }
    for (int trial = 0; trial < 4; trial++) {
        hipStream_t a, b;
        hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
        hipMemcpyAsync(dp, hp, B, hipMemcpyHostToDevice, a); hipStreamSynchronize(a);
        hipMemcpyAsync(hp + B, dp + B, B, hipMemcpyDeviceToHost, b); hipStreamSynchronize(b);
        show("C first copies one after the other (never together)", measure(a, b));
        hipStreamDestroy(a); hipStreamDestroy(b);
    }
    for (int trial = 0; trial < 4; trial++) {
        hipStream_t a, b;
        hipStreamCreateWithPriority(&a, hipStreamNonBlocking, 0); hipStreamCreateWithPriority(&b, hipStreamNonBlocking, -1);
        show("D two priorities, back to back", measure(a, b));
        hipStreamDestroy(a); hipStreamDestroy(b);
    }
    {   // E: a pool of streams: which pairs overlap?
        const int N = 6;
        hipStream_t s[N];
        for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
        for (int i = 0; i < N; i++) { hipMemcpyAsync(dp, hp, 1 << 20, hipMemcpyHostToDevice, s[i]); hipStreamSynchronize(s[i]); }
        for (int i = 0; i < 1; i++) for (int j = 0; j < N; j++) if (i != j) { char w[64]; snprintf(w, sizeof w, "E pool of %d, up on %d, down on %d", N, i, j); show(w, measure(s[i], s[j])); }
        for (auto& x : s) hipStreamDestroy(x);
    }
    {   // E2: the default pool is "polluted" now (every queue's first copy was an upload): what does a fresh default pair get?
        const int N = 6;
        hipStream_t s[N];
        for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
        for (int i = 0; i < N; i++) { hipMemcpyAsync(dp, hp, 1 << 20, hipMemcpyHostToDevice, s[i]); hipStreamSynchronize(s[i]); }
        for (int trial = 0; trial < 3; trial++) {
            hipStream_t a, b;
            hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
            show("E2 fresh default pair beside 6 upload-first streams", measure(a, b));
            hipStreamDestroy(a); hipStreamDestroy(b);
        }
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        printf("priority range: least %d greatest %d\n", lo, hi);
        for (int trial = 0; trial < 3; trial++) {
            hipStream_t a, b;
            hipStreamCreateWithPriority(&a, hipStreamNonBlocking, lo); hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi);
            show("E3 low-priority up, high-priority down, same state", measure(a, b));
            hipStreamDestroy(a); hipStreamDestroy(b);
        }
        for (int trial = 0; trial < 3; trial++) {
            hipStream_t a, b;
            hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi);
            show("E4 default up, high-priority down, same state", measure(a, b));
            hipStreamDestroy(a); hipStreamDestroy(b);
        }
        for (auto& x : s) hipStreamDestroy(x);
        for (int trial = 0; trial < 3; trial++) {
            hipStream_t a, b;
            hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
            show("E5 fresh default pair after the 6 are destroyed", measure(a, b));
            hipStreamDestroy(a); hipStreamDestroy(b);
        }
    }
    {   // H: what does pinning the caller's pages cost?
        const size_t n = (size_t)1 << 30;
        char* m = (char*)malloc(n);
        for (size_t i = 0; i < n; i += 4096) m[i] = 1;
        double t0 = now();
        hipError_t e = hipHostRegister(m, n, hipHostRegisterDefault);
        double t1 = now();
        hipHostUnregister(m);
        double t2 = now();
        printf("H hipHostRegister of 1 GiB: %.1f ms (%s), unregister %.1f ms\n", t1 - t0, hipGetErrorString(e), t2 - t1);
        hipStream_t a; hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
        for (int k = 0; k < 3; k++) { t0 = now(); hipMemcpyAsync(m, dp, B, hipMemcpyDeviceToHost, a); hipStreamSynchronize(a); t1 = now(); printf("H pageable download of 32 MiB: %.2f ms\n", t1 - t0); }
        for (int k = 0; k < 3; k++) { t0 = now(); hipMemcpyAsync(dp, m, B, hipMemcpyHostToDevice, a); hipStreamSynchronize(a); t1 = now(); printf("H pageable upload of 32 MiB: %.2f ms\n", t1 - t0); }
        free(m);
    }
    {   // F: the same stream for both
        hipStream_t a; hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
        show("F one stream for both (the floor)", measure(a, a));
    }
    {   // G: pageable memory, two threads is what the library does; here only: does a D2H kernel copy overlap an SDMA upload?
    }
    return 0;
}
