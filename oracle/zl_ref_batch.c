/*
 * zl_ref_batch.c -- TEST / BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline side of config C4; nothing under
 * zstdlite_amd/ links or calls it).
 *
 * The reference's per-object loop -- vignettes/dictionaries.Rmd:129-138: lapply over the objects, zstd_serialize() with a
 * cctx that holds the dictionary (src/cctx.c:300-312 ZSTD_CCtx_loadDictionary, src/serialize.c:91 ZSTD_compress2) -- on
 * as many host threads as asked for, each with a context of its own over a contiguous share of the objects: what "the
 * bundled libzstd on every core" gives for many small objects.  Links against oracle/_ref/libzstd_ref.so (the reference's
 * own source, compiled where it lies) by name at run time; the prototypes below are zstd.h's (zstd.h:232,266,290,599; zstd.h:1040 for the
 * dictionary loaders).
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
#include <time.h>

#include <dlfcn.h>

/* the reference library's entry points, looked up in THAT file (zlr_init): a process that has PyTorch in it also holds the
 * system's libzstd 1.4.8, and plain linking could bind these names there */
static void* (*ZSTD_createCCtx)(void);
static size_t (*ZSTD_freeCCtx)(void*);
static size_t (*ZSTD_CCtx_setParameter)(void*, int, int);
static size_t (*ZSTD_CCtx_loadDictionary)(void*, const void*, size_t);
static size_t (*ZSTD_compress2)(void*, void*, size_t, const void*, size_t);
static void* (*ZSTD_createDCtx)(void);
static size_t (*ZSTD_freeDCtx)(void*);
static size_t (*ZSTD_DCtx_loadDictionary)(void*, const void*, size_t);
static size_t (*ZSTD_decompressDCtx)(void*, void*, size_t, const void*, size_t);

/* path: oracle/_ref/libzstd_ref.so.  Returns 0 when every entry point was found. */
int zlr_init(const char* path) {
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return 1;
    *(void**)&ZSTD_createCCtx = dlsym(h, "ZSTD_createCCtx");
    *(void**)&ZSTD_freeCCtx = dlsym(h, "ZSTD_freeCCtx");
    *(void**)&ZSTD_CCtx_setParameter = dlsym(h, "ZSTD_CCtx_setParameter");
    *(void**)&ZSTD_CCtx_loadDictionary = dlsym(h, "ZSTD_CCtx_loadDictionary");
    *(void**)&ZSTD_compress2 = dlsym(h, "ZSTD_compress2");
    *(void**)&ZSTD_createDCtx = dlsym(h, "ZSTD_createDCtx");
    *(void**)&ZSTD_freeDCtx = dlsym(h, "ZSTD_freeDCtx");
    *(void**)&ZSTD_DCtx_loadDictionary = dlsym(h, "ZSTD_DCtx_loadDictionary");
    *(void**)&ZSTD_decompressDCtx = dlsym(h, "ZSTD_decompressDCtx");
    return !(ZSTD_createCCtx && ZSTD_freeCCtx && ZSTD_CCtx_setParameter && ZSTD_CCtx_loadDictionary && ZSTD_compress2 && ZSTD_createDCtx && ZSTD_freeDCtx &&
             ZSTD_DCtx_loadDictionary && ZSTD_decompressDCtx);
}

typedef struct {
    int decompress, level;
    const void* dict; size_t dict_len;
    const unsigned char* in; const size_t* in_off; const size_t* in_size;
    unsigned char* out; const size_t* out_off; const size_t* out_cap; size_t* out_size;
    size_t first, last;
} zlr_job;

static void* zlr_work(void* arg) {
    zlr_job* j = (zlr_job*)arg;
    if (j->decompress) {
        void* d = ZSTD_createDCtx();
        if (j->dict_len) ZSTD_DCtx_loadDictionary(d, j->dict, j->dict_len);
        for (size_t i = j->first; i < j->last; i++) j->out_size[i] = ZSTD_decompressDCtx(d, j->out + j->out_off[i], j->out_cap[i], j->in + j->in_off[i], j->in_size[i]);
        ZSTD_freeDCtx(d);
    } else {
        void* c = ZSTD_createCCtx();
        ZSTD_CCtx_setParameter(c, 100 /* ZSTD_c_compressionLevel */, j->level);
        if (j->dict_len) ZSTD_CCtx_loadDictionary(c, j->dict, j->dict_len);
        for (size_t i = j->first; i < j->last; i++) j->out_size[i] = ZSTD_compress2(c, j->out + j->out_off[i], j->out_cap[i], j->in + j->in_off[i], j->in_size[i]);
        ZSTD_freeCCtx(c);
    }
    return NULL;
}

/* n objects (in + in_off / in_size) -> out + out_off (out_cap bytes each; sizes or error codes in out_size) on nthreads threads.
 * Returns the wall seconds from the first thread's start to the last one's end (contexts and dictionary loading included:
 * they are part of what every thread has to do once). */
double zlr_batch(int decompress, int level, int nthreads, const void* dict, size_t dict_len, size_t n, const unsigned char* in, const size_t* in_off,
                 const size_t* in_size, unsigned char* out, const size_t* out_off, const size_t* out_cap, size_t* out_size) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = (int)(n ? n : 1);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
    zlr_job* jobs = (zlr_job*)malloc(sizeof(zlr_job) * (size_t)nthreads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 0; k < nthreads; k++) {
        zlr_job* j = &jobs[k];
        j->decompress = decompress; j->level = level; j->dict = dict; j->dict_len = dict_len;
        j->in = in; j->in_off = in_off; j->in_size = in_size; j->out = out; j->out_off = out_off; j->out_cap = out_cap; j->out_size = out_size;
        j->first = n * (size_t)k / (size_t)nthreads; j->last = n * (size_t)(k + 1) / (size_t)nthreads;
        pthread_create(&th[k], NULL, zlr_work, j);
    }
    for (int k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th); free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
