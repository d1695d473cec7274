// CPU model of the DEEP match finder planned for levels >= 7 (development only): every position takes the best of the K most
// recent earlier positions with the same hash inside the window (what a sort of the section's positions by hash hands a GPU
// thread: its K predecessors in the sorted order), then the match kernel's window parse (256 positions at once: repeat
// offset of the window in front, lazy rule, greedy path, backward extension, merged continuation pieces).  The sequences go
// through the REFERENCE's entropy stage (ZSTD_compressSequences of oracle/_ref) so the size printed is a real frame's.
// build: gcc -O2 -o /tmp/deep_sim scripts/dev/sim/deep_sim.c oracle/_ref/libzstd_ref.so -Wl,-rpath,$PWD/oracle/_ref
// run:   /tmp/deep_sim file [key=value ...]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
typedef struct { unsigned offset, litLength, matchLength, rep; } ZSTD_Sequence;
extern void* ZSTD_createCCtx(void);
extern size_t ZSTD_CCtx_setParameter(void*, int, int);
extern size_t ZSTD_compressSequences(void*, void*, size_t, const ZSTD_Sequence*, size_t, const void*, size_t);
extern size_t ZSTD_compressBound(size_t);
extern unsigned ZSTD_isError(size_t);
extern const char* ZSTD_getErrorName(size_t);
extern size_t ZSTD_compress2(void*, void*, size_t, const void*, size_t);

static int K = 16, WLOG = 21, MML = 5, CAP = 72, LAZY = 2, CHUNK = 131072, WIN = 256, BACK = 8, HLOG = 24, LEVEL = 9, FRAMES = 0, RULE = 0, SECLOG = 26, REPIN = 0, MINLEN = 5, FARMIN = 0, NOREP = 0;
static u64 ld64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
static u32 hb(u32 v) { return 31 - __builtin_clz(v); }
static u32 mlen(const u8* s, long p, long q, long n, u32 cap) { u32 l = 0; while (l < cap && p + l < n && s[p + l] == s[q + l]) l++; return l; }
static int gain(u32 len, u32 off) { return 4 * (int)len - (int)hb(off + 1); }

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long N = ftell(f); fseek(f, 0, SEEK_SET);
    u8* s = malloc(N + 64); if (fread(s, 1, N, f) != (size_t)N) return 1; fclose(f); memset(s + N, 0, 64);
    for (int i = 2; i < argc; i++) {
        char* e = strchr(argv[i], '='); if (!e) continue; *e = 0; int v = atoi(e + 1);
#define P(n_, v_) else if (!strcmp(argv[i], n_)) v_ = v
        if (0) {} P("k", K); P("wlog", WLOG); P("mml", MML); P("cap", CAP); P("lazy", LAZY); P("back", BACK); P("hlog", HLOG); P("level", LEVEL);
        P("frames", FRAMES); P("rule", RULE); P("repin", REPIN); P("minlen", MINLEN); P("farmin", FARMIN); P("win", WIN); P("norep", NOREP);
    }
    // ---- deep candidates: best[p] over the K most recent same-hash positions within the window (and the chunk, in the frames layout; and the section)
    u32* blen = calloc(N + 1, 4); u32* boff = calloc(N + 1, 4);
    {
        int32_t* head = malloc(sizeof(int32_t) << HLOG); int32_t* prev = malloc(sizeof(int32_t) * (N + 1));
        for (long i = 0; i < (1L << HLOG); i++) head[i] = -1;
        const u64 W = 1ull << WLOG;
        for (long p = 0; p + 8 <= N; p++) {
            const u64 v = ld64(s + p);
            const u32 h = (u32)(((v << (64 - 8 * MML)) * 0x9E3779B185EBCA87ULL) >> (64 - HLOG));
            const long lo_sec = (p >> SECLOG) << SECLOG, lo_chunk = FRAMES ? (p / CHUNK) * (long)CHUNK : 0;
            long lo = p > (long)W ? p - (long)W : 0; if (lo < lo_sec) lo = lo_sec; if (lo < lo_chunk) lo = lo_chunk;
            const long nend = FRAMES ? ((p / CHUNK) + 1) * (long)CHUNK < N ? ((p / CHUNK) + 1) * (long)CHUNK : N : N;
            u32 bl = 0, bo = 0; int bg = -1000;
            long q = head[h];
            for (int k = 0; k < K && q >= lo; k++, q = prev[q]) {
                u32 l = mlen(s, p, q, nend, CAP);
                if (l < (u32)MINLEN) continue;
                const u32 off = (u32)(p - q);
                if (FARMIN && off > 65536 && l < (u32)FARMIN) continue;
                if (RULE == 0) { if (l > bl) { bl = l; bo = off; } }                         // longest, nearest among equals
                else { int g = gain(l, off); if (g > bg) { bg = g; bl = l; bo = off; } }      // best gain
            }
            blen[p] = bl; boff[p] = bo;
            prev[p] = head[h]; head[h] = (int32_t)p;
        }
        free(head); free(prev);
    }
    // ---- the match kernel's parse, chunk by chunk, window by window
    ZSTD_Sequence* seqs = malloc(sizeof(ZSTD_Sequence) * (N / 3 + 16)); size_t nseq = 0;
    u32* wl = malloc(4 * (WIN + 2)); u32* wo = malloc(4 * (WIN + 2));
    long anchor = 0; u64 nrep = 0, ndeep = 0;
    long prev_end = -1; u32 prev_off = 0;
    for (long c0 = 0; c0 < N; c0 += CHUNK) {
        const long n = (N - c0 < CHUNK) ? N - c0 : CHUNK;
        const long lo_ok = FRAMES ? c0 : (c0 >> SECLOG) << SECLOG;   // lowest source position a match of this chunk may name
        u32 rep = 0; long carry = 0;
        if (FRAMES) { prev_end = -1; }
        for (long ip = 0; ip < n; ip += WIN) {
            const long wend = ip + WIN < n ? ip + WIN : n;
            for (long p = ip; p < wend; p++) {
                const long g = c0 + p; u32 bl = 0, bo = 0;
                if (p + 8 <= n) {
                    bl = blen[g]; bo = boff[g];
                    if (bl && g + bl > c0 + n) bl = (u32)(c0 + n - g);
                    if (bl < 4) { bl = 0; bo = 0; }
                    if (!NOREP && rep && g - (long)rep >= lo_ok) {
                        const u32 lr = mlen(s, g, g - rep, c0 + n, CAP);
                        // the kernel's rule: a table candidate must be LONGER than the repeat offset's
                        if (lr >= 4 && (RULE == 0 ? lr >= bl : gain(lr, 0) + 4 >= gain(bl, bo))) { bl = lr; bo = rep; }
                    }
                }
                wl[p - ip] = bl; wo[p - ip] = bo;
            }
            wl[wend - ip] = wl[wend - ip + 1] = 0;
            long p = ip + carry; u32 last_off = 0;
            while (p < wend) {
                const long w = p - ip; u32 bl = wl[w], bo = wo[w];
                if (bl && LAZY >= 1 && wl[w + 1] && gain(wl[w + 1], wo[w + 1]) > gain(bl, bo) + 4) bl = 0;
                if (bl && LAZY >= 2 && w + 2 < WIN && wl[w + 2] && gain(wl[w + 2], wo[w + 2]) > gain(bl, bo) + 7) bl = 0;
                if (!bl) { p++; continue; }
                long ps = c0 + p, g = c0 + p; u32 ml = bl;
                int b = 0;
                while (b < BACK && ps > anchor && ps - 1 - (long)bo >= lo_ok && s[ps - 1] == s[ps - 1 - bo]) { ps--; ml++; b++; }
                if (ps == prev_end && bo == prev_off && nseq && ps == anchor) { seqs[nseq - 1].matchLength += ml; }   // a continuation piece
                else { seqs[nseq].offset = bo; seqs[nseq].litLength = (u32)(ps - anchor); seqs[nseq].matchLength = ml; seqs[nseq].rep = 0; nseq++; if (bo == rep) nrep++; else ndeep++; }
                p += bl; anchor = c0 + p; prev_end = anchor; prev_off = bo; last_off = bo; (void)g;
                if (REPIN) rep = bo;
            }
            if (last_off) rep = last_off;
            carry = p - wend;
        }
        if (anchor > c0 + n) { fprintf(stderr, "parse ran past the chunk\n"); return 2; }
        if (FRAMES) { /* literals up to the chunk's end are closed by the next sequence's litLength: fine in one frame of sequences */ }
    }
    // ---- the reference's entropy stage
    void* cctx = ZSTD_createCCtx();
    ZSTD_CCtx_setParameter(cctx, 100, LEVEL);
    ZSTD_CCtx_setParameter(cctx, 101, WLOG < 10 ? 10 : WLOG);
    size_t cap = ZSTD_compressBound(N); u8* dst = malloc(cap);
    size_t r = ZSTD_compressSequences(cctx, dst, cap, seqs, nseq, s, N);
    if (ZSTD_isError(r)) { fprintf(stderr, "compressSequences: %s\n", ZSTD_getErrorName(r)); return 3; }
    void* c2 = ZSTD_createCCtx(); ZSTD_CCtx_setParameter(c2, 100, LEVEL);
    size_t rr = ZSTD_compress2(c2, dst, cap, s, N);
    u64 lits = 0; for (size_t i = 0; i < nseq; i++) lits += seqs[i].litLength;
    printf("k=%d wlog=%d mml=%d cap=%d lazy=%d back=%d rule=%d frames=%d: seqs %zu (rep %llu) lits %llu  size %zu ratio %.3f | ref level %d: %zu ratio %.3f | ours/ref %.3f\n",
           K, WLOG, MML, CAP, LAZY, BACK, RULE, FRAMES, nseq, (unsigned long long)nrep, (unsigned long long)lits, r, (double)N / r, LEVEL, rr, (double)N / rr, (double)r / rr);
    return 0;
}
