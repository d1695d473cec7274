// CPU model: an optimal (price-based, forward DP) parse over the deep match finder's candidates -- what levels 16+ could gain.
// Per position: up to K candidates (the K most recent same-hash positions in the window), each (len, off).  DP over the whole input
// (blocks ignored), static prices from a first greedy pass; repeat offsets tracked along the best path (as zstd's btopt does).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
typedef struct { unsigned offset, litLength, matchLength, rep; } ZSTD_Sequence;
extern void* ZSTD_createCCtx(void);
extern size_t ZSTD_CCtx_setParameter(void*, int, int);
extern size_t ZSTD_compressSequences(void*, void*, size_t, const ZSTD_Sequence*, size_t, const void*, size_t);
extern size_t ZSTD_compressBound(size_t);
extern unsigned ZSTD_isError(size_t);
extern const char* ZSTD_getErrorName(size_t);
extern size_t ZSTD_compress2(void*, void*, size_t, const void*, size_t);
static int K = 32, WLOG = 22, MML = 4, CAP = 255, HLOG = 24, LEVEL = 16, MINLEN = 4, TOPC = 4, ALLLEN = 1;
static u64 ld64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
static u32 hb(u32 v) { return 31 - __builtin_clz(v); }
static u32 mlen(const u8* s, long p, long q, long n, u32 cap) { u32 l = 0; while (l < cap && p + l < n && s[p + l] == s[q + l]) l++; return l; }
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long N = ftell(f); fseek(f, 0, SEEK_SET);
    u8* s = malloc(N + 64); if (fread(s, 1, N, f) != (size_t)N) return 1; fclose(f); memset(s + N, 0, 64);
    for (int i = 2; i < argc; i++) { char* e = strchr(argv[i], '='); if (!e) continue; *e = 0; int v = atoi(e + 1);
#define P(n_, v_) else if (!strcmp(argv[i], n_)) v_ = v
        if (0) {} P("k", K); P("wlog", WLOG); P("mml", MML); P("cap", CAP); P("hlog", HLOG); P("level", LEVEL); P("minlen", MINLEN); P("topc", TOPC); P("alllen", ALLLEN); }
    // literal prices: order-0 over the whole input (bits x 256 fixed point as floats)
    double lp[256]; { u64 h[256] = {0}; for (long i = 0; i < N; i++) h[s[i]]++; for (int b = 0; b < 256; b++) lp[b] = h[b] ? -log2((double)h[b] / N) : 12.0; }
    // candidates: per position up to TOPC (len, off) with strictly increasing len (nearest first among the K most recent)
    u32* cl = calloc((size_t)N * TOPC, 4); u32* co = calloc((size_t)N * TOPC, 4); u8* cn = calloc(N, 1);
    { int32_t* head = malloc(sizeof(int32_t) << HLOG); int32_t* prev = malloc(sizeof(int32_t) * (N + 1));
      for (long i = 0; i < (1L << HLOG); i++) head[i] = -1;
      const long W = 1L << WLOG;
      for (long p = 0; p + 8 <= N; p++) {
        const u64 v = ld64(s + p); const u32 h = (u32)(((v << (64 - 8 * MML)) * 0x9E3779B185EBCA87ULL) >> (64 - HLOG));
        long lo = p > W ? p - W : 0; long q = head[h]; u32 best = MINLEN - 1; int n = 0;
        u32 tl[64], to[64]; int tn = 0;
        for (int k = 0; k < K && q >= lo; k++, q = prev[q]) { u32 l = mlen(s, p, q, N, CAP); if (l > best) { best = l; if (tn < 64) { tl[tn] = l; to[tn] = (u32)(p - q); tn++; } } }
        // keep the TOPC longest of the increasing chain (the last ones)
        int st = tn > TOPC ? tn - TOPC : 0; for (int k = st; k < tn; k++) { cl[(size_t)p * TOPC + n] = tl[k]; co[(size_t)p * TOPC + n] = to[k]; n++; }
        cn[p] = (u8)n; prev[p] = head[h]; head[h] = (int32_t)p; }
      free(head); free(prev); }
    // forward DP
    float* cost = malloc(sizeof(float) * (N + 2)); u32* fl = malloc(4 * (N + 2)); u32* fo = malloc(4 * (N + 2));   // best arrival: length (0 = literal) and offset
    u32 (*rep)[3] = malloc(sizeof(u32[3]) * (N + 2));
    for (long i = 0; i <= N; i++) cost[i] = 1e30f;
    cost[0] = 0; rep[0][0] = 1; rep[0][1] = 4; rep[0][2] = 8; fl[0] = 0;
    for (long p = 0; p < N; p++) {
        const float c0 = cost[p];
        // literal
        { float c = c0 + (float)lp[s[p]]; if (c < cost[p + 1]) { cost[p + 1] = c; fl[p + 1] = 0; fo[p + 1] = 0; memcpy(rep[p + 1], rep[p], 12); } }
        // repeat offsets as candidates too
        for (int r = 0; r < 3; r++) { u32 o = rep[p][r]; if (o && (long)o <= p) { u32 L = mlen(s, p, p - o, N, CAP); if (L >= 3) {
            for (u32 l = ALLLEN ? 3 : L; l <= L; l++) { float c = c0 + 4.5f + (r ? 1.5f : 0.0f) + 4.0f + (l > 34 ? 2.0f * hb(l) - 8 : 0) + 3.5f; if (c < cost[p + l]) { cost[p + l] = c; fl[p + l] = l; fo[p + l] = o; u32 nr[3]; if (r == 0) memcpy(nr, rep[p], 12); else if (r == 1) { nr[0] = rep[p][1]; nr[1] = rep[p][0]; nr[2] = rep[p][2]; } else { nr[0] = rep[p][2]; nr[1] = rep[p][0]; nr[2] = rep[p][1]; } memcpy(rep[p + l], nr, 12); } } } } }
        for (int k = 0; k < cn[p]; k++) { const u32 L = cl[(size_t)p * TOPC + k], o = co[(size_t)p * TOPC + k]; const u32 lmin = k ? cl[(size_t)p * TOPC + k - 1] + 1 : MINLEN;
            for (u32 l = ALLLEN ? lmin : L; l <= L; l++) { float c = c0 + (float)(hb(o + 3)) + 6.0f + 4.0f + (l > 34 ? 2.0f * hb(l) - 8 : 0) + 3.5f; if (c < cost[p + l]) { cost[p + l] = c; fl[p + l] = l; fo[p + l] = o; rep[p + l][0] = o; rep[p + l][1] = rep[p][0]; rep[p + l][2] = rep[p][1]; } } }
    }
    // backtrack
    ZSTD_Sequence* seqs = malloc(sizeof(ZSTD_Sequence) * (N / 3 + 16)); size_t ns = 0;
    { long p = N; size_t cap = N / 3 + 16; ZSTD_Sequence* rev = malloc(sizeof(ZSTD_Sequence) * cap); size_t nr = 0; u32 lits = 0;
      while (p > 0) { if (fl[p] == 0) { lits++; p--; } else { rev[nr].matchLength = fl[p]; rev[nr].offset = fo[p]; rev[nr].litLength = 0; rev[nr].rep = 0; if (nr) rev[nr - 1].litLength = lits; else {/* trailing literals */} lits = 0; nr++; p -= fl[p]; } }
      if (nr) rev[nr - 1].litLength = lits;
      for (size_t i = 0; i < nr; i++) seqs[ns++] = rev[nr - 1 - i]; free(rev); }
    void* cctx = ZSTD_createCCtx(); ZSTD_CCtx_setParameter(cctx, 100, LEVEL); ZSTD_CCtx_setParameter(cctx, 101, WLOG);
    size_t cap = ZSTD_compressBound(N); u8* dst = malloc(cap);
    size_t r = ZSTD_compressSequences(cctx, dst, cap, seqs, ns, s, N);
    if (ZSTD_isError(r)) { fprintf(stderr, "compressSequences: %s\n", ZSTD_getErrorName(r)); return 3; }
    void* c2 = ZSTD_createCCtx(); ZSTD_CCtx_setParameter(c2, 100, LEVEL); size_t rr = ZSTD_compress2(c2, dst, cap, s, N);
    printf("opt k=%d topc=%d wlog=%d mml=%d alllen=%d: seqs %zu size %zu ratio %.3f | ref level %d ratio %.3f | ours/ref %.3f\n", K, TOPC, WLOG, MML, ALLLEN, ns, r, (double)N / r, LEVEL, (double)N / rr, (double)r / rr);
    return 0;
}
