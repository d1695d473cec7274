// CPU model, round 6: a price-based forward parse over the deep match finder's candidates, with prices taken from the STATISTICS of a
// parse of the same data (zstd.c:34586-36163: btopt / btultra / btultra2 price a choice by what its codes cost under the running
// symbol statistics; btultra2 parses a block twice to seed them) -- what levels 16 ... 22 could be on a machine that parses a chunk in
// passes instead of adapting symbol by symbol.  The sequences go through the REFERENCE's entropy stage (ZSTD_compressSequences), so the
// number printed is the parse's doing alone.  opt_sim.c (round 5) had fixed prices and no literal-length state: 1.07 - 1.10x the reference.
//
//   gcc -O2 -o opt2_sim opt2_sim.c oracle/_ref/libzstd_ref.so -lm -Wl,-rpath,$PWD/oracle/_ref
//   ./opt2_sim file level=19 k=256 topc=8 wlog=23 mml=3 passes=2 chunk=0
// chunk=N (bytes): statistics per chunk of N bytes instead of over the whole input (what a kernel that owns a chunk can have).
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
static int K = 64, WLOG = 22, MML = 4, CAP = 255, HLOG = 24, LEVEL = 16, TOPC = 8, PASSES = 2, CHUNK = 0, REPML = 3, SUFF = 0, LAZYSEED = 1, MML2 = 0, K2 = 64;
static u64 ld64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
static u32 hb(u32 v) { return 31 - __builtin_clz(v); }
static u32 mlen(const u8* s, long p, long q, long n, u32 cap) { u32 l = 0; while (l < cap && p + l < n && s[p + l] == s[q + l]) l++; return l; }
static u32 ll_code(u32 l) { static const u8 t[64] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24}; return l > 63 ? hb(l) + 19 : t[l]; }
static u32 ml_code(u32 mlb) { static const u8 t[128] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42}; return mlb > 127 ? hb(mlb) + 36 : t[mlb]; }
static const u8 ll_bits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const u8 ml_bits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
typedef struct { double lit[256], ll[36], ml[53], of[32]; } Stats;
typedef struct { float lit[256], ll[36], ml[53], of[32]; } Prices;
static void prices_from(const Stats* st, Prices* P) {
    double sl = 0, sll = 0, sml = 0, sof = 0;
    for (int i = 0; i < 256; i++) sl += st->lit[i] + 0.25; for (int i = 0; i < 36; i++) sll += st->ll[i] + 0.25; for (int i = 0; i < 53; i++) sml += st->ml[i] + 0.25; for (int i = 0; i < 32; i++) sof += st->of[i] + 0.25;
    for (int i = 0; i < 256; i++) P->lit[i] = (float)-log2((st->lit[i] + 0.25) / sl);
    for (int i = 0; i < 36; i++) P->ll[i] = (float)-log2((st->ll[i] + 0.25) / sll) + ll_bits[i];
    for (int i = 0; i < 53; i++) P->ml[i] = (float)-log2((st->ml[i] + 0.25) / sml) + ml_bits[i];
    for (int i = 0; i < 32; i++) P->of[i] = (float)-log2((st->of[i] + 0.25) / sof) + i;
}
typedef struct { float price; u32 len, off, litlen; u32 rep[3]; } Node;  // arrival at a position: by a match (len, off) or a literal (len 0); litlen: literals pending
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long N = ftell(f); fseek(f, 0, SEEK_SET);
    u8* s = malloc(N + 64); if (fread(s, 1, N, f) != (size_t)N) return 1; fclose(f); memset(s + N, 0, 64);
    for (int i = 2; i < argc; i++) { char* e = strchr(argv[i], '='); if (!e) continue; *e = 0; int v = atoi(e + 1);
#define P_(n_, v_) else if (!strcmp(argv[i], n_)) v_ = v
        if (0) {} P_("k", K); P_("wlog", WLOG); P_("mml", MML); P_("cap", CAP); P_("hlog", HLOG); P_("level", LEVEL); P_("topc", TOPC); P_("passes", PASSES); P_("chunk", CHUNK); P_("repml", REPML); P_("suff", SUFF); P_("lazyseed", LAZYSEED); P_("mml2", MML2); P_("k2", K2); }
    if (CHUNK == 0) CHUNK = (int)N;
    // candidates: per position up to TOPC (len, off), strictly increasing len, nearest first among the K most recent with the same MML bytes
    u32* cl = calloc((size_t)N * TOPC, 4); u32* co = calloc((size_t)N * TOPC, 4); u8* cn = calloc(N, 1);
    // (MML2 > 0: a second chain over MML2-byte hashes finds the LONG matches that the K most recent MML-byte twins do not reach -- what a
    // binary tree over the whole window finds; the two increasing chains are merged: per length the nearest offset)
    for (int which = 0; which < (MML2 ? 2 : 1); which++) {
      const int mm = which ? MML2 : MML;
      int32_t* head = malloc(sizeof(int32_t) << HLOG); int32_t* prev = malloc(sizeof(int32_t) * (N + 1));
      for (long i = 0; i < (1L << HLOG); i++) head[i] = -1;
      const long W = 1L << WLOG;
      for (long p = 0; p + 8 <= N; p++) {
        const u64 v = ld64(s + p); const u32 h = (u32)(((mm >= 8 ? v : (v << (64 - 8 * mm))) * 0x9E3779B185EBCA87ULL) >> (64 - HLOG));
        long lo = p > W ? p - W : 0; long q = head[h]; u32 best = (which ? MML2 : MML) - 1;
        u32 tl[512], to[512]; int tn = 0;
        // what the first chain left for this position comes first
        if (which) { for (int k = 0; k < cn[p]; k++) { tl[tn] = cl[(size_t)p * TOPC + k]; to[tn] = co[(size_t)p * TOPC + k]; tn++; } }
        int base_n = tn;
        for (int k = 0; k < (which ? K2 : K) && q >= lo; k++, q = prev[q]) { u32 l = mlen(s, p, q, N, CAP); if (l > best) { best = l; if (tn < 512) { tl[tn] = l; to[tn] = (u32)(p - q); tn++; } } }
        if (which && tn > base_n) {  // merge: sort by length, keep per length the nearest, strictly increasing length with decreasing... (pareto: longer must be kept; equal length: nearer)
            for (int a = 1; a < tn; a++) { u32 L = tl[a], O = to[a]; int b = a - 1; while (b >= 0 && (tl[b] > L || (tl[b] == L && to[b] > O))) { tl[b + 1] = tl[b]; to[b + 1] = to[b]; b--; } tl[b + 1] = L; to[b + 1] = O; }
            // drop entries dominated by a longer-or-equal one with a smaller-or-equal offset
            int m = 0; for (int a = 0; a < tn; a++) { int dom = 0; for (int b = a + 1; b < tn; b++) if (to[b] <= to[a]) { dom = 1; break; } if (a + 1 < tn && tl[a + 1] == tl[a]) dom = 1; if (!dom) { tl[m] = tl[a]; to[m] = to[a]; m++; } } tn = m; }
        int n = 0; int st = tn > TOPC ? tn - TOPC : 0; for (int k = st; k < tn; k++) { cl[(size_t)p * TOPC + n] = tl[k]; co[(size_t)p * TOPC + n] = to[k]; n++; }
        cn[p] = (u8)n; prev[p] = head[h]; head[h] = (int32_t)p; }
      free(head); free(prev); }
    ZSTD_Sequence* seqs = malloc(sizeof(ZSTD_Sequence) * (N / 3 + 16)); size_t ns = 0;
    Node* nd = malloc(sizeof(Node) * (size_t)(CHUNK + CAP + 2));
    void* cctx = ZSTD_createCCtx(); ZSTD_CCtx_setParameter(cctx, 100, LEVEL); ZSTD_CCtx_setParameter(cctx, 101, WLOG);
    size_t capb = ZSTD_compressBound(N); u8* dst = malloc(capb);
    u32 rep_in[3] = {1, 4, 8};
    for (long c0 = 0; c0 < N; c0 += CHUNK) {
        const long c1 = c0 + CHUNK < N ? c0 + CHUNK : N, n = c1 - c0;
        Stats st; Prices P;
        // seed statistics: a greedy parse (longest candidate, repeat offset first) of this chunk
        { memset(&st, 0, sizeof st); u32 rep = rep_in[0]; long p = c0, anchor = c0;
          while (p < c1) { u32 L = 0, O = 0; if (rep <= p) { u32 l = mlen(s, p, p - rep, c1, CAP); if (l >= 3) { L = l; O = rep; } }
              if (cn[p]) { u32 l = cl[(size_t)p * TOPC + cn[p] - 1], o = co[(size_t)p * TOPC + cn[p] - 1]; if (l > c1 - p) l = c1 - p; if (l >= (u32)MML && l > L + 1) { L = l; O = o; } }
              if (!L) { p++; continue; }
              for (long i = anchor; i < p; i++) st.lit[s[i]]++; st.ll[ll_code(p - anchor)]++; st.ml[ml_code(L - 3)]++; st.of[O == rep ? 0 : hb(O + 3)]++; rep = O; p += L; anchor = p; }
          for (long i = anchor; i < c1; i++) st.lit[s[i]]++; }
        size_t ns0 = ns;
        for (int pass = 0; pass < PASSES; pass++) {
            prices_from(&st, &P);
            ns = ns0;
            for (long i = 0; i <= n + CAP; i++) nd[i].price = 1e30f;
            nd[0].price = P.ll[0]; nd[0].len = 0; nd[0].off = 0; nd[0].litlen = 0; memcpy(nd[0].rep, rep_in, 12);
            for (long i = 0; i < n; i++) {
                const Node cur = nd[i]; const long p = c0 + i;
                if (cur.price >= 1e29f) continue;
                // literal
                { const u32 ll = cur.litlen + 1; const float c = cur.price + P.lit[s[p]] + P.ll[ll_code(ll)] - P.ll[ll_code(cur.litlen)];
                  if (c < nd[i + 1].price) { nd[i + 1] = cur; nd[i + 1].price = c; nd[i + 1].len = 0; nd[i + 1].litlen = ll; } }
                const float base = cur.price + P.ll[0];  // (the run of literals in front is paid for; the next run starts at length 0)
                // repeat offsets (zstd.c:19680: with no literal in front the codes mean rep[1], rep[2], rep[0] - 1)
                const int ll0 = cur.litlen == 0;
                for (int r = 0; r < 3; r++) {
                    u32 o; if (!ll0) o = cur.rep[r]; else o = r == 0 ? cur.rep[1] : r == 1 ? cur.rep[2] : cur.rep[0] - 1;
                    if (!o || (long)o > p) continue;
                    if (ll0 && r < 2 && o == cur.rep[0]) continue;
                    u32 L = mlen(s, p, p - o, c1, CAP); if (L < (u32)REPML) continue;
                    const float pc = base + P.of[r == 0 ? 0 : 1] + (r == 2 ? 1.0f : 0.0f);
                    for (u32 l = REPML; l <= L; l++) { const float c = pc + P.ml[ml_code(l - 3)];
                        if (c < nd[i + l].price) { Node* t = &nd[i + l]; t->price = c; t->len = l; t->off = o; t->litlen = 0;
                            // history after a repeat code
                            u32 h0 = cur.rep[0], h1 = cur.rep[1], h2 = cur.rep[2];
                            if (o == h0) { t->rep[0] = h0; t->rep[1] = h1; t->rep[2] = h2; } else if (o == h1) { t->rep[0] = h1; t->rep[1] = h0; t->rep[2] = h2; } else { t->rep[0] = o; t->rep[1] = h0; t->rep[2] = h1; } } }
                }
                u32 lmin = MML;
                for (int k = 0; k < cn[p]; k++) { u32 L = cl[(size_t)p * TOPC + k]; const u32 o = co[(size_t)p * TOPC + k]; if (L > c1 - p) L = (u32)(c1 - p);
                    if (o == cur.rep[0] && !ll0) { lmin = L + 1; continue; }
                    const float pc = base + P.of[hb(o + 3)];
                    for (u32 l = lmin; l <= L; l++) { const float c = pc + P.ml[ml_code(l - 3)];
                        if (c < nd[i + l].price) { Node* t = &nd[i + l]; t->price = c; t->len = l; t->off = o; t->litlen = 0; t->rep[0] = o; t->rep[1] = cur.rep[0]; t->rep[2] = cur.rep[1]; } }
                    if (L + 1 > lmin) lmin = L + 1; }
            }
            // backtrack
            { long i = n; size_t nr = 0; ZSTD_Sequence* rev = seqs + (N / 3 + 16) - 1;  // (filled from the back of the array's free part)
              u32 lits = 0; Stats s2; memset(&s2, 0, sizeof s2);
              ZSTD_Sequence* tmp = malloc(sizeof(ZSTD_Sequence) * (n / 3 + 16));
              while (i > 0) { if (nd[i].len == 0) { s2.lit[s[c0 + i - 1]]++; lits++; i--; }
                  else { if (nr) { tmp[nr - 1].litLength = lits; s2.ll[ll_code(lits)]++; } lits = 0; tmp[nr].matchLength = nd[i].len; tmp[nr].offset = nd[i].off; tmp[nr].rep = 0; tmp[nr].litLength = 0; s2.ml[ml_code(nd[i].len - 3)]++;
                         // offset code statistics: a repeat if it equals an entry of the history at the match's start
                         { const Node* fr = &nd[i - nd[i].len]; u32 o = nd[i].off; int code = hb(o + 3);
                           if (fr->litlen) { if (o == fr->rep[0]) code = 0; else if (o == fr->rep[1] || o == fr->rep[2]) code = 1; }
                           else { if (o == fr->rep[1]) code = 0; else if (o == fr->rep[2] || o == fr->rep[0] - 1) code = 1; }
                           s2.of[code]++; }
                         nr++; i -= nd[i].len; } }
              if (nr) { tmp[nr - 1].litLength = lits; s2.ll[ll_code(lits)]++; }
              u32 trailing = nr ? 0 : lits; (void)rev;
              // trailing literals of the chunk: carried into the next chunk's first sequence (block delimiters are not used)
              static u32 carry = 0;
              for (size_t q = 0; q < nr; q++) { seqs[ns] = tmp[nr - 1 - q]; if (q == 0) { seqs[ns].litLength += carry; } ns++; }
              if (pass == PASSES - 1) { if (nr) { u32 tl = 0; long e = n; while (e > 0 && nd[e].len == 0) { tl++; e--; } carry = tl; } else carry += trailing; }
              free(tmp);
              if (getenv("OPT_DBG") && pass == PASSES - 1 && c0 == 0) { double nl = 0, nll = 0, nml = 0, nof = 0, rep0 = s2.of[0], rep12 = s2.of[1]; for (int q = 0; q < 256; q++) nl += s2.lit[q]; for (int q = 0; q < 36; q++) nll += s2.ll[q]; for (int q = 0; q < 53; q++) nml += s2.ml[q]; for (int q = 0; q < 32; q++) nof += s2.of[q];
                  double mlsum = 0; for (size_t q = ns0; q < ns; q++) mlsum += seqs[q].matchLength;
                  fprintf(stderr, "chunk 0: est price %.0f bits (%.0f bytes), lits %.0f seqs %.0f avg ml %.2f rep0 %.0f rep12 %.0f\n", nd[n].price, nd[n].price / 8, nl, nof, mlsum / (nof + 1e-9), rep0, rep12);
                  for (int q = 0; q < 32; q++) if (s2.of[q]) fprintf(stderr, " of[%d]=%.0f(%.1f)", q, s2.of[q], P.of[q]); fprintf(stderr, "\n");
                  for (int q = 0; q < 12; q++) fprintf(stderr, " ml[%d]=%.0f(%.1f)", q, s2.ml[q], P.ml[q]); fprintf(stderr, "\n");
                  for (int q = 0; q < 12; q++) fprintf(stderr, " ll[%d]=%.0f(%.1f)", q, s2.ll[q], P.ll[q]); fprintf(stderr, "\n"); }
              st = s2; }
        }
        memcpy(rep_in, nd[n].rep, 12);
    }
    size_t r = ZSTD_compressSequences(cctx, dst, capb, seqs, ns, s, N);
    if (ZSTD_isError(r)) { fprintf(stderr, "compressSequences: %s\n", ZSTD_getErrorName(r)); return 3; }
    void* c2 = ZSTD_createCCtx(); ZSTD_CCtx_setParameter(c2, 100, LEVEL); size_t rr = ZSTD_compress2(c2, dst, capb, s, N);
    printf("opt2 k=%d topc=%d wlog=%d mml=%d passes=%d chunk=%d: seqs %zu size %zu ratio %.3f | ref level %d ratio %.3f | ours/ref %.3f\n", K, TOPC, WLOG, MML, PASSES, CHUNK, ns, r, (double)N / r, LEVEL, (double)N / rr, (double)r / rr);
    return 0;
}
