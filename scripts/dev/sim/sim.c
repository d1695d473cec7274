// CPU model of the HIP match kernel's dense body (development only): counts literals / sequences under
// variations of the parse, to see where ratio is lost against the reference.  Cost model: bits = BL*nlit + BS*nseq.
// build: gcc -O2 -o /tmp/sim scripts/dev/sim/sim.c ; run: /tmp/sim file [key=value ...]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
static int LONG_LOG = 13, SHORT_LOG = 12, SHORT_BYTES = 6, CAP = 40, WIN = 256, INTRA = 0, BACK = 0, LAZY = 1, CHUNK = 131072, REP23 = 0, FARLOG = 0, INSERT_ALL = 1;
static double BL = 5.49, BS = 20.56;
static int REACH = 65535; static int FAR = 0, FSAMP = 5, FBUCK = 13, FGRP = 16, FMIN = 8, FHINT = 5; static int MERGE = 0; static int IW = 512, IB = 5, WAVE = 64, BACKMAX = 1000, INSFILT = 0, IMIN = 4;
static u64 ld64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
static u32 h8(u64 v) { return (u32)((v * 0x9E3779B185EBCA87ULL) >> (64 - LONG_LOG)); }
static u32 hs(u64 v) { return (u32)(((v << (64 - 8 * SHORT_BYTES)) * 0xC2B2AE3D27D4EB4FULL) >> (64 - SHORT_LOG)); }
static u32 cb(u64 a, u64 b) { u64 x = a ^ b; return x ? (u32)(__builtin_ctzll(x) >> 3) : 8; }
static u32 mlen(const u8* s, u32 p, long q, u32 n, u32 cap) {  // common length of s[p..] and s[q..], p + len <= n
    u32 l = 0;
    while (l < cap && p + l < n && s[p + l] == s[q + l]) l++;
    return l;
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long N = ftell(f); fseek(f, 0, SEEK_SET);
    u8* data = malloc(N + 16); fread(data, 1, N, f); fclose(f); memset(data + N, 0, 16);
    for (int i = 2; i < argc; i++) {
        char* e = strchr(argv[i], '='); if (!e) continue; *e = 0; double v = atof(e + 1);
        if (!strcmp(argv[i], "long")) LONG_LOG = v; else if (!strcmp(argv[i], "short")) SHORT_LOG = v; else if (!strcmp(argv[i], "sb")) SHORT_BYTES = v;
        else if (!strcmp(argv[i], "cap")) CAP = v; else if (!strcmp(argv[i], "win")) WIN = v; else if (!strcmp(argv[i], "intra")) INTRA = v;
        else if (!strcmp(argv[i], "back")) BACK = v; else if (!strcmp(argv[i], "lazy")) LAZY = v; else if (!strcmp(argv[i], "chunk")) CHUNK = v;
        else if (!strcmp(argv[i], "rep23")) REP23 = v; else if (!strcmp(argv[i], "iw")) IW = v; else if (!strcmp(argv[i], "reach")) REACH = v; else if (!strcmp(argv[i], "merge")) MERGE = v; else if (!strcmp(argv[i], "far")) FAR = v; else if (!strcmp(argv[i], "fsamp")) FSAMP = v;
        else if (!strcmp(argv[i], "fbuck")) FBUCK = v; else if (!strcmp(argv[i], "fgrp")) FGRP = v; else if (!strcmp(argv[i], "fmin")) FMIN = v; else if (!strcmp(argv[i], "fhint")) FHINT = v; else if (!strcmp(argv[i], "ib")) IB = v; else if (!strcmp(argv[i], "wave")) WAVE = v;
        else if (!strcmp(argv[i], "backmax")) BACKMAX = v; else if (!strcmp(argv[i], "insfilt")) INSFILT = v; else if (!strcmp(argv[i], "imin")) IMIN = v; else if (!strcmp(argv[i], "insall")) INSERT_ALL = v;
    }
    u64 tot_lit = 0, tot_seq = 0, tot_rep = 0, r1 = 0, r2 = 0, r3 = 0, r1m = 0, offbits = 0; u32 R[3] = {1,4,8};
    int32_t* tl = malloc(sizeof(int32_t) << LONG_LOG); int32_t* ts = malloc(sizeof(int32_t) << SHORT_LOG);
    u32* blen = malloc(4 * WIN); u32* boff = malloc(4 * WIN);
    long nch = (N + CHUNK - 1) / CHUNK;
    int64_t* ft = NULL; u32* hints = NULL;
    if (FAR) {
        ft = malloc(sizeof(int64_t) * nch << FBUCK);
        for (long i = 0; i < (nch << FBUCK); i++) ft[i] = -1;
        for (long p = 0; p + 8 <= N; p++) {
            u64 h = ld64(data + p) * 0x9E3779B185EBCA87ULL;
            if ((h >> (64 - FSAMP)) != 0) continue;
            u32 b = (u32)(h >> (64 - FSAMP - FBUCK)) & ((1u << FBUCK) - 1);
            ft[((p / CHUNK) << FBUCK) + b] = p;  // max position wins
        }
        hints = calloc((N >> FHINT) + 2, 4);
        u8* hl = calloc((N >> FHINT) + 2, 1);
        for (long p = 0; p + 8 <= N; p++) {
            u64 v = ld64(data + p), h = v * 0x9E3779B185EBCA87ULL;
            if ((h >> (64 - FSAMP)) != 0) continue;
            u32 b = (u32)(h >> (64 - FSAMP - FBUCK)) & ((1u << FBUCK) - 1);
            long c = p / CHUNK;
            for (long k = 1; k < FGRP && c - k >= 0; k++) {
                int64_t q = ft[((c - k) << FBUCK) + b];
                if (q < 0 || ld64(data + q) != v) continue;
                u32 l = 8; while (l < 63 && p + l < N && data[p + l] == data[q + l]) l++;
                if (l > hl[p >> FHINT]) { hl[p >> FHINT] = l; hints[p >> FHINT] = (u32)(p - q); }
                break;
            }
        }
        free(hl);
    }
    for (long c0 = 0; c0 < N; c0 += CHUNK) {
        const u8* s = data + c0; u32 n = (u32)((N - c0 < CHUNK) ? N - c0 : CHUNK);
        for (u32 i = 0; i < (1u << LONG_LOG); i++) tl[i] = -1;
        for (u32 i = 0; i < (1u << SHORT_LOG); i++) ts[i] = -1;
        u32 rep = 0, rep2 = 0, carry = 0, anchor = 0, prev_off_m = 0, prev_end_m = 0; if (!FAR) { R[0] = 1; R[1] = 4; R[2] = 8; }
        int32_t* ti = malloc(4 * IW); u8* inmatch = calloc(WIN, 1);
        for (u32 ip = 0; ip < n; ip += WIN) {
            u32 wend = ip + WIN < n ? ip + WIN : n;
            // ---- find: every position against the table state at window start (INTRA: tables updated as we go)
            for (u32 i = 0; i < (u32)IW; i++) ti[i] = -1;
            for (u32 p = ip; p < wend; p++) {
                u32 bl = 0, bo = 0;
                if (INTRA == 2 && ((p - ip) % WAVE) == 0 && p > ip)  // the previous wave's positions become visible
                    for (u32 q = p - WAVE; q < p && q + 8 <= n; q++) { u64 v = ld64(s + q); ti[(u32)(((v << (64 - 8 * IB)) * 0xC2B2AE3D27D4EB4FULL) >> 40) % IW] = q; }
                if (p + 8 <= n) {
                    u64 v = ld64(s + p);
                    u32 hl = h8(v), hh = hs(v);
                    int32_t ql = tl[hl], qs = ts[hh];
                    if (rep && (u64)c0 + p >= rep) { u32 l = cb(v, ld64(s + p - (long)rep)); if (l >= 4) { bl = l; bo = rep; } }
                    if (REP23 && rep2 && (u64)c0 + p >= rep2) { u32 l = cb(v, ld64(s + p - (long)rep2)); if (l >= 4 && l > bl) { bl = l; bo = rep2; } }
                    if (ql >= 0 && p - ql <= (u32)REACH) { u32 l = cb(v, ld64(s + ql)); if (l >= 5 && l > bl) { bl = l; bo = p - ql; } }
                    if (qs >= 0 && p - qs <= (u32)REACH && bl < 8) { u32 l = cb(v, ld64(s + qs)); if (l >= 5 && l > bl) { bl = l; bo = p - qs; } }
                    if (FAR) { u32 O = hints[(c0 + p) >> FHINT]; if (O && (u64)c0 + p >= O) { u32 l = cb(v, ld64(s + p - (long)O)); if (l >= (u32)FMIN && l > bl) { bl = l; bo = O; } } }
                    if (INTRA == 2) { int32_t qi = ti[(u32)(((v << (64 - 8 * IB)) * 0xC2B2AE3D27D4EB4FULL) >> 40) % IW]; if (qi >= 0) { u32 l = cb(v, ld64(s + qi)); if (l >= (u32)IMIN && (l > bl || (l == bl && p - qi < bo))) { bl = l; bo = p - qi; } } }
                    if (bl >= 8) bl = mlen(s, p, (long)p - (long)bo, n - 7 > p ? n : n, CAP + 7);
                    if (bl > (u32)CAP + 7) bl = CAP + 7;
                    if (INTRA == 1) { tl[hl] = p; ts[hh] = p; }
                }
                blen[p - ip] = bl >= 4 ? bl : 0; boff[p - ip] = bo;
            }
            // ---- lazy + greedy parse
            u32 p = ip + carry;
            u32 last_off = 0;
            while (p < wend) {
                u32 w = p - ip, bl = blen[w];
                if (bl && LAZY && w + 1 < (u32)WIN && blen[w + 1]) {
                    int g1 = 4 * (int)bl - (31 - __builtin_clz(boff[w] + 1)) + 4, g2 = 4 * (int)blen[w + 1] - (31 - __builtin_clz(boff[w + 1] + 1));
                    if (g2 > g1) bl = 0;
                }
                if (bl) {
                    u32 ps = p, o = boff[w];
                    if (BACK) while (ps > anchor && (u64)c0 + ps > o && p - ps < (u32)BACKMAX && s[ps - 1] == s[(long)ps - 1 - (long)o]) { ps--; bl++; }
                    tot_lit += ps - anchor; tot_seq++;
                    {   // zstd.c:19680 repcode rules, for counting only
                        u32 ll = ps - anchor; int hit = 0;
                        if (MERGE && ps == anchor && o == prev_off_m && anchor == prev_end_m && anchor > 0) hit = -1;  // merged away
                        if (hit == 0) {
                            if (ll > 0) { if (o == R[0]) { r1++; hit = 1; } else if (o == R[1]) { r2++; hit = 1; u32 t = R[1]; R[1] = R[0]; R[0] = t; } else if (o == R[2]) { r3++; hit = 1; u32 t = R[2]; R[2] = R[1]; R[1] = R[0]; R[0] = t; } }
                            else { if (o == R[1]) { r2++; hit = 1; u32 t = R[1]; R[1] = R[0]; R[0] = t; } else if (o == R[2]) { r3++; hit = 1; u32 t = R[2]; R[2] = R[1]; R[1] = R[0]; R[0] = t; } else if (o == R[0] - 1 && o) { r1m++; hit = 1; R[2] = R[1]; R[1] = R[0]; R[0] = o; } }
                            if (!hit) { offbits += 32 - __builtin_clz(o + 3); R[2] = R[1]; R[1] = R[0]; R[0] = o; }
                        }
                    }
                    if (MERGE && ps == anchor && o == prev_off_m && anchor == prev_end_m && anchor > 0) tot_seq--;
                    if (o == last_off || o == rep) tot_rep++;
                    if (o != last_off && last_off) rep2 = last_off;
                    last_off = o;
                    { u32 e = p + (bl - (p - ps)); for (u32 x = p + 1; x < e && x < wend; x++) inmatch[x - ip] = 1; }
                    p += blen[w] ? (bl - (p - ps)) : 1; anchor = p; prev_off_m = o; prev_end_m = p;
                } else p++;
            }
            if (last_off) { if (rep != last_off) rep2 = rep; rep = last_off; }
            carry = p - wend; if (p > n) p = n;
            if (INTRA != 1) for (u32 q = ip; q < wend && q + 8 <= n; q++) { if (INSFILT && inmatch[q - ip]) continue; u64 v = ld64(s + q); tl[h8(v)] = q; ts[hs(v)] = q; }
            memset(inmatch, 0, WIN);
            { u32 e = p; for (u32 x = wend; x < e && x < wend + WIN; x++) if (x - wend < (u32)WIN) inmatch[x - wend] = 1; }
        }
        tot_lit += n - (anchor < n ? anchor : n);
    }
    printf("rep1 %llu rep2 %llu rep3 %llu rep1-1 %llu explicit-offset extra bits %llu\n", (unsigned long long)r1, (unsigned long long)r2, (unsigned long long)r3, (unsigned long long)r1m, (unsigned long long)offbits);
    double bits = BL * tot_lit + BS * tot_seq;
    printf("lits %llu seqs %llu (rep-ish %llu) est bytes %.0f ratio %.3f\n", (unsigned long long)tot_lit, (unsigned long long)tot_seq, (unsigned long long)tot_rep, bits / 8, N / (bits / 8));
    return 0;
}
