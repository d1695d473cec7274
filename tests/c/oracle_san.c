/* The oracle restatement under AddressSanitizer / UBSan (CPU only: GPU sanitizers are not available on this pool): every file named on the
 * command line is decoded as it is and in 200 damaged copies (bit flips, truncations, with and without the checksum check).
 * scripts/oracle_sanitize.sh builds it, makes a few reference frames and runs it.  Test infrastructure, like oracle/ itself. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
size_t zlo_decompress(void* dst, size_t cap, const void* src, size_t n, const void* dict, size_t dn, int ign);
unsigned zlo_is_error(size_t c);
int main(int argc, char** argv) {
    for (int a = 1; a < argc; a++) {
        FILE* f = fopen(argv[a], "rb"); if (!f) continue;
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        unsigned char* s = malloc(n); if (fread(s, 1, n, f) != (size_t)n) return 2; fclose(f);
        size_t cap = 64u << 20; unsigned char* d = malloc(cap);
        size_t r = zlo_decompress(d, cap, s, n, 0, 0, 0);
        printf("%s: %s %zu\n", argv[a], zlo_is_error(r) ? "error" : "ok", r);
        /* damaged copies: flips and truncations */
        unsigned seed = 12345;
        for (int k = 0; k < 200; k++) {
            unsigned char* b = malloc(n); memcpy(b, s, n);
            seed = seed * 1664525u + 1013904223u; size_t pos = seed % n; b[pos] ^= 1u << ((seed >> 20) & 7);
            seed = seed * 1664525u + 1013904223u; size_t len = (k & 1) ? n : 1 + seed % n;
            (void)zlo_decompress(d, cap, b, len, 0, 0, (k >> 1) & 1);
            free(b);
        }
        free(s); free(d);
    }
    return 0;
}
