/* gcd of two polynomials over Fp (Montgomery limbs from the oracle), classical Euclid with OpenMP saxpy.
 * in: a.bin (na coefficients, low first), b.bin (nb).  out: gcd.bin (monic), prints degree. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fp252.h"      /* oracle/fp252.h: build with -I oracle -L oracle/_build -loracle */

static fp_t *load(const char *path, long *n) {
    FILE *f = fopen(path, "rb");
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    fp_t *p = malloc(sz); if (fread(p, 1, sz, f) != (size_t)sz) exit(2); fclose(f);
    *n = sz / 32; return p;
}
static long degree(const fp_t *p, long n) { while (n > 0 && fp_is_zero(p[n - 1])) --n; return n - 1; }

int main(int argc, char **argv) {
    fp_init();
    long na, nb;
    fp_t *a = load(argv[1], &na), *b = load(argv[2], &nb);
    long da = degree(a, na), db = degree(b, nb);
    long steps = 0;
    while (db >= 0) {
        if (da < db) { fp_t *t = a; a = b; b = t; long d = da; da = db; db = d; }
        if (db < 0) break;
        /* reduce a by b until deg a < deg b */
        const fp_t inv = fp_inv(b[db]);
        while (da >= db) {
            const fp_t q = fp_mul(a[da], inv);
            const long sh = da - db;
            #pragma omp parallel for schedule(static)
            for (long i = 0; i < db; ++i) a[i + sh] = fp_sub(a[i + sh], fp_mul(q, b[i]));
            fp_t z = {{0,0,0,0}}; a[da] = z;
            da = degree(a, da);
            if ((++steps & 0x3fff) == 0) { fprintf(stderr, "deg %ld / %ld\n", da, db); }
        }
        fp_t *t = a; a = b; b = t; long d = da; da = db; db = d;
    }
    /* a is the gcd; make monic */
    fp_t inv = fp_inv(a[da]);
    for (long i = 0; i <= da; ++i) a[i] = fp_mul(a[i], inv);
    printf("gcd degree %ld\n", da);
    FILE *f = fopen(argv[3], "wb"); fwrite(a, 32, da + 1, f); fclose(f);
    return 0;
}
