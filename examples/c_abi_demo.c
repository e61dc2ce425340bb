/*
 * Plain-C user of the C ABI (include/overiva_hip.h): OverIVA on a synthetic (T, F, M) complex64 tensor.
 *   gcc -std=c99 -I include examples/c_abi_demo.c -L overiva_amd -loveriva_hip -Wl,-rpath,$PWD/overiva_amd -lm -o c_abi_demo
 * Exit code 0 on success, 2 when no GPU is present (the library reports it; there is no CPU fallback).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "overiva_hip.h"

static float frand(unsigned *s) { /* uniform (-1, 1), LCG */
    *s = *s * 1664525u + 1013904223u;
    return (float)((*s >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

int main(void) {
    const int T = 256, F = 65, M = 4, K = 2;
    int ndev = 0;
    if (oiva_device_count(&ndev) != OIVA_OK || ndev == 0) {
        fprintf(stderr, "no GPU: %s\n", oiva_last_error());
        return 2;
    }
    float *X = malloc(sizeof(float) * 2 * T * F * M);
    float *Y = malloc(sizeof(float) * 2 * T * F * K);
    float *W = malloc(sizeof(float) * 2 * F * M * K);
    unsigned seed = 12345u;
    for (long i = 0; i < 2L * T * F * M; ++i) X[i] = frand(&seed);

    oiva_plan *p = NULL;
    if (oiva_plan_create(&p, 0, T, F, M, K, OIVA_MODEL_LAPLACE, F, NULL) != OIVA_OK ||
        oiva_plan_set_x_host(p, X, 0) != OIVA_OK ||      /* overiva.py:132 */
        oiva_plan_covariance(p) != OIVA_OK ||            /* overiva.py:87 */
        oiva_plan_set_w(p, NULL, 0) != OIVA_OK ||        /* overiva.py:89-123, identity start */
        oiva_plan_iterate(p, 20) != OIVA_OK ||           /* overiva.py:138-190 */
        oiva_plan_demix(p, Y, 0, 1) != OIVA_OK ||        /* overiva.py:192-199 */
        oiva_plan_get_w(p, W, 0) != OIVA_OK) {           /* overiva.py:201-202 */
        fprintf(stderr, "overiva_hip: %s\n", oiva_last_error());
        return 1;
    }
    double e = 0.0;
    for (long i = 0; i < 2L * T * F * K; ++i) e += (double)Y[i] * Y[i];
    printf("OverIVA %d x %d x %d / %d, 20 iterations: output energy %.6g, W[0] = (%g, %g)\n", F, T, M, K, e, W[0], W[1]);
    oiva_plan_destroy(p);
    free(X); free(Y); free(W);
    return isfinite(e) ? 0 : 1;
}
