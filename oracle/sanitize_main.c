/*
 * sanitize_main.c — TEST INFRASTRUCTURE ONLY (like csr_oracle.c).  Drives every function of the C oracle on small inputs —
 * including empty matrices, empty rows and single entries — so that a build with -fsanitize=address,undefined
 * (tests/test_oracle_golden.py::test_c_oracle_is_clean_under_address_and_undefined_sanitizers; CPU only, never on the GPU
 * box's device) sees every loop.  Exits 0 when the results are the ones a dense evaluation gives.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "csr_oracle.c"

#define N 5
#define P 3

static int close_to(double a, double b) { return fabs(a - b) <= 1e-9 * (1.0 + fabs(b)); }

int main(void) {
    /* lower triangular 5 x 5 with an empty-below-diagonal row: rows {0}, {0,1}, {2}, {1,3}, {0,2,4} */
    const int64_t crow[N + 1] = {0, 1, 3, 4, 6, 9};
    const int64_t col[9] = {0, 0, 1, 2, 1, 3, 0, 2, 4};
    const double val[9] = {2.0, -1.0, 3.0, 1.5, 0.5, 4.0, -2.0, 1.0, 2.5};
    double dense[N][N] = {{0}};
    for (int i = 0; i < N; ++i)
        for (int64_t k = crow[i]; k < crow[i + 1]; ++k) dense[i][col[k]] = val[k];
    double B[N * P], G[N * P], C[N * P], D[N * P], X[N * P], out[9];
    for (int i = 0; i < N * P; ++i) B[i] = 0.25 * (i + 1), G[i] = 1.0 - 0.1 * i;
    int bad = 0;

    oracle_csr_spmm_f64(N, crow, col, val, B, P, C, P, P);
    oracle_csr_spmm_t_f64(N, N, crow, col, val, G, P, D, P, P);
    oracle_csr_sddmm_f64(N, crow, col, G, P, B, P, -1.0, out, P);
    for (int i = 0; i < N; ++i)
        for (int q = 0; q < P; ++q) {
            double c = 0, d = 0;
            for (int j = 0; j < N; ++j) c += dense[i][j] * B[j * P + q], d += dense[j][i] * G[j * P + q];
            bad += !close_to(C[i * P + q], c) + !close_to(D[i * P + q], d);
        }
    for (int i = 0; i < N; ++i)
        for (int64_t k = crow[i]; k < crow[i + 1]; ++k) {
            double s = 0;
            for (int q = 0; q < P; ++q) s += G[i * P + q] * B[col[k] * P + q];
            bad += !close_to(out[k], -s);
        }
    {   /* COO form of the same rule */
        int64_t row[9];
        for (int i = 0; i < N; ++i)
            for (int64_t k = crow[i]; k < crow[i + 1]; ++k) row[k] = i;
        double out2[9];
        oracle_coo_sddmm_f64(9, row, col, G, P, B, P, -1.0, out2, P);
        for (int k = 0; k < 9; ++k) bad += !close_to(out2[k], out[k]);
    }
    /* triangular solves: all flag combinations on the lower matrix and on its transpose pattern read as upper */
    for (int unit = 0; unit < 2; ++unit)
        for (int tr = 0; tr < 2; ++tr) {
            if (oracle_csr_sptrsm_f64(N, crow, col, val, 0, unit, tr, B, P, X, P, P) != 0) bad += 100;
            for (int i = 0; i < N; ++i)
                for (int q = 0; q < P; ++q) {
                    double s = 0;
                    for (int j = 0; j < N; ++j) {
                        double t = tr ? dense[j][i] : dense[i][j];
                        if (unit && i == j) t = 1.0;
                        s += t * X[j * P + q];
                    }
                    bad += !close_to(s, B[i * P + q]);
                }
        }
    if (oracle_csr_levels(N, crow, col, 0) < 1) bad += 1000;
    /* float instantiations + empty inputs */
    {
        const float valf[9] = {2.f, -1.f, 3.f, 1.5f, .5f, 4.f, -2.f, 1.f, 2.5f};
        float Bf[N * P], Cf[N * P], of[9];
        for (int i = 0; i < N * P; ++i) Bf[i] = 0.25f * (i + 1);
        oracle_csr_spmm_f32(N, crow, col, valf, Bf, P, Cf, P, P);
        oracle_csr_sddmm_f32(N, crow, col, Bf, P, Bf, P, 1.f, of, P);
        const int64_t crow0[1] = {0};
        oracle_csr_spmm_f32(0, crow0, col, valf, Bf, P, Cf, P, P);
        oracle_csr_spmm_t_f32(0, 0, crow0, col, valf, Bf, P, Cf, P, P);
        oracle_csr_sddmm_f32(0, crow0, col, Bf, P, Bf, P, 1.f, of, P);
        oracle_coo_sddmm_f32(0, col, col, Bf, P, Bf, P, 1.f, of, P);
        if (oracle_csr_sptrsm_f32(0, crow0, col, valf, 1, 0, 0, Bf, P, Cf, P, P) != 0) bad += 100;
        /* p = 0 columns */
        oracle_csr_spmm_f32(N, crow, col, valf, Bf, 0, Cf, 0, 0);
    }
    if (oracle_abi_version() != 1) bad += 10000;
    printf("sanitize_main: %d mismatches\n", bad);
    return bad ? 1 : 0;
}
