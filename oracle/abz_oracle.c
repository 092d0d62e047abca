/*
 * C restatement of the reference's timed hot loops -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.
 * ("CPU restatement, not Julia": the Julia reference cannot run in this pipeline.)
 *
 * Follows the reference's loop structure so that timing it is a fair stand-in for the reference's
 * CPU path:
 *   orc_fourier_ptr   hierarchical evaluation on a PTR grid, contracting the outermost dimension
 *                     first, vals[i1,...,id] column-major, threads over the outermost grid index
 *                     with one workspace per thread       ref: src/fourier.jl:132-164 (156-161)
 *   orc_eig_herm      eigen(Hermitian(h)) by cyclic Jacobi on the upper triangle
 *                                                         ref: src/dos_ggr.jl:19,34
 *   orc_dos_scan      per-omega weighted sum over cached H(k) of -Im tr inv((w+i eta)I - H)/pi
 *                                                         ref: quadsum at src/fourier.jl:204-207 with
 *                                                         the integrand of aps_example/aps_example.jl:30
 *   orc_fourier_ptr3 / orc_dos_scan3   the same two loops for n = 3 bands at fixed size, doing what the
 *                     reference does for SMatrix{3,3}: series terms as unrolled 3x3 complex multiply-adds,
 *                     eigenvalues by StaticArrays' closed form for 3x3 Hermitian matrices (trigonometric
 *                     roots of the characteristic cubic, ref: src/dos_ggr.jl:19), inverse by the adjugate
 *                     (StaticArrays inv of a 3x3, ref: aps_example/aps_example.jl:30).  Built with
 *                     -fcx-limited-range: Julia's complex multiply has no Annex-G NaN recovery branch.
 *                     These are the cpu_baseline of bench.py.
 * Parity status: pinned through tests/test_oracle_c.py against the numpy oracle (itself pinned
 * analytically, oracle/abz_oracle.py).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex cd;

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* eigenvalues (ascending) of the Hermitian matrix given by the upper triangle of h (n x n,
 * column-major like Julia: h[a + n*b]). */
void orc_eig_herm(const cd* h, int n, double* e) {
    cd A[32 * 32];
    double norm2 = 0.0;
    for (int b = 0; b < n; ++b)
        for (int a = 0; a <= b; ++a) {
            cd v = (a == b) ? creal(h[a + n * b]) : h[a + n * b];
            A[a + n * b] = v;
            A[b + n * a] = conj(v);
            norm2 += (a == b ? 1.0 : 2.0) * (creal(v) * creal(v) + cimag(v) * cimag(v));
        }
    const double tiny = 1e-34 * norm2;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off2 = 0.0;
        for (int q = 1; q < n; ++q)
            for (int p = 0; p < q; ++p) off2 += creal(A[p + n * q]) * creal(A[p + n * q]) + cimag(A[p + n * q]) * cimag(A[p + n * q]);
        if (!(off2 > tiny)) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const cd al = A[p + n * q];
                const double b2 = creal(al) * creal(al) + cimag(al) * cimag(al);
                if (!(b2 > tiny)) continue;
                const double b = sqrt(b2);
                const cd g = conj(al) / b;
                const double d = creal(A[q + n * q]) - creal(A[p + n * p]);
                const double t = copysign(2.0 * b, d) / (fabs(d) + sqrt(d * d + 4.0 * b2));
                const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                A[p + n * p] = creal(A[p + n * p]) - t * b;
                A[q + n * q] = creal(A[q + n * q]) + t * b;
                A[p + n * q] = 0.0;
                A[q + n * p] = 0.0;
                for (int r = 0; r < n; ++r) {
                    if (r == p || r == q) continue;
                    const cd x = A[r + n * p], y = A[r + n * q];
                    const cd np_ = c * x - s * g * y, nq_ = s * x + c * g * y;
                    A[r + n * p] = np_;
                    A[r + n * q] = nq_;
                    A[p + n * r] = conj(np_);
                    A[q + n * r] = conj(nq_);
                }
            }
    }
    for (int a = 0; a < n; ++a) e[a] = creal(A[a + n * a]);
    for (int i = 1; i < n; ++i) { /* insertion sort */
        double v = e[i];
        int j = i - 1;
        while (j >= 0 && e[j] > v) {
            e[j + 1] = e[j];
            --j;
        }
        e[j + 1] = v;
    }
}

/* tr inv(A) for a general complex n x n matrix (column-major), Gauss-Jordan with partial pivoting */
static cd trace_inverse(const cd* Ain, int n) {
    cd A[32 * 32], X[32 * 32];
    memcpy(A, Ain, sizeof(cd) * (size_t)n * n);
    for (int i = 0; i < n * n; ++i) X[i] = 0.0;
    for (int i = 0; i < n; ++i) X[i + n * i] = 1.0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = cabs(A[c + n * c]);
        for (int r = c + 1; r < n; ++r)
            if (cabs(A[r + n * c]) > best) {
                best = cabs(A[r + n * c]);
                piv = r;
            }
        if (piv != c)
            for (int b = 0; b < n; ++b) {
                cd t = A[c + n * b];
                A[c + n * b] = A[piv + n * b];
                A[piv + n * b] = t;
                t = X[c + n * b];
                X[c + n * b] = X[piv + n * b];
                X[piv + n * b] = t;
            }
        const cd ip = 1.0 / A[c + n * c];
        for (int b = 0; b < n; ++b) {
            A[c + n * b] *= ip;
            X[c + n * b] *= ip;
        }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const cd f = A[r + n * c];
            for (int b = 0; b < n; ++b) {
                A[r + n * b] -= f * A[c + n * b];
                X[r + n * b] -= f * X[c + n * b];
            }
        }
    }
    cd tr = 0.0;
    for (int i = 0; i < n; ++i) tr += X[i + n * i];
    return tr;
}

/* Hierarchical PTR evaluation, d = 1..3.  coef in Julia order [M_d]..[M_1][n*n]; vals [npt^d][n*n]
 * with i_1 fastest; eig [npt^d][n] or NULL.  Phase of frequency f at grid index i: e^{2 pi i f i / npt}. */
void orc_fourier_ptr(const cd* coef, int d, const int* dims, const int* first, int n, int npt, cd* vals, double* eig) {
    const int nn = n * n;
    int M[3] = {1, 1, 1}, F[3] = {0, 0, 0}, G[3] = {1, 1, 1};
    for (int j = 0; j < d; ++j) {
        M[j] = dims[j];
        F[j] = first[j];
        G[j] = npt;
    }
    /* phase tables ph[j][i*M + m] */
    cd* ph[3];
    for (int j = 0; j < 3; ++j) {
        ph[j] = (cd*)malloc(sizeof(cd) * (size_t)G[j] * M[j]);
        for (int i = 0; i < G[j]; ++i)
            for (int m = 0; m < M[j]; ++m) {
                const double ang = 2.0 * M_PI * (double)(F[j] + m) * ((double)i / (double)npt);
                ph[j][i * M[j] + m] = (j < d) ? (cos(ang) + I * sin(ang)) : 1.0;
            }
    }
    const int64_t L2 = (int64_t)M[1] * M[0] * nn, L1 = (int64_t)M[0] * nn;
#pragma omp parallel
    {
        cd* c2 = (cd*)malloc(sizeof(cd) * (size_t)L2);
        cd* c1 = (cd*)malloc(sizeof(cd) * (size_t)L1);
#pragma omp for schedule(static)
        for (int i3 = 0; i3 < G[2]; ++i3) {
            /* contract dim 3 */
            for (int64_t l = 0; l < L2; ++l) c2[l] = 0.0;
            for (int m = 0; m < M[2]; ++m) {
                const cd p = ph[2][i3 * M[2] + m];
                const cd* src = coef + (int64_t)m * L2;
                for (int64_t l = 0; l < L2; ++l) c2[l] += src[l] * p;
            }
            for (int i2 = 0; i2 < G[1]; ++i2) {
                for (int64_t l = 0; l < L1; ++l) c1[l] = 0.0;
                for (int m = 0; m < M[1]; ++m) {
                    const cd p = ph[1][i2 * M[1] + m];
                    const cd* src = c2 + (int64_t)m * L1;
                    for (int64_t l = 0; l < L1; ++l) c1[l] += src[l] * p;
                }
                for (int i1 = 0; i1 < G[0]; ++i1) {
                    const int64_t k = ((int64_t)i3 * G[1] + i2) * G[0] + i1;
                    cd* out = vals + k * nn;
                    for (int a = 0; a < nn; ++a) out[a] = 0.0;
                    for (int m = 0; m < M[0]; ++m) {
                        const cd p = ph[0][i1 * M[0] + m];
                        const cd* src = c1 + (int64_t)m * nn;
                        for (int a = 0; a < nn; ++a) out[a] += src[a] * p;
                    }
                    if (eig) orc_eig_herm(out, n, eig + k * n);
                }
            }
        }
        free(c2);
        free(c1);
    }
    for (int j = 0; j < 3; ++j) free(ph[j]);
}

/* out[w] = (1/nk) sum_k -Im tr inv((omega_w + i eta) I - H_k) / pi   (one pass over vals per omega,
 * like quadsum with the user integrand; threads over k) */
void orc_dos_scan(const cd* vals, int64_t nk, int n, double eta, const double* omegas, int nw, double* out) {
    const int nn = n * n;
    for (int w = 0; w < nw; ++w) {
        const cd z = omegas[w] + I * eta;
        double acc = 0.0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
        for (int64_t k = 0; k < nk; ++k) {
            cd A[32 * 32];
            const cd* h = vals + k * nn;
            for (int i = 0; i < nn; ++i) A[i] = -h[i];
            for (int i = 0; i < n; ++i) A[i + n * i] += z;
            acc += -cimag(trace_inverse(A, n)) / M_PI;
        }
        out[w] = acc / (double)nk;
    }
}

/* ---------------------------------------------------------------------------------------------
 * n = 3 at fixed size (the SVO workload): what the reference's SMatrix{3,3} code paths do.
 * ------------------------------------------------------------------------------------------- */
/* eigvals(Hermitian(h)) for 3x3, upper triangle, column-major: closed form (trigonometric solution of the
 * characteristic cubic), ascending.  StaticArrays uses this form for 3x3 Hermitian matrices. */
static inline void eig3_closed(const cd* h, double* e) {
    const double a00 = creal(h[0]), a11 = creal(h[4]), a22 = creal(h[8]);
    const cd a01 = h[3], a02 = h[6], a12 = h[7];
    const double n01 = creal(a01) * creal(a01) + cimag(a01) * cimag(a01);
    const double n02 = creal(a02) * creal(a02) + cimag(a02) * cimag(a02);
    const double n12 = creal(a12) * creal(a12) + cimag(a12) * cimag(a12);
    const double p1 = n01 + n02 + n12;
    const double q = (a00 + a11 + a22) / 3.0;
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p2 = b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1;
    if (!(p2 > 0.0)) {
        e[0] = e[1] = e[2] = q;
        return;
    }
    const double p = sqrt(p2 / 6.0);
    /* det(B), B = A - qI Hermitian: b00 b11 b22 + 2 Re(a01 a12 conj(a02)) - b00|a12|^2 - b11|a02|^2 - b22|a01|^2 */
    const cd t = a01 * a12 * conj(a02);
    const double detB = b00 * b11 * b22 + 2.0 * creal(t) - b00 * n12 - b11 * n02 - b22 * n01;
    double r = detB / (2.0 * p * p * p);
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    const double phi = acos(r) / 3.0;
    const double e3 = q + 2.0 * p * cos(phi);
    const double e1 = q + 2.0 * p * cos(phi + 2.0 * M_PI / 3.0);
    e[0] = e1;
    e[1] = 3.0 * q - e1 - e3;
    e[2] = e3;
}

void orc_eig3_closed(const cd* h, double* e) { eig3_closed(h, e); }

/* tr inv(A), A general complex 3x3 column-major, by the adjugate (only the diagonal cofactors survive the trace) */
static inline cd trace_inverse3(const cd* A) {
    const cd c00 = A[4] * A[8] - A[7] * A[5];
    const cd c11 = A[0] * A[8] - A[6] * A[2];
    const cd c22 = A[0] * A[4] - A[3] * A[1];
    const cd c01 = A[7] * A[2] - A[1] * A[8]; /* cofactors of the first column for the determinant */
    const cd c02 = A[1] * A[5] - A[4] * A[2];
    const cd det = A[0] * c00 + A[3] * c01 + A[6] * c02;
    return (c00 + c11 + c22) / det;
}

void orc_fourier_ptr3(const cd* coef, int d, const int* dims, const int* first, int n, int npt, cd* vals, double* eig) {
    if (n != 3 || d != 3) {
        orc_fourier_ptr(coef, d, dims, first, n, npt, vals, eig);
        return;
    }
    enum { NN = 9 };
    const int M0 = dims[0], M1 = dims[1], M2 = dims[2];
    cd* ph[3];
    for (int j = 0; j < 3; ++j) {
        ph[j] = (cd*)malloc(sizeof(cd) * (size_t)npt * dims[j]);
        for (int i = 0; i < npt; ++i)
            for (int m = 0; m < dims[j]; ++m) {
                const double ang = 2.0 * M_PI * (double)(first[j] + m) * ((double)i / (double)npt);
                ph[j][i * dims[j] + m] = cos(ang) + I * sin(ang);
            }
    }
    const int64_t L2 = (int64_t)M1 * M0 * NN, L1 = (int64_t)M0 * NN;
#pragma omp parallel
    {
        cd* c2 = (cd*)malloc(sizeof(cd) * (size_t)L2);
        cd* c1 = (cd*)malloc(sizeof(cd) * (size_t)L1);
#pragma omp for schedule(static)
        for (int i3 = 0; i3 < npt; ++i3) {
            for (int64_t l = 0; l < L2; ++l) c2[l] = 0.0;
            for (int m = 0; m < M2; ++m) {
                const cd p = ph[2][i3 * M2 + m];
                const cd* src = coef + (int64_t)m * L2;
                for (int64_t l = 0; l < L2; ++l) c2[l] += src[l] * p;
            }
            for (int i2 = 0; i2 < npt; ++i2) {
                for (int64_t l = 0; l < L1; ++l) c1[l] = 0.0;
                for (int m = 0; m < M1; ++m) {
                    const cd p = ph[1][i2 * M1 + m];
                    const cd* src = c2 + (int64_t)m * L1;
                    for (int64_t l = 0; l < L1; ++l) c1[l] += src[l] * p;
                }
                for (int i1 = 0; i1 < npt; ++i1) {
                    const int64_t k = ((int64_t)i3 * npt + i2) * npt + i1;
                    cd acc[NN];
                    for (int a = 0; a < NN; ++a) acc[a] = 0.0;
                    const cd* pp = ph[0] + (int64_t)i1 * M0;
                    for (int m = 0; m < M0; ++m) {
                        const cd p = pp[m];
                        const cd* src = c1 + (int64_t)m * NN;
                        for (int a = 0; a < NN; ++a) acc[a] += src[a] * p;
                    }
                    cd* out = vals + k * NN;
                    for (int a = 0; a < NN; ++a) out[a] = acc[a];
                    if (eig) eig3_closed(acc, eig + k * 3);
                }
            }
        }
        free(c2);
        free(c1);
    }
    for (int j = 0; j < 3; ++j) free(ph[j]);
}

void orc_dos_scan3(const cd* vals, int64_t nk, int n, double eta, const double* omegas, int nw, double* out) {
    if (n != 3) {
        orc_dos_scan(vals, nk, n, eta, omegas, nw, out);
        return;
    }
    for (int w = 0; w < nw; ++w) {
        const cd z = omegas[w] + I * eta;
        double acc = 0.0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
        for (int64_t k = 0; k < nk; ++k) {
            cd A[9];
            const cd* h = vals + k * 9;
            for (int i = 0; i < 9; ++i) A[i] = -h[i];
            A[0] += z;
            A[4] += z;
            A[8] += z;
            acc += -cimag(trace_inverse3(A)) / M_PI;
        }
        out[w] = acc / (double)nk;
    }
}
