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
#include <time.h>
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

/* tr inv(A) for a general complex n x n matrix (column-major): in-place Gauss-Jordan inversion with partial pivoting
 * (n^3 complex multiply-adds: what a full `inv` costs, which is what the reference's integrand computes before `tr`,
 * aps_example/aps_example.jl:30), the row interchanges undone on the columns at the end. */
static cd trace_inverse(const cd* Ain, int n) {
    cd A[32 * 32];
    int piv[32];
    memcpy(A, Ain, sizeof(cd) * (size_t)n * n);
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = creal(A[c + n * c]) * creal(A[c + n * c]) + cimag(A[c + n * c]) * cimag(A[c + n * c]);
        for (int r = c + 1; r < n; ++r) {
            const double v = creal(A[r + n * c]) * creal(A[r + n * c]) + cimag(A[r + n * c]) * cimag(A[r + n * c]);
            if (v > best) {
                best = v;
                p = r;
            }
        }
        piv[c] = p;
        if (p != c)
            for (int b = 0; b < n; ++b) {
                const cd t = A[c + n * b];
                A[c + n * b] = A[p + n * b];
                A[p + n * b] = t;
            }
        const cd ip = 1.0 / A[c + n * c];
        A[c + n * c] = 1.0;
        for (int b = 0; b < n; ++b) A[c + n * b] *= ip;
        for (int b = 0; b < n; ++b) { /* column b of every other row: A[r][b] -= A[r][c] * A[c][b], A[r][c] <- -A[r][c] / pivot */
            if (b == c) continue;
            const cd u = A[c + n * b];
            for (int r = 0; r < n; ++r)
                if (r != c) A[r + n * b] -= A[r + n * c] * u;
        }
        for (int r = 0; r < n; ++r)
            if (r != c) A[r + n * c] = -A[r + n * c] * ip;
    }
    for (int c = n - 1; c >= 0; --c) /* inv(P A) = inv(A) P': swap the columns back */
        if (piv[c] != c)
            for (int r = 0; r < n; ++r) {
                const cd t = A[r + n * c];
                A[r + n * c] = A[r + n * piv[c]];
                A[r + n * piv[c]] = t;
            }
    cd tr = 0.0;
    for (int i = 0; i < n; ++i) tr += A[i + n * i];
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

/* =============================================================================================
 * IAI: nested globally adaptive GK(7,15), depth first, for the DOS integrand on a box of limits.
 * ref: src/fourier.jl:432-510 (outer levels contract the series at the node and recurse with abstol / len, reltol
 * unchanged; the innermost level evaluates the 1-D series and calls f), src/algorithms.jl:202-240 (AuxQuadGKJL, order 7),
 * QuadGK's adapt loop (pop the worst segment, bisect, 30 new nodes, push; E = |I_K - I_G| h; DataStructures' binary heap).
 * Threads: the nodes of an OUTERMOST round (15, then 30 per bisection) are dealt to threads, every thread with its own
 * contraction buffers -- what NestedBatchIntegrand does with its per-thread workers (src/fourier.jl:441-473); the inner
 * levels run serially inside a thread.  Same panels, same numevals as oracle/abz_oracle.py::solve_iai (scalar refinement).
 * ============================================================================================= */
static const double GKX[7] = {-0.991455371120812639206854697526329, -0.949107912342758524526189684047851,
                              -0.864864423359769072789712788640926, -0.741531185599394439863864773280788,
                              -0.586087235467691130294144838258730, -0.405845151377397166906606412076961,
                              -0.207784955007898467600689403773245};
static const double GKW[8] = {0.022935322010529224963732008058970, 0.063092092629978553290700663189204,
                              0.104790010322250183839876322541518, 0.140653259715525918745189590510238,
                              0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
                              0.204432940075298892414161999234649, 0.209482141084727828012999174891714};
static const double GKGW[4] = {0.129484966168869693270611432679082, 0.279705391489276667901467771423780,
                               0.381830050505118944950369775488975, 0.417959183673469387755102040816327};

static void gk_nodes15(double a, double b, double* x) {
    const double s = 0.5 * (b - a);
    for (int i = 0; i < 7; ++i) {
        x[2 * i] = a + (1 + GKX[i]) * s;
        x[2 * i + 1] = a + (1 - GKX[i]) * s;
    }
    x[14] = a + s;
}

/* Kronrod value and error of one segment from its 15 values (QuadGK.evalrule's order and summation order) */
static double gk_rule15(const double* fv, double a, double b, double* E) {
    const double s = 0.5 * (b - a);
    double fg = fv[2] + fv[3], fk = fv[0] + fv[1];
    double Ig = fg * GKGW[0];
    double Ik = fg * GKW[1] + fk * GKW[0];
    for (int i = 2; i < 4; ++i) {
        fg = fv[2 * (2 * i - 1)] + fv[2 * (2 * i - 1) + 1];
        fk = fv[2 * (2 * i - 2)] + fv[2 * (2 * i - 2) + 1];
        Ig = Ig + fg * GKGW[i - 1];
        Ik = Ik + fg * GKW[2 * i - 1] + fk * GKW[2 * i - 2];
    }
    Ig = Ig + fv[14] * GKGW[3];
    Ik = Ik + fv[14] * GKW[7] + (fv[12] + fv[13]) * GKW[6];
    *E = fabs(Ik * s - Ig * s);
    return Ik * s;
}

typedef struct {
    double a, b, v, E;
} seg_t;

/* DataStructures.jl's binary heap ordered by Reverse on E (max-heap), percolate_down / percolate_up */
static void heap_down(seg_t* xs, int i, seg_t x, int len) {
    for (;;) {
        const int l = 2 * i + 1;
        if (l >= len) break;
        const int r = l + 1;
        const int j = (r >= len || xs[r].E < xs[l].E) ? l : r;
        if (!(x.E < xs[j].E)) break;
        xs[i] = xs[j];
        i = j;
    }
    xs[i] = x;
}
static void heap_up(seg_t* xs, int i, seg_t x) {
    while (i > 0) {
        const int j = (i - 1) / 2;
        if (!(xs[j].E < x.E)) break;
        xs[i] = xs[j];
        i = j;
    }
    xs[i] = x;
}

typedef void (*batch_fn)(void* ctx, const double* xs, int n, double* out);

/* A wall-clock budget for bench.py's bounded CPU samples (not part of the algorithm: with a deadline set, every level stops
 * refining once it has passed; the evaluations made until then are counted as usual, so nodes / second stays meaningful and
 * the value does not).  0: none. */
static double orc_deadline = 0.0;
static double orc_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void orc_set_deadline(double seconds_from_now) { orc_deadline = seconds_from_now > 0.0 ? orc_now() + seconds_from_now : 0.0; }
int orc_deadline_passed(void) { return orc_deadline > 0.0 && orc_now() > orc_deadline; }

/* auxquadgk on [a, b], scalar refinement; returns I, *Eout the error estimate */
static double adapt_gk(batch_fn f, void* ctx, double a, double b, double atol, double rtol, int64_t maxevals, double* Eout) {
    int cap = 64, len = 1;
    seg_t* xs = (seg_t*)malloc(sizeof(seg_t) * (size_t)cap);
    double x[30], fv[30];
    gk_nodes15(a, b, x);
    f(ctx, x, 15, fv);
    xs[0].a = a;
    xs[0].b = b;
    xs[0].v = gk_rule15(fv, a, b, &xs[0].E);
    double Iv = xs[0].v, E = xs[0].E;
    int64_t numevals = 15;
    if (!(E <= fmax(atol, rtol * fabs(Iv)) || numevals >= maxevals)) {
        while (E > fmax(atol, rtol * fabs(Iv)) && numevals < maxevals && !orc_deadline_passed()) {
            const seg_t s = xs[0];
            const seg_t y = xs[--len];
            if (len > 0) heap_down(xs, 0, y, len);
            numevals += 30;
            const double mid = (s.a + s.b) / 2;
            gk_nodes15(s.a, mid, x);
            gk_nodes15(mid, s.b, x + 15);
            f(ctx, x, 30, fv);
            seg_t s1, s2;
            s1.a = s.a;
            s1.b = mid;
            s1.v = gk_rule15(fv, s.a, mid, &s1.E);
            s2.a = mid;
            s2.b = s.b;
            s2.v = gk_rule15(fv + 15, mid, s.b, &s2.E);
            Iv = (Iv - s.v) + s1.v + s2.v;
            E = (E - s.E) + s1.E + s2.E;
            if (len + 2 > cap) {
                cap *= 2;
                xs = (seg_t*)realloc(xs, sizeof(seg_t) * (size_t)cap);
            }
            heap_up(xs, len, s1);
            ++len;
            heap_up(xs, len, s2);
            ++len;
        }
        Iv = xs[0].v;
        E = xs[0].E;
        for (int i = 1; i < len; ++i) {
            Iv = Iv + xs[i].v;
            E = E + xs[i].E;
        }
    }
    free(xs);
    *Eout = E;
    return Iv;
}

typedef struct {
    const cd* coef;  /* [M3][M2][M1][n*n] */
    int n, M[3], F[3];
    double lo[3], hi[3];
    double eta, omega, rtol;
    int64_t maxevals;
    int64_t* counters; /* per thread */
} iai_t;

typedef struct {
    const iai_t* p;
    cd* c2;          /* this thread's level-2 set (after fixing variable 3) */
    cd* c1;          /* level-1 set */
    double atol_inner; /* tolerance of the level below */
    int64_t* count;
} iai_thread_t;

/* innermost: DOS at the nodes of one variable-1 batch */
static void iai_level1(void* vctx, const double* xs, int nx, double* out) {
    iai_thread_t* t = (iai_thread_t*)vctx;
    const iai_t* p = t->p;
    const int n = p->n, nn = n * n, M = p->M[0];
    cd A[32 * 32];
    for (int k = 0; k < nx; ++k) {
        for (int i = 0; i < nn; ++i) A[i] = 0.0;
        for (int m = 0; m < M; ++m) {
            const double ang = 2.0 * M_PI * (double)(p->F[0] + m) * xs[k];
            const cd ph = cos(ang) + I * sin(ang);
            const cd* src = t->c1 + (int64_t)m * nn;
            for (int i = 0; i < nn; ++i) A[i] -= src[i] * ph;
        }
        const cd z = p->omega + I * p->eta;
        for (int i = 0; i < n; ++i) A[i + n * i] += z;
        out[k] = -cimag(n == 3 ? trace_inverse3(A) : trace_inverse(A, n)) / M_PI;
    }
    *t->count += nx;
}

static void contract_level(const cd* src, int Msrc, int first, int64_t L, double x, cd* dst) {
    for (int64_t l = 0; l < L; ++l) dst[l] = 0.0;
    for (int m = 0; m < Msrc; ++m) {
        const double ang = 2.0 * M_PI * (double)(first + m) * x;
        const cd ph = cos(ang) + I * sin(ang);
        const cd* s = src + (int64_t)m * L;
        for (int64_t l = 0; l < L; ++l) dst[l] += s[l] * ph;
    }
}

/* middle level: for each y, contract variable 2 and integrate over variable 1 */
static void iai_level2(void* vctx, const double* ys, int ny, double* out) {
    iai_thread_t* t = (iai_thread_t*)vctx;
    const iai_t* p = t->p;
    const int nn = p->n * p->n;
    for (int k = 0; k < ny; ++k) {
        contract_level(t->c2, p->M[1], p->F[1], (int64_t)p->M[0] * nn, ys[k], t->c1);
        double E;
        out[k] = adapt_gk(iai_level1, t, p->lo[0], p->hi[0], t->atol_inner, p->rtol, p->maxevals, &E);
    }
}

typedef struct {
    const iai_t* p;
    double atol2, atol1; /* tolerances handed to levels 2 and 1 */
} iai_top_t;

/* outermost level: the nodes of a round are dealt to threads */
static void iai_level3(void* vctx, const double* zs, int nz, double* out) {
    const iai_top_t* top = (const iai_top_t*)vctx;
    const iai_t* p = top->p;
    const int nn = p->n * p->n;
#pragma omp parallel
    {
        iai_thread_t t;
        t.p = p;
        t.c2 = (cd*)malloc(sizeof(cd) * (size_t)p->M[1] * p->M[0] * nn);
        t.c1 = (cd*)malloc(sizeof(cd) * (size_t)p->M[0] * nn);
        t.atol_inner = top->atol1;
#ifdef _OPENMP
        t.count = p->counters + 8 * omp_get_thread_num();
#else
        t.count = p->counters;
#endif
#pragma omp for schedule(dynamic, 1)
        for (int k = 0; k < nz; ++k) {
            contract_level(p->coef, p->M[2], p->F[2], (int64_t)p->M[1] * p->M[0] * nn, zs[k], t.c2);
            double E;
            out[k] = adapt_gk(iai_level2, &t, p->lo[1], p->hi[1], top->atol2, p->rtol, p->maxevals, &E);
        }
        free(t.c2);
        free(t.c1);
    }
}

/* DOS(omega) = int over the box of -Im tr inv((omega + i eta) I - H(k)) / pi dk, d = 3.  abstol is the tolerance of the
 * NESTED quadrature (the caller has divided by |det B| nsyms like do_solve_autobz, src/brillouin.jl:340-342); reltol < 0:
 * the default (0 when abstol > 0, sqrt(eps) otherwise).  Returns the integral; *err, *numevals. */
double orc_iai_dos3(const cd* coef, const int* dims, const int* first, int n, const double* lo, const double* hi, double eta,
                    double omega, double abstol, double reltol, int64_t maxevals, double* err, int64_t* numevals) {
    iai_t p;
    p.coef = coef;
    p.n = n;
    for (int j = 0; j < 3; ++j) {
        p.M[j] = dims[j];
        p.F[j] = first[j];
        p.lo[j] = lo[j];
        p.hi[j] = hi[j];
    }
    p.eta = eta;
    p.omega = omega;
    p.rtol = reltol >= 0.0 ? reltol : (abstol > 0.0 ? 0.0 : sqrt(2.220446049250313e-16));
    p.maxevals = maxevals;
    const int nt = orc_num_threads();
    p.counters = (int64_t*)calloc((size_t)(8 * (nt > 0 ? nt : 1)), sizeof(int64_t));
    iai_top_t top;
    top.p = &p;
    /* abstol / len at every level (src/fourier.jl:466-467,479-480) */
    top.atol2 = abstol / (hi[1] - lo[1]);
    top.atol1 = top.atol2 / (hi[0] - lo[0]);
    double E;
    const double v = adapt_gk(iai_level3, &top, lo[2], hi[2], abstol, p.rtol, maxevals, &E);
    int64_t cnt = 0;
    for (int i = 0; i < (nt > 0 ? nt : 1); ++i) cnt += p.counters[8 * i];
    free(p.counters);
    if (err) *err = E;
    if (numevals) *numevals = cnt;
    return v;
}

/* =============================================================================================
 * GGR: get_ggr_data + sum_ggr on the full PTR grid, d = 3.  ref: src/dos_ggr.jl:1-65,90-104
 * JacobianSeries contracts hierarchically like the series itself: the derivative factor 2 pi i f / t goes on the variable
 * being contracted, so three families live at level 1: plain, d/dk_2, d/dk_3.  Per node:
 * e, U = eigen(Hermitian(h)); v_j = Re diag(U' dH/dk_j U) t_j.  n = 3 (the SVO workload, SMatrix{3,3} in the reference):
 * closed-form eigenvalues, eigenvectors from cross products of rows of H - e I (what StaticArrays' 3x3 Hermitian eigen
 * does), a Jacobi fallback for close pairs; other n: Jacobi with accumulated rotations.
 * Threads over the outermost grid index for the build (the reference threads the rule evaluation the same way,
 * src/fourier.jl:156-161; its eigen loop and sum_ggr are serial Julia code) and over the nodes for the scan.
 * ============================================================================================= */
static void eig_herm_vec(const cd* h, int n, double* e, cd* V) { /* Jacobi with vectors, ascending */
    cd A[32 * 32];
    double norm2 = 0.0;
    for (int b = 0; b < n; ++b)
        for (int a = 0; a <= b; ++a) {
            cd v = (a == b) ? creal(h[a + n * b]) : h[a + n * b];
            A[a + n * b] = v;
            A[b + n * a] = conj(v);
            norm2 += (a == b ? 1.0 : 2.0) * (creal(v) * creal(v) + cimag(v) * cimag(v));
        }
    for (int i = 0; i < n * n; ++i) V[i] = 0.0;
    for (int i = 0; i < n; ++i) V[i + n * i] = 1.0;
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
                    if (r != p && r != q) {
                        const cd x = A[r + n * p], y = A[r + n * q];
                        const cd np_ = c * x - s * g * y, nq_ = s * x + c * g * y;
                        A[r + n * p] = np_;
                        A[r + n * q] = nq_;
                        A[p + n * r] = conj(np_);
                        A[q + n * r] = conj(nq_);
                    }
                    const cd vx = V[r + n * p], vy = V[r + n * q];
                    V[r + n * p] = c * vx - s * g * vy;
                    V[r + n * q] = s * vx + c * g * vy;
                }
            }
    }
    /* sort ascending (selection on columns) */
    double d[32];
    for (int a = 0; a < n; ++a) d[a] = creal(A[a + n * a]);
    for (int i = 0; i < n; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j)
            if (d[j] < d[m]) m = j;
        if (m != i) {
            const double td = d[i];
            d[i] = d[m];
            d[m] = td;
            for (int r = 0; r < n; ++r) {
                const cd tv = V[r + n * i];
                V[r + n * i] = V[r + n * m];
                V[r + n * m] = tv;
            }
        }
        e[i] = d[i];
    }
}

/* 3x3: closed-form eigenvalues, each eigenvector as the largest cross product of two rows of H - e I */
static void eig3_vec(const cd* h, double* e, cd* V) {
    eig3_closed(h, e);
    const double scale = fabs(e[0]) + fabs(e[2]) + 1e-300;
    if (!(e[1] - e[0] > 1e-5 * scale && e[2] - e[1] > 1e-5 * scale)) {
        eig_herm_vec(h, 3, e, V);
        return;
    }
    cd H[9];
    for (int b = 0; b < 3; ++b)
        for (int a = 0; a <= b; ++a) {
            const cd v = (a == b) ? creal(h[a + 3 * b]) : h[a + 3 * b];
            H[a + 3 * b] = v;
            H[b + 3 * a] = conj(v);
        }
    for (int k = 0; k < 3; ++k) {
        cd r[3][3]; /* rows of H - e I */
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) r[a][b] = H[a + 3 * b] - (a == b ? e[k] : 0.0);
        cd best[3] = {0, 0, 0};
        double bn = -1.0;
        for (int a = 0; a < 3; ++a) {
            const int b = (a + 1) % 3;
            /* x with r_a . x = 0 and r_b . x = 0 (bilinear products): x = r_a x r_b */
            cd c0 = r[a][1] * r[b][2] - r[a][2] * r[b][1];
            cd c1 = r[a][2] * r[b][0] - r[a][0] * r[b][2];
            cd c2 = r[a][0] * r[b][1] - r[a][1] * r[b][0];
            const double nn = creal(c0) * creal(c0) + cimag(c0) * cimag(c0) + creal(c1) * creal(c1) + cimag(c1) * cimag(c1) +
                              creal(c2) * creal(c2) + cimag(c2) * cimag(c2);
            if (nn > bn) {
                bn = nn;
                best[0] = c0;
                best[1] = c1;
                best[2] = c2;
            }
        }
        const double inv = 1.0 / sqrt(bn);
        for (int a = 0; a < 3; ++a) V[a + 3 * k] = best[a] * inv;
    }
}

void orc_ggr_data(const cd* coef, const int* dims, const int* first, int n, int npt, const double* period, double* eig, double* vel) {
    const int nn = n * n, M0 = dims[0], M1 = dims[1], M2 = dims[2];
    cd* ph[3];
    for (int j = 0; j < 3; ++j) {
        ph[j] = (cd*)malloc(sizeof(cd) * (size_t)npt * dims[j]);
        for (int i = 0; i < npt; ++i)
            for (int m = 0; m < dims[j]; ++m) {
                const double ang = 2.0 * M_PI * (double)(first[j] + m) * ((double)i / (double)npt);
                ph[j][i * dims[j] + m] = cos(ang) + I * sin(ang);
            }
    }
    const int64_t L2 = (int64_t)M1 * M0 * nn, L1 = (int64_t)M0 * nn;
#pragma omp parallel
    {
        cd* c2 = (cd*)malloc(sizeof(cd) * (size_t)L2 * 2);      /* plain, d3 */
        cd* c1 = (cd*)malloc(sizeof(cd) * (size_t)L1 * 3);      /* plain, d2, d3 */
        cd* Hd = (cd*)malloc(sizeof(cd) * (size_t)nn * 4);      /* H, d1, d2, d3 */
        cd* V = (cd*)malloc(sizeof(cd) * (size_t)nn);
        cd* T = (cd*)malloc(sizeof(cd) * (size_t)n);
#pragma omp for schedule(static)
        for (int i3 = 0; i3 < npt; ++i3) {
            for (int64_t l = 0; l < 2 * L2; ++l) c2[l] = 0.0;
            for (int m = 0; m < M2; ++m) {
                const cd p = ph[2][i3 * M2 + m];
                const cd dp = p * (I * (2.0 * M_PI * (double)(first[2] + m) / period[2]));
                const cd* src = coef + (int64_t)m * L2;
                for (int64_t l = 0; l < L2; ++l) {
                    c2[l] += src[l] * p;
                    c2[L2 + l] += src[l] * dp;
                }
            }
            for (int i2 = 0; i2 < npt; ++i2) {
                for (int64_t l = 0; l < 3 * L1; ++l) c1[l] = 0.0;
                for (int m = 0; m < M1; ++m) {
                    const cd p = ph[1][i2 * M1 + m];
                    const cd dp = p * (I * (2.0 * M_PI * (double)(first[1] + m) / period[1]));
                    const cd* s0 = c2 + (int64_t)m * L1;
                    const cd* s3 = c2 + L2 + (int64_t)m * L1;
                    for (int64_t l = 0; l < L1; ++l) {
                        c1[l] += s0[l] * p;
                        c1[L1 + l] += s0[l] * dp;
                        c1[2 * L1 + l] += s3[l] * p;
                    }
                }
                for (int i1 = 0; i1 < npt; ++i1) {
                    const int64_t k = ((int64_t)i3 * npt + i2) * npt + i1;
                    for (int a = 0; a < 4 * nn; ++a) Hd[a] = 0.0;
                    for (int m = 0; m < M0; ++m) {
                        const cd p = ph[0][i1 * M0 + m];
                        const cd dp = p * (I * (2.0 * M_PI * (double)(first[0] + m) / period[0]));
                        const cd* s0 = c1 + (int64_t)m * nn;
                        const cd* s2 = c1 + L1 + (int64_t)m * nn;
                        const cd* s3 = c1 + 2 * L1 + (int64_t)m * nn;
                        for (int a = 0; a < nn; ++a) {
                            Hd[a] += s0[a] * p;
                            Hd[nn + a] += s0[a] * dp;
                            Hd[2 * nn + a] += s2[a] * p;
                            Hd[3 * nn + a] += s3[a] * p;
                        }
                    }
                    double* e = eig + k * n;
                    if (n == 3)
                        eig3_vec(Hd, e, V);
                    else
                        eig_herm_vec(Hd, n, e, V);
                    for (int j = 0; j < 3; ++j) {
                        const cd* D = Hd + (int64_t)(1 + j) * nn;
                        for (int b = 0; b < n; ++b) {
                            /* Re u' D u */
                            for (int a = 0; a < n; ++a) {
                                cd t = 0.0;
                                for (int c = 0; c < n; ++c) t += D[a + n * c] * V[c + n * b];
                                T[a] = t;
                            }
                            double acc = 0.0;
                            for (int a = 0; a < n; ++a) acc += creal(conj(V[a + n * b]) * T[a]);
                            vel[(k * 3 + j) * n + b] = acc * period[j];
                        }
                    }
                }
            }
        }
        free(c2);
        free(c1);
        free(Hd);
        free(V);
        free(T);
    }
    for (int j = 0; j < 3; ++j) free(ph[j]);
}

static inline double ggr3(double b, double E, double e, double va, double vb, double vc) { /* src/dos_ggr.jl:90-104 */
    double v1 = fabs(va), v2 = fabs(vb), v3 = fabs(vc), t;
    if (v1 < v2) { t = v1; v1 = v2; v2 = t; }
    if (v2 < v3) { t = v2; v2 = v3; v3 = t; }
    if (v1 < v2) { t = v1; v1 = v2; v2 = t; }
    const double dw = fabs(E - e);
    const double w1 = b * fabs(v1 - v2 - v3), w2 = b * (v1 - v2 + v3), w3 = b * (v1 + v2 - v3), w4 = b * (v1 + v2 + v3);
    const double v = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    if (v1 >= v2 + v3 && 0 <= dw && dw <= w1) return 4 * b * b / v1;
    if (v1 <= v2 + v3 && 0 <= dw && dw <= w1) return (2 * b * b * (v1 * v2 + v2 * v3 + v3 * v1) - (dw * dw + (v * b) * (v * b))) / (v1 * v2 * v3);
    if (w1 <= dw && dw <= w2)
        return (b * b * (v1 * v2 + 3 * v2 * v3 + v3 * v1) - b * dw * (-v1 + v2 + v3) - (dw * dw + (v * b) * (v * b)) / 2) / (v1 * v2 * v3);
    if (w2 <= dw && dw <= w3) return 2 * b * (b * (v1 + v2) - dw) / (v1 * v2);
    if (w3 <= dw && dw <= w4) return (b * (v1 + v2 + v3) - dw) * (b * (v1 + v2 + v3) - dw) / (2 * v1 * v2 * v3);
    return 0.0;
}

/* out[iE] = sum_k sum_bands ggr_formula(1 / (2 npt), E, e, v1, v2, v3) (weights one: the full grid), one pass per energy */
void orc_sum_ggr3(int npt, const double* Es, int nE, int64_t nk, int n, const double* eig, const double* vel, double* out) {
    const double b = 1.0 / (2.0 * npt);
    for (int iE = 0; iE < nE; ++iE) {
        const double E = Es[iE];
        double acc = 0.0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
        for (int64_t k = 0; k < nk; ++k) {
            const double* e = eig + k * n;
            const double* v = vel + k * 3 * n;
            double s = 0.0;
            for (int bnd = 0; bnd < n; ++bnd) s += ggr3(b, E, e[bnd], v[bnd], v[n + bnd], v[2 * n + bnd]);
            acc += s;
        }
        out[iE] = acc;
    }
}
