// Per-lane small-matrix complex arithmetic for n <= 4 bands: everything is fully unrolled so that
// the matrices live in VGPRs (runtime-indexed arrays would go to scratch on gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace abz {

template <int N>
struct CMat {
    double re[N][N];  // [row][col]
    double im[N][N];
};

__device__ __forceinline__ void cmul(double ar, double ai, double br, double bi, double& cr, double& ci) {
    cr = ar * br - ai * bi;
    ci = ar * bi + ai * br;
}

// Hermitian eigensolver: cyclic Jacobi on the UPPER triangle of h (like Julia's Hermitian(h),
// src/dos_ggr.jl:19,34).  e ascending.  If VEC, V's columns are the orthonormal eigenvectors.
// Rotation for pivot (p,q), alpha = a_pq = b e^{i phi}, g = e^{-i phi}:
//   t = sgn(d) 2b / (|d| + sqrt(d^2 + 4 b^2)), d = a_qq - a_pp, c = 1/sqrt(1+t^2), s = t c
//   a_rp' = c a_rp - s g a_rq ; a_rq' = s a_rp + c g a_rq ; a_pp -= t b ; a_qq += t b
template <int N, bool VEC>
__device__ __forceinline__ void herm_eig(const CMat<N>& h, double (&e)[N], CMat<N>& V) {
    if constexpr (N == 1) {
        e[0] = h.re[0][0];
        if constexpr (VEC) {
            V.re[0][0] = 1.0;
            V.im[0][0] = 0.0;
        }
        return;
    } else {
        CMat<N> A;
        double norm2 = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
#pragma unroll
            for (int b = 0; b < N; ++b) {
                if (a < b) {
                    A.re[a][b] = h.re[a][b];
                    A.im[a][b] = h.im[a][b];
                    A.re[b][a] = h.re[a][b];
                    A.im[b][a] = -h.im[a][b];
                    norm2 += 2.0 * (h.re[a][b] * h.re[a][b] + h.im[a][b] * h.im[a][b]);
                } else if (a == b) {
                    A.re[a][a] = h.re[a][a];
                    A.im[a][a] = 0.0;
                    norm2 += h.re[a][a] * h.re[a][a];
                }
                if constexpr (VEC) {
                    V.re[a][b] = (a == b) ? 1.0 : 0.0;
                    V.im[a][b] = 0.0;
                }
            }
        }
        const double tiny = 1e-34 * norm2;  // |a_pq| <= 1e-17 ||A||_F counts as zero
        constexpr int MAXSWEEP = (N == 2) ? 1 : 8;
        for (int sweep = 0; sweep < MAXSWEEP; ++sweep) {
            double off2 = 0.0;
#pragma unroll
            for (int p = 0; p < N - 1; ++p) {
#pragma unroll
                for (int q = p + 1; q < N; ++q) off2 += A.re[p][q] * A.re[p][q] + A.im[p][q] * A.im[p][q];
            }
            if (!(off2 > tiny)) break;  // per-lane exit; lanes reconverge after the loop
#pragma unroll
            for (int p = 0; p < N - 1; ++p) {
#pragma unroll
                for (int q = p + 1; q < N; ++q) {
                    const double ar = A.re[p][q], ai = A.im[p][q];
                    const double b2 = ar * ar + ai * ai;
                    if (b2 > tiny) {
                        const double rb = rsqrt(b2);
                        const double b = b2 * rb;
                        const double gr = ar * rb, gi = -ai * rb;  // g = conj(alpha)/b
                        const double d = A.re[q][q] - A.re[p][p];
                        const double t = copysign(2.0 * b, d) / (fabs(d) + sqrt(d * d + 4.0 * b2));
                        const double c = rsqrt(1.0 + t * t);
                        const double s = t * c;
                        const double sgr = s * gr, sgi = s * gi, cgr = c * gr, cgi = c * gi;
                        A.re[p][p] -= t * b;
                        A.re[q][q] += t * b;
                        A.re[p][q] = 0.0;
                        A.im[p][q] = 0.0;
                        A.re[q][p] = 0.0;
                        A.im[q][p] = 0.0;
#pragma unroll
                        for (int r = 0; r < N; ++r) {
                            if (r != p && r != q) {
                                const double xr = A.re[r][p], xi = A.im[r][p];
                                const double yr = A.re[r][q], yi = A.im[r][q];
                                // a_rp' = c x - (s g) y ; a_rq' = s x + (c g) y
                                const double npr = c * xr - (sgr * yr - sgi * yi);
                                const double npi = c * xi - (sgr * yi + sgi * yr);
                                const double nqr = s * xr + (cgr * yr - cgi * yi);
                                const double nqi = s * xi + (cgr * yi + cgi * yr);
                                A.re[r][p] = npr;
                                A.im[r][p] = npi;
                                A.re[r][q] = nqr;
                                A.im[r][q] = nqi;
                                A.re[p][r] = npr;
                                A.im[p][r] = -npi;
                                A.re[q][r] = nqr;
                                A.im[q][r] = -nqi;
                            }
                        }
                        if constexpr (VEC) {
#pragma unroll
                            for (int r = 0; r < N; ++r) {
                                const double xr = V.re[r][p], xi = V.im[r][p];
                                const double yr = V.re[r][q], yi = V.im[r][q];
                                V.re[r][p] = c * xr - (sgr * yr - sgi * yi);
                                V.im[r][p] = c * xi - (sgr * yi + sgi * yr);
                                V.re[r][q] = s * xr + (cgr * yr - cgi * yi);
                                V.im[r][q] = s * xi + (cgr * yi + cgi * yr);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < N; ++a) e[a] = A.re[a][a];
        // ascending sort (bubble network) carrying eigenvector columns
#pragma unroll
        for (int pass = 0; pass < N - 1; ++pass) {
#pragma unroll
            for (int a = 0; a < N - 1 - pass; ++a) {
                const bool sw = e[a] > e[a + 1];
                const double lo = sw ? e[a + 1] : e[a];
                const double hi = sw ? e[a] : e[a + 1];
                e[a] = lo;
                e[a + 1] = hi;
                if constexpr (VEC) {
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        const double x0 = V.re[r][a], x1 = V.re[r][a + 1];
                        const double y0 = V.im[r][a], y1 = V.im[r][a + 1];
                        V.re[r][a] = sw ? x1 : x0;
                        V.re[r][a + 1] = sw ? x0 : x1;
                        V.im[r][a] = sw ? y1 : y0;
                        V.im[r][a + 1] = sw ? y0 : y1;
                    }
                }
            }
        }
    }
}

// Inverse of the general complex matrix A by Gauss-Jordan with partial pivoting; row exchanges are
// done with selects so every index stays a compile-time constant.
template <int N>
__device__ __forceinline__ void cinv(CMat<N>& A, CMat<N>& X) {
#pragma unroll
    for (int a = 0; a < N; ++a) {
#pragma unroll
        for (int b = 0; b < N; ++b) {
            X.re[a][b] = (a == b) ? 1.0 : 0.0;
            X.im[a][b] = 0.0;
        }
    }
#pragma unroll
    for (int c = 0; c < N; ++c) {
        // bring the largest |A[r][c]|, r >= c, to row c
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            const double mc = A.re[c][c] * A.re[c][c] + A.im[c][c] * A.im[c][c];
            const double mr = A.re[r][c] * A.re[r][c] + A.im[r][c] * A.im[r][c];
            const bool sw = mr > mc;
#pragma unroll
            for (int b = 0; b < N; ++b) {
                double t0 = A.re[c][b], t1 = A.re[r][b];
                A.re[c][b] = sw ? t1 : t0;
                A.re[r][b] = sw ? t0 : t1;
                t0 = A.im[c][b];
                t1 = A.im[r][b];
                A.im[c][b] = sw ? t1 : t0;
                A.im[r][b] = sw ? t0 : t1;
                t0 = X.re[c][b];
                t1 = X.re[r][b];
                X.re[c][b] = sw ? t1 : t0;
                X.re[r][b] = sw ? t0 : t1;
                t0 = X.im[c][b];
                t1 = X.im[r][b];
                X.im[c][b] = sw ? t1 : t0;
                X.im[r][b] = sw ? t0 : t1;
            }
        }
        const double pr = A.re[c][c], pi = A.im[c][c];
        const double inv = 1.0 / (pr * pr + pi * pi);
        const double ir = pr * inv, ii = -pi * inv;  // 1/pivot
#pragma unroll
        for (int b = 0; b < N; ++b) {
            double tr, ti;
            cmul(A.re[c][b], A.im[c][b], ir, ii, tr, ti);
            A.re[c][b] = tr;
            A.im[c][b] = ti;
            cmul(X.re[c][b], X.im[c][b], ir, ii, tr, ti);
            X.re[c][b] = tr;
            X.im[c][b] = ti;
        }
#pragma unroll
        for (int r = 0; r < N; ++r) {
            if (r != c) {
                const double fr = A.re[r][c], fi = A.im[r][c];
#pragma unroll
                for (int b = 0; b < N; ++b) {
                    A.re[r][b] -= fr * A.re[c][b] - fi * A.im[c][b];
                    A.im[r][b] -= fr * A.im[c][b] + fi * A.re[c][b];
                    X.re[r][b] -= fr * X.re[c][b] - fi * X.im[c][b];
                    X.im[r][b] -= fr * X.im[c][b] + fi * X.re[c][b];
                }
            }
        }
    }
}

// G = inv((w + i eta) I - H)
template <int N>
__device__ __forceinline__ void gloc(const CMat<N>& H, double w, double eta, CMat<N>& G) {
    if constexpr (N == 1) {
        const double ar = w - H.re[0][0], ai = eta - H.im[0][0];
        const double inv = 1.0 / (ar * ar + ai * ai);
        G.re[0][0] = ar * inv;
        G.im[0][0] = -ai * inv;
    } else {
        CMat<N> A;
#pragma unroll
        for (int a = 0; a < N; ++a) {
#pragma unroll
            for (int b = 0; b < N; ++b) {
                A.re[a][b] = ((a == b) ? w : 0.0) - H.re[a][b];
                A.im[a][b] = ((a == b) ? eta : 0.0) - H.im[a][b];
            }
        }
        cinv<N>(A, G);
    }
}

// wave64 sum via DPP-free shuffles (6 steps)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

}  // namespace abz
