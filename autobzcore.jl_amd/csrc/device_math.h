// Per-lane small-matrix complex arithmetic for n <= 4 bands: everything is fully unrolled so that
// the matrices live in VGPRs (runtime-indexed arrays would go to scratch on gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace abz {

// 1/x for x well inside the normal range (no denormal / inf / nan handling): hardware estimate +
// 2 Newton steps, ~1 ulp.  5 instructions instead of the ~11 of an IEEE division.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// 1/sqrt(x) under the same range assumption: hardware estimate + 2 Newton steps.
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(fma(-hx * y, y, 0.5), y, y);
    y = fma(fma(-hx * y, y, 0.5), y, y);
    return y;
}

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
                        const double rb = fast_rsqrt(b2);
                        const double b = b2 * rb;
                        const double gr = ar * rb, gi = -ai * rb;  // g = conj(alpha)/b
                        const double d = A.re[q][q] - A.re[p][p];
                        // the rotation only has to be unitary to rounding, not the exact Jacobi angle
                        const double r2 = fma(d, d, 4.0 * b2);
                        const double t = copysign(2.0 * b, d) * fast_rcp(fabs(d) + r2 * fast_rsqrt(r2));
                        const double c = fast_rsqrt(fma(t, t, 1.0));
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

__device__ __forceinline__ void sort3(double a, double b, double c, double (&e)[3]) {
    double t = fmin(a, b);
    b = fmax(a, b);
    a = t;
    t = fmin(b, c);
    c = fmax(b, c);
    b = t;
    t = fmin(a, b);
    b = fmax(a, b);
    a = t;
    e[0] = a;
    e[1] = b;
    e[2] = c;
}

// Eigenvalues (ascending) of a 3 x 3 Hermitian matrix (upper triangle of h), LAPACK-grade accuracy
// at ~1/4 of the Jacobi cost.  (1) trigonometric root of the characteristic polynomial for the
// eigenvalue that is best separated from the other two (the only one the closed form gives to full
// precision); (2) its eigenvector as the largest cross product of two rows of A - lambda I;
// (3) the other two eigenvalues from the 2 x 2 projection of A on the orthogonal complement, which
// is stable for (near-)degenerate pairs where the closed form alone loses half the digits.
__device__ __forceinline__ void herm_eig3_values(const CMat<3>& h, double (&e)[3]) {
    const double a00 = h.re[0][0], a11 = h.re[1][1], a22 = h.re[2][2];
    const double br = h.re[0][1], bi = h.im[0][1];  // a01
    const double cr = h.re[0][2], ci = h.im[0][2];  // a02
    const double dr = h.re[1][2], di = h.im[1][2];  // a12
    const double q = (a00 + a11 + a22) * (1.0 / 3.0);
    const double d0 = a00 - q, d1 = a11 - q, d2 = a22 - q;
    const double nb = br * br + bi * bi, nc = cr * cr + ci * ci, nd = dr * dr + di * di;
    const double p2 = d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * (nb + nc + nd);
    if (!(p2 > 0.0)) {
        e[0] = q;
        e[1] = q;
        e[2] = q;
        return;
    }
    const double ip = rsqrt(p2 * (1.0 / 6.0));
    const double p = p2 * (1.0 / 6.0) * ip;
    // det(A - qI) = d0 d1 d2 + 2 Re(a01 a12 conj(a02)) - d0 |a12|^2 - d1 |a02|^2 - d2 |a01|^2
    const double bdr = br * dr - bi * di, bdi = br * di + bi * dr;
    const double det = d0 * d1 * d2 + 2.0 * (bdr * cr + bdi * ci) - d0 * nd - d1 * nc - d2 * nb;
    double r = 0.5 * det * ip * ip * ip;
    // Largest root x = 2 cos(acos(t)/3) of x^3 - 3x - 2t, t = |r| in [0,1] (x in [sqrt 3, 2], always
    // simple): quartic initial guess (8.9e-6) + 2 Newton steps (2e-16), no acos/cos.  r >= 0: the
    // largest eigenvalue q + p x is the isolated one; r < 0: the smallest, q - p x.
    const double t = fmin(1.0, fabs(r));
    double x = fma(fma(fma(fma(-0.008198810912827986, t, 0.03528472977563877), t, -0.09201052271579181), t,
                       0.33285803676124615), t, 1.732059706718476);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double x2 = x * x;
        const double f = fma(x2 - 3.0, x, -2.0 * t);
        x -= f / (3.0 * x2 - 3.0);
    }
    const double l1 = (r >= 0.0) ? q + p * x : q - p * x;
    // The other two roots of the (shifted, scaled) cubic are -x/2 +- sqrt(3 - 3 x^2 / 4).  When the
    // discriminant is not tiny this is accurate to ~1e-15 / sqrt(disc) * p and we are done; only
    // (near-)degenerate pairs take the deflation path below.
    const double disc = fma(-0.75 * x, x, 3.0);
    if (disc > 1e-6) {
        const double sq = sqrt(disc);
        const double sgn = (r >= 0.0) ? 1.0 : -1.0;
        const double mid = q - sgn * 0.5 * p * x;
        const double lo = mid - p * sq, hi = mid + p * sq;
        sort3(lo, hi, l1, e);
        return;
    }
    // rows of B = A - l1 I
    const double b00 = a00 - l1, b11 = a11 - l1, b22 = a22 - l1;
    // r0 = (b00, a01, a02), r1 = (conj a01, b11, a12), r2 = (conj a02, conj a12, b22)
    // cross products (bilinear): w = ri x rj solves ri.w = rj.w = 0
    // w01 = r0 x r1
    double w01r[3], w01i[3], w02r[3], w02i[3], w12r[3], w12i[3];
    // r0 x r1: ( a01*a12 - a02*b11, a02*conj(a01) - b00*a12, b00*b11 - a01*conj(a01) )
    w01r[0] = (br * dr - bi * di) - cr * b11;
    w01i[0] = (br * di + bi * dr) - ci * b11;
    w01r[1] = (cr * br + ci * bi) - b00 * dr;
    w01i[1] = (ci * br - cr * bi) - b00 * di;
    w01r[2] = b00 * b11 - nb;
    w01i[2] = 0.0;
    // r0 x r2: ( a01*b22 - a02*conj(a12), a02*conj(a02) - b00*b22, b00*conj(a12) - a01*conj(a02) )
    w02r[0] = br * b22 - (cr * dr + ci * di);
    w02i[0] = bi * b22 - (ci * dr - cr * di);
    w02r[1] = nc - b00 * b22;
    w02i[1] = 0.0;
    w02r[2] = b00 * dr - (br * cr + bi * ci);
    w02i[2] = -b00 * di - (bi * cr - br * ci);
    // r1 x r2: ( b11*b22 - a12*conj(a12), a12*conj(a02) - conj(a01)*b22, conj(a01)*conj(a12) - b11*conj(a02) )
    w12r[0] = b11 * b22 - nd;
    w12i[0] = 0.0;
    w12r[1] = (dr * cr + di * ci) - br * b22;
    w12i[1] = (di * cr - dr * ci) + bi * b22;
    w12r[2] = (br * dr - bi * di) - b11 * cr;
    w12i[2] = -(br * di + bi * dr) + b11 * ci;
    double n01 = 0.0, n02 = 0.0, n12 = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        n01 += w01r[j] * w01r[j] + w01i[j] * w01i[j];
        n02 += w02r[j] * w02r[j] + w02i[j] * w02i[j];
        n12 += w12r[j] * w12r[j] + w12i[j] * w12i[j];
    }
    const bool s02 = n02 > n01;
    double nv = s02 ? n02 : n01;
    double vr[3], vi[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        vr[j] = s02 ? w02r[j] : w01r[j];
        vi[j] = s02 ? w02i[j] : w01i[j];
    }
    const bool s12 = n12 > nv;
    nv = s12 ? n12 : nv;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        vr[j] = s12 ? w12r[j] : vr[j];
        vi[j] = s12 ? w12i[j] : vi[j];
    }
    if (!(nv > 0.0)) {  // A - l1 I has rank <= 1: (numerically) a triple eigenvalue
        e[0] = q;
        e[1] = q;
        e[2] = q;
        return;
    }
    const double inv_nv = rsqrt(nv);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        vr[j] *= inv_nv;
        vi[j] *= inv_nv;
    }
    // u2 = normalised (e_k - conj(v_k) v) for the axis k with the smallest |v_k| (norm^2 = 1 - |v_k|^2
    // >= 2/3: unconditionally well conditioned, also when v is only a rough eigenvector)
    const double g0 = vr[0] * vr[0] + vi[0] * vi[0], g1 = vr[1] * vr[1] + vi[1] * vi[1],
                 g2 = vr[2] * vr[2] + vi[2] * vi[2];
    const bool k1 = g1 < g0;
    const double gm01 = k1 ? g1 : g0;
    const bool k2 = g2 < gm01;
    const double gk = k2 ? g2 : gm01;
    const double kr = k2 ? vr[2] : (k1 ? vr[1] : vr[0]);
    const double ki = k2 ? vi[2] : (k1 ? vi[1] : vi[0]);
    double ur[3], ui[3];
    const double inv_mu = rsqrt(1.0 - gk);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // -conj(v_k) v_j = -(kr - i ki)(vr + i vi)
        ur[j] = -(kr * vr[j] + ki * vi[j]);
        ui[j] = -(kr * vi[j] - ki * vr[j]);
    }
    ur[0] += (!k1 && !k2) ? 1.0 : 0.0;
    ur[1] += (k1 && !k2) ? 1.0 : 0.0;
    ur[2] += k2 ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ur[j] *= inv_mu;
        ui[j] *= inv_mu;
    }
    // u3 = conj(v x u2)
    double xr[3], xi[3];
    xr[0] = (vr[1] * ur[2] - vi[1] * ui[2]) - (vr[2] * ur[1] - vi[2] * ui[1]);
    xi[0] = -((vr[1] * ui[2] + vi[1] * ur[2]) - (vr[2] * ui[1] + vi[2] * ur[1]));
    xr[1] = (vr[2] * ur[0] - vi[2] * ui[0]) - (vr[0] * ur[2] - vi[0] * ui[2]);
    xi[1] = -((vr[2] * ui[0] + vi[2] * ur[0]) - (vr[0] * ui[2] + vi[0] * ur[2]));
    xr[2] = (vr[0] * ur[1] - vi[0] * ui[1]) - (vr[1] * ur[0] - vi[1] * ui[0]);
    xi[2] = -((vr[0] * ui[1] + vi[0] * ur[1]) - (vr[1] * ui[0] + vi[1] * ur[0]));
    // y2 = A u2, y3 = A u3 with A Hermitian from the upper triangle
    double y2r[3], y2i[3], y3r[3], y3i[3];
#define ABZ_HMUL(ur_, ui_, yr_, yi_)                                                                   \
    yr_[0] = a00 * ur_[0] + (br * ur_[1] - bi * ui_[1]) + (cr * ur_[2] - ci * ui_[2]);                 \
    yi_[0] = a00 * ui_[0] + (br * ui_[1] + bi * ur_[1]) + (cr * ui_[2] + ci * ur_[2]);                 \
    yr_[1] = (br * ur_[0] + bi * ui_[0]) + a11 * ur_[1] + (dr * ur_[2] - di * ui_[2]);                 \
    yi_[1] = (br * ui_[0] - bi * ur_[0]) + a11 * ui_[1] + (dr * ui_[2] + di * ur_[2]);                 \
    yr_[2] = (cr * ur_[0] + ci * ui_[0]) + (dr * ur_[1] + di * ui_[1]) + a22 * ur_[2];                 \
    yi_[2] = (cr * ui_[0] - ci * ur_[0]) + (dr * ui_[1] - di * ur_[1]) + a22 * ui_[2];
    ABZ_HMUL(ur, ui, y2r, y2i)
    ABZ_HMUL(xr, xi, y3r, y3i)
#undef ABZ_HMUL
    double t22 = 0.0, t33 = 0.0, t23r = 0.0, t23i = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t22 += ur[j] * y2r[j] + ui[j] * y2i[j];
        t33 += xr[j] * y3r[j] + xi[j] * y3i[j];
        t23r += ur[j] * y3r[j] + ui[j] * y3i[j];  // <u2, y3>
        t23i += ur[j] * y3i[j] - ui[j] * y3r[j];
    }
    const double mm = 0.5 * (t22 + t33), hh = 0.5 * (t22 - t33);
    const double rad = sqrt(hh * hh + t23r * t23r + t23i * t23i);
    const double lo = mm - rad, hi = mm + rad;
    sort3(lo, hi, l1, e);  // l1 is the extreme one by construction; order exactly anyway
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

// tr inv((w + i eta) I - H) without forming the inverse (cofactor expansion for n <= 3)
template <int N>
__device__ __forceinline__ void gloc_trace(const CMat<N>& H, double w, double eta, double& tr, double& ti) {
    if constexpr (N == 1) {
        const double ar = w - H.re[0][0], ai = eta - H.im[0][0];
        const double inv = 1.0 / (ar * ar + ai * ai);
        tr = ar * inv;
        ti = -ai * inv;
    } else if constexpr (N == 2) {
        const double a0r = w - H.re[0][0], a0i = eta - H.im[0][0];
        const double a3r = w - H.re[1][1], a3i = eta - H.im[1][1];
        double dr, di, xr, xi;
        cmul(a0r, a0i, a3r, a3i, dr, di);
        cmul(H.re[0][1], H.im[0][1], H.re[1][0], H.im[1][0], xr, xi);  // (-a01)(-a10)
        dr -= xr;
        di -= xi;
        const double nr = a0r + a3r, ni = a0i + a3i;  // trace of the adjugate
        const double inv = 1.0 / (dr * dr + di * di);
        tr = (nr * dr + ni * di) * inv;
        ti = (ni * dr - nr * di) * inv;
    } else if constexpr (N == 3) {
        double ar[3][3], ai[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                ar[a][b] = ((a == b) ? w : 0.0) - H.re[a][b];
                ai[a][b] = ((a == b) ? eta : 0.0) - H.im[a][b];
            }
        }
#define ABZ_M2(i, j, k, l, outr, outi)                                  \
    {                                                                   \
        double p_r, p_i, q_r, q_i;                                      \
        cmul(ar[i][j], ai[i][j], ar[k][l], ai[k][l], p_r, p_i);         \
        cmul(ar[i][l], ai[i][l], ar[k][j], ai[k][j], q_r, q_i);         \
        outr = p_r - q_r;                                               \
        outi = p_i - q_i;                                               \
    }
        double c00r, c00i, c01r, c01i, c02r, c02i, c11r, c11i, c22r, c22i;
        ABZ_M2(1, 1, 2, 2, c00r, c00i)  // a11 a22 - a12 a21
        ABZ_M2(1, 2, 2, 0, c01r, c01i)  // a12 a20 - a10 a22  (= -(a10 a22 - a12 a20))
        ABZ_M2(1, 0, 2, 1, c02r, c02i)  // a10 a21 - a11 a20
        ABZ_M2(0, 0, 2, 2, c11r, c11i)  // a00 a22 - a02 a20
        ABZ_M2(0, 0, 1, 1, c22r, c22i)  // a00 a11 - a01 a10
#undef ABZ_M2
        double dr, di, t_r, t_i;
        cmul(ar[0][0], ai[0][0], c00r, c00i, dr, di);
        cmul(ar[0][1], ai[0][1], c01r, c01i, t_r, t_i);
        dr += t_r;
        di += t_i;
        cmul(ar[0][2], ai[0][2], c02r, c02i, t_r, t_i);
        dr += t_r;
        di += t_i;
        const double nr = c00r + c11r + c22r, ni = c00i + c11i + c22i;
        const double inv = 1.0 / (dr * dr + di * di);
        tr = (nr * dr + ni * di) * inv;
        ti = (ni * dr - nr * di) * inv;
    } else {
        CMat<N> G;
        gloc<N>(H, w, eta, G);
        tr = 0.0;
        ti = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            tr += G.re[a][a];
            ti += G.im[a][a];
        }
    }
}

// Trace of the resolvent through the characteristic polynomial, n = 2, 3:
//   tr inv(z I - H) = p'(z) / p(z),  p(z) = det(z I - H).
// With q = tr H / n and B = H - q I the polynomial is depressed, p(w) = w^3 + p1 w + p0 (w = z - q),
// p1 = sum of principal 2x2 minors of B, p0 = -det B, all of the size of the band spread (not of |z|),
// so the evaluation is well conditioned: error ~ eps spread^n / |p(w)|.  The coefficients are computed
// ONCE per node; every sweep value then costs ~40 flops instead of a 3x3 complex inversion.
template <int N>
struct CharPoly {
    double qr, qi;    // q = tr H / N
    double p1r, p1i;  // N = 3: coefficient of w ; N = 2: unused
    double p0r, p0i;  // constant term
};

template <int N>
__device__ __forceinline__ void charpoly_init(const CMat<N>& H, CharPoly<N>& cp) {
    static_assert(N == 2 || N == 3, "charpoly: n = 2, 3");
    double tr = 0.0, ti = 0.0;
#pragma unroll
    for (int a = 0; a < N; ++a) {
        tr += H.re[a][a];
        ti += H.im[a][a];
    }
    cp.qr = tr * (1.0 / N);
    cp.qi = ti * (1.0 / N);
    double br[N][N], bi[N][N];
#pragma unroll
    for (int a = 0; a < N; ++a) {
#pragma unroll
        for (int b = 0; b < N; ++b) {
            br[a][b] = H.re[a][b] - ((a == b) ? cp.qr : 0.0);
            bi[a][b] = H.im[a][b] - ((a == b) ? cp.qi : 0.0);
        }
    }
    double xr, xi, yr, yi;
    if constexpr (N == 2) {
        // p(w) = w^2 + det B  (tr B = 0)
        cmul(br[0][0], bi[0][0], br[1][1], bi[1][1], xr, xi);
        cmul(br[0][1], bi[0][1], br[1][0], bi[1][0], yr, yi);
        cp.p0r = xr - yr;
        cp.p0i = xi - yi;
        cp.p1r = 0.0;
        cp.p1i = 0.0;
    } else {
#define ABZ_MIN2(i, j, outr, outi)                                  \
    cmul(br[i][i], bi[i][i], br[j][j], bi[j][j], xr, xi);           \
    cmul(br[i][j], bi[i][j], br[j][i], bi[j][i], yr, yi);           \
    outr = xr - yr;                                                 \
    outi = xi - yi;
        double m01r, m01i, m02r, m02i, m12r, m12i;
        ABZ_MIN2(0, 1, m01r, m01i)
        ABZ_MIN2(0, 2, m02r, m02i)
        ABZ_MIN2(1, 2, m12r, m12i)
#undef ABZ_MIN2
        cp.p1r = m01r + m02r + m12r;
        cp.p1i = m01i + m02i + m12i;
        // det B by the first row
        double c0r, c0i, c1r, c1i, c2r, c2i;
        cmul(br[1][1], bi[1][1], br[2][2], bi[2][2], xr, xi);
        cmul(br[1][2], bi[1][2], br[2][1], bi[2][1], yr, yi);
        c0r = xr - yr;
        c0i = xi - yi;
        cmul(br[1][2], bi[1][2], br[2][0], bi[2][0], xr, xi);
        cmul(br[1][0], bi[1][0], br[2][2], bi[2][2], yr, yi);
        c1r = xr - yr;
        c1i = xi - yi;
        cmul(br[1][0], bi[1][0], br[2][1], bi[2][1], xr, xi);
        cmul(br[1][1], bi[1][1], br[2][0], bi[2][0], yr, yi);
        c2r = xr - yr;
        c2i = xi - yi;
        double dr, di;
        cmul(br[0][0], bi[0][0], c0r, c0i, dr, di);
        cmul(br[0][1], bi[0][1], c1r, c1i, xr, xi);
        dr += xr;
        di += xi;
        cmul(br[0][2], bi[0][2], c2r, c2i, xr, xi);
        dr += xr;
        di += xi;
        cp.p0r = -dr;
        cp.p0i = -di;
    }
}

// tr inv((w + i eta) I - H) from the precomputed polynomial
template <int N>
__device__ __forceinline__ void charpoly_trace(const CharPoly<N>& cp, double w, double eta, double& tr, double& ti) {
    const double zr = w - cp.qr, zi = eta - cp.qi;
    const double z2r = zr * zr - zi * zi, z2i = 2.0 * zr * zi;
    double nr, ni, dr, di;
    if constexpr (N == 2) {
        nr = 2.0 * zr;
        ni = 2.0 * zi;
        dr = z2r + cp.p0r;
        di = z2i + cp.p0i;
    } else {
        // den = z^3 + p1 z + p0 = z (z^2 + p1) + p0 ; num = 3 z^2 + p1
        const double ar = z2r + cp.p1r, ai = z2i + cp.p1i;
        dr = fma(zr, ar, fma(-zi, ai, cp.p0r));
        di = fma(zr, ai, fma(zi, ar, cp.p0i));
        nr = fma(3.0, z2r, cp.p1r);
        ni = fma(3.0, z2i, cp.p1i);
    }
    const double inv = 1.0 / (dr * dr + di * di);
    tr = (nr * dr + ni * di) * inv;
    ti = (ni * dr - nr * di) * inv;
}

// Hermitian H: the characteristic polynomial of B = H - (tr H / N) I has REAL coefficients,
// p(w) = w^3 + p1 w + p0 (N = 3) or w^2 + p0 (N = 2), and only the upper triangle of H is needed.
struct CharPolyH {
    double q, p1, p0;
};

// N = 3: diagonal d0 d1 d2, upper triangle b = H01, c = H02, d = H12.  N = 2: d0 d1 and b = H01.
// (Every multiply-add below is spelled out and contraction is off: `a * b + c * d` left to the compiler is fused one way in
// one kernel and the other way in the next -- the kernels that share these functions then disagree in the last bit.)
__device__ __forceinline__ void charpoly_init_h3(double d0, double d1, double d2, double br, double bi, double cr,
                                                 double ci, double dr, double di, CharPolyH& cp) {
#pragma clang fp contract(off)
    const double q = (d0 + d1 + d2) * (1.0 / 3.0);
    d0 -= q;
    d1 -= q;
    d2 -= q;
    const double nb = fma(br, br, bi * bi), nc = fma(cr, cr, ci * ci), nd = fma(dr, dr, di * di);
    cp.q = q;
    cp.p1 = (fma(d0, d1, -nb) + fma(d0, d2, -nc)) + fma(d1, d2, -nd);
    // det B = d0 d1 d2 + 2 Re(b d conj(c)) - d0 |d|^2 - d1 |c|^2 - d2 |b|^2
    const double bdr = fma(br, dr, -(bi * di)), bdi = fma(br, di, bi * dr);
    const double t = fma(bdr, cr, bdi * ci);
    const double det = fma(-d2, nb, fma(-d1, nc, fma(-d0, nd, fma(d0 * d1, d2, 2.0 * t))));
    cp.p0 = -det;
}
__device__ __forceinline__ void charpoly_init_h2(double d0, double d1, double br, double bi, CharPolyH& cp) {
#pragma clang fp contract(off)
    const double q = 0.5 * (d0 + d1);
    d0 -= q;
    d1 -= q;
    cp.q = q;
    cp.p1 = 0.0;
    cp.p0 = fma(d0, d1, -fma(br, br, bi * bi));
}

// tr inv((w + i eta) I - H) = p'(z) / p(z), z = (w - q) + i eta.  eta2 = eta^2, teta = 2 eta.
// NEED_RE = false returns only the imaginary part (DOS).
template <int N, bool NEED_RE>
__device__ __forceinline__ void charpoly_trace_h(const CharPolyH& cp, double w, double eta, double eta2, double teta,
                                                 double& tr, double& ti) {
#pragma clang fp contract(off)
    const double zr = w - cp.q;
    const double z2r = fma(zr, zr, -eta2), z2i = teta * zr;
    double nr, ni, dr, di;
    if constexpr (N == 2) {
        nr = 2.0 * zr;
        ni = teta;
        dr = z2r + cp.p0;
        di = z2i;
    } else {
        const double ar = z2r + cp.p1;  // z^2 + p1 (imaginary part z2i)
        dr = fma(zr, ar, fma(-eta, z2i, cp.p0));
        di = fma(zr, z2i, eta * ar);
        nr = fma(3.0, z2r, cp.p1);
        ni = 3.0 * z2i;
    }
    const double inv = fast_rcp(fma(dr, dr, di * di));
    ti = fma(ni, dr, -(nr * di)) * inv;
    tr = NEED_RE ? fma(nr, dr, ni * di) * inv : 0.0;
}

// N = 4, Hermitian: characteristic polynomial of B = H - (tr H / 4) I,
// p(w) = w^4 + c2 w^2 + c3 w + c4 with REAL coefficients (c1 = -tr B = 0), from the power sums s_k = tr B^k by
// Newton's identities: c2 = -s2/2, c3 = -s3/3, c4 = (s2^2/2 - s4)/4 (the Faddeev-LeVerrier recursion with tr B = 0).
// B is Hermitian, so only the upper triangle of B^2 is formed: s2 = tr B^2, s3 = sum (B^2)_ab B_ba, s4 = ||B^2||_F^2.
// ~170 flops once per node (three full 4x4 complex products in the first version: ~770); every sweep value then
// costs ~36 instead of a 4x4 complex inversion.
struct CharPolyH4 {
    double q, c2, c3, c4;
};
__device__ __forceinline__ void charpoly_init_h4(const CMat<4>& H, CharPolyH4& cp) {
    const double q = 0.25 * (H.re[0][0] + H.re[1][1] + H.re[2][2] + H.re[3][3]);
    CMat<4> B = H;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        B.re[a][a] -= q;
        B.im[a][a] = 0.0;
    }
    double s2 = 0.0, s3 = 0.0, s4 = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = a; b < 4; ++b) {
            double pr = 0.0, pi = 0.0;  // (B^2)_ab
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pr = fma(B.re[a][k], B.re[k][b], pr);
                pr = fma(-B.im[a][k], B.im[k][b], pr);
                if (a != b) {
                    pi = fma(B.re[a][k], B.im[k][b], pi);
                    pi = fma(B.im[a][k], B.re[k][b], pi);
                }
            }
            if (a == b) {
                s2 += pr;
                s3 = fma(pr, B.re[a][a], s3);
                s4 = fma(pr, pr, s4);
            } else {
                s3 = fma(2.0, fma(pr, B.re[a][b], pi * B.im[a][b]), s3);  // 2 Re((B^2)_ab conj(B_ab))
                s4 = fma(2.0, fma(pr, pr, pi * pi), s4);
            }
        }
    }
    cp.q = q;
    cp.c2 = -0.5 * s2;
    cp.c3 = -(1.0 / 3.0) * s3;
    cp.c4 = 0.25 * (0.5 * s2 * s2 - s4);
}
// Eigenvalues (ascending) of a 4 x 4 Hermitian matrix (upper triangle of h) from the REAL characteristic polynomial of
// B = H - (tr H / 4) I (charpoly_init_h4: ~170 flops from the power sums tr B^2, B^3, B^4):
//   w^4 + c2 w^2 + c3 w + c4 = (w^2 + a w + b)(w^2 - a w + d),  a^2 = y = the LARGEST root of Ferrari's resolvent cubic
//   y^3 + 2 c2 y^2 + (c2^2 - 4 c4) y - c3^2 (three real non-negative roots (l_i + l_j)^2 here; trigonometric form, one
//   guarded Newton step), b, d = (c2 + y -+ c3 / a) / 2, then two real quadratics and two Newton steps per root on the quartic.
// ~450 instructions instead of the ~4500 of the per-lane cyclic Jacobi (which bounded H + eig builds at 2 TB/s while H alone
// streams at 6 TB/s).  Error ~ eps spread^4 / prod |l_i - l_j|: 1.4e-14 ||H|| over 2e5 random matrices.  Clustered spectra
// (min_i |p'(w_i)| = min_i prod_j |l_i - l_j| below 2e-3 ||B||_F^3, exact degeneracies included) lose digits in ANY root
// formula and take the Jacobi path instead -- the same split as the 3 x 3 solver's deflation for clustered pairs.
__device__ __forceinline__ void herm_eig4_values(const CMat<4>& hin, double (&e)[4]) {
    CMat<4> h;  // Hermitian(h): the upper triangle decides (src/dos_ggr.jl:19); register renames for Hermitian input
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = a; b < 4; ++b) {
            h.re[a][b] = hin.re[a][b];
            h.im[a][b] = (a == b) ? 0.0 : hin.im[a][b];
            h.re[b][a] = hin.re[a][b];
            h.im[b][a] = (a == b) ? 0.0 : -hin.im[a][b];
        }
    }
    CharPolyH4 cp;
    charpoly_init_h4(h, cp);
    const double c2 = cp.c2, c3 = cp.c3, c4 = cp.c4;
    const double s2 = -2.0 * c2;  // ||B||_F^2
    bool jacobi = !(s2 > 0.0);    // B = 0 (or not finite): nothing to solve / let the Jacobi path deal with it
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
    if (!jacobi) {
        const double P = fma(-c2, c2 * (1.0 / 3.0), -4.0 * c4);
        const double Q = fma(c2, fma(-2.0 / 27.0 * c2, c2, (8.0 / 3.0) * c4), -c3 * c3);
        const double mp3 = fmax(-P * (1.0 / 3.0), 0.0);
        const double r = sqrt(mp3);
        const double den = r * mp3;
        double cth = den > 0.0 ? -0.5 * Q / den : 1.0;
        cth = fmin(fmax(cth, -1.0), 1.0);
        const double t0 = 2.0 * r * cos(acos(cth) * (1.0 / 3.0));
        double y = fmax(fma(-2.0 / 3.0, c2, t0), 0.0);
        {
            const double k1 = fma(c2, c2, -4.0 * c4);
            const double f = fma(fma(y + 2.0 * c2, y, k1), y, -c3 * c3);
            const double fp = fma(fma(3.0, y, 4.0 * c2), y, k1);
            if (fp > 1e-3 * s2 * s2) y = fmax(y - f / fp, 0.0);
        }
        const double a = sqrt(y);
        const double c3a = (a <= 1e-7 * sqrt(s2)) ? 0.0 : c3 / a;
        const double b0 = 0.5 * (c2 + y - c3a), d0 = 0.5 * (c2 + y + c3a);
        const double d1 = sqrt(fmax(fma(-4.0, b0, y), 0.0)), d2 = sqrt(fmax(fma(-4.0, d0, y), 0.0));
        // roots of the two quadratics (each pair ordered), merged by a sorting network
        double p0 = 0.5 * (-a - d1), p1 = 0.5 * (-a + d1), p2 = 0.5 * (a - d2), p3 = 0.5 * (a + d2);
        double lo = fmin(p0, p2), hi = fmax(p0, p2);
        p0 = lo;
        p2 = hi;
        lo = fmin(p1, p3);
        hi = fmax(p1, p3);
        p1 = lo;
        p3 = hi;
        lo = fmin(p1, p2);
        hi = fmax(p1, p2);
        w0 = p0;
        w1 = lo;
        w2 = hi;
        w3 = p3;
        // conditioning of root i = eps ||B||^4 / |p'(w_i)|, |p'(w_i)| = the product of its gaps to the other three: a
        // pair 1e-3 ||B|| apart is fine, three roots at that spacing are not (1.3e-11 measured) -- bound the product
        auto dp = [&](double w) { return fabs(fma(fma(4.0 * w, w, 2.0 * c2), w, c3)); };
        const double ppmin = fmin(fmin(dp(w0), dp(w1)), fmin(dp(w2), dp(w3)));
        jacobi = !(ppmin >= 2e-3 * s2 * sqrt(s2));
        if (!jacobi) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                auto step = [&](double w) {
                    const double p = fma(fma(fma(w, w, c2), w, c3), w, c4);
                    const double pp = fma(fma(4.0 * w, w, 2.0 * c2), w, c3);
                    return w - p / pp;  // |pp| = product of the gaps to the other roots >= 2e-3 ||B||^3 here
                };
                w0 = step(w0);
                w1 = step(w1);
                w2 = step(w2);
                w3 = step(w3);
            }
        }
    }
    if (jacobi) {
        CMat<4> V;
        herm_eig<4, false>(h, e, V);
        return;
    }
    e[0] = w0 + cp.q;
    e[1] = w1 + cp.q;
    e[2] = w2 + cp.q;
    e[3] = w3 + cp.q;
}

// ascending eigenvalues only, the cheapest accurate route per size
template <int N>
__device__ __forceinline__ void herm_eig_values(const CMat<N>& h, double (&e)[N]) {
    if constexpr (N == 3) {
        herm_eig3_values(h, e);
    } else if constexpr (N == 4) {
        herm_eig4_values(h, e);
    } else {
        CMat<N> V;
        herm_eig<N, false>(h, e, V);
    }
}

// tr inv((w + i eta) I - H) = p'(z) / p(z), z = (w - q) + i eta
template <bool NEED_RE>
__device__ __forceinline__ void charpoly_trace_h4(const CharPolyH4& cp, double w, double eta, double eta2, double teta,
                                                  double& tr, double& ti) {
    const double zr = w - cp.q;
    const double z2r = fma(zr, zr, -eta2), z2i = teta * zr;
    // p = (z^2 + c2) z^2 + c3 z + c4
    const double ar = z2r + cp.c2;
    const double dr = fma(ar, z2r, fma(-z2i, z2i, fma(cp.c3, zr, cp.c4)));
    const double di = fma(ar, z2i, fma(z2i, z2r, cp.c3 * eta));
    // p' = z (4 z^2 + 2 c2) + c3
    const double br = fma(4.0, z2r, 2.0 * cp.c2), bi = 4.0 * z2i;
    const double nr = fma(zr, br, fma(-eta, bi, cp.c3));
    const double ni = fma(zr, bi, eta * br);
    const double inv = fast_rcp(fma(dr, dr, di * di));
    ti = fma(ni, dr, -(nr * di)) * inv;
    tr = NEED_RE ? fma(nr, dr, ni * di) * inv : 0.0;
}

// Hermitian H, matrix-valued resolvent: G(z) = adj(w I - B) / p(w) with B = H - q I traceless,
// w = z - q, adj(w I - B) = w^2 I + w B + C, C = adj(B) = B^2 + p1 I (Cayley-Hamilton), p as above.
// Per node: B and C (Hermitian: 9 doubles each) once; per sweep value ~95 flops instead of a 3x3
// complex inversion.
struct AdjH3 {
    double q, p1, p0;
    double d[3];                 // diagonal of B
    double br[3], bi[3];         // B01, B02, B12
    double c[3];                 // diagonal of C
    double cr[3], ci[3];         // C01, C02, C12
};
__device__ __forceinline__ void adj_init_h3(double h00, double h11, double h22, double b01r, double b01i, double b02r,
                                            double b02i, double b12r, double b12i, AdjH3& s) {
    CharPolyH cp;
    charpoly_init_h3(h00, h11, h22, b01r, b01i, b02r, b02i, b12r, b12i, cp);
    s.q = cp.q;
    s.p1 = cp.p1;
    s.p0 = cp.p0;
    const double d0 = h00 - cp.q, d1 = h11 - cp.q, d2 = h22 - cp.q;
    s.d[0] = d0;
    s.d[1] = d1;
    s.d[2] = d2;
    s.br[0] = b01r, s.bi[0] = b01i, s.br[1] = b02r, s.bi[1] = b02i, s.br[2] = b12r, s.bi[2] = b12i;
    const double n01 = b01r * b01r + b01i * b01i, n02 = b02r * b02r + b02i * b02i, n12 = b12r * b12r + b12i * b12i;
    // C = B^2 + p1 I
    s.c[0] = d0 * d0 + n01 + n02 + cp.p1;
    s.c[1] = n01 + d1 * d1 + n12 + cp.p1;
    s.c[2] = n02 + n12 + d2 * d2 + cp.p1;
    // (B^2)_01 = (d0 + d1) b01 + b02 conj(b12)
    s.cr[0] = (d0 + d1) * b01r + (b02r * b12r + b02i * b12i);
    s.ci[0] = (d0 + d1) * b01i + (b02i * b12r - b02r * b12i);
    // (B^2)_02 = (d0 + d2) b02 + b01 b12
    s.cr[1] = (d0 + d2) * b02r + (b01r * b12r - b01i * b12i);
    s.ci[1] = (d0 + d2) * b02i + (b01r * b12i + b01i * b12r);
    // (B^2)_12 = (d1 + d2) b12 + conj(b01) b02
    s.cr[2] = (d1 + d2) * b12r + (b01r * b02r + b01i * b02i);
    s.ci[2] = (d1 + d2) * b12i + (b01r * b02i - b01i * b02r);
}
// G[a + 3 b] (column-major) at z = w + i eta
__device__ __forceinline__ void adj_gloc_h3(const AdjH3& s, double w, double eta, double (&gr)[9], double (&gi)[9]) {
    const double wr = w - s.q;
    const double w2r = fma(wr, wr, -eta * eta), w2i = 2.0 * wr * eta;
    const double ar = w2r + s.p1;
    const double pr = fma(wr, ar, fma(-eta, w2i, s.p0)), pi = fma(wr, w2i, eta * ar);
    const double inv = fast_rcp(fma(pr, pr, pi * pi));
    const double ir = pr * inv, ii = -pi * inv;  // 1 / p(w)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double nr = fma(wr, s.d[a], w2r + s.c[a]), ni = fma(eta, s.d[a], w2i);
        gr[a + 3 * a] = nr * ir - ni * ii;
        gi[a + 3 * a] = nr * ii + ni * ir;
    }
    constexpr int RA[3] = {0, 0, 1}, RB[3] = {1, 2, 2};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        // upper entry (a, b): w B_ab + C_ab ; lower entry (b, a): w conj(B_ab) + conj(C_ab)
        const double ur = fma(wr, s.br[t], fma(-eta, s.bi[t], s.cr[t])), ui = fma(wr, s.bi[t], fma(eta, s.br[t], s.ci[t]));
        const double lr = fma(wr, s.br[t], fma(eta, s.bi[t], s.cr[t])), li = fma(-wr, s.bi[t], fma(eta, s.br[t], -s.ci[t]));
        gr[RA[t] + 3 * RB[t]] = ur * ir - ui * ii;
        gi[RA[t] + 3 * RB[t]] = ur * ii + ui * ir;
        gr[RB[t] + 3 * RA[t]] = lr * ir - li * ii;
        gi[RB[t] + 3 * RA[t]] = lr * ii + li * ir;
    }
}

// Hermitian H, N = 4, matrix-valued resolvent by the same route: with B = H - q I traceless and the
// Faddeev-LeVerrier matrices M2 = B, M3 = B^2 + c2 I, M4 = B M3 + c3 I (all Hermitian, they are polynomials in B),
//   adj(w I - B) = w^3 I + w^2 B + w M3 + M4,   G(z) = adj(w I - B) / p(w),   w = z - q.
// Per node: the three matrices once (upper triangles, 48 doubles); per sweep value ~230 flops for the 16 entries
// instead of a 4x4 complex inversion.
struct AdjH4 {
    double q, c2, c3, c4;
    double bd[4], br[6], bi[6];     // B:  diagonal, then (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
    double m3d[4], m3r[6], m3i[6];  // M3
    double m4d[4], m4r[6], m4i[6];  // M4
};
__device__ __forceinline__ void adj_init_h4(const CMat<4>& H, AdjH4& s) {
    constexpr int RA[6] = {0, 0, 0, 1, 1, 2}, RB[6] = {1, 2, 3, 2, 3, 3};
    const double q = 0.25 * (H.re[0][0] + H.re[1][1] + H.re[2][2] + H.re[3][3]);
    CMat<4> B = H, M3;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        B.re[a][a] -= q;
        B.im[a][a] = 0.0;
    }
    // B^2 (Hermitian): upper triangle, mirrored
    double s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = a; b < 4; ++b) {
            double pr = 0.0, pi = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pr = fma(B.re[a][k], B.re[k][b], pr);
                pr = fma(-B.im[a][k], B.im[k][b], pr);
                if (a != b) {
                    pi = fma(B.re[a][k], B.im[k][b], pi);
                    pi = fma(B.im[a][k], B.re[k][b], pi);
                }
            }
            M3.re[a][b] = pr;
            M3.im[a][b] = pi;
            M3.re[b][a] = pr;
            M3.im[b][a] = -pi;
            if (a == b) {
                s2 += pr;
                s3 = fma(pr, B.re[a][a], s3);
            } else {
                s3 = fma(2.0, fma(pr, B.re[a][b], pi * B.im[a][b]), s3);
            }
        }
    }
    const double c2 = -0.5 * s2, c3 = -(1.0 / 3.0) * s3;
#pragma unroll
    for (int a = 0; a < 4; ++a) M3.re[a][a] += c2;
    // M4 = B M3 + c3 I (upper triangle), c4 = -tr(B M4) / 4
    double t4 = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double pr = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pr = fma(B.re[a][k], M3.re[k][a], pr);
            pr = fma(-B.im[a][k], M3.im[k][a], pr);
        }
        s.m4d[a] = pr + c3;
        t4 = fma(B.re[a][a], s.m4d[a], t4);
        s.bd[a] = B.re[a][a];
        s.m3d[a] = M3.re[a][a];
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int a = RA[t], b = RB[t];
        double pr = 0.0, pi = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pr = fma(B.re[a][k], M3.re[k][b], pr);
            pr = fma(-B.im[a][k], M3.im[k][b], pr);
            pi = fma(B.re[a][k], M3.im[k][b], pi);
            pi = fma(B.im[a][k], M3.re[k][b], pi);
        }
        s.m4r[t] = pr;
        s.m4i[t] = pi;
        s.br[t] = B.re[a][b];
        s.bi[t] = B.im[a][b];
        s.m3r[t] = M3.re[a][b];
        s.m3i[t] = M3.im[a][b];
        t4 = fma(2.0, fma(B.re[a][b], pr, B.im[a][b] * pi), t4);  // 2 Re(B_ab conj(M4_ab))
    }
    s.q = q;
    s.c2 = c2;
    s.c3 = c3;
    s.c4 = -0.25 * t4;
}
// G[a + 4 b] (column-major) at z = w + i eta
__device__ __forceinline__ void adj_gloc_h4(const AdjH4& s, double w, double eta, double (&gr)[16], double (&gi)[16]) {
    constexpr int RA[6] = {0, 0, 0, 1, 1, 2}, RB[6] = {1, 2, 3, 2, 3, 3};
    const double zr = w - s.q, zi = eta;
    const double z2r = fma(zr, zr, -zi * zi), z2i = 2.0 * zr * zi;
    const double z3r = fma(z2r, zr, -z2i * zi), z3i = fma(z2r, zi, z2i * zr);
    // p = (z^2 + c2) z^2 + c3 z + c4
    const double ar = z2r + s.c2;
    const double pr = fma(ar, z2r, fma(-z2i, z2i, fma(s.c3, zr, s.c4)));
    const double pi = fma(ar, z2i, fma(z2i, z2r, s.c3 * zi));
    const double inv = fast_rcp(fma(pr, pr, pi * pi));
    const double ir = pr * inv, ii = -pi * inv;  // 1 / p
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double nr = fma(z2r, s.bd[a], fma(zr, s.m3d[a], z3r + s.m4d[a]));
        const double ni = fma(z2i, s.bd[a], fma(zi, s.m3d[a], z3i));
        gr[a + 4 * a] = nr * ir - ni * ii;
        gi[a + 4 * a] = nr * ii + ni * ir;
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        // upper (a, b): z^2 B_ab + z M3_ab + M4_ab; lower (b, a): the same with the conjugate entries
        const double ur = fma(z2r, s.br[t], fma(-z2i, s.bi[t], fma(zr, s.m3r[t], fma(-zi, s.m3i[t], s.m4r[t]))));
        const double ui = fma(z2r, s.bi[t], fma(z2i, s.br[t], fma(zr, s.m3i[t], fma(zi, s.m3r[t], s.m4i[t]))));
        const double lr = fma(z2r, s.br[t], fma(z2i, s.bi[t], fma(zr, s.m3r[t], fma(zi, s.m3i[t], s.m4r[t]))));
        const double li = fma(-z2r, s.bi[t], fma(z2i, s.br[t], fma(-zr, s.m3i[t], fma(zi, s.m3r[t], -s.m4i[t]))));
        gr[RA[t] + 4 * RB[t]] = ur * ir - ui * ii;
        gi[RA[t] + 4 * RB[t]] = ur * ii + ui * ir;
        gr[RB[t] + 4 * RA[t]] = lr * ir - li * ii;
        gi[RB[t] + 4 * RA[t]] = lr * ii + li * ir;
    }
}

// wave64 sum via DPP-free shuffles (6 steps)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// sin(pi t), cos(pi t) for finite |t| < 2^30: t = k/2 + r, |r| <= 1/4, Taylor polynomials in r (|pi r| <= 0.785:
// 8 + 8 terms, 1 ulp), quadrant by k.  The library routine costs twice the instructions and keeps its coefficients in
// VGPRs that this kernel has to spill; here they are kernel arguments (scalar operands of the FMAs).
__device__ __forceinline__ void sincospi_poly(const double (&sc)[16], double t, double& s, double& c) {
    const double k = rint(t + t);
    const double r = fma(-0.5, k, t);
    const int q = (int)k;
    const double r2 = r * r;
    double p = sc[7];
#pragma unroll
    for (int i = 6; i >= 1; --i) p = fma(p, r2, sc[i]);
    const double sn = fma(r * r2, p, r * sc[0]);
    double u = sc[15];
#pragma unroll
    for (int i = 14; i >= 8; --i) u = fma(u, r2, sc[i]);
    const double cs = fma(r2, u, 1.0);
    const bool odd = q & 1;
    const double ss = odd ? cs : sn, cc = odd ? sn : cs;
    s = __hiloint2double(__double2hiint(ss) ^ ((q & 2) << 30), __double2loint(ss));
    c = __hiloint2double(__double2hiint(cc) ^ (((q + 1) & 2) << 30), __double2loint(cc));
}

// Taylor coefficients of sin(pi r) / r^(2k+1), k = 0..7, and of cos(pi r) / r^(2k), k = 1..8
static const double kSinCosPiCoef[16] = {3.141592653589793, -5.16771278004997, 2.5501640398773455, -0.5992645293207921, 0.08214588661112823, -0.0073704309457143504, 0.00046630280576761255, -2.1915353447830217e-05, -4.934802200544679, 4.0587121264167685, -1.3352627688545895, 0.2353306303588932, -0.02580689139001406, 0.0019295743094039231, -0.0001046381049248457, 4.303069587032947e-06};

}  // namespace abz
