// Fused GGR build for 5...32 bands (ref: src/dos_ggr.jl:14-44 -- `e, U = eigen(Hermitian(h))`, `v_j = Re diag(U' dH/dk_j U) t_j`,
// LAPACK there): ONE kernel per build, row layout (NP = 8 / 16 / 32 lanes per node), only (e, v) reach HBM.
//
// Per node, with lane r of its NP lanes owning row r of every matrix in registers:
//   (1) H(k) from the line's level-1 coefficient set staged in LDS (rows_device.h);
//   (2) Householder tridiagonalisation, the reflectors KEPT: lane r holds component r of every v_K, beta_K lives in lane
//       K + 1, the accumulated phase of the complex subdiagonal in the lane of its row (T_complex = P T_real P^H);
//   (3) eigenvalue b of the real tridiagonal in lane b: Sturm bisection + interpolation (tri_eigval_bisect);
//   (4) its eigenvector IN THE SAME LANE, no cross-lane traffic: inverse iteration on T - lambda_b I with the pivoted LU of
//       LAPACK's dlagtf / dlagts (dstein's scheme: three solves from a lane-dependent start vector, tiny pivots replaced by
//       eps ||T||); members of a cluster (chain of gaps <= 1e-5 ||T||: degenerate levels, where ANY orthonormal basis of the
//       eigenspace is an answer) are perturbed apart by 10 eps like dstein does and re-orthogonalised, lowest first, in
//       wave-uniform rounds that run only when a wave holds a cluster;
//   (5) back-transformation u_b = H_0 ... H_{n-3} P z_b: lane b owns COLUMN b of U.  The reflectors wait in LDS, in the room
//       of the coefficient set (idle between two series evaluations; packed lower triangle per node, written by the lanes
//       that own the components during the Householder steps) and come back by broadcast reads: kept in registers they
//       cost 2 NP doubles per lane beside the 4 NP of the tridiagonal LU -- 1.8 KB of scratch per lane at 16 bands;
//   (6) per direction j: the line's level-1 set of dH/dk_j is staged (variable 1: the same set with the factor 2 pi i f
//       applied on its way into LDS; variables 2, 3: families contracted with the factor on that variable), row r of it
//       evaluated in lane r, its upper triangle parked in the same LDS room, and v_b = u_b^H D u_b = sum_r D_rr |u_r|^2 +
//       2 Re sum_{r<c} conj(u_r) D_rc u_c from broadcast reads -- n^2 / 2 per direction, no matrix reaches HBM;
//   (7) e and v leave through an LDS tile [plane][node] as whole 128-B lines of the rule's planes.
// The round-4 route for these band counts wrote U and every dH/dk_j to HBM (4 KB per node and matrix at 16 bands), found
// the eigenvectors by n dense inverse iterations (n^4) and, above 16 bands, fell back to one wave per node:
// 24^3 nodes: 16 bands 1.54 ms, 17 bands 13.8 ms, 32 bands 29.5 ms.
#include <utility>

#include "abz_internal.h"
#include "rows_device.h"

namespace abz {

namespace {

constexpr double TWO_PI_R = 6.283185307179586476925286766559;

// a wave's LDS writes visible to its other lanes (the rooms are wave-private: no block barrier)
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct GgrRowsArgs {
    const double2* src[3];  // level-1 sets [line][M][n * n]: plain, derivative factor on variable 2, on variable 3
    const double2* tab;     // e^{2 pi i j / npt}
    PlaneView E, V;
    int64_t nlines;
    const int64_t* run_start;  // irregular lists: line l owns nodes [run_start[l], run_start[l + 1]) with grid indices gi
    const int32_t* gi;
    int n, M, first, npt, d;
    int mc;  // coefficients staged at a time (= M when the set fits the LDS whole)
    int fold;  // padded sets with a symmetric frequency range: staged folded (c_0, c_f +- c_f^T), half the series FMAs
    int coef_elems;  // complex numbers of the coefficient room (>= the staged chunk and >= the nodes' scratchpad rooms)
};

// Room of one node's kept reflectors / derivative rows in LDS (complex numbers; + 1: the nodes of a wave start on different banks)
template <int NP>
constexpr int REFL_STRIDE = NP * (NP - 1) / 2 + 1;  // v_K[i], i > K, at K NP - K (K + 1) / 2 + (i - K - 1)
template <int NP>
constexpr int DUP_STRIDE = NP * (NP + 1) / 2 + 1;   // D[R][c], R <= c, at c (c + 1) / 2 + R
template <int NP>
constexpr int PARK_STRIDE = DUP_STRIDE<NP>;

// what the Householder steps leave behind (rows_device.h: hh_step): the reflector components go to `park` (this node's room)
template <int NP>
struct HhKeep {
    double2* park;
    double beta = 0.0;        // lane K + 1: beta of step K
    double phr = 1.0, phi = 0.0;  // lane j: accumulated phase p_j of the subdiagonal (p_0 = 1, p_{K+1} = p_K e_K / |e_K|)
    double cr = 1.0, ci = 0.0;    // the running product (uniform inside the node)
    template <int NPX, int K>
    __device__ __forceinline__ void reflect(int r, double vr, double vi, double b, double x1r, double x1i, double a1sq, double sigma) {
        if (r > K && r < NPX) park[K * NPX - K * (K + 1) / 2 + (r - K - 1)] = make_double2(vr, vi);
        beta = (r == K + 1) ? b : beta;
        // e_K = -(x1 / |x1|) sqrt(sigma); x1 = 0: -sqrt(sigma); sigma = 0: no coupling, phase 1
        double ur = -1.0, ui = 0.0;
        if (a1sq > 0.0) {
            const double inv = a1sq >= 1e-280 ? rsqrt_nr(a1sq) : 1.0 / sqrt(a1sq);
            ur = -x1r * inv;
            ui = -x1i * inv;
        }
        if (!(sigma > 0.0)) {
            ur = 1.0;
            ui = 0.0;
        }
        const double nr = cr * ur - ci * ui, ni = cr * ui + ci * ur;
        cr = nr;
        ci = ni;
        phr = (r == K + 1) ? cr : phr;
        phi = (r == K + 1) ? ci : phi;
    }
    template <int NPX, int K>
    __device__ __forceinline__ void last(int r, double xr, double xi) {
        if constexpr (K + 1 < NPX) {
            const double x1r = group_bcast<NPX, K + 1>(xr), x1i = group_bcast<NPX, K + 1>(xi);
            const double a1sq = x1r * x1r + x1i * x1i;
            double ur = 1.0, ui = 0.0;
            if (a1sq > 0.0) {
                const double inv = a1sq >= 1e-280 ? rsqrt_nr(a1sq) : 1.0 / sqrt(a1sq);
                ur = x1r * inv;
                ui = x1i * inv;
            }
            const double nr = cr * ur - ci * ui, ni = cr * ui + ci * ur;
            cr = nr;
            ci = ni;
            phr = (r == K + 1) ? cr : phr;
            phi = (r == K + 1) ? ci : phi;
        }
    }
};

// ---- eigenvector of the real symmetric tridiagonal (d, |e|^2) for this lane's eigenvalue `lam`: everything in the lane's own
// registers, all indices compile-time.  Scaled to unit Gershgorin radius like the bisection.
template <int NP>
struct TriLU {
    double a[NP], b[NP], dd[NP], c[NP], ia[NP];  // U: diagonal, first and second superdiagonal; multipliers of L; 1 / pivots
    unsigned swapped = 0;                        // bit k: rows k, k + 1 were interchanged
};

// The loops below run over all NP rows without a condition on n: tri_eigvec pads the matrix with a decoupled block (zero
// coupling, diagonal far outside the spectrum) and the start vector with zeros, so rows >= n stay exactly zero.  (Guards
// `k < n` on these straight-line bodies were turned into selects by the compiler anyway: both sides computed, the uniform
// masks kept in dozens of SGPR pairs.)
template <int NP>
__device__ __forceinline__ void tri_factor(const double (&ds)[NP], const double (&off)[NP], double lam, TriLU<NP>& f) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        f.a[k] = ds[k] - lam;
        f.b[k] = off[k];
        f.c[k] = off[k];
        f.dd[k] = 0.0;
    }
    f.swapped = 0;
    double scale1 = fabs(f.a[0]) + fabs(f.b[0]);
#pragma unroll
    for (int k = 0; k + 1 < NP; ++k) {
        const double bk1 = (k + 2 < NP) ? f.b[k + 1] : 0.0;
        const double ak = f.a[k], ak1 = f.a[k + 1], ck = f.c[k], bk = f.b[k];
        const double scale2 = fabs(ck) + fabs(ak1) + fabs(bk1);
        // dlagtf: interchange when |c| / scale2 > |a| / scale1 (real rows: c is never zero, the couplings are floored; the
        // padding: c = 0, no interchange, multiplier 0)
        const bool sw = fabs(ck) * scale1 > fabs(ak) * scale2;
        const double piv = sw ? ck : ak;
        const double ip = rcp_nr(fabs(piv) < 1e-290 ? 1e-290 : piv);
        const double mult = (sw ? ak : ck) * ip;
        // no interchange: a[k+1] -= mult b[k].   interchange: a[k] = c, a[k+1] = b[k] - mult a[k+1], d[k] = b[k+1],
        // b[k+1] = -mult b[k+1], b[k] = old a[k+1]
        f.a[k] = piv;
        f.a[k + 1] = sw ? fma(-mult, ak1, bk) : fma(-mult, bk, ak1);
        f.b[k] = sw ? ak1 : bk;
        f.dd[k] = sw ? bk1 : 0.0;
        if (k + 2 < NP) f.b[k + 1] = sw ? -mult * bk1 : f.b[k + 1];
        f.c[k] = mult;
        f.swapped |= sw ? (1u << k) : 0u;
        scale1 = sw ? scale1 : scale2;
    }
    // reciprocal pivots; a pivot below eps (unit scale) is replaced by +-eps as dlagts does with job = -1
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const double ak = f.a[k];
        const double pk = fabs(ak) < 2.3e-16 ? (ak < 0.0 ? -2.3e-16 : 2.3e-16) : ak;
        f.ia[k] = rcp_nr(pk);
    }
}

// y <- inv(T - lam I) y, then y scaled to unit maximum norm
template <int NP>
__device__ __forceinline__ void tri_solve(const TriLU<NP>& f, double (&y)[NP]) {
#pragma unroll
    for (int k = 0; k + 1 < NP; ++k) {
        const bool sw = (f.swapped >> k) & 1u;
        const double yk = y[k], yk1 = y[k + 1];
        y[k] = sw ? yk1 : yk;
        y[k + 1] = sw ? fma(-f.c[k], yk1, yk) : fma(-f.c[k], yk, yk1);
    }
#pragma unroll
    for (int k = NP - 1; k >= 0; --k) {
        double t = y[k];
        if (k + 1 < NP) t = fma(-f.b[k], y[k + 1 < NP ? k + 1 : k], t);
        if (k + 2 < NP) t = fma(-f.dd[k], y[k + 2 < NP ? k + 2 : k], t);
        y[k] = t * f.ia[k];
    }
    double mx = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) mx = fmax(mx, fabs(y[k]));
    const double s = (mx > 1e-290 && mx < 1e290) ? rcp_nr(mx) : 1.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) y[k] *= s;
}

template <int NP>
__device__ __forceinline__ void unit2(double (&y)[NP]) {
    double nn = 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) nn = fma(y[k], y[k], nn);
    const double s = nn > 1e-290 ? rsqrt_nr(nn) : 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) y[k] *= s;
}

template <int NP>
__device__ __forceinline__ void tri_eigvec(int n, int r, int lane, const double (&d)[NP], const double (&e2)[NP], double lam, double (&z)[NP]) {
    double lo = d[0], hi = d[0], eprev = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < n) {
            const double en = (i + 1 < n) ? sqrt(e2[i]) : 0.0;
            lo = fmin(lo, d[i] - eprev - en);
            hi = fmax(hi, d[i] + eprev + en);
            eprev = en;
        }
    }
    const double span = fmax(fabs(lo), fabs(hi));
    const double sc = span > 0.0 ? 1.0 / span : 1.0;
    double ds[NP], off[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const double y = fmax(e2[i] * sc * sc, 4.9e-32);  // the floor of the bisection: the matrix it found the eigenvalues of
        // rows >= n: a decoupled block with its diagonal at 4 (the spectrum lies in [-1, 1])
        ds[i] = (i < n) ? d[i] * sc : 4.0;
        off[i] = (i + 1 < n) ? y * rsqrt_nr(y) : 0.0;
    }
    const double lams = lam * sc;
    // clusters: lane r is linked to lane r - 1 when their eigenvalues are within 1e-5 of the scale; pos = links below it
    const double lprev = __shfl(lams, lane > 0 ? lane - 1 : 0, 64);
    const bool link = r > 0 && r < n && (lams - lprev) <= 1e-5;
    const unsigned long long links = __builtin_amdgcn_ballot_w64(link);
    const unsigned long long below = (~links) & ((2ull << lane) - 1ull);  // (bit of the node's first lane is always set)
    const int pos = lane - (63 - __builtin_clzll(below));
    TriLU<NP> f;
    tri_factor<NP>(ds, off, lams + 2.3e-15 * (double)pos, f);
    // start vector: lane dependent, no zeros, no symmetry
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const unsigned h = (unsigned)(r * 40503 + i * 30011 + 12345) * 2654435761u;
        z[i] = (i < n) ? (double)((h >> 8) & 0xffffu) * (1.0 / 65536.0) + 0.25 : 0.0;
        z[i] = ((h >> 30) & 1u) ? -z[i] : z[i];
    }
    tri_solve<NP>(f, z);
    tri_solve<NP>(f, z);
    tri_solve<NP>(f, z);
    unit2<NP>(z);
    // cluster members, lowest first: Gram-Schmidt against the members below (final by then), two more solves each
    for (int p = 1; __builtin_amdgcn_ballot_w64(pos >= p) != 0ull; ++p) {  // wave-uniform; not entered without a cluster
        for (int it = 0; it < 3; ++it) {
            double y[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) y[i] = z[i];
            if (it > 0) tri_solve<NP>(f, y);
            for (int t = 1; t <= p; ++t) {  // (the member's vector is fetched twice rather than held: registers)
                double dot = 0.0;
                const int src = lane - t >= 0 ? lane - t : 0;
#pragma unroll
                for (int i = 0; i < NP; ++i) dot = fma(__shfl(z[i], src, 64), y[i], dot);
                dot = (t <= pos) ? dot : 0.0;
#pragma unroll
                for (int i = 0; i < NP; ++i) y[i] = fma(-dot, __shfl(z[i], src, 64), y[i]);
            }
            unit2<NP>(y);
            if (pos == p) {
#pragma unroll
                for (int i = 0; i < NP; ++i) z[i] = y[i];
            }
        }
    }
}

// y = P z (complex), then u = H_0 ... H_{n-3} y.  Lane b works on its own column; v_K comes back from the node's room in LDS
// (the same address for all lanes of a node: broadcast reads).
template <int NP, int K, int... I>
__device__ __forceinline__ void back_step(int n, const HhKeep<NP>& kp, double (&ur)[NP], double (&ui)[NP], std::integer_sequence<int, I...>) {
    if (K + 2 >= n) return;  // uniform: no reflector for this step
    const double beta = group_bcast<NP, K + 1>(kp.beta);
    const double2* __restrict__ vk = kp.park + (K * NP - K * (K + 1) / 2);
    double2 v[NP - K - 1];
    // w = v^H u = sum_i conj(v_i) u_i: two accumulator pairs (even / odd i) halve the dependent chains
    double wr[2] = {0.0, 0.0}, wi[2] = {0.0, 0.0};
    ((void)([&] {
         constexpr int i = K + 1 + I;  // (rows >= n: v = 0 was stored, u = 0)
         v[I] = vk[I];
         wr[I & 1] = fma(v[I].x, ur[i], wr[I & 1]);
         wr[I & 1] = fma(v[I].y, ui[i], wr[I & 1]);
         wi[I & 1] = fma(v[I].x, ui[i], wi[I & 1]);
         wi[I & 1] = fma(-v[I].y, ur[i], wi[I & 1]);
     }()),
     ...);
    const double wwr = (wr[0] + wr[1]) * beta, wwi = (wi[0] + wi[1]) * beta;
    ((void)([&] {
         constexpr int i = K + 1 + I;  // u_i -= w v_i
         ur[i] = fma(-wwr, v[I].x, ur[i]);
         ur[i] = fma(wwi, v[I].y, ur[i]);
         ui[i] = fma(-wwr, v[I].y, ui[i]);
         ui[i] = fma(-wwi, v[I].x, ui[i]);
     }()),
     ...);
}
template <int NP, int... KK>
__device__ __forceinline__ void back_steps(int n, const HhKeep<NP>& kp, double (&ur)[NP], double (&ui)[NP], std::integer_sequence<int, KK...>) {
    // K = NP - 3 - KK: the last reflector first
    (back_step<NP, NP - 3 - KK>(n, kp, ur, ui, std::make_integer_sequence<int, NP - (NP - 3 - KK) - 1>()), ...);
}
template <int NP, int... J>
__device__ __forceinline__ void phase_apply(int n, const HhKeep<NP>& kp, const double (&z)[NP], double (&ur)[NP], double (&ui)[NP],
                                            std::integer_sequence<int, J...>) {
    ((void)([&] {
         const double pr = group_bcast<NP, J>(kp.phr), pi = group_bcast<NP, J>(kp.phi);
         ur[J] = (J < n) ? pr * z[J] : 0.0;
         ui[J] = (J < n) ? pi * z[J] : 0.0;
     }()),
     ...);
}

// eigenvalue b (ascending) and column b of U in lane b, from the rows (ar, ai) of the Hermitian matrix (destroyed); `park`:
// this node's room in LDS
template <int NP>
__device__ __forceinline__ void rows_eigh_columns(int n, int r, int lane, double2* park, double (&ar)[NP], double (&ai)[NP], double& myeig,
                                                  double (&ur)[NP], double (&ui)[NP]) {
    HhKeep<NP> kp;
    kp.park = park;
    double z[NP];
    {
        double e2[NP], d[NP];
        hh_steps_keep<NP>(n, r, ar, ai, e2, kp, std::make_integer_sequence<int, NP>());
        diag_gather<NP>(ar, d, std::make_integer_sequence<int, NP>());
        myeig = tri_eigval_bisect<NP>(n, r, d, e2);
        tri_eigvec<NP>(n, r, lane, d, e2, myeig, z);
    }
    phase_apply<NP>(n, kp, z, ur, ui, std::make_integer_sequence<int, NP>());
    wave_sync_lds();  // the reflectors were written by other lanes of this wave
    if constexpr (NP >= 3) back_steps<NP>(n, kp, ur, ui, std::make_integer_sequence<int, NP - 2>());
}

// u^H D u for this lane's column u, D Hermitian: lane r parks the upper triangle of its row (D[r][c], c >= r) in the node's
// room, every lane reads D[R][c] back (broadcast reads), column c at a time
template <int NP>
__device__ __forceinline__ void park_upper(int n, int r, double2* park, const double (&dr)[NP], const double (&di)[NP]) {
#pragma unroll
    for (int c = 0; c < NP; ++c)
        if (c < n && r <= c) park[c * (c + 1) / 2 + r] = make_double2(dr[c], di[c]);
}
template <int NP, int C, int... RR>
__device__ __forceinline__ void quad_col(int n, const double2* __restrict__ park, const double (&ur)[NP], const double (&ui)[NP], double& diag,
                                         double (&offd)[2], std::integer_sequence<int, RR...>) {
    if (C >= n) return;  // uniform
    const double2* __restrict__ col = park + C * (C + 1) / 2;
    // u_C through an opaque move: the products conj(u_R) u_C do not depend on the direction, and the compiler hoisted all
    // n (n - 1) of them out of the loop over the directions -- through scratch memory (186 stores, 1.5 KB per lane)
    double ucr = ur[C], uci = ui[C];
    asm volatile("" : "+v"(ucr), "+v"(uci));
    diag = fma(col[C].x, fma(ucr, ucr, uci * uci), diag);
    ((void)([&] {
         constexpr int R = RR;  // R < C
         const double2 dv = col[R];
         const double tr = fma(ur[R], ucr, ui[R] * uci);   // conj(u_R) u_C
         const double ti = fma(ur[R], uci, -ui[R] * ucr);
         offd[R & 1] = fma(dv.x, tr, offd[R & 1]);
         offd[R & 1] = fma(-dv.y, ti, offd[R & 1]);
     }()),
     ...);
}
template <int NP, int... CC>
__device__ __forceinline__ double quad_form(int n, const double2* __restrict__ park, const double (&ur)[NP], const double (&ui)[NP],
                                            std::integer_sequence<int, CC...>) {
    double diag = 0.0, offd[2] = {0.0, 0.0};
    (quad_col<NP, CC>(n, park, ur, ui, diag, offd, std::make_integer_sequence<int, CC>()), ...);
    return fma(2.0, offd[0] + offd[1], diag);
}

// stage coefficients [m0, m0 + mcur) of one level-1 set; DERIV: times 2 pi i (first + m) on the way (d/dk_1)
template <int NP, bool PAD>
__device__ __forceinline__ void ggr_stage(double2* coef, const double2* __restrict__ src, int n, int m0, int mcur, int first, bool deriv) {
    const int nn = n * n;
    if constexpr (PAD) {
        for (int t = threadIdx.x; t < mcur * NP * NP; t += blockDim.x) {
            const int m = t / (NP * NP), e = t - m * (NP * NP);
            const int rr = e % NP, j = e / NP;
            double2 c = (rr < n && j < n) ? src[(size_t)(m0 + m) * nn + rr + n * j] : make_double2(0.0, 0.0);
            if (deriv) {
                const double tf = TWO_PI_R * (double)(first + m0 + m);
                c = make_double2(-tf * c.y, tf * c.x);
            }
            coef[t] = c;
        }
    } else {
        for (int t = threadIdx.x; t < mcur * nn; t += blockDim.x) {
            double2 c = src[(size_t)m0 * nn + t];
            if (deriv) {
                const double tf = TWO_PI_R * (double)(first + m0 + t / nn);
                c = make_double2(-tf * c.y, tf * c.x);
            }
            coef[t] = c;
        }
    }
}

// the folded form of a set (panel_stage_fold of rows_device.h, here with the derivative factor of variable 1 and no shift):
// block 0 = -c_0 (zero for d/dk_1), block 2 f - 1 = s_f = c_f + c_f^T, block 2 f = t_f = c_f - c_f^T, c_f times 2 pi i f first
template <int NP>
__device__ __forceinline__ void ggr_stage_fold(double2* coef, const double2* __restrict__ src, int n, int M, bool deriv) {
    const int F = (M - 1) / 2, nn = n * n;
    for (int t = threadIdx.x; t < M * NP * NP; t += blockDim.x) {
        const int b = t / (NP * NP), e = t - b * (NP * NP);
        const int rr = e % NP, j = e / NP;
        double2 v = make_double2(0.0, 0.0);
        if (rr < n && j < n) {
            if (b == 0) {
                const double2 c = src[(size_t)F * nn + rr + n * j];
                v = deriv ? v : make_double2(-c.x, -c.y);
            } else {
                const int f = (b + 1) >> 1;
                double2 c = src[(size_t)(F + f) * nn + rr + n * j], cp = src[(size_t)(F + f) * nn + j + n * rr];
                if (deriv) {
                    const double tf = TWO_PI_R * (double)f;
                    c = make_double2(-tf * c.y, tf * c.x);
                    cp = make_double2(-tf * cp.y, tf * cp.x);
                }
                v = (b & 1) ? make_double2(c.x + cp.x, c.y + cp.y) : make_double2(c.x - cp.x, c.y - cp.y);
            }
        }
        coef[t] = v;
    }
}

template <int NP, bool PAD>
__global__ __launch_bounds__(256, NP <= 16 ? 2 : 1) void ggr_rows_kernel(GgrRowsArgs a) {
    extern __shared__ double2 lds_gr[];
    constexpr int SLOTS = 256 / NP;
    constexpr int TS = SLOTS + 1;
    const int n0 = a.n, nn = n0 * n0, M = a.M, mc = a.mc, d = a.d;
    double2* const coef = lds_gr;  // the staged coefficients; between two series evaluations: the nodes' rooms (PARK_STRIDE each)
    double* const tile = reinterpret_cast<double*>(coef + a.coef_elems);  // [(1 + d) NP][TS]
    const int slot = threadIdx.x / NP, r0 = threadIdx.x % NP, lane = threadIdx.x & 63;
    double2* const park = coef + (size_t)slot * PARK_STRIDE<NP>;
    int fm = a.first % a.npt;
    if (fm < 0) fm += a.npt;
    // work item = one pass (SLOTS nodes) of one line: the room of the coefficient set is reused between the series
    // evaluations, so every pass stages its sets anyway, and items of this size let the hardware's workgroup scheduler balance
    // a small grid (24^3 nodes, 16 bands: 576 lines = 1.1 rounds of workgroups when a workgroup took a whole line)
    const int ppl = (a.npt + SLOTS - 1) / SLOTS;  // passes per line (node lists: runs are at most npt long)
    for (int64_t item = blockIdx.x; item < a.nlines * ppl; item += gridDim.x) {
        const int64_t line = item / ppl;
        const int64_t kbase = a.run_start ? a.run_start[line] : line * a.npt;
        const int count = a.run_start ? (int)(a.run_start[line + 1] - kbase) : a.npt;
        const int i0 = (int)(item - line * ppl) * SLOTS;
        if (i0 < count) {  // (uniform)
            // n and r go through an opaque move once per pass: everything derived from them alone -- dozens of uniform
            // comparisons with n, the start vectors of the inverse iteration -- was hoisted out of both loops and lived (or
            // was spilled: 1.9 KB of scratch per lane) through the whole kernel
            int n = n0, r = r0;
            asm volatile("" : "+s"(n));
            asm volatile("" : "+v"(r));
            // a wave without a node in this pass computes nothing but keeps the block's barriers
            const bool wave_on = i0 + (int)(threadIdx.x >> 6) * (64 / NP) < count;
            const int i1 = i0 + slot;
            const bool act = i1 < count;
            const int ii = act ? i1 : 0;
            const int ic = a.gi ? a.gi[kbase + ii] : ii;
            const double2 z = a.tab[ic];
            const double2 w = a.tab[(int)(((unsigned)fm * (unsigned)ic) % (unsigned)a.npt)];
            double hr[NP], hi[NP];
            // row r of the series of family `fam` (deriv1: d/dk_1 of the plain family); every thread keeps the barriers
            auto series = [&](int fam, bool deriv1) {
                const double2* __restrict__ src = a.src[fam] + line * ((int64_t)M * nn);
                double pr = w.x, pi = w.y;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    hr[j] = 0.0;
                    hi[j] = 0.0;
                }
                bool folded = false;
                if constexpr (PAD) {
                    if (a.fold) {  // (uniform) the whole set, folded: rows of -H from half the FMAs
                        folded = true;
                        __syncthreads();
                        ggr_stage_fold<NP>(coef, src, n, M, deriv1);
                        __syncthreads();
                        if (wave_on) panel_series_row_fold<NP>(coef, M, z.x, z.y, r, hr, hi);
                    }
                }
                for (int m0 = 0; m0 < M && !folded; m0 += mc) {
                    const int mcur = min(mc, M - m0);
                    __syncthreads();
                    ggr_stage<NP, PAD>(coef, src, n, m0, mcur, a.first, deriv1);
                    __syncthreads();
                    if (wave_on) {
                        if constexpr (PAD)  // (the padded set is always staged whole: one chunk)
                            panel_series_row<NP, true>(coef, n, mcur, z.x, z.y, pr, pi, r, hr, hi);
                        else
                            panel_series_row_chunk<NP>(coef, n, mcur, z.x, z.y, pr, pi, r, hr, hi);
                    }
                }
                __syncthreads();  // every wave is done with the set: its room serves as the nodes' scratchpad now
#pragma unroll
                for (int j = 0; j < NP; ++j) {  // the series routines accumulate -H; rows / columns >= n: zero
                    const bool real = r < n && j < n;
                    hr[j] = real ? -hr[j] : 0.0;
                    hi[j] = real ? -hi[j] : 0.0;
                }
            };
            series(0, false);
            double myeig = 0.0, ur[NP], ui[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                ur[j] = 0.0;
                ui[j] = 0.0;
            }
            if (wave_on) rows_eigh_columns<NP>(n, r, lane, park, hr, hi, myeig, ur, ui);
            const bool keep = act && r < n;
            if (keep) tile[r * TS + slot] = myeig;
            for (int j = 0; j < d; ++j) {
                series(j, j == 0);
                if (wave_on) {
                    park_upper<NP>(n, r, park, hr, hi);
                    wave_sync_lds();
                    const double v = quad_form<NP>(n, park, ur, ui, std::make_integer_sequence<int, NP>());
                    if (keep) tile[((1 + j) * NP + r) * TS + slot] = v;
                }
            }
            __syncthreads();
            const int nplanes = (1 + d) * n;
            for (int idx = threadIdx.x; idx < nplanes * SLOTS; idx += 256) {
                const int pl = idx / SLOTS, sl = idx - pl * SLOTS;
                if (i0 + sl >= count) continue;
                const int64_t k = kbase + i0 + sl;
                if (pl < n) {
                    a.E.base[view_off(a.E, k) + (int64_t)pl * a.E.pitch] = tile[pl * TS + sl];
                } else {
                    const int q = pl - n, j = q / n, b = q - j * n;
                    a.V.base[view_off(a.V, k) + (int64_t)(j * n + b) * a.V.pitch] = tile[((1 + j) * NP + b) * TS + sl];
                }
            }
            // (the next pass begins with a barrier before it touches the LDS again)
        }
    }
}

size_t ggr_rows_tile_bytes(int np, int d) { return sizeof(double) * (size_t)((1 + d) * np) * (size_t)(256 / np + 1); }
// the nodes' scratchpad rooms of one workgroup (complex numbers)
size_t ggr_rows_park_elems(int np) { return (size_t)(256 / np) * (size_t)(np * (np + 1) / 2 + 1); }

}  // namespace

// Hermitian series of 5...32 bands, full grids or node lists that come in runs per level-1 set (d >= 2)
bool ggr_rows_supported(int n, int d, int M, int npt, bool herm) {
    if (!herm || n <= 4 || n > 32 || d < 1 || d > 3 || npt < 1 || npt >= 65536 || M < 1) return false;
    const int np = n <= 8 ? 8 : (n <= 16 ? 16 : 32);
    // one coefficient at a time always fits beside the rooms' minimum
    return sizeof(double2) * std::max((size_t)n * n, ggr_rows_park_elems(np)) + ggr_rows_tile_bytes(np, d) <= 150 * 1024;
}

int launch_ggr_rows(abz_ctx* ctx, const GgrRowsSpec& gs) {
    if (gs.nlines <= 0) return ABZ_OK;
    GgrRowsArgs a;
    for (int j = 0; j < 3; ++j) a.src[j] = gs.src[j];
    a.tab = gs.tab;
    a.E = gs.E;
    a.V = gs.V;
    a.nlines = gs.nlines;
    a.run_start = gs.run_start;
    a.gi = gs.gi;
    a.n = gs.n;
    a.M = gs.M;
    a.first = gs.first;
    a.npt = gs.npt;
    a.d = gs.d;
    const int np = gs.n <= 8 ? 8 : (gs.n <= 16 ? 16 : 32);
    const size_t tile = ggr_rows_tile_bytes(np, gs.d), park = ggr_rows_park_elems(np);
    // the zero-padded set whole when two workgroups per CU still fit (<= 72 KB each); otherwise unpadded, as many coefficients
    // at a time as fit (whole up to 150 KB)
    const bool pad = sizeof(double2) * std::max((size_t)gs.M * np * np, park) + tile <= 72 * 1024;
    size_t elems;
    if (pad) {
        a.mc = gs.M;
        elems = (size_t)gs.M * np * np;
    } else {
        const size_t per = (size_t)gs.n * gs.n;
        if (sizeof(double2) * std::max(per * (size_t)gs.M, park) + tile <= 150 * 1024) {
            a.mc = gs.M;
        } else {
            a.mc = (int)((72 * 1024 - tile) / (sizeof(double2) * per));
            if (a.mc < 1) a.mc = (int)((150 * 1024 - tile) / (sizeof(double2) * per));
            if (a.mc < 1) {
                set_error("GGR build: one %d-band coefficient block does not fit the LDS", gs.n);
                return ABZ_ERR_UNSUPPORTED;
            }
        }
        elems = per * (size_t)a.mc;
    }
    a.fold = (pad && (gs.M & 1) && gs.first == -((gs.M - 1) / 2) && abz_switch(SW_EIG_FOLD) != 0) ? 1 : 0;
    elems = std::max(elems, park);
    a.coef_elems = (int)elems;
    const size_t lds = sizeof(double2) * elems + tile;
    const int64_t blocks = std::min<int64_t>(gs.nlines * ((gs.npt + 256 / np - 1) / (256 / np)), 256 * 64);
    ProfScope ps(ctx, ABZ_K_EVAL);
#define ABZ_GR(NPV, PV)                                                                                                              \
    {                                                                                                                                \
        ABZ_HIP(hipFuncSetAttribute((const void*)ggr_rows_kernel<NPV, PV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
        hipLaunchKernelGGL((ggr_rows_kernel<NPV, PV>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);                      \
    }
    if (np == 8 && pad) ABZ_GR(8, true)
    else if (np == 8) ABZ_GR(8, false)
    else if (np == 16 && pad) ABZ_GR(16, true)
    else if (np == 16) ABZ_GR(16, false)
    else if (pad) ABZ_GR(32, true)
    else ABZ_GR(32, false)
#undef ABZ_GR
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

}  // namespace abz
