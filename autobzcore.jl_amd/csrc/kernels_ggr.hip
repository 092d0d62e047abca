// Fused GGR build (gfx950 only): eigenvalues and band velocities of every PTR node in ONE kernel.
//
// ref: src/dos_ggr.jl:14-44 (get_ggr_data): per node  h, V = x.s ; e, U = eigen(Hermitian(h)) ;
//      v_j = real(diag(U' V_j U)) * t_j.
//
// The unfused build (api.cpp::rule_fill, rounds 1-2) ran 1 + 3 x (series launch + velocity_kernel) and moved the
// eigenvectors and every dH/dk_j through HBM: ~1.5 KB per node against the 8 n (1 + d) bytes the rule keeps.
// Here a wave evaluates H and the d derivative matrices of its nodes from the same staged coefficients, solves
// the eigenproblem in registers and stores only (e, v): 96 B per node for 3 bands in 3 dimensions.
//
//  * Hermitian series only (c(-R) = c(R)^dagger, symmetric frequency range): H and every dH/dk_j are Hermitian, and
//    the level-1 coefficients of a grid line obey c1[-f] = c1[f]^dagger.  So only frequencies f >= 0 are kept, as the
//    PACKED set of a line (pk_* below): c1[0] (upper triangle), and per f > 0 the combinations
//        dd_a = 2 c1[f]_aa,   s_ab = c1[f]_ab + c1[f]_ba,   t_ab = c1[f]_ab - c1[f]_ba   (a < b)
//    with which  H_ab += (s.x pr - s.y pi) + i (t.x pi + t.y pr),  H_aa += dd.x pr - dd.y pi,  p = z^f:
//    one FMA group serves +f and -f, half the Fourier work (and half the LDS) of the plain sum over 2F + 1 terms.
//  * dH/dk_1 shares the coefficients of H (phase q = 2 pi i f p); dH/dk_j, j >= 2, need sets contracted with the
//    derivative factor on variable j.  FUSE: the wave contracts the packed sets of its lines itself from the packed
//    level-2 sets of its block (LDS), so level-1 sets never exist in HBM and the work loop holds no global load.
//    !FUSE (d = 1, or sets too large for LDS): the d level-1 families are read from HBM and packed on the way in.
//  * Lane mapping: a wave works on TWO grid lines at a time, one per half-wave (32 lanes); a pass covers 32 KPL
//    nodes of each line, KPL = 2 for the body (two nodes share every broadcast ds_read_b128 of a coefficient) and
//    KPL = 1 for a tail of <= 32 nodes.  npt = 150: 2 + 2 + 1 half-passes = 160 lane-slots per line (94 %), where
//    whole-wave passes would need 192 (78 %).
//  * Velocities without eigenvectors (n <= 3): v_b = tr(P_b D) with the spectral projector
//    P_b = prod_{c != b} (B - w_c) / p'(w_b), B = H - (tr H / n) I, so for n = 3
//        v_b = [tr(B^2 D) + w_b tr(B D) + (c2 + w_b^2) tr D] / (3 w_b^2 + c2),   c2 = -tr(B^2) / 2.
//    Conditioning eps ||B||^2 ||D|| / |p'(w_b)|: nodes whose closest pair of bands is nearer than ~2e-3 of the
//    spectrum's scale (high-symmetry points and lines) take Jacobi eigenvectors instead, as do 4 bands.
//  * No scratch: a scratch reload is a vector-memory operation and its s_waitcnt vmcnt drains every store the wave
//    has in flight (measured: the whole store time became serial).  Row bases are scalar, the lane offset is one
//    32-bit register.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "abz_internal.h"
#include "device_math.h"
#include "packed_herm.h"

namespace abz {

namespace {

constexpr double TWO_PI = 6.283185307179586476925286766559;

inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct GgrBuildArgs {
    const double2* src[3];   // !FUSE: level-1 sets, slot stride M n n: [0] plain, [j-1] derivative on variable j
    const double2* src2[2];  // FUSE: level-2 sets, slot stride M2 M n n: [0] plain, [1] derivative on variable 3
    const double2* tab;      // e^{2 pi i j / npt}
    PlaneView E, V;
    int64_t nlines;
    int M, first, npt;
    int M2, first2, gbeg, gcnt, nseg;
    int nt;     // non-temporal stores
    int pitch;  // padded row length: columns npt..pitch-1 are written too (whole 128-B lines)
    // Constants of the 3-band solve that are no inline operands: as kernel arguments they sit in scalar registers.  As
    // literals the compiler kept them in VGPR pairs hoisted to the kernel's entry, across the accumulation loops, and
    // spilled them (a scratch reload in the solve drains the wave's stores).
    double cq[8];
    // node lists (ggr_build_nodes_kernel)
    const int64_t* parents;
    const int32_t* gi;
    int64_t nk;
};

template <bool NT>
__device__ __forceinline__ void st_f64(double* p, double v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// u^dagger D u for Hermitian D given by its upper triangle; u = column b of V
template <int N>
__device__ __forceinline__ double herm_form(const CMat<N>& Dm, const CMat<N>& V, int b) {
    double v = 0.0;
#pragma unroll
    for (int a = 0; a < N; ++a) {
        v = fma(Dm.re[a][a], V.re[a][b] * V.re[a][b] + V.im[a][b] * V.im[a][b], v);
#pragma unroll
        for (int c = a + 1; c < N; ++c) {
            // 2 Re(conj(u_a) D_ac u_c)
            const double tr = Dm.re[a][c] * V.re[c][b] - Dm.im[a][c] * V.im[c][b];
            const double ti = Dm.re[a][c] * V.im[c][b] + Dm.im[a][c] * V.re[c][b];
            v = fma(2.0, V.re[a][b] * tr + V.im[a][b] * ti, v);
        }
    }
    return v;
}

template <int N, int D>
__device__ __forceinline__ void ggr_node_jacobi(const CMat<N> (&A)[D + 1], double (&e)[N], double (&v)[D][N]) {
    CMat<N> V;
    herm_eig<N, true>(A[0], e, V);
#pragma unroll
    for (int j = 0; j < D; ++j) {
#pragma unroll
        for (int b = 0; b < N; ++b) v[j][b] = herm_form<N>(A[j + 1], V, b);
    }
}

// tr(X D) for Hermitian X, D (upper triangles)
#define ABZ_TRHD3(x00, x11, x22, x01r, x01i, x02r, x02i, x12r, x12i, Dm)                                              \
    (x00 * Dm.re[0][0] + x11 * Dm.re[1][1] + x22 * Dm.re[2][2] +                                                      \
     2.0 * ((x01r * Dm.re[0][1] + x01i * Dm.im[0][1]) + (x02r * Dm.re[0][2] + x02i * Dm.im[0][2]) +                   \
            (x12r * Dm.re[1][2] + x12i * Dm.im[1][2])))

// Eigenvalues (ascending) and band velocities of one node from H = A[0] and dH/dk_j = A[j] (upper triangles), in
// straight-line code.  Returns true when the node has to be redone by the eigenvector route (ggr_node_jacobi):
// (nearly) degenerate bands; then e and v hold no result.
template <int N, int D>
__device__ __forceinline__ bool ggr_node_fast(const GgrBuildArgs& a, const CMat<N> (&A)[D + 1], double (&e)[N], double (&v)[D][N]) {
    if constexpr (N == 1) {
        e[0] = A[0].re[0][0];
#pragma unroll
        for (int j = 0; j < D; ++j) v[j][0] = A[j + 1].re[0][0];
        return false;
    } else if constexpr (N == 2) {
        const CMat<2>& H = A[0];
        const double q = 0.5 * (H.re[0][0] + H.re[1][1]);
        const double d0 = 0.5 * (H.re[0][0] - H.re[1][1]);
        const double br = H.re[0][1], bi = H.im[0][1];
        const double r2 = d0 * d0 + br * br + bi * bi;
        const double s = fabs(H.re[0][0]) + fabs(H.re[1][1]);
        const bool fast = r2 > 1e-26 * s * s && r2 > 1e-290;
        const double ir = fast_rsqrt(fast ? r2 : 1.0);
        const double r = r2 * ir;
        e[0] = q - r;
        e[1] = q + r;
        const double hr = 0.5 * ir;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const CMat<2>& Dm = A[j + 1];
            const double t0 = Dm.re[0][0] + Dm.re[1][1];
            const double t1 = d0 * (Dm.re[0][0] - Dm.re[1][1]) + 2.0 * (br * Dm.re[0][1] + bi * Dm.im[0][1]);
            // P_b = (B + w_b) / (2 w_b), w_b = -+ r
            v[j][0] = 0.5 * t0 - t1 * hr;
            v[j][1] = 0.5 * t0 + t1 * hr;
        }
        return !fast;
    } else if constexpr (N == 3) {
        // Roots of w^3 + c2 w + c3 (B = H - q I, c2 = -tr B^2 / 2, c3 = -det B) in the trigonometric form of
        // herm_eig3_values (device_math.h): x = 2 cos(acos(|r|) / 3) by a quartic seed + 2 Newton steps gives the
        // isolated root, the other two follow from the quadratic factor; each root is then polished by one Newton
        // step of the cubic itself, whose 1 / p'(w) the velocities need anyway.  Reciprocals and square roots are
        // hardware estimates + Newton (operands are O(1) after scaling).  A close pair (disc <= 1e-6, i.e. a gap below
        // ~2e-3 of the scale p) or a zero matrix B is computed like any other node and flagged for the redo.
        const CMat<3>& H = A[0];
        const double q = (H.re[0][0] + H.re[1][1] + H.re[2][2]) * a.cq[5];
        const double d0 = H.re[0][0] - q, d1 = H.re[1][1] - q, d2 = H.re[2][2] - q;
        const double br = H.re[0][1], bi = H.im[0][1];  // B01
        const double cr = H.re[0][2], ci = H.im[0][2];  // B02
        const double dr = H.re[1][2], di = H.im[1][2];  // B12
        const double nb = br * br + bi * bi, nc = cr * cr + ci * ci, nd = dr * dr + di * di;
        const double p2 = d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * (nb + nc + nd);
        const double p26 = fmax(p2 * a.cq[6], 1e-290);
        const double ip = fast_rsqrt(p26);
        const double p = p26 * ip;
        // det B = d0 d1 d2 + 2 Re(b d conj(c)) - d0 |d|^2 - d1 |c|^2 - d2 |b|^2
        const double bdr = br * dr - bi * di, bdi = br * di + bi * dr;
        const double det = d0 * d1 * d2 + 2.0 * (bdr * cr + bdi * ci) - d0 * nd - d1 * nc - d2 * nb;
        const double r = 0.5 * det * ip * ip * ip;
        const double t = fmin(1.0, fabs(r));
        double x = fma(fma(fma(fma(a.cq[0], t, a.cq[1]), t, a.cq[2]), t, a.cq[3]), t, a.cq[4]);  // quartic seed, device_math.h
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const double x2 = x * x;
            const double f = fma(x2 - 3.0, x, -2.0 * t);
            x = fma(-f, fast_rcp(fma(3.0, x2, -3.0)), x);
        }
        const double disc = fma(-0.75 * x, x, 3.0);
        const bool fast = disc > a.cq[7] && p2 > 1e-290;
        const double dsafe = fmax(disc, a.cq[7]);
        const double sq = dsafe * fast_rsqrt(dsafe);
        const bool pos = r >= 0.0;
        const double wi = pos ? p * x : -(p * x);  // isolated root: largest for r >= 0, smallest otherwise
        const double wm = -0.5 * wi;               // the pair: wm -+ p sq
        const double wl = wm - p * sq, wh = wm + p * sq;
        double w[3];
        w[0] = pos ? wl : wi;
        w[1] = pos ? wh : wl;
        w[2] = pos ? wi : wh;
        const double c2 = -0.5 * p2;
        const double c3 = -det;
        double rp[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            double pp = fma(3.0 * w[b], w[b], c2);
            pp = fast ? pp : 1.0;
            const double ri = fast_rcp(pp);
            w[b] = fma(-fma(fma(w[b], w[b], c2), w[b], c3), ri, w[b]);  // Newton on the cubic
            pp = fma(3.0 * w[b], w[b], c2);
            rp[b] = fma(fma(-pp, ri, 1.0), ri, ri);
            e[b] = q + w[b];
        }
        // B^2, upper triangle
        const double s00 = d0 * d0 + nb + nc, s11 = nb + d1 * d1 + nd, s22 = nc + nd + d2 * d2;
        const double s01r = br * (d0 + d1) + (cr * dr + ci * di), s01i = bi * (d0 + d1) + (ci * dr - cr * di);  // b (d0+d1) + c conj(d)
        const double s02r = cr * (d0 + d2) + (br * dr - bi * di), s02i = ci * (d0 + d2) + (br * di + bi * dr);  // c (d0+d2) + b d
        const double s12r = dr * (d1 + d2) + (br * cr + bi * ci), s12i = di * (d1 + d2) + (br * ci - bi * cr);  // d (d1+d2) + conj(b) c
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const CMat<3>& Dm = A[j + 1];
            const double t0 = Dm.re[0][0] + Dm.re[1][1] + Dm.re[2][2];
            const double t1 = ABZ_TRHD3(d0, d1, d2, br, bi, cr, ci, dr, di, Dm);
            const double t2 = ABZ_TRHD3(s00, s11, s22, s01r, s01i, s02r, s02i, s12r, s12i, Dm);
#pragma unroll
            for (int b = 0; b < 3; ++b) v[j][b] = fma(fma(w[b], w[b], c2), t0, fma(w[b], t1, t2)) * rp[b];
        }
        return !fast;
    } else {
        return true;  // 4 bands: eigenvectors always
    }
}
#undef ABZ_TRHD3

// H and dH/dk_j (upper triangles) of KPL nodes per lane, i1 = i0 + sub + 32 j, from the line's D packed sets in LDS
// (cset[set * P + e]).
template <int N, int D, int KPL>
__device__ __forceinline__ void ggr_accumulate(const GgrBuildArgs& a, const double2* __restrict__ cset, int F, int P, const double2* tab_l,
                                               int i0, int sub, CMat<N> (&A)[KPL][D + 1]) {
    double zr[KPL], zi[KPL], pr[KPL], pi[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int i1 = i0 + sub + 32 * j;
        const double2 z = tab_l[i1 < a.npt ? i1 : 0];
        zr[j] = z.x;
        zi[j] = z.y;
        pr[j] = 1.0;
        pi[j] = 0.0;
    }
    // frequency 0: the accumulators start from c1[0]
#pragma unroll
    for (int bb = 0; bb < N; ++bb) {
#pragma unroll
        for (int aa = 0; aa <= bb; ++aa) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const double2 c = cset[s * P + Pk<N>::tri(aa, bb)];
                const int mat = s == 0 ? 0 : s + 1;
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    A[j][mat].re[aa][bb] = c.x;
                    A[j][mat].im[aa][bb] = c.y;
                }
            }
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                A[j][1].re[aa][bb] = 0.0;
                A[j][1].im[aa][bb] = 0.0;
            }
        }
    }
    for (int f = 1; f <= F; ++f) {
        const double tf = TWO_PI * (double)f;
        double qr[KPL], qi[KPL];  // 2 pi i f p: phase of d/dx_1
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const double nr = pr[j] * zr[j] - pi[j] * zi[j];
            const double ni = pr[j] * zi[j] + pi[j] * zr[j];
            pr[j] = nr;
            pi[j] = ni;
            qr[j] = -tf * ni;
            qi[j] = tf * nr;
        }
        const double2* __restrict__ cf = cset + Pk<N>::blk(1) + (f - 1) * (N * N);  // block f; element offsets below are those of block 1
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int mat = s == 0 ? 0 : s + 1;
#pragma unroll
            for (int aa = 0; aa < N; ++aa) {
                const double2 dd = cf[s * P + aa];
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    A[j][mat].re[aa][aa] = fma(dd.x, pr[j], A[j][mat].re[aa][aa]);
                    A[j][mat].re[aa][aa] = fma(-dd.y, pi[j], A[j][mat].re[aa][aa]);
                    if (s == 0) {
                        A[j][1].re[aa][aa] = fma(dd.x, qr[j], A[j][1].re[aa][aa]);
                        A[j][1].re[aa][aa] = fma(-dd.y, qi[j], A[j][1].re[aa][aa]);
                    }
                }
            }
#pragma unroll
            for (int bb = 1; bb < N; ++bb) {
#pragma unroll
                for (int aa = 0; aa < bb; ++aa) {
                    const double2 sv = cf[s * P + N + 2 * Pk<N>::pair(aa, bb)];
                    const double2 tv = cf[s * P + N + 2 * Pk<N>::pair(aa, bb) + 1];
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        A[j][mat].re[aa][bb] = fma(sv.x, pr[j], A[j][mat].re[aa][bb]);
                        A[j][mat].re[aa][bb] = fma(-sv.y, pi[j], A[j][mat].re[aa][bb]);
                        A[j][mat].im[aa][bb] = fma(tv.x, pi[j], A[j][mat].im[aa][bb]);
                        A[j][mat].im[aa][bb] = fma(tv.y, pr[j], A[j][mat].im[aa][bb]);
                        if (s == 0) {
                            A[j][1].re[aa][bb] = fma(sv.x, qr[j], A[j][1].re[aa][bb]);
                            A[j][1].re[aa][bb] = fma(-sv.y, qi[j], A[j][1].re[aa][bb]);
                            A[j][1].im[aa][bb] = fma(tv.x, qi[j], A[j][1].im[aa][bb]);
                            A[j][1].im[aa][bb] = fma(tv.y, qr[j], A[j][1].im[aa][bb]);
                        }
                    }
                }
            }
        }
    }
}

template <int N, int D, bool NT>
__device__ __forceinline__ void ggr_store(const GgrBuildArgs& a, double* __restrict__ erow, double* __restrict__ vrow, unsigned u,
                                          const double (&e)[N], const double (&v)[D][N]) {
#pragma unroll
    for (int b = 0; b < N; ++b) st_f64<NT>((erow + (int64_t)b * a.E.pitch) + u, e[b]);
#pragma unroll
    for (int jj = 0; jj < D; ++jj) {
#pragma unroll
        for (int b = 0; b < N; ++b) st_f64<NT>((vrow + (int64_t)(jj * N + b) * a.V.pitch) + u, v[jj][b]);
    }
}

// One unit of work: KPL nodes per lane -- accumulate, then node after node: the straight-line solve, the Jacobi redo of
// flagged nodes (grid nodes on high-symmetry lines), stores.
template <int N, int D, int KPL, bool NT>
__device__ __forceinline__ void ggr_unit(const GgrBuildArgs& a, const double2* __restrict__ cset, int F, int P, const double2* tab_l,
                                         int i0, int sub, int64_t lineA, int half, bool active) {
    CMat<N> A[KPL][D + 1];
    ggr_accumulate<N, D, KPL>(a, cset, F, P, tab_l, i0, sub, A);
    // wave-uniform row bases (scalar registers) + one 32-bit lane offset: the half-wave's own line is half * tile
    // further on.  64-bit per-lane addresses here were hoisted out of the loops and spilled.
    double* __restrict__ erow = a.E.base + lineA * a.E.tile;
    double* __restrict__ vrow = a.V.base + lineA * a.V.tile;
    const unsigned lane_off = (unsigned)sub + (half ? (unsigned)a.E.tile : 0u);  // E and V are views of one tile: same stride
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int i1 = i0 + sub + 32 * j;
        __builtin_amdgcn_sched_barrier(0);  // one node after the other: interleaved solves need both working sets
        if (active && i1 < a.pitch) {
            double e[N], v[D][N];
            if (ggr_node_fast<N, D>(a, A[j], e, v)) ggr_node_jacobi<N, D>(A[j], e, v);
            unsigned u = lane_off;
            asm volatile("" : "+v"(u));
            u += (unsigned)(i0 + 32 * j);
            ggr_store<N, D, NT>(a, erow, vrow, u, e, v);
        }
    }
}

// all passes of a pair of lines whose packed sets are staged in the wave's LDS buffer
template <int N, int D, bool NT>
__device__ __forceinline__ void ggr_line_pair(const GgrBuildArgs& a, const double2* wbuf, int F, int P, const double2* tab_l,
                                              int lane, int64_t lineA, bool haveB) {
    // per-lane values are re-derived from `lane` for every pair of lines (a handful of integer operations): as loop
    // invariants the compiler kept one set of LDS addresses and offsets per unit variant alive across the whole
    // kernel and spilled them
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5, sub = lane & 31;
    const double2* cset = wbuf + (size_t)half * D * P;
    const bool active = half == 0 || haveB;
    if constexpr (N <= 3) {  // two nodes per lane share every broadcast coefficient read (one per lane at 3 waves/SIMD measured the same)
        const int nfull = a.pitch / 64;
        const int rem = a.pitch - 64 * nfull;  // pitch is a multiple of 16: rem in {0, 16, 32, 48}
        for (int p = 0; p < nfull; ++p) ggr_unit<N, D, 2, NT>(a, cset, F, P, tab_l, 64 * p, sub, lineA, half, active);
        if (rem > 32)
            ggr_unit<N, D, 2, NT>(a, cset, F, P, tab_l, 64 * nfull, sub, lineA, half, active);
        else if (rem > 0)
            ggr_unit<N, D, 1, NT>(a, cset, F, P, tab_l, 64 * nfull, sub, lineA, half, active);
    } else {  // 4 bands: 64 accumulator doubles per node, one node per lane
        for (int i0 = 0; i0 < a.pitch; i0 += 32) ggr_unit<N, D, 1, NT>(a, cset, F, P, tab_l, i0, sub, lineA, half, active);
    }
}

constexpr int GGR_MAX_P = 256;  // packed elements per set

// ---- FUSE: block = (parent, segment of the i2 range); packed level-2 sets of the parent in LDS
template <int N, int D, bool NT>
__global__ __launch_bounds__(256, 2) void ggr_build_fused_kernel(GgrBuildArgs a) {
    static_assert(D >= 2, "the fused build contracts variable 2 in the kernel");
    extern __shared__ double2 lds_g[];  // [D-1][M2][P] packed level-2 sets | [npt] phase table | [4 waves][2 lines][D][P]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int F = (a.M - 1) / 2;
    const int P = Pk<N>::size(F);
    const int L2 = a.M2 * P;
    double2* const c2s = lds_g;
    double2* const tab_l = c2s + (size_t)(D - 1) * L2;
    double2* const wbuf = tab_l + a.npt + (size_t)wave * 2 * D * P;
    int fm2 = a.first2 % a.npt;
    if (fm2 < 0) fm2 += a.npt;
    for (int i = threadIdx.x; i < a.npt; i += 256) tab_l[i] = a.tab[i];
    // block = (parent, segment of its line pairs): many more blocks than resident slots, dealt by the dispatcher (one
    // round of resident blocks with equal contiguous shares measured 15 % slower: launch_ggr_build)
    const int ppp = (a.gcnt + 1) / 2;  // pairs per parent
    const int64_t par = blockIdx.x / a.nseg;
    const int seg = blockIdx.x - (int)(par * a.nseg);
    const int64_t g0 = par * ppp + ((int64_t)ppp * seg) / a.nseg;
    const int64_t g1 = par * ppp + ((int64_t)ppp * (seg + 1)) / a.nseg;
    const int nT = (P + 63) / 64;
    // packed level-1 sets of grid indices i2 (line A) and i2 + dB (line B) into dst[line][set][e]: set 0 plain, set 1
    // derivative on variable 2, set 2 from c2s'.  Both lines share every coefficient read.
    auto contract_pair = [&](int i2, int dB, double2* dst) {
        const unsigned ipA0 = (unsigned)(((unsigned)fm2 * (unsigned)i2) % (unsigned)a.npt);
        const unsigned ipB0 = (unsigned)(((unsigned)fm2 * (unsigned)(i2 + dB)) % (unsigned)a.npt);
        for (int t = 0; t < nT; ++t) {
            const int idx = lane + 64 * t;
            const int ii = idx < P ? idx : P - 1;
            double acr[2][D], aci[2][D];
#pragma unroll
            for (int l = 0; l < 2; ++l) {
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    acr[l][s] = 0.0;
                    aci[l][s] = 0.0;
                }
            }
            unsigned ip[2] = {ipA0, ipB0};
            const unsigned step[2] = {(unsigned)i2, (unsigned)(i2 + dB)};
            const double2* __restrict__ row = c2s + ii;
            for (int m2 = 0; m2 < a.M2; ++m2) {
                const double f = TWO_PI * (double)(a.first2 + m2);
                const double2 c = row[(size_t)m2 * P];
                double2 c3 = c;
                if constexpr (D == 3) c3 = row[(size_t)L2 + (size_t)m2 * P];
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    const double2 ph = tab_l[ip[l]];
                    const double dhx = -f * ph.y, dhy = f * ph.x;  // i f ph
                    acr[l][0] = fma(c.x, ph.x, acr[l][0]);
                    acr[l][0] = fma(-c.y, ph.y, acr[l][0]);
                    aci[l][0] = fma(c.x, ph.y, aci[l][0]);
                    aci[l][0] = fma(c.y, ph.x, aci[l][0]);
                    acr[l][1] = fma(c.x, dhx, acr[l][1]);
                    acr[l][1] = fma(-c.y, dhy, acr[l][1]);
                    aci[l][1] = fma(c.x, dhy, aci[l][1]);
                    aci[l][1] = fma(c.y, dhx, aci[l][1]);
                    if constexpr (D == 3) {
                        acr[l][2] = fma(c3.x, ph.x, acr[l][2]);
                        acr[l][2] = fma(-c3.y, ph.y, acr[l][2]);
                        aci[l][2] = fma(c3.x, ph.y, aci[l][2]);
                        aci[l][2] = fma(c3.y, ph.x, aci[l][2]);
                    }
                    ip[l] += step[l];
                    if (ip[l] >= (unsigned)a.npt) ip[l] -= (unsigned)a.npt;
                }
            }
            if (idx < P) {
#pragma unroll
                for (int l = 0; l < 2; ++l) {
#pragma unroll
                    for (int s = 0; s < D; ++s) dst[(l * D + s) * P + idx] = make_double2(acr[l][s], aci[l][s]);
                }
            }
        }
    };
    int64_t g = g0;
    while (g < g1) {
        const int64_t parent = g / ppp;
        const int64_t gend = g1 < (parent + 1) * ppp ? g1 : (parent + 1) * ppp;
        __syncthreads();  // every wave is done with the previous parent's sets (and the phase table is in place)
        {
            const double2* __restrict__ src = a.src2[0] + parent * (int64_t)(D - 1) * L2;  // packed by ggr_pack2_kernel
            for (int i = threadIdx.x; i < (D - 1) * L2; i += 256) c2s[i] = src[i];
        }
        __syncthreads();
        for (int64_t gg = g + ((wave - (int)(g - g0)) & 3); gg < gend; gg += 4) {  // pair j of the share goes to wave j mod 4
            const int p = (int)(gg - parent * ppp);
            const int i2 = a.gbeg + 2 * p;
            const bool haveB = 2 * p + 1 < a.gcnt;
            wave_lds_fence();  // the previous pair's reads of wbuf are done
            contract_pair(i2, haveB ? 1 : 0, wbuf);
            wave_lds_fence();
            const int64_t lineA = parent * a.gcnt + 2 * p;
            ggr_line_pair<N, D, NT>(a, wbuf, F, P, tab_l, lane, lineA, haveB);
        }
        g = gend;
    }
}

// packed level-2 sets for the fused kernel: out[parent][s][m2][e], s < nsrc = D - 1 (plain, derivative on variable 3)
template <int N>
__global__ __launch_bounds__(256) void ggr_pack2_kernel(const double2* __restrict__ src0, const double2* __restrict__ src1, int nsrc,
                                                        int M, int M2, double2* __restrict__ out) {
    const int F = (M - 1) / 2;
    const int P = Pk<N>::size(F);
    const int L2 = M2 * P;
    const int64_t parent = blockIdx.x;
    const int64_t full2 = (int64_t)M2 * M * (N * N);
    for (int s = 0; s < nsrc; ++s) {
        const double2* __restrict__ g = (s == 0 ? src0 : src1) + parent * full2;
        for (int i = threadIdx.x; i < L2; i += 256) {
            const int m2 = i / P, e = i - m2 * P;
            out[(parent * nsrc + s) * L2 + i] = pk_from_full<N>(g + (int64_t)m2 * M * (N * N), F, e);
        }
    }
}

// ---- !FUSE: level-1 families from HBM, packed on their way into LDS; pairs of consecutive lines, grid-strided
template <int N, int D, bool NT>
__global__ __launch_bounds__(256, 2) void ggr_build_lines_kernel(GgrBuildArgs a) {
    extern __shared__ double2 lds_g[];  // [npt] phase table | [4 waves][2 lines][D][P]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int F = (a.M - 1) / 2;
    const int P = Pk<N>::size(F);
    const int MNN = a.M * N * N;
    double2* const tab_l = lds_g;
    double2* const wbuf = tab_l + a.npt + (size_t)wave * 2 * D * P;
    for (int i = threadIdx.x; i < a.npt; i += 256) tab_l[i] = a.tab[i];
    __syncthreads();
    const int64_t npairs = (a.nlines + 1) / 2;
    for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < npairs; p += (int64_t)gridDim.x * 4) {
        const int64_t lineA = 2 * p;
        const bool haveB = lineA + 1 < a.nlines;
        wave_lds_fence();
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int64_t line = (hb == 0 || haveB) ? lineA + hb : lineA;
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const double2* __restrict__ src = a.src[s] + line * MNN;
                double2* dst = wbuf + (size_t)(hb * D + s) * P;
                for (int e = lane; e < P; e += 64) dst[e] = pk_from_full<N>(src, F, e);
            }
        }
        wave_lds_fence();
        ggr_line_pair<N, D, NT>(a, wbuf, F, P, tab_l, lane, lineA, haveB);
    }
}

// ---- irregular node lists (symmetric rules): one lane per node, level-1 sets through the constant address space
// (scalar loads wherever a wave's nodes share their set); +f and -f folded as above.
struct cpod {
    double x, y;
};
typedef const __attribute__((address_space(4))) cpod* cp4_t;
__device__ __forceinline__ cp4_t as_c4(const double2* p) {
    return (cp4_t)(const __attribute__((address_space(1))) cpod*)(const void*)p;
}

template <int N, int D>
__global__ __launch_bounds__(256) void ggr_build_nodes_kernel(GgrBuildArgs a) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.nk) return;
    constexpr int NN = N * N;
    const int MNN = a.M * NN;
    const int F = (a.M - 1) / 2;
    const int64_t slot = a.parents ? a.parents[k] : 0;
    const double2 z = a.tab[a.gi[k]];
    cp4_t cs[3] = {as_c4(a.src[0] + slot * MNN), as_c4(a.src[D >= 2 ? 1 : 0] + slot * MNN), as_c4(a.src[D >= 3 ? 2 : 0] + slot * MNN)};
    CMat<N> A[D + 1];
#pragma unroll
    for (int bb = 0; bb < N; ++bb) {
#pragma unroll
        for (int aa = 0; aa <= bb; ++aa) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const int mat = s == 0 ? 0 : s + 1;
                A[mat].re[aa][bb] = cs[s][F * NN + aa + N * bb].x;
                A[mat].im[aa][bb] = cs[s][F * NN + aa + N * bb].y;
            }
            A[1].re[aa][bb] = 0.0;
            A[1].im[aa][bb] = 0.0;
        }
    }
    double pr = 1.0, pi = 0.0;
    for (int f = 1; f <= F; ++f) {
        const double nr = pr * z.x - pi * z.y, ni = pr * z.y + pi * z.x;
        pr = nr;
        pi = ni;
        const double tf = TWO_PI * (double)f;
        const double qr = -tf * pi, qi = tf * pr;
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int mat = s == 0 ? 0 : s + 1;
            cp4_t cf = cs[s] + (F + f) * NN;
#pragma unroll
            for (int aa = 0; aa < N; ++aa) {
                const double dx = 2.0 * cf[aa + N * aa].x, dy = 2.0 * cf[aa + N * aa].y;
                A[mat].re[aa][aa] = fma(dx, pr, A[mat].re[aa][aa]);
                A[mat].re[aa][aa] = fma(-dy, pi, A[mat].re[aa][aa]);
                if (s == 0) {
                    A[1].re[aa][aa] = fma(dx, qr, A[1].re[aa][aa]);
                    A[1].re[aa][aa] = fma(-dy, qi, A[1].re[aa][aa]);
                }
            }
#pragma unroll
            for (int bb = 1; bb < N; ++bb) {
#pragma unroll
                for (int aa = 0; aa < bb; ++aa) {
                    const double ux = cf[aa + N * bb].x, uy = cf[aa + N * bb].y;
                    const double vx = cf[bb + N * aa].x, vy = cf[bb + N * aa].y;
                    const double sx = ux + vx, sy = uy + vy, tx = ux - vx, ty = uy - vy;
                    A[mat].re[aa][bb] = fma(sx, pr, A[mat].re[aa][bb]);
                    A[mat].re[aa][bb] = fma(-sy, pi, A[mat].re[aa][bb]);
                    A[mat].im[aa][bb] = fma(tx, pi, A[mat].im[aa][bb]);
                    A[mat].im[aa][bb] = fma(ty, pr, A[mat].im[aa][bb]);
                    if (s == 0) {
                        A[1].re[aa][bb] = fma(sx, qr, A[1].re[aa][bb]);
                        A[1].re[aa][bb] = fma(-sy, qi, A[1].re[aa][bb]);
                        A[1].im[aa][bb] = fma(tx, qi, A[1].im[aa][bb]);
                        A[1].im[aa][bb] = fma(ty, qr, A[1].im[aa][bb]);
                    }
                }
            }
        }
    }
    double e[N], v[D][N];
    if (ggr_node_fast<N, D>(a, A, e, v)) ggr_node_jacobi<N, D>(A, e, v);
    const int ll = a.E.line_len;
    const int64_t line = k / ll;
    const unsigned u = (unsigned)(k - line * ll);
    double* __restrict__ erow = a.E.base + line * a.E.tile;
    double* __restrict__ vrow = a.V.base + line * a.V.tile;
#pragma unroll
    for (int b = 0; b < N; ++b) (erow + (int64_t)b * a.E.pitch)[u] = e[b];
#pragma unroll
    for (int jj = 0; jj < D; ++jj) {
#pragma unroll
        for (int b = 0; b < N; ++b) (vrow + (int64_t)(jj * N + b) * a.V.pitch)[u] = v[jj][b];
    }
}

}  // namespace

// rows of full coefficients [M][n n] -> packed rows (packed_herm.h); linear, so it commutes with every contraction
template <int N>
__global__ __launch_bounds__(256) void pack_rows_kernel(const double2* __restrict__ src, int64_t nrows, int M, double2* __restrict__ out) {
    const int F = (M - 1) / 2;
    const int P = Pk<N>::size(F);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nrows * P) return;
    const int64_t row = t / P;
    const int e = (int)(t - row * P);
    out[t] = pk_from_full<N>(src + row * ((int64_t)M * N * N), F, e);
}

size_t packed_row_elems(int n, int M) { return (size_t)(n * (n + 1) / 2 + ((M - 1) / 2) * n * n); }

int launch_pack_rows(abz_ctx* ctx, int n, int M, const double2* src, int64_t nrows, double2* out) {
    const int64_t tot = nrows * (int64_t)packed_row_elems(n, M);
    if (tot == 0) return ABZ_OK;
    const unsigned blocks = (unsigned)cdiv64(tot, 256);
    switch (n) {
        case 1: hipLaunchKernelGGL(pack_rows_kernel<1>, dim3(blocks), dim3(256), 0, ctx->stream, src, nrows, M, out); break;
        case 2: hipLaunchKernelGGL(pack_rows_kernel<2>, dim3(blocks), dim3(256), 0, ctx->stream, src, nrows, M, out); break;
        case 3: hipLaunchKernelGGL(pack_rows_kernel<3>, dim3(blocks), dim3(256), 0, ctx->stream, src, nrows, M, out); break;
        case 4: hipLaunchKernelGGL(pack_rows_kernel<4>, dim3(blocks), dim3(256), 0, ctx->stream, src, nrows, M, out); break;
        default: set_error("packed rows exist for n <= 4"); return ABZ_ERR_UNSUPPORTED;
    }
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// LDS bytes of the kernels
static size_t ggr_packed(int n, int M) { return (size_t)(n * (n + 1) / 2 + ((M - 1) / 2) * n * n); }
static size_t ggr_fused_lds(int n, int d, int M, int M2, int npt) {
    const size_t P = ggr_packed(n, M);
    return sizeof(double2) * ((size_t)(d - 1) * M2 * P + (size_t)npt + 4 * 2 * (size_t)d * P);
}
static size_t ggr_lines_lds(int n, int d, int M, int npt) {
    return sizeof(double2) * ((size_t)npt + 4 * 2 * (size_t)d * ggr_packed(n, M));
}

bool ggr_build_supported(int n, int d, int M, int npt, bool herm) {
    const bool off = !abz_switch(SW_GGR_FUSED);  // per call: tests compare both builds
    // Hermitian series have an odd number of symmetric frequencies per variable (detect_hermitian, api.cpp)
    if (off || !herm || (M & 1) == 0 || n < 1 || n > 4 || d < 1 || d > 3 || npt >= 65536) return false;
    return ggr_packed(n, M) <= (size_t)GGR_MAX_P && ggr_lines_lds(n, d, M, npt) <= 64 * 1024;
}

size_t ggr_build_pack2_elems(int n, int d, int M, int M2, int64_t nparents) {
    return (size_t)nparents * (size_t)(d - 1) * (size_t)M2 * ggr_packed(n, M);
}

bool ggr_build_can_fuse(int n, int d, int M, int M2, int npt) {
    const bool off = !abz_switch(SW_GGR_FUSE2);  // per call
    // two blocks per CU must fit the 160 KB of LDS
    return !off && d >= 2 && ggr_fused_lds(n, d, M, M2, npt) <= 78 * 1024;
}

int launch_ggr_build(abz_ctx* ctx, const GgrBuildSpec& gs) {
    GgrBuildArgs a;
    for (int j = 0; j < 3; ++j) a.src[j] = gs.src[j];
    a.src2[0] = gs.src2[0];
    a.src2[1] = gs.src2[1];
    a.tab = gs.tab;
    a.E = gs.E;
    a.V = gs.V;
    a.nlines = gs.nlines;
    a.M = gs.M;
    a.first = gs.first;
    a.npt = gs.npt;
    a.M2 = gs.M2;
    a.first2 = gs.first2;
    a.gbeg = gs.gbeg;
    a.gcnt = gs.gcnt;
    a.nseg = 1;
    a.pitch = gs.E.row;
    a.parents = gs.parents;
    a.gi = gs.gi;
    a.nk = gs.nk;
    const int n = gs.n, d = gs.d;
    {
        const double cq[8] = {-0.008198810912827986, 0.03528472977563877, -0.09201052271579181, 0.33285803676124615,
                              1.732059706718476, 1.0 / 3.0, 1.0 / 6.0, 1e-6};
        for (int i = 0; i < 8; ++i) a.cq[i] = cq[i];
    }
    ProfScope ps(ctx, ABZ_K_GGRBUILD);
    if (!gs.grid) {
        if (gs.nk == 0) return ABZ_OK;
        const unsigned blocks = (unsigned)cdiv64(gs.nk, 256);
#define GN(NN, DD) hipLaunchKernelGGL((ggr_build_nodes_kernel<NN, DD>), dim3(blocks), dim3(256), 0, ctx->stream, a)
#define GND(NN)                    \
    switch (d) {                   \
        case 1: GN(NN, 1); break;  \
        case 2: GN(NN, 2); break;  \
        default: GN(NN, 3); break; \
    }
        switch (n) {
            case 1: GND(1) break;
            case 2: GND(2) break;
            case 3: GND(3) break;
            default: GND(4) break;
        }
#undef GND
#undef GN
        ABZ_HIP(hipGetLastError());
        return ABZ_OK;
    }
    if (gs.nlines == 0) return ABZ_OK;
    {
        const double bytes = 8.0 * (double)(n * (1 + d)) * (double)gs.E.row * (double)gs.nlines;
        a.nt = bytes > 256.0 * 1024 * 1024 ? 1 : 0;
    }
    if (gs.fuse) {
        const size_t lds = ggr_fused_lds(n, d, gs.M, gs.M2, gs.npt);
        const int64_t nparents = gs.nlines / std::max(gs.gcnt, 1);
        // the level-2 sets, packed once per parent (the blocks then copy them into LDS as they are)
        {
            const unsigned pb = (unsigned)nparents;
            switch (n) {
                case 1: hipLaunchKernelGGL(ggr_pack2_kernel<1>, dim3(pb), dim3(256), 0, ctx->stream, gs.src2[0], gs.src2[1], d - 1, gs.M, gs.M2, gs.pack2); break;
                case 2: hipLaunchKernelGGL(ggr_pack2_kernel<2>, dim3(pb), dim3(256), 0, ctx->stream, gs.src2[0], gs.src2[1], d - 1, gs.M, gs.M2, gs.pack2); break;
                case 3: hipLaunchKernelGGL(ggr_pack2_kernel<3>, dim3(pb), dim3(256), 0, ctx->stream, gs.src2[0], gs.src2[1], d - 1, gs.M, gs.M2, gs.pack2); break;
                default: hipLaunchKernelGGL(ggr_pack2_kernel<4>, dim3(pb), dim3(256), 0, ctx->stream, gs.src2[0], gs.src2[1], d - 1, gs.M, gs.M2, gs.pack2); break;
            }
            ABZ_HIP(hipGetLastError());
            a.src2[0] = gs.pack2;
            a.src2[1] = nullptr;
        }
        // block = (parent, segment of ~8 of its line pairs), dealt by the dispatcher: a single round of resident blocks
        // with equal contiguous shares was 15 % slower (the lines of the k_3 = 0, 1/2 planes all take the Jacobi redo and
        // one static share held 22 of them)
        const int ppp = (gs.gcnt + 1) / 2;
        a.nseg = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(ppp, 8), ppp));
        const unsigned blocks = (unsigned)(nparents * a.nseg);
        // above 64 KB of dynamic LDS a launch is rejected unless the function was told so (every other launch of that
        // size in the library does the same); a 4-band d = 3 model with M = M2 = 11 at npt = 150 needs 68.6 KB
#define GF1(NN, DD, NT)                                                                                                   \
    do {                                                                                                                  \
        auto kfn = ggr_build_fused_kernel<NN, DD, NT>;                                                                    \
        if (lds > 64 * 1024)                                                                                              \
            ABZ_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
        hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), lds, ctx->stream, a);                                            \
    } while (0)
#define GF(NN, DD)        \
    if (a.nt)             \
        GF1(NN, DD, true);  \
    else                  \
        GF1(NN, DD, false)
#define GFD(NN)              \
    if (d == 2) {            \
        GF(NN, 2);           \
    } else {                 \
        GF(NN, 3);           \
    }
        switch (n) {
            case 1: GFD(1) break;
            case 2: GFD(2) break;
            case 3: GFD(3) break;
            default: GFD(4) break;
        }
#undef GFD
#undef GF
#undef GF1
    } else {
        const size_t lds = ggr_lines_lds(n, d, gs.M, gs.npt);
        const int64_t npairs = (gs.nlines + 1) / 2;
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv64(npairs, 4), 256 * 12);
#define GL(NN, DD)                                                                                                      \
    if (a.nt)                                                                                                           \
        hipLaunchKernelGGL((ggr_build_lines_kernel<NN, DD, true>), dim3(blocks), dim3(256), lds, ctx->stream, a);      \
    else                                                                                                                \
        hipLaunchKernelGGL((ggr_build_lines_kernel<NN, DD, false>), dim3(blocks), dim3(256), lds, ctx->stream, a)
#define GLD(NN)                    \
    switch (d) {                   \
        case 1: GL(NN, 1); break;  \
        case 2: GL(NN, 2); break;  \
        default: GL(NN, 3); break; \
    }
        switch (n) {
            case 1: GLD(1) break;
            case 2: GLD(2) break;
            case 3: GLD(3) break;
            default: GLD(4) break;
        }
#undef GLD
#undef GL
    }
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}


// ------------------------------------------------------------------------------------------
// GGR scan: sum_k w_k sum_bands ggr_formula(b, E, e, v...) for a list of energies.  ref: src/dos_ggr.jl:58-104
// ------------------------------------------------------------------------------------------
namespace {

// ref: src/dos_ggr.jl:75-104 (same branch order)
__device__ __forceinline__ double ggr1(double b, double E, double e, double v1) {
    v1 = fabs(v1);
    const double dw = fabs(E - e);
    return (dw <= b * v1) ? 1.0 / v1 : 0.0;
}
__device__ __forceinline__ double ggr2(double b, double E, double e, double va, double vb) {
    va = fabs(va);
    vb = fabs(vb);
    const double v1 = fmax(va, vb), v2 = fmin(va, vb);
    const double dw = fabs(E - e);
    const double w1 = b * fabs(v1 - v2), w3 = b * (v1 + v2);
    if (dw <= w1) return 2.0 * b / v1;
    if (dw <= w3) return (b * (v1 + v2) - dw) / (v1 * v2);
    return 0.0;
}
__device__ __forceinline__ double ggr3(double b, double E, double e, double va, double vb, double vc) {
    va = fabs(va);
    vb = fabs(vb);
    vc = fabs(vc);
    const double v1 = fmax(va, fmax(vb, vc));
    const double v3 = fmin(va, fmin(vb, vc));
    const double v2 = (va + vb + vc) - v1 - v3;
    const double dw = fabs(E - e);
    const double w1 = b * fabs(v1 - v2 - v3);
    const double w2 = b * (v1 - v2 + v3);
    const double w3 = b * (v1 + v2 - v3);
    const double w4 = b * (v1 + v2 + v3);
    const double vn2 = v1 * v1 + v2 * v2 + v3 * v3;
    const double p = v1 * v2 * v3;
    if (v1 >= v2 + v3 && dw <= w1) return 4.0 * b * b / v1;
    if (v1 <= v2 + v3 && dw <= w1) return (2.0 * b * b * (v1 * v2 + v2 * v3 + v3 * v1) - (dw * dw + vn2 * b * b)) / p;
    if (w1 <= dw && dw <= w2)
        return (b * b * (v1 * v2 + 3.0 * v2 * v3 + v3 * v1) - b * dw * (-v1 + v2 + v3) - (dw * dw + vn2 * b * b) * 0.5) / p;
    if (w2 <= dw && dw <= w3) return 2.0 * b * (b * (v1 + v2) - dw) / (v1 * v2);
    if (w3 <= dw && dw <= w4) {
        const double t = b * (v1 + v2 + v3) - dw;
        return t * t / (2.0 * p);
    }
    return 0.0;
}

template <int D>
__device__ __forceinline__ double ggr_formula(double b, double E, double e, const double (&v)[D]) {
    if constexpr (D == 1)
        return ggr1(b, E, e, v[0]);
    else if constexpr (D == 2)
        return ggr2(b, E, e, v[0], v[1]);
    else
        return ggr3(b, E, e, v[0], v[1], v[2]);
}

__device__ __forceinline__ int64_t plane_off(const PlaneView& v, int64_t k) {
    const int64_t line = k / v.line_len;
    return line * v.tile + (k - line * v.line_len);
}

struct GgrArgs {
    PlaneView E, V;
    const double* w;
    const double* Es;  // device; the windowed kernel needs them ascending
    int64_t nk;
    int n, d, nE;
    int vstride;  // planes between the velocity components of a band (= number of bands of the rule)
    double b;
    double inv_step = 0.0;  // > 0: the energies are equispaced (a linspace sweep): Es[i] = Es[0] + i / inv_step to rounding
};

// The 3-d formula with everything that does not depend on the energy taken out of the energy loop (ref: src/dos_ggr.jl:90-104,
// same regions in the same order): per (node, band) the sorted |v|, the four break points w1 <= w2 <= w3 <= w4 and, per
// region, the coefficients of f(dw) = A + B dw + C dw^2 with the three reciprocals (1/v1, 1/(v1 v2), 1/(v1 v2 v3)) formed
// ONCE; an energy then costs |E - e|, four compares, the selects and two FMAs instead of the sort, ten products and an
// IEEE division.  The last region keeps its squared form t^2 / (2 p), t = w4 - dw (no cancellation at the edge of the
// window).  Plain local variables in the kernel below: as members of a struct they went to scratch (104 B) and the scan
// took twice as long.
// Every formula is zero outside |E - e| <= b (|v_1| + ... + |v_d|) (the last branch of each ggr_formula method), a
// window of ~2 b |v| around the band energy: with 256 energies over the band width a (node, band) pair meets one
// or two of them.  So a thread finds the first energy of its window in the ASCENDING list (binary search in LDS),
// evaluates the formula only on the energies inside and adds w_k f into its wave's histogram (LDS f64 atomics; a
// wave's own adds come in program order, so sums do not depend on scheduling); the histograms of a block are
// summed in a fixed order.  The all-pairs kernel below evaluated n nE formulas per node (SVO 150^3 x 256 energies:
// 2.9 ms + a 3.2 ms serial reduction of 13 184 partial rows); this one reads the rule once.
// N bands per thread.  n <= 4: N = n, one block row.  n > 4: N = 1 and blockIdx.y is the band.
template <int N, int D>
__global__ __launch_bounds__(256) void ggr_window_kernel(GgrArgs a, double* __restrict__ partial, int64_t nrows) {
    extern __shared__ double ldsw[];  // [nE] energies | [4 waves][nE] histograms
    const int wave = threadIdx.x >> 6;
    double* const Esl = ldsw;
    double* const hist = ldsw + (size_t)(1 + wave) * a.nE;
    for (int i = threadIdx.x; i < a.nE; i += 256) Esl[i] = a.Es[i];
    for (int i = threadIdx.x; i < 4 * a.nE; i += 256) ldsw[a.nE + i] = 0.0;
    __syncthreads();
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < a.nk; k += (int64_t)gridDim.x * 256) {
        const double wk = a.w ? a.w[k] : 1.0;
        const double* __restrict__ ei = a.E.base + plane_off(a.E, k) + (int64_t)blockIdx.y * a.E.pitch;
        const double* __restrict__ vi = a.V.base + plane_off(a.V, k) + (int64_t)blockIdx.y * a.V.pitch;
        double e[N], v[N][D];
#pragma unroll
        for (int bnd = 0; bnd < N; ++bnd) {
            e[bnd] = ei[(int64_t)bnd * a.E.pitch];
#pragma unroll
            for (int j = 0; j < D; ++j) v[bnd][j] = vi[(int64_t)(j * a.vstride + bnd) * a.V.pitch];
        }
#pragma unroll
        for (int bnd = 0; bnd < N; ++bnd) {
            double top = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) top += fabs(v[bnd][j]);
            top *= a.b;
            // a little wider than the formula's own test, which then decides exactly
            const double slack = 8.0 * 2.220446049250313e-16 * (fabs(e[bnd]) + top);
            const double lo = e[bnd] - top - slack, hi = e[bnd] + top + slack;
            int i0 = 0;  // first energy >= lo
            if (a.inv_step > 0.0) {
                // equispaced sweep: the index by arithmetic, made exact against the list itself (two LDS reads instead of
                // the log2(nE) dependent ones of the search)
                const double g = (lo - Esl[0]) * a.inv_step;
                i0 = g <= 0.0 ? 0 : (g >= (double)a.nE ? a.nE : (int)g);
                while (i0 > 0 && Esl[i0 - 1] >= lo) --i0;
                while (i0 < a.nE && Esl[i0] < lo) ++i0;
            } else {
                int len = a.nE;
                while (len > 0) {
                    const int half = len >> 1;
                    const bool right = Esl[i0 + half] < lo;
                    i0 = right ? i0 + half + 1 : i0;
                    len = right ? len - half - 1 : half;
                }
            }
            if constexpr (D == 3) {
                if (i0 < a.nE && Esl[i0] <= hi) {  // most windows of a coarse sweep hold no energy at all
                    const double b = a.b, b2 = b * b;
                    const double va = fabs(v[bnd][0]), vb = fabs(v[bnd][1]), vc = fabs(v[bnd][2]);
                    const double v1 = fmax(va, fmax(vb, vc)), v3 = fmin(va, fmin(vb, vc));
                    const double v2 = (va + vb + vc) - v1 - v3;
                    const double w1 = b * fabs(v1 - v2 - v3), w2 = b * (v1 - v2 + v3), w3 = b * (v1 + v2 - v3), w4 = b * (v1 + v2 + v3);
                    const double vn2 = v1 * v1 + v2 * v2 + v3 * v3, v12 = v1 * v2;
                    const double ip = 1.0 / (v12 * v3), i12 = 1.0 / v12;
                    const double s2 = v12 + v2 * v3 + v3 * v1;
                    const bool thin = v1 >= v2 + v3;
                    const double A1 = thin ? 4.0 * b2 / v1 : b2 * (2.0 * s2 - vn2) * ip, C1 = thin ? 0.0 : -ip;
                    const double A3 = b2 * ((s2 + 2.0 * v2 * v3) - 0.5 * vn2) * ip, B3 = -b * (-v1 + v2 + v3) * ip, C3 = -0.5 * ip;
                    const double A4 = 2.0 * b2 * (v1 + v2) * i12, B4 = -2.0 * b * i12, h5 = 0.5 * ip;
                    for (int i = i0; i < a.nE; ++i) {
                        const double En = Esl[i];
                        if (!(En <= hi)) break;
                        const double dw = fabs(En - e[bnd]);
                        const bool r1 = dw <= w1, r3 = dw <= w2, r4 = dw <= w3, r5 = dw <= w4;
                        const double A = r1 ? A1 : (r3 ? A3 : A4);
                        const double B = r1 ? 0.0 : (r3 ? B3 : B4);
                        const double C = r1 ? C1 : (r3 ? C3 : 0.0);
                        const double t = w4 - dw;
                        const double fp = fma(fma(C, dw, B), dw, A);
                        const double f = r4 ? fp : (r5 ? t * t * h5 : 0.0);
                        if (f != 0.0) __hip_atomic_fetch_add(hist + i, wk * f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            } else {
                for (int i = i0; i < a.nE; ++i) {
                    const double En = Esl[i];
                    if (!(En <= hi)) break;
                    const double f = ggr_formula<D>(a.b, En, e[bnd], v[bnd]);
                    if (f != 0.0) __hip_atomic_fetch_add(hist + i, wk * f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads();
    const int64_t prow = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const double* h = ldsw + a.nE;
    // transposed partials [energy][row]: the final reduction reads rows of one energy contiguously
    for (int t = threadIdx.x; t < a.nE; t += 256)
        partial[(int64_t)t * nrows + prow] = (h[t] + h[a.nE + t]) + (h[2 * a.nE + t] + h[3 * a.nE + t]);
}

// out[col] = sum_rows partial[col][row], one block per column, fixed summation order
__global__ __launch_bounds__(256) void ggr_final_kernel(const double* __restrict__ partial, int64_t nrows, double* __restrict__ out) {
    __shared__ double red[256];
    const double* __restrict__ p = partial + (int64_t)blockIdx.x * nrows;
    double s = 0.0;
    for (int64_t r = threadIdx.x; r < nrows; r += 256) s += p[r];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

}  // namespace

#define ABZ_GGR_ND(KERNEL, ...)                                          \
    do {                                                                  \
        switch (n <= 4 ? n : 1) {                                         \
            case 1: ABZ_GGR_D(KERNEL, 1, __VA_ARGS__); break;             \
            case 2: ABZ_GGR_D(KERNEL, 2, __VA_ARGS__); break;             \
            case 3: ABZ_GGR_D(KERNEL, 3, __VA_ARGS__); break;             \
            default: ABZ_GGR_D(KERNEL, 4, __VA_ARGS__); break;            \
        }                                                                 \
    } while (0)
#define ABZ_GGR_D(KERNEL, NN, ...)                                                                              \
    switch (d) {                                                                                                \
        case 1: hipLaunchKernelGGL((KERNEL<NN, 1>), grid, dim3(256), lds, ctx->stream, __VA_ARGS__); break;     \
        case 2: hipLaunchKernelGGL((KERNEL<NN, 2>), grid, dim3(256), lds, ctx->stream, __VA_ARGS__); break;     \
        default: hipLaunchKernelGGL((KERNEL<NN, 3>), grid, dim3(256), lds, ctx->stream, __VA_ARGS__); break;    \
    }

int launch_ggr(abz_ctx* ctx, int n, int d, int npt, PlaneView E, PlaneView V, const double* w, int64_t nk,
               const double* Es_host, int nE, double* out_host) {
    const int brows = n > 4 ? n : 1;  // n > 4: one block row per band
    GgrArgs a;
    a.E = E;
    a.V = V;
    a.w = w;
    a.nk = nk;
    a.n = n;
    a.d = d;
    a.vstride = n;
    a.b = 1.0 / (2.0 * (double)npt);
    // ascending energies (stable order of equal ones), results go back through the permutation
    std::vector<int> perm((size_t)nE);
    for (int i = 0; i < nE; ++i) perm[(size_t)i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return Es_host[x] < Es_host[y]; });
    std::vector<double> Es((size_t)nE), res((size_t)nE);
    for (int i = 0; i < nE; ++i) Es[(size_t)i] = Es_host[perm[(size_t)i]];
    constexpr int CH = 1024;  // energies per launch: 5 x 8 KB of LDS
    const int64_t nblocks = std::min<int64_t>(cdiv64(nk, 256), 256 * 8);
    const int64_t nrows = nblocks * brows;
    int rc = ctx->scratch[1].reserve(sizeof(double) * (size_t)(nrows * std::min(nE, CH)));
    if (rc) return rc;
    if ((rc = ctx->scratch[2].reserve(sizeof(double) * (size_t)nE * 2))) return rc;
    double* partial = ctx->scratch[1].as<double>();
    double* Es_dev = ctx->scratch[2].as<double>();
    double* outd = Es_dev + nE;
    // energies in and sums out through the pinned mailbox: an asynchronous copy, the last kernel writes the sums into
    // host memory itself, one stream synchronisation per call
    const bool mb = mbox_reserve(ctx) == ABZ_OK && sizeof(double) * (size_t)nE <= ctx->mbox_cap / 2;
    const double* res_host = res.data();
    if (mb) {
        std::memcpy(ctx->mbox, Es.data(), sizeof(double) * (size_t)nE);
        ABZ_HIP(hipMemcpyAsync(Es_dev, ctx->mbox, sizeof(double) * (size_t)nE, hipMemcpyHostToDevice, ctx->stream));
        outd = reinterpret_cast<double*>(static_cast<char*>(ctx->mbox_dev) + ctx->mbox_cap / 2);
        res_host = reinterpret_cast<const double*>(static_cast<const char*>(ctx->mbox) + ctx->mbox_cap / 2);
    } else {
        ABZ_HIP(hipMemcpyAsync(Es_dev, Es.data(), sizeof(double) * (size_t)nE, hipMemcpyHostToDevice, ctx->stream));
    }
    // an equispaced list (the usual linspace sweep) lets a thread compute its window's first index instead of searching
    {
        const bool off = !abz_switch(SW_GGR_UNIFORM);  // per call: tests compare both
        const double step = nE >= 2 ? (Es[(size_t)nE - 1] - Es[0]) / (double)(nE - 1) : 0.0;
        bool uni = !off && nE >= 8 && step > 0.0;
        for (int i = 0; i < nE && uni; ++i) uni = std::fabs(Es[(size_t)i] - (Es[0] + (double)i * step)) <= 1e-6 * step;
        a.inv_step = uni ? 1.0 / step : 0.0;
    }
    for (int s0 = 0; s0 < nE; s0 += CH) {
        const int cnt = std::min(CH, nE - s0);
        a.Es = Es_dev + s0;
        a.nE = cnt;
        ProfScope ps(ctx, ABZ_K_GGR);
        const size_t lds = sizeof(double) * 5 * (size_t)cnt;
        const dim3 grid((unsigned)nblocks, (unsigned)brows);
        ABZ_GGR_ND(ggr_window_kernel, a, partial, nrows);
        ABZ_HIP(hipGetLastError());
        hipLaunchKernelGGL(ggr_final_kernel, dim3((unsigned)cnt), dim3(256), 0, ctx->stream, partial, nrows, outd + s0);
        ABZ_HIP(hipGetLastError());
    }
    if (!mb) ABZ_HIP(hipMemcpyAsync(res.data(), outd, sizeof(double) * (size_t)nE, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < nE; ++i) out_host[perm[(size_t)i]] = res_host[(size_t)i];
    return ABZ_OK;
}
#undef ABZ_GGR_D
#undef ABZ_GGR_ND

}  // namespace abz
