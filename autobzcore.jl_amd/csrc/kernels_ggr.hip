// Fused GGR build (gfx950 only): eigenvalues and band velocities of every PTR node in ONE kernel.
//
// ref: src/dos_ggr.jl:14-44 (get_ggr_data): per node  h, V = x.s ; e, U = eigen(Hermitian(h)) ;
//      v_j = real(diag(U' V_j U)) * t_j.
//
// The unfused build (api.cpp::rule_fill, round 1/2) ran 1 + 3 x (series launch + velocity_kernel) and moved the
// eigenvectors and every dH/dk_j through HBM: ~1.5 KB per node against the 8 n (1 + d) bytes the rule keeps.
// Here a wave evaluates H and the d derivative matrices of its nodes from the same staged coefficients, solves
// the eigenproblem in registers and stores only (e, v): 96 B per node for 3 bands in 3 dimensions.
//
//  * Hermitian series only (c(-R) = c(R)^dagger): H and dH/dk_j are Hermitian, upper triangles are accumulated.
//  * dH/dk_1 shares the level-1 coefficients with H (phase times 2 pi i (first + m)); dH/dk_j, j >= 2, need level-1
//    sets contracted with the derivative factor on variable j.  FUSE: the wave contracts the level-1 sets of its
//    lines itself from the level-2 sets of its block (LDS), so the level-1 sets never exist in HBM and the work
//    loop holds no global load.  !FUSE (d = 1, or sets too large for LDS): the d level-1 families are read.
//  * Lane mapping: a wave works on TWO grid lines at a time, one per half-wave (32 lanes); a pass covers 32 KPL
//    nodes of each line, KPL = 2 for the body (two nodes share every broadcast ds_read_b128 of a coefficient:
//    18 reads per 144 FMAs) and KPL = 1 for a tail of <= 32 nodes.  npt = 150: 2 + 2 + 1 half-passes = 160
//    lane-slots per line (94 %), where whole-wave passes would need 192 (one line per wave, 78 %).
//  * Velocities without eigenvectors (n <= 3): v_b = tr(P_b D) with the spectral projector
//    P_b = prod_{c != b} (B - w_c) / p'(w_b), B = H - (tr H / n) I, so for n = 3
//        v_b = [tr(B^2 D) + w_b tr(B D) + (c2 + w_b^2) tr D] / (3 w_b^2 + c2),   c2 = -tr(B^2) / 2.
//    Conditioning eps ||B||^2 ||D|| / |p'(w_b)|: nodes with min_b |p'(w_b)| < 1e-6 ||B||_F^2 (degenerate or nearly
//    degenerate bands: high-symmetry points and lines) take the Jacobi eigenvectors instead, as do 4 bands.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "abz_internal.h"
#include "device_math.h"

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
    int dbg;    // experiments (ABZ_GGR_DEBUG): bit 0 skip the node solve, bit 1 one m-iteration only, bit 2 skip the contraction
    int pitch;  // padded row length: columns npt..pitch-1 are written too (whole 128-B lines)
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

template <int N, int D>
__device__ __forceinline__ void ggr_node(const CMat<N> (&A)[D + 1], double (&e)[N], double (&v)[D][N]) {
    if constexpr (N == 1) {
        e[0] = A[0].re[0][0];
#pragma unroll
        for (int j = 0; j < D; ++j) v[j][0] = A[j + 1].re[0][0];
    } else if constexpr (N == 2) {
        const CMat<2>& H = A[0];
        const double q = 0.5 * (H.re[0][0] + H.re[1][1]);
        const double d0 = 0.5 * (H.re[0][0] - H.re[1][1]);
        const double br = H.re[0][1], bi = H.im[0][1];
        const double r2 = d0 * d0 + br * br + bi * bi;
        const double r = sqrt(r2);
        if (r > 1e-13 * (fabs(H.re[0][0]) + fabs(H.re[1][1]) + r)) {
            e[0] = q - r;
            e[1] = q + r;
            const double hr = 0.5 / r;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const CMat<2>& Dm = A[j + 1];
                const double t0 = Dm.re[0][0] + Dm.re[1][1];
                const double t1 = d0 * (Dm.re[0][0] - Dm.re[1][1]) + 2.0 * (br * Dm.re[0][1] + bi * Dm.im[0][1]);
                // P_b = (B + w_b) / (2 w_b), w_b = -+ r
                v[j][0] = 0.5 * t0 - t1 * hr;
                v[j][1] = 0.5 * t0 + t1 * hr;
            }
        } else {
            ggr_node_jacobi<2, D>(A, e, v);
        }
    } else if constexpr (N == 3) {
        const CMat<3>& H = A[0];
        herm_eig3_values(H, e);
        const double q = (H.re[0][0] + H.re[1][1] + H.re[2][2]) * (1.0 / 3.0);
        const double d0 = H.re[0][0] - q, d1 = H.re[1][1] - q, d2 = H.re[2][2] - q;
        const double br = H.re[0][1], bi = H.im[0][1];  // B01
        const double cr = H.re[0][2], ci = H.im[0][2];  // B02
        const double dr = H.re[1][2], di = H.im[1][2];  // B12
        const double nb = br * br + bi * bi, nc = cr * cr + ci * ci, nd = dr * dr + di * di;
        const double p2 = d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * (nb + nc + nd);
        const double c2 = -0.5 * p2;
        double w[3], rp[3];
        bool ok = true;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            w[b] = e[b] - q;
            const double pp = fma(3.0 * w[b], w[b], c2);
            ok = ok && (fabs(pp) > 1e-6 * p2);
            rp[b] = 1.0 / pp;
        }
        if (ok) {
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
        } else {
            ggr_node_jacobi<3, D>(A, e, v);
        }
    } else {
        ggr_node_jacobi<N, D>(A, e, v);
    }
}
#undef ABZ_TRHD3

// Accumulators of one node: H and D derivative matrices, upper triangles.  cset: the lane's line, D sets of MNN
// complex coefficients in LDS ([set][m][a + N b]).
template <int N, int D, int KPL, bool NT>
__device__ __forceinline__ void ggr_unit(const GgrBuildArgs& a, const double2* __restrict__ cset, int MNN, const double2* tab_l,
                                         int fm, int i0, int sub, int64_t lineA, int half, bool active) {
    CMat<N> A[KPL][D + 1];
    double zr[KPL], zi[KPL], pr[KPL], pi[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int i1 = i0 + sub + 32 * j;
        const int ic = i1 < a.npt ? i1 : 0;
        const int iw = (int)(((unsigned)fm * (unsigned)ic) % (unsigned)a.npt);
        const double2 z = tab_l[ic];
        const double2 w = tab_l[iw];
        zr[j] = z.x;
        zi[j] = z.y;
        pr[j] = w.x;
        pi[j] = w.y;
#pragma unroll
        for (int s = 0; s <= D; ++s) {
#pragma unroll
            for (int bb = 0; bb < N; ++bb) {
#pragma unroll
                for (int aa = 0; aa <= bb; ++aa) {
                    A[j][s].re[aa][bb] = 0.0;
                    A[j][s].im[aa][bb] = 0.0;
                }
            }
        }
    }
    const int Mrun = (a.dbg & 2) ? 1 : a.M;
    for (int m = 0; m < Mrun; ++m) {
        const double f = TWO_PI * (double)(a.first + m);
        double qr[KPL], qi[KPL];  // i f p: phase of d/dx_1
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            qr[j] = -f * pi[j];
            qi[j] = f * pr[j];
        }
        const double2* __restrict__ cm = cset + m * (N * N);
#pragma unroll
        for (int bb = 0; bb < N; ++bb) {
#pragma unroll
            for (int aa = 0; aa <= bb; ++aa) {
                const double2 c = cm[aa + N * bb];
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    A[j][0].re[aa][bb] = fma(c.x, pr[j], A[j][0].re[aa][bb]);
                    A[j][0].re[aa][bb] = fma(-c.y, pi[j], A[j][0].re[aa][bb]);
                    A[j][1].re[aa][bb] = fma(c.x, qr[j], A[j][1].re[aa][bb]);
                    A[j][1].re[aa][bb] = fma(-c.y, qi[j], A[j][1].re[aa][bb]);
                    if (aa != bb) {
                        A[j][0].im[aa][bb] = fma(c.x, pi[j], A[j][0].im[aa][bb]);
                        A[j][0].im[aa][bb] = fma(c.y, pr[j], A[j][0].im[aa][bb]);
                        A[j][1].im[aa][bb] = fma(c.x, qi[j], A[j][1].im[aa][bb]);
                        A[j][1].im[aa][bb] = fma(c.y, qr[j], A[j][1].im[aa][bb]);
                    }
                }
#pragma unroll
                for (int s = 1; s < D; ++s) {
                    const double2 cs = cm[s * MNN + aa + N * bb];
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        A[j][s + 1].re[aa][bb] = fma(cs.x, pr[j], A[j][s + 1].re[aa][bb]);
                        A[j][s + 1].re[aa][bb] = fma(-cs.y, pi[j], A[j][s + 1].re[aa][bb]);
                        if (aa != bb) {
                            A[j][s + 1].im[aa][bb] = fma(cs.x, pi[j], A[j][s + 1].im[aa][bb]);
                            A[j][s + 1].im[aa][bb] = fma(cs.y, pr[j], A[j][s + 1].im[aa][bb]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const double nr = pr[j] * zr[j] - pi[j] * zi[j];
            const double ni = pr[j] * zi[j] + pi[j] * zr[j];
            pr[j] = nr;
            pi[j] = ni;
        }
    }
    // wave-uniform row bases (scalar registers) + one 32-bit lane offset: the half-wave's own line is half * tile
    // further on.  64-bit per-lane addresses here spilled to scratch, and a scratch reload is a vector-memory
    // operation: its s_waitcnt vmcnt drains every store the wave has in flight (measured: the whole store time
    // became serial, 0.075 ms of 0.28 at 150^3).
    double* __restrict__ erow = a.E.base + lineA * a.E.tile;
    double* __restrict__ vrow = a.V.base + lineA * a.V.tile;
    const unsigned lane_off = (unsigned)sub + (half ? (unsigned)a.E.tile : 0u);  // E and V are views of one tile: same stride
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int i1 = i0 + sub + 32 * j;
        __builtin_amdgcn_sched_barrier(0);  // one node after the other: interleaved solves need both working sets
        if (active && i1 < a.pitch) {
            double e[N], v[D][N];
            if (a.dbg & 1) {
#pragma unroll
                for (int b = 0; b < N; ++b) {
                    e[b] = A[j][0].re[b][b] + A[j][0].im[0][N - 1] + A[j][0].re[0][N - 1];
#pragma unroll
                    for (int jj = 0; jj < D; ++jj) v[jj][b] = A[j][jj + 1].re[b][b] + A[j][jj + 1].im[0][N - 1] + A[j][jj + 1].re[0][N - 1];
                }
            } else {
                ggr_node<N, D>(A[j], e, v);
            }
            // the lane offset stays ONE 32-bit register next to scalar row bases: nothing per-lane to hoist (and spill)
            unsigned u = lane_off;
            asm volatile("" : "+v"(u));
            u += (unsigned)(i0 + 32 * j);
#pragma unroll
            for (int b = 0; b < N; ++b) st_f64<NT>((erow + (int64_t)b * a.E.pitch) + u, e[b]);
#pragma unroll
            for (int jj = 0; jj < D; ++jj) {
#pragma unroll
                for (int b = 0; b < N; ++b) st_f64<NT>((vrow + (int64_t)(jj * N + b) * a.V.pitch) + u, v[jj][b]);
            }
        }
    }
}

// all passes of a pair of lines whose sets are staged in the wave's LDS buffer
template <int N, int D, bool NT, int KB>
__device__ __forceinline__ void ggr_line_pair(const GgrBuildArgs& a, const double2* wbuf, int MNN, const double2* tab_l, int fm,
                                              int lane, int64_t lineA, bool haveB) {
    const int half = lane >> 5, sub = lane & 31;
    const double2* cset = wbuf + (size_t)half * D * MNN;
    const bool active = half == 0 || haveB;
    if constexpr (N <= 3 && KB == 2) {
        const int nfull = a.pitch / 64;
        const int rem = a.pitch - 64 * nfull;  // pitch is a multiple of 16: rem in {0, 16, 32, 48}
        for (int p = 0; p < nfull; ++p) ggr_unit<N, D, 2, NT>(a, cset, MNN, tab_l, fm, 64 * p, sub, lineA, half, active);
        if (rem > 32)
            ggr_unit<N, D, 2, NT>(a, cset, MNN, tab_l, fm, 64 * nfull, sub, lineA, half, active);
        else if (rem > 0)
            ggr_unit<N, D, 1, NT>(a, cset, MNN, tab_l, fm, 64 * nfull, sub, lineA, half, active);
    } else {  // 4 bands: 64 accumulator doubles per node, one node per lane
        for (int i0 = 0; i0 < a.pitch; i0 += 32) ggr_unit<N, D, 1, NT>(a, cset, MNN, tab_l, fm, i0, sub, lineA, half, active);
    }
}

constexpr int GGR_MAX_T = 4;  // coefficient elements per lane and set: M n n <= 256

// ---- FUSE: block = (parent, segment of the i2 range); level-2 sets of the parent in LDS
template <int N, int D, bool NT, int KB>
__global__ __launch_bounds__(256, 2) void ggr_build_fused_kernel(GgrBuildArgs a) {
    static_assert(D >= 2, "the fused build contracts variable 2 in the kernel");
    extern __shared__ double2 lds_g[];  // [D-1][M2][MNN] level-2 sets | [npt] phase table | [4 waves][2 lines][D][MNN]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int MNN = a.M * N * N;
    const int L2 = a.M2 * MNN;
    double2* const c2s = lds_g;
    double2* const tab_l = c2s + (size_t)(D - 1) * L2;
    double2* const wbuf = tab_l + a.npt + (size_t)wave * 2 * D * MNN;
    int fm = a.first % a.npt;
    if (fm < 0) fm += a.npt;
    int fm2 = a.first2 % a.npt;
    if (fm2 < 0) fm2 += a.npt;
    const int64_t parent = blockIdx.x / a.nseg;
    const int seg = blockIdx.x - (int)(parent * a.nseg);
    {
        const double2* __restrict__ s0 = a.src2[0] + parent * (int64_t)L2;
        for (int i = threadIdx.x; i < L2; i += 256) c2s[i] = s0[i];
        if constexpr (D == 3) {
            const double2* __restrict__ s1 = a.src2[1] + parent * (int64_t)L2;
            for (int i = threadIdx.x; i < L2; i += 256) c2s[L2 + i] = s1[i];
        }
        for (int i = threadIdx.x; i < a.npt; i += 256) tab_l[i] = a.tab[i];
    }
    __syncthreads();
    // pairs of lines of this block's segment
    const int npairs_all = (a.gcnt + 1) / 2;
    const int plo = (int)(((int64_t)npairs_all * seg) / a.nseg);
    const int phi = (int)(((int64_t)npairs_all * (seg + 1)) / a.nseg);
    const int nT = (MNN + 63) / 64;
    // level-1 sets of grid index i2 into dst[set][idx]: set 0 plain, set 1 derivative on variable 2, set 2 from c2s'
    // level-1 sets of grid indices i2 (line A) and i2 + dB (line B) into dst[line][set][idx]: set 0 plain, set 1 derivative
    // on variable 2, set 2 from c2s'.  Both lines share every coefficient read; 4 D independent accumulation chains.
    auto contract_pair = [&](int i2, int dB, double2* dst) {
        const unsigned ipA0 = (unsigned)(((unsigned)fm2 * (unsigned)i2) % (unsigned)a.npt);
        const unsigned ipB0 = (unsigned)(((unsigned)fm2 * (unsigned)(i2 + dB)) % (unsigned)a.npt);
        for (int t = 0; t < nT; ++t) {
            const int idx = lane + 64 * t;
            const int ii = idx < MNN ? idx : MNN - 1;
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
                const double2 c = row[(size_t)m2 * MNN];
                double2 c3 = c;
                if constexpr (D == 3) c3 = row[(size_t)L2 + (size_t)m2 * MNN];
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
            if (idx < MNN) {
#pragma unroll
                for (int l = 0; l < 2; ++l) {
#pragma unroll
                    for (int s = 0; s < D; ++s) dst[(l * D + s) * MNN + idx] = make_double2(acr[l][s], aci[l][s]);
                }
            }
        }
    };
    for (int p = plo + wave; p < phi; p += 4) {
        const int i2 = a.gbeg + 2 * p;
        const bool haveB = 2 * p + 1 < a.gcnt;
        wave_lds_fence();  // the previous pair's reads of wbuf are done
        if (!(a.dbg & 4)) contract_pair(i2, haveB ? 1 : 0, wbuf);
        wave_lds_fence();
        const int64_t lineA = parent * a.gcnt + 2 * p;
        ggr_line_pair<N, D, NT, KB>(a, wbuf, MNN, tab_l, fm, lane, lineA, haveB);
    }
}

// ---- !FUSE: level-1 families from HBM; pairs of consecutive lines, grid-strided
template <int N, int D, bool NT, int KB>
__global__ __launch_bounds__(256, KB == 1 ? 3 : 2) void ggr_build_lines_kernel(GgrBuildArgs a) {
    extern __shared__ double2 lds_g[];  // [npt] phase table | [4 waves][2 lines][D][MNN]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int MNN = a.M * N * N;
    double2* const tab_l = lds_g;
    double2* const wbuf = tab_l + a.npt + (size_t)wave * 2 * D * MNN;
    int fm = a.first % a.npt;
    if (fm < 0) fm += a.npt;
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
                double2* dst = wbuf + (size_t)(hb * D + s) * MNN;
                for (int idx = lane; idx < MNN; idx += 64) dst[idx] = src[idx];
            }
        }
        wave_lds_fence();
        ggr_line_pair<N, D, NT, KB>(a, wbuf, MNN, tab_l, fm, lane, lineA, haveB);
    }
}

// ---- irregular node lists (symmetric rules): one lane per node, level-1 sets through the constant address space
// (scalar loads wherever a wave's nodes share their set).
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
    const int MNN = a.M * N * N;
    const int64_t slot = a.parents ? a.parents[k] : 0;
    int fm = a.first % a.npt;
    if (fm < 0) fm += a.npt;
    const int i1 = a.gi[k];
    const double2 z = a.tab[i1];
    const double2 w0 = a.tab[(int)(((int64_t)fm * i1) % a.npt)];
    CMat<N> A[D + 1];
#pragma unroll
    for (int s = 0; s <= D; ++s) {
#pragma unroll
        for (int bb = 0; bb < N; ++bb) {
#pragma unroll
            for (int aa = 0; aa <= bb; ++aa) {
                A[s].re[aa][bb] = 0.0;
                A[s].im[aa][bb] = 0.0;
            }
        }
    }
    double pr = w0.x, pi = w0.y;
    cp4_t c0 = as_c4(a.src[0] + slot * MNN);
    cp4_t c1 = as_c4(a.src[D >= 2 ? 1 : 0] + slot * MNN);
    cp4_t c2 = as_c4(a.src[D >= 3 ? 2 : 0] + slot * MNN);
    for (int m = 0; m < a.M; ++m) {
        const double f = TWO_PI * (double)(a.first + m);
        const double qr = -f * pi, qi = f * pr;
#pragma unroll
        for (int bb = 0; bb < N; ++bb) {
#pragma unroll
            for (int aa = 0; aa <= bb; ++aa) {
                const int o = m * (N * N) + aa + N * bb;
                const double cx = c0[o].x, cy = c0[o].y;
                A[0].re[aa][bb] = fma(cx, pr, A[0].re[aa][bb]);
                A[0].re[aa][bb] = fma(-cy, pi, A[0].re[aa][bb]);
                A[1].re[aa][bb] = fma(cx, qr, A[1].re[aa][bb]);
                A[1].re[aa][bb] = fma(-cy, qi, A[1].re[aa][bb]);
                if (aa != bb) {
                    A[0].im[aa][bb] = fma(cx, pi, A[0].im[aa][bb]);
                    A[0].im[aa][bb] = fma(cy, pr, A[0].im[aa][bb]);
                    A[1].im[aa][bb] = fma(cx, qi, A[1].im[aa][bb]);
                    A[1].im[aa][bb] = fma(cy, qr, A[1].im[aa][bb]);
                }
                if constexpr (D >= 2) {
                    const double sx = c1[o].x, sy = c1[o].y;
                    A[2].re[aa][bb] = fma(sx, pr, A[2].re[aa][bb]);
                    A[2].re[aa][bb] = fma(-sy, pi, A[2].re[aa][bb]);
                    if (aa != bb) {
                        A[2].im[aa][bb] = fma(sx, pi, A[2].im[aa][bb]);
                        A[2].im[aa][bb] = fma(sy, pr, A[2].im[aa][bb]);
                    }
                }
                if constexpr (D >= 3) {
                    const double sx = c2[o].x, sy = c2[o].y;
                    A[3].re[aa][bb] = fma(sx, pr, A[3].re[aa][bb]);
                    A[3].re[aa][bb] = fma(-sy, pi, A[3].re[aa][bb]);
                    if (aa != bb) {
                        A[3].im[aa][bb] = fma(sx, pi, A[3].im[aa][bb]);
                        A[3].im[aa][bb] = fma(sy, pr, A[3].im[aa][bb]);
                    }
                }
            }
        }
        const double nr = pr * z.x - pi * z.y;
        const double ni = pr * z.y + pi * z.x;
        pr = nr;
        pi = ni;
    }
    double e[N], v[D][N];
    ggr_node<N, D>(A, e, v);
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

// LDS bytes of the fused kernel; 0 when the case is not supported (then the line kernel or the unfused build runs)
static size_t ggr_fused_lds(int n, int d, int M, int M2, int npt) {
    const size_t mnn = (size_t)M * n * n;
    return sizeof(double2) * ((size_t)(d - 1) * M2 * mnn + (size_t)npt + 4 * 2 * (size_t)d * mnn);
}
static size_t ggr_lines_lds(int n, int d, int M, int npt) {
    const size_t mnn = (size_t)M * n * n;
    return sizeof(double2) * ((size_t)npt + 4 * 2 * (size_t)d * mnn);
}

bool ggr_build_supported(int n, int d, int M, int npt, bool herm) {
    static const bool off = [] { const char* e = getenv("ABZ_GGR_FUSED"); return e && e[0] == '0'; }();
    if (off || !herm || n < 1 || n > 4 || d < 1 || d > 3 || npt >= 65536) return false;
    return M * n * n <= 64 * GGR_MAX_T && ggr_lines_lds(n, d, M, npt) <= 64 * 1024;
}

bool ggr_build_can_fuse(int n, int d, int M, int M2, int npt) {
    static const bool off = [] { const char* e = getenv("ABZ_GGR_FUSE2"); return e && e[0] == '0'; }();
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
    a.dbg = [] { const char* e = getenv("ABZ_GGR_DEBUG"); return e ? atoi(e) : 0; }();
    const int kb = [] { const char* e = getenv("ABZ_GGR_KB"); return e ? atoi(e) : 2; }();  // nodes per lane in body passes
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
        const int force = [] { const char* e = getenv("ABZ_NT_STORES"); return e ? atoi(e) : -1; }();
        const double bytes = 8.0 * (double)(n * (1 + d)) * (double)gs.E.row * (double)gs.nlines;
        a.nt = force >= 0 ? force : (bytes > 256.0 * 1024 * 1024 ? 1 : 0);
    }
    if (gs.fuse) {
        const size_t lds = ggr_fused_lds(n, d, gs.M, gs.M2, gs.npt);
        const int64_t nparents = gs.nlines / std::max(gs.gcnt, 1);
        // segments of a parent's line pairs: ~2 pairs per wave and block, at least enough blocks for two rounds of the chip
        const int npairs = (gs.gcnt + 1) / 2;
        static const int ppb = [] { const char* e = getenv("ABZ_GGR_PAIRS_PER_BLOCK"); return e ? atoi(e) : 8; }();
        int64_t nseg = std::max<int64_t>(1, cdiv64(npairs, std::max(ppb, 1)));
        nseg = std::min<int64_t>(nseg, npairs);
        a.nseg = (int)nseg;
        const unsigned blocks = (unsigned)(nparents * nseg);
#define GF(NN, DD)                                                                                                        \
    if (kb == 1 && a.nt)                                                                                                  \
        hipLaunchKernelGGL((ggr_build_fused_kernel<NN, DD, true, 1>), dim3(blocks), dim3(256), lds, ctx->stream, a);     \
    else if (kb == 1)                                                                                                     \
        hipLaunchKernelGGL((ggr_build_fused_kernel<NN, DD, false, 1>), dim3(blocks), dim3(256), lds, ctx->stream, a);    \
    else if (a.nt)                                                                                                        \
        hipLaunchKernelGGL((ggr_build_fused_kernel<NN, DD, true, 2>), dim3(blocks), dim3(256), lds, ctx->stream, a);     \
    else                                                                                                                  \
        hipLaunchKernelGGL((ggr_build_fused_kernel<NN, DD, false, 2>), dim3(blocks), dim3(256), lds, ctx->stream, a)
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
    } else {
        const size_t lds = ggr_lines_lds(n, d, gs.M, gs.npt);
        const int64_t npairs = (gs.nlines + 1) / 2;
        const unsigned blocks = (unsigned)std::min<int64_t>(cdiv64(npairs, 4), 256 * 12);
#define GL(NN, DD)                                                                                                        \
    if (kb == 1 && a.nt)                                                                                                  \
        hipLaunchKernelGGL((ggr_build_lines_kernel<NN, DD, true, 1>), dim3(blocks), dim3(256), lds, ctx->stream, a);     \
    else if (kb == 1)                                                                                                     \
        hipLaunchKernelGGL((ggr_build_lines_kernel<NN, DD, false, 1>), dim3(blocks), dim3(256), lds, ctx->stream, a);    \
    else if (a.nt)                                                                                                        \
        hipLaunchKernelGGL((ggr_build_lines_kernel<NN, DD, true, 2>), dim3(blocks), dim3(256), lds, ctx->stream, a);     \
    else                                                                                                                  \
        hipLaunchKernelGGL((ggr_build_lines_kernel<NN, DD, false, 2>), dim3(blocks), dim3(256), lds, ctx->stream, a)
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
};

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
            int i0 = 0, len = a.nE;  // first energy >= lo
            while (len > 0) {
                const int half = len >> 1;
                const bool right = Esl[i0 + half] < lo;
                i0 = right ? i0 + half + 1 : i0;
                len = right ? len - half - 1 : half;
            }
            for (int i = i0; i < a.nE; ++i) {
                const double En = Esl[i];
                if (!(En <= hi)) break;
                const double f = ggr_formula<D>(a.b, En, e[bnd], v[bnd]);
                if (f != 0.0) __hip_atomic_fetch_add(hist + i, wk * f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

// The all-pairs scan of rounds 1-2 (ABZ_GGR_SCAN=0; unsorted energy lists need no permutation here): every thread
// evaluates every energy for its node.
template <int N, int D>
__global__ __launch_bounds__(256) void ggr_allpairs_kernel(GgrArgs a, double* __restrict__ partial) {
    extern __shared__ double ldsd[];  // [nE chunk][4]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = k < a.nk;
    const int64_t kk = ok ? k : 0;
    const double wk = ok ? (a.w ? a.w[kk] : 1.0) : 0.0;
    double e[N];
    double v[N][D];
    const double* __restrict__ ei = a.E.base + plane_off(a.E, kk) + (int64_t)blockIdx.y * a.E.pitch;
    const double* __restrict__ vi = a.V.base + plane_off(a.V, kk) + (int64_t)blockIdx.y * a.V.pitch;
#pragma unroll
    for (int bnd = 0; bnd < N; ++bnd) {
        e[bnd] = ei[(int64_t)bnd * a.E.pitch];
#pragma unroll
        for (int j = 0; j < D; ++j) v[bnd][j] = vi[(int64_t)(j * a.vstride + bnd) * a.V.pitch];
    }
    const int64_t prow = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int chunk = 1024;
    for (int s0 = 0; s0 < a.nE; s0 += chunk) {
        const int s1 = min(a.nE, s0 + chunk);
        for (int s = s0; s < s1; ++s) {
            const double En = a.Es[s];
            double acc = 0.0;
#pragma unroll
            for (int bnd = 0; bnd < N; ++bnd) acc += ggr_formula<D>(a.b, En, e[bnd], v[bnd]);
            acc = wave_sum(wk * acc);
            if (lane == 0) ldsd[(s - s0) * 4 + wave] = acc;
        }
        __syncthreads();
        for (int t = threadIdx.x; t < s1 - s0; t += 256)
            partial[prow * a.nE + s0 + t] = ldsd[t * 4] + ldsd[t * 4 + 1] + ldsd[t * 4 + 2] + ldsd[t * 4 + 3];
        __syncthreads();
    }
}

__global__ void final_reduce_real_kernel(const double* __restrict__ partial, int64_t nblocks, int64_t ncols,
                                         double* __restrict__ out) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    double s = 0.0;
    for (int64_t b = 0; b < nblocks; ++b) s += partial[b * ncols + col];
    out[col] = s;
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
    static const bool windowed = [] { const char* e = getenv("ABZ_GGR_SCAN"); return !(e && e[0] == '0'); }();
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
    if (!windowed) {
        const int64_t nblocks = cdiv64(nk, 256);
        int rc = ctx->scratch[1].reserve(sizeof(double) * (size_t)(nblocks * brows * nE));
        if (rc) return rc;
        if ((rc = ctx->scratch[2].reserve(sizeof(double) * (size_t)nE * 2))) return rc;
        double* partial = ctx->scratch[1].as<double>();
        double* Es_dev = ctx->scratch[2].as<double>();
        double* outd = Es_dev + nE;
        ABZ_HIP(hipMemcpyAsync(Es_dev, Es_host, sizeof(double) * (size_t)nE, hipMemcpyHostToDevice, ctx->stream));
        a.Es = Es_dev;
        a.nE = nE;
        {
            ProfScope ps(ctx, ABZ_K_GGR);
            const size_t lds = sizeof(double) * 4 * (size_t)std::min(nE, 1024);
            const dim3 grid((unsigned)nblocks, (unsigned)brows);
            ABZ_GGR_ND(ggr_allpairs_kernel, a, partial);
            ABZ_HIP(hipGetLastError());
            hipLaunchKernelGGL(final_reduce_real_kernel, dim3((unsigned)cdiv64(nE, 256)), dim3(256), 0, ctx->stream, partial,
                               nblocks * brows, (int64_t)nE, outd);
            ABZ_HIP(hipGetLastError());
        }
        ABZ_HIP(hipMemcpyAsync(out_host, outd, sizeof(double) * (size_t)nE, hipMemcpyDeviceToHost, ctx->stream));
        ABZ_HIP(hipStreamSynchronize(ctx->stream));
        return ABZ_OK;
    }
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
    ABZ_HIP(hipMemcpyAsync(Es_dev, Es.data(), sizeof(double) * (size_t)nE, hipMemcpyHostToDevice, ctx->stream));
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
    ABZ_HIP(hipMemcpyAsync(res.data(), outd, sizeof(double) * (size_t)nE, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < nE; ++i) out_host[perm[(size_t)i]] = res[(size_t)i];
    return ABZ_OK;
}
#undef ABZ_GGR_D
#undef ABZ_GGR_ND

}  // namespace abz
