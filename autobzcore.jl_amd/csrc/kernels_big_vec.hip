// Band velocities for 33...64 bands (ref: src/dos_ggr.jl:14-44 -- `e, U = eigen(Hermitian(h))`, `v_j = Re diag(U' dH/dk_j U) t_j`,
// LAPACK there): the eigenvector half of the GGR build of kernels_big.hip.  ONE WORKGROUP OF TWO WAVES PER NODE, lane b of either
// wave owns band b:
//   (1) the node's real tridiagonal (big_tridiag_kernel) and its eigenvalues (big_qr_kernel) are there already; lane b of wave 0
//       finds the eigenvector z_b of its eigenvalue by inverse iteration on T - lambda_b I with the pivoted LU of LAPACK's
//       dlagtf / dlagts (dstein's scheme, the one of kernels_ggr_rows.hip: three solves from a lane-dependent start vector,
//       tiny pivots replaced by eps ||T||, members of a cluster perturbed apart and re-orthogonalised lowest first).  The
//       tridiagonal is the same for all lanes (LDS, broadcast reads); the factors are not, and 64 rows x (pivot, superdiagonal,
//       multiplier) per lane beside the vector do not fit the 256 registers the vector ALU can address: the forward pass keeps
//       the elimination state at every eighth row only, and the back substitution re-eliminates a chunk of eight rows from its
//       checkpoint before it substitutes them;
//   (2) u_b = H_0 ... H_{n-3} P z_b with the rows split between the waves in chunks of eight (wave w: chunks w, w + 2, ...: the
//       triangle of reflector x row work is shared evenly, 2 x 32 doubles per lane), the two partial sums v_K^H u meeting in LDS,
//       one barrier per reflector.  The reflectors v_K (kept by big_tridiag_kernel, column-packed) are the same for all lanes:
//       staged in LDS once, read back by broadcast;
//   (3) v_{b,j} = Re u_b^H D_j u_b = sum_r [Re u_r t1_r + Im u_r (t2_r + 2 t4_r)], t1 = Re D Re u, t2 = Re D Im u, t4 = Im D Re u
//       (D_j Hermitian: three real FMAs per matrix element instead of four).  Eight rows of D_j at a time are staged in LDS
//       (column r of the stored matrix is the conjugate of row r; the next eight are in flight meanwhile) and read by
//       broadcast; every wave sums over its own columns, the partial t of the eight rows go to the wave that owns them (LDS),
//       whose u_r come out of the register file by a uniform switch.
//       (First version: reflectors and D_j as scalar-cache operands of the FMAs -- no LDS traffic, but a 64-B s_load feeds twelve
//       FMAs, every one misses the 16-KB scalar cache, and ~100 SGPRs hold too few of them in flight: 1.8 ms per 1 152 nodes
//       of 64 bands, nine times the issue time of its instructions.)
// Only (e, v) reach the rule; nothing of U is stored.
#include <cstdlib>
#include <utility>

#include "abz_internal.h"
#include "rows_device.h"

namespace abz {

namespace {

constexpr int NM = 64;
constexpr int BIG_NP_V = 64;  // rows of the tridiagonal scratch (kernels_big.hip: BIG_NP)

__device__ __forceinline__ void vwave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int64_t vview_off(const PlaneView& v, int64_t k) {
    const int64_t line = k / v.line_len;
    return line * v.tile + (k - line * v.line_len);
}

// f(integral_constant<int, C>) for the chunks C of eight rows that hold rows < n8 (n8: n rounded up to a multiple of 8)
template <typename F, int... C>
__device__ __forceinline__ void chunks_up(int n8, F&& f, std::integer_sequence<int, C...>) {
    ((void)((8 * C < n8) ? (f(std::integral_constant<int, C>()), 0) : 0), ...);
}
template <typename F, int... C>
__device__ __forceinline__ void chunks_down(int n8, F&& f, std::integer_sequence<int, C...>) {
    ((void)((8 * (NM / 8 - 1 - C) < n8) ? (f(std::integral_constant<int, NM / 8 - 1 - C>()), 0) : 0), ...);
}
using Chunks = std::make_integer_sequence<int, NM / 8>;

// One elimination step of dlagtf on rows k, k + 1 of T - lam I.  State: (ca, cb) = the current row k (diagonal, superdiagonal),
// scale1.  Out: U's pivot and first superdiagonal of row k, the multiplier, whether the rows were interchanged.
struct TriStep {
    double piv, b, mult;
    bool sw;
};
__device__ __forceinline__ TriStep tri_step(const double* sd, const double* so, int k, double lam, double& ca, double& cb, double& scale1) {
    const double ck = so[k], ak1 = sd[k + 1] - lam, bk1 = so[k + 1];
    const double scale2 = fabs(ck) + fabs(ak1) + fabs(bk1);
    TriStep s;
    s.sw = fabs(ck) * scale1 > fabs(ca) * scale2;  // dlagtf: interchange when |c| / scale2 > |a| / scale1
    s.piv = s.sw ? ck : ca;
    const double ip = rcp_nr(fabs(s.piv) < 1e-290 ? 1e-290 : s.piv);
    s.mult = (s.sw ? ca : ck) * ip;
    s.b = s.sw ? ak1 : cb;
    const double na = s.sw ? fma(-s.mult, ak1, cb) : fma(-s.mult, cb, ak1);
    const double nb = s.sw ? -s.mult * bk1 : bk1;
    ca = na;
    cb = nb;
    scale1 = s.sw ? scale1 : scale2;
    return s;
}
__device__ __forceinline__ double tri_rpiv(double piv) {  // dlagts, job = -1: a pivot below eps (unit scale) is replaced by +-eps
    const double pk = fabs(piv) < 2.3e-16 ? (piv < 0.0 ? -2.3e-16 : 2.3e-16) : piv;
    return rcp_nr(pk);
}

// y <- inv(T - lam I) y, scaled to unit maximum norm.  T: diagonal sd, couplings so (LDS; rows >= n: a decoupled block with
// diagonal 4 and coupling 0, so[NM - 1 ...] = 0; sd, so hold NM + 2 entries).
__device__ __forceinline__ void tri_fsolve(const double* sd, const double* so, int n8, double lam, double (&y)[NM]) {
    double ka[NM / 8], kb[NM / 8], ks[NM / 8];  // the elimination state at the first row of every chunk
    {
        double ca = sd[0] - lam, cb = so[0];
        double scale1 = fabs(ca) + fabs(cb);
        chunks_up(
            n8,
            [&](auto cc) {
                constexpr int C = decltype(cc)::value;
                ka[C] = ca;
                kb[C] = cb;
                ks[C] = scale1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 8 * C + j;
                    if (k + 1 < NM) {
                        const TriStep s = tri_step(sd, so, k, lam, ca, cb, scale1);
                        const double yk = y[k], yk1 = y[k + 1 < NM ? k + 1 : k];
                        y[k] = s.sw ? yk1 : yk;
                        y[k + 1 < NM ? k + 1 : k] = s.sw ? fma(-s.mult, yk1, yk) : fma(-s.mult, yk, yk1);
                    }
                }
            },
            Chunks());
    }
    chunks_down(
        n8,
        [&](auto cc) {
            constexpr int C = decltype(cc)::value;
            double ca = ka[C], cb = kb[C], scale1 = ks[C];
            double ia[8], b[8];
            unsigned sw8 = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * C + j;
                if (k + 1 < NM) {
                    const TriStep s = tri_step(sd, so, k, lam, ca, cb, scale1);
                    ia[j] = tri_rpiv(s.piv);
                    b[j] = s.b;
                    sw8 |= s.sw ? (1u << j) : 0u;
                } else {
                    ia[j] = tri_rpiv(ca);
                    b[j] = 0.0;
                }
            }
#pragma unroll
            for (int j = 7; j >= 0; --j) {
                const int k = 8 * C + j;
                double t = y[k];
                if (k + 1 < NM) t = fma(-b[j], y[k + 1 < NM ? k + 1 : k], t);
                if (k + 2 < NM) {
                    const double dd = ((sw8 >> j) & 1u) ? so[k + 1] : 0.0;  // U's second superdiagonal: the old b[k + 1] of an interchange
                    t = fma(-dd, y[k + 2 < NM ? k + 2 : k], t);
                }
                y[k] = t * ia[j];
            }
        },
        Chunks());
    double mx = 0.0;
    chunks_up(
        n8,
        [&](auto cc) {
            constexpr int C = decltype(cc)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmax(mx, fabs(y[8 * C + j]));
        },
        Chunks());
    const double s = (mx > 1e-290 && mx < 1e290) ? rcp_nr(mx) : 1.0;
    chunks_up(
        n8,
        [&](auto cc) {
            constexpr int C = decltype(cc)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) y[8 * C + j] *= s;
        },
        Chunks());
}

__device__ __forceinline__ void unit2v(int n8, double (&y)[NM]) {
    double nn = 0.0;
    chunks_up(
        n8,
        [&](auto cc) {
            constexpr int C = decltype(cc)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) nn = fma(y[8 * C + j], y[8 * C + j], nn);
        },
        Chunks());
    const double s = nn > 1e-290 ? rsqrt_nr(nn) : 0.0;
    chunks_up(
        n8,
        [&](auto cc) {
            constexpr int C = decltype(cc)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) y[8 * C + j] *= s;
        },
        Chunks());
}

// ---- the two-wave part: wave W owns the chunks C = 2 LC + W, LC = 0 ... 3, of eight rows each; row 8 C + j sits at 8 LC + j
constexpr int NH = NM / 2;
using HalfChunks = std::make_integer_sequence<int, NM / 16>;
template <int W, typename F, int... LC>
__device__ __forceinline__ void own_chunks(int n8, F&& f, std::integer_sequence<int, LC...>) {
    ((void)((8 * (2 * LC + W) < n8) ? (f(std::integral_constant<int, LC>()), 0) : 0), ...);
}

// rooms in LDS (doubles)
constexpr int XW = 2 * 2 * 2 * 64;  // [parity][wave][re, im][lane]: the partial v_K^H u
constexpr int STG = 8 * NM * 2;     // eight rows of D_j, complex, zero beyond n
constexpr int PB = 24 * 64;         // [8 rows x (t1, t2, t4)][lane]: the partial rows of D u of the wave that does not own the rows

struct BigVecLds {
    double sd[NM + 2], so[NM + 2], sbeta[NM], spr[NM], spi[NM];
    double xw[XW];
    // the eigenvectors of the tridiagonal [row][lane]; then the reflectors (column K: rows K + 1 ... n8 - 1 at K n8 - K (K + 1) / 2
    // + i - K - 1, complex, zero beyond n); then the staged rows of D_j and the partial rows of D u
    double big[NM * 64];
    double accx[64];
};
static_assert(STG + PB <= NM * 64 && (NM * (NM - 1) / 2) * 2 <= NM * 64, "the rooms that share BigVecLds::big");

// u <- H_K u for the reflectors K of block KB (K = 8 KB + 7 ... 8 KB), rolled over K; this wave's rows i > 8 KB, compile-time.
// The reflectors are the same for all lanes: broadcast reads from LDS.
template <int W, int KB>
__device__ __forceinline__ void back_block(int n, int n8, int lane, const double2* refl, const double* sbeta, double* xw, int& xp,
                                           double (&ur)[NH], double (&ui)[NH]) {
#pragma unroll 1
    for (int kk = 7; kk >= 0; --kk) {
        const int K = 8 * KB + kk;
        if (K + 2 >= n) continue;  // uniform in the workgroup
        const double beta = sbeta[K];
        if (beta == 0.0) continue;  // (a column that was zero already: no reflector)
        const double2* vk = refl + (K * n8 - K * (K + 1) / 2 - K - 1);
        double wr[2] = {0.0, 0.0}, wi[2] = {0.0, 0.0};
        own_chunks<W>(
            n8,
            [&](auto lc) {
                constexpr int LC = decltype(lc)::value, C = 2 * LC + W;
                if constexpr (C >= KB) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 8 * C + j;
                        if (C == KB && i <= 8 * KB) continue;  // (compile time)
                        double2 v = vk[(C == KB && i <= K) ? K + 1 : i];
                        if (C == KB && i <= K) v = make_double2(0.0, 0.0);
                        // w += conj(v_i) u_i
                        wr[j & 1] = fma(v.x, ur[8 * LC + j], wr[j & 1]);
                        wr[j & 1] = fma(v.y, ui[8 * LC + j], wr[j & 1]);
                        wi[j & 1] = fma(v.x, ui[8 * LC + j], wi[j & 1]);
                        wi[j & 1] = fma(-v.y, ur[8 * LC + j], wi[j & 1]);
                    }
                }
            },
            HalfChunks());
        double pr = wr[0] + wr[1], pi = wi[0] + wi[1];
        xw[((xp * 2 + W) * 2 + 0) * 64 + lane] = pr;
        xw[((xp * 2 + W) * 2 + 1) * 64 + lane] = pi;
        __syncthreads();
        pr += xw[((xp * 2 + (1 - W)) * 2 + 0) * 64 + lane];
        pi += xw[((xp * 2 + (1 - W)) * 2 + 1) * 64 + lane];
        xp ^= 1;  // (the other room next time: nobody writes this one before everybody has passed the next barrier)
        const double wwr = pr * beta, wwi = pi * beta;
        own_chunks<W>(
            n8,
            [&](auto lc) {
                constexpr int LC = decltype(lc)::value, C = 2 * LC + W;
                if constexpr (C >= KB) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 8 * C + j;
                        if (C == KB && i <= 8 * KB) continue;
                        double2 v = vk[(C == KB && i <= K) ? K + 1 : i];
                        if (C == KB && i <= K) v = make_double2(0.0, 0.0);
                        // u_i -= w v_i
                        ur[8 * LC + j] = fma(-wwr, v.x, ur[8 * LC + j]);
                        ur[8 * LC + j] = fma(wwi, v.y, ur[8 * LC + j]);
                        ui[8 * LC + j] = fma(-wwr, v.y, ui[8 * LC + j]);
                        ui[8 * LC + j] = fma(-wwi, v.x, ui[8 * LC + j]);
                    }
                }
            },
            HalfChunks());
    }
}

template <int W, int... KB>
__device__ __forceinline__ void back_blocks(int n, int n8, int lane, const double2* refl, const double* sbeta, double* xw, int& xp,
                                            double (&ur)[NH], double (&ui)[NH], std::integer_sequence<int, KB...>) {
    // the last reflector first
    ((void)((8 * (NM / 8 - 1 - KB) + 2 < n) ? (back_block<W, NM / 8 - 1 - KB>(n, n8, lane, refl, sbeta, xw, xp, ur, ui), 0) : 0), ...);
}

template <int LC>
__device__ __forceinline__ void take8(const double (&ur)[NH], const double (&ui)[NH], double (&ar)[8], double (&ai)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ar[j] = ur[8 * LC + j];
        ai[j] = ui[8 * LC + j];
    }
}

// rows 8 rb ... 8 rb + 7 of the Hermitian matrix whose element (a, b) sits at D[a + n b]: row r is read as the conjugate of column
// r (contiguous); thread t of the 128 takes the elements t, t + 128, ... of the [8][NM] block, zero beyond n
__device__ __forceinline__ void stage_load(const double2* __restrict__ D, int n, int rb, int tid, double2 (&g)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 128 * q, jr = e >> 6, c = e & 63, r = 8 * rb + jr;
        g[q] = (r < n && c < n) ? D[c + n * r] : make_double2(0.0, 0.0);
    }
}

// This wave's share of Re u^H D u.  Returns the sum over the rows this wave owns (the other wave returns the rest).
template <int W>
__device__ __forceinline__ double quad_form_k(const double2* __restrict__ D, int n, int n8, int tid, int lane, double2* stage, double* pb,
                                              const double (&ur)[NH], const double (&ui)[NH]) {
    double acc = 0.0;
    double2 g[4];
    stage_load(D, n, 0, tid, g);
#pragma unroll 1
    for (int rb = 0; 8 * rb < n; ++rb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) stage[tid + 128 * q] = g[q];
        __syncthreads();
        if (8 * (rb + 1) < n) stage_load(D, n, rb + 1, tid, g);  // in flight during the sums
        double t1[8], t2[8], t4[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            t1[j] = 0.0;
            t2[j] = 0.0;
            t4[j] = 0.0;
        }
        own_chunks<W>(
            n8,
            [&](auto lc) {
                constexpr int LC = decltype(lc)::value, c0 = 8 * (2 * LC + W);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const double2 dv = stage[j * NM + c0 + c];  // D_rc = (dv.x, -dv.y)
                        t1[j] = fma(dv.x, ur[8 * LC + c], t1[j]);
                        t2[j] = fma(dv.x, ui[8 * LC + c], t2[j]);
                        t4[j] = fma(-dv.y, ur[8 * LC + c], t4[j]);
                    }
                }
            },
            HalfChunks());
        const bool mine = (rb & 1) == W;  // uniform: the rows of chunk rb live in wave rb & 1
        if (!mine) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pb[(3 * j + 0) * 64 + lane] = t1[j];
                pb[(3 * j + 1) * 64 + lane] = t2[j];
                pb[(3 * j + 2) * 64 + lane] = t4[j];
            }
        }
        __syncthreads();
        if (mine) {
            double ar[8], ai[8];
            switch (rb >> 1) {
                case 0: take8<0>(ur, ui, ar, ai); break;
                case 1: take8<1>(ur, ui, ar, ai); break;
                case 2: take8<2>(ur, ui, ar, ai); break;
                default: take8<3>(ur, ui, ar, ai); break;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double s1 = t1[j] + pb[(3 * j + 0) * 64 + lane];
                const double s2 = t2[j] + pb[(3 * j + 1) * 64 + lane];
                const double s4 = t4[j] + pb[(3 * j + 2) * 64 + lane];
                acc = fma(ar[j], s1, acc);  // (rows >= n: u_r = 0)
                acc = fma(ai[j], fma(2.0, s4, s2), acc);
            }
        }
    }
    return acc;
}

struct BigVecArgs {
    int64_t tri_nk, node0, nnodes, dstride;  // dstride: complex numbers from one matrix array to the next
    int n, d;
    int phases;  // (timing experiments: bit 0 inverse iteration, 1 back-transformation, 2 quadratic forms; 7 in production)
    PlaneView E, V;
};

template <int W>
__device__ __forceinline__ void big_vec_wave(const BigVecArgs& a, BigVecLds& L, int lane, int64_t node, const double2* __restrict__ kn,
                                             const double2* __restrict__ Dm) {
    const int n = a.n, n8 = (n + 7) & ~7, tid = lane + 64 * W;
    // ---- this wave's rows of P z
    double ur[NH], ui[NH];
    own_chunks<W>(
        NM,
        [&](auto lc) {
            constexpr int LC = decltype(lc)::value, C = 2 * LC + W;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = 8 * C + j;
                const double zv = (i < n) ? L.big[i * 64 + lane] : 0.0;
                ur[8 * LC + j] = L.spr[i] * zv;
                ui[8 * LC + j] = L.spi[i] * zv;
            }
        },
        HalfChunks());
    __syncthreads();  // the room of z is free from here on: the reflectors move in
    double2* refl = reinterpret_cast<double2*>(L.big);
    for (int K0 = 0; K0 + 2 < n; K0 += 8) {  // (eight columns' loads in flight)
        double2 g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int K = K0 + q, i = K + 1 + tid;
            g[q] = (K + 2 < n && i < n) ? kn[K * n - K * (K + 1) / 2 + (i - K - 1)] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int K = K0 + q, i = K + 1 + tid;
            if (K + 2 < n && i < n8) refl[K * n8 - K * (K + 1) / 2 + (i - K - 1)] = g[q];
        }
    }
    __syncthreads();
    // ---- u = H_0 ... H_{n-3} (P z)
    int xp = 0;
    if (a.phases & 2) back_blocks<W>(n, n8, lane, refl, L.sbeta, L.xw, xp, ur, ui, Chunks());
    __syncthreads();
    // ---- velocities
    double2* stage = reinterpret_cast<double2*>(L.big);
    double* pb = L.big + STG;
    for (int j = 0; j < a.d; ++j) {
        const double2* __restrict__ D = Dm + (int64_t)j * a.dstride + node * (int64_t)n * n;
        const double part = (a.phases & 4) ? quad_form_k<W>(D, n, n8, tid, lane, stage, pb, ur, ui) : ur[0];
        if (W == 1) L.accx[lane] = part;
        __syncthreads();
        if (W == 0 && lane < n)
            a.V.base[vview_off(a.V, a.node0 + node) + (int64_t)(j * n + lane) * a.V.pitch] = part + L.accx[lane];
        __syncthreads();
    }
}

__global__ __launch_bounds__(128, 2) void big_ggr_kernel(const double* __restrict__ tri, const double2* __restrict__ keep,
                                                         const double2* __restrict__ Dm, BigVecArgs a) {
    __shared__ BigVecLds L;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = a.n;
    const int n8 = (n + 7) & ~7;
    const int np = n * (n + 1) / 2, ncol = np - n;
    for (int64_t node = blockIdx.x; node < a.nnodes; node += gridDim.x) {
        __syncthreads();
        const double2* __restrict__ kn = keep + node * (int64_t)np;
        if (wv == 0) {
            // ---- the tridiagonal, scaled to unit Gershgorin radius; couplings floored like the eigenvalue kernels do
            const double dv = lane < n ? tri[(int64_t)lane * a.tri_nk + node] : 0.0;
            const double e2 = lane + 1 < n ? tri[(int64_t)(BIG_NP_V + lane) * a.tri_nk + node] : 0.0;
            const double en = sqrt(e2);
            const double ep = __shfl_up(en, 1, 64);
            const double eprev = lane > 0 ? ep : 0.0;
            double lo = lane < n ? dv - eprev - en : 1e300, hi = lane < n ? dv + eprev + en : -1e300;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lo = fmin(lo, __shfl_xor(lo, off, 64));
                hi = fmax(hi, __shfl_xor(hi, off, 64));
            }
            const double span = fmax(fabs(lo), fabs(hi));
            const double sc = span > 0.0 ? 1.0 / span : 1.0;
            {
                const double y = fmax(e2 * sc * sc, 4.9e-32);
                L.sd[lane] = lane < n ? dv * sc : 4.0;
                L.so[lane] = lane + 1 < n ? sqrt(y) : 0.0;
                if (lane < 2) {
                    L.sd[NM + lane] = 4.0;
                    L.so[NM + lane] = 0.0;
                }
                L.sbeta[lane] = lane < n ? kn[ncol + lane].x : 0.0;
            }
            // ---- phases of the complex subdiagonal: T = P T_real P^H, p_0 = 1, p_{k+1} = p_k t_k / |t_k|; t_k = -(x1 / |x1|) ||x|| and
            // the kept v1 = x1 (1 + ||x|| / |x1|) (or ||x|| when x1 = 0) has the phase of x1; the last coupling is the entry itself
            {
                double fr = 1.0, fi = 0.0;
                if (lane + 1 < n) {
                    const double2 v1 = kn[(int64_t)lane * n - (int64_t)lane * (lane + 1) / 2];
                    const double a2 = v1.x * v1.x + v1.y * v1.y;
                    const bool refl = lane + 2 < n;
                    const bool live = refl ? kn[ncol + lane].x != 0.0 : true;
                    if (live && a2 > 0.0) {
                        const double inv = 1.0 / sqrt(a2);
                        fr = (refl ? -v1.x : v1.x) * inv;
                        fi = (refl ? -v1.y : v1.y) * inv;
                    }
                }
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const double tr = __shfl_up(fr, off, 64), ti = __shfl_up(fi, off, 64);
                    if (lane >= off) {
                        const double nr = fr * tr - fi * ti, ni = fr * ti + fi * tr;
                        fr = nr;
                        fi = ni;
                    }
                }
                const double pr = __shfl_up(fr, 1, 64), pi = __shfl_up(fi, 1, 64);
                L.spr[lane] = lane == 0 ? 1.0 : pr;
                L.spi[lane] = lane == 0 ? 0.0 : pi;
            }
            vwave_sync();
            const double lam = lane < n ? a.E.base[vview_off(a.E, a.node0 + node) + (int64_t)lane * a.E.pitch] * sc : 2.0;
            // clusters: lane b is linked to lane b - 1 when their eigenvalues are within 1e-5 of the scale; pos = links below it
            const double lprev = __shfl(lam, lane > 0 ? lane - 1 : 0, 64);
            const bool link = lane > 0 && lane < n && (lam - lprev) <= 1e-5;
            const unsigned long long links = __builtin_amdgcn_ballot_w64(link);
            const unsigned long long below = (~links) & ((2ull << lane) - 1ull);
            const int pos = lane - (63 - __builtin_clzll(below));
            const double lamp = lam + 2.3e-15 * (double)pos;
            double z[NM];
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const unsigned h = (unsigned)(lane * 40503 + i * 30011 + 12345) * 2654435761u;
                const double v = (i < n) ? (double)((h >> 8) & 0xffffu) * (1.0 / 65536.0) + 0.25 : 0.0;
                z[i] = ((h >> 30) & 1u) ? -v : v;
            }
            if (a.phases & 1) {
                tri_fsolve(L.sd, L.so, n8, lamp, z);
                tri_fsolve(L.sd, L.so, n8, lamp, z);
                tri_fsolve(L.sd, L.so, n8, lamp, z);
            }
            unit2v(n8, z);
#pragma unroll
            for (int i = 0; i < NM; ++i) L.big[i * 64 + lane] = z[i];
            vwave_sync();
            // cluster members, lowest first: Gram-Schmidt against the members below (final by then), two more solves each
            for (int p = 1; __builtin_amdgcn_ballot_w64(pos >= p) != 0ull; ++p) {  // wave-uniform; not entered without a cluster
                for (int it = 0; it < 3; ++it) {
                    if (it > 0) tri_fsolve(L.sd, L.so, n8, lamp, z);
                    for (int t = 1; t <= p; ++t) {
                        const int src = lane - t >= 0 ? lane - t : 0;
                        double dot = 0.0;
#pragma unroll
                        for (int i = 0; i < NM; ++i) dot = fma(L.big[i * 64 + src], z[i], dot);
                        dot = (t <= pos) ? dot : 0.0;
#pragma unroll
                        for (int i = 0; i < NM; ++i) z[i] = fma(-dot, L.big[i * 64 + src], z[i]);
                    }
                    unit2v(n8, z);
                    vwave_sync();
                    if (pos == p) {
#pragma unroll
                        for (int i = 0; i < NM; ++i) L.big[i * 64 + lane] = z[i];
                    }
                    vwave_sync();
#pragma unroll
                    for (int i = 0; i < NM; ++i) z[i] = L.big[i * 64 + lane];  // (the others: their vector as it was)
                }
            }
        }
        __syncthreads();
        if (wv == 0)
            big_vec_wave<0>(a, L, lane, node, kn, Dm);
        else
            big_vec_wave<1>(a, L, lane, node, kn, Dm);
    }
}

}  // namespace

// keep: [node][n (n - 1) / 2 reflector components, column-packed | n x (beta, 0)]; Dm: the derivative matrices, direction j of
// node k at Dm[j dstride + k n n]
int launch_big_vec(abz_ctx* ctx, const double* tri, int64_t tri_nk, const double2* keep, const double2* Dm, int64_t dstride, int64_t node0,
                   int64_t nnodes, int n, int d, PlaneView E, PlaneView V) {
    if (nnodes <= 0) return ABZ_OK;
    BigVecArgs a;
    a.tri_nk = tri_nk;
    a.node0 = node0;
    a.nnodes = nnodes;
    a.dstride = dstride;
    a.n = n;
    a.d = d;
    {
        const char* ph = getenv("ABZ_BIG_VEC_PHASES");
        a.phases = ph ? atoi(ph) : 7;
    }
    a.E = E;
    a.V = V;
    const int64_t blocks = std::min<int64_t>(nnodes, 256 * 4 * 4);
    hipLaunchKernelGGL(big_ggr_kernel, dim3((unsigned)blocks), dim3(128), 0, ctx->stream, tri, keep, Dm, a);
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

}  // namespace abz
