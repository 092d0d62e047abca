// 33...64 bands (ref: src/fourier.jl:22-58 is generic in the matrix size; eigen(Hermitian(h)) goes through LAPACK there,
// src/dos_ggr.jl:19): the row layout of kernels_generic.hip ends at 32 lanes per node (two DPP rows, 512 registers for a
// 32 x 32 complex row block), so these sizes take a third route, generic in n at run time:
//   big_series_kernel      H(k) of a chunk of nodes into a scratch array [node][n * n] -- the level-1 evaluation
//                          [nodes x M] . [M x n^2] as plain FMAs (a tile of nodes shares every coefficient read) or, on full grid
//                          lines, as a real GEMM [16 nodes x 2M] . [2M x 16 columns] on v_mfma_f64_16x16x4_f64
//                          (ABZ_BIG_MFMA; measured in profiles/r05_big_series_mfma_vs_fma.txt -- f64 MFMA and f64 FMA share
//                          the DP units);
//   big_tridiag_kernel     Householder tridiagonalisation, ONE WAVE PER NODE, the lower triangle of the matrix packed in a
//                          wave-private LDS slab, lane r owning row r (the part right of the diagonal is read as the
//                          conjugate of column r);
//   big_qr_kernel          the eigenvalues of 64 tridiagonals per workgroup by the per-lane root-free QR iteration of
//                          tri_eig_kernel (big_bisect_kernel, one thread per (node, band), serves the AoS output);
//   big_trace_kernel       tr inv((w + i eta) I - H) = p'(z) / p(z) from the three-term recurrence of the tridiagonal, one
//                          thread per (node, swept value): node values (IAI) or weighted partial sums (rules, store-free sums).
// Serves abz_eval_nodes, rule builds (H and / or eigenvalues, full layout), scans of cached rules (DOS / tr G from the
// matrices or the eigenvalues), store-free PTR sums and the IAI node path; big_inverse_kernel (one workgroup per node, Gauss-
// Jordan in registers) adds matrix-valued G and the traces of series that are not Hermitian; GGR builds (eigenvalues + band
// velocities): launch_big_ggr below with kernels_big_vec.hip.  (The Hermitian-compact layout ends at 16 bands.)
#include <utility>

#include "abz_internal.h"
#include "rows_device.h"

namespace abz {

namespace {

constexpr int BIG_NP = 64;  // rows of the tridiagonal scratch [2 BIG_NP][tri_nk]
constexpr int BIG_TN = 8;   // nodes per tile of the FMA series kernel

static inline int64_t cdivb(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ int64_t bview_off(const PlaneView& v, int64_t k) {
    const int64_t line = k / v.line_len;
    return line * v.tile + (k - line * v.line_len);
}
__device__ __forceinline__ double bwsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double brl(double v, int lane) {  // lane: uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void bwave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct BigSeriesArgs {
    const double2* src;      // level-1 sets [item][M][n * n]
    const int64_t* parents;  // node -> item (nullptr: grid lines, item = node / npt)
    const int32_t* gi;       // node -> grid index of variable 1 (nullptr with grid: node % npt)
    const double* x;         // explicit coordinate of variable 1 (then gi is ignored)
    const double2* tab;      // e^{2 pi i j / npt}
    int64_t node0, nnodes;   // this chunk: nodes [node0, node0 + nnodes)
    int n, M, first, npt, grid;
    int deriv = 0;           // d/dx_1 in units of the period: coefficient f carries the factor 2 pi i f (JacobianSeries, src/dos_ggr.jl:18-20)
    double inv_period;
    double2* Hbuf;           // [nnodes][n * n], element (a, b) at a + n b like the coefficient blocks
};

__device__ __forceinline__ double2 big_phase(const BigSeriesArgs& a, int64_t k, int m) {
    const int f = a.first + m;
    double2 p;
    if (a.x) {
        double s, c;
        sincospi(2.0 * ((double)f * a.x[k] * a.inv_period), &s, &c);
        p = make_double2(c, s);
    } else {
        const int64_t i1 = a.gi ? (int64_t)a.gi[k] : k % a.npt;
        int64_t fm = f % a.npt;
        if (fm < 0) fm += a.npt;
        p = a.tab[(fm * i1) % a.npt];
    }
    if (a.deriv) {
        const double g = 6.283185307179586476925286766559 * (double)f;
        p = make_double2(-g * p.y, g * p.x);
    }
    return p;
}

// tile of BIG_TN consecutive nodes: they share the coefficient reads when they share their item (grid lines: a tile never
// crosses a line; node lists: checked per tile, otherwise node by node)
__global__ __launch_bounds__(256) void big_series_kernel(BigSeriesArgs a, int tiles_per_line) {
    __shared__ double2 ph[BIG_TN][64];  // [node][m] of a piece of <= 64 coefficients
    __shared__ int64_t item_s[BIG_TN];
    const int nn = a.n * a.n;
    int64_t t0;
    int cnt;
    if (a.grid) {
        const int64_t line = blockIdx.x / tiles_per_line;  // (of this chunk: node0 is a multiple of npt in grid mode)
        const int tl = (int)(blockIdx.x - line * tiles_per_line);
        t0 = line * a.npt + (int64_t)tl * BIG_TN;
        cnt = min(BIG_TN, a.npt - tl * BIG_TN);
    } else {
        t0 = (int64_t)blockIdx.x * BIG_TN;
        cnt = (int)min((int64_t)BIG_TN, a.nnodes - t0);
    }
    if (cnt <= 0) return;
    if (threadIdx.x < cnt) {
        const int64_t k = a.node0 + t0 + threadIdx.x;
        item_s[threadIdx.x] = a.parents ? a.parents[k] : (a.grid ? k / a.npt : 0);  // (node lists without parents: d = 1, one set)
    }
    // coefficients in pieces of 64 (the room of the phases): a later piece adds to what the earlier ones left in Hbuf
    for (int m0 = 0; m0 < a.M; m0 += 64) {
        const int mc = min(64, a.M - m0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * mc; t += 256) {
            const int j = t / mc, m = t - j * mc;
            ph[j][m] = big_phase(a, a.node0 + t0 + j, m0 + m);
        }
        __syncthreads();
        bool same = true;
        for (int j = 1; j < cnt; ++j) same = same && item_s[j] == item_s[0];
        if (same) {
            const double2* __restrict__ c1 = a.src + item_s[0] * ((int64_t)a.M * nn) + (int64_t)m0 * nn;
            for (int e = threadIdx.x; e < nn; e += 256) {
                double ar[BIG_TN], ai[BIG_TN];
#pragma unroll
                for (int j = 0; j < BIG_TN; ++j) {
                    const double2 h0 = (m0 > 0 && j < cnt) ? a.Hbuf[(t0 + j) * nn + e] : make_double2(0.0, 0.0);
                    ar[j] = h0.x;
                    ai[j] = h0.y;
                }
                for (int m = 0; m < mc; ++m) {
                    const double2 c = c1[(int64_t)m * nn + e];
#pragma unroll
                    for (int j = 0; j < BIG_TN; ++j) {
                        const double2 p = ph[j][m];
                        ar[j] = fma(c.x, p.x, ar[j]);
                        ar[j] = fma(-c.y, p.y, ar[j]);
                        ai[j] = fma(c.x, p.y, ai[j]);
                        ai[j] = fma(c.y, p.x, ai[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < BIG_TN; ++j)
                    if (j < cnt) a.Hbuf[(t0 + j) * nn + e] = make_double2(ar[j], ai[j]);
            }
        } else {
            for (int j = 0; j < cnt; ++j) {
                const double2* __restrict__ c1 = a.src + item_s[j] * ((int64_t)a.M * nn) + (int64_t)m0 * nn;
                for (int e = threadIdx.x; e < nn; e += 256) {
                    const double2 h0 = m0 > 0 ? a.Hbuf[(t0 + j) * nn + e] : make_double2(0.0, 0.0);
                    double ar = h0.x, ai = h0.y;
                    for (int m = 0; m < mc; ++m) {
                        const double2 c = c1[(int64_t)m * nn + e], p = ph[j][m];
                        ar = fma(c.x, p.x, ar);
                        ar = fma(-c.y, p.y, ar);
                        ai = fma(c.x, p.y, ai);
                        ai = fma(c.y, p.x, ai);
                    }
                    a.Hbuf[(t0 + j) * nn + e] = make_double2(ar, ai);
                }
            }
        }
    }
}

// The same on the matrix cores, full grid lines only: a workgroup takes 16 consecutive nodes of a line; wave w the column
// blocks e0 = 16 (w + 4 i) of the n^2 elements.  Real GEMM over k = (m, part):  Re H = sum_m c.re p.re - c.im p.im,
// Im H = sum_m c.re p.im + c.im p.re, i.e. two accumulations D_re, D_im that share the B operand (the coefficients).
// v_mfma_f64_16x16x4_f64 operands (one double per lane): A[i = lane % 16][k = lane / 16], B[k = lane / 16][j = lane % 16],
// D: four doubles per lane, D[i = lane / 16 + 4 r][j = lane % 16] (found by the parity test: the other blocking fails it).
typedef double bdouble4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void big_series_mfma_kernel(BigSeriesArgs a, int tiles_per_line) {
    __shared__ double2 ph[16][64];
    const int nn = a.n * a.n;
    const int64_t line = blockIdx.x / tiles_per_line;
    const int tl = (int)(blockIdx.x - line * tiles_per_line);
    const int64_t t0 = line * a.npt + (int64_t)tl * 16;
    const int cnt = min(16, a.npt - tl * 16);
    for (int t = threadIdx.x; t < 16 * a.M; t += 256) {
        const int j = t / a.M, m = t - j * a.M;
        ph[j][m] = j < cnt ? big_phase(a, a.node0 + t0 + j, m) : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const double2* __restrict__ c1 = a.src + ((a.node0 + t0) / a.npt) * ((int64_t)a.M * nn);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;  // operand row / column, k inside a group of four
    const int K = 2 * a.M;
    for (int e0 = 16 * wave; e0 < nn; e0 += 64) {
        bdouble4 dre = {0.0, 0.0, 0.0, 0.0}, dim = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < K; k0 += 4) {
            const int k = k0 + lk;            // this lane's k: coefficient m = k / 2, part = k % 2 (0: re, 1: im)
            const int m = k >> 1, part = k & 1;
            const bool in = k < K;
            const int mm = in ? m : 0;
            const double2 p = ph[li][mm];     // A operand: node li
            const double2 c = (e0 + li < nn) ? c1[(int64_t)mm * nn + e0 + li] : make_double2(0.0, 0.0);  // B operand: column li
            const double bv = in ? (part ? c.y : c.x) : 0.0;
            const double are = in ? (part ? -p.y : p.x) : 0.0;  // Re H: + c.re p.re - c.im p.im
            const double aim = in ? (part ? p.x : p.y) : 0.0;   // Im H: + c.re p.im + c.im p.re
#if defined(__HIP_DEVICE_COMPILE__)
            dre = __builtin_amdgcn_mfma_f64_16x16x4f64(are, bv, dre, 0, 0, 0);
            dim = __builtin_amdgcn_mfma_f64_16x16x4f64(aim, bv, dim, 0, 0, 0);
#endif
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int node = lk + 4 * r;
            if (node < cnt && e0 + li < nn) a.Hbuf[(t0 + node) * nn + e0 + li] = make_double2(dre[r], dim[r]);
        }
    }
}

// Hbuf -> the rule's planes (full layout: plane 2 (a + n b) + {0, 1}) and / or the AoS array of abz_eval_nodes
__global__ __launch_bounds__(256) void big_store_h_kernel(const double2* __restrict__ Hbuf, int64_t node0, int64_t nnodes, int n, PlaneView H,
                                                          double2* Haos) {
    const int nn = n * n;
    const int64_t total = (nnodes + 63) / 64 * 64 * nn;  // (groups of 64 nodes, the last one padded)
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        // plane-major inside a group of 64 nodes: consecutive threads write consecutive nodes of one plane
        const int64_t grp = t / (64 * nn);
        const int64_t rem = t - grp * (64 * nn);
        const int e = (int)(rem / 64);
        const int64_t node = grp * 64 + (rem & 63);
        if (node >= nnodes) continue;
        const double2 v = Hbuf[node * nn + e];
        if (H.base) {
            double* ho = H.base + bview_off(H, node0 + node);
            ho[(int64_t)(2 * e) * H.pitch] = v.x;
            ho[(int64_t)(2 * e + 1) * H.pitch] = v.y;
        }
        if (Haos) Haos[(node0 + node) * nn + e] = v;
    }
}

// a cached rule's planes -> Hbuf (scans of the matrices)
__global__ __launch_bounds__(256) void big_load_h_kernel(double2* __restrict__ Hbuf, int64_t node0, int64_t nnodes, int n, PlaneView H) {
    const int nn = n * n;
    const int64_t total = (nnodes + 63) / 64 * 64 * nn;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t grp = t / (64 * nn);
        const int64_t rem = t - grp * (64 * nn);
        const int e = (int)(rem / 64);
        const int64_t node = grp * 64 + (rem & 63);
        if (node >= nnodes) continue;
        const double* hi = H.base + bview_off(H, node0 + node);
        Hbuf[node * nn + e] = make_double2(hi[(int64_t)(2 * e) * H.pitch], hi[(int64_t)(2 * e + 1) * H.pitch]);
    }
}

// Householder tridiagonalisation of Hermitian(h) (the upper triangle of each n x n block of Hbuf, like the reference's
// Hermitian wrapper): d_j -> tri[j][t], |e_j|^2 -> tri[BIG_NP + j][t], t = t0 + node.  One wave per node; lane i owns row i of
// the LOWER triangle, packed row by row (L(i, j), j <= i, at i (i + 1) / 2 + j): half the LDS of a full matrix -- twice the
// waves per CU under these latency-bound row loops -- and half the rank-2 update; the part of row i right of the diagonal is
// read as the conjugate of column i (consecutive lanes, consecutive addresses).
// `keep` (GGR builds, kernels_big_vec.hip): per node what the steps leave below the diagonal, column by column -- column k is the
// reflector v_k (its first component v_k[k + 1] replaces the eliminated entry; component i at k n - k (k + 1) / 2 + i - k - 1) --
// then n pairs (beta_k, 0): H_k = I - beta_k v_k v_k^H.
__global__ __launch_bounds__(64) void big_tridiag_kernel(const double2* __restrict__ Hbuf, int64_t nnodes, int n, double* __restrict__ tri,
                                                         int64_t tri_nk, int64_t t0, double2* __restrict__ keep) {
    extern __shared__ double2 lds_bt[];  // L [n (n + 1) / 2], then v [n], q [n]
    const int np = n * (n + 1) / 2;
    double2* __restrict__ const A = lds_bt;
    double2* __restrict__ const vv = lds_bt + np;
    double2* __restrict__ const qq = vv + n;
    const int lane = threadIdx.x, i = lane;
    const bool row = i < n;
    const int ri = i * (i + 1) / 2;  // start of this lane's row
    for (int64_t node = blockIdx.x; node < nnodes; node += gridDim.x) {
        const double2* __restrict__ h = Hbuf + node * ((int64_t)n * n);
        bwave_sync();
        for (int e = lane; e < np; e += 64) {
            // packed index e -> (r, c), c <= r
            int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
            while ((r + 1) * (r + 2) / 2 <= e) ++r;
            while (r * (r + 1) / 2 > e) --r;
            const int c = e - r * (r + 1) / 2;
            double2 v = h[c + n * r];  // H(c, r), c <= r: the upper triangle; L(r, c) = conj
            v.y = r == c ? 0.0 : -v.y;
            A[e] = v;
        }
        bwave_sync();
        for (int k = 0; k + 1 < n; ++k) {
            const bool below = row && i > k;
            const double2 x = below ? A[ri + k] : make_double2(0.0, 0.0);
            const double sigma = bwsum(x.x * x.x + x.y * x.y);
            if (lane == k) {
                tri[(int64_t)k * tri_nk + t0 + node] = A[ri + k].x;
                tri[(int64_t)(BIG_NP + k) * tri_nk + t0 + node] = sigma;
            }
            if (k + 2 >= n || !(sigma > 0.0)) {  // uniform
                if (keep && lane == 0) keep[node * (int64_t)np + (np - n) + k] = make_double2(0.0, 0.0);
                continue;
            }
            const double x1r = __shfl(x.x, k + 1, 64), x1i = __shfl(x.y, k + 1, 64);
            const double a1sq = x1r * x1r + x1i * x1i;
            // v = x + (x1 / |x1|) ||x|| e_1, beta = 1 / (sigma + ||x|| |x1|)
            const double nrm = sqrt(sigma), a1 = sqrt(a1sq);
            double v1r, v1i;
            if (a1 > 0.0) {
                const double fac = 1.0 + nrm / a1;
                v1r = x1r * fac;
                v1i = x1i * fac;
            } else {
                v1r = nrm;
                v1i = 0.0;
            }
            const double beta = 1.0 / (sigma + nrm * a1);
            const double2 v = (i == k + 1) ? make_double2(v1r, v1i) : x;  // zero in rows <= k and >= n
            if (row) vv[i] = v;
            bwave_sync();
            // p_i = beta sum_{j > k} A(i, j) v_j:  A(i, j) = L(i, j) for j <= i, conj L(j, i) for j > i
            double pr = 0.0, pi = 0.0;
            if (below) {
                double pr2 = 0.0, pi2 = 0.0;  // (two chains: the FMAs of consecutive j do not wait for one another)
                int j = k + 1;
#pragma unroll 2
                for (; j + 1 < n; j += 2) {
                    const bool own0 = j <= i, own1 = j + 1 <= i;
                    double2 a0 = A[own0 ? ri + j : j * (j + 1) / 2 + i], a1 = A[own1 ? ri + j + 1 : (j + 1) * (j + 2) / 2 + i];
                    a0.y = own0 ? a0.y : -a0.y;
                    a1.y = own1 ? a1.y : -a1.y;
                    const double2 v0 = vv[j], v1 = vv[j + 1];
                    pr = fma(a0.x, v0.x, pr);
                    pr2 = fma(a1.x, v1.x, pr2);
                    pi = fma(a0.x, v0.y, pi);
                    pi2 = fma(a1.x, v1.y, pi2);
                    pr = fma(-a0.y, v0.y, pr);
                    pr2 = fma(-a1.y, v1.y, pr2);
                    pi = fma(a0.y, v0.x, pi);
                    pi2 = fma(a1.y, v1.x, pi2);
                }
                if (j < n) {
                    const bool own = j <= i;
                    double2 aij = A[own ? ri + j : j * (j + 1) / 2 + i];
                    aij.y = own ? aij.y : -aij.y;
                    const double2 vj = vv[j];
                    pr = fma(aij.x, vj.x, pr);
                    pr = fma(-aij.y, vj.y, pr);
                    pi = fma(aij.x, vj.y, pi);
                    pi = fma(aij.y, vj.x, pi);
                }
                pr += pr2;
                pi += pi2;
            }
            pr *= beta;
            pi *= beta;
            // kappa = (beta / 2) v^H p;  q = p - kappa v
            const double kr = 0.5 * beta * bwsum(v.x * pr + v.y * pi);
            const double ki = 0.5 * beta * bwsum(v.x * pi - v.y * pr);
            const double qr = pr - (kr * v.x - ki * v.y), qi = pi - (kr * v.y + ki * v.x);
            if (row) qq[i] = make_double2(qr, qi);
            if (keep && i == k + 1) {
                A[ri + k] = make_double2(v1r, v1i);
                keep[node * (int64_t)np + (np - n) + k] = make_double2(beta, 0.0);
            }
            bwave_sync();
            // L(i, j) -= v_i conj(q_j) + q_i conj(v_j), k < j <= i (this lane's own row)
            if (below) {
#pragma unroll 4
                for (int j = k + 1; j <= i; ++j) {
                    const double2 vj = vv[j], qj = qq[j];
                    double2 aij = A[ri + j];
                    aij.x -= (v.x * qj.x + v.y * qj.y) + (qr * vj.x + qi * vj.y);
                    aij.y -= (v.y * qj.x - v.x * qj.y) + (qi * vj.x - qr * vj.y);
                    A[ri + j] = aij;
                }
            }
            bwave_sync();
        }
        if (lane == n - 1) {
            tri[(int64_t)(n - 1) * tri_nk + t0 + node] = A[ri + (n - 1)].x;
            tri[(int64_t)(BIG_NP + n - 1) * tri_nk + t0 + node] = 0.0;
            if (keep) keep[node * (int64_t)np + (np - n) + (n - 1)] = make_double2(0.0, 0.0);
        }
        if (keep) {
            bwave_sync();
            double2* __restrict__ ko = keep + node * (int64_t)np;  // n (n - 1) / 2 components + n betas = np numbers per node
            for (int e = lane; e < np; e += 64) {
                int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                while ((r + 1) * (r + 2) / 2 <= e) ++r;
                while (r * (r + 1) / 2 > e) --r;
                const int c = e - r * (r + 1) / 2;
                if (c < r) ko[c * n - c * (c + 1) / 2 + (r - c - 1)] = A[e];
            }
        }
    }
}

// The same with NW waves per node (a workgroup of 64 NW threads; lane i of every wave stands for row i): the sums over j of a step
// -- p = A v and the rank-2 update of a row -- are dealt to the waves by j mod NW, the partial p meet in LDS.  Four block barriers
// per step instead of three wave barriers, but NW times the waves in flight per matrix held in LDS: one wave per node leaves a CU
// with four waves at 64 bands (33 KB per matrix), all of them waiting on their own LDS round trips.
template <int NW>
__global__ __launch_bounds__(64 * NW) void big_tridiag_mw_kernel(const double2* __restrict__ Hbuf, int64_t nnodes, int n, double* __restrict__ tri,
                                                                 int64_t tri_nk, int64_t t0, double2* __restrict__ keep) {
    extern __shared__ double2 lds_bt[];  // L [n (n + 1) / 2], then v [n], q [n], partial p [NW][64]
    const int np = n * (n + 1) / 2;
    double2* __restrict__ const A = lds_bt;
    double2* __restrict__ const vv = lds_bt + np;
    double2* __restrict__ const qq = vv + n;
    double2* __restrict__ const part = qq + n;
    const int tid = threadIdx.x, lane = tid & 63, i = lane, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool row = i < n;
    const int ri = i * (i + 1) / 2;  // start of this lane's row
    for (int64_t node = blockIdx.x; node < nnodes; node += gridDim.x) {
        const double2* __restrict__ h = Hbuf + node * ((int64_t)n * n);
        __syncthreads();
        for (int e = tid; e < np; e += 64 * NW) {
            int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
            while ((r + 1) * (r + 2) / 2 <= e) ++r;
            while (r * (r + 1) / 2 > e) --r;
            const int c = e - r * (r + 1) / 2;
            double2 v = h[c + n * r];  // H(c, r), c <= r: the upper triangle; L(r, c) = conj
            v.y = r == c ? 0.0 : -v.y;
            A[e] = v;
        }
        __syncthreads();
        for (int k = 0; k + 1 < n; ++k) {
            const bool below = row && i > k;
            const double2 x = below ? A[ri + k] : make_double2(0.0, 0.0);  // (every wave: the same values)
            const double sigma = bwsum(x.x * x.x + x.y * x.y);
            if (w == 0 && lane == k) {
                tri[(int64_t)k * tri_nk + t0 + node] = A[ri + k].x;
                tri[(int64_t)(BIG_NP + k) * tri_nk + t0 + node] = sigma;
            }
            if (k + 2 >= n || !(sigma > 0.0)) {  // uniform in the workgroup
                if (keep && tid == 0) keep[node * (int64_t)np + (np - n) + k] = make_double2(0.0, 0.0);
                continue;
            }
            const double x1r = __shfl(x.x, k + 1, 64), x1i = __shfl(x.y, k + 1, 64);
            const double a1sq = x1r * x1r + x1i * x1i;
            const double nrm = sqrt(sigma), a1 = sqrt(a1sq);
            double v1r, v1i;
            if (a1 > 0.0) {
                const double fac = 1.0 + nrm / a1;
                v1r = x1r * fac;
                v1i = x1i * fac;
            } else {
                v1r = nrm;
                v1i = 0.0;
            }
            const double beta = 1.0 / (sigma + nrm * a1);
            const double2 v = (i == k + 1) ? make_double2(v1r, v1i) : x;  // zero in rows <= k and >= n
            if (w == 0 && row) vv[i] = v;
            __syncthreads();
            double pr = 0.0, pi = 0.0;
            if (below) {
                double pr2 = 0.0, pi2 = 0.0;  // (two chains)
                int j = k + 1 + w;
                for (; j + NW < n; j += 2 * NW) {
                    const int j1 = j + NW;
                    const bool own0 = j <= i, own1 = j1 <= i;
                    double2 a0 = A[own0 ? ri + j : j * (j + 1) / 2 + i], a1 = A[own1 ? ri + j1 : j1 * (j1 + 1) / 2 + i];
                    a0.y = own0 ? a0.y : -a0.y;
                    a1.y = own1 ? a1.y : -a1.y;
                    const double2 v0 = vv[j], v1 = vv[j1];
                    pr = fma(a0.x, v0.x, pr);
                    pr2 = fma(a1.x, v1.x, pr2);
                    pi = fma(a0.x, v0.y, pi);
                    pi2 = fma(a1.x, v1.y, pi2);
                    pr = fma(-a0.y, v0.y, pr);
                    pr2 = fma(-a1.y, v1.y, pr2);
                    pi = fma(a0.y, v0.x, pi);
                    pi2 = fma(a1.y, v1.x, pi2);
                }
                if (j < n) {
                    const bool own = j <= i;
                    double2 aij = A[own ? ri + j : j * (j + 1) / 2 + i];
                    aij.y = own ? aij.y : -aij.y;
                    const double2 vj = vv[j];
                    pr = fma(aij.x, vj.x, pr);
                    pr = fma(-aij.y, vj.y, pr);
                    pi = fma(aij.x, vj.y, pi);
                    pi = fma(aij.y, vj.x, pi);
                }
                pr += pr2;
                pi += pi2;
            }
            part[w * 64 + lane] = make_double2(pr, pi);
            __syncthreads();
            pr = 0.0;
            pi = 0.0;
#pragma unroll
            for (int u = 0; u < NW; ++u) {  // (the same order in every wave: the same q everywhere)
                pr += part[u * 64 + lane].x;
                pi += part[u * 64 + lane].y;
            }
            pr *= beta;
            pi *= beta;
            const double kr = 0.5 * beta * bwsum(v.x * pr + v.y * pi);
            const double ki = 0.5 * beta * bwsum(v.x * pi - v.y * pr);
            const double qr = pr - (kr * v.x - ki * v.y), qi = pi - (kr * v.y + ki * v.x);
            if (w == 0 && row) qq[i] = make_double2(qr, qi);
            if (keep && w == 0 && i == k + 1) {
                A[ri + k] = make_double2(v1r, v1i);
                keep[node * (int64_t)np + (np - n) + k] = make_double2(beta, 0.0);
            }
            __syncthreads();
            if (below) {
#pragma unroll 2
                for (int j = k + 1 + w; j <= i; j += NW) {
                    const double2 vj = vv[j], qj = qq[j];
                    double2 aij = A[ri + j];
                    aij.x -= (v.x * qj.x + v.y * qj.y) + (qr * vj.x + qi * vj.y);
                    aij.y -= (v.y * qj.x - v.x * qj.y) + (qi * vj.x - qr * vj.y);
                    A[ri + j] = aij;
                }
            }
            __syncthreads();
        }
        if (w == 0 && lane == n - 1) {
            tri[(int64_t)(n - 1) * tri_nk + t0 + node] = A[ri + (n - 1)].x;
            tri[(int64_t)(BIG_NP + n - 1) * tri_nk + t0 + node] = 0.0;
            if (keep) keep[node * (int64_t)np + (np - n) + (n - 1)] = make_double2(0.0, 0.0);
        }
        if (keep) {
            __syncthreads();
            double2* __restrict__ ko = keep + node * (int64_t)np;
            for (int e = tid; e < np; e += 64 * NW) {
                int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                while ((r + 1) * (r + 2) / 2 <= e) ++r;
                while (r * (r + 1) / 2 > e) --r;
                const int c = e - r * (r + 1) / 2;
                if (c < r) ko[c * n - c * (c + 1) / 2 + (r - c - 1)] = A[e];
            }
        }
    }
}

// eigenvalue `band` (ascending) of each tridiagonal: bisection on the Sturm count (ratio form with LAPACK's pivot guard),
// 4 nodes x 64 bands per workgroup
__global__ __launch_bounds__(256) void big_bisect_kernel(const double* __restrict__ tri, int64_t tri_nk, int64_t t0, int64_t node0, int64_t nnodes,
                                                         int n, PlaneView E, double* __restrict__ Eaos) {
    __shared__ double ds[4][BIG_NP], es[4][BIG_NP];
    const int slot = threadIdx.x >> 6, band = threadIdx.x & 63;
    const int64_t node = (int64_t)blockIdx.x * 4 + slot;
    const bool act = node < nnodes;
    const int64_t kk = act ? node : nnodes - 1;
    ds[slot][band] = band < n ? tri[(int64_t)band * tri_nk + t0 + kk] : 0.0;
    es[slot][band] = band + 1 < n ? tri[(int64_t)(BIG_NP + band) * tri_nk + t0 + kk] : 0.0;
    __syncthreads();
    const double* __restrict__ d = ds[slot];
    const double* __restrict__ e2 = es[slot];
    double lo = d[0], hi = d[0], eprev = 0.0, emax = 0.0;
    for (int j = 0; j < n; ++j) {
        const double en = j + 1 < n ? sqrt(e2[j]) : 0.0;
        lo = fmin(lo, d[j] - eprev - en);
        hi = fmax(hi, d[j] + eprev + en);
        emax = fmax(emax, e2[j]);
        eprev = en;
    }
    const double span = fmax(fabs(lo), fabs(hi));
    lo -= 2.3e-16 * span * n + 4.9e-324;
    hi += 2.3e-16 * span * n + 4.9e-324;
    const double pivmin = fmax(2.3e-308 * fmax(emax, 1.0), 4.9e-324);
    const int want = band < n ? band : n - 1;
    for (int it = 0; it < 120; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (!(mid > lo && mid < hi)) break;
        int cnt = 0;
        double q = d[0] - mid;
        if (fabs(q) < pivmin) q = -pivmin;
        cnt += q < 0.0;
        for (int j = 1; j < n; ++j) {
            q = d[j] - mid - e2[j - 1] / q;
            if (fabs(q) < pivmin) q = -pivmin;
            cnt += q < 0.0;
        }
        if (cnt > want)
            hi = mid;
        else
            lo = mid;
    }
    if (act && band < n) {
        const double ev = 0.5 * (lo + hi);
        if (E.base) E.base[bview_off(E, node0 + node) + (int64_t)band * E.pitch] = ev;
        if (Eaos) Eaos[(node0 + node) * n + band] = ev;
    }
}

// The eigenvalues of 64 tridiagonals per workgroup by the per-lane root-free QR iteration (tri_qr_lane, rows_device.h: what
// tri_eig_kernel runs for <= 32 bands) -- O(n^2) per matrix where the bisection above spends O(n^2) per EIGENVALUE; sorted by
// an odd-even transposition in the lane's LDS column.  Rule planes only (the AoS output keeps the bisection kernel).
__global__ __launch_bounds__(64) void big_qr_kernel(const double* __restrict__ tri, int64_t tri_nk, int64_t node0, int64_t nnodes, int n, PlaneView E) {
    extern __shared__ double lds_bq[];  // ld [n + 2][64] | le [n + 2][64]
    double(*const ld)[64] = reinterpret_cast<double(*)[64]>(lds_bq);
    double(*const le)[64] = ld + (n + 2);
    const int lane = threadIdx.x;
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;
    const bool act = k < nnodes;
    const int64_t kk = act ? k : nnodes - 1;
    double anorm2 = 0.0;
    for (int j = 0; j < n + 2; ++j) {
        const double dj = j < n ? tri[(int64_t)j * tri_nk + kk] : 0.0;
        const double ej = j + 1 < n ? tri[(int64_t)(BIG_NP + j) * tri_nk + kk] : 0.0;
        ld[j][lane] = dj;
        le[j][lane] = ej;
        anorm2 = fmax(anorm2, fmax(dj * dj, ej));
    }
    const int left = tri_qr_lane(ld, le, n, lane, anorm2);
    for (int pass = 0; pass < n; ++pass)
        for (int j = pass & 1; j + 1 < n; j += 2) {
            const double a = ld[j][lane], b = ld[j + 1][lane];
            ld[j][lane] = fmin(a, b);
            ld[j + 1][lane] = fmax(a, b);
        }
    if (act) {
        double* __restrict__ eo = E.base + bview_off(E, node0 + k);
        for (int j = 0; j < n; ++j) eo[(int64_t)j * E.pitch] = left > 0 ? __builtin_nan("") : ld[j][lane];
    }
}

// tr inv((w + i eta) I - H) from the tridiagonal: p_0 = 1, p_1 = z - d_0, p_{j+1} = (z - d_j) p_j - |e_{j-1}|^2 p_{j-1}, the
// same recurrence for p'; scaled to unit Gershgorin radius (rows_trace_resolvent_tri of rows_device.h does the same for <= 32)
struct BigTraceArgs {
    const double* tri;
    int64_t tri_nk, t0, node0, nnodes;
    int n, n_sweep, is_dos;
    double eta;
    const double* sweep;           // device [n_sweep] (or null: sweep0)
    const double* sweep_per_node;  // device [all nodes]: one value per node
    double sweep0;
    const double* w;               // weights of the nodes (sum mode) or null
    double2* values;               // node mode: [node][n_sweep]
    double2* partial;              // sum mode: [gridDim.x][n_sweep]
};
__global__ __launch_bounds__(256) void big_trace_kernel(BigTraceArgs a) {
    __shared__ double red[2][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int s = 0; s < a.n_sweep; ++s) {
        double accr = 0.0, acci = 0.0;
        for (int64_t node = (int64_t)blockIdx.x * 256 + threadIdx.x; node < a.nnodes; node += (int64_t)gridDim.x * 256) {
            const double sw = a.sweep_per_node ? a.sweep_per_node[a.node0 + node] : (a.sweep ? a.sweep[s] : a.sweep0);
            const double* __restrict__ d = a.tri + a.t0 + node;
            double rad = 0.0, eprev = 0.0;
            for (int j = 0; j < a.n; ++j) {
                const double en = j + 1 < a.n ? sqrt(d[(int64_t)(BIG_NP + j) * a.tri_nk]) : 0.0;
                rad = fmax(rad, fabs(d[(int64_t)j * a.tri_nk]) + eprev + en);
                eprev = en;
            }
            const double sc = 1.0 / (rad + fabs(sw) + a.eta);
            const double zr = sw * sc, zi = a.eta * sc;
            double p0r = 1.0, p0i = 0.0, p1r = zr - d[0] * sc, p1i = zi;
            double q0r = 0.0, q0i = 0.0, q1r = 1.0, q1i = 0.0;
            for (int j = 1; j < a.n; ++j) {
                const double ar = zr - d[(int64_t)j * a.tri_nk] * sc, ai = zi;
                const double ee = d[(int64_t)(BIG_NP + j - 1) * a.tri_nk] * sc * sc;
                const double npr = fma(ar, p1r, fma(-ai, p1i, -ee * p0r));
                const double npi = fma(ar, p1i, fma(ai, p1r, -ee * p0i));
                const double nqr = p1r + fma(ar, q1r, fma(-ai, q1i, -ee * q0r));
                const double nqi = p1i + fma(ar, q1i, fma(ai, q1r, -ee * q0i));
                p0r = p1r;
                p0i = p1i;
                p1r = npr;
                p1i = npi;
                q0r = q1r;
                q0i = q1i;
                q1r = nqr;
                q1i = nqi;
            }
            const double ip = sc / (p1r * p1r + p1i * p1i);
            double tr = (q1r * p1r + q1i * p1i) * ip, ti = (q1i * p1r - q1r * p1i) * ip;
            if (a.is_dos) {
                tr = -ti * 0.31830988618379067153776752674503;
                ti = 0.0;
            }
            if (a.values) {
                a.values[(a.node0 + node) * a.n_sweep + s] = make_double2(tr, ti);
            } else {
                const double wk = a.w ? a.w[a.node0 + node] : 1.0;
                accr = fma(wk, tr, accr);
                acci = fma(wk, ti, acci);
            }
        }
        if (a.partial) {
            __syncthreads();
            accr = bwsum(accr);
            acci = bwsum(acci);
            if (lane == 0) {
                red[0][wave] = accr;
                red[1][wave] = acci;
            }
            __syncthreads();
            if (threadIdx.x == 0)
                a.partial[(int64_t)blockIdx.x * a.n_sweep + s] =
                    make_double2((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
        }
    }
}

__global__ __launch_bounds__(256) void big_accumulate_kernel(double2* __restrict__ total, const double2* __restrict__ part, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) {
        total[t].x += part[t].x;
        total[t].y += part[t].y;
    }
}

// inv((w + i eta) I - H) of a node, ONE WORKGROUP OF FOUR WAVES PER NODE (two nodes up to 32 bands), the matrix in registers,
// Gauss-Jordan in place without pivoting (what the <= 32-band kernels do: for Hermitian H the matrix is eta I plus a skew-
// Hermitian part times i -- no small pivots; a general H gets no more here than there); big_inv_pivot4 below has the step.
// Serves what the tridiagonal cannot: matrix-valued G (ABZ_F_GLOC) and the traces of series that are not Hermitian.  Node mode:
// values[node][swept value][component]; sum mode: every workgroup adds the weighted values of its nodes (registers) and leaves
// partial[block][swept value][component] for launch_final_reduce.
struct BigInvArgs {
    const double2* Hbuf;  // [nnodes][n n] of this chunk
    int64_t node0, nnodes;
    int n, n_sweep, kind;  // kind 0: G (n n components), 1: tr G, 2: DOS = -Im tr G / pi
    double eta;
    const double* sweep;           // device [n_sweep] (or null: sweep0)
    const double* sweep_per_node;  // device [all nodes]
    double sweep0;
    const double* w;      // sum mode: weights of all nodes (null: 1)
    double2* values;      // node mode
    double2* partial;     // sum mode, ACCUMULATED over chunks (zeroed by the caller)
    int nodes_per_block;  // sum mode
};
// One block of four pivots c = 4 CQ ... 4 CQ + 3 of the register-resident Gauss-Jordan (big_inverse_kernel).  Column c -- the
// f = A(r, c) of every row -- lives in wave c & 3 and reaches the other waves through LDS (two rooms in turn, one barrier per
// pivot); row c -- A(c, j) of this wave's sixteen columns -- lives in lane c of THIS wave: v_readlane, scalar operands of the FMAs
// (SUBS = 2, 4, 8: that many nodes of <= 32, 16, 8 bands side by side in every wave: lane c of the lane's own part, a shuffle).
template <int NQ, int CQ, int SUBS>
__device__ __forceinline__ void big_inv_pivot4(int n, int lane, int r, int jq, double2 (*colb)[64], double2 (&W)[NQ]) {
    constexpr int LW = 64 / SUBS;
#pragma unroll 1
    for (int cm = 0; cm < 4; ++cm) {
        const int c = 4 * CQ + cm;
        if (c >= n) break;  // uniform
        const int pb = c & 1;
        if (jq == cm) colb[pb][lane] = W[CQ];
        __syncthreads();
        const int lc = (lane & ~(LW - 1)) + c;  // the lane that holds row c of this lane's node
        const double2 p = colb[pb][lc], f = colb[pb][lane];
        const double ipn = 1.0 / (p.x * p.x + p.y * p.y);
        const double ipr = p.x * ipn, ipi = -p.y * ipn;  // 1 / pivot
        const bool prow = r == c;
        const double qr = f.x * ipr - f.y * ipi, qi = f.x * ipi + f.y * ipr;  // f / pivot
        // one straight line for every entry: A(r, j) - (f / p) A(c, j); in the pivot row itself 0 + (1 / p) A(c, j)
        const double fr = prow ? -ipr : qr, fi = prow ? -ipi : qi;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double gx, gy;  // A(c, jq + 4 q) before the step
            if constexpr (SUBS == 1) {
                gx = brl(W[q].x, c);
                gy = brl(W[q].y, c);
            } else {
                gx = __shfl(W[q].x, lc, 64);
                gy = __shfl(W[q].y, lc, 64);
            }
            const double ox = prow ? 0.0 : W[q].x, oy = prow ? 0.0 : W[q].y;
            W[q] = make_double2(ox - (fr * gx - fi * gy), oy - (fr * gy + fi * gx));
        }
        if (jq == cm) W[CQ] = prow ? make_double2(ipr, ipi) : make_double2(-qr, -qi);  // the pivot column: -f / p, the pivot 1 / p
    }
}
template <int NQ, int SUBS, int... CQ>
__device__ __forceinline__ void big_inv_pivots(int n, int lane, int r, int jq, double2 (*colb)[64], double2 (&W)[NQ],
                                               std::integer_sequence<int, CQ...>) {
    ((void)((4 * CQ < n) ? (big_inv_pivot4<NQ, CQ, SUBS>(n, lane, r, jq, colb, W), 0) : 0), ...);
}

// NQ: column groups of four the instance holds (n <= 4 NQ); SUBS: nodes side by side in a wave (2: n <= 32, rows in 32 lanes)
template <int NQ, int SUBS>
__global__ __launch_bounds__(256) void big_inverse_kernel(BigInvArgs a) {
    // The matrix lives in REGISTERS: lane r of wave jq holds the entries (r, jq + 4 q), q < NQ (big_inv_pivot4 above).
    static_assert(4 * NQ <= 64 / SUBS, "SUBS nodes per wave: 64 / SUBS lanes of rows each");
    constexpr int LW = 64 / SUBS;
    __shared__ double2 colb[2][64];
    __shared__ double2 trb[4][8];
    const int n = a.n, nn = n * n, tid = threadIdx.x;
    const int lane = tid & 63, r = lane & (LW - 1), sub = lane / LW, jq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool rowon = r < n;
    const int ncomp = a.kind == 0 ? nn : 1;
    const bool sum = a.partial != nullptr;
    const int64_t k0 = sum ? (int64_t)blockIdx.x * a.nodes_per_block : blockIdx.x;
    const int64_t k1 = sum ? min(a.nnodes, k0 + a.nodes_per_block) : a.nnodes;
    const int64_t kstep = sum ? 1 : gridDim.x;
    for (int s = 0; s < a.n_sweep; ++s) {
        double2 acc[NQ];  // sum mode: the weighted sum of this thread's entries, or the trace in acc[0] of the node's first lane
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = make_double2(0.0, 0.0);
        for (int64_t kb = k0; kb < k1; kb += kstep * SUBS) {
            const int64_t k = kb + sub * kstep;
            const bool valid = k < k1;  // (the second half of a wave may be without a node: it inverts z I)
            const int64_t kk = valid ? k : kb;
            const double sw = a.sweep_per_node ? a.sweep_per_node[a.node0 + kk] : (a.sweep ? a.sweep[s] : a.sweep0);
            const double2* __restrict__ h = a.Hbuf + kk * (int64_t)nn;
            double2 W[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int j = jq + 4 * q;
                W[q] = make_double2(0.0, 0.0);
                if (rowon && j < n) {
                    const double2 hv = valid ? h[r + n * j] : make_double2(0.0, 0.0);
                    W[q] = make_double2((r == j ? sw : 0.0) - hv.x, (r == j ? a.eta : 0.0) - hv.y);
                }
            }
            // pivots c = 4 cq + cm: cq unrolled (the entry of column c in its wave's registers is W[cq], a static index), cm rolled
            big_inv_pivots<NQ, SUBS>(n, lane, r, jq, colb, W, std::make_integer_sequence<int, NQ>());
            const double wk = (sum && valid) ? (a.w ? a.w[a.node0 + k] : 1.0) : (valid ? 1.0 : 0.0);
            if (a.kind == 0) {
                if (sum) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        acc[q].x = fma(wk, W[q].x, acc[q].x);
                        acc[q].y = fma(wk, W[q].y, acc[q].y);
                    }
                } else if (rowon && valid) {
                    double2* __restrict__ vo = a.values + ((a.node0 + k) * a.n_sweep + s) * (int64_t)nn;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int j = jq + 4 * q;
                        if (j < n) vo[r + n * j] = W[q];
                    }
                }
            } else {
                // the diagonal entry (r, r) sits in wave r & 3 at q = r >> 2: partial traces per wave and node, met in LDS
                double tr = 0.0, ti = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const bool d = rowon && (jq + 4 * q) == r;
                    tr += d ? W[q].x : 0.0;
                    ti += d ? W[q].y : 0.0;
                }
#pragma unroll
                for (int off = LW / 2; off > 0; off >>= 1) {
                    tr += __shfl_xor(tr, off, 64);
                    ti += __shfl_xor(ti, off, 64);
                }
                if (r == 0) trb[jq][sub] = make_double2(tr, ti);
                __syncthreads();
                if (jq == 0 && r == 0) {
                    tr = (trb[0][sub].x + trb[1][sub].x) + (trb[2][sub].x + trb[3][sub].x);
                    ti = (trb[0][sub].y + trb[1][sub].y) + (trb[2][sub].y + trb[3][sub].y);
                    if (a.kind == 2) {
                        tr = -ti * 0.31830988618379067153776752674503;
                        ti = 0.0;
                    }
                    if (sum) {
                        acc[0].x = fma(wk, tr, acc[0].x);
                        acc[0].y = fma(wk, ti, acc[0].y);
                    } else if (valid) {
                        a.values[(a.node0 + k) * a.n_sweep + s] = make_double2(tr, ti);
                    }
                }
                __syncthreads();
            }
        }
        if (sum) {
            double2* __restrict__ po = a.partial + ((int64_t)blockIdx.x * a.n_sweep + s) * ncomp;
            if constexpr (SUBS > 1) {  // the nodes' sums of the same entries
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
#pragma unroll
                    for (int off = LW; off < 64; off <<= 1) {
                        acc[q].x += __shfl_xor(acc[q].x, off, 64);
                        acc[q].y += __shfl_xor(acc[q].y, off, 64);
                    }
                }
            }
            if (a.kind == 0) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int j = jq + 4 * q;
                    if (rowon && sub == 0 && j < n) {
                        po[r + n * j].x += acc[q].x;
                        po[r + n * j].y += acc[q].y;
                    }
                }
            } else if (tid == 0) {
                po[0].x += acc[0].x;
                po[0].y += acc[0].y;
            }
        }
    }
}

struct BigWork {
    double2* Hbuf = nullptr;
    double* tri = nullptr;
    int64_t chunk = 0, tri_nk = 0;
};

int big_reserve(abz_ctx* ctx, int n, int npt_or_zero, int64_t nnodes, BigWork& w) {
    // chunks of nodes: <= 256 MB of matrices, whole grid lines in grid mode
    const int64_t mb = abz_switch(SW_BIG_CHUNK_MB) > 0 ? abz_switch(SW_BIG_CHUNK_MB) : 256;
    int64_t c = std::max<int64_t>(1, (mb << 20) / ((int64_t)sizeof(double2) * n * n));
    if (npt_or_zero > 0) c = std::max<int64_t>(1, c / npt_or_zero) * npt_or_zero;
    c = std::min(c, nnodes);
    w.chunk = c;
    w.tri_nk = (c + 63) / 64 * 64;
    int rc = ctx->scratch[6].reserve(sizeof(double2) * (size_t)c * n * n);
    if (rc) return rc;
    if ((rc = ctx->scratch[4].reserve(sizeof(double) * (size_t)(2 * BIG_NP) * (size_t)w.tri_nk))) return rc;
    w.Hbuf = ctx->scratch[6].as<double2>();
    w.tri = ctx->scratch[4].as<double>();
    return ABZ_OK;
}

// (tri_to / tri_nk_to / t0: write the tridiagonal forms into another array at node offset t0 -- a rule's cache)
int big_tridiag(abz_ctx* ctx, const BigWork& w, int n, int64_t cn, double2* keep = nullptr, double* tri_to = nullptr, int64_t tri_nk_to = 0,
                int64_t t0 = 0) {
    double* const tri = tri_to ? tri_to : w.tri;
    const int64_t tri_nk = tri_to ? tri_nk_to : w.tri_nk;
    // two waves per node up to 44 bands, one up to 48, four above (24^3 nodes, H + eig with one / two / four waves: 64 bands
    // 11.0 / 8.4 / 7.8 ms, 56 bands 7.6 / 6.0 / 5.9, 52 bands 6.6 / 5.3 / 5.3, 48 bands 3.83 / 4.01 / 4.15, 46 bands 3.47 / 3.55,
    // 44 bands 3.31 / 2.97, 40 bands 2.83 / 2.59 / 3.16, 36 bands 2.37 / 2.22, 33 bands 1.66 / 1.57 / 1.98)
    const int sw = abz_switch(SW_BIG_TRI_WAVES);
    const int nw = sw > 0 ? sw : (n > 48 ? 4 : (n <= 44 ? 2 : 1));
    const size_t lds = sizeof(double2) * ((size_t)n * (n + 1) / 2 + 2 * (size_t)n + (nw > 1 ? 64 * (size_t)nw : 0));
    const int64_t blocks = std::min<int64_t>(cn, 256 * 8);
    if (nw >= 4) {
        ABZ_HIP(hipFuncSetAttribute((const void*)big_tridiag_mw_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(big_tridiag_mw_kernel<4>, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, w.Hbuf, cn, n, tri, tri_nk, t0, keep);
    } else if (nw >= 2) {
        ABZ_HIP(hipFuncSetAttribute((const void*)big_tridiag_mw_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(big_tridiag_mw_kernel<2>, dim3((unsigned)blocks), dim3(128), lds, ctx->stream, w.Hbuf, cn, n, tri, tri_nk, t0, keep);
    } else {
        ABZ_HIP(hipFuncSetAttribute((const void*)big_tridiag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(big_tridiag_kernel, dim3((unsigned)blocks), dim3(64), lds, ctx->stream, w.Hbuf, cn, n, tri, tri_nk, t0, keep);
    }
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

static int big_inverse(abz_ctx* ctx, BigInvArgs& ia, int64_t blocks) {
    if (ia.n <= 32 && !ia.partial) {  // (several nodes per workgroup pass)
        const int subs = ia.n <= 8 ? 8 : (ia.n <= 16 ? 4 : 2);
        blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, (ia.nnodes + subs - 1) / subs));
    }
    if (ia.n <= 4)
        hipLaunchKernelGGL((big_inverse_kernel<1, 8>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else if (ia.n <= 8)
        hipLaunchKernelGGL((big_inverse_kernel<2, 8>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else if (ia.n <= 12)
        hipLaunchKernelGGL((big_inverse_kernel<3, 4>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else if (ia.n <= 16)
        hipLaunchKernelGGL((big_inverse_kernel<4, 4>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else if (ia.n <= 20)
        hipLaunchKernelGGL((big_inverse_kernel<5, 2>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else if (ia.n <= 24)
        hipLaunchKernelGGL((big_inverse_kernel<6, 2>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else if (ia.n <= 28)
        hipLaunchKernelGGL((big_inverse_kernel<7, 2>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else if (ia.n <= 32)
        hipLaunchKernelGGL((big_inverse_kernel<8, 2>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else if (ia.n <= 40)
        hipLaunchKernelGGL((big_inverse_kernel<10, 1>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else if (ia.n <= 48)
        hipLaunchKernelGGL((big_inverse_kernel<12, 1>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else if (ia.n <= 56)
        hipLaunchKernelGGL((big_inverse_kernel<14, 1>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    else
        hipLaunchKernelGGL((big_inverse_kernel<16, 1>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ia);
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}
static int big_inv_kind(int integrand) { return integrand == ABZ_F_GLOC ? 0 : (integrand == ABZ_F_TRGLOC ? 1 : (integrand == ABZ_F_DOS ? 2 : -1)); }

int big_series(abz_ctx* ctx, BigSeriesArgs& sa, int64_t c0, int64_t cn) {
    sa.node0 = c0;
    sa.nnodes = cn;
    if (sa.grid) {
        const int64_t lines = cn / sa.npt;
        if (abz_switch(SW_BIG_MFMA) && sa.M <= 64) {
            const int tpl = (sa.npt + 15) / 16;
            hipLaunchKernelGGL(big_series_mfma_kernel, dim3((unsigned)(lines * tpl)), dim3(256), 0, ctx->stream, sa, tpl);
        } else {
            const int tpl = (sa.npt + BIG_TN - 1) / BIG_TN;
            hipLaunchKernelGGL(big_series_kernel, dim3((unsigned)(lines * tpl)), dim3(256), 0, ctx->stream, sa, tpl);
        }
    } else {
        hipLaunchKernelGGL(big_series_kernel, dim3((unsigned)cdivb(cn, BIG_TN)), dim3(256), 0, ctx->stream, sa, 0);
    }
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

}  // namespace

bool big_supported(int n) { return n > 32 && n <= ABZ_MAX_BANDS; }

// 5...32 bands borrow big_inverse_kernel for what their row kernels do not serve (the wave-per-node Gauss-Jordan in LDS took
// 21 ms for a 4-omega G scan of 24^3 nodes at 32 bands; 3.8 ms at 33 bands here): series that are not Hermitian, and from 17
// bands on the matrix-valued G (up to 16 bands the 16-lane row kernel inverts in registers)
bool big_inverse_wanted(int n, int integrand, bool herm) {
    return n > 4 && n <= ABZ_MAX_BANDS && big_inv_kind(integrand) >= 0 && (!herm || (integrand == ABZ_F_GLOC && n > 16));
}
// store-free sums have no row kernel for G at all: 5 bands and more
bool big_inverse_sum_wanted(int n, int integrand, bool herm) {
    return n > 4 && n <= ABZ_MAX_BANDS && big_inv_kind(integrand) >= 0 && (!herm || integrand == ABZ_F_GLOC);
}

// GGR builds (eigenvalues + band velocities, nothing else stored): chunk by chunk H -> tridiagonal with the reflectors kept ->
// eigenvalues into the rule -> dH/dk_j, j = 1 ... d, side by side -> eigenvectors, back-transformation and the quadratic forms in
// big_ggr_kernel (kernels_big_vec.hip).  ref: src/dos_ggr.jl:14-44
bool big_ggr_supported(int n, int d, int M, int npt, bool herm) {
    return big_supported(n) && herm && d >= 1 && d <= 3 && M >= 1 && npt >= 1 && npt < 65536;
}

int launch_big_ggr(abz_ctx* ctx, const GgrRowsSpec& gs) {
    const bool grid = gs.nk == 0;  // (node lists: nk nodes with their grid indices and, for d >= 2, the level-1 set of every node)
    const int64_t nnodes = grid ? gs.nlines * gs.npt : gs.nk;
    if (nnodes == 0) return ABZ_OK;
    const int n = gs.n, nn = n * n, np = n * (n + 1) / 2;
    // chunks of nodes: <= 2 GB of matrices (H and d derivatives) and kept reflectors, whole grid lines in grid mode (a chunk wants
    // several thousand nodes: the kernels run one wave or one small workgroup per node)
    const int64_t mb = abz_switch(SW_BIG_CHUNK_MB) > 0 ? abz_switch(SW_BIG_CHUNK_MB) : 2048;
    int64_t c = std::max<int64_t>(1, (mb << 20) / ((int64_t)sizeof(double2) * ((int64_t)nn * gs.d + np)));
    if (grid) c = std::max<int64_t>(1, c / gs.npt) * gs.npt;
    c = std::min(c, nnodes);
    BigWork w;
    w.chunk = c;
    w.tri_nk = (c + 63) / 64 * 64;
    int rc;
    // (H shares its room with the first derivative: it is dead once the tridiagonal is there)
    if ((rc = ctx->scratch[6].reserve(sizeof(double2) * ((size_t)c * nn * gs.d + 64)))) return rc;
    if ((rc = ctx->scratch[4].reserve(sizeof(double) * (size_t)(2 * BIG_NP) * (size_t)w.tri_nk))) return rc;
    if ((rc = ctx->scratch[7].reserve(sizeof(double2) * ((size_t)c * np + 256)))) return rc;
    w.Hbuf = ctx->scratch[6].as<double2>();
    w.tri = ctx->scratch[4].as<double>();
    double2* keep = ctx->scratch[7].as<double2>();
    BigSeriesArgs sa;
    sa.parents = grid ? nullptr : gs.parents;
    sa.gi = grid ? nullptr : gs.gi;
    sa.x = nullptr;
    sa.tab = gs.tab;
    sa.n = n;
    sa.M = gs.M;
    sa.first = gs.first;
    sa.npt = gs.npt;
    sa.grid = grid ? 1 : 0;
    sa.inv_period = 1.0;
    ProfScope ps(ctx, ABZ_K_EVAL);
    for (int64_t c0 = 0; c0 < nnodes; c0 += w.chunk) {
        const int64_t cn = std::min(w.chunk, nnodes - c0);
        sa.src = gs.src[0];
        sa.deriv = 0;
        sa.Hbuf = w.Hbuf;
        if ((rc = big_series(ctx, sa, c0, cn))) return rc;
        if ((rc = big_tridiag(ctx, w, n, cn, keep))) return rc;
        {
            const size_t qlds = sizeof(double) * 2 * (size_t)(n + 2) * 64;
            ABZ_HIP(hipFuncSetAttribute((const void*)big_qr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qlds));
            hipLaunchKernelGGL(big_qr_kernel, dim3((unsigned)cdivb(cn, 64)), dim3(64), qlds, ctx->stream, w.tri, w.tri_nk, c0, cn, n, gs.E);
            ABZ_HIP(hipGetLastError());
        }
        for (int j = 0; j < gs.d; ++j) {
            sa.src = gs.src[j];
            sa.deriv = j == 0 ? 1 : 0;  // (variables 2, 3: the family was contracted with the factor on that variable)
            sa.Hbuf = w.Hbuf + (int64_t)j * w.chunk * nn;
            if ((rc = big_series(ctx, sa, c0, cn))) return rc;
        }
        if ((rc = launch_big_vec(ctx, w.tri, w.tri_nk, keep, w.Hbuf, w.chunk * nn, c0, cn, n, gs.d, gs.E, gs.V))) return rc;
    }
    return ABZ_OK;
}

// abz_eval_nodes, rule builds (full layout) and the IAI node path
int launch_big_nodes(abz_ctx* ctx, const GenSpec& gs) {
    if (gs.nnodes == 0) return ABZ_OK;
    if (gs.deriv || gs.Uplanes.base || gs.Hplanes.compact) {
        set_error("n = %d bands: stored eigenvectors, derivative matrices and the upper-triangle layout are built for n <= 32 (band velocities "
                  "of 33...64 bands: rules of full grids or of lists of grid nodes with ABZ_WANT_VEL)", gs.n);
        return ABZ_ERR_UNSUPPORTED;
    }
    if (gs.values && big_inv_kind(gs.integrand) < 0) {
        set_error("n = %d bands: integrand %d is built for n <= 32 (G, tr G and DOS are available)", gs.n, gs.integrand);
        return ABZ_ERR_UNSUPPORTED;
    }
    // values: the traces of a Hermitian series come from the tridiagonal; matrix-valued G and series that are not Hermitian
    // from the inverse (eigenvalues are those of Hermitian(h), the upper triangle, whatever the series)
    const bool inv_values = gs.values && (gs.integrand == ABZ_F_GLOC || !gs.herm);
    BigWork w;
    int rc = big_reserve(ctx, gs.n, gs.grid ? gs.npt : 0, gs.nnodes, w);
    if (rc) return rc;
    BigSeriesArgs sa;
    sa.src = gs.src;
    sa.parents = gs.grid ? nullptr : gs.parents;
    sa.gi = gs.grid ? nullptr : gs.gi;
    sa.x = gs.x;
    sa.tab = gs.tab;
    sa.n = gs.n;
    sa.M = gs.M;
    sa.first = gs.first;
    sa.npt = gs.npt > 0 ? gs.npt : 1;
    sa.grid = gs.grid ? 1 : 0;
    sa.inv_period = 1.0 / gs.period;
    sa.Hbuf = w.Hbuf;
    ProfScope ps(ctx, ABZ_K_EVAL);
    for (int64_t c0 = 0; c0 < gs.nnodes; c0 += w.chunk) {
        const int64_t cn = std::min(w.chunk, gs.nnodes - c0);
        if ((rc = big_series(ctx, sa, c0, cn))) return rc;
        if (gs.Hplanes.base || gs.Haos) {
            hipLaunchKernelGGL(big_store_h_kernel, dim3((unsigned)std::min<int64_t>(cdivb((cn + 63) / 64 * 64 * gs.n * gs.n, 256), 256 * 16)), dim3(256), 0, ctx->stream,
                               w.Hbuf, c0, cn, gs.n, gs.Hplanes, gs.Haos);
            ABZ_HIP(hipGetLastError());
        }
        if (inv_values) {
            BigInvArgs ia;
            ia.Hbuf = w.Hbuf;
            ia.node0 = c0;
            ia.nnodes = cn;
            ia.n = gs.n;
            ia.n_sweep = gs.n_sweep > 0 ? gs.n_sweep : 1;
            ia.kind = big_inv_kind(gs.integrand);
            ia.eta = gs.params[0];
            ia.sweep = gs.sweep_dev;
            ia.sweep_per_node = gs.sweep_per_node;
            ia.sweep0 = gs.sweep0;
            ia.w = nullptr;
            ia.values = gs.values;
            ia.partial = nullptr;
            ia.nodes_per_block = 0;
            if ((rc = big_inverse(ctx, ia, std::min<int64_t>(cn, 256 * 8)))) return rc;
        }
        if (!(gs.Eplanes.base || gs.Eaos || (gs.values && !inv_values))) continue;
        if ((rc = big_tridiag(ctx, w, gs.n, cn))) return rc;
        // (one lane per matrix: the QR kernel wants >= 64 matrices per CU's worth of workgroups; a small chunk of the largest
        // matrices is faster by bisection, one thread per eigenvalue)
        if (gs.Eplanes.base && !gs.Eaos && (gs.n <= 56 || cn >= 64 * 512)) {
            const size_t qlds = sizeof(double) * 2 * (size_t)(gs.n + 2) * 64;
            ABZ_HIP(hipFuncSetAttribute((const void*)big_qr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qlds));
            hipLaunchKernelGGL(big_qr_kernel, dim3((unsigned)cdivb(cn, 64)), dim3(64), qlds, ctx->stream, w.tri, w.tri_nk, c0, cn, gs.n, gs.Eplanes);
            ABZ_HIP(hipGetLastError());
        } else if (gs.Eplanes.base || gs.Eaos) {
            hipLaunchKernelGGL(big_bisect_kernel, dim3((unsigned)cdivb(cn, 4)), dim3(256), 0, ctx->stream, w.tri, w.tri_nk, (int64_t)0, c0, cn, gs.n,
                               gs.Eplanes, gs.Eaos);
            ABZ_HIP(hipGetLastError());
        }
        if (gs.values && !inv_values) {
            BigTraceArgs ta;
            ta.tri = w.tri;
            ta.tri_nk = w.tri_nk;
            ta.t0 = 0;
            ta.node0 = c0;
            ta.nnodes = cn;
            ta.n = gs.n;
            ta.n_sweep = gs.n_sweep > 0 ? gs.n_sweep : 1;
            ta.is_dos = gs.integrand == ABZ_F_DOS ? 1 : 0;
            ta.eta = gs.params[0];
            ta.sweep = gs.sweep_dev;
            ta.sweep_per_node = gs.sweep_per_node;
            ta.sweep0 = gs.sweep0;
            ta.w = nullptr;
            ta.values = gs.values;
            ta.partial = nullptr;
            hipLaunchKernelGGL(big_trace_kernel, dim3((unsigned)std::min<int64_t>(cdivb(cn, 256), 256 * 4)), dim3(256), 0, ctx->stream, ta);
            ABZ_HIP(hipGetLastError());
        }
    }
    return ABZ_OK;
}

// sums of resolvent traces over tridiagonals of chunks: shared by the store-free sums and the scans of cached matrices
static int big_sum_chunk(abz_ctx* ctx, const BigWork& w, int n, int64_t c0, int64_t cn, int is_dos, double eta, const double* sweep_dev, int n_sweep,
                         const double* weights, double2* total, bool first) {
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(cdivb(cn, 256), 256 * 2));
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * n_sweep + n_sweep));
    if (rc) return rc;
    BigTraceArgs ta;
    ta.tri = w.tri;
    ta.tri_nk = w.tri_nk;
    ta.t0 = 0;
    ta.node0 = c0;
    ta.nnodes = cn;
    ta.n = n;
    ta.n_sweep = n_sweep;
    ta.is_dos = is_dos;
    ta.eta = eta;
    ta.sweep = sweep_dev;
    ta.sweep_per_node = nullptr;
    ta.sweep0 = 0.0;
    ta.w = weights;
    ta.values = nullptr;
    ta.partial = ctx->scratch[1].as<double2>();
    hipLaunchKernelGGL(big_trace_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ta);
    ABZ_HIP(hipGetLastError());
    double2* part = ta.partial + blocks * n_sweep;
    if ((rc = launch_final_reduce(ctx, ta.partial, blocks, n_sweep, 1.0, first ? total : part))) return rc;
    if (!first) {
        hipLaunchKernelGGL(big_accumulate_kernel, dim3((unsigned)cdivb(n_sweep, 256)), dim3(256), 0, ctx->stream, total, part, n_sweep);
        ABZ_HIP(hipGetLastError());
    }
    return ABZ_OK;
}

__global__ __launch_bounds__(256) void big_scale_kernel(double2* v, int n, double s) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) {
        v[t].x *= s;
        v[t].y *= s;
    }
}

bool big_sum_supported(int n, int M, int npt, int integrand, bool herm) {
    if (npt < 1 || npt >= 65536) return false;
    if (big_inverse_sum_wanted(n, integrand, herm)) return true;  // G, or a series that is not Hermitian: the inverse of every node
    return big_supported(n) && herm && (integrand == ABZ_F_DOS || integrand == ABZ_F_TRGLOC);
}

// store-free PTR sums (abz_ptr_sum)
int launch_big_sum(abz_ctx* ctx, const SumSpec& ss, double* out_reim) {
    const int64_t nnodes = ss.nlines * ss.npt;
    BigWork w;
    int rc = big_reserve(ctx, ss.n, ss.npt, nnodes, w);
    if (rc) return rc;
    if ((rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)ss.n_sweep))) return rc;
    if ((rc = ctx->scratch[5].reserve(sizeof(double) * (size_t)ss.n_sweep))) return rc;
    double* sw = ctx->scratch[5].as<double>();
    ABZ_HIP(hipMemcpyAsync(sw, ss.sweep_host, sizeof(double) * (size_t)ss.n_sweep, hipMemcpyHostToDevice, ctx->stream));
    double2* total = ctx->scratch[2].as<double2>();
    BigSeriesArgs sa;
    sa.src = ss.src;
    sa.parents = nullptr;
    sa.gi = nullptr;
    sa.x = nullptr;
    sa.tab = ss.tab;
    sa.n = ss.n;
    sa.M = ss.M;
    sa.first = ss.first;
    sa.npt = ss.npt;
    sa.grid = 1;
    sa.inv_period = 1.0;
    sa.Hbuf = w.Hbuf;
    if (big_inverse_sum_wanted(ss.n, ss.integrand, ss.herm) || (ss.force_inverse && big_inv_kind(ss.integrand) >= 0)) {
        // matrix-valued G, or a series that is not Hermitian: H(k) of a chunk, the inverse of every node, weighted sums in registers;
        // a group of swept values per pass (the workgroups' partial sums stay under 256 MB), the chunks re-evaluated per group
        const int nn = ss.n * ss.n, kind = big_inv_kind(ss.integrand);
        const int64_t ncomp = kind == 0 ? nn : 1;
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(nnodes, 256 * 2));
        const int group = (int)std::max<int64_t>(1, std::min<int64_t>(ss.n_sweep, (256ll << 20) / (int64_t)(sizeof(double2) * blocks * ncomp)));
        if ((rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * group * ncomp)))) return rc;
        if ((rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)(group * ncomp)))) return rc;
        double2* partial = ctx->scratch[1].as<double2>();
        double2* outd = ctx->scratch[2].as<double2>();
        ProfScope ps(ctx, ABZ_K_EVAL);
        for (int s0 = 0; s0 < ss.n_sweep; s0 += group) {
            const int ns = std::min(group, ss.n_sweep - s0);
            ABZ_HIP(hipMemsetAsync(partial, 0, sizeof(double2) * (size_t)(blocks * ns * ncomp), ctx->stream));
            for (int64_t c0 = 0; c0 < nnodes; c0 += w.chunk) {
                const int64_t cn = std::min(w.chunk, nnodes - c0);
                if ((rc = big_series(ctx, sa, c0, cn))) return rc;
                BigInvArgs ia;
                ia.Hbuf = w.Hbuf;
                ia.node0 = c0;
                ia.nnodes = cn;
                ia.n = ss.n;
                ia.n_sweep = ns;
                ia.kind = kind;
                ia.eta = ss.params[0];
                ia.sweep = sw + s0;
                ia.sweep_per_node = nullptr;
                ia.sweep0 = 0.0;
                ia.w = nullptr;
                ia.values = nullptr;
                ia.partial = partial;
                ia.nodes_per_block = (int)cdivb(cn, blocks);
                if ((rc = big_inverse(ctx, ia, cdivb(cn, ia.nodes_per_block)))) return rc;
            }
            if ((rc = launch_final_reduce(ctx, partial, blocks, (int64_t)ns * ncomp, ss.scale, outd))) return rc;
            ABZ_HIP(hipMemcpyAsync(out_reim + 2 * (size_t)s0 * ncomp, outd, sizeof(double2) * (size_t)ns * ncomp, hipMemcpyDeviceToHost, ctx->stream));
            ABZ_HIP(hipStreamSynchronize(ctx->stream));
        }
        return ABZ_OK;
    }
    ProfScope ps(ctx, ABZ_K_EVAL);
    for (int64_t c0 = 0; c0 < nnodes; c0 += w.chunk) {
        const int64_t cn = std::min(w.chunk, nnodes - c0);
        if ((rc = big_series(ctx, sa, c0, cn))) return rc;
        if ((rc = big_tridiag(ctx, w, ss.n, cn))) return rc;
        if ((rc = big_sum_chunk(ctx, w, ss.n, c0, cn, ss.integrand == ABZ_F_DOS, ss.params[0], sw, ss.n_sweep, nullptr, total, c0 == 0))) return rc;
    }
    hipLaunchKernelGGL(big_scale_kernel, dim3((unsigned)cdivb(ss.n_sweep, 256)), dim3(256), 0, ctx->stream, total, ss.n_sweep, ss.scale);
    ABZ_HIP(hipGetLastError());
    ABZ_HIP(hipMemcpyAsync(out_reim, total, sizeof(double2) * (size_t)ss.n_sweep, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    return ABZ_OK;
}

// scans of a cached rule: DOS / tr G from the matrices (tridiagonalised chunk by chunk); the eigenvalue form goes through
// gen_eig_dos_kernel (kernels_generic.hip), which is generic in n
int launch_big_reduce(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim) {
    if (big_inv_kind(rs.integrand) < 0 || !rs.H.base || rs.H.compact) {
        set_error("n = %d bands: scans of a cached rule offer G, tr G and DOS from the matrices (full layout) and DOS from eigenvalues", rs.n);
        return ABZ_ERR_UNSUPPORTED;
    }
    BigWork w;
    int rc = big_reserve(ctx, rs.n, 0, rs.nk, w);
    if (rc) return rc;
    if (rs.integrand == ABZ_F_GLOC || !rs.herm) {
        // matrix-valued G, or matrices that are not Hermitian: the inverse of every node (big_inverse_kernel), a group of swept
        // values per pass so that the workgroups' partial sums stay under 256 MB
        const int nn = rs.n * rs.n, kind = big_inv_kind(rs.integrand);
        const int64_t ncomp = kind == 0 ? nn : 1;
        const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(rs.nk, 256 * 2));
        const int group = (int)std::max<int64_t>(1, std::min<int64_t>(rs.n_sweep, (256ll << 20) / (int64_t)(sizeof(double2) * blocks * ncomp)));
        if ((rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * group * ncomp)))) return rc;
        if ((rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)(group * ncomp)))) return rc;
        double2* partial = ctx->scratch[1].as<double2>();
        double2* outd = ctx->scratch[2].as<double2>();
        for (int s0 = 0; s0 < rs.n_sweep; s0 += group) {
            const int ns = std::min(group, rs.n_sweep - s0);
            {
                ProfScope ps(ctx, ABZ_K_REDUCE);
                ABZ_HIP(hipMemsetAsync(partial, 0, sizeof(double2) * (size_t)(blocks * ns * ncomp), ctx->stream));
                for (int64_t c0 = 0; c0 < rs.nk; c0 += w.chunk) {
                    const int64_t cn = std::min(w.chunk, rs.nk - c0);
                    hipLaunchKernelGGL(big_load_h_kernel, dim3((unsigned)std::min<int64_t>(cdivb((cn + 63) / 64 * 64 * nn, 256), 256 * 16)), dim3(256), 0,
                                       ctx->stream, w.Hbuf, c0, cn, rs.n, rs.H);
                    ABZ_HIP(hipGetLastError());
                    BigInvArgs ia;
                    ia.Hbuf = w.Hbuf;
                    ia.node0 = c0;
                    ia.nnodes = cn;
                    ia.n = rs.n;
                    ia.n_sweep = ns;
                    ia.kind = kind;
                    ia.eta = rs.params[0];
                    ia.sweep = rs.sweep_dev + s0;
                    ia.sweep_per_node = nullptr;
                    ia.sweep0 = 0.0;
                    ia.w = rs.w;
                    ia.values = nullptr;
                    ia.partial = partial;
                    ia.nodes_per_block = (int)cdivb(cn, blocks);
                    if ((rc = big_inverse(ctx, ia, cdivb(cn, ia.nodes_per_block)))) return rc;
                }
                if ((rc = launch_final_reduce(ctx, partial, blocks, (int64_t)ns * ncomp, rs.scale, outd))) return rc;
            }
            if (rs.out_dev) {
                ABZ_HIP(hipMemcpyAsync(rs.out_dev + 2 * (size_t)s0 * ncomp, outd, sizeof(double2) * (size_t)ns * ncomp, hipMemcpyDeviceToDevice,
                                       ctx->stream));
                continue;
            }
            ABZ_HIP(hipMemcpyAsync(out_reim + 2 * (size_t)s0 * ncomp, outd, sizeof(double2) * (size_t)ns * ncomp, hipMemcpyDeviceToHost, ctx->stream));
            ABZ_HIP(hipStreamSynchronize(ctx->stream));
        }
        return ABZ_OK;
    }
    if ((rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)rs.n_sweep))) return rc;
    double2* total = ctx->scratch[2].as<double2>();
    if (rs.tri_cache && rs.tri_state) {
        // the rule keeps the tridiagonal forms of its nodes: tridiagonalise once per fill of the rule, then every scan is the p'/p pass
        ProfScope ps(ctx, ABZ_K_REDUCE);
        if (*rs.tri_state == 0) {
            for (int64_t c0 = 0; c0 < rs.nk; c0 += w.chunk) {
                const int64_t cn = std::min(w.chunk, rs.nk - c0);
                hipLaunchKernelGGL(big_load_h_kernel, dim3((unsigned)std::min<int64_t>(cdivb((cn + 63) / 64 * 64 * rs.n * rs.n, 256), 256 * 16)), dim3(256), 0,
                                   ctx->stream, w.Hbuf, c0, cn, rs.n, rs.H);
                ABZ_HIP(hipGetLastError());
                if ((rc = big_tridiag(ctx, w, rs.n, cn, nullptr, rs.tri_cache, rs.tri_nk, c0))) return rc;
            }
            *rs.tri_state = 1;
        }
        BigWork wc = w;
        wc.tri = rs.tri_cache;
        wc.tri_nk = rs.tri_nk;
        if ((rc = big_sum_chunk(ctx, wc, rs.n, 0, rs.nk, rs.integrand == ABZ_F_DOS, rs.params[0], rs.sweep_dev, rs.n_sweep, rs.w, total, true))) return rc;
        hipLaunchKernelGGL(big_scale_kernel, dim3((unsigned)cdivb(rs.n_sweep, 256)), dim3(256), 0, ctx->stream, total, rs.n_sweep, rs.scale);
        ABZ_HIP(hipGetLastError());
    } else {
        ProfScope ps(ctx, ABZ_K_REDUCE);
        for (int64_t c0 = 0; c0 < rs.nk; c0 += w.chunk) {
            const int64_t cn = std::min(w.chunk, rs.nk - c0);
            hipLaunchKernelGGL(big_load_h_kernel, dim3((unsigned)std::min<int64_t>(cdivb((cn + 63) / 64 * 64 * rs.n * rs.n, 256), 256 * 16)), dim3(256), 0, ctx->stream,
                               w.Hbuf, c0, cn, rs.n, rs.H);
            ABZ_HIP(hipGetLastError());
            if ((rc = big_tridiag(ctx, w, rs.n, cn))) return rc;
            if ((rc = big_sum_chunk(ctx, w, rs.n, c0, cn, rs.integrand == ABZ_F_DOS, rs.params[0], rs.sweep_dev, rs.n_sweep, rs.w, total, c0 == 0)))
                return rc;
        }
        hipLaunchKernelGGL(big_scale_kernel, dim3((unsigned)cdivb(rs.n_sweep, 256)), dim3(256), 0, ctx->stream, total, rs.n_sweep, rs.scale);
        ABZ_HIP(hipGetLastError());
    }
    if (rs.out_dev) {
        ABZ_HIP(hipMemcpyAsync(rs.out_dev, total, sizeof(double2) * (size_t)rs.n_sweep, hipMemcpyDeviceToDevice, ctx->stream));
        return ABZ_OK;
    }
    ABZ_HIP(hipMemcpyAsync(out_reim, total, sizeof(double2) * (size_t)rs.n_sweep, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    return ABZ_OK;
}

}  // namespace abz
