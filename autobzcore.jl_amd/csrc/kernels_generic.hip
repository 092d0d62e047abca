// Generic-n (5 <= n <= 32 bands) kernels: ONE WAVEFRONT PER NODE, the n x n matrices live in a
// wave-private LDS slab (no block barriers: a wave only reads LDS bytes it wrote itself).
//
//   gen_node_kernel    series value H(k) = sum_m c1[m] w z^m at a node (lanes stride over the n^2
//                      entries, coefficient reads coalesced), then per `mode`: store H / eigenvalues
//                      (cooperative cyclic Jacobi on the upper triangle) / integrand value
//                      (tr / full inverse of (omega + i eta) I - H by Gauss-Jordan in LDS).
//                      Serves abz_eval_nodes, PTR rule builds and the IAI innermost nodes for n > 4.
//   gen_reduce_kernel  sum_k w_k f(H(k); omega_i) over a cached rule for n > 4.
//
// A = (omega + i eta) I - H has a positive-definite anti-Hermitian part (eta > 0) whenever H is
// Hermitian, so elimination without pivoting is backward stable up to a growth factor <= ||A||/eta;
// that is what the Gauss-Jordan below relies on.
#include <utility>

#include "abz_internal.h"
#include "inner_adapt.h"
#include "device_math.h"
#include "rows_device.h"

namespace abz {

static inline int64_t cdiv2(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// In-LDS Gauss-Jordan: X <- inv(A), A destroyed.  A, X: n*n double2, column-major (a + n*b).
__device__ void wave_inverse(double2* A, double2* X, int n, int lane) {
    const int nn = n * n;
    for (int t = lane; t < nn; t += 64) {
        const int a = t % n, b = t / n;
        X[t] = make_double2(a == b ? 1.0 : 0.0, 0.0);
    }
    wave_sync();
    for (int c = 0; c < n; ++c) {
        const double2 p = A[c + n * c];
        const double ipn = 1.0 / (p.x * p.x + p.y * p.y);
        const double ipr = p.x * ipn, ipi = -p.y * ipn;  // 1 / pivot
        // new values into registers (reads see the state before this column's update)
        double2 na[16], nx[16];  // n*n / 64 <= 16 entries per lane for n <= 32 (unrolled: registers)
#pragma unroll
        for (int cnt = 0; cnt < 16; ++cnt) {
            const int t = lane + 64 * cnt;
            if (t < nn) {
                const int r = t % n, cc = t / n;
                const double2 arow = A[c + n * cc], xrow = X[c + n * cc];  // pivot row entries (old)
                if (r == c) {
                    na[cnt] = make_double2(arow.x * ipr - arow.y * ipi, arow.x * ipi + arow.y * ipr);
                    nx[cnt] = make_double2(xrow.x * ipr - xrow.y * ipi, xrow.x * ipi + xrow.y * ipr);
                } else {
                    const double2 f0 = A[r + n * c];
                    const double fr = f0.x * ipr - f0.y * ipi, fi = f0.x * ipi + f0.y * ipr;  // f / pivot
                    const double2 av = A[t], xv = X[t];
                    na[cnt] = make_double2(av.x - (fr * arow.x - fi * arow.y), av.y - (fr * arow.y + fi * arow.x));
                    nx[cnt] = make_double2(xv.x - (fr * xrow.x - fi * xrow.y), xv.y - (fr * xrow.y + fi * xrow.x));
                }
            }
        }
        wave_sync();
#pragma unroll
        for (int cnt = 0; cnt < 16; ++cnt) {
            const int t = lane + 64 * cnt;
            if (t < nn) {
                A[t] = na[cnt];
                X[t] = nx[cnt];
            }
        }
        wave_sync();
    }
}

// Cooperative cyclic Jacobi on the Hermitian matrix given by the UPPER triangle of A (column-major,
// destroyed).  Eigenvalues ascending into e[0..n) (LDS doubles).  V (optional, n*n in LDS): the rotations
// are accumulated, column j of V is the eigenvector of A's final diagonal entry j, and the rank of that
// entry in the ascending order is left in A[j + n*j].y (the diagonal's imaginary part is otherwise zero).
__device__ void wave_eig(double2* A, double* e, int n, int lane, double2* V = nullptr) {
    const int nn = n * n;
    if (V) {
        for (int t = lane; t < nn; t += 64) V[t] = make_double2((t % n == t / n) ? 1.0 : 0.0, 0.0);
    }
    // hermitise from the upper triangle, real diagonal
    for (int t = lane; t < nn; t += 64) {
        const int a = t % n, b = t / n;
        if (a > b) {
            const double2 u = A[b + n * a];
            A[t] = make_double2(u.x, -u.y);
        }
    }
    wave_sync();
    for (int t = lane; t < n; t += 64) A[t + n * t].y = 0.0;
    wave_sync();
    double norm2 = 0.0;
    for (int t = lane; t < nn; t += 64) norm2 += A[t].x * A[t].x + A[t].y * A[t].y;
    norm2 = wsum(norm2);
    const double tiny = 1e-34 * norm2;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off2 = 0.0;
        for (int t = lane; t < nn; t += 64) {
            const int a = t % n, b = t / n;
            if (a < b) off2 += A[t].x * A[t].x + A[t].y * A[t].y;
        }
        off2 = wsum(off2);
        if (!(off2 > tiny)) break;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                const double2 al = A[p + n * q];
                const double b2 = al.x * al.x + al.y * al.y;
                if (!(b2 > tiny)) continue;  // wave-uniform
                const double rb = rsqrt(b2), b = b2 * rb;
                const double gr = al.x * rb, gi = -al.y * rb;
                const double app = A[p + n * p].x, aqq = A[q + n * q].x;
                const double dd = aqq - app;
                const double t = copysign(2.0 * b, dd) / (fabs(dd) + sqrt(dd * dd + 4.0 * b2));
                const double c = rsqrt(1.0 + t * t), s = t * c;
                const double sgr = s * gr, sgi = s * gi, cgr = c * gr, cgi = c * gi;
                // lane r updates (r,p),(r,q) and the mirrored entries
                double2 np_[1], nq_[1];
                const int r = lane;
                const bool act = r < n && r != p && r != q;
                if (act) {
                    const double2 x = A[r + n * p], y = A[r + n * q];
                    np_[0] = make_double2(c * x.x - (sgr * y.x - sgi * y.y), c * x.y - (sgr * y.y + sgi * y.x));
                    nq_[0] = make_double2(s * x.x + (cgr * y.x - cgi * y.y), s * x.y + (cgr * y.y + cgi * y.x));
                }
                if (V && r < n) {  // V <- V J: the same column rotation, every row (a lane owns its row of V)
                    const double2 x = V[r + n * p], y = V[r + n * q];
                    V[r + n * p] = make_double2(c * x.x - (sgr * y.x - sgi * y.y), c * x.y - (sgr * y.y + sgi * y.x));
                    V[r + n * q] = make_double2(s * x.x + (cgr * y.x - cgi * y.y), s * x.y + (cgr * y.y + cgi * y.x));
                }
                wave_sync();
                if (act) {
                    A[r + n * p] = np_[0];
                    A[r + n * q] = nq_[0];
                    A[p + n * r] = make_double2(np_[0].x, -np_[0].y);
                    A[q + n * r] = make_double2(nq_[0].x, -nq_[0].y);
                }
                if (lane == 0) {
                    A[p + n * p] = make_double2(app - t * b, 0.0);
                    A[q + n * q] = make_double2(aqq + t * b, 0.0);
                    A[p + n * q] = make_double2(0.0, 0.0);
                    A[q + n * p] = make_double2(0.0, 0.0);
                }
                wave_sync();
            }
        }
    }
    // ascending order by rank counting (n <= 64 lanes)
    double v = 0.0;
    if (lane < n) v = A[lane + n * lane].x;
    wave_sync();
    if (lane < n) {
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double u = A[j + n * j].x;
            rank += (u < v || (u == v && j < lane)) ? 1 : 0;
        }
        e[rank] = v;
        if (V) A[lane + n * lane].y = (double)rank;
    }
    wave_sync();
}

struct GenArgs {
    const double2* src;      // coefficient sets [slot][M][n*n]
    const int64_t* parents;  // per node (null in grid mode)
    const double* x;         // per node coordinate (null: grid index)
    const int32_t* gi;       // per node grid index (null in grid mode / x mode)
    const double2* tab;
    const double* tail;
    int64_t nnodes;
    int n, M, first, npt, d, grid, deriv;
    double inv_period;
    // outputs
    PlaneView Hplanes;
    PlaneView Eplanes;
    PlaneView Uplanes;  // eigenvectors, planes 2*(a + n*band) + {re, im} (null: eigenvalues only)
    double2* Haos;  // [node][n*n]
    double* Eaos;   // [node][n]
    // integrand
    int integrand, n_sweep, ncomp;
    double p[4];
    const double* sweep;  // device [n_sweep] (null: use sweep0)
    const double* sweep_per_node;  // device [nnodes] (n_sweep = 1)
    double sweep0;
    double2* values;  // [node][n_sweep][ncomp]
};

// value of the integrand at the node whose H sits in LDS `H`; W, X: scratch n*n; out via lanes
__device__ void gen_integrand(const GenArgs& a, const double2* H, double2* W, double2* X, double* ev, double sw,
                              int lane, double2* out /* global, ncomp */) {
    const int n = a.n, nn = n * n;
    if (a.integrand == ABZ_F_ONE) {
        if (lane == 0) out[0] = make_double2(1.0, 0.0);
        return;
    }
    if (a.integrand == ABZ_F_DOS_EIG) {
        double acc = 0.0;
        for (int b = lane; b < n; b += 64) {
            const double de = sw - ev[b];
            acc += a.p[0] / (de * de + a.p[0] * a.p[0]);
        }
        acc = wsum(acc);
        if (lane == 0) out[0] = make_double2(acc * 0.31830988618379067153776752674503, 0.0);
        return;
    }
    for (int t = lane; t < nn; t += 64) {
        const int r = t % n, c = t / n;
        W[t] = make_double2((r == c ? sw : 0.0) - H[t].x, (r == c ? a.p[0] : 0.0) - H[t].y);
    }
    wave_sync();
    wave_inverse(W, X, n, lane);
    if (a.integrand == ABZ_F_GLOC) {
        for (int t = lane; t < nn; t += 64) out[t] = X[t];
    } else {
        double tr = 0.0, ti = 0.0;
        for (int t = lane; t < n; t += 64) {
            tr += X[t + n * t].x;
            ti += X[t + n * t].y;
        }
        tr = wsum(tr);
        ti = wsum(ti);
        if (lane == 0)
            out[0] = (a.integrand == ABZ_F_DOS) ? make_double2(-ti * 0.31830988618379067153776752674503, 0.0)
                                                : make_double2(tr, ti);
    }
    wave_sync();
}

__global__ __launch_bounds__(256) void gen_node_kernel(GenArgs a, int waves_per_block) {
    extern __shared__ double2 lds_g[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= waves_per_block) return;
    const int n = a.n, nn = n * n;
    // per-wave slab: H[nn] W[nn] X[nn] ph[M] ev[n doubles]
    double2* H = lds_g + (size_t)wave * (3 * nn + a.M + (n + 1) / 2);
    double2* W = H + nn;
    double2* X = W + nn;
    double2* ph = X + nn;
    double* ev = reinterpret_cast<double*>(ph + a.M);
    int fm = 0;
    if (a.npt > 0) {
        fm = a.first % a.npt;
        if (fm < 0) fm += a.npt;
    }
    const int64_t wstride = (int64_t)gridDim.x * waves_per_block;
    for (int64_t k = (int64_t)blockIdx.x * waves_per_block + wave; k < a.nnodes; k += wstride) {
        int64_t slot;
        double zr, zi, wr, wi;
        if (a.grid) {
            slot = k / a.npt;
            const int i1 = (int)(k - slot * a.npt);
            const double2 z = a.tab[i1];
            const double2 w = a.tab[(int)(((int64_t)fm * i1) % a.npt)];
            zr = z.x, zi = z.y, wr = w.x, wi = w.y;
        } else {
            slot = a.parents ? a.parents[k] : 0;
            if (a.x) {
                const double xx = a.x[k] * a.inv_period;
                sincospi(2.0 * xx, &zi, &zr);
                sincospi(2.0 * ((double)a.first * xx), &wi, &wr);
            } else {
                const int i1 = a.gi[k];
                const double2 z = a.tab[i1];
                const double2 w = a.tab[(int)(((int64_t)fm * i1) % a.npt)];
                zr = z.x, zi = z.y, wr = w.x, wi = w.y;
            }
        }
        // phases w z^m (every lane runs the short recurrence; lane m keeps its own)
        {
            double pr = wr, pi = wi;
            for (int m = 0; m < a.M; ++m) {
                if (lane == (m & 63)) {
                    double qr = pr, qi = pi;
                    if (a.deriv) {
                        const double f = 6.283185307179586476925286766559 * (double)(a.first + m);
                        qr = -f * pi;
                        qi = f * pr;
                    }
                    ph[m] = make_double2(qr, qi);
                }
                const double nr = pr * zr - pi * zi, ni = pr * zi + pi * zr;
                pr = nr;
                pi = ni;
            }
        }
        wave_sync();
        const double2* __restrict__ c1 = a.src + slot * ((int64_t)a.M * nn);
        for (int t = lane; t < nn; t += 64) {
            double hr = 0.0, hi = 0.0;
            for (int m = 0; m < a.M; ++m) {
                const double2 c = c1[(int64_t)m * nn + t];
                const double2 q = ph[m];
                hr = fma(c.x, q.x, hr);
                hr = fma(-c.y, q.y, hr);
                hi = fma(c.x, q.y, hi);
                hi = fma(c.y, q.x, hi);
            }
            H[t] = make_double2(hr, hi);
            if (a.Hplanes.base) {
                double* ho = a.Hplanes.base + view_off(a.Hplanes, k);
                ho[(int64_t)(2 * t) * a.Hplanes.pitch] = hr;
                ho[(int64_t)(2 * t + 1) * a.Hplanes.pitch] = hi;
            }
            if (a.Haos) a.Haos[k * nn + t] = make_double2(hr, hi);
        }
        wave_sync();
        const bool need_eig = a.Eplanes.base || a.Eaos || a.Uplanes.base || (a.values && a.integrand == ABZ_F_DOS_EIG);
        if (need_eig) {
            for (int t = lane; t < nn; t += 64) W[t] = H[t];
            wave_sync();
            wave_eig(W, ev, n, lane, a.Uplanes.base ? X : nullptr);
            for (int b = lane; b < n; b += 64) {
                if (a.Eplanes.base) a.Eplanes.base[view_off(a.Eplanes, k) + (int64_t)b * a.Eplanes.pitch] = ev[b];
                if (a.Eaos) a.Eaos[k * n + b] = ev[b];
            }
            if (a.Uplanes.base) {  // U[:, rank(j)] = V[:, j]
                double* uo = a.Uplanes.base + view_off(a.Uplanes, k);
                for (int t = lane; t < nn; t += 64) {
                    const int ra = t % n, j = t / n;
                    const int band = (int)W[j + n * j].y;
                    uo[(int64_t)(2 * (ra + n * band)) * a.Uplanes.pitch] = X[t].x;
                    uo[(int64_t)(2 * (ra + n * band) + 1) * a.Uplanes.pitch] = X[t].y;
                }
            }
        }
        if (a.values) {
            for (int s = 0; s < a.n_sweep; ++s) {
                const double sw = a.sweep_per_node ? a.sweep_per_node[k] : (a.sweep ? a.sweep[s] : a.sweep0);
                gen_integrand(a, H, W, X, ev, sw, lane, a.values + (k * a.n_sweep + s) * a.ncomp);
            }
        }
        wave_sync();
    }
}

// Band velocities for n > 4: v_b = Re sum_{a,c} conj(U[a][b]) D[a][c] U[c][b], one thread per (node, band),
// nodes fastest (every plane read is coalesced across the lanes; a band's threads re-read D through L2).
__global__ __launch_bounds__(256) void gen_velocity_kernel(PlaneView Uv, PlaneView Dv, PlaneView Vv, int64_t nk, int n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (k >= nk) return;
    const double* __restrict__ u = Uv.base + view_off(Uv, k) + (int64_t)(2 * n * b) * Uv.pitch;  // column b of U
    const double* __restrict__ dm = Dv.base + view_off(Dv, k);
    double v = 0.0;
    for (int a = 0; a < n; ++a) {
        double tr = 0.0, ti = 0.0;  // t = sum_c D[a][c] U[c][b]
        for (int c = 0; c < n; ++c) {
            const double dr = dm[(int64_t)(2 * (a + n * c)) * Dv.pitch], di = dm[(int64_t)(2 * (a + n * c) + 1) * Dv.pitch];
            const double ur = u[(int64_t)(2 * c) * Uv.pitch], ui = u[(int64_t)(2 * c + 1) * Uv.pitch];
            tr += dr * ur - di * ui;
            ti += dr * ui + di * ur;
        }
        v += u[(int64_t)(2 * a) * Uv.pitch] * tr + u[(int64_t)(2 * a + 1) * Uv.pitch] * ti;
    }
    Vv.base[view_off(Vv, k) + (int64_t)b * Vv.pitch] = v;
}

static bool launch_gen_velocity_rows(abz_ctx* ctx, int n, PlaneView U, PlaneView dH, PlaneView Vj, int64_t nk);

int launch_gen_velocity(abz_ctx* ctx, int n, PlaneView U, PlaneView dH, PlaneView Vj, int64_t nk) {
    if (nk == 0) return ABZ_OK;
    if (launch_gen_velocity_rows(ctx, n, U, dH, Vj, nk)) {
        ABZ_HIP(hipGetLastError());
        return ABZ_OK;
    }
    hipLaunchKernelGGL(gen_velocity_kernel, dim3((unsigned)cdiv2(nk, 256), (unsigned)n), dim3(256), 0, ctx->stream, U, dH, Vj, nk, n);
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// IAI innermost nodes for n > 4, GK panel at a time.
// One workgroup per panel (15 nodes of one 1-D integral = one coefficient set): the set is staged in
// LDS once and shared by all nodes; a node is evaluated by NP lanes (NP = n rounded up to 8/16/32),
// lane r owning ROW r of the matrix in registers:
//   series     row_r(H) = sum_m c_m[r, :] p_m          (coefficient reads: LDS, 16 B per lane,
//                                                        identical across the nodes of a wave)
//   resolvent  in-place Gauss-Jordan inversion of A = (omega + i eta) I - H without pivoting (see the
//              header of this file): per pivot c the lane that owns row c publishes it through a small
//              LDS row buffer, every lane does one rank-1 row update  a_r += g_r u  with
//              g_r = -a_rc / p (g_c = 1/p - 1), a_rc <- g_r (a_cc <- 1/p).  n^3 complex FMA per node in
//              registers instead of the 2 n^3 LDS-resident updates of the wave-per-node kernel.
// ------------------------------------------------------------------------------------------


// row r of A = (sw + i eta) I - H from the row of -H; padding rows become identity rows
template <int NP, bool PAD>
__device__ __forceinline__ void panel_shift_row(int n, double sw, double eta, int r, double (&ar)[NP], double (&ai)[NP]) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (!PAD && r >= n) {  // padding rows: identity, decoupled from the n x n block
            ar[j] = 0.0;
            ai[j] = 0.0;
        }
        if (j == r) {
            ar[j] += (r < n) ? sw : 1.0;
            ai[j] += (r < n) ? eta : 0.0;
        }
    }
}


// one Gauss-Jordan pivot (column C) of the zero-padded NP x NP matrix whose row r this lane holds: the pivot
// row comes from lane C of the group, eight columns at a time (the eight that hold the pivot first)
template <int NP, int C>
__device__ __forceinline__ void panel_pivot(int r, double (&ar)[NP], double (&ai)[NP]) {
    constexpr int NB = NP / 8;
    double gr = 0.0, gi = 0.0, ipr = 0.0, ipi = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j0 = 8 * ((C / 8 + b) % NB);
        double ur[8], ui[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ur[j] = group_bcast<NP, C>(ar[j0 + j]);
            ui[j] = group_bcast<NP, C>(ai[j0 + j]);
        }
        if (b == 0) {
            const double pr = ur[C % 8], pi = ui[C % 8];
            const double inv = rcp_nr(pr * pr + pi * pi);
            ipr = pr * inv;
            ipi = -pi * inv;  // 1 / pivot
            const double fr = ar[C], fi = ai[C];
            // g = -f / p for the other rows, 1/p - 1 for the pivot row itself
            gr = -(fr * ipr - fi * ipi);
            gi = -(fr * ipi + fi * ipr);
            if (r == C) {
                gr = ipr - 1.0;
                gi = ipi;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j0 + j != C) {
                ar[j0 + j] = fma(gr, ur[j], ar[j0 + j]);
                ar[j0 + j] = fma(-gi, ui[j], ar[j0 + j]);
                ai[j0 + j] = fma(gr, ui[j], ai[j0 + j]);
                ai[j0 + j] = fma(gi, ur[j], ai[j0 + j]);
            }
        }
    }
    ar[C] = (r == C) ? ipr : gr;
    ai[C] = (r == C) ? ipi : gi;
}

template <int NP, bool GUARD, int... C>
__device__ __forceinline__ void panel_pivots(int n, int r, double (&ar)[NP], double (&ai)[NP], std::integer_sequence<int, C...>) {
    // GUARD: the rows / columns >= n are an identity block kept in registers only (unpadded LDS layout); their
    // pivots are 1 and are skipped (uniform)
    ((!GUARD || C < n ? panel_pivot<NP, C>(r, ar, ai) : (void)0), ...);
}

// in-place inversion of the matrix whose row r this lane holds (rows >= n: identity rows, columns >= n of the real
// rows: zeros -- by the zero-padded staging (PAD) or by panel_series_row / panel_shift_row (!PAD)).  The pivot rows
// travel by `group_bcast` (the first version published them through per-slot LDS row buffers).
template <int NP, bool PAD>
__device__ __forceinline__ void panel_invert_rows(int n, int r, double (&ar)[NP], double (&ai)[NP]) {
    panel_pivots<NP, !PAD>(n, r, ar, ai, std::make_integer_sequence<int, NP>());
}


// ---- Gauss-Jordan with the pivot-row broadcast folded INTO the FMAs (16 lanes per node, trace of the inverse only) ----
// `v_fmac_f64_dpp dst, src0, src1 row_newbcast:C` computes dst += src0[lane C of the row] * src1: no separate broadcast
// move (panel_pivot above: 2 moves + 4 FMAs per complex column, here 4 FMAs).  Reading the pivot row straight out of the
// registers that the same FMAs update is only legal if lane C's row does NOT change while the others read it, so the
// pivot row's own scaling by 1/p is deferred: lane C runs the column updates with g = 0 (a + 0 * b = a, its registers
// are rewritten with the same values) and keeps q = 1/p; the row it holds from then on is p times the true row, which
// the later pivots' updates preserve (g_C' = -a_CC'/p' on the stored row is p times the true multiplier), and the
// diagonal element of the inverse is a_rr * q_r at the end.  The compiler does not fold `v_mov_b64_dpp` into the FMA by
// itself and its hazard recogniser does not look inside inline asm: every asm block starts with the two wait states a
// DPP read needs after a VALU write of its source (`s_nop 1`); inside a block the only DPP reads of a register written
// one instruction earlier are reads of lane C's unchanged values.
template <int C>
__device__ __forceinline__ void fmac_col_bcast(double& xr, double& xi, double gr, double gi) {
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, -%3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %0, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
        : "+v"(xr), "+v"(xi)
        : "v"(gr), "v"(gi), "n"(C));
}

template <int C>
__device__ __forceinline__ void pivot_bcast(double ar, double ai, double& pr, double& pi) {
    asm("s_nop 1\n\t"
        "v_mov_b64_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
        : "=&v"(pr), "=&v"(pi)
        : "v"(ar), "v"(ai), "n"(C));
}

// Lane C's special cases of pivot C in one exec-masked block (lane C of each 16-lane row): g = 0, q = 1/p.  Four
// `v_mov_b64` under a scalar exec switch instead of a dozen `v_cndmask_b32`; the scalar instructions issue beside the
// other waves' VALU work.  No DPP instruction runs under the narrowed exec.
template <int C>
__device__ __forceinline__ void pivot_lane_fixup(double& gr, double& gi, double& qr, double& qi, double ipr, double ipi) {
    constexpr unsigned half = 0x00010001u << C;  // lane C of the two rows of each half of the wave
    unsigned long long saved;
    asm("s_mov_b64 %[sv], exec\n\t"
        "s_and_b32 exec_lo, exec_lo, %[hm]\n\t"
        "s_and_b32 exec_hi, exec_hi, %[hm]\n\t"
        "v_mov_b64 %[gr], 0\n\t"
        "v_mov_b64 %[gi], 0\n\t"
        "v_mov_b64 %[qr], %[ipr]\n\t"
        "v_mov_b64 %[qi], %[ipi]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [gr] "+v"(gr), [gi] "+v"(gi), [qr] "+v"(qr), [qi] "+v"(qi), [sv] "=&s"(saved)
        : [hm] "n"(half), [ipr] "v"(ipr), [ipi] "v"(ipi)
        : "scc");
}

template <int C>
__device__ __forceinline__ void fmac_pivot(int r, double (&ar)[16], double (&ai)[16], double& qr, double& qi) {
    double pr, pi;
    pivot_bcast<C>(ar[C], ai[C], pr, pi);
    const double inv = rcp_nr(pr * pr + pi * pi);
    const double ipr = pr * inv, ipi = -pi * inv;  // 1 / pivot
    const double fr = ar[C], fi = ai[C];
    double gr = -(fr * ipr - fi * ipi);
    double gi = -(fr * ipi + fi * ipr);
    pivot_lane_fixup<C>(gr, gi, qr, qi, ipr, ipi);
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j != C) fmac_col_bcast<C>(ar[j], ai[j], gr, gi);
    // column C: the multipliers; the pivot row stays unscaled, its column-C element is p * (1/p) = 1 (g = 0 there: only
    // the high word differs from the multiplier's)
    ar[C] = __hiloint2double(r == C ? 0x3ff00000 : __double2hiint(gr), __double2loint(gr));
    ai[C] = gi;
}

template <bool GUARD, int... C>
__device__ __forceinline__ void fmac_pivots(int n, int r, double (&ar)[16], double (&ai)[16], double& qr, double& qi,
                                            std::integer_sequence<int, C...>) {
    // GUARD: the pivots of the identity rows >= n are skipped (uniform), as in panel_pivots
    ((!GUARD || C < n ? fmac_pivot<C>(r, ar, ai, qr, qi) : (void)0), ...);
}

#define ABZ_DIAG_STEP(j) "v_mov_b64 %[dr], %[r" #j "]\n\tv_mov_b64 %[di], %[i" #j "]\n\ts_lshl_b64 exec, exec, 1\n\t"
// the exec mask walks from lane J0 of every row upwards (the wave runs with all lanes on here: the shifted mask never
// turns on a lane the caller had off)
template <int J0>
__device__ __forceinline__ void diag_capture8(const double (&ar)[16], const double (&ai)[16], double& dr, double& di) {
    constexpr unsigned half = 0x00010001u << J0;
    unsigned long long saved;
    asm("s_mov_b64 %[sv], exec\n\t"
        "s_and_b32 exec_lo, exec_lo, %[hm]\n\t"
        "s_and_b32 exec_hi, exec_hi, %[hm]\n\t"
        ABZ_DIAG_STEP(0) ABZ_DIAG_STEP(1) ABZ_DIAG_STEP(2) ABZ_DIAG_STEP(3) ABZ_DIAG_STEP(4) ABZ_DIAG_STEP(5) ABZ_DIAG_STEP(6) ABZ_DIAG_STEP(7)
        "s_mov_b64 exec, %[sv]"
        : [dr] "+v"(dr), [di] "+v"(di), [sv] "=&s"(saved)
        : [hm] "n"(half), [r0] "v"(ar[J0]), [r1] "v"(ar[J0 + 1]), [r2] "v"(ar[J0 + 2]), [r3] "v"(ar[J0 + 3]), [r4] "v"(ar[J0 + 4]),
          [r5] "v"(ar[J0 + 5]), [r6] "v"(ar[J0 + 6]), [r7] "v"(ar[J0 + 7]), [i0] "v"(ai[J0]), [i1] "v"(ai[J0 + 1]), [i2] "v"(ai[J0 + 2]),
          [i3] "v"(ai[J0 + 3]), [i4] "v"(ai[J0 + 4]), [i5] "v"(ai[J0 + 5]), [i6] "v"(ai[J0 + 6]), [i7] "v"(ai[J0 + 7])
        : "scc");
}
#undef ABZ_DIAG_STEP

// trace of the inverse of the zero-padded 16 x 16 matrix whose row r this lane holds (rows >= n: identity rows)
template <bool GUARD = false>
__device__ __forceinline__ void panel_inverse_trace_fmac(int n, int r, double (&ar)[16], double (&ai)[16], double& tr, double& ti) {
    double qr = 1.0, qi = 0.0;
    fmac_pivots<GUARD>(n, r, ar, ai, qr, qi, std::make_integer_sequence<int, 16>());
    // lane r's diagonal element a_rr: two moves per column under the exec mask of lane j of every row (a select chain
    // costs four `v_cndmask_b32` per column)
    double dr = 0.0, di = 0.0;
    diag_capture8<0>(ar, ai, dr, di);
    diag_capture8<8>(ar, ai, dr, di);
    tr = r < n ? dr * qr - di * qi : 0.0;
    ti = r < n ? dr * qi + di * qr : 0.0;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        tr += __shfl_xor(tr, off, 64);
        ti += __shfl_xor(ti, off, 64);
    }
}

template <int NP, bool PAD>
__device__ __forceinline__ void panel_inverse_row(const double2* coef, int n, int M, int first,
                                                  double xx, double sw, double eta, int r, double (&ar)[NP],
                                                  double (&ai)[NP]) {
    double zr, zi, pr, pi;
    sincospi(2.0 * xx, &zi, &zr);
    sincospi(2.0 * ((double)first * xx), &pi, &pr);
    panel_series_row<NP, PAD>(coef, n, M, zr, zi, pr, pi, r, ar, ai);
    panel_shift_row<NP, PAD>(n, sw, eta, r, ar, ai);
    panel_invert_rows<NP, PAD>(n, r, ar, ai);
}


// (A "duo" layout -- 8 lanes per node, two rows per lane, one pivot broadcast for both -- was built and measured in round 2:
// 255 against 420 M nodes/s on config 5; at 256 registers it runs 2 waves/SIMD, which no longer hide the broadcast -> FMA
// latency of the pivot chain.  DESIGN.md section 4.)



// trace of the inverse from its rows (sum over the node's NP lanes; every lane gets it)
template <int NP>
__device__ __forceinline__ void panel_trace(const double (&ar)[NP], const double (&ai)[NP], int n, int r, double& tr,
                                            double& ti) {
    tr = 0.0;
    ti = 0.0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (j == r && r < n) {
            tr = ar[j];
            ti = ai[j];
        }
    }
#pragma unroll
    for (int off = NP / 2; off > 0; off >>= 1) {
        tr += __shfl_xor(tr, off, 64);
        ti += __shfl_xor(ti, off, 64);
    }
}

template <int NP>
__device__ __forceinline__ void rows_trace_resolvent_tri(int n, int r, double (&hr)[NP], double (&hi)[NP], double sw, double eta, double& tr,
                                                         double& ti);  // (defined with the Householder code below)

template <int NP, bool PAD>
__global__ __launch_bounds__(256) void gen_panel_kernel(GenArgs a) {
    extern __shared__ double2 lds_p[];
    constexpr int SLOTS = 256 / NP;
    const int n = a.n, nn = n * n, M = a.M;
    double2* coef = lds_p;                    // [M][nn] or [M][NP*NP]
    const int slot = threadIdx.x / NP, r = threadIdx.x % NP;
    const int64_t ngroups = a.nnodes / 15;
    for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int64_t node0 = g * 15;
        const int64_t parent = a.parents ? a.parents[node0] : 0;
        __syncthreads();  // the previous panel's readers are done with `coef`
        panel_stage<NP, PAD>(coef, a.src + parent * ((int64_t)M * nn), n, M);
        __syncthreads();
        for (int q0 = 0; q0 < 15; q0 += SLOTS) {
            const int q = q0 + slot;
            const bool act = q < 15;
            const int64_t k = node0 + (act ? q : 0);
            const double sw = a.sweep_per_node ? a.sweep_per_node[k] : a.sweep0;
            if constexpr (NP == 32) {
                if (a.integrand != ABZ_F_GLOC) {  // (uniform) 17...32 bands, traces: from the tridiagonal form, see rows_trace_resolvent_tri
                    double hr[NP], hi[NP];
                    double zr, zi, pr, pi, tr, ti;
                    const double xx = a.x[k] * a.inv_period;
                    sincospi(2.0 * xx, &zi, &zr);
                    sincospi(2.0 * ((double)a.first * xx), &pi, &pr);
                    panel_series_row<NP, PAD>(coef, n, M, zr, zi, pr, pi, r, hr, hi);
                    if (!PAD) {
#pragma unroll
                        for (int j = 0; j < NP; ++j) {
                            const bool real = r < n && j < n;
                            hr[j] = real ? hr[j] : 0.0;
                            hi[j] = real ? hi[j] : 0.0;
                        }
                    }
                    rows_trace_resolvent_tri<NP>(n, r, hr, hi, sw, a.p[0], tr, ti);
                    if (act && r == 0)
                        a.values[k * a.ncomp] = (a.integrand == ABZ_F_DOS)
                                                    ? make_double2(-ti * 0.31830988618379067153776752674503, 0.0)
                                                    : make_double2(tr, ti);
                    continue;
                }
            }
            double ar[NP], ai[NP];
            panel_inverse_row<NP, PAD>(coef, n, M, a.first, a.x[k] * a.inv_period, sw, a.p[0], r, ar, ai);
            if (a.integrand == ABZ_F_GLOC) {
                if (act && r < n) {
                    double2* out = a.values + k * a.ncomp;
#pragma unroll
                    for (int j = 0; j < NP; ++j)
                        if (j < n) out[r + n * j] = make_double2(ar[j], ai[j]);
                }
            } else {
                double tr, ti;
                panel_trace<NP>(ar, ai, n, r, tr, ti);
                if (act && r == 0)
                    a.values[k * a.ncomp] = (a.integrand == ABZ_F_DOS)
                                                ? make_double2(-ti * 0.31830988618379067153776752674503, 0.0)
                                                : make_double2(tr, ti);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Store-free PTR for n > 4: rule(f, B) of a full grid for resolvent-trace integrands.  One workgroup per
// grid line (its coefficient set staged in LDS, as in the panel kernels), a node per NP-lane group: the row
// of -H(k) is computed once and re-used for up to 4 sweep values (shift, invert, trace), the traces are
// accumulated per group and reduced at the end.  Nothing is written per node.
// ------------------------------------------------------------------------------------------
struct GenSumArgs {
    const double2* src;   // level-1 sets [nlines][M][n*n]
    const double2* tab;   // phase table [npt]
    double2* partial;     // [gridDim.x][nw]
    int64_t nlines;
    int n, M, first, npt, nw, is_dos;
    double eta;
    double sweep[4];
};

template <int NP, bool PAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void gen_grid_sum_kernel(GenSumArgs a) {
    extern __shared__ double2 lds_gs[];
    constexpr int SLOTS = 256 / NP;
    const int n = a.n, nn = n * n, M = a.M;
    double2* coef = lds_gs;
    double2* red = coef + (size_t)M * (PAD ? NP * NP : nn);  // [SLOTS][4]
    const int slot = threadIdx.x / NP, r = threadIdx.x % NP;
    int fm = a.first % a.npt;
    if (fm < 0) fm += a.npt;
    double accr[4] = {0.0, 0.0, 0.0, 0.0}, acci[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t line = blockIdx.x; line < a.nlines; line += gridDim.x) {
        __syncthreads();
        panel_stage<NP, PAD>(coef, a.src + line * ((int64_t)M * nn), n, M);
        __syncthreads();
        for (int i0 = 0; i0 < a.npt; i0 += SLOTS) {
            const int i1 = i0 + slot;
            const bool act = i1 < a.npt;
            const int ic = act ? i1 : 0;
            const double2 z = a.tab[ic];
            const double2 w = a.tab[(int)(((unsigned)fm * (unsigned)ic) % (unsigned)a.npt)];
            double hr[NP], hi[NP];
            panel_series_row<NP, PAD>(coef, n, M, z.x, z.y, w.x, w.y, r, hr, hi);
#pragma unroll 1
            for (int q = 0; q < a.nw; ++q) {  // uniform; one copy of the inversion in the code
                double ar[NP], ai[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    ar[j] = hr[j];
                    ai[j] = hi[j];
                }
                panel_shift_row<NP, PAD>(n, a.sweep[q], a.eta, r, ar, ai);
                double tr, ti;
                if constexpr (NP == 16) {  // broadcast inside the FMAs, trace only (see panel_inverse_trace_fmac)
                    panel_inverse_trace_fmac<!PAD>(n, r, ar, ai, tr, ti);
                } else {
                    panel_invert_rows<NP, PAD>(n, r, ar, ai);
                    panel_trace<NP>(ar, ai, n, r, tr, ti);
                }
                const double dr = !act ? 0.0 : (a.is_dos ? -ti * 0.31830988618379067153776752674503 : tr);
                const double di = (!act || a.is_dos) ? 0.0 : ti;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {  // static indices: the accumulators stay in registers
                    accr[qq] += (qq == q) ? dr : 0.0;
                    acci[qq] += (qq == q) ? di : 0.0;
                }
            }
        }
    }
    __syncthreads();
    if (r == 0)
        for (int q = 0; q < 4; ++q) red[slot * 4 + q] = make_double2(accr[q], acci[q]);
    __syncthreads();
    if (threadIdx.x < a.nw) {
        double sr = 0.0, si = 0.0;
        for (int sl = 0; sl < SLOTS; ++sl) {
            sr += red[sl * 4 + threadIdx.x].x;
            si += red[sl * 4 + threadIdx.x].y;
        }
        a.partial[(int64_t)blockIdx.x * a.nw + threadIdx.x] = make_double2(sr, si);
    }
}

bool gen_sum_supported(int n, int M, int npt, int integrand, bool herm) {
    if (big_supported(n) || big_inverse_sum_wanted(n, integrand, herm)) return big_sum_supported(n, M, npt, integrand, herm);
    if (n <= 4 || n > ABZ_MAX_BANDS || !herm || npt < 1 || npt >= 65536) return false;
    if (!(integrand == ABZ_F_DOS || integrand == ABZ_F_TRGLOC)) return false;
    const int np = n <= 8 ? 8 : (n <= 16 ? 16 : 32);
    const size_t rest = sizeof(double2) * (size_t)(256 / np) * 4;
    // (17...32 bands, and any set that does not fit the LDS whole, go through the tridiagonal kernel, which stages in chunks)
    return n > 16 ? abz_switch(SW_GEN_SUM_TRI) != 0 || sizeof(double2) * (size_t)M * n * n + rest <= 159 * 1024
                  : abz_switch(SW_GEN_SUM_TRI) != 0 || sizeof(double2) * (size_t)M * n * n + rest <= 150 * 1024;
}

static bool gen_sum_tri_wanted(const SumSpec& ss);
static int launch_gen_sum_tri(abz_ctx* ctx, const SumSpec& ss, double* out_reim);

int launch_gen_sum(abz_ctx* ctx, const SumSpec& ss, double* out_reim) {
    if (big_supported(ss.n) || big_inverse_sum_wanted(ss.n, ss.integrand, ss.herm)) return launch_big_sum(ctx, ss, out_reim);
    if (lane_sum_supported(ss.n, ss.M, ss.first, ss.npt, ss.integrand, ss.n_sweep)) return launch_lane_sum(ctx, ss, out_reim);
    if (gen_sum_tri_wanted(ss)) {
        const int rc = launch_gen_sum_tri(ctx, ss, out_reim);
        if (rc != ABZ_ERR_UNSUPPORTED) return rc;
    }
    const int n = ss.n, M = ss.M;
    const int np = n <= 8 ? 8 : (n <= 16 ? 16 : 32);
    const size_t rest = sizeof(double2) * (size_t)(256 / np) * 4;
    size_t lds = sizeof(double2) * (size_t)M * np * np + rest;
    const bool pad = lds <= 150 * 1024;
    if (!pad) lds = sizeof(double2) * (size_t)M * n * n + rest;
    if (lds > 160 * 1024) return ABZ_ERR_UNSUPPORTED;
    const int64_t blocks = std::min<int64_t>(ss.nlines, 256 * 2);
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * 4));
    if (rc) return rc;
    if ((rc = ctx->scratch[2].reserve(sizeof(double2) * 4))) return rc;
    GenSumArgs a;
    a.src = ss.src;
    a.tab = ss.tab;
    a.partial = ctx->scratch[1].as<double2>();
    a.nlines = ss.nlines;
    a.n = n;
    a.M = M;
    a.first = ss.first;
    a.npt = ss.npt;
    a.is_dos = ss.integrand == ABZ_F_DOS ? 1 : 0;
    a.eta = ss.params[0];
    for (int s0 = 0; s0 < ss.n_sweep; s0 += 4) {
        a.nw = std::min(4, ss.n_sweep - s0);
        for (int q = 0; q < 4; ++q) a.sweep[q] = q < a.nw ? ss.sweep_host[s0 + q] : 0.0;
        {
            ProfScope ps(ctx, ABZ_K_EVAL);
#define ABZ_GS2(NPV, PV)                                                                                              \
    ABZ_HIP(hipFuncSetAttribute((const void*)gen_grid_sum_kernel<NPV, PV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds));                                                                           \
    hipLaunchKernelGGL((gen_grid_sum_kernel<NPV, PV>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
#define ABZ_GS(NPV)   \
    if (pad) {        \
        ABZ_GS2(NPV, true) \
    } else {          \
        ABZ_GS2(NPV, false) \
    }
            if (np == 8) {
                ABZ_GS(8)
            } else if (np == 16) {
                ABZ_GS(16)
            } else {
                ABZ_GS(32)
            }
#undef ABZ_GS
#undef ABZ_GS2
            ABZ_HIP(hipGetLastError());
            rc = launch_final_reduce(ctx, a.partial, blocks, a.nw, ss.scale, ctx->scratch[2].as<double2>());
            if (rc) return rc;
        }
        ABZ_HIP(hipMemcpyAsync(out_reim + 2 * (size_t)s0, ctx->scratch[2].p, sizeof(double2) * (size_t)a.nw, hipMemcpyDeviceToHost,
                               ctx->stream));
        ABZ_HIP(hipStreamSynchronize(ctx->stream));
    }
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// Rule builds with eigenvalues (and eigenvectors) on full grids for 5..16 bands: the row layout of the panel
// kernels (lane r of a node's NP lanes owns row r) with a PARALLEL-ORDER two-sided Jacobi.  One step of the
// round-robin tournament rotates NP/2 disjoint pairs (p, q) at once:
//   rows     row_p' = c row_p - s conj(g) row_q,  row_q' = s row_p + c conj(g) row_q   -- partner lanes swap rows
//            (`ds_bpermute_b32`), the p lane computes the angle and hands it to the q lane
//   columns  x' = c x - s g y,  y' = s x + c g y  on (x, y) = (a_rp, a_rq) of every row, for every pair of the step
//            -- local to the lane, the pairs' (c, s, g) by `group_bcast` (pair indices are compile-time: the NP-1
//            steps are separate template instances)
// The diagonal is carried in a scalar per lane and updated by +- t |a_pq| as in the per-lane solver (device_math.h).
// NP/2 rotations per ~450 instructions instead of one rotation per 2 wave barriers of `wave_eig`.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_read(double v, int addr4) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr4, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr4, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// pair I (0 .. NP/2-1) of step T (0 .. NP-2) of the round-robin ordering on NP players: {0, T+1}, {(T+i)%m+1, (T-i)%m+1}
template <int NP, int T, int I>
struct RRPair {
    static constexpr int m = NP - 1;
    static constexpr int a = I == 0 ? 0 : (T + I) % m + 1;
    static constexpr int b = I == 0 ? T + 1 : (T - I + m) % m + 1;
    static constexpr int lo = a < b ? a : b;
    static constexpr int hi = a < b ? b : a;
};

template <int NP, int T, int I, bool VEC>
__device__ __forceinline__ void jacobi_columns(double c, double s, double gr, double gi, double (&ar)[NP], double (&ai)[NP],
                                               double (&vr)[NP], double (&vi)[NP]) {
    constexpr int P = RRPair<NP, T, I>::lo, Q = RRPair<NP, T, I>::hi;
    const double cc = group_bcast<NP, P>(c), ss = group_bcast<NP, P>(s);
    const double ggr = group_bcast<NP, P>(gr), ggi = group_bcast<NP, P>(gi);
    const double sgr = ss * ggr, sgi = ss * ggi, cgr = cc * ggr, cgi = cc * ggi;
    {
        const double xr = ar[P], xi = ai[P], yr = ar[Q], yi = ai[Q];
        ar[P] = cc * xr - (sgr * yr - sgi * yi);
        ai[P] = cc * xi - (sgr * yi + sgi * yr);
        ar[Q] = ss * xr + (cgr * yr - cgi * yi);
        ai[Q] = ss * xi + (cgr * yi + cgi * yr);
    }
    if constexpr (VEC) {
        const double xr = vr[P], xi = vi[P], yr = vr[Q], yi = vi[Q];
        vr[P] = cc * xr - (sgr * yr - sgi * yi);
        vi[P] = cc * xi - (sgr * yi + sgi * yr);
        vr[Q] = ss * xr + (cgr * yr - cgi * yi);
        vi[Q] = ss * xi + (cgr * yi + cgi * yr);
    }
}

template <int NP, int T, bool VEC, int... I>
__device__ __forceinline__ void jacobi_all_columns(double c, double s, double gr, double gi, double (&ar)[NP], double (&ai)[NP],
                                                   double (&vr)[NP], double (&vi)[NP], std::integer_sequence<int, I...>) {
    (jacobi_columns<NP, T, I, VEC>(c, s, gr, gi, ar, ai, vr, vi), ...);
}

template <int NP, int T, bool VEC>
__device__ __forceinline__ void jacobi_step(int r, int lane, double tiny, double& dg, double (&ar)[NP], double (&ai)[NP],
                                            double (&vr)[NP], double (&vi)[NP]) {
    constexpr int m = NP - 1;
    int pr;  // this lane's partner in step T
    {
        const int x = r - 1;
        int y = (2 * T - x) % m;
        if (y < 0) y += m;
        pr = (r == 0) ? T + 1 : (x == T ? 0 : y + 1);
    }
    const int addr = ((lane & ~(NP - 1)) | pr) << 2;
    const bool prole = r < pr;
    double er = 0.0, ei = 0.0;  // a[r][pr]
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (j == pr) {
            er = ar[j];
            ei = ai[j];
        }
    }
    const double dgp = lane_read(dg, addr);
    // the pair's rotation from the p lane's data (alpha = a_pq, d = a_qq - a_pp)
    const double b2 = er * er + ei * ei;
    const bool ok = b2 > tiny;
    const double b2s = ok ? b2 : 1.0;
    const double rb = rsqrt_nr(b2s), b = b2s * rb;
    double gr = er * rb, gi = -ei * rb;  // g = conj(alpha) / |alpha|
    const double d = dgp - dg;
    const double r2 = fma(d, d, 4.0 * b2s);
    const double t = copysign(2.0 * b, d) * rcp_nr(fabs(d) + r2 * rsqrt_nr(r2));
    double c = rsqrt_nr(fma(t, t, 1.0));
    double s = t * c;
    double tb = t * b;
    if (!ok) {
        c = 1.0;
        s = 0.0;
        tb = 0.0;
        gr = 1.0;
        gi = 0.0;
    }
    {  // the q lane works with exactly the p lane's numbers
        const double c2 = lane_read(c, addr), s2 = lane_read(s, addr), g2r = lane_read(gr, addr), g2i = lane_read(gi, addr),
                     tb2 = lane_read(tb, addr);
        if (!prole) {
            c = c2;
            s = s2;
            gr = g2r;
            gi = g2i;
            tb = tb2;
        }
    }
    dg += prole ? -tb : tb;
    // rows: new = ks * mine + kp * partner's
    const double ksr = prole ? c : c * gr, ksi = prole ? 0.0 : -c * gi;
    const double kpr = prole ? -s * gr : s, kpi = prole ? s * gi : 0.0;
#pragma unroll
    for (int j0 = 0; j0 < NP; j0 += 8) {
        double br[8], bi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            br[j] = lane_read(ar[j0 + j], addr);
            bi[j] = lane_read(ai[j0 + j], addr);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double xr = ar[j0 + j], xi = ai[j0 + j];
            ar[j0 + j] = (ksr * xr - ksi * xi) + (kpr * br[j] - kpi * bi[j]);
            ai[j0 + j] = (ksr * xi + ksi * xr) + (kpr * bi[j] + kpi * br[j]);
        }
    }
    jacobi_all_columns<NP, T, VEC>(c, s, gr, gi, ar, ai, vr, vi, std::make_integer_sequence<int, NP / 2>());
    // the rotated pair's off-diagonal entry is zero by construction: make it exact (left to rounding it stalls at
    // ~1e-16 ||A|| and the sweeps never see convergence)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (j == pr) {
            ar[j] = 0.0;
            ai[j] = 0.0;
        }
    }
}

template <int NP, bool VEC, int... T>
__device__ __forceinline__ void jacobi_sweep(int r, int lane, double tiny, double& dg, double (&ar)[NP], double (&ai)[NP],
                                             double (&vr)[NP], double (&vi)[NP], std::integer_sequence<int, T...>) {
    (jacobi_step<NP, T, VEC>(r, lane, tiny, dg, ar, ai, vr, vi), ...);
}


// Eigenvalues (ascending rank of this lane's eigenvalue in `rank`, the value in `dg`) of the Hermitian matrix whose
// row r this lane holds (rows / columns >= n: zero, they never couple); VEC: row r of the eigenvector matrix in vr/vi,
// its column j belongs to the eigenvalue held by lane j.
template <int NP, bool VEC>
__device__ __forceinline__ void rows_eig(int n, int r, int lane, double (&ar)[NP], double (&ai)[NP], double (&vr)[NP],
                                         double (&vi)[NP], double& dg, int& rank) {
    dg = 0.0;
    double nrm = 0.0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (j == r) {
            dg = ar[j];
            ai[j] = 0.0;
        }
        nrm += ar[j] * ar[j] + ai[j] * ai[j];
        if constexpr (VEC) {
            vr[j] = (j == r) ? 1.0 : 0.0;
            vi[j] = 0.0;
        }
    }
    const double tiny = 1e-34 * group_sum<NP>(nrm);  // |a_pq| <= 1e-17 ||A||_F counts as zero
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off2 = 0.0;
#pragma unroll
        for (int j = 0; j < NP; ++j) off2 += (j == r) ? 0.0 : ar[j] * ar[j] + ai[j] * ai[j];
        off2 = group_sum<NP>(off2);
        if (!__any(off2 > tiny)) break;  // wave-uniform: the nodes of a wave sweep together
        jacobi_sweep<NP, VEC>(r, lane, tiny, dg, ar, ai, vr, vi, std::make_integer_sequence<int, NP - 1>());
    }
    const double mine = r < n ? dg : __builtin_huge_val();  // padding rows rank last
    double vals[NP];
    group_gather<NP>(mine, vals, std::make_integer_sequence<int, NP>());
    rank = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) rank += (vals[j] < mine || (vals[j] == mine && j < r)) ? 1 : 0;
}

struct GenEigArgs {
    const double2* src;  // level-1 coefficient sets, one per grid line
    const double2* tab;
    PlaneView H, E, U;
    int64_t nlines;
    const int64_t* run_start = nullptr;  // irregular lists: line l owns nodes [run_start[l], run_start[l + 1]) with grid indices gi
    const int32_t* gi = nullptr;
    int n, M, first, npt;
    int herm = 0;  // Hermitian series: H(k) leaves the kernel as its upper triangle (mirrored into the full layout's lower planes)
    double* tri = nullptr;  // eigenvalues only: the tridiagonals (d | |e|^2, [2 NP][tri_nk]) for tri_eig_kernel instead of E
    int64_t tri_nk = 0;
    int fold = 0;  // Hermitian level-1 sets with first = -(M - 1) / 2: the series from c_0 and c_f +- c_f^T (half the FMAs)
    int mc = 0;    // > 0 (unpadded layout): the set is staged mc coefficients at a time, per group of nodes (it does not fit the LDS whole)
};




// tr inv((sw + i eta) I - H) of the Hermitian matrix whose NEGATED row r this lane holds (hr, hi = row r of -H; rows /
// columns >= n: zero): Householder tridiagonalisation, then p'(z) / p(z) by the three-term recurrence of the tridiagonal
// (see the sweep kernel below, which runs the same recurrence for many values).  Every lane of the node gets the trace.
// The 32-lane instances of the IAI kernels use it instead of the Gauss-Jordan inverse: the unrolled elimination of a
// 32 x 32 matrix leaves most of the row arrays in scratch memory (7 M nodes/s), and the trace needs no inverse.
template <int NP>
__device__ __forceinline__ void rows_trace_resolvent_tri(int n, int r, double (&hr)[NP], double (&hi)[NP], double sw, double eta, double& tr,
                                                         double& ti) {
    double e2[NP], b[NP];
    hh_steps<NP>(n, r, hr, hi, e2, std::make_integer_sequence<int, NP>());
    diag_gather<NP>(hr, b, std::make_integer_sequence<int, NP>());
    double rad = 0.0, eprev = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < n) {
            const double en = (i + 1 < n) ? sqrt(e2[i]) : 0.0;
            rad = fmax(rad, fabs(b[i]) + eprev + en);
            eprev = en;
        }
    }
    const double sc = rcp_nr(rad + fabs(sw) + eta);
    const double zr = sw * sc, zi = eta * sc;
    double p0r = 1.0, p0i = 0.0, p1r = fma(b[0], sc, zr), p1i = zi;
    double q0r = 0.0, q0i = 0.0, q1r = 1.0, q1i = 0.0;
#pragma unroll
    for (int i = 1; i < NP; ++i) {
        if (i < n) {  // uniform
            const double ar = fma(b[i], sc, zr), ai = zi;
            const double ee = e2[i - 1] * sc * sc;
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
    }
    const double ip = rcp_nr(p1r * p1r + p1i * p1i) * sc;
    tr = (q1r * p1r + q1i * p1i) * ip;
    ti = (q1i * p1r - q1r * p1i) * ip;
}

// ------------------------------------------------------------------------------------------
// Store-free PTR sums of resolvent traces for MANY swept values (5..16 bands, Hermitian series): one Householder
// tridiagonalisation per node, then tr inv(z I - H) = p'(z) / p(z) from the three-term recurrence of the real
// symmetric tridiagonal (d, |e|^2) -- O(n) per swept value instead of an O(n^3) Gauss-Jordan inversion each:
//   p_k = (z + b_k) p_{k-1} - |e_{k-1}|^2 p_{k-2},   p'_k = p_{k-1} + (z + b_k) p'_{k-1} - |e_{k-1}|^2 p'_{k-2}
// (b = the diagonal of tridiag(-H), the row kernels carry -H).  Lane r of a node's NP lanes takes swept value r: up
// to NP values per pass, every lane busy, no cross-lane traffic after the reduction.  Im z = eta > 0 keeps p away from
// zero; the matrix is scaled to unit Gershgorin radius so that p stays far inside the double range for any spectrum.
// 16 bands, 16 swept values: ~3.6 k instructions per four nodes against 16 x 1.3 k for the inversions.
// ------------------------------------------------------------------------------------------
struct GenSumTriArgs {
    const double2* src;
    const double2* tab;
    double2* partial;  // [gridDim.x][nw]
    int64_t nlines;
    int n, M, first, npt, nw, is_dos;
    int mc = 0;  // > 0 (unpadded layout): the set is staged mc coefficients at a time, per group of nodes (it does not fit the LDS whole)
    double eta;
    double sweep[32];
};

template <int NP, bool PAD>
__global__ __launch_bounds__(256) void gen_grid_sum_tri_kernel(GenSumTriArgs a) {
    extern __shared__ double2 lds_gt[];
    constexpr int SLOTS = 256 / NP;
    const int n = a.n, nn = n * n, M = a.M;
    double2* coef = lds_gt;
    double2* red = coef + (size_t)((!PAD && a.mc > 0) ? a.mc : M) * (PAD ? NP * NP : nn);  // [SLOTS][NP]
    const int slot = threadIdx.x / NP, r = threadIdx.x % NP;
    int fm = a.first % a.npt;
    if (fm < 0) fm += a.npt;
    double mysw = 0.0;  // this lane's swept value (lane r of the node takes value r)
#pragma unroll
    for (int q = 0; q < NP; ++q) mysw = (q == r) ? a.sweep[q] : mysw;
    const bool lane_act = r < a.nw;
    double accr = 0.0, acci = 0.0;
    const bool chunked = !PAD && a.mc > 0;
    for (int64_t line = blockIdx.x; line < a.nlines; line += gridDim.x) {
        if (!chunked) {
            __syncthreads();
            panel_stage<NP, PAD>(coef, a.src + line * ((int64_t)M * nn), n, M);
            __syncthreads();
        }
        for (int i0 = 0; i0 < a.npt; i0 += SLOTS) {
            const bool wave_on = i0 + (int)(threadIdx.x >> 6) * (64 / NP) < a.npt;
            if (!chunked && !wave_on) continue;  // no node for this wave (wave-level sync only below; chunked: the staging barriers need every wave)
            const int i1 = i0 + slot;
            const bool act = i1 < a.npt;
            const int ic = act ? i1 : 0;
            const double2 z = a.tab[ic];
            const double2 w = a.tab[(int)(((unsigned)fm * (unsigned)ic) % (unsigned)a.npt)];
            double hr[NP], hi[NP];
            if (chunked) {
                double pr = w.x, pi = w.y;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    hr[j] = 0.0;
                    hi[j] = 0.0;
                }
                for (int m0 = 0; m0 < M; m0 += a.mc) {
                    const int mcur = min(a.mc, M - m0);
                    __syncthreads();  // the previous chunk's readers are done
                    panel_stage<NP, false>(coef, a.src + line * ((int64_t)M * nn) + (int64_t)m0 * nn, n, mcur);
                    __syncthreads();
                    if (wave_on) panel_series_row_chunk<NP>(coef, n, mcur, z.x, z.y, pr, pi, r, hr, hi);
                }
                if (!wave_on) continue;
            } else {
                panel_series_row<NP, PAD>(coef, n, M, z.x, z.y, w.x, w.y, r, hr, hi);  // row r of B = -H(k)
            }
            if (!PAD) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const bool real = r < n && j < n;
                    hr[j] = real ? hr[j] : 0.0;
                    hi[j] = real ? hi[j] : 0.0;
                }
            }
            double e2[NP], b[NP];
            hh_steps<NP>(n, r, hr, hi, e2, std::make_integer_sequence<int, NP>());
            diag_gather<NP>(hr, b, std::make_integer_sequence<int, NP>());
            // scale: s = 1 / (Gershgorin radius of T + |z|): the recurrence runs on T / s-free numbers of size <= 1
            double rad = 0.0, eprev = 0.0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (i < n) {
                    const double en = (i + 1 < n) ? sqrt(e2[i]) : 0.0;
                    rad = fmax(rad, fabs(b[i]) + eprev + en);
                    eprev = en;
                }
            }
            const double sc = rcp_nr(rad + fabs(mysw) + a.eta);
            const double zr = mysw * sc, zi = a.eta * sc;
            // p(z) = det(z I + T) and p'(z) by the three-term recurrence (everything scaled by sc)
            double p0r = 1.0, p0i = 0.0, p1r = fma(b[0], sc, zr), p1i = zi;  // p_0, p_1
            double q0r = 0.0, q0i = 0.0, q1r = 1.0, q1i = 0.0;              // p'_0, p'_1 (per unit of the scaled variable)
#pragma unroll
            for (int i = 1; i < NP; ++i) {
                if (i < n) {  // uniform
                    const double ar = fma(b[i], sc, zr), ai = zi;
                    const double ee = e2[i - 1] * sc * sc;
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
            }
            // tr inv(z I - H) = p'(z) / p(z); the scaled variable gives sc * (q / p)
            const double ip = rcp_nr(p1r * p1r + p1i * p1i) * sc;
            const double tr = (q1r * p1r + q1i * p1i) * ip, ti = (q1i * p1r - q1r * p1i) * ip;
            const bool use = act && lane_act;
            accr += !use ? 0.0 : (a.is_dos ? -ti * 0.31830988618379067153776752674503 : tr);
            acci += (!use || a.is_dos) ? 0.0 : ti;
        }
    }
    __syncthreads();
    red[slot * NP + r] = make_double2(accr, acci);
    __syncthreads();
    if (threadIdx.x < a.nw) {
        double sr = 0.0, si = 0.0;
        for (int sl = 0; sl < SLOTS; ++sl) {
            sr += red[sl * NP + threadIdx.x].x;
            si += red[sl * NP + threadIdx.x].y;
        }
        a.partial[(int64_t)blockIdx.x * a.nw + threadIdx.x] = make_double2(sr, si);
    }
}

// sweeps of at least 3 values on 5..32 bands take the tridiagonal route
static bool gen_sum_tri_wanted(const SumSpec& ss) {
    // (17...32 bands, two nodes per wave: the tridiagonal route wins for a single value as well -- 0.9 against 4.2 ms at 32^3)
    const int np = ss.n <= 8 ? 8 : 16;
    const bool whole = sizeof(double2) * (size_t)ss.M * ss.n * ss.n + sizeof(double2) * (size_t)(256 / np) * 4 <= 150 * 1024;  // gen_sum_supported's bound
    return abz_switch(SW_GEN_SUM_TRI) && ss.n > 4 && ss.n <= 32 && (ss.n_sweep >= 3 || ss.n > 16 || !whole);
}

static int launch_gen_sum_tri(abz_ctx* ctx, const SumSpec& ss, double* out_reim) {
    const int n = ss.n, M = ss.M;
    const int np = n <= 8 ? 8 : (n <= 16 ? 16 : 32);
    const size_t rest = sizeof(double2) * (size_t)256;  // [SLOTS][NP] partial sums
    size_t lds = sizeof(double2) * (size_t)M * np * np + rest;
    // (17...32 bands: the kernel fits 256 registers -- the set without its zero padding when that lets two workgroups share a CU)
    const size_t bare = sizeof(double2) * (size_t)M * n * n + rest;
    const bool pad = np == 32 ? (lds <= 72 * 1024 || bare > 72 * 1024) && lds <= 150 * 1024 : lds <= 150 * 1024;
    if (!pad) lds = bare;
    int mc = 0;
    if (lds > 160 * 1024) {  // the set does not fit the LDS whole: staged in chunks of mc coefficients (two workgroups per CU)
        mc = (int)((72 * 1024 - rest) / (sizeof(double2) * (size_t)n * n));
        if (mc < 1) return ABZ_ERR_UNSUPPORTED;
        lds = sizeof(double2) * (size_t)mc * n * n + rest;
    }
    const int64_t blocks = std::min<int64_t>(ss.nlines, 256 * 4);
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * 32));
    if (rc) return rc;
    if ((rc = ctx->scratch[2].reserve(sizeof(double2) * 32))) return rc;
    GenSumTriArgs a;
    a.src = ss.src;
    a.tab = ss.tab;
    a.partial = ctx->scratch[1].as<double2>();
    a.nlines = ss.nlines;
    a.n = n;
    a.M = M;
    a.first = ss.first;
    a.npt = ss.npt;
    a.is_dos = ss.integrand == ABZ_F_DOS ? 1 : 0;
    a.eta = ss.params[0];
    a.mc = mc;
    for (int s0 = 0; s0 < ss.n_sweep; s0 += np) {
        a.nw = std::min(np, ss.n_sweep - s0);
        for (int q = 0; q < 32; ++q) a.sweep[q] = q < a.nw ? ss.sweep_host[s0 + q] : 0.0;
        {
            ProfScope ps(ctx, ABZ_K_EVAL);
            if (np == 32) {  // 17...32 bands: two nodes per wave
                if (pad) {
                    ABZ_HIP(hipFuncSetAttribute((const void*)gen_grid_sum_tri_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((gen_grid_sum_tri_kernel<32, true>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
                } else {
                    ABZ_HIP(hipFuncSetAttribute((const void*)gen_grid_sum_tri_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((gen_grid_sum_tri_kernel<32, false>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
                }
            } else if (np == 8) {
                if (pad) {
                    ABZ_HIP(hipFuncSetAttribute((const void*)gen_grid_sum_tri_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((gen_grid_sum_tri_kernel<8, true>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
                } else {
                    ABZ_HIP(hipFuncSetAttribute((const void*)gen_grid_sum_tri_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((gen_grid_sum_tri_kernel<8, false>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
                }
            } else {
                if (pad) {
                    ABZ_HIP(hipFuncSetAttribute((const void*)gen_grid_sum_tri_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((gen_grid_sum_tri_kernel<16, true>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
                } else {
                    ABZ_HIP(hipFuncSetAttribute((const void*)gen_grid_sum_tri_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL((gen_grid_sum_tri_kernel<16, false>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
                }
            }
            ABZ_HIP(hipGetLastError());
            rc = launch_final_reduce(ctx, a.partial, blocks, a.nw, ss.scale, ctx->scratch[2].as<double2>());
            if (rc) return rc;
        }
        ABZ_HIP(hipMemcpyAsync(out_reim + 2 * (size_t)s0, ctx->scratch[2].p, sizeof(double2) * (size_t)a.nw, hipMemcpyDeviceToHost,
                               ctx->stream));
        ABZ_HIP(hipStreamSynchronize(ctx->stream));
    }
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// Eigenvectors for 9..16 bands in the row layout (GGR builds; ref src/dos_ggr.jl:31-44 calls LAPACK's eigen there):
// eigenvalues by Householder + bisection (above), then per band b INVERSE ITERATION with the machinery of the IAI panel
// kernels -- X = inv(H - (lambda_b + i eps) I) by the in-register Gauss-Jordan elimination (the complex shift keeps every
// leading minor regular, so no pivoting), x <- X x three times from a band-dependent start vector (contamination
// (eps / gap)^k), modified Gram-Schmidt against earlier members of the same cluster (|lambda_b - lambda_c| <= 1e-8 scale:
// degenerate bands, where any orthonormal basis of the eigenspace is an answer).  ~3 k instructions per band and four
// matrices; the 16-row parallel-order Jacobi with accumulated rotations spilled 4.6 KB and the wave-per-node Jacobi
// took 11.2 ms for 24^3 nodes.  Lane r ends with row r of U (component r of every eigenvector), band b in column b.
// ------------------------------------------------------------------------------------------
template <int NP, int... J>
__device__ __forceinline__ void row_matvec(const double (&xr_)[NP], const double (&xi_)[NP], double vr, double vi, double& outr,
                                           double& outi, std::integer_sequence<int, J...>) {
    outr = 0.0;
    outi = 0.0;
    // out_r = sum_j X[r][j] v_j, v_j from lane j of the group
    ((void)([&] {
         const double br = group_bcast<NP, J>(vr), bi = group_bcast<NP, J>(vi);
         outr = fma(xr_[J], br, outr);
         outr = fma(-xi_[J], bi, outr);
         outi = fma(xr_[J], bi, outi);
         outi = fma(xi_[J], br, outi);
     }()),
     ...);
}

template <int NP>
__device__ __forceinline__ void rows_eigvecs_invit(int n, int r, int lane, const double (&hr)[NP], const double (&hi)[NP], double myeig,
                                                   double (&vr)[NP], double (&vi)[NP]) {
    const int gbase = lane & ~(NP - 1);
    // scale of the spectrum (for the shift and the cluster test)
    double sc = fabs(myeig);
#pragma unroll
    for (int off = NP / 2; off > 0; off >>= 1) sc = fmax(sc, __shfl_xor(sc, off, 64));
    double offd = 0.0;
#pragma unroll
    for (int j = 0; j < NP; ++j) offd += fabs(hr[j]) + fabs(hi[j]);
#pragma unroll
    for (int off = NP / 2; off > 0; off >>= 1) offd = fmax(offd, __shfl_xor(offd, off, 64));
    sc = fmax(fmax(sc, offd), 1e-300);
    const double eps = 1e-10 * sc, ctol = 1e-8 * sc;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        vr[j] = 0.0;
        vi[j] = 0.0;
    }
    for (int b = 0; b < n; ++b) {  // uniform
        const double lam = __shfl(myeig, gbase + b, 64);
        double wr[NP], wi[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            wr[j] = hr[j];
            wi[j] = hi[j];
            if (j == r) {  // rows >= n of the zero-padded matrix: identity rows, decoupled
                wr[j] = r < n ? hr[j] - lam : 1.0;
                wi[j] = r < n ? hi[j] - eps : 0.0;
            }
        }
        panel_invert_rows<NP, true>(n, r, wr, wi);
        // start vector: band dependent, so that the members of a degenerate cluster start differently
        double xr = 0.0, xi = 0.0;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const double y = (j < n) ? 1.0 + 0.61803398874989484820 * (double)(((j + 1) * (b + 3)) % 7) : 0.0;
            xr = fma(wr[j], y, xr);
            xi = fma(wi[j], y, xi);
        }
        for (int it = 0; it < 3; ++it) {
            if (it > 0) {
                double yr, yi;
                row_matvec<NP>(wr, wi, xr, xi, yr, yi, std::make_integer_sequence<int, NP>());
                xr = yr;
                xi = yi;
            }
            if (r >= n) {
                xr = 0.0;
                xi = 0.0;
            }
            // modified Gram-Schmidt against earlier bands of the same cluster (wave-uniform skip: clusters are rare)
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                const double lc = __shfl(myeig, gbase + (c < n ? c : 0), 64);
                const bool need = c < b && fabs(lc - lam) <= ctol;
                if (__builtin_amdgcn_ballot_w64(need) != 0) {
                    double dr = vr[c] * xr + vi[c] * xi, di = vr[c] * xi - vi[c] * xr;  // conj(u_c) x, summed over the rows
                    dr = group_sum<NP>(dr);
                    di = group_sum<NP>(di);
                    if (need) {
                        xr -= dr * vr[c] - di * vi[c];
                        xi -= dr * vi[c] + di * vr[c];
                    }
                }
            }
            const double nn = group_sum<NP>(xr * xr + xi * xi);
            const double inv = nn > 0.0 ? rsqrt_nr(nn) : 0.0;
            xr *= inv;
            xi *= inv;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            vr[j] = (j == b) ? xr : vr[j];
            vi[j] = (j == b) ? xi : vi[j];
        }
    }
}

// Band velocities v_b = Re sum_{r,c} conj(U[r][b]) D[r][c] U[c][b], NP lanes per node, lane r owning row r of
// T = D U: the node's U and D = dH/dk_j are staged in LDS ([slot][row][col], 16 B per element), the loop over c reads
// D[r][c] (one element per lane) and the row U[c][.] (the same 16 B for all lanes of the node: broadcast reads) and
// accumulates T[r][b] in registers; the sum over r is a group reduction per band.  One thread per (node, band) re-read
// the whole D (4 KB at 16 bands) sixteen times per node: 0.25 ms per direction for 24^3 nodes.  (A register-only
// version -- rows of U travelling by DPP broadcasts -- spilled 670 registers at 16 rows whatever the ordering.)
template <int NP, int B>
__device__ __forceinline__ void vel_store1(const double2* __restrict__ urow, const double (&tr)[NP], const double (&ti)[NP], bool act, int r,
                                           double* __restrict__ vo, int pitch) {
    const double2 u = urow[B];
    const double v = group_sum<NP>(u.x * tr[B] + u.y * ti[B]);
    if (act && r == B) vo[(int64_t)B * pitch] = v;
}
template <int NP, int... B>
__device__ __forceinline__ void vel_store(const double2* __restrict__ urow, const double (&tr)[NP], const double (&ti)[NP], bool act, int r,
                                          double* __restrict__ vo, int pitch, std::integer_sequence<int, B...>) {
    (vel_store1<NP, B>(urow, tr, ti, act, r, vo, pitch), ...);
}

template <int NP>
__global__ __launch_bounds__(128) void gen_velocity_rows_kernel(PlaneView Uv, PlaneView Dv, PlaneView Vv, int64_t nk, int n) {
    extern __shared__ double2 lds_v[];  // [SLOTS][NP][NP] U | [SLOTS][NP][NP] D
    constexpr int SLOTS = 128 / NP;
    const int slot = threadIdx.x / NP, r = threadIdx.x % NP;
    const int64_t k = (int64_t)blockIdx.x * SLOTS + slot;
    const bool act = k < nk && r < n;
    const int64_t kk = k < nk ? k : nk - 1;
    double2* const us = lds_v + (size_t)slot * NP * NP;
    double2* const ds = lds_v + (size_t)(SLOTS + slot) * NP * NP;
    {
        const double* __restrict__ u = Uv.base + view_off(Uv, kk);
        const double* __restrict__ dm = Dv.base + view_off(Dv, kk);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const bool in = r < n && j < n;
            const int rr = in ? r : 0, jj = in ? j : 0;
            const double a0 = u[(int64_t)(2 * (rr + n * jj)) * Uv.pitch], a1 = u[(int64_t)(2 * (rr + n * jj) + 1) * Uv.pitch];
            const double b0 = dm[(int64_t)(2 * (rr + n * jj)) * Dv.pitch], b1 = dm[(int64_t)(2 * (rr + n * jj) + 1) * Dv.pitch];
            us[r * NP + j] = in ? make_double2(a0, a1) : make_double2(0.0, 0.0);
            ds[r * NP + j] = in ? make_double2(b0, b1) : make_double2(0.0, 0.0);
        }
    }
    __syncthreads();
    double tr[NP], ti[NP];
#pragma unroll
    for (int b = 0; b < NP; ++b) {
        tr[b] = 0.0;
        ti[b] = 0.0;
    }
    for (int c = 0; c < n; ++c) {
        const double2 dd = ds[r * NP + c];
        const double2* __restrict__ uc = us + c * NP;
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            const double2 u = uc[b];
            tr[b] = fma(dd.x, u.x, tr[b]);
            tr[b] = fma(-dd.y, u.y, tr[b]);
            ti[b] = fma(dd.x, u.y, ti[b]);
            ti[b] = fma(dd.y, u.x, ti[b]);
        }
    }
    double* __restrict__ vo = Vv.base + view_off(Vv, kk);
    vel_store<NP>(us + r * NP, tr, ti, act, r, vo, Vv.pitch, std::make_integer_sequence<int, NP>());
}

static bool launch_gen_velocity_rows(abz_ctx* ctx, int n, PlaneView U, PlaneView dH, PlaneView Vj, int64_t nk) {
    if (n <= 4 || n > 16) return false;
    if (n <= 8)
        hipLaunchKernelGGL(gen_velocity_rows_kernel<8>, dim3((unsigned)cdiv2(nk, 16)), dim3(128), sizeof(double2) * 2 * 16 * 64, ctx->stream, U, dH,
                           Vj, nk, n);
    else
        hipLaunchKernelGGL(gen_velocity_rows_kernel<16>, dim3((unsigned)cdiv2(nk, 8)), dim3(128), sizeof(double2) * 2 * 8 * 256, ctx->stream, U, dH,
                           Vj, nk, n);
    return true;
}

// TRI: eigenvalues only by Householder + Sturm bisection (its own instance: the Jacobi path of the same kernel costs
// it 90 more registers and the second wave per SIMD)
// SPLIT: the tridiagonals go to tri_eig_kernel (an instance of its own: with the bisection beside it the kernel spilled 33
// registers)
template <int NP, bool PAD, bool VEC, bool TRI, bool SPLIT = false>
__global__ __launch_bounds__(256, (TRI && !VEC && NP <= 16) ? 2 : 1) void gen_grid_eig_kernel(GenEigArgs a) {
    static_assert(!SPLIT || (TRI && !VEC), "the split build computes eigenvalues only");
    static_assert(NP <= 16 || SPLIT || !TRI, "17...32 bands: eigenvalues through the tridiagonal kernel");
    static_assert(!(VEC && TRI) || PAD, "inverse iteration works on the zero-padded layout");
    extern __shared__ double2 lds_ge[];
    constexpr int SLOTS = 256 / NP;
    constexpr int TS = SLOTS + 1;        // tile row stride (doubles): one bank further per plane
    constexpr int PC = NP == 32 ? 176 : (NP == 16 ? 144 : 64);  // planes per tile pass: columns 0...11 of 16 rows (16 nodes) / all 64 of 8 rows (32 nodes) / at most 171 of 32 rows (8 nodes)
    const int n = a.n, nn = n * n, M = a.M;
    double2* coef = lds_ge;
    const bool chunked = !PAD && a.mc > 0;
    double* const tile = reinterpret_cast<double*>(coef + (size_t)(chunked ? a.mc : M) * (PAD ? NP * NP : nn));  // [PC][TS]
    const int slot = threadIdx.x / NP, r = threadIdx.x % NP, lane = threadIdx.x & 63;
    int fm = a.first % a.npt;
    if (fm < 0) fm += a.npt;
    for (int64_t line = blockIdx.x; line < a.nlines; line += gridDim.x) {
        if (!chunked) {
            __syncthreads();
            if (PAD && a.fold)
                panel_stage_fold<NP>(coef, a.src + line * ((int64_t)M * nn), n, M, 0.0, 0.0);
            else
                panel_stage<NP, PAD>(coef, a.src + line * ((int64_t)M * nn), n, M);
            __syncthreads();
        }
        // nodes of this coefficient set: the whole grid line, or its run of an irregular (symmetric-rule) list
        const int64_t kbase = a.run_start ? a.run_start[line] : line * a.npt;
        const int count = a.run_start ? (int)(a.run_start[line + 1] - kbase) : a.npt;
        for (int i0 = 0; i0 < count; i0 += SLOTS) {
            // a wave without a node in this pass computes nothing but keeps the block's barriers (the stores go through LDS)
            const bool wave_on = i0 + (int)(threadIdx.x >> 6) * (64 / NP) < count;
            const int i1 = i0 + slot;
            const bool act = i1 < count;
            const int ii = act ? i1 : 0;
            const int64_t k = kbase + ii;
            const bool wr = act && r < n;
            double hr[NP], hi[NP];
            if (chunked) {  // the set in chunks of a.mc coefficients, staged per group of nodes: every wave keeps the barriers
                const int ic = a.gi ? a.gi[kbase + ii] : ii;
                const double2 z = a.tab[ic];
                const double2 w = a.tab[(int)(((unsigned)fm * (unsigned)ic) % (unsigned)a.npt)];
                double pr = w.x, pi = w.y;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    hr[j] = 0.0;
                    hi[j] = 0.0;
                }
                for (int m0 = 0; m0 < M; m0 += a.mc) {
                    const int mcur = min(a.mc, M - m0);
                    __syncthreads();
                    panel_stage<NP, false>(coef, a.src + line * ((int64_t)M * nn) + (int64_t)m0 * nn, n, mcur);
                    __syncthreads();
                    if (wave_on) panel_series_row_chunk<NP>(coef, n, mcur, z.x, z.y, pr, pi, r, hr, hi);
                }
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const bool real = r < n && j < n;
                    hr[j] = real ? -hr[j] : 0.0;
                    hi[j] = real ? -hi[j] : 0.0;
                }
            } else if (wave_on) {
                const int ic = a.gi ? a.gi[kbase + ii] : ii;
                const double2 z = a.tab[ic];
                const double2 w = a.tab[(int)(((unsigned)fm * (unsigned)ic) % (unsigned)a.npt)];
                if (PAD && a.fold)
                    panel_series_row_fold<NP>(coef, M, z.x, z.y, r, hr, hi);  // row r of -H(k) (its padding diagonal: 1)
                else
                    panel_series_row<NP, PAD>(coef, n, M, z.x, z.y, w.x, w.y, r, hr, hi);  // row r of -H(k)
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const bool real = (PAD && !a.fold) || (r < n && j < n);
                    hr[j] = real ? -hr[j] : 0.0;
                    hi[j] = real ? -hi[j] : 0.0;
                }
            }
            if (a.H.base && a.herm) {
                // Hermitian H(k): the upper triangle (compact plane order: Re / Im of H[a][b], a < b, at b^2 + 2a, + 1; H[b][b]
                // at b^2 + 2b) goes through an LDS tile [plane][node] so that every plane row leaves the block as runs of SLOTS
                // consecutive nodes (whole 128-B lines; lane-by-lane each store instruction wrote sixteen 32-B pieces).
                // Full layout: the lower triangle is the conjugate of the same numbers, the diagonal's imaginary plane 0.
                // Two tile passes by COLUMNS (16 rows: columns 0...11 = planes 0...143, then 12...15 = planes 144...255; 8 rows:
                // one pass), so that plane indices are compile-time offsets from one lane-dependent address.
                auto pass = [&](auto j0c, auto j1c) {
                    constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
                    if (n <= J0) return;  // uniform
                    if (wave_on && wr) {
                        double* const tp = tile + (2 * r) * TS + slot;
#pragma unroll
                        for (int j = J0; j < J1; ++j) {
                            if (j < n && r <= j) {
                                tp[(j * j - J0 * J0) * TS] = hr[j];
                                if (r < j) tp[(j * j - J0 * J0 + 1) * TS] = hi[j];
                            }
                        }
                    }
                    __syncthreads();
                    const int jend = n < J1 ? n : J1;
                    const int npl = jend * jend - J0 * J0;
                    for (int idx = threadIdx.x; idx < npl * SLOTS; idx += 256) {
                        const int pl = idx / SLOTS, sl = idx - pl * SLOTS;
                        if (i0 + sl >= count) continue;
                        const double val = tile[pl * TS + sl];
                        const int pc = J0 * J0 + pl;
                        double* ho = a.H.base + view_off(a.H, kbase + i0 + sl);
                        if (a.H.compact) {
                            ho[(int64_t)pc * a.H.pitch] = val;
                        } else {
                            int bb = (int)sqrtf((float)pc);
                            bb -= (bb * bb > pc) ? 1 : 0;
                            bb += ((bb + 1) * (bb + 1) <= pc) ? 1 : 0;
                            const int rem = pc - bb * bb, aa = rem >> 1, im = rem & 1;
                            ho[(int64_t)(2 * (aa + n * bb) + im) * a.H.pitch] = val;
                            if (aa != bb)
                                ho[(int64_t)(2 * (bb + n * aa) + im) * a.H.pitch] = im ? -val : val;
                            else
                                ho[(int64_t)(2 * (aa + n * aa) + 1) * a.H.pitch] = 0.0;
                        }
                    }
                    __syncthreads();
                };
                if constexpr (NP == 32) {  // 17...32 bands: seven passes of <= 171 planes
                    pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 12>{});
                    pass(std::integral_constant<int, 12>{}, std::integral_constant<int, 17>{});
                    pass(std::integral_constant<int, 17>{}, std::integral_constant<int, 21>{});
                    pass(std::integral_constant<int, 21>{}, std::integral_constant<int, 24>{});
                    pass(std::integral_constant<int, 24>{}, std::integral_constant<int, 27>{});
                    pass(std::integral_constant<int, 27>{}, std::integral_constant<int, 30>{});
                    pass(std::integral_constant<int, 30>{}, std::integral_constant<int, 32>{});
                } else if constexpr (NP == 16) {
                    pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 12>{});
                    pass(std::integral_constant<int, 12>{}, std::integral_constant<int, 16>{});
                } else {
                    pass(std::integral_constant<int, 0>{}, std::integral_constant<int, NP>{});
                }
            } else if (a.H.base && wave_on && wr) {  // a series that is not Hermitian: the rows as they are
                double* ho = a.H.base + view_off(a.H, k);
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    if (j < n) {
                        ho[(int64_t)(2 * (r + n * j)) * a.H.pitch] = hr[j];
                        ho[(int64_t)(2 * (r + n * j) + 1) * a.H.pitch] = hi[j];
                    }
                }
            }
            if (!a.E.base) continue;  // values only (uniform)
            double vr[NP], vi[NP], dg = 0.0;
            int rank = r;
            if (wave_on) {
            if constexpr (TRI && VEC) {  // eigenvalues as below, eigenvectors by inverse iteration on H itself
                double tr_[NP], ti_[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    tr_[j] = hr[j];
                    ti_[j] = hi[j];
                }
                dg = rows_eigvals_tridiag<NP>(n, r, tr_, ti_);
                rank = r;
                rows_eigvecs_invit<NP>(n, r, lane, hr, hi, dg, vr, vi);
            } else if constexpr (TRI) {  // eigenvalues only
                if constexpr (SPLIT) {
                    // Householder here, the eigenvalues of the tridiagonals in tri_eig_kernel (one LANE per matrix there: a
                    // fraction of the instructions of the bisection below, which keeps 16 lanes busy per matrix)
                    double e2[NP], d[NP];
                    hh_steps<NP>(n, r, hr, hi, e2, std::make_integer_sequence<int, NP>());
                    diag_gather<NP>(hr, d, std::make_integer_sequence<int, NP>());
                    double dr = 0.0, er = 0.0;
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        dr = (j == r) ? d[j] : dr;
                        er = (j == r) ? e2[j] : er;
                    }
                    if (wr) {
                        a.tri[(int64_t)r * a.tri_nk + k] = dr;
                        a.tri[(int64_t)(NP + r) * a.tri_nk + k] = er;
                    }
                } else {
                    dg = rows_eigvals_tridiag<NP>(n, r, hr, hi);  // Householder + Sturm bisection, lane r gets eigenvalue r
                    rank = r;
                }
            } else {
                rows_eig<NP, VEC>(n, r, lane, hr, hi, vr, vi, dg, rank);
            }
            }
            if constexpr (SPLIT) continue;  // (the whole block: no barrier below is skipped by a part of it)
            {  // eigenvalue planes: through the tile as well ([band][node] -> SLOTS consecutive nodes per band)
                if (wave_on && wr) tile[rank * TS + slot] = dg;
                __syncthreads();
                for (int idx = threadIdx.x; idx < n * SLOTS; idx += 256) {
                    const int pl = idx / SLOTS, sl = idx - pl * SLOTS;
                    if (i0 + sl < count) a.E.base[view_off(a.E, kbase + i0 + sl) + (int64_t)pl * a.E.pitch] = tile[pl * TS + sl];
                }
                __syncthreads();
            }
            if (!wave_on) continue;
            if constexpr (VEC) {
                double ranks[NP];
                if constexpr (TRI) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) ranks[j] = (double)j;  // inverse iteration: column j is band j
                } else {
                    group_gather<NP>((double)rank, ranks, std::make_integer_sequence<int, NP>());
                }
                if (wr) {
                    double* uo = a.U.base + view_off(a.U, k);
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        if (j < n) {
                            const int band = (int)ranks[j];
                            uo[(int64_t)(2 * (r + n * band)) * a.U.pitch] = vr[j];
                            uo[(int64_t)(2 * (r + n * band) + 1) * a.U.pitch] = vi[j];
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Eigenvalues of the real symmetric tridiagonals (d, |e|^2) that gen_grid_eig_kernel leaves behind: ONE LANE PER MATRIX,
// the root-free QR iteration of Pal, Walker and Kahan as LAPACK's dsterf runs it (its "QR iteration" branch: Wilkinson
// shift from the bottom 2 x 2, one sweep of rational rotations on d and e^2 -- no square roots in the sweep --, deflation
// at the bottom), on the whole leading block [0, L] (interior splits are not looked for: a rotation across a zero
// coupling is the identity).  ~1.7 sweeps per eigenvalue, ~60 instructions per rotation: ~30 k instructions per 64
// matrices against ~4 k per FOUR for the bisection in the row layout.  The arrays of a lane live in LDS ([j][lane]: the
// bottom index L differs from lane to lane), sorted at the end by an odd-even transposition network in registers.
// ------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(64) void tri_eig_kernel(const double* __restrict__ tri, int64_t tri_nk, int64_t nk, int n, PlaneView E) {
    __shared__ double ld[NP + 2][64], le[NP + 2][64];
    const int lane = threadIdx.x;
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;
    const bool act = k < nk;
    const int64_t kk = act ? k : nk - 1;
    double anorm2 = 0.0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const double dj = j < n ? tri[(int64_t)j * tri_nk + kk] : 0.0;
        const double ej = j + 1 < n ? tri[(int64_t)(NP + j) * tri_nk + kk] : 0.0;
        ld[j][lane] = dj;
        le[j][lane] = ej;
        anorm2 = fmax(anorm2, fmax(dj * dj, ej));
    }
    int L = tri_qr_lane(ld, le, n, lane, anorm2);
    // A block that is still coupled when the budget runs out (LAPACK's dsterf returns info > 0 there) must not leave as
    // plausible numbers: the node's eigenvalues become NaN, which every sum over the rule carries to the caller.
    const bool failed = L > 0;
    double v[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) v[j] = j < n ? (failed ? __builtin_nan("") : ld[j][lane]) : __builtin_huge_val();
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
#pragma unroll
        for (int j = pass & 1; j + 1 < NP; j += 2) {
            const double lo = fmin(v[j], v[j + 1]), hi = fmax(v[j], v[j + 1]);
            v[j] = lo;
            v[j + 1] = hi;
        }
    }
    if (act) {
        double* eo = E.base + view_off(E, k);
#pragma unroll
        for (int j = 0; j < NP; ++j)
            if (j < n) eo[(int64_t)j * E.pitch] = v[j];
    }
}

// full-grid rule values (and eigenvalues, Hermitian series) for 5..16 bands
static bool gen_grid_eig_supported(const GenSpec& gs, int* np_out, size_t* lds_out, bool* pad_out) {
    const bool runs = !gs.grid && gs.run_start && gs.gi && !gs.x && gs.nruns > 0;  // symmetric rules: runs of grid-index nodes
    if (!(gs.grid || runs) || gs.deriv || gs.values || gs.Haos || gs.Eaos || !(gs.Eplanes.base || gs.Hplanes.base)) return false;
    if (gs.Eplanes.base && !gs.herm) return false;  // the Jacobi works on full rows: H(k) must be Hermitian to rounding
    // 9..16 bands with eigenvectors: inverse iteration (needs the zero-padded layout, checked below)
    if (gs.n <= 4 || gs.n > 32 || gs.npt < 1 || gs.npt >= 65536) return false;
    // 17...32 bands (two nodes per wave): H and eigenvalues -- through the tridiagonal kernel -- only
    if (gs.n > 16 && (gs.Uplanes.base || (gs.Eplanes.base && abz_switch(SW_EIG_SPLIT) == 0))) return false;
    const int np = gs.n <= 8 ? 8 : (gs.n <= 16 ? 16 : 32);
    const size_t tile_bytes = sizeof(double) * (size_t)(np == 32 ? 176 : (np == 16 ? 144 : 64)) * (size_t)(256 / np + 1);  // the store tile [PC][TS]
    size_t lds = sizeof(double2) * (size_t)gs.M * np * np + tile_bytes;
    *pad_out = lds <= 150 * 1024;
    if (!*pad_out) lds = sizeof(double2) * (size_t)gs.M * gs.n * gs.n + tile_bytes;
    if (lds > 160 * 1024) {  // (unpadded: up to the whole LDS of a CU; beyond it the set is staged in chunks, two workgroups per CU)
        if (gs.Uplanes.base) return false;
        const int mc = (int)((72 * 1024 - tile_bytes) / (sizeof(double2) * (size_t)gs.n * gs.n));
        if (mc < 1) return false;
        lds = sizeof(double2) * (size_t)mc * gs.n * gs.n + tile_bytes;
    }
    if (gs.Uplanes.base && gs.n > 8 && !*pad_out) return false;
    *np_out = np;
    *lds_out = lds;
    return true;
}

bool gen_compact_supported(int n, int M, int npt) {
    if (n <= 4 || n > 16 || npt < 1 || npt >= 65536) return false;
    const int np = n <= 8 ? 8 : 16;
    const size_t tile_bytes = sizeof(double) * (size_t)(np == 16 ? 144 : 64) * (size_t)(256 / np + 1);
    return sizeof(double2) * (size_t)M * n * n + tile_bytes <= 150 * 1024;  // gen_grid_eig_supported's bound (unpadded set)
}

static int launch_gen_grid_eig(abz_ctx* ctx, const GenSpec& gs, int np, size_t lds, bool pad) {
    GenEigArgs a;
    a.src = gs.src;
    a.tab = gs.tab;
    a.H = gs.Hplanes;
    a.E = gs.Eplanes;
    a.U = gs.Uplanes;
    a.nlines = gs.grid ? gs.nnodes / gs.npt : gs.nruns;
    a.run_start = gs.grid ? nullptr : gs.run_start;
    a.gi = gs.grid ? nullptr : gs.gi;
    a.n = gs.n;
    a.M = gs.M;
    a.first = gs.first;
    a.npt = gs.npt;
    a.herm = gs.herm ? 1 : 0;
    a.fold = (gs.herm && pad && (gs.M & 1) && gs.first == -((gs.M - 1) / 2) && abz_switch(SW_EIG_FOLD) != 0) ? 1 : 0;
    {
        const size_t tile_bytes = sizeof(double) * (size_t)(np == 32 ? 176 : (np == 16 ? 144 : 64)) * (size_t)(256 / np + 1);
        if (!pad && sizeof(double2) * (size_t)gs.M * gs.n * gs.n + tile_bytes > 160 * 1024)  // gen_grid_eig_supported's chunk length
            a.mc = (int)((72 * 1024 - tile_bytes) / (sizeof(double2) * (size_t)gs.n * gs.n));
    }
    // one workgroup per line (or run): the set is staged per line whatever the grid, and the hardware balances a grid that is not a
    // multiple of the resident workgroups (48^3 x 16 bands: 2 304 lines over 1 024 workgroups left a quarter of the time to a tail)
    const int64_t blocks = std::min<int64_t>(a.nlines, abz_switch(SW_EIG_SPLIT) != 0 ? (int64_t)1 << 20 : 256 * 4);
    const bool vec = gs.Uplanes.base != nullptr;
    // eigenvalues without eigenvectors: the tridiagonals go through scratch to tri_eig_kernel (ABZ_EIG_SPLIT=0: bisection
    // inside the grid kernel).  (Cutting the grid into four chunks of lines with chunk c's tridiagonal kernel on a second
    // stream beside chunk c + 1's grid kernel was slower, 0.76 against 0.51 ms at 48^3: four short launches with their own
    // tails.)
    const bool split = !vec && gs.Eplanes.base && abz_switch(SW_EIG_SPLIT) != 0;
    if (split) {
        a.tri_nk = (gs.nnodes + 63) / 64 * 64;
        int rc = ctx->scratch[4].reserve(sizeof(double) * (size_t)(2 * np) * (size_t)a.tri_nk);
        if (rc) return rc;
        a.tri = ctx->scratch[4].as<double>();
    }
    ProfScope ps(ctx, ABZ_K_EVAL);
    // eigenvalues only: Householder + Sturm bisection (TRI); with eigenvectors: the parallel-order Jacobi up to 8 bands,
    // inverse iteration on the zero-padded layout for 9...16 (the 16-row Jacobi with accumulated rotations spilled 4.6 KB)
#define ABZ_GE4(NPV, PV, VV, TV, SV)                                                                                              \
    {                                                                                                                             \
        ABZ_HIP(hipFuncSetAttribute((const void*)gen_grid_eig_kernel<NPV, PV, VV, TV, SV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds));                                                                                   \
        hipLaunchKernelGGL((gen_grid_eig_kernel<NPV, PV, VV, TV, SV>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);   \
    }
#define ABZ_GE3(NPV, PV, VV, TV) ABZ_GE4(NPV, PV, VV, TV, false)
    if (!vec && (split || !gs.Eplanes.base)) {  // (H only: the lean instance as well -- it never reaches the eigenvalue stage)
        if (np == 8 && pad) ABZ_GE4(8, true, false, true, true)
        else if (np == 8) ABZ_GE4(8, false, false, true, true)
        else if (np == 32 && pad) ABZ_GE4(32, true, false, true, true)
        else if (np == 32) ABZ_GE4(32, false, false, true, true)
        else if (pad) ABZ_GE4(16, true, false, true, true)
        else ABZ_GE4(16, false, false, true, true)
    } else if (!vec) {
        if (np == 8 && pad) ABZ_GE3(8, true, false, true)
        else if (np == 8) ABZ_GE3(8, false, false, true)
        else if (pad) ABZ_GE3(16, true, false, true)
        else ABZ_GE3(16, false, false, true)
    } else if (np == 8) {
        if (pad) ABZ_GE3(8, true, true, false)
        else ABZ_GE3(8, false, true, false)
    } else {
        ABZ_GE3(16, true, true, true)  // gen_grid_eig_supported: pad is set
    }
#undef ABZ_GE3
#undef ABZ_GE4
    if (split) {
        const unsigned tb = (unsigned)cdiv2(gs.nnodes, 64);
        if (np == 8)
            hipLaunchKernelGGL(tri_eig_kernel<8>, dim3(tb), dim3(64), 0, ctx->stream, a.tri, a.tri_nk, gs.nnodes, gs.n, gs.Eplanes);
        else if (np == 32)
            hipLaunchKernelGGL(tri_eig_kernel<32>, dim3(tb), dim3(64), 0, ctx->stream, a.tri, a.tri_nk, gs.nnodes, gs.n, gs.Eplanes);
        else
            hipLaunchKernelGGL(tri_eig_kernel<16>, dim3(tb), dim3(64), 0, ctx->stream, a.tri, a.tri_nk, gs.nnodes, gs.n, gs.Eplanes);
    }
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

static bool gen_panel_supported(const GenSpec& gs, int* np_out, size_t* lds_out, bool* pad_out) {
    if (!gs.values || !gs.panels15 || gs.grid || !gs.x || gs.deriv || gs.nnodes % 15 != 0) return false;
    if (gs.n_sweep > 1 || gs.sweep_dev) return false;
    if (!(gs.integrand == ABZ_F_DOS || gs.integrand == ABZ_F_TRGLOC || gs.integrand == ABZ_F_GLOC)) return false;
    if (gs.Hplanes.base || gs.Eplanes.base || gs.Uplanes.base || gs.Haos || gs.Eaos) return false;
    if (gs.n > 16 && gs.integrand != ABZ_F_GLOC && !gs.herm) return false;  // (the 32-lane instance takes traces from the tridiagonal of Hermitian(h))
    const int np = gs.n <= 8 ? 8 : (gs.n <= 16 ? 16 : 32);
    size_t lds = sizeof(double2) * (size_t)gs.M * np * np;  // zero-padded set
    *pad_out = lds <= 150 * 1024;
    if (!*pad_out) lds = sizeof(double2) * (size_t)gs.M * gs.n * gs.n;
    if (lds > 150 * 1024) return false;
    *np_out = np;
    *lds_out = lds;
    return true;
}

static int gen_waves_per_block(int n, int M) {
    const size_t per = sizeof(double2) * (size_t)(3 * n * n + M + (n + 1) / 2);
    int w = (int)((150 * 1024) / per);
    if (w > 4) w = 4;
    if (w < 1) w = 1;
    return w;
}

int launch_gen_nodes(abz_ctx* ctx, const GenSpec& gs) {
    if (gs.nnodes == 0) return ABZ_OK;
    if (big_supported(gs.n)) return launch_big_nodes(ctx, gs);  // 33...64 bands: kernels_big.hip
    // arbitrary nodes (abz_eval_nodes) with eigenvalues, 9...32 bands: Householder + QR through the kernels of kernels_big.hip (generic in
    // n) instead of the wave-per-node Jacobi below (4 096 nodes of 32 bands: 18 ms of kernels; a rule build of 13 824 nodes 0.8 ms)
    if (!gs.grid && gs.x && !gs.values && (gs.Eplanes.base || gs.Eaos) && !gs.Uplanes.base && !gs.deriv && !gs.Hplanes.compact && gs.n > 4)
        return launch_big_nodes(ctx, gs);
    if (gs.values && gs.n > 16 && big_inverse_wanted(gs.n, gs.integrand, gs.herm) && !gs.Hplanes.base && !gs.Haos && !gs.Eplanes.base && !gs.Eaos &&
        !gs.Uplanes.base && !gs.deriv)
        return launch_big_nodes(ctx, gs);  // 17...32 bands, values only (IAI node path): the inverse of every node in registers
    if (lane_grid_supported(gs)) return launch_lane_grid(ctx, gs);  // 5...8 bands on full grids: one node per lane
    if (gs.n > ABZ_MAX_BANDS) {
        set_error("n = %d bands exceeds ABZ_MAX_BANDS", gs.n);
        return ABZ_ERR_UNSUPPORTED;
    }
    GenArgs a;
    a.src = gs.src;
    a.parents = gs.parents;
    a.x = gs.x;
    a.gi = gs.gi;
    a.tab = gs.tab;
    a.tail = nullptr;
    a.nnodes = gs.nnodes;
    a.n = gs.n;
    a.M = gs.M;
    a.first = gs.first;
    a.npt = gs.npt;
    a.d = gs.d;
    a.grid = gs.grid ? 1 : 0;
    a.deriv = gs.deriv ? 1 : 0;
    a.inv_period = 1.0 / gs.period;
    a.Hplanes = gs.Hplanes;
    a.Eplanes = gs.Eplanes;
    a.Uplanes = gs.Uplanes;
    a.Haos = gs.Haos;
    a.Eaos = gs.Eaos;
    a.integrand = gs.integrand;
    a.n_sweep = gs.values ? (gs.n_sweep > 0 ? gs.n_sweep : 1) : 0;
    a.ncomp = gs.values ? integrand_ncomp(gs.integrand, gs.n, gs.d) : 0;
    for (int i = 0; i < 4; ++i) a.p[i] = gs.params[i];
    a.sweep = gs.sweep_dev;
    a.sweep_per_node = gs.sweep_per_node;
    a.sweep0 = gs.sweep0;
    a.values = gs.values;
    if (gs.values && (gs.integrand == ABZ_F_LINEAR || gs.integrand == ABZ_F_LINEAR_X)) {
        set_error("ABZ_F_LINEAR(_X) needs a scalar (n = 1) series");
        return ABZ_ERR_ARG;
    }
    {
        int np = 0;
        size_t plds = 0;
        bool pad = false;
        if (gen_grid_eig_supported(gs, &np, &plds, &pad)) return launch_gen_grid_eig(ctx, gs, np, plds, pad);
    }
    {
        int np = 0;
        size_t plds = 0;
        bool pad = false;
        if (gen_panel_supported(gs, &np, &plds, &pad)) {
            const int64_t blocks = std::min<int64_t>(gs.nnodes / 15, 256 * 8);
            ProfScope ps(ctx, ABZ_K_EVAL);
#define ABZ_PANEL2(NPV, PV)                                                                                                  \
    ABZ_HIP(hipFuncSetAttribute((const void*)gen_panel_kernel<NPV, PV>, hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                (int)plds));                                                                               \
    hipLaunchKernelGGL((gen_panel_kernel<NPV, PV>), dim3((unsigned)blocks), dim3(256), plds, ctx->stream, a);
#define ABZ_PANEL(NPV) \
    if (pad) {         \
        ABZ_PANEL2(NPV, true) \
    } else {           \
        ABZ_PANEL2(NPV, false) \
    }
            if (np == 8) {
                ABZ_PANEL(8)
            } else if (np == 16) {
                ABZ_PANEL(16)
            } else {
                ABZ_PANEL(32)
            }
#undef ABZ_PANEL
#undef ABZ_PANEL2
            ABZ_HIP(hipGetLastError());
            return ABZ_OK;
        }
    }
    if (gs.Hplanes.base && gs.Hplanes.compact) {
        set_error("internal: an upper-triangle rule reached the wave-per-node kernel (full layout only)");
        return ABZ_ERR_UNSUPPORTED;
    }
    const int wpb = gen_waves_per_block(gs.n, gs.M);
    const size_t lds = sizeof(double2) * (size_t)(3 * gs.n * gs.n + gs.M + (gs.n + 1) / 2) * wpb;
    const int64_t blocks = std::min<int64_t>(cdiv2(gs.nnodes, wpb), 256 * 16);
    ProfScope ps(ctx, ABZ_K_EVAL);
    ABZ_HIP(hipFuncSetAttribute((const void*)gen_node_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(gen_node_kernel, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a, wpb);
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// reduce over a cached rule, n > 4: one wave per node chunk, per-omega partial sums in LDS
// ------------------------------------------------------------------------------------------
// element (r, j) of the rule's H(k) at a node whose plane 0 is `hin`: full layout 2 (r + n j) + {re, im}; upper-triangle
// layout (PlaneView::compact): Re / Im of H[a][b], a < b, at b^2 + 2a, + 1, H[b][b] at b^2 + 2b, lower = conjugate
__device__ __forceinline__ void rule_h_elem(const PlaneView& v, const double* __restrict__ hin, int n, int r, int j, double& vr, double& vi) {
    if (v.compact) {
        const int lo = r < j ? r : j, hi = r < j ? j : r;
        vr = hin[(int64_t)(hi * hi + 2 * lo) * v.pitch];
        const double t = (r == j) ? 0.0 : hin[(int64_t)(hi * hi + 2 * lo + 1) * v.pitch];
        vi = r > j ? -t : t;
    } else {
        vr = hin[(int64_t)(2 * (r + n * j)) * v.pitch];
        vi = hin[(int64_t)(2 * (r + n * j) + 1) * v.pitch];
    }
}

struct GenReduceArgs {
    PlaneView Hplanes;
    PlaneView Eplanes;
    const double* w;
    const double* sweep;
    int64_t nk, chunk;
    int n, n_sweep, ncomp, integrand;
    double p[4];
};

__global__ __launch_bounds__(256) void gen_reduce_kernel(GenReduceArgs a, int waves_per_block, double2* partial) {
    extern __shared__ double2 lds_r[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= waves_per_block) return;
    const int n = a.n, nn = n * n;
    const int ncols = a.n_sweep * a.ncomp;
    double2* H = lds_r + (size_t)wave * (3 * nn + ncols + (n + 1) / 2);
    double2* W = H + nn;
    double2* X = W + nn;
    double2* acc = X + nn;
    double* ev = reinterpret_cast<double*>(acc + ncols);
    for (int t = lane; t < ncols; t += 64) acc[t] = make_double2(0.0, 0.0);
    const int64_t gw = (int64_t)blockIdx.x * waves_per_block + wave;
    const int64_t k0 = gw * a.chunk, k1 = min(a.nk, k0 + a.chunk);
    GenArgs ga;  // only the fields gen_integrand reads
    ga.n = n;
    ga.integrand = a.integrand;
    for (int i = 0; i < 4; ++i) ga.p[i] = a.p[i];
    for (int64_t k = k0; k < k1; ++k) {
        const double wk = a.w ? a.w[k] : 1.0;
        if (a.integrand == ABZ_F_DOS_EIG) {
            const double* ei = a.Eplanes.base + view_off(a.Eplanes, k);
            for (int b = lane; b < n; b += 64) ev[b] = ei[(int64_t)b * a.Eplanes.pitch];
        } else if (a.integrand != ABZ_F_ONE) {
            const double* hi_ = a.Hplanes.base + view_off(a.Hplanes, k);
            for (int t = lane; t < nn; t += 64) {
                double vr, vi;
                rule_h_elem(a.Hplanes, hi_, n, t % n, t / n, vr, vi);  // H[t]: row t % n, column t / n
                H[t] = make_double2(vr, vi);
            }
        }
        wave_sync();
        for (int s = 0; s < a.n_sweep; ++s) {
            const double sw = a.sweep ? a.sweep[s] : 0.0;
            // value into X-adjacent scratch: reuse global-free path by writing to LDS acc directly
            if (a.integrand == ABZ_F_ONE) {
                if (lane == 0) acc[s * a.ncomp].x += wk;
            } else if (a.integrand == ABZ_F_DOS_EIG) {
                double v = 0.0;
                for (int b = lane; b < n; b += 64) {
                    const double de = sw - ev[b];
                    v += a.p[0] / (de * de + a.p[0] * a.p[0]);
                }
                v = wsum(v);
                if (lane == 0) acc[s].x += wk * v * 0.31830988618379067153776752674503;
            } else {
                for (int t = lane; t < nn; t += 64) {
                    const int r = t % n, c = t / n;
                    W[t] = make_double2((r == c ? sw : 0.0) - H[t].x, (r == c ? a.p[0] : 0.0) - H[t].y);
                }
                wave_sync();
                wave_inverse(W, X, n, lane);
                if (a.integrand == ABZ_F_GLOC) {
                    for (int t = lane; t < nn; t += 64) {
                        acc[s * nn + t].x += wk * X[t].x;
                        acc[s * nn + t].y += wk * X[t].y;
                    }
                } else {
                    double tr = 0.0, ti = 0.0;
                    for (int t = lane; t < n; t += 64) {
                        tr += X[t + n * t].x;
                        ti += X[t + n * t].y;
                    }
                    tr = wsum(tr);
                    ti = wsum(ti);
                    if (lane == 0) {
                        if (a.integrand == ABZ_F_DOS)
                            acc[s].x += wk * (-ti * 0.31830988618379067153776752674503);
                        else {
                            acc[s].x += wk * tr;
                            acc[s].y += wk * ti;
                        }
                    }
                }
            }
            wave_sync();
        }
    }
    wave_sync();
    for (int t = lane; t < ncols; t += 64) partial[gw * ncols + t] = acc[t];
}

__global__ __launch_bounds__(256) void final_reduce2_kernel(const double2* __restrict__ partial, int64_t nrows,
                                                            int64_t ncols, double scale, double2* __restrict__ out) {
    __shared__ double2 sh[4];
    const int64_t col = blockIdx.x;
    double sr = 0.0, si = 0.0;
    for (int64_t b = threadIdx.x; b < nrows; b += 256) {
        const double2 v = partial[b * ncols + col];
        sr += v.x;
        si += v.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sr += __shfl_down(sr, off, 64);
        si += __shfl_down(si, off, 64);
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = make_double2(sr, si);
    __syncthreads();
    if (threadIdx.x == 0)
        out[col] = make_double2((sh[0].x + sh[1].x + sh[2].x + sh[3].x) * scale, (sh[0].y + sh[1].y + sh[2].y + sh[3].y) * scale);
}

// Resolvent-trace scans of a cached rule for 5..16 bands (Hermitian values) on the row kernels: lane r of a node's
// NP lanes loads row r of H(k) from the rule's planes, then per swept value shift, swizzle-pivot inversion, trace.
// Four swept values per pass over the nodes (the rule is re-read per pass: 16 n^2 B per node against 4 inversions).
struct GenRowsReduceArgs {
    PlaneView H;
    const double* w;
    const double* sweep;  // device [n_sweep]
    double2* partial;     // [blocks][n_sweep]
    int64_t nk;
    int n, n_sweep, is_dos;
    double eta;
};

template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void gen_rows_reduce_kernel(GenRowsReduceArgs a) {
    constexpr int SLOTS = 256 / NP;
    __shared__ double2 red[SLOTS * 4];
    const int n = a.n;
    const int slot = threadIdx.x / NP, r = threadIdx.x % NP;
    for (int s0 = 0; s0 < a.n_sweep; s0 += 4) {
        const int nw = min(4, a.n_sweep - s0);
        double accr[4] = {0.0, 0.0, 0.0, 0.0}, acci[4] = {0.0, 0.0, 0.0, 0.0};
        for (int64_t k0 = (int64_t)blockIdx.x * SLOTS; k0 < a.nk; k0 += (int64_t)gridDim.x * SLOTS) {
            const int64_t k = k0 + slot;
            const bool act = k < a.nk;
            const int64_t kc = act ? k : 0;
            const double wk = act ? (a.w ? a.w[kc] : 1.0) : 0.0;
            const double* __restrict__ hin = a.H.base + view_off(a.H, kc);
            const int rr = r < n ? r : 0;
            double hr[NP], hi[NP];  // row r of -H; rows / columns >= n: zero
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const bool real = r < n && j < n;
                const int jj = j < n ? j : 0;
                double vr, vi;
                rule_h_elem(a.H, hin, n, rr, jj, vr, vi);
                hr[j] = real ? -vr : 0.0;
                hi[j] = real ? -vi : 0.0;
            }
#pragma unroll 1
            for (int q = 0; q < nw; ++q) {
                double ar[NP], ai[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    ar[j] = hr[j];
                    ai[j] = hi[j];
                }
                panel_shift_row<NP, false>(n, a.sweep[s0 + q], a.eta, r, ar, ai);
                double tr, ti;
                if constexpr (NP == 16) {
                    panel_inverse_trace_fmac<true>(n, r, ar, ai, tr, ti);
                } else {
                    panel_invert_rows<NP, false>(n, r, ar, ai);
                    panel_trace<NP>(ar, ai, n, r, tr, ti);
                }
                const double dr = wk * (a.is_dos ? -ti * 0.31830988618379067153776752674503 : tr);
                const double di = a.is_dos ? 0.0 : wk * ti;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    accr[qq] += (qq == q) ? dr : 0.0;
                    acci[qq] += (qq == q) ? di : 0.0;
                }
            }
        }
        __syncthreads();
        if (r == 0)
            for (int q = 0; q < 4; ++q) red[slot * 4 + q] = make_double2(accr[q], acci[q]);
        __syncthreads();
        if ((int)threadIdx.x < nw) {
            double sr = 0.0, si = 0.0;
            for (int sl = 0; sl < SLOTS; ++sl) {
                sr += red[sl * 4 + threadIdx.x].x;
                si += red[sl * 4 + threadIdx.x].y;
            }
            a.partial[(int64_t)blockIdx.x * a.n_sweep + s0 + threadIdx.x] = make_double2(sr, si);
        }
    }
}

// The same for scans of a cached rule (5..16 bands, resolvent traces): lane r loads row r of H(k) from the rule's planes,
// the node is tridiagonalised once per chunk of 4 NP swept values, lane r takes values r, r + NP, ... of the chunk.
// 256 omega at 16 bands: 4 passes over the rule and ~(4 x 2.5 k + 256 / 16 x 250) instructions per four nodes instead
// of 64 passes and 256 x 1.3 k.
template <int NP>
__global__ __launch_bounds__(256, NP <= 16 ? 2 : 1) void gen_rows_reduce_tri_kernel(GenRowsReduceArgs a) {
    constexpr int SLOTS = 256 / NP;
    __shared__ double2 red[SLOTS * 4 * NP];
    const int n = a.n;
    const int slot = threadIdx.x / NP, r = threadIdx.x % NP;
    for (int s0 = 0; s0 < a.n_sweep; s0 += 4 * NP) {
        const int left = a.n_sweep - s0;
        const int npass = min(4, (left + NP - 1) / NP);
        double sw[4], accr[4] = {0.0, 0.0, 0.0, 0.0}, acci[4] = {0.0, 0.0, 0.0, 0.0};
        bool mine[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            mine[p] = p * NP + r < left;
            sw[p] = mine[p] ? a.sweep[s0 + p * NP + r] : 0.0;
        }
        for (int64_t k0 = (int64_t)blockIdx.x * SLOTS; k0 < a.nk; k0 += (int64_t)gridDim.x * SLOTS) {
            if (k0 + (int64_t)(threadIdx.x >> 6) * (64 / NP) >= a.nk) continue;  // no node for this wave (wave-level sync only below)
            const int64_t k = k0 + slot;
            const bool act = k < a.nk;
            const int64_t kc = act ? k : 0;
            const double wk = act ? (a.w ? a.w[kc] : 1.0) : 0.0;
            const double* __restrict__ hin = a.H.base + view_off(a.H, kc);
            const int rr = r < n ? r : 0;
            double hr[NP], hi[NP];  // row r of B = -H; rows / columns >= n: zero
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const bool real = r < n && j < n;
                const int jj = j < n ? j : 0;
                double vr, vi;
                rule_h_elem(a.H, hin, n, rr, jj, vr, vi);
                hr[j] = real ? -vr : 0.0;
                hi[j] = real ? -vi : 0.0;
            }
            double e2[NP], b[NP];
            hh_steps<NP>(n, r, hr, hi, e2, std::make_integer_sequence<int, NP>());
            diag_gather<NP>(hr, b, std::make_integer_sequence<int, NP>());
            double rad = 0.0, eprev = 0.0;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (i < n) {
                    const double en = (i + 1 < n) ? sqrt(e2[i]) : 0.0;
                    rad = fmax(rad, fabs(b[i]) + eprev + en);
                    eprev = en;
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (p < npass) {  // uniform
                    const double sc = rcp_nr(rad + fabs(sw[p]) + a.eta);
                    const double zr = sw[p] * sc, zi = a.eta * sc;
                    double p0r = 1.0, p0i = 0.0, p1r = fma(b[0], sc, zr), p1i = zi;
                    double q0r = 0.0, q0i = 0.0, q1r = 1.0, q1i = 0.0;
#pragma unroll
                    for (int i = 1; i < NP; ++i) {
                        if (i < n) {  // uniform
                            const double ar = fma(b[i], sc, zr), ai = zi;
                            const double ee = e2[i - 1] * sc * sc;
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
                    }
                    const double ip = rcp_nr(p1r * p1r + p1i * p1i) * sc;
                    const double tr = (q1r * p1r + q1i * p1i) * ip, ti = (q1i * p1r - q1r * p1i) * ip;
                    const double wv = mine[p] ? wk : 0.0;
                    accr[p] = fma(wv, a.is_dos ? -ti * 0.31830988618379067153776752674503 : tr, accr[p]);
                    acci[p] = a.is_dos ? 0.0 : fma(wv, ti, acci[p]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) red[(slot * 4 + p) * NP + r] = make_double2(accr[p], acci[p]);
        __syncthreads();
        for (int t = threadIdx.x; t < 4 * NP; t += 256) {
            const int p = t / NP, rr = t % NP;
            if (p * NP + rr < left) {
                double sr = 0.0, si = 0.0;
                for (int sl = 0; sl < SLOTS; ++sl) {
                    sr += red[(sl * 4 + p) * NP + rr].x;
                    si += red[(sl * 4 + p) * NP + rr].y;
                }
                a.partial[(int64_t)blockIdx.x * a.n_sweep + s0 + p * NP + rr] = make_double2(sr, si);
            }
        }
    }
}

// Matrix-valued scan (G_loc = sum_k w_k inv((omega + i eta) I - H(k))) on the same rows: one swept value per pass,
// lane r accumulates row r of the resolvent; the node slots of a wave are summed by shuffles, the waves of a block
// through LDS.  partial: [blocks][n_sweep][n*n] complex, component r + n*c (column-major, as `integrand_value`).
template <int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void gen_rows_gloc_kernel(GenRowsReduceArgs a) {
    __shared__ double2 red[4][NP * NP];
    constexpr int SLOTS = 256 / NP;
    const int n = a.n, nn = n * n;
    const int slot = threadIdx.x / NP, r = threadIdx.x % NP, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int s0 = 0; s0 < a.n_sweep; ++s0) {
        const double sw = a.sweep[s0];
        double gr[NP], gi[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            gr[j] = 0.0;
            gi[j] = 0.0;
        }
        for (int64_t k0 = (int64_t)blockIdx.x * SLOTS; k0 < a.nk; k0 += (int64_t)gridDim.x * SLOTS) {
            const int64_t k = k0 + slot;
            const bool act = k < a.nk;
            const int64_t kc = act ? k : 0;
            const double wk = act ? (a.w ? a.w[kc] : 1.0) : 0.0;
            const double* __restrict__ hin = a.H.base + view_off(a.H, kc);
            const int rr = r < n ? r : 0;
            double ar[NP], ai[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const bool real = r < n && j < n;
                const int jj = j < n ? j : 0;
                double vr, vi;
                rule_h_elem(a.H, hin, n, rr, jj, vr, vi);
                ar[j] = real ? -vr : 0.0;
                ai[j] = real ? -vi : 0.0;
            }
            panel_shift_row<NP, false>(n, sw, a.eta, r, ar, ai);
            panel_invert_rows<NP, false>(n, r, ar, ai);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                gr[j] = fma(wk, ar[j], gr[j]);
                gi[j] = fma(wk, ai[j], gi[j]);
            }
        }
        // the 64 / NP node slots of a wave
#pragma unroll
        for (int j = 0; j < NP; ++j) {
#pragma unroll
            for (int off = NP; off < 64; off <<= 1) {
                gr[j] += __shfl_xor(gr[j], off, 64);
                gi[j] += __shfl_xor(gi[j], off, 64);
            }
        }
        __syncthreads();  // the previous swept value's readers are done with `red`
        if (lane < NP) {
#pragma unroll
            for (int j = 0; j < NP; ++j) red[wave][r + NP * j] = make_double2(gr[j], gi[j]);
        }
        __syncthreads();
        for (int t = threadIdx.x; t < nn; t += 256) {
            const int rw = t % n, c = t / n;
            const int e = rw + NP * c;
            const double2 v0 = red[0][e], v1 = red[1][e], v2 = red[2][e], v3 = red[3][e];
            a.partial[((int64_t)blockIdx.x * a.n_sweep + s0) * nn + t] = make_double2((v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y));
        }
    }
}

static bool gen_rows_reduce_supported(const ReduceSpec& rs) {
    // 17...32 bands (two nodes per wave): resolvent traces through the tridiagonal kernel; matrix-valued scans stay wave-per-node
    return rs.herm && rs.H.base && rs.n > 4 &&
           (rs.n <= 16 ? (rs.integrand == ABZ_F_DOS || rs.integrand == ABZ_F_TRGLOC || rs.integrand == ABZ_F_GLOC)
                       : (rs.n <= 32 && !rs.H.compact && (rs.integrand == ABZ_F_DOS || rs.integrand == ABZ_F_TRGLOC)));
}

static int launch_gen_rows_gloc(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim) {
    const int np = rs.n <= 8 ? 8 : 16, nn = rs.n * rs.n;
    const int64_t blocks = std::min<int64_t>(cdiv2(rs.nk, 256 / np), 256);
    // swept values per launch: partial sums of at most 64 MB
    const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(rs.n_sweep, (64ll << 20) / (int64_t)(sizeof(double2) * blocks * nn)));
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * chunk * nn));
    if (rc) return rc;
    if ((rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)chunk * nn))) return rc;
    GenRowsReduceArgs a;
    a.H = rs.H;
    a.w = rs.w;
    a.partial = ctx->scratch[1].as<double2>();
    a.nk = rs.nk;
    a.n = rs.n;
    a.is_dos = 0;
    a.eta = rs.params[0];
    double2* outd = ctx->scratch[2].as<double2>();
    for (int s0 = 0; s0 < rs.n_sweep; s0 += chunk) {
        a.n_sweep = std::min(chunk, rs.n_sweep - s0);
        a.sweep = rs.sweep_dev + s0;
        {
            ProfScope ps(ctx, ABZ_K_REDUCE);
            if (np == 8)
                hipLaunchKernelGGL(gen_rows_gloc_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
            else
                hipLaunchKernelGGL(gen_rows_gloc_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
            ABZ_HIP(hipGetLastError());
            if ((rc = launch_final_reduce(ctx, a.partial, blocks, (int64_t)a.n_sweep * nn, rs.scale, outd))) return rc;
        }
        if (rs.out_dev) {  // chunk results stay in HBM; the next chunk reuses `outd` in stream order
            ABZ_HIP(hipMemcpyAsync(rs.out_dev + 2 * (size_t)s0 * nn, outd, sizeof(double2) * (size_t)a.n_sweep * nn,
                                   hipMemcpyDeviceToDevice, ctx->stream));
            continue;
        }
        ABZ_HIP(hipMemcpyAsync(out_reim + 2 * (size_t)s0 * nn, outd, sizeof(double2) * (size_t)a.n_sweep * nn, hipMemcpyDeviceToHost,
                               ctx->stream));
        ABZ_HIP(hipStreamSynchronize(ctx->stream));
    }
    return ABZ_OK;
}

static int launch_gen_rows_reduce(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim) {
    if (rs.integrand == ABZ_F_GLOC) return launch_gen_rows_gloc(ctx, rs, out_reim);
    const int np = rs.n <= 8 ? 8 : (rs.n <= 16 ? 16 : 32);
    const int64_t blocks = std::min<int64_t>(cdiv2(rs.nk, 256 / np), 256 * 2);
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * rs.n_sweep));
    if (rc) return rc;
    if ((rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)rs.n_sweep))) return rc;
    GenRowsReduceArgs a;
    a.H = rs.H;
    a.w = rs.w;
    a.sweep = rs.sweep_dev;
    a.partial = ctx->scratch[1].as<double2>();
    a.nk = rs.nk;
    a.n = rs.n;
    a.n_sweep = rs.n_sweep;
    a.is_dos = rs.integrand == ABZ_F_DOS ? 1 : 0;
    a.eta = rs.params[0];
    double2* outd = ctx->scratch[2].as<double2>();
    {
        ProfScope ps(ctx, ABZ_K_REDUCE);
        const bool tri = (abz_switch(SW_GEN_SUM_TRI) && rs.n_sweep >= 3) || np == 32;  // sweeps: tridiagonalise once, p'/p per swept value
        if (np == 32)
            hipLaunchKernelGGL(gen_rows_reduce_tri_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
        else if (tri && np == 8)
            hipLaunchKernelGGL(gen_rows_reduce_tri_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
        else if (tri)
            hipLaunchKernelGGL(gen_rows_reduce_tri_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
        else if (np == 8)
            hipLaunchKernelGGL(gen_rows_reduce_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
        else
            hipLaunchKernelGGL(gen_rows_reduce_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
        ABZ_HIP(hipGetLastError());
        if ((rc = launch_final_reduce(ctx, a.partial, blocks, rs.n_sweep, rs.scale, outd))) return rc;
    }
    if (rs.out_dev) {
        ABZ_HIP(hipMemcpyAsync(rs.out_dev, outd, sizeof(double2) * (size_t)rs.n_sweep, hipMemcpyDeviceToDevice, ctx->stream));
        return ABZ_OK;
    }
    ABZ_HIP(hipMemcpyAsync(out_reim, outd, sizeof(double2) * (size_t)rs.n_sweep, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    return ABZ_OK;
}

// DOS from CACHED EIGENVALUES for more than four bands: sum_k w_k sum_b (eta / pi) / ((omega - e_b(k))^2 + eta^2).
// One thread per node (the eigenvalue planes are read coalesced), eight swept values per pass in registers, one
// reciprocal (estimate + 2 Newton steps) per (k, band, omega); wave sums by shuffles, four waves through LDS, the block
// partials summed by final_reduce_kernel in a fixed order.  Replaces the wave-per-node loop of gen_reduce_kernel for this
// integrand (16 bands, 48^3, 16 omega: 0.94 ms there).
struct GenEigDosArgs {
    PlaneView E;
    const double* w;
    const double* sweep;
    double2* partial;  // [gridDim.x][n_sweep]
    int64_t nk;
    int n, n_sweep;
    double eta;
};

__global__ __launch_bounds__(256) void gen_eig_dos_kernel(GenEigDosArgs a) {
    constexpr int SW = 8;
    __shared__ double red[4][SW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double eta2 = a.eta * a.eta;
    for (int s0 = 0; s0 < a.n_sweep; s0 += SW) {
        double sw[SW], acc[SW];
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            sw[j] = a.sweep[min(s0 + j, a.n_sweep - 1)];
            acc[j] = 0.0;
        }
        for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < a.nk; k += (int64_t)gridDim.x * 256) {
            const double* __restrict__ e = a.E.base + view_off(a.E, k);
            double part[SW];
#pragma unroll
            for (int j = 0; j < SW; ++j) part[j] = 0.0;
            for (int b = 0; b < a.n; ++b) {
                const double eb = e[(int64_t)b * a.E.pitch];
#pragma unroll
                for (int j = 0; j < SW; ++j) {
                    const double de = sw[j] - eb;
                    part[j] += rcp_nr(fma(de, de, eta2));
                }
            }
            const double wk = a.w ? a.w[k] : 1.0;
#pragma unroll
            for (int j = 0; j < SW; ++j) acc[j] = fma(wk, part[j], acc[j]);
        }
        __syncthreads();  // the previous pass's readers are done with `red`
#pragma unroll
        for (int j = 0; j < SW; ++j) {
            const double v = wsum(acc[j]);
            if (lane == 0) red[wave][j] = v;
        }
        __syncthreads();
        if (threadIdx.x < SW && s0 + (int)threadIdx.x < a.n_sweep) {
            const int j = threadIdx.x;
            const double v = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
            a.partial[(int64_t)blockIdx.x * a.n_sweep + s0 + j] = make_double2(v * a.eta * 0.31830988618379067153776752674503, 0.0);
        }
    }
}

static int launch_gen_eig_dos(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim) {
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(cdiv2(rs.nk, 256), 256 * 4));
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * rs.n_sweep));
    if (rc) return rc;
    if ((rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)rs.n_sweep))) return rc;
    GenEigDosArgs a;
    a.E = rs.E;
    a.w = rs.w;
    a.sweep = rs.sweep_dev;
    a.partial = ctx->scratch[1].as<double2>();
    a.nk = rs.nk;
    a.n = rs.n;
    a.n_sweep = rs.n_sweep;
    a.eta = rs.params[0];
    double2* outd = ctx->scratch[2].as<double2>();
    {
        ProfScope ps(ctx, ABZ_K_REDUCE);
        hipLaunchKernelGGL(gen_eig_dos_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
        ABZ_HIP(hipGetLastError());
        if ((rc = launch_final_reduce(ctx, a.partial, blocks, rs.n_sweep, rs.scale, outd))) return rc;
    }
    if (rs.out_dev) {
        ABZ_HIP(hipMemcpyAsync(rs.out_dev, outd, sizeof(double2) * (size_t)rs.n_sweep, hipMemcpyDeviceToDevice, ctx->stream));
        return ABZ_OK;
    }
    ABZ_HIP(hipMemcpyAsync(out_reim, outd, sizeof(double2) * (size_t)rs.n_sweep, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    return ABZ_OK;
}

int launch_gen_reduce(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim) {
    if (lane_scan_supported(rs)) return launch_lane_scan(ctx, rs, out_reim);  // 5...8 bands, DOS / tr G: one node per lane
    if (gen_rows_reduce_supported(rs)) return launch_gen_rows_reduce(ctx, rs, out_reim);
    if (rs.integrand == ABZ_F_DOS_EIG && rs.E.base && rs.sweep_dev && rs.n_sweep >= 1) return launch_gen_eig_dos(ctx, rs, out_reim);
    if (big_supported(rs.n)) return launch_big_reduce(ctx, rs, out_reim);  // 33...64 bands: kernels_big.hip
    if (big_inverse_wanted(rs.n, rs.integrand, rs.herm) && rs.H.base && !rs.H.compact) return launch_big_reduce(ctx, rs, out_reim);
    const int ncomp = integrand_ncomp(rs.integrand, rs.n, rs.d);
    if (ncomp < 0 || rs.integrand == ABZ_F_LINEAR || rs.integrand == ABZ_F_LINEAR_X) {
        set_error("integrand %d is not available for n = %d bands", rs.integrand, rs.n);
        return ABZ_ERR_ARG;
    }
    const int n = rs.n, nn = n * n;
    const int64_t ncols = (int64_t)rs.n_sweep * ncomp;
    const size_t per = sizeof(double2) * (size_t)(3 * nn + ncols + (n + 1) / 2);
    if (per > 150 * 1024) {
        set_error("sweep of %d values x %d components does not fit the LDS accumulators; split the sweep", rs.n_sweep, ncomp);
        return ABZ_ERR_UNSUPPORTED;
    }
    int wpb = (int)((150 * 1024) / per);
    if (wpb > 4) wpb = 4;
    // about 8 waves per CU over the chip
    const int64_t target_waves = 256 * 8;
    int64_t chunk = std::max<int64_t>(1, cdiv2(rs.nk, target_waves));
    const int64_t nwaves = cdiv2(rs.nk, chunk);
    const int64_t blocks = cdiv2(nwaves, wpb);
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * wpb * ncols));
    if (rc) return rc;
    rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)ncols);
    if (rc) return rc;
    double2* partial = ctx->scratch[1].as<double2>();
    double2* outd = ctx->scratch[2].as<double2>();
    ABZ_HIP(hipMemsetAsync(partial, 0, sizeof(double2) * (size_t)(blocks * wpb * ncols), ctx->stream));
    GenReduceArgs a;
    a.Hplanes = rs.H;
    a.Eplanes = rs.E;
    a.w = rs.w;
    a.sweep = rs.sweep_dev;
    a.nk = rs.nk;
    a.chunk = chunk;
    a.n = n;
    a.n_sweep = rs.n_sweep;
    a.ncomp = ncomp;
    a.integrand = rs.integrand;
    for (int i = 0; i < 4; ++i) a.p[i] = rs.params[i];
    {
        ProfScope ps(ctx, ABZ_K_REDUCE);
        const size_t lds = per * wpb;
        ABZ_HIP(hipFuncSetAttribute((const void*)gen_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(gen_reduce_kernel, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a, wpb, partial);
        ABZ_HIP(hipGetLastError());
        hipLaunchKernelGGL(final_reduce2_kernel, dim3((unsigned)ncols), dim3(256), 0, ctx->stream, partial, blocks * wpb,
                           ncols, rs.scale, outd);
        ABZ_HIP(hipGetLastError());
    }
    if (rs.out_dev) {
        ABZ_HIP(hipMemcpyAsync(rs.out_dev, outd, sizeof(double2) * (size_t)ncols, hipMemcpyDeviceToDevice, ctx->stream));
        return ABZ_OK;
    }
    ABZ_HIP(hipMemcpyAsync(out_reim, outd, sizeof(double2) * (size_t)ncols, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// IAI innermost level on the device for n > 4: one WORKGROUP per 1-D integral (gen_inner_panel_kernel).  (The first
// version, one wavefront per integral with the node's matrices in LDS, was slower than the host-driven loop at 16
// bands and is gone.)
// ------------------------------------------------------------------------------------------
struct GenInnerArgs {
    const double2* src;
    const int64_t* slot;
    const double* lo;
    const double* hi;
    const double* atol;
    const double* sweep_arr;
    int64_t nint, maxevals;
    int n, M, first, d, ncomp, integrand, has_rtol;
    int herm;  // Hermitian series with first = -(M - 1) / 2: +f and -f can be folded
    double inv_period, sweep, rtol_user;
    double p[4];
    double2* I_out;
    double* E_out;
    int64_t* nev_out;
    int* status_out;
    int pair;       // adapt_step_pair instead of adapt_step (ABZ_ADAPT_PAIR=0: the one-lane step)
    double sc[16];  // sincospi_poly's coefficients (kernel arguments stay in scalar registers / the scalar cache)
};


// Resolvent-trace integrands: the integral's coefficient
// set is staged in LDS once and stays there for its whole adaptive loop; each round's 15 / 30 GK nodes
// are evaluated 256/NP at a time by NP-lane groups (panel_inverse_row), thread 0 runs adapt_step.
template <int NP, bool PAD, int NT, int WPE, bool FOLD = false, bool FMAC = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE ? WPE : 1, WPE ? WPE : 8))) void gen_inner_panel_kernel(GenInnerArgs a) {
    static_assert(!FMAC || (FOLD && NP == 16), "the FMA-with-broadcast inversion is written for 16 lanes per node");
    static_assert(!FOLD || PAD, "the folded series is built for the zero-padded layout");
    extern __shared__ double2 lds_ip[];
    constexpr int LPN = NP;  // lanes per node: lane r owns row r
    constexpr int SLOTS = NT / LPN;
    constexpr int MS = ABZ_PANEL_MAXSEG;
    const int n = a.n, nn = n * n, M = a.M, nc = a.ncomp;  // nc = 1 (DOS, TRGLOC)
    double2* coef = lds_ip;
    double* g = reinterpret_cast<double*>(coef + (size_t)M * (PAD ? NP * NP : nn));
    double* seg_a = g;
    double* seg_b = seg_a + MS;
    double* seg_E = seg_b + MS;
    gkc* seg_I = reinterpret_cast<gkc*>(seg_E + MS);
    gkc* vals = seg_I + (size_t)MS * nc;
    int* heap = reinterpret_cast<int*>(vals + (size_t)30 * nc);
    double* ctl = reinterpret_cast<double*>(heap + MS);
    double* heapE = ctl + 8;  // adapt_step_pair's mirror of seg_E[heap[.]]
    const int slot = threadIdx.x / LPN, r = threadIdx.x % LPN;
    for (int64_t q = blockIdx.x; q < a.nint; q += gridDim.x) {
        AdaptStateT<1> st;
        AdaptParent par;
        __syncthreads();  // the previous integral's readers are done with coef / ctl
        if constexpr (FOLD)
            panel_stage_fold<NP>(coef, a.src + a.slot[q] * ((int64_t)M * nn), n, M, a.sweep_arr ? a.sweep_arr[q] : a.sweep, a.p[0]);
        else
            panel_stage<NP, PAD>(coef, a.src + a.slot[q] * ((int64_t)M * nn), n, M);
        if (threadIdx.x == 0) {
            adapt_init(st, a.atol[q], a.has_rtol != 0, a.rtol_user, a.lo[q], a.hi[q], ctl);
            ctl[5] = 0.0;  // done flag
        }
        const double swq = a.sweep_arr ? a.sweep_arr[q] : a.sweep;
        while (true) {
            __syncthreads();
            if (ctl[5] != 0.0) break;  // uniform
            const int nnodes = 15 * (int)ctl[0];
            for (int t0 = 0; t0 < nnodes; t0 += SLOTS) {
                const int t = t0 + slot;
                if (t0 + (int)(threadIdx.x >> 6) * (64 / LPN) >= nnodes) continue;  // no node for this wave in the pass
                const bool act = t < nnodes;
                const int tt = act ? t : 0;
                const int pnl = tt / 15, i = tt - 15 * pnl;
                const double x = gk15_node(ctl[1 + 2 * pnl], ctl[2 + 2 * pnl], i);
                double tr, ti;
                if constexpr (FOLD) {
                    double ar[NP], ai[NP];
                    double zr, zi;
                    if constexpr (FMAC)
                        sincospi_poly(a.sc, 2.0 * (x * a.inv_period), zi, zr);
                    else
                        sincospi(2.0 * (x * a.inv_period), &zi, &zr);
                    panel_series_row_fold<NP>(coef, M, zr, zi, r, ar, ai);
                    if constexpr (FMAC) {
                        panel_inverse_trace_fmac(n, r, ar, ai, tr, ti);
                    } else {
                        panel_invert_rows<NP, PAD>(n, r, ar, ai);
                        panel_trace<NP>(ar, ai, n, r, tr, ti);
                    }
                } else if constexpr (NP == 32) {  // 17...32 bands: the trace from the tridiagonal form (no inverse)
                    double hr[NP], hi[NP];
                    double zr, zi, pr, pi;
                    sincospi(2.0 * (x * a.inv_period), &zi, &zr);
                    sincospi(2.0 * ((double)a.first * (x * a.inv_period)), &pi, &pr);
                    panel_series_row<NP, PAD>(coef, n, M, zr, zi, pr, pi, r, hr, hi);  // row r of -H
                    if (!PAD) {
#pragma unroll
                        for (int j = 0; j < NP; ++j) {
                            const bool real = r < n && j < n;
                            hr[j] = real ? hr[j] : 0.0;
                            hi[j] = real ? hi[j] : 0.0;
                        }
                    }
                    rows_trace_resolvent_tri<NP>(n, r, hr, hi, swq, a.p[0], tr, ti);
                } else {
                    double ar[NP], ai[NP];
                    panel_inverse_row<NP, PAD>(coef, n, M, a.first, x * a.inv_period, swq, a.p[0], r, ar, ai);
                    panel_trace<NP>(ar, ai, n, r, tr, ti);
                }
                if (act && r == 0) {
                    if (a.integrand == ABZ_F_DOS) {
                        vals[t].re = -ti * 0.31830988618379067153776752674503;
                        vals[t].im = 0.0;
                    } else {
                        vals[t].re = tr;
                        vals[t].im = ti;
                    }
                }
            }
            __syncthreads();
            if (threadIdx.x < 2) {
                InnerOut out;
                out.I = a.I_out + q * nc;
                out.E = a.E_out + q;
                out.nev = a.nev_out + q;
                out.status = a.status_out + q;
                if (a.pair) {  // uniform: lanes 0 and 1 share the two panel rules
                    if (adapt_step_pair<MS>(st, par, (int)threadIdx.x, seg_a, seg_b, seg_E, seg_I, vals, heap, heapE, ctl, a.maxevals, out))
                        ctl[5] = 1.0;
                } else if (threadIdx.x == 0) {
                    if (adapt_step<false, 1, MS>(st, nc, seg_a, seg_b, seg_E, seg_I, vals, heap, ctl, a.maxevals, out)) ctl[5] = 1.0;
                }
            }
        }
    }
}

static bool gen_inner_panel_fits(int n, int M, int integrand, int* np_out, size_t* lds_out, bool* pad_out) {
    if (!(integrand == ABZ_F_DOS || integrand == ABZ_F_TRGLOC) || n > 32) return false;  // (33...64 bands: the host-driven node path)
    const int np = n <= 8 ? 8 : (n <= 16 ? 16 : 32);
    const size_t rest = sizeof(double) * ((size_t)inner_group_doubles(1, ABZ_PANEL_MAXSEG) + ABZ_PANEL_MAXSEG);  // + heapE
    size_t lds = sizeof(double2) * (size_t)M * np * np + rest;  // zero-padded set
    // 17...32 bands: the kernel fits 256 registers, so two workgroups share a CU when their sets do -- the set without its zero
    // padding if that is what it takes (17 bands x 5 coefficients: 80 KB padded, 23 KB without; IAI 55 -> 72 M nodes/s)
    const size_t bare = sizeof(double2) * (size_t)M * n * n + rest;
    const bool pad = np == 32 ? (lds <= 72 * 1024 || bare > 72 * 1024) && lds <= 150 * 1024 : lds <= 150 * 1024;
    if (!pad) lds = bare;
    if (lds > 150 * 1024) return false;
    if (np_out) *np_out = np;
    if (lds_out) *lds_out = lds;
    if (pad_out) *pad_out = pad;
    return true;
}

// the block-per-integral kernel is the one worth running by default (see iai_host.cpp)
bool gen_inner_panel_supported(int n, int M, int integrand, bool herm) {
    // (17...32 bands: the 32-lane instance takes the trace from the tridiagonal form of Hermitian(h); a series that is not
    // Hermitian goes through the host-driven node path and big_inverse_kernel)
    return n > 4 && n <= ABZ_MAX_BANDS && (n <= 16 || herm) && gen_inner_panel_fits(n, M, integrand, nullptr, nullptr, nullptr);
}

int launch_gen_inner_adaptive(abz_ctx* ctx, const InnerSpec& is) {
    if (is.nint == 0) return ABZ_OK;
    GenInnerArgs a;
    a.src = is.src;
    a.slot = is.slot;
    a.lo = is.lo;
    a.hi = is.hi;
    a.atol = is.atol;
    a.sweep_arr = is.sweep_arr;
    a.nint = is.nint;
    a.maxevals = is.maxevals;
    a.n = is.n;
    a.M = is.M;
    a.first = is.first;
    a.d = is.d;
    a.ncomp = integrand_ncomp(is.integrand, is.n, is.d);
    a.integrand = is.integrand;
    a.has_rtol = is.has_rtol ? 1 : 0;
    a.herm = (is.herm && (is.M & 1) && is.first == -((is.M - 1) / 2)) ? 1 : 0;
    a.inv_period = 1.0 / is.period;
    a.sweep = is.sweep;
    a.rtol_user = is.rtol_user;
    for (int i = 0; i < 4; ++i) a.p[i] = is.params[i];
    a.I_out = is.I_out;
    a.E_out = is.E_out;
    a.nev_out = is.nev_out;
    a.status_out = is.status_out;
    for (int i = 0; i < 16; ++i) a.sc[i] = kSinCosPiCoef[i];
    a.pair = abz_switch(SW_ADAPT_PAIR) != 0;  // per call: tests compare both
    int np = 0;
    size_t plds = 0;
    bool pad = false;
    if (!gen_inner_panel_fits(is.n, is.M, is.integrand, &np, &plds, &pad)) {
        set_error("inner panel kernel: n = %d, M = %d, integrand %d do not fit (the caller falls back to host-driven rounds)", is.n, is.M, is.integrand);
        return ABZ_ERR_UNSUPPORTED;
    }
    // one workgroup per integral: the dispatcher hands the next integral to whichever CU frees a slot (a static share of
    // integrals per persistent workgroup left the second half of a launch half empty: 3.1 of 4 wave slots occupied)
    const int64_t blocks = std::min<int64_t>(is.nint, (int64_t)1 << 20);
    ProfScope ps(ctx, ABZ_K_EVAL);
    // workgroup: a round's 30 nodes in one pass where the lane groups allow it -- 256 threads = 32 nodes of 8 lanes, 512
    // threads = 32 nodes of 16 lanes at 4 waves per SIMD (32 rows x 2 arrays alone fill 128 VGPRs: no bound there)
#define ABZ_IPANEL3(NPV, PV, NTV, WV, ...)                                                                                \
    {                                                                                                                     \
        auto kfn = gen_inner_panel_kernel<NPV, PV, NTV, WV, ##__VA_ARGS__>;                                               \
        ABZ_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));            \
        hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(NTV), plds, ctx->stream, a);                                 \
    }
    const bool fold = abz_switch(SW_IPANEL_FOLD) != 0, fmac = abz_switch(SW_IPANEL_FMAC) != 0;  // per call: tests compare the variants
    if (np == 16 && pad && a.herm && fold && fmac) {  // config 5's shape
        ABZ_IPANEL3(16, true, 512, 4, true, true)
    } else if (np == 16 && pad && a.herm && fold) {  // the same with the pivot rows broadcast by separate DPP moves
        ABZ_IPANEL3(16, true, 512, 4, true)
    } else if (np == 8) {
        if (pad) ABZ_IPANEL3(8, true, 256, 0)
        else ABZ_IPANEL3(8, false, 256, 0)
    } else if (np == 16) {
        if (pad) ABZ_IPANEL3(16, true, 512, 4)
        else ABZ_IPANEL3(16, false, 512, 4)
    } else {  // 17...32 bands: 256 threads = 8 nodes of 32 lanes
        const bool two = plds <= 72 * 1024;  // (two workgroups per CU: hold the kernel to 256 registers)
        if (pad && two) ABZ_IPANEL3(32, true, 256, 2)
        else if (pad) ABZ_IPANEL3(32, true, 256, 0)
        else if (two) ABZ_IPANEL3(32, false, 256, 2)
        else ABZ_IPANEL3(32, false, 256, 0)
    }
#undef ABZ_IPANEL3
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

}  // namespace abz
