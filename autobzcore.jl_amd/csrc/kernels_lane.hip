// 5...8 bands on full grids, ONE NODE PER LANE (ref: src/fourier.jl:127-174 evaluates the series on the PTR grid,
// eigen(Hermitian(h)) at every node is LAPACK for these sizes, src/dos_ggr.jl:19): the row layout of kernels_generic.hip
// gives a 5 x 5 matrix eight lanes and pays a cross-lane sum or broadcast for every column of every Householder step --
// ~1000 wave instructions per EIGHT nodes, 15x what a node of a 4-band model costs in the closed-form kernels (round 4:
// 64^3 nodes, H + eigenvalues: 4 bands 0.032 ms, 5 bands 0.158 ms).  Here a lane keeps the upper triangle of its node's
// matrix in registers (N (N + 1) / 2 complex numbers, N a template parameter: every index is static):
//   series        folded level-1 set of the line in a wave-private LDS slab (c_0, s_f = c_f + c_f^T, t_f = c_f - c_f^T,
//                 rows_device.h), broadcast reads, 4 FMAs per pair and frequency pair (+f, -f);
//   tridiagonal   LAPACK's zhetd2 (UPLO = 'U') unrolled: zlarfg, the Hermitian matrix-vector product on the stored
//                 triangle, the rank-2 update -- no cross-lane traffic at all;
//   eigenvalues   the root-free QR iteration of tri_eig_kernel, one matrix per lane as there (tri_qr_lane), on the
//                 (d, e^2) the lane has just produced -- nothing goes through HBM;
//   sums          tr inv(z I - H) = p'(z) / p(z) from the tridiagonal for every swept value, wave-reduced per value into
//                 a wave-private accumulator row: any number of values in one pass over the grid.
// Serves rule builds (H in either layout and / or eigenvalues; full grids and the runs of a symmetric node list), store-free
// sums and DOS / tr G scans of cached rules for Hermitian series with a symmetric frequency range; matrix-valued scans, GGR
// builds and the IAI panels stay on the row kernels.
#include <utility>

#include "abz_internal.h"
#include "rows_device.h"

namespace abz {

namespace {

template <int N>
struct LaneIdx {
    static constexpr int P = N * (N + 1) / 2;                       // stored pairs a <= b
    static constexpr int pid(int a, int b) { return b * (b + 1) / 2 + a; }  // a <= b
};

struct LaneArgs {
    const double2* src;  // level-1 sets [line][M][N * N]
    const double2* tab;
    PlaneView H, E;
    int64_t nlines;
    int npt, M, first;
    const int64_t* run_start;  // node lists: line l owns nodes [run_start[l], run_start[l + 1]) with grid indices gi (null: full grid lines)
    const int32_t* gi;
    // sums
    const double* sweep;  // device [n_sweep]
    int n_sweep, is_dos;
    double eta;
    double2* partial;  // [waves][n_sweep]
};

// the folded set of a line in the wave's slab: block 0: c_0 (pair p at [p]); block f >= 1: s_f at [P + (f - 1) (2 P) + p],
// t_f at [P + (f - 1) (2 P) + P + p]
template <int N>
__device__ __forceinline__ void lane_stage(double2* __restrict__ slab, const double2* __restrict__ src, int M, int lane) {
    constexpr int P = LaneIdx<N>::P;
    const int F = (M - 1) / 2;
    for (int t = lane; t < (1 + 2 * F) * P; t += 64) {
        const int blk = t / P, p = t - blk * P;
        // pair p -> (a, b): b = largest with b (b + 1) / 2 <= p
        int b = 0;
        while ((b + 1) * (b + 2) / 2 <= p) ++b;
        const int a = p - b * (b + 1) / 2;
        double2 v;
        if (blk == 0) {
            v = src[(size_t)F * N * N + a + N * b];
            if (a == b) v.y = 0.0;
        } else {
            const int f = (blk + 1) >> 1;
            const double2 c = src[(size_t)(F + f) * N * N + a + N * b], ct = src[(size_t)(F + f) * N * N + b + N * a];
            v = (blk & 1) ? make_double2(c.x + ct.x, c.y + ct.y) : make_double2(c.x - ct.x, c.y - ct.y);
        }
        slab[t] = v;
    }
}

// upper triangle of H(k) at this lane's node: z = e^{2 pi i k_1}
template <int N>
__device__ __forceinline__ void lane_series(const double2* __restrict__ slab, int M, double zr, double zi, double (&hr)[LaneIdx<N>::P],
                                            double (&hi)[LaneIdx<N>::P]) {
    constexpr int P = LaneIdx<N>::P;
    const int F = (M - 1) / 2;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const double2 c = slab[p];
        hr[p] = c.x;
        hi[p] = c.y;
    }
    double pr = 1.0, pi = 0.0;
    for (int f = 1; f <= F; ++f) {
        const double nr = pr * zr - pi * zi, ni = pr * zi + pi * zr;
        pr = nr;
        pi = ni;
        const double2* __restrict__ sf = slab + P + (size_t)(f - 1) * (2 * P);
        const double2* __restrict__ tf = sf + P;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const double2 s = sf[p], t = tf[p];
            hr[p] = fma(s.x, pr, hr[p]);
            hr[p] = fma(-s.y, pi, hr[p]);
            hi[p] = fma(t.x, pi, hi[p]);
            hi[p] = fma(t.y, pr, hi[p]);
        }
    }
}

// zhetd2, UPLO = 'U', on the lane's own matrix: d[0 .. N-1], e2[0 .. N-2] = squares of the (real) off-diagonal
template <int N>
__device__ __forceinline__ void lane_tridiag(double (&hr)[LaneIdx<N>::P], double (&hi)[LaneIdx<N>::P], double (&d)[N], double (&e2)[N]) {
    using I = LaneIdx<N>;
#pragma unroll
    for (int i = N - 2; i >= 0; --i) {
        // H(i) annihilates A(0 : i-1, i+1); alpha = A(i, i+1)
        const double alr = hr[I::pid(i, i + 1)], ali = hi[I::pid(i, i + 1)];
        double xn2 = 0.0;
#pragma unroll
        for (int r = 0; r < i; ++r) xn2 += hr[I::pid(r, i + 1)] * hr[I::pid(r, i + 1)] + hi[I::pid(r, i + 1)] * hi[I::pid(r, i + 1)];
        const bool refl = xn2 > 0.0 || ali != 0.0;
        const double nrm2 = alr * alr + ali * ali + xn2;
        const double nrm = nrm2 > 0.0 ? sqrt(nrm2) : 0.0;
        const double beta = refl ? (alr >= 0.0 ? -nrm : nrm) : alr;  // zlarfg: beta = -sign(alphr) ||(alpha, x)||
        e2[i] = beta * beta;
        d[i + 1] = hr[I::pid(i + 1, i + 1)];
        if (i == 0) continue;  // (a 1 x 1 block is left: only the phase of alpha was removed, nothing to update)
        const double ib = refl ? 1.0 / beta : 0.0;
        const double taur = refl ? (beta - alr) * ib : 0.0, taui = refl ? -ali * ib : 0.0;
        // x *= 1 / (alpha - beta)
        const double dr = alr - beta, di = ali;
        const double idn = refl ? 1.0 / (dr * dr + di * di) : 0.0;
        const double scr = dr * idn, sci = -di * idn;
        double vr[N], vi[N];
#pragma unroll
        for (int r = 0; r < N; ++r) {
            vr[r] = 0.0;
            vi[r] = 0.0;
        }
#pragma unroll
        for (int r = 0; r < i; ++r) {
            const double xr = hr[I::pid(r, i + 1)], xi = hi[I::pid(r, i + 1)];
            vr[r] = xr * scr - xi * sci;
            vi[r] = xr * sci + xi * scr;
        }
        vr[i] = 1.0;
        // p = tau A(0:i, 0:i) v on the stored triangle
        double pr[N], pi[N];
#pragma unroll
        for (int r = 0; r < N; ++r) {
            pr[r] = 0.0;
            pi[r] = 0.0;
        }
#pragma unroll
        for (int c = 0; c <= i; ++c) {
            pr[c] = fma(hr[I::pid(c, c)], vr[c], pr[c]);
            pi[c] = fma(hr[I::pid(c, c)], vi[c], pi[c]);
#pragma unroll
            for (int r = 0; r < c; ++r) {
                const double ar = hr[I::pid(r, c)], ai = hi[I::pid(r, c)];
                // p_r += A_rc v_c ;  p_c += conj(A_rc) v_r
                pr[r] = fma(ar, vr[c], pr[r]);
                pr[r] = fma(-ai, vi[c], pr[r]);
                pi[r] = fma(ar, vi[c], pi[r]);
                pi[r] = fma(ai, vr[c], pi[r]);
                pr[c] = fma(ar, vr[r], pr[c]);
                pr[c] = fma(ai, vi[r], pr[c]);
                pi[c] = fma(ar, vi[r], pi[c]);
                pi[c] = fma(-ai, vr[r], pi[c]);
            }
        }
        double dotr = 0.0, doti = 0.0;  // (tau A v)^H v
#pragma unroll
        for (int r = 0; r <= i; ++r) {
            const double tr = taur * pr[r] - taui * pi[r], ti = taur * pi[r] + taui * pr[r];
            pr[r] = tr;
            pi[r] = ti;
            dotr = fma(tr, vr[r], dotr);
            dotr = fma(ti, vi[r], dotr);
            doti = fma(tr, vi[r], doti);
            doti = fma(-ti, vr[r], doti);
        }
        // alpha2 = -1/2 tau dot;  p += alpha2 v
        const double a2r = -0.5 * (taur * dotr - taui * doti), a2i = -0.5 * (taur * doti + taui * dotr);
#pragma unroll
        for (int r = 0; r <= i; ++r) {
            pr[r] = fma(a2r, vr[r], pr[r]);
            pr[r] = fma(-a2i, vi[r], pr[r]);
            pi[r] = fma(a2r, vi[r], pi[r]);
            pi[r] = fma(a2i, vr[r], pi[r]);
        }
        // A_rc -= v_r conj(p_c) + p_r conj(v_c), r <= c <= i
#pragma unroll
        for (int c = 0; c <= i; ++c) {
#pragma unroll
            for (int r = 0; r <= c; ++r) {
                hr[I::pid(r, c)] -= (vr[r] * pr[c] + vi[r] * pi[c]) + (pr[r] * vr[c] + pi[r] * vi[c]);
                if (r < c) hi[I::pid(r, c)] -= (vi[r] * pr[c] - vr[r] * pi[c]) + (pi[r] * vr[c] - pr[r] * vi[c]);
            }
        }
    }
    d[0] = hr[I::pid(0, 0)];
    e2[N - 1] = 0.0;
}

// sum over the wave's nodes of tr inv((w + i eta) I - H) for every swept value, from the lanes' tridiagonals (d, e2): p' / p by
// the three-term recurrence, scaled to unit Gershgorin radius; `wk`: the node's weight (0 for lanes without a node)
template <int N>
__device__ __forceinline__ void lane_trace_sums(const double (&d)[N], const double (&e2)[N], double wk, const double* __restrict__ sweep, int n_sweep,
                                                double eta, int lane, double2* __restrict__ acc) {
    double rad = 0.0, eprev = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double en = (j + 1 < N && e2[j] > 0.0) ? sqrt(e2[j]) : 0.0;
        rad = fmax(rad, fabs(d[j]) + eprev + en);
        eprev = en;
    }
    for (int s = 0; s < n_sweep; ++s) {
        const double sw = sweep[s];
        const double sc = 1.0 / (rad + fabs(sw) + eta);
        const double zr = sw * sc, zi = eta * sc;
        double p0r = 1.0, p0i = 0.0, p1r = zr - d[0] * sc, p1i = zi;
        double q0r = 0.0, q0i = 0.0, q1r = 1.0, q1i = 0.0;
#pragma unroll
        for (int j = 1; j < N; ++j) {
            const double ar = zr - d[j] * sc, ai = zi;
            const double ee = e2[j - 1] * sc * sc;
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
        const double ip = wk * sc / (p1r * p1r + p1i * p1i);
        double tr = (q1r * p1r + q1i * p1i) * ip, ti = (q1i * p1r - q1r * p1i) * ip;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            tr += __shfl_xor(tr, off, 64);
            ti += __shfl_xor(ti, off, 64);
        }
        if (lane == 0) {
            double2 t = acc[s];
            t.x += tr;
            t.y += ti;
            acc[s] = t;
        }
    }
}

// MODE bit 0: store H, bit 1: eigenvalues, bit 2: sums of resolvent traces
template <int N, int MODE>
__global__ __launch_bounds__(256) void lane_grid_kernel(LaneArgs a) {
    constexpr int P = LaneIdx<N>::P;
    extern __shared__ double2 lds_ln[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int F = (a.M - 1) / 2;
    const int slab_elems = (1 + 2 * F) * P;
    // per wave: the folded set | (eigenvalues) ld, le [N + 2][64] doubles | (sums) accumulators [n_sweep] complex
    const size_t wave_bytes = sizeof(double2) * (size_t)slab_elems + ((MODE & 2) ? sizeof(double) * 2 * (N + 2) * 64 : 0) +
                              ((MODE & 4) ? sizeof(double2) * (size_t)a.n_sweep : 0);
    char* const base = reinterpret_cast<char*>(lds_ln) + (size_t)wave * ((wave_bytes + 15) / 16 * 16);
    double2* const slab = reinterpret_cast<double2*>(base);
    double(*const ld)[64] = reinterpret_cast<double(*)[64]>(base + sizeof(double2) * (size_t)slab_elems);
    double(*const le)[64] = ld + (N + 2);
    double2* const acc = reinterpret_cast<double2*>(base + sizeof(double2) * (size_t)slab_elems + ((MODE & 2) ? sizeof(double) * 2 * (N + 2) * 64 : 0));
    if constexpr (MODE & 4) {
        for (int s = lane; s < a.n_sweep; s += 64) acc[s] = make_double2(0.0, 0.0);
    }
    const int ppl = (a.npt + 63) / 64;
    const int64_t units = a.nlines * ppl;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    int64_t staged = -1;
    for (int64_t u = (int64_t)blockIdx.x * 4 + wave; u < units; u += nwaves) {
        const int64_t line = u / ppl;
        const int i0 = (int)(u - line * ppl) * 64;
        if (line != staged) {
            __builtin_amdgcn_wave_barrier();
            lane_stage<N>(slab, a.src + line * ((int64_t)a.M * N * N), a.M, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            staged = line;
        }
        const int64_t kbase = a.run_start ? a.run_start[line] : line * a.npt;
        const int count = a.run_start ? (int)(a.run_start[line + 1] - kbase) : a.npt;
        if (i0 >= count) continue;  // (node lists: a run is at most npt long, usually shorter)
        const bool act = i0 + lane < count;
        const int64_t k = kbase + (act ? i0 + lane : 0);
        const double2 z = a.tab[a.gi ? a.gi[k] : (int)(k - kbase)];
        double hr[P], hi[P];
        lane_series<N>(slab, a.M, z.x, z.y, hr, hi);
        if constexpr (MODE & 1) {
            if (act) {
                double* __restrict__ ho = a.H.base + view_off(a.H, k);
#pragma unroll
                for (int b = 0; b < N; ++b) {
#pragma unroll
                    for (int r = 0; r <= b; ++r) {
                        const double vr = hr[LaneIdx<N>::pid(r, b)], vi = r < b ? hi[LaneIdx<N>::pid(r, b)] : 0.0;
                        if (a.H.compact) {
                            ho[(int64_t)(b * b + 2 * r) * a.H.pitch] = vr;
                            if (r < b) ho[(int64_t)(b * b + 2 * r + 1) * a.H.pitch] = vi;
                        } else {
                            ho[(int64_t)(2 * (r + N * b)) * a.H.pitch] = vr;
                            ho[(int64_t)(2 * (r + N * b) + 1) * a.H.pitch] = vi;
                            if (r < b) {
                                ho[(int64_t)(2 * (b + N * r)) * a.H.pitch] = vr;
                                ho[(int64_t)(2 * (b + N * r) + 1) * a.H.pitch] = -vi;
                            }
                        }
                    }
                }
            }
        }
        if constexpr ((MODE & 6) != 0) {
            double d[N], e2[N];
            lane_tridiag<N>(hr, hi, d, e2);
            if constexpr (MODE & 2) {
                double anorm2 = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    ld[j][lane] = d[j];
                    le[j][lane] = e2[j];
                    anorm2 = fmax(anorm2, fmax(d[j] * d[j], e2[j]));
                }
                ld[N][lane] = 0.0;
                le[N][lane] = 0.0;
                ld[N + 1][lane] = 0.0;
                le[N + 1][lane] = 0.0;
                const int left = tri_qr_lane(ld, le, N, lane, anorm2);
                double v[N];
#pragma unroll
                for (int j = 0; j < N; ++j) v[j] = left > 0 ? __builtin_nan("") : ld[j][lane];
#pragma unroll
                for (int pass = 0; pass < N; ++pass) {
#pragma unroll
                    for (int j = pass & 1; j + 1 < N; j += 2) {
                        const double lo = fmin(v[j], v[j + 1]), hh = fmax(v[j], v[j + 1]);
                        v[j] = lo;
                        v[j + 1] = hh;
                    }
                }
                if (act) {
                    double* __restrict__ eo = a.E.base + view_off(a.E, k);
#pragma unroll
                    for (int j = 0; j < N; ++j) eo[(int64_t)j * a.E.pitch] = v[j];
                }
            }
            if constexpr (MODE & 4) lane_trace_sums<N>(d, e2, act ? 1.0 : 0.0, a.sweep, a.n_sweep, a.eta, lane, acc);
        }
    }
    if constexpr (MODE & 4) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t wg = (int64_t)blockIdx.x * 4 + wave;
        for (int s = lane; s < a.n_sweep; s += 64) {
            double2 t = acc[s];
            if (a.is_dos) t = make_double2(-t.y * 0.31830988618379067153776752674503, 0.0);
            a.partial[wg * a.n_sweep + s] = t;
        }
    }
}


// DOS / tr G scans of a cached rule (any node list): a lane loads the upper triangle of its node's matrix from the rule's planes
// (either layout; consecutive lanes read consecutive doubles of a plane), tridiagonalises it in registers and adds every
// swept value's trace to the wave's accumulator row
struct LaneScanArgs {
    PlaneView H;
    const double* w;  // node weights or null
    int64_t nk;
    const double* sweep;
    int n_sweep, is_dos;
    double eta;
    double2* partial;
};
template <int N>
__global__ __launch_bounds__(256) void lane_scan_kernel(LaneScanArgs a) {
    constexpr int P = LaneIdx<N>::P;
    extern __shared__ double2 lds_ls[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double2* const acc = lds_ls + (size_t)wave * a.n_sweep;
    for (int s = lane; s < a.n_sweep; s += 64) acc[s] = make_double2(0.0, 0.0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t units = (a.nk + 63) / 64, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t u = (int64_t)blockIdx.x * 4 + wave; u < units; u += nwaves) {
        const int64_t k = u * 64 + lane;
        const bool act = k < a.nk;
        const int64_t kk = act ? k : a.nk - 1;
        const double* __restrict__ hin = a.H.base + view_off(a.H, kk);
        double hr[P], hi[P];
#pragma unroll
        for (int b = 0; b < N; ++b) {
#pragma unroll
            for (int r = 0; r <= b; ++r) {
                const int64_t pl = a.H.compact ? (int64_t)(b * b + 2 * r) : (int64_t)(2 * (r + N * b));
                hr[LaneIdx<N>::pid(r, b)] = hin[pl * a.H.pitch];
                hi[LaneIdx<N>::pid(r, b)] = r < b ? hin[(pl + 1) * a.H.pitch] : 0.0;
            }
        }
        double d[N], e2[N];
        lane_tridiag<N>(hr, hi, d, e2);
        lane_trace_sums<N>(d, e2, act ? (a.w ? a.w[kk] : 1.0) : 0.0, a.sweep, a.n_sweep, a.eta, lane, acc);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t wg = (int64_t)blockIdx.x * 4 + wave;
    for (int s = lane; s < a.n_sweep; s += 64) {
        double2 t = acc[s];
        if (a.is_dos) t = make_double2(-t.y * 0.31830988618379067153776752674503, 0.0);
        a.partial[wg * a.n_sweep + s] = t;
    }
}

size_t lane_wave_bytes(int n, int M, int mode, int n_sweep) {
    const int P = n * (n + 1) / 2, F = (M - 1) / 2;
    const size_t b = sizeof(double2) * (size_t)(1 + 2 * F) * P + ((mode & 2) ? sizeof(double) * 2 * (size_t)(n + 2) * 64 : 0) +
                     ((mode & 4) ? sizeof(double2) * (size_t)n_sweep : 0);
    return (b + 15) / 16 * 16;
}

template <int N>
int lane_launch_n(abz_ctx* ctx, const LaneArgs& a, int mode, size_t lds, int64_t blocks) {
#define ABZ_LN(MV)                                                                                                             \
    {                                                                                                                          \
        ABZ_HIP(hipFuncSetAttribute((const void*)lane_grid_kernel<N, MV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((lane_grid_kernel<N, MV>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);                 \
    }
    if (mode == 1) ABZ_LN(1)
    else if (mode == 2) ABZ_LN(2)
    else if (mode == 3) ABZ_LN(3)
    else if (mode == 4) ABZ_LN(4)
    else return ABZ_ERR_UNSUPPORTED;
#undef ABZ_LN
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

int lane_launch(abz_ctx* ctx, int n, const LaneArgs& a, int mode, size_t lds, int64_t blocks) {
    switch (n) {
        case 5: return lane_launch_n<5>(ctx, a, mode, lds, blocks);
        case 6: return lane_launch_n<6>(ctx, a, mode, lds, blocks);
        case 7: return lane_launch_n<7>(ctx, a, mode, lds, blocks);
        case 8: return lane_launch_n<8>(ctx, a, mode, lds, blocks);
    }
    return ABZ_ERR_UNSUPPORTED;
}

bool lane_shape_ok(int n, int M, int first, int npt, bool herm) {
    return abz_switch(SW_LANE_KERNELS) != 0 && herm && n >= 5 && n <= 8 && (M & 1) && first == -((M - 1) / 2) && npt >= 1 && npt < 65536;
}

}  // namespace

// rule builds on full grids: H (either layout) and / or eigenvalues
bool lane_grid_supported(const GenSpec& gs) {
    const bool runs = !gs.grid && gs.run_start && gs.gi && !gs.x && gs.nruns > 0;  // symmetric rules: runs of grid-index nodes
    if (!(gs.grid || runs) || gs.deriv || gs.values || gs.Haos || gs.Eaos || gs.Uplanes.base || !(gs.Hplanes.base || gs.Eplanes.base)) return false;
    if (!lane_shape_ok(gs.n, gs.M, gs.first, gs.npt, gs.herm)) return false;
    const int mode = (gs.Hplanes.base ? 1 : 0) | (gs.Eplanes.base ? 2 : 0);
    return 4 * lane_wave_bytes(gs.n, gs.M, mode, 0) <= 150 * 1024;
}

int launch_lane_grid(abz_ctx* ctx, const GenSpec& gs) {
    LaneArgs a;
    a.src = gs.src;
    a.tab = gs.tab;
    a.H = gs.Hplanes;
    a.E = gs.Eplanes;
    a.nlines = gs.grid ? gs.nnodes / gs.npt : gs.nruns;
    a.npt = gs.npt;
    a.M = gs.M;
    a.first = gs.first;
    a.run_start = gs.grid ? nullptr : gs.run_start;
    a.gi = gs.grid ? nullptr : gs.gi;
    a.sweep = nullptr;
    a.n_sweep = 0;
    a.is_dos = 0;
    a.eta = 0.0;
    a.partial = nullptr;
    const int mode = (gs.Hplanes.base ? 1 : 0) | (gs.Eplanes.base ? 2 : 0);
    const size_t lds = 4 * lane_wave_bytes(gs.n, gs.M, mode, 0);
    const int64_t units = a.nlines * ((gs.npt + 63) / 64);
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((units + 3) / 4, 256 * 8));
    ProfScope ps(ctx, ABZ_K_EVAL);
    return lane_launch(ctx, gs.n, a, mode, lds, blocks);
}

bool lane_sum_supported(int n, int M, int first, int npt, int integrand, int n_sweep) {
    if (!(integrand == ABZ_F_DOS || integrand == ABZ_F_TRGLOC)) return false;
    if (!lane_shape_ok(n, M, first, npt, true)) return false;  // (store-free sums exist for Hermitian series only)
    return 4 * lane_wave_bytes(n, M, 4, n_sweep) <= 150 * 1024;
}

int launch_lane_sum(abz_ctx* ctx, const SumSpec& ss, double* out_reim) {
    const int64_t units = ss.nlines * ((ss.npt + 63) / 64);
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((units + 3) / 4, 256 * 8));
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * 4 * ss.n_sweep));
    if (rc) return rc;
    if ((rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)ss.n_sweep))) return rc;
    if ((rc = ctx->scratch[5].reserve(sizeof(double) * (size_t)ss.n_sweep))) return rc;
    double* sw = ctx->scratch[5].as<double>();
    ABZ_HIP(hipMemcpyAsync(sw, ss.sweep_host, sizeof(double) * (size_t)ss.n_sweep, hipMemcpyHostToDevice, ctx->stream));
    LaneArgs a;
    a.src = ss.src;
    a.tab = ss.tab;
    a.nlines = ss.nlines;
    a.npt = ss.npt;
    a.M = ss.M;
    a.first = ss.first;
    a.run_start = nullptr;
    a.gi = nullptr;
    a.sweep = sw;
    a.n_sweep = ss.n_sweep;
    a.is_dos = ss.integrand == ABZ_F_DOS ? 1 : 0;
    a.eta = ss.params[0];
    a.partial = ctx->scratch[1].as<double2>();
    const size_t lds = 4 * lane_wave_bytes(ss.n, ss.M, 4, ss.n_sweep);
    {
        ProfScope ps(ctx, ABZ_K_EVAL);
        if ((rc = lane_launch(ctx, ss.n, a, 4, lds, blocks))) return rc;
        if ((rc = launch_final_reduce(ctx, a.partial, blocks * 4, ss.n_sweep, ss.scale, ctx->scratch[2].as<double2>()))) return rc;
    }
    ABZ_HIP(hipMemcpyAsync(out_reim, ctx->scratch[2].p, sizeof(double2) * (size_t)ss.n_sweep, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    return ABZ_OK;
}

bool lane_scan_supported(const ReduceSpec& rs) {
    return abz_switch(SW_LANE_KERNELS) != 0 && rs.herm && rs.H.base && rs.n >= 5 && rs.n <= 8 && rs.sweep_dev && rs.n_sweep >= 1 &&
           (rs.integrand == ABZ_F_DOS || rs.integrand == ABZ_F_TRGLOC) && sizeof(double2) * 4 * (size_t)rs.n_sweep <= 60 * 1024;
}

int launch_lane_scan(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim) {
    const int64_t units = (rs.nk + 63) / 64;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((units + 3) / 4, 256 * 8));
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * 4 * rs.n_sweep));
    if (rc) return rc;
    if ((rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)rs.n_sweep))) return rc;
    LaneScanArgs a;
    a.H = rs.H;
    a.w = rs.w;
    a.nk = rs.nk;
    a.sweep = rs.sweep_dev;
    a.n_sweep = rs.n_sweep;
    a.is_dos = rs.integrand == ABZ_F_DOS ? 1 : 0;
    a.eta = rs.params[0];
    a.partial = ctx->scratch[1].as<double2>();
    const size_t lds = sizeof(double2) * 4 * (size_t)rs.n_sweep;
    double2* outd = ctx->scratch[2].as<double2>();
    {
        ProfScope ps(ctx, ABZ_K_REDUCE);
        switch (rs.n) {
            case 5: hipLaunchKernelGGL(lane_scan_kernel<5>, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); break;
            case 6: hipLaunchKernelGGL(lane_scan_kernel<6>, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); break;
            case 7: hipLaunchKernelGGL(lane_scan_kernel<7>, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); break;
            default: hipLaunchKernelGGL(lane_scan_kernel<8>, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); break;
        }
        ABZ_HIP(hipGetLastError());
        if ((rc = launch_final_reduce(ctx, a.partial, blocks * 4, rs.n_sweep, rs.scale, outd))) return rc;
    }
    if (rs.out_dev) {
        ABZ_HIP(hipMemcpyAsync(rs.out_dev, outd, sizeof(double2) * (size_t)rs.n_sweep, hipMemcpyDeviceToDevice, ctx->stream));
        return ABZ_OK;
    }
    ABZ_HIP(hipMemcpyAsync(out_reim, outd, sizeof(double2) * (size_t)rs.n_sweep, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    return ABZ_OK;
}

}  // namespace abz
