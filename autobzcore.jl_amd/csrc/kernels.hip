// HIP kernels of the hot path, written for gfx950 (MI355X, wave64) only.
//
//   phase_kernel            e^{2 pi i f x / t} tables for a batch of nodes                     (tiny)
//   contract_grid_s_kernel  full-grid contraction of the outermost remaining variable, phases as scalars
//   contract_grid_kernel    the same with LDS-staged phases (derivative builds); contract_kernel: node lists
//                           = workspace_contract!  (ref src/fourier.jl:152,158,242,252,468,478)
//   eval_grid_kernel        innermost 1-D series on a full PTR grid: one wavefront per pass of a line (i2,i3),
//                           KPL nodes per lane, the line's coefficients in a wave-private double-buffered LDS
//                           slot, one load group and one wait per unit of work (in-order vmcnt), fused
//                           Hermitian eigensolve, tiled-planar stores (H first, non-temporal on large rules)
//                           = workspace_evaluate! in fourier_ptr!  (ref src/fourier.jl:132-147)
//   eval_node_kernel        the series at explicit nodes (symmetric rules, abz_eval_nodes)
//   reduce_kernel           sum_k w_k f(H(k); omega_i) for all omega_i of a sweep in one pass over the cached
//                           rule = quadsum (ref src/fourier.jl:204-207,289-292); Hermitian rules: real
//                           characteristic polynomial / adjugate forms
//   ggr_kernel              GGR formula scan (ref src/dos_ggr.jl:58-104)
//   node_integrand_kernel   IAI innermost nodes (n <= 4); inner_adaptive_kernel: whole innermost adaptive
//                           loops on the device, half a wavefront per 1-D integral
//   (generic n: kernels_generic.hip; symmetric-rule tables: kernels_symptr.hip)
#include "abz_internal.h"
#include "device_math.h"
#include "gk15.h"
#include "inner_adapt.h"
#include "packed_herm.h"

#include <cmath>
#include <cstring>
#include <type_traits>
#include <utility>
#include <cstdlib>

namespace abz {

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct cplx_pod {
    double x, y;
};
typedef const cplx_pod __attribute__((address_space(4))) * cptr_t;
__device__ __forceinline__ cptr_t as_const(const double2* p) {
    return (cptr_t)(unsigned long long)p;
}

// ------------------------------------------------------------------------------------------
// phases
// ------------------------------------------------------------------------------------------
struct PhaseArgs {
    int64_t B;
    int M, first, npt, deriv;
    int g0, gcnt;  // implicit grid index of item b: g0 + b % gcnt
    const int32_t* gi;
    const double* x;
    const double2* tab;
    double inv_period;
};

__global__ void phase_kernel(PhaseArgs a, double2* __restrict__ phs) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.B * a.M) return;
    const int64_t b = t / a.M;
    const int m = (int)(t - b * a.M);
    const int f = a.first + m;
    double c, s;
    if (a.x == nullptr) {
        const int64_t gi = a.gi ? (int64_t)a.gi[b] : (a.g0 + b % a.gcnt);
        int64_t fm = f % a.npt;
        if (fm < 0) fm += a.npt;
        const double2 ph = a.tab[(fm * gi) % a.npt];
        c = ph.x;
        s = ph.y;
    } else {
        sincospi(2.0 * ((double)f * a.x[b] * a.inv_period), &s, &c);
    }
    if (a.deriv) {
        const double w = 6.283185307179586476925286766559 * (double)f;
        const double c2 = -w * s, s2 = w * c;
        c = c2;
        s = s2;
    }
    phs[t] = make_double2(c, s);
}

int make_phase_table(abz_ctx* ctx, int npt, DevBuf& buf) {
    int rc0 = buf.reserve(sizeof(double2) * (size_t)npt);
    if (rc0) return rc0;
    for (auto& e : ctx->phase_cache) {
        if (e.first == npt && e.second.p) {  // same stream: ordered before every launch that reads `buf`
            ABZ_HIP(hipMemcpyAsync(buf.p, e.second.p, sizeof(double2) * (size_t)npt, hipMemcpyDeviceToDevice, ctx->stream));
            return ABZ_OK;
        }
    }
    std::vector<double2> tab(npt);
    const long double twopi = 6.283185307179586476925286766559005768L;
    for (int i = 0; i < npt; ++i) {
        // reduce to the first octant-ish range for accuracy: angle = 2 pi i / npt
        long double ang = twopi * (long double)i / (long double)npt;
        tab[i] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    int rc = buf.reserve(sizeof(double2) * (size_t)npt);
    if (rc) return rc;
    ABZ_HIP(hipMemcpyAsync(buf.p, tab.data(), sizeof(double2) * (size_t)npt, hipMemcpyHostToDevice, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));  // tab is a stack-lifetime host vector
    if (ctx->phase_cache.size() < 64 && npt <= (1 << 16)) {  // a handful of grids per context in practice
        ctx->phase_cache.emplace_back(npt, DevBuf());
        DevBuf& c = ctx->phase_cache.back().second;
        if (c.reserve(sizeof(double2) * (size_t)npt) == ABZ_OK)
            ABZ_HIP(hipMemcpyAsync(c.p, buf.p, sizeof(double2) * (size_t)npt, hipMemcpyDeviceToDevice, ctx->stream));
        else
            ctx->phase_cache.pop_back();
    }
    return ABZ_OK;
}

int launch_phases(abz_ctx* ctx, const PhaseSpec& ps, double2* phs) {
    PhaseArgs a;
    a.B = ps.B;
    a.M = ps.M;
    a.first = ps.first;
    a.npt = ps.npt > 0 ? ps.npt : 1;
    a.deriv = ps.deriv ? 1 : 0;
    a.gi = ps.gi;
    a.g0 = ps.g0;
    a.gcnt = ps.gcnt > 0 ? ps.gcnt : a.npt;
    a.x = ps.x;
    a.tab = ps.tab;
    a.inv_period = 1.0 / ps.period;
    const int64_t total = ps.B * ps.M;
    if (total == 0) return ABZ_OK;
    hipLaunchKernelGGL(phase_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, ctx->stream, a, phs);
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// contraction of the outermost remaining dimension
// ------------------------------------------------------------------------------------------
__global__ void contract_kernel(const double2* __restrict__ src, int64_t slot_elems,
                                const int64_t* __restrict__ parents, int64_t per_parent,
                                const double2* __restrict__ phs, double2* __restrict__ out, int64_t B,
                                int64_t L, int M) {
    const int64_t b = blockIdx.x;
    const int64_t l = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const int64_t parent = parents ? parents[b] : (b / per_parent);
    const double2* __restrict__ s = src + parent * slot_elems + l;
    const double2* __restrict__ p = phs + b * M;
    double ar = 0.0, ai = 0.0;
    for (int m = 0; m < M; ++m) {
        const double2 c = s[(int64_t)m * L];
        const double2 ph = p[m];
        ar = fma(c.x, ph.x, ar);
        ar = fma(-c.y, ph.y, ar);
        ai = fma(c.x, ph.y, ai);
        ai = fma(c.y, ph.x, ai);
    }
    out[b * L + l] = make_double2(ar, ai);
}

int launch_contract(abz_ctx* ctx, const double2* src, int64_t src_slot_elems, const int64_t* parents,
                    int64_t per_parent, const double2* phs, double2* out, int64_t B, int64_t L, int M) {
    if (B == 0 || L == 0) return ABZ_OK;
    ProfScope ps(ctx, ABZ_K_CONTRACT);
    const int bs = L <= 64 ? 64 : (L <= 128 ? 128 : 256);
    const int64_t gy = cdiv(L, bs);
    if (gy > 65535) {
        set_error("contract: row length %lld too large", (long long)L);
        return ABZ_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(contract_kernel, dim3((unsigned)B, (unsigned)gy), dim3(bs), 0, ctx->stream, src,
                       src_slot_elems, parents, per_parent > 0 ? per_parent : 1, phs, out, B, L, M);
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// Full-grid contraction: out[(parent*npt + gi)][l] = sum_m tab[(f_m gi) mod npt] C[parent][m][l].
// A thread keeps the M coefficients of its column l in registers and sweeps a chunk of grid indices
// gi, so every coefficient is fetched once per (parent, chunk) instead of once per output; phases are
// exact table look-ups whose index advances by f_m per step (no phase kernel, no modulo).
constexpr int CONTRACT_MAXM = 16;

__global__ __launch_bounds__(128) void contract_grid_kernel(const double2* __restrict__ src, int64_t slot_elems,
                                                            const double2* __restrict__ tab, double2* __restrict__ out,
                                                            int64_t L, int M, int first, int npt, int chunk, int deriv,
                                                            int gbeg, int gcnt) {
    extern __shared__ double2 phs[];  // [chunk][M] phases of this block's grid indices
    // grid indices gbeg .. gbeg + gcnt - 1 of the contracted variable (a slab of the outermost
    // variable, or the whole range); output item = parent * gcnt + (gi - gbeg)
    const int g0 = gbeg + blockIdx.z * chunk;
    const int g1 = min(gbeg + gcnt, g0 + chunk);
    for (int t = threadIdx.x; t < (g1 - g0) * M; t += 128) {
        const int gi = g0 + t / M, m = t % M;
        int fm = (first + m) % npt;
        if (fm < 0) fm += npt;
        double2 ph = tab[(int)(((int64_t)fm * gi) % npt)];
        if (deriv) {
            const double f = 6.283185307179586476925286766559 * (double)(first + m);
            ph = make_double2(-f * ph.y, f * ph.x);
        }
        phs[t] = ph;
    }
    __syncthreads();
    const int64_t l = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (l >= L) return;
    const int64_t parent = blockIdx.y;
    double2 c[CONTRACT_MAXM];
#pragma unroll
    for (int m = 0; m < CONTRACT_MAXM; ++m)
        if (m < M) c[m] = src[parent * slot_elems + (int64_t)m * L + l];
    for (int gi = g0; gi < g1; ++gi) {
        const double2* __restrict__ p = phs + (gi - g0) * M;
        double ar = 0.0, ai = 0.0;
#pragma unroll
        for (int m = 0; m < CONTRACT_MAXM; ++m) {
            if (m < M) {
                const double2 ph = p[m];  // LDS broadcast
                ar = fma(c[m].x, ph.x, ar);
                ar = fma(-c[m].y, ph.y, ar);
                ai = fma(c[m].x, ph.y, ai);
                ai = fma(c[m].y, ph.x, ai);
            }
        }
        out[(parent * gcnt + (gi - gbeg)) * L + l] = make_double2(ar, ai);
    }
}

// The same with the phases read as SCALARS from a per-level table phs[gi][m] (global memory, wave-uniform
// address -> s_load through the scalar cache, SGPR operands of v_fma_f64).  The LDS-phase kernel above is
// bound by LDS bandwidth: a broadcast ds_read_b128 still costs 64 lanes x 16 B of the LDS pipe for 4 FMAs
// per lane (18.8 us for the 36 MB level-1 sets of the 150^3 SVO grid); here the phases cost no vector or
// LDS cycles at all.
template <int M>  // exact coefficient count: straight-line loads, no per-m branches
__global__ __launch_bounds__(128) void contract_grid_s_kernel(const double2* __restrict__ src, int64_t slot_elems,
                                                              const double2* __restrict__ phs, double2* __restrict__ out,
                                                              int64_t L, int chunk, int gbeg, int gcnt) {
    const int64_t l = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (l >= L) return;
    const int64_t parent = blockIdx.y;
    const int g0 = gbeg + blockIdx.z * chunk;
    const int g1 = min(gbeg + gcnt, g0 + chunk);
    double2 c[M];
#pragma unroll
    for (int m = 0; m < M; ++m) c[m] = src[parent * slot_elems + (int64_t)m * L + l];
    for (int gi = g0; gi < g1; ++gi) {
        cptr_t p = as_const(phs + (int64_t)gi * M);
        double ar = 0.0, ai = 0.0;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const double px = p[m].x, py = p[m].y;  // scalar loads
            ar = fma(c[m].x, px, ar);
            ar = fma(-c[m].y, py, ar);
            ai = fma(c[m].x, py, ai);
            ai = fma(c[m].y, px, ai);
        }
        out[(parent * gcnt + (gi - gbeg)) * L + l] = make_double2(ar, ai);
    }
}

int launch_contract_grid(abz_ctx* ctx, const double2* src, int64_t src_slot_elems, int64_t nparents, const double2* tab,
                         double2* out, int64_t L, int M, int first, int npt, bool deriv, int gbeg, int gcnt,
                         const double2* phs_table) {
    if (nparents == 0 || L == 0 || gcnt == 0) return ABZ_OK;
    ProfScope ps(ctx, ABZ_K_CONTRACT);
    const int64_t gx = cdiv(L, 128);
    // enough blocks to fill the chip, at least ~4 grid indices per thread to amortise the loads
    const int target_blocks = 4096;
    int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>(cdiv(gcnt, 4), cdiv(target_blocks, gx * nparents)));
    const int chunk = (int)cdiv(gcnt, nsplit);
    nsplit = cdiv(gcnt, chunk);
    if (nparents > 65535 || nsplit > 65535) {
        set_error("contract_grid: grid too large");
        return ABZ_ERR_UNSUPPORTED;
    }
    if (phs_table && !deriv) {
#define CS(MM)                                                                                                         \
    case MM:                                                                                                           \
        hipLaunchKernelGGL(contract_grid_s_kernel<MM>, dim3((unsigned)gx, (unsigned)nparents, (unsigned)nsplit), dim3(128), 0, \
                           ctx->stream, src, src_slot_elems, phs_table, out, L, chunk, gbeg, gcnt);                    \
        break;
        switch (M) {
            CS(1) CS(2) CS(3) CS(4) CS(5) CS(6) CS(7) CS(8) CS(9) CS(10) CS(11) CS(12) CS(13) CS(14) CS(15) CS(16)
            default: set_error("contract_grid: M = %d", M); return ABZ_ERR_UNSUPPORTED;
        }
#undef CS
        ABZ_HIP(hipGetLastError());
        return ABZ_OK;
    }
    hipLaunchKernelGGL(contract_grid_kernel, dim3((unsigned)gx, (unsigned)nparents, (unsigned)nsplit), dim3(128),
                       sizeof(double2) * (size_t)chunk * M, ctx->stream, src, src_slot_elems, tab, out, L, M, first, npt,
                       chunk, deriv ? 1 : 0, gbeg, gcnt);
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// innermost evaluation
// ------------------------------------------------------------------------------------------
// Constant-address-space view of read-only coefficient sets: a wave-uniform address in this address
// space is fetched through the scalar cache (s_load_dwordx4/x8) and reaches v_fma_f64 as an SGPR
// operand; a divergent address still becomes an ordinary global_load.  Legal because the sets are
// written by an EARLIER kernel on the same stream and never inside the kernel that reads them.
// (cplx_pod / cptr_t / as_const are defined at the top of this file)

// H = sum_m c1[m] * (w z^m)  (optionally times i 2 pi f_m).
template <int N>
__device__ __forceinline__ void series_lane(cptr_t c1, int M, int first, double zr,
                                            double zi, double wr, double wi, bool deriv, CMat<N>& H) {
#pragma unroll
    for (int a = 0; a < N; ++a) {
#pragma unroll
        for (int b = 0; b < N; ++b) {
            H.re[a][b] = 0.0;
            H.im[a][b] = 0.0;
        }
    }
    double pr = wr, pi = wi;
    for (int m = 0; m < M; ++m) {
        double qr = pr, qi = pi;
        if (deriv) {
            const double f = 6.283185307179586476925286766559 * (double)(first + m);
            qr = -f * pi;
            qi = f * pr;
        }
        cptr_t cm = c1 + (int64_t)m * (N * N);
#pragma unroll
        for (int b = 0; b < N; ++b) {
#pragma unroll
            for (int a = 0; a < N; ++a) {
                const double cx = cm[a + N * b].x, cy = cm[a + N * b].y;  // column-major block
                H.re[a][b] = fma(cx, qr, H.re[a][b]);
                H.re[a][b] = fma(-cy, qi, H.re[a][b]);
                H.im[a][b] = fma(cx, qi, H.im[a][b]);
                H.im[a][b] = fma(cy, qr, H.im[a][b]);
            }
        }
        const double nr = pr * zr - pi * zi;
        const double ni = pr * zi + pi * zr;
        pr = nr;
        pi = ni;
    }
}

// offset of (node k, plane 0) in a tiled planar view
__device__ __forceinline__ int64_t view_off(const PlaneView& v, int64_t k) {
    const int64_t line = k / v.line_len;
    return line * v.tile + (k - line * v.line_len);
}

// Plane of Re H[a][b], a <= b, in a view (Im is the next plane).  Full layout, the reference's SMatrix order:
// 2 (a + N b).  Hermitian-compact layout (PlaneView::compact, ABZ_WANT_H_COMPACT in abzhip.h): the upper triangle column
// by column, b^2 + 2 a; the real diagonal entry H[b][b] is the single plane b^2 + 2 b.
template <int N>
__device__ __forceinline__ int hplane(const PlaneView& v, int a, int b) {
    return v.compact ? b * b + 2 * a : 2 * (a + N * b);
}

template <int N>
__device__ __forceinline__ void store_planes(const CMat<N>& H, const PlaneView& v, int64_t off) {
    double* __restrict__ out = v.base + off;
    if (v.compact) {
#pragma unroll
        for (int b = 0; b < N; ++b) {
#pragma unroll
            for (int a = 0; a <= b; ++a) {
                out[(int64_t)(b * b + 2 * a) * v.pitch] = H.re[a][b];
                if (a < b) out[(int64_t)(b * b + 2 * a + 1) * v.pitch] = H.im[a][b];
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < N; ++b) {
#pragma unroll
        for (int a = 0; a < N; ++a) {
            out[(int64_t)(2 * (a + N * b)) * v.pitch] = H.re[a][b];
            out[(int64_t)(2 * (a + N * b) + 1) * v.pitch] = H.im[a][b];
        }
    }
}

// Hermitian values: the upper triangle's planes only (n^2 loads instead of 2 n^2), the rest mirrored in registers
template <int N>
__device__ __forceinline__ void load_planes_herm(CMat<N>& H, const PlaneView& v, int64_t off) {
    const double* __restrict__ in = v.base + off;
#pragma unroll
    for (int b = 0; b < N; ++b) {
#pragma unroll
        for (int a = 0; a <= b; ++a) {
            const int pl = hplane<N>(v, a, b);
            H.re[a][b] = in[(int64_t)pl * v.pitch];
            H.im[a][b] = (a == b) ? 0.0 : in[(int64_t)(pl + 1) * v.pitch];
        }
    }
#pragma unroll
    for (int b = 0; b < N; ++b) {
#pragma unroll
        for (int a = b + 1; a < N; ++a) {
            H.re[a][b] = H.re[b][a];
            H.im[a][b] = -H.im[b][a];
        }
    }
}

template <int N>
__device__ __forceinline__ void load_planes(CMat<N>& H, const PlaneView& v, int64_t off) {
    if (v.compact) {  // only the upper triangle exists
        load_planes_herm<N>(H, v, off);
        return;
    }
    const double* __restrict__ in = v.base + off;
#pragma unroll
    for (int b = 0; b < N; ++b) {
#pragma unroll
        for (int a = 0; a < N; ++a) {
            H.re[a][b] = in[(int64_t)(2 * (a + N * b)) * v.pitch];
            H.im[a][b] = in[(int64_t)(2 * (a + N * b) + 1) * v.pitch];
        }
    }
}

struct EvalArgs {
    const double2* src;
    const double2* tab;
    const int64_t* parents;
    const int32_t* gi;
    const double* x;
    PlaneView H, E, U;
    int64_t nlines, nk;
    int64_t line0;  // store-free sums on a slab: global index of line 0 (for node coordinates)
    int M, first, npt, deriv, herm;
    int nt;  // non-temporal stores (rule values larger than the Infinity Cache)
    int pk;    // the level-1 sets are PACKED Hermitian sets (packed_herm.h): Pk<N>::size((M - 1) / 2) numbers per line
    double inv_period;
};

// planes of one node at column i1 of tile `line`; with a wave-uniform line every plane row is a scalar
// base and the lane offset i1 is the same 32-bit register for all of them.  NT: non-temporal stores
// (streaming: the values are not re-read by this kernel and a large rule does not fit the caches anyway).
template <bool NT>
__device__ __forceinline__ void store_f64(double* p, double v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}
template <int N, bool NT = false, bool HC = false>
__device__ __forceinline__ void store_planes_at(const CMat<N>& H, const PlaneView& v, int64_t line, int i1) {
    double* __restrict__ row = v.base + line * v.tile;
    const unsigned u = (unsigned)i1;
    if constexpr (HC) {  // Hermitian-compact planes: the upper triangle
#pragma unroll
        for (int b = 0; b < N; ++b) {
#pragma unroll
            for (int a = 0; a <= b; ++a) {
                store_f64<NT>((row + (int64_t)(b * b + 2 * a) * v.pitch) + u, H.re[a][b]);
                if (a < b) store_f64<NT>((row + (int64_t)(b * b + 2 * a + 1) * v.pitch) + u, H.im[a][b]);
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < N; ++b) {
#pragma unroll
            for (int a = 0; a < N; ++a) {
                store_f64<NT>((row + (int64_t)(2 * (a + N * b)) * v.pitch) + u, H.re[a][b]);
                store_f64<NT>((row + (int64_t)(2 * (a + N * b) + 1) * v.pitch) + u, H.im[a][b]);
            }
        }
    }
}

template <int N, bool VEC = true>
__device__ __forceinline__ void eval_epilogue(const EvalArgs& a, CMat<N>& H, int64_t line, int i1) {
    if (a.H.base) {
        if (a.H.compact)
            store_planes_at<N, false, true>(H, a.H, line, i1);
        else
            store_planes_at<N>(H, a.H, line, i1);
    }
    if (a.E.base || (VEC && a.U.base)) {
        double e[N];
        CMat<N> V;
        if (VEC && a.U.base) {
            herm_eig<N, true>(H, e, V);
            store_planes_at<N>(V, a.U, line, i1);
        } else {
            herm_eig_values<N>(H, e);
        }
        if (a.E.base) {
            double* __restrict__ row = a.E.base + line * a.E.tile;
            const unsigned u = (unsigned)i1;
#pragma unroll
            for (int b = 0; b < N; ++b) (row + (int64_t)b * a.E.pitch)[u] = e[b];
        }
    }
}

// One wavefront per line (i2,i3); each lane owns KPL nodes i1 = i0 + lane + 64 j of that line.
// The line's coefficients c1[M][N*N] are staged in a wave-private LDS buffer (double-buffered: the
// next line's set is fetched into registers while the current one is consumed), read back with
// broadcast ds_read_b128 and reused for the KPL nodes of every lane; phases w z^m by recurrence.
// No block-level barrier: a wave only ever reads the LDS bytes it wrote itself.
constexpr int EVAL_MAX_MNN = 256;  // complex coefficients per line held in LDS (M * N * N)

// One unit of work of the grid kernels: pass `pass` (64 KPL nodes starting at i0) of line `line`, from the
// line's coefficients c1 (LDS).  `mid` runs between the m-loop and the stores: the place where the next
// unit's coefficients are handed to the other LDS buffer.
template <int N, int KPL, bool HERM, class MID, bool PK = false>
__device__ __forceinline__ void eval_unit_core(const EvalArgs& a, const double2* __restrict__ c1, const double2* tab_l, int fm,
                                               const int (&iz0)[KPL], const int (&iw0)[KPL], int npass, int i0, int lane,
                                               MID&& mid, CMat<N> (&H)[KPL]) {
    if constexpr (PK) {
        // packed Hermitian set (packed_herm.h): +f and -f in one FMA group, phases z^f from z alone -- half the Fourier
        // work of the loop over 2 F + 1 terms below
        static_assert(HERM, "packed sets exist for Hermitian series only");
        const int F = (a.M - 1) / 2;
        double zr[KPL], zi[KPL], pr[KPL], pi[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            int ic = iz0[j];
            if (npass > 1) {
                const int i1 = i0 + lane + 64 * j;
                ic = i1 < a.npt ? i1 : 0;
            }
            const double2 z = tab_l[ic];
            zr[j] = z.x;
            zi[j] = z.y;
            pr[j] = 1.0;
            pi[j] = 0.0;
        }
#pragma unroll
        for (int bb = 0; bb < N; ++bb) {
#pragma unroll
            for (int aa = 0; aa <= bb; ++aa) {
                const double2 c = c1[Pk<N>::tri(aa, bb)];
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    H[j].re[aa][bb] = c.x;
                    H[j].im[aa][bb] = (aa == bb) ? 0.0 : c.y;
                }
            }
        }
        for (int f = 1; f <= F; ++f) {
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const double nr = pr[j] * zr[j] - pi[j] * zi[j];
                const double ni = pr[j] * zi[j] + pi[j] * zr[j];
                pr[j] = nr;
                pi[j] = ni;
            }
            const double2* __restrict__ cf = c1 + Pk<N>::blk(1) + (f - 1) * (N * N);
#pragma unroll
            for (int aa = 0; aa < N; ++aa) {
                const double2 dd = cf[aa];
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    H[j].re[aa][aa] = fma(dd.x, pr[j], H[j].re[aa][aa]);
                    H[j].re[aa][aa] = fma(-dd.y, pi[j], H[j].re[aa][aa]);
                }
            }
#pragma unroll
            for (int bb = 1; bb < N; ++bb) {
#pragma unroll
                for (int aa = 0; aa < bb; ++aa) {
                    const double2 sv = cf[N + 2 * Pk<N>::pair(aa, bb)];
                    const double2 tv = cf[N + 2 * Pk<N>::pair(aa, bb) + 1];
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        H[j].re[aa][bb] = fma(sv.x, pr[j], H[j].re[aa][bb]);
                        H[j].re[aa][bb] = fma(-sv.y, pi[j], H[j].re[aa][bb]);
                        H[j].im[aa][bb] = fma(tv.x, pi[j], H[j].im[aa][bb]);
                        H[j].im[aa][bb] = fma(tv.y, pr[j], H[j].im[aa][bb]);
                    }
                }
            }
        }
        mid();
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
#pragma unroll
            for (int bb = 0; bb < N; ++bb) {
#pragma unroll
                for (int aa = bb + 1; aa < N; ++aa) {
                    H[j].re[aa][bb] = H[j].re[bb][aa];
                    H[j].im[aa][bb] = -H[j].im[bb][aa];
                }
            }
        }
        return;
    }
    double zr[KPL], zi[KPL], pr[KPL], pi[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        int ic = iz0[j], iw = iw0[j];
        if (npass > 1) {
            const int i1 = i0 + lane + 64 * j;
            ic = i1 < a.npt ? i1 : 0;
            iw = (int)(((unsigned)fm * (unsigned)ic) % (unsigned)a.npt);
        }
        const double2 z = tab_l[ic];
        const double2 w = tab_l[iw];
        zr[j] = z.x;
        zi[j] = z.y;
        pr[j] = w.x;
        pi[j] = w.y;
#pragma unroll
        for (int aa = 0; aa < N; ++aa) {
#pragma unroll
            for (int bb = 0; bb < N; ++bb) {
                H[j].re[aa][bb] = 0.0;
                H[j].im[aa][bb] = 0.0;
            }
        }
    }
    for (int m = 0; m < a.M; ++m) {
        const double2* __restrict__ cm = c1 + m * (N * N);
#pragma unroll
        for (int bb = 0; bb < N; ++bb) {
#pragma unroll
            for (int aa = 0; aa < N; ++aa) {
                if (HERM && aa > bb) continue;  // upper triangle only; mirrored below
                const double2 c = cm[aa + N * bb];
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    H[j].re[aa][bb] = fma(c.x, pr[j], H[j].re[aa][bb]);
                    H[j].re[aa][bb] = fma(-c.y, pi[j], H[j].re[aa][bb]);
                    if (!(HERM && aa == bb)) {
                        H[j].im[aa][bb] = fma(c.x, pi[j], H[j].im[aa][bb]);
                        H[j].im[aa][bb] = fma(c.y, pr[j], H[j].im[aa][bb]);
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
    mid();
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        if constexpr (HERM) {
#pragma unroll
            for (int bb = 0; bb < N; ++bb) {
#pragma unroll
                for (int aa = bb + 1; aa < N; ++aa) {
                    H[j].re[aa][bb] = H[j].re[bb][aa];
                    H[j].im[aa][bb] = -H[j].im[bb][aa];
                }
            }
        }
    }
}

// The store epilogue of a unit: values of its 64 KPL nodes into the tiled-planar rule arrays.
template <int N, int KPL, bool VEC>
__device__ __forceinline__ void eval_unit_store(const EvalArgs& a, CMat<N> (&H)[KPL], int i0, int lane, int64_t line) {
    if constexpr (VEC) {
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int i1 = i0 + lane + 64 * j;
            const int pitch = a.H.base ? a.H.row : a.E.row;
            if (i1 < pitch) eval_epilogue<N, VEC>(a, H[j], line, i1);
        }
    }
    if constexpr (!VEC) {
        // columns npt..pitch-1 are padding: written (finite filler) so every 128-B line of the
        // tile leaves the CU whole; never read back (leaving them unwritten measured 17 % slower: partial lines).
        // Order: ALL H(k) stores of the unit first, then the eigensolves (they overlap the drain of those
        // stores), then the eigenvalue stores.  With non-temporal stores on rules larger than the Infinity
        // Cache this is worth 19 % at 150^3 (neither change alone is: the interleaved order leaves the
        // store queue empty during every eigensolve, and temporal stores make the 605 MB fight for L2/MALL).
        const int pitch = a.H.base ? a.H.row : a.E.row;
        auto epilogue = [&](auto nt, auto hc) {
            constexpr bool NT = decltype(nt)::value;
            constexpr bool HC = decltype(hc)::value;
            if (a.H.base) {
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int i1 = i0 + lane + 64 * j;
                    if (i1 < pitch) store_planes_at<N, NT, HC>(H[j], a.H, line, i1);
                }
            }
            if (a.E.base) {
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const int i1 = i0 + lane + 64 * j;
                    if (i1 < pitch) {
                        double e[N];
                        herm_eig_values<N>(H[j], e);
                        double* __restrict__ row = a.E.base + line * a.E.tile;
                        const unsigned u = (unsigned)i1;
#pragma unroll
                        for (int b = 0; b < N; ++b) store_f64<NT>((row + (int64_t)b * a.E.pitch) + u, e[b]);
                    }
                }
            }
        };
        if (a.H.compact) {
            if (a.nt)
                epilogue(std::true_type{}, std::true_type{});
            else
                epilogue(std::false_type{}, std::true_type{});
        } else if (a.nt)
            epilogue(std::true_type{}, std::false_type{});
        else
            epilogue(std::false_type{}, std::false_type{});
    }
}

template <int N, int KPL, bool HERM, bool VEC, class MID, bool PK = false>
__device__ __forceinline__ void eval_unit(const EvalArgs& a, const double2* __restrict__ c1, const double2* tab_l, int fm,
                                          const int (&iz0)[KPL], const int (&iw0)[KPL], int npass, int i0, int lane,
                                          int64_t line, MID&& mid) {
    CMat<N> H[KPL];
    eval_unit_core<N, KPL, HERM, MID&, PK>(a, c1, tab_l, fm, iz0, iw0, npass, i0, lane, mid, H);
    eval_unit_store<N, KPL, VEC>(a, H, i0, lane, line);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int N, int KPL, bool HERM, bool VEC, int OCC, bool PK = false>
__global__ __launch_bounds__(256, OCC) void eval_grid_kernel(EvalArgs a) {
    extern __shared__ double2 lds_c[];  // [4 waves][2 buffers][MNN]
    // the wave index is made an SGPR value: every per-line quantity (tile base, coefficient row) is
    // then scalar arithmetic
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int MNN = PK ? Pk<N>::size((a.M - 1) / 2) : a.M * N * N;  // numbers per line
    // derivative series: coefficient m is scaled by 2 pi i (first + m) once, on its way into LDS
    auto stage = [&](double2 c, int idx) -> double2 {
        if (!a.deriv) return c;
        const double f = 6.283185307179586476925286766559 * (double)(a.first + idx / (N * N));
        return make_double2(-f * c.y, f * c.x);
    };
    double2* const mybuf = lds_c + (size_t)wave * 2 * MNN;
    int fm = a.first % a.npt;
    if (fm < 0) fm += a.npt;
    // Every global load issued inside the work loop shares the in-order vmcnt with the stores, and a
    // wait on any of them is a wait on every older store of the wave.  So the loop body is straight
    // line code with exactly one group of loads (the next unit's coefficients) and one wait, placed
    // right after the m-loop when the previous unit's stores are a whole m-loop old; phase seeds
    // come from an LDS copy of the table.  A unit of work is one pass (64 KPL nodes) of one line.
    const int npass = (a.npt + 64 * KPL - 1) / (64 * KPL);
    double2* const tab_l = lds_c + (size_t)4 * 2 * MNN;  // LDS copy of the phase table, shared by the block
    for (int i = threadIdx.x; i < a.npt; i += 256) tab_l[i] = a.tab[i];
    __syncthreads();
    const int64_t lstride = (int64_t)gridDim.x * 4;
    int64_t line = (int64_t)blockIdx.x * 4 + wave;
    if (line >= a.nlines) return;
    // stage the first line
    {
        const double2* __restrict__ src = a.src + line * MNN;
#pragma unroll
        for (int t = 0; t < EVAL_MAX_MNN / 64; ++t) {
            const int idx = lane + 64 * t;
            if (idx < MNN) mybuf[idx] = stage(src[idx], idx);
        }
    }
    // table indices of z and of the seed w = z^first for the nodes of pass 0
    int iz0[KPL], iw0[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int i1 = lane + 64 * j;
        iz0[j] = i1 < a.npt ? i1 : 0;
        iw0[j] = (int)(((unsigned)fm * (unsigned)iz0[j]) % (unsigned)a.npt);
    }
    int cur = 0, pass = 0;
    while (line < a.nlines) {
        // fetch the next unit's coefficients into registers (the same line again if it has more passes)
        const bool last_pass = pass + 1 >= npass;
        const int64_t nline = last_pass ? line + lstride : line;
        const bool have_next = nline < a.nlines;
        double2 pre[EVAL_MAX_MNN / 64];
        {
            const double2* __restrict__ src = a.src + (have_next ? nline : line) * MNN;
#pragma unroll
            for (int t = 0; t < EVAL_MAX_MNN / 64; ++t) {
                const int idx = lane + 64 * t;
                pre[t] = src[idx < MNN ? idx : MNN - 1];  // unconditional: no select waits on the data
            }
        }
        wave_lds_sync();
        auto mid_fn = [&]() {
            // The one wait of the loop body: the fetched registers are consumed here, unconditionally and
            // before this unit's stores are issued (the asm pins the point; nothing is hoisted above it).
#pragma unroll
            for (int t = 0; t < EVAL_MAX_MNN / 64; ++t) asm volatile("" : "+v"(pre[t].x), "+v"(pre[t].y));
            if (have_next) {
                double2* dst = mybuf + (size_t)(cur ^ 1) * MNN;
#pragma unroll
                for (int t = 0; t < EVAL_MAX_MNN / 64; ++t) {
                    const int idx = lane + 64 * t;
                    if (idx < MNN) dst[idx] = stage(pre[t], idx);
                }
            }
        };
        eval_unit<N, KPL, HERM, VEC, decltype(mid_fn)&, PK>(a, mybuf + (size_t)cur * MNN, tab_l, fm, iz0, iw0, npass, pass * (64 * KPL), lane, line,
                                                            mid_fn);
        cur ^= 1;
        pass = last_pass ? 0 : pass + 1;
        line = nline;
    }
}

// Fallback for coefficient sets too large for the LDS staging above: wave-uniform scalar loads.
template <int N>
__global__ __launch_bounds__(256) void eval_grid_kernel_scalar(EvalArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int fm = a.first % a.npt;
    if (fm < 0) fm += a.npt;
    for (int64_t line = (int64_t)blockIdx.x * 4 + wave; line < a.nlines; line += (int64_t)gridDim.x * 4) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(line & 0xffffffffu));
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)line >> 32));
        const int64_t line_u = (int64_t)(((unsigned long long)hi << 32) | lo);
        cptr_t c1 = as_const(a.src + line_u * ((int64_t)a.M * N * N));
        for (int i0 = 0; i0 < a.npt; i0 += 64) {
            const int i1 = i0 + lane;
            if (i1 < a.npt) {
                const double2 z = a.tab[i1];
                const double2 w = a.tab[(int)(((int64_t)fm * i1) % a.npt)];
                CMat<N> H;
                series_lane<N>(c1, a.M, a.first, z.x, z.y, w.x, w.y, a.deriv != 0, H);
                eval_epilogue<N>(a, H, line_u, i1);
            }
        }
    }
}

template <int N>
__global__ __launch_bounds__(256) void eval_node_kernel(EvalArgs a) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.nk) return;
    const int64_t slot = a.parents ? a.parents[k] : 0;
    cptr_t c1 = as_const(a.src + slot * ((int64_t)a.M * N * N));
    double zr, zi, wr, wi;
    if (a.x == nullptr) {
        const int i1 = a.gi[k];
        int fm = a.first % a.npt;
        if (fm < 0) fm += a.npt;
        const double2 z = a.tab[i1];
        const double2 w = a.tab[(int)(((int64_t)fm * i1) % a.npt)];
        zr = z.x;
        zi = z.y;
        wr = w.x;
        wi = w.y;
    } else {
        const double xx = a.x[k] * a.inv_period;
        sincospi(2.0 * xx, &zi, &zr);
        sincospi(2.0 * ((double)a.first * xx), &wi, &wr);
    }
    CMat<N> H;
    series_lane<N>(c1, a.M, a.first, zr, zi, wr, wi, a.deriv != 0, H);
    const int ll = a.H.base ? a.H.line_len : (a.E.base ? a.E.line_len : a.U.line_len);
    const int64_t line = k / ll;
    eval_epilogue<N>(a, H, line, (int)(k - line * ll));
}

#define ABZ_DISPATCH_N(n, FN)                                       \
    switch (n) {                                                    \
        case 1: FN(1); break;                                       \
        case 2: FN(2); break;                                       \
        case 3: FN(3); break;                                       \
        case 4: FN(4); break;                                       \
        default:                                                    \
            set_error("n = %d bands: only n <= 4 is built in this round", n); \
            return ABZ_ERR_UNSUPPORTED;                             \
    }

// Can the grid kernel take packed Hermitian level-1 sets (packed_herm.h)?  Same limits as its LDS-staged path.
bool eval_packed_supported(int n, int M, int npt) {
    if (!abz_switch(SW_EVAL_PACKED) || n < 1 || n > 4 || (M & 1) == 0) return false;
    const size_t P = packed_row_elems(n, M);
    return P <= (size_t)EVAL_MAX_MNN && npt < 65536 && sizeof(double2) * (4 * 2 * P + (size_t)npt) <= 64 * 1024;
}

int launch_eval(abz_ctx* ctx, const EvalSpec& es) {
    if (es.n > 4) {
        GenSpec gs;
        gs.Uplanes = es.U;
        gs.herm = es.herm && !es.deriv;
        gs.n = es.n;
        gs.M = es.M;
        gs.first = es.first;
        gs.npt = es.npt;
        gs.d = 0;
        gs.period = es.period;
        gs.src = es.src;
        gs.grid = es.grid;
        gs.parents = es.parents;
        gs.run_start = es.run_start;
        gs.nruns = es.nruns;
        gs.gi = es.gi;
        gs.x = es.x;
        gs.tab = es.tab;
        gs.deriv = es.deriv;
        gs.nnodes = es.grid ? es.nlines * es.npt : es.nk;
        gs.Hplanes = es.H;
        gs.Eplanes = es.E;
        gs.Haos = nullptr;
        gs.Eaos = nullptr;
        gs.integrand = ABZ_F_ONE;
        for (int i = 0; i < 4; ++i) gs.params[i] = 0.0;
        gs.sweep_dev = nullptr;
        gs.sweep0 = 0.0;
        gs.n_sweep = 0;
        gs.values = nullptr;
        return launch_gen_nodes(ctx, gs);
    }
    EvalArgs a;
    a.src = es.src;
    a.tab = es.tab;
    a.parents = es.parents;
    a.gi = es.gi;
    a.x = es.x;
    a.H = es.H;
    a.E = es.E;
    a.U = es.U;
    a.nlines = es.nlines;
    a.nk = es.nk;
    a.line0 = 0;
    a.M = es.M;
    a.first = es.first;
    a.npt = es.npt > 0 ? es.npt : 1;
    a.deriv = es.deriv ? 1 : 0;
    a.herm = (es.herm && !es.deriv) ? 1 : 0;
    a.inv_period = 1.0 / es.period;
    {
        // streaming stores once the values written by this launch exceed the 256 MB Infinity Cache
        // (npt = 100, 168 MB: 7 % slower with them; 150, 605 MB: 19 % faster)
        const PlaneView& pv = es.H.base ? es.H : es.E;
        const double planes_out = (es.H.base ? (es.H.compact ? 1.0 : 2.0) * es.n * es.n : 0.0) + (es.E.base ? (double)es.n : 0.0) + (es.U.base ? 2.0 * es.n * es.n : 0.0);
        const double bytes = 8.0 * planes_out * (double)pv.row * (double)(es.grid ? es.nlines : (es.nk + 63) / 64);
        a.nt = bytes > 256.0 * 1024 * 1024 ? 1 : 0;
    }
    a.pk = (es.packed && a.herm && !es.U.base && es.grid) ? 1 : 0;
    ProfScope ps(ctx, ABZ_K_EVAL);
    if (es.grid) {
        if (es.nlines == 0) return ABZ_OK;
        // Workgroups: 8 per CU up to ~140^3, 16 per CU beyond (3 bands: npt 150 0.1133 -> 0.1101 ms on a fast box,
        // 0.1234 -> 0.1175 on a slow one; 200 -6 %, 250 -15 %, 300 -3 %, 500 -6 %; 110 / 128 are 3-5 % better with 8 per
        // CU, 180 and 400 lose 2-3 % with 16); compact H planes: 24 per CU -- at 150^3 one line per wave -- measured
        // 1.5 % better than 16, the same at 200^3
        const int64_t quads = cdiv(es.nlines, 4);
        const int64_t blocks = std::min<int64_t>(quads, quads < 5000 ? 256 * 8 : (es.H.compact ? 256 * 24 : 256 * 16));
        const int mnn = a.pk ? es.n * (es.n + 1) / 2 + ((es.M - 1) / 2) * es.n * es.n : es.M * es.n * es.n;  // numbers per line (packed_herm.h)
        // nodes per lane: minimise lane-rounds per line, ceil(npt / (64 kpl)) * kpl, weighted by the LDS
        // operand reads that are shared by the kpl nodes of a lane (200 points: 2 x 2 rounds, not 2 x 3)
        int kpl = 1;
        {
            double best = 1e30;
            for (int k = 1; k <= 3; ++k) {
                const double cost = (double)(cdiv(es.npt, 64 * k) * k) * (1.0 + 0.3 / k);
                if (cost < best - 1e-12) {
                    best = cost;
                    kpl = k;
                }
            }
        }
        // + an LDS copy of the phase table (fm * i1 < npt^2 must fit 32 bits)
        const size_t lds = sizeof(double2) * (4 * 2 * (size_t)mnn + (size_t)a.npt);
        if (mnn <= EVAL_MAX_MNN && lds <= 64 * 1024 && a.npt < 65536) {
            // waves per SIMD the register allocator is asked for: 3 on the Hermitian 3-band paths (2 and 4 measured
            // slower there), 2 elsewhere
#define LK(NN, KK)                                                                                                             \
    {                                                                                                                          \
        constexpr int OO = NN == 3 ? 3 : 2;                                                                                    \
        if (a.pk)                                                                                                              \
            hipLaunchKernelGGL((eval_grid_kernel<NN, KK, true, false, OO, true>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); \
        else if (a.U.base)                                                                                                     \
            hipLaunchKernelGGL((eval_grid_kernel<NN, KK, false, true, 2>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);  \
        else if (a.herm)                                                                                                       \
            hipLaunchKernelGGL((eval_grid_kernel<NN, KK, true, false, OO>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); \
        else                                                                                                                   \
            hipLaunchKernelGGL((eval_grid_kernel<NN, KK, false, false, 2>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a); \
    }
#define FN(NN)                    \
    switch (kpl) {                \
        case 1: LK(NN, 1) break;  \
        case 2: LK(NN, 2) break;  \
        default: LK(NN, 3) break; \
    }
            ABZ_DISPATCH_N(es.n, FN)
#undef FN
#undef LK
        } else {
#define FN(NN) hipLaunchKernelGGL(eval_grid_kernel_scalar<NN>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a)
            ABZ_DISPATCH_N(es.n, FN)
#undef FN
        }
    } else {
        if (es.nk == 0) return ABZ_OK;
        const int64_t blocks = cdiv(es.nk, 256);
#define FN(NN) hipLaunchKernelGGL(eval_node_kernel<NN>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a)
        ABZ_DISPATCH_N(es.n, FN)
#undef FN
    }
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// stand-alone eigensolve on planes, velocities
// ------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void eig_planes_kernel(PlaneView Hv, PlaneView Ev, PlaneView Uv, int64_t nk) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nk) return;
    CMat<N> H, V;
    load_planes<N>(H, Hv, view_off(Hv, k));
    double e[N];
    if (Uv.base) {
        herm_eig<N, true>(H, e, V);
        store_planes<N>(V, Uv, view_off(Uv, k));
    } else {
        herm_eig_values<N>(H, e);
    }
    double* __restrict__ eo = Ev.base + view_off(Ev, k);
#pragma unroll
    for (int b = 0; b < N; ++b) eo[(int64_t)b * Ev.pitch] = e[b];
}

int launch_eig_planes(abz_ctx* ctx, int n, PlaneView H, PlaneView E, PlaneView U, int64_t nk) {
    if (nk == 0) return ABZ_OK;
    ProfScope ps(ctx, ABZ_K_EIG);
#define FN(NN) hipLaunchKernelGGL(eig_planes_kernel<NN>, dim3((unsigned)cdiv(nk, 256)), dim3(256), 0, ctx->stream, H, E, U, nk)
    ABZ_DISPATCH_N(n, FN)
#undef FN
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

template <int N>
__global__ __launch_bounds__(256) void velocity_kernel(PlaneView Uv, PlaneView Dv, PlaneView Vv, int64_t nk) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nk) return;
    CMat<N> U, D;
    load_planes<N>(U, Uv, view_off(Uv, k));
    load_planes<N>(D, Dv, view_off(Dv, k));
    double* __restrict__ vo = Vv.base + view_off(Vv, k);
#pragma unroll
    for (int b = 0; b < N; ++b) {
        // v_b = Re sum_{a,c} conj(U[a][b]) D[a][c] U[c][b]
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
            double tr = 0.0, ti = 0.0;  // t = sum_c D[a][c] U[c][b]
#pragma unroll
            for (int c = 0; c < N; ++c) {
                tr += D.re[a][c] * U.re[c][b] - D.im[a][c] * U.im[c][b];
                ti += D.re[a][c] * U.im[c][b] + D.im[a][c] * U.re[c][b];
            }
            v += U.re[a][b] * tr + U.im[a][b] * ti;
        }
        vo[(int64_t)b * Vv.pitch] = v;
    }
}

int launch_velocity(abz_ctx* ctx, int n, PlaneView U, PlaneView dH, PlaneView Vj, int64_t nk) {
    if (nk == 0) return ABZ_OK;
    if (n > 4) return launch_gen_velocity(ctx, n, U, dH, Vj, nk);
#define FN(NN) hipLaunchKernelGGL(velocity_kernel<NN>, dim3((unsigned)cdiv(nk, 256)), dim3(256), 0, ctx->stream, U, dH, Vj, nk)
    ABZ_DISPATCH_N(n, FN)
#undef FN
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// integrands
// ------------------------------------------------------------------------------------------
int integrand_ncomp(int integrand, int n, int d) {
    switch (integrand) {
        case ABZ_F_ONE:
        case ABZ_F_LINEAR:
        case ABZ_F_DOS:
        case ABZ_F_TRGLOC:
        case ABZ_F_DOS_EIG: return 1;
        case ABZ_F_LINEAR_X: return d;
        case ABZ_F_GLOC: return n * n;
        default: return -1;
    }
}

constexpr int MAXC = 16;  // max components per integrand value (n <= 4: n*n)

// value of integrand FID at one node; out[c] complex components.
template <int N, int FID>
__device__ __forceinline__ void integrand_value(const CMat<N>& H, const double (&e)[N], const double* xk, int d,
                                                const double* p, double sw, double (&vr)[MAXC], double (&vi)[MAXC]) {
    if constexpr (FID == ABZ_F_ONE) {
        vr[0] = 1.0;
        vi[0] = 0.0;
    } else if constexpr (FID == ABZ_F_LINEAR) {
        vr[0] = p[0] * H.re[0][0] + p[1];
        vi[0] = p[0] * H.im[0][0];
    } else if constexpr (FID == ABZ_F_LINEAR_X) {
#pragma unroll
        for (int j = 0; j < ABZ_MAX_DIM; ++j) {
            vr[j] = (j < d) ? p[0] * H.re[0][0] * xk[j] + p[1] : 0.0;
            vi[j] = (j < d) ? p[0] * H.im[0][0] * xk[j] : 0.0;
        }
    } else if constexpr (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC || FID == ABZ_F_GLOC) {
        if constexpr (FID == ABZ_F_GLOC) {
            CMat<N> G;
            gloc<N>(H, sw, p[0], G);
#pragma unroll
            for (int b = 0; b < N; ++b) {
#pragma unroll
                for (int a = 0; a < N; ++a) {
                    vr[a + N * b] = G.re[a][b];
                    vi[a + N * b] = G.im[a][b];
                }
            }
        } else {
            double tr, ti;
            gloc_trace<N>(H, sw, p[0], tr, ti);
            if constexpr (FID == ABZ_F_DOS) {
                vr[0] = -ti * 0.31830988618379067153776752674503;  // -Im tr G / pi
                vi[0] = 0.0;
            } else {
                vr[0] = tr;
                vi[0] = ti;
            }
        }
    } else if constexpr (FID == ABZ_F_DOS_EIG) {
        double acc = 0.0;
        const double eta = p[0], eta2 = eta * eta;
        if constexpr (N == 3) {  // sum of three fractions over one reciprocal
            const double d0 = sw - e[0], d1 = sw - e[1], d2 = sw - e[2];
            const double x0 = fma(d0, d0, eta2), x1 = fma(d1, d1, eta2), x2 = fma(d2, d2, eta2);
            const double x12 = x1 * x2;
            acc = fma(x0, x1 + x2, x12) * fast_rcp(x0 * x12);
        } else {
#pragma unroll
            for (int b = 0; b < N; ++b) {
                const double de = sw - e[b];
                acc += fast_rcp(fma(de, de, eta2));
            }
        }
        vr[0] = acc * (eta * 0.31830988618379067153776752674503);
        vi[0] = 0.0;
    }
}

template <int FID>
struct NComp {
    template <int N>
    static constexpr int value() {
        return FID == ABZ_F_GLOC ? N * N : (FID == ABZ_F_LINEAR_X ? ABZ_MAX_DIM : 1);
    }
};

// Cross-lane reduction of the NC components of one sweep value.  NC <= 2: wave shuffles.  NC > 2 (matrix
// and vector valued integrands): through a wave-private LDS tile -- every lane writes its NC values,
// then lane (c, g) sums every G-th row of column c: ~(NC + 64/G) LDS operations per lane instead of
// 12 NC dependent shuffles.
template <int NC>
struct ReduceShape {
    static constexpr bool viaLds = NC > 2;
    static constexpr int G = viaLds ? 64 / NC : 1;        // row groups (partial sums) per wave
    static constexpr int PW = 4 * G;                      // partial rows per sweep value and block
    static constexpr int chunk = viaLds ? (NC > 9 ? 4 : 8) : 512 / NC;  // sweep values per LDS pass
    static constexpr int tile = viaLds ? 64 * NC : 0;     // double2 per wave
};

struct ReduceArgs {
    PlaneView H, E;
    const double* w;
    const int32_t* idx;
    const double* sweep;
    int64_t nk, k_offset;
    int d, npt, n_sweep, ncomp;
    int sweep_per_row = 1 << 30;  // swept values per blockIdx.y row; default: all in row 0
    double p[4];
};

// Block tile = 256 * KT nodes held in registers; loop over the whole sweep; per omega one wave
// reduction, wave partials parked in LDS, one barrier at the end of each sweep chunk.
// HERM (rules of a Hermitian series, resolvent traces of n = 2, 3): real polynomial, upper triangle of
// H only (half of the planes are never read), ~21 instructions per (node, sweep value).
template <int N, int FID, int KT, bool HERM>
__global__ __launch_bounds__(256, FID == ABZ_F_GLOC ? (HERM ? 2 : 1) : ((HERM || FID == ABZ_F_DOS_EIG) ? (N == 4 ? 2 : 3) : 1)) void reduce_kernel(ReduceArgs a, double2* __restrict__ partial) {
    constexpr int NC = NComp<FID>::template value<N>();
    extern __shared__ double2 lds[];  // [chunk][4 waves][NC]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t base = (int64_t)blockIdx.x * (256 * KT);
    constexpr bool polyH = HERM && (N == 2 || N == 3) && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC);
    constexpr bool usePoly0 = !polyH && (N == 2 || N == 3) && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC);
    CMat<N> H[(usePoly0 || (HERM && (N == 3 || N == 4) && FID == ABZ_F_GLOC) ||
               (HERM && N == 4 && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC))) ? 1 : KT];  // polynomial modes: H is never kept
    double e[KT][N];
    double wk[KT];
    double xk[KT][ABZ_MAX_DIM];
    constexpr bool needH = (FID == ABZ_F_LINEAR || FID == ABZ_F_LINEAR_X || FID == ABZ_F_DOS ||
                            FID == ABZ_F_TRGLOC || FID == ABZ_F_GLOC);
    // n = 2, 3 resolvent traces: characteristic polynomial per node, ~40 flops per sweep value
    constexpr bool usePoly = usePoly0;
    constexpr bool adjG = HERM && N == 3 && FID == ABZ_F_GLOC;  // adjugate form of the 3x3 resolvent
    constexpr bool poly4 = HERM && N == 4 && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC);  // Faddeev-LeVerrier
    CharPolyH4 cp4[poly4 ? KT : 1];
    CharPoly<(usePoly ? N : 2)> cp[usePoly ? KT : 1];
    CharPolyH cph[polyH ? KT : 1];
    AdjH3 adj[adjG ? KT : 1];
    constexpr bool adjG4 = HERM && N == 4 && FID == ABZ_F_GLOC;  // the same for 4 bands
    AdjH4 adj4[adjG4 ? KT : 1];
    // (line, column) of this thread's first node; the following ones are 256 apart, so one 64-bit
    // division per thread and a 32-bit one per node (all views of a rule share line_len and tile)
    const int LL = a.H.line_len;
    int64_t vline = (base + threadIdx.x) / LL;
    unsigned vcol = (unsigned)((base + threadIdx.x) - vline * LL);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const int64_t k = base + threadIdx.x + 256 * j;
        const bool ok = k < a.nk;
        const int64_t kk = ok ? k : 0;
        const int64_t voff = ok ? vline * a.H.tile + vcol : 0;
        {
            vcol += 256u;
            const unsigned qd = vcol / (unsigned)LL;
            vline += qd;
            vcol -= qd * (unsigned)LL;
        }
        wk[j] = ok ? (a.w ? a.w[kk] : 1.0) : 0.0;
        if constexpr (adjG) {
            const double* __restrict__ in = a.H.base + voff;
            const int64_t pp = a.H.pitch;
            const int p01 = hplane<3>(a.H, 0, 1), p02 = hplane<3>(a.H, 0, 2), p12 = hplane<3>(a.H, 1, 2);
            adj_init_h3(in[0], in[hplane<3>(a.H, 1, 1) * pp], in[hplane<3>(a.H, 2, 2) * pp], in[p01 * pp], in[(p01 + 1) * pp], in[p02 * pp],
                        in[(p02 + 1) * pp], in[p12 * pp], in[(p12 + 1) * pp], adj[j]);
        } else if constexpr (polyH) {
            const double* __restrict__ in = a.H.base + voff;
            const int64_t pp = a.H.pitch;
            // plane of Re H[r][c]: hplane (full or Hermitian-compact layout), Im the next one
            if constexpr (N == 3) {
                const int p01 = hplane<3>(a.H, 0, 1), p02 = hplane<3>(a.H, 0, 2), p12 = hplane<3>(a.H, 1, 2);
                charpoly_init_h3(in[0], in[hplane<3>(a.H, 1, 1) * pp], in[hplane<3>(a.H, 2, 2) * pp], in[p01 * pp], in[(p01 + 1) * pp],
                                 in[p02 * pp], in[(p02 + 1) * pp], in[p12 * pp], in[(p12 + 1) * pp], cph[j]);
            } else {
                const int p01 = hplane<2>(a.H, 0, 1);
                charpoly_init_h2(in[0], in[hplane<2>(a.H, 1, 1) * pp], in[p01 * pp], in[(p01 + 1) * pp], cph[j]);
            }
        } else if constexpr (adjG4) {
            load_planes_herm<N>(H[0], a.H, voff);
            if constexpr (N == 4) adj_init_h4(H[0], adj4[j]);
        } else if constexpr (poly4) {
            load_planes_herm<N>(H[0], a.H, voff);
            if constexpr (N == 4) charpoly_init_h4(H[0], cp4[j]);
        } else if constexpr (needH) {
            load_planes<N>(H[(usePoly0 || adjG) ? 0 : j], a.H, voff);
        }
        if constexpr (usePoly) charpoly_init<N>(H[0], cp[j]);
        if constexpr (FID == ABZ_F_DOS_EIG) {
            const double* __restrict__ ei = a.E.base + voff;
#pragma unroll
            for (int b = 0; b < N; ++b) e[j][b] = ei[(int64_t)b * a.E.pitch];
        }
        if constexpr (FID == ABZ_F_LINEAR_X) {
            int64_t r = kk + a.k_offset;
            for (int t = 0; t < a.d; ++t) {
                int gi;
                if (a.idx) {
                    gi = a.idx[(int64_t)t * a.nk + kk];
                } else {
                    gi = (int)(r % a.npt);
                    r /= a.npt;
                }
                xk[j][t] = (double)gi / (double)a.npt;
            }
        }
    }
    using RS = ReduceShape<NC>;
    constexpr int chunk = RS::chunk;
    constexpr int PW = RS::PW;
    [[maybe_unused]] double2* const T = lds + (size_t)chunk * PW * NC + (size_t)wave * RS::tile;
    const int sw_lo = (int)blockIdx.y * a.sweep_per_row;  // this block row's share of the sweep
    const int sw_hi = min(a.n_sweep, sw_lo + a.sweep_per_row);
    for (int s0 = sw_lo; s0 < sw_hi; s0 += chunk) {
        const int s1 = min(sw_hi, s0 + chunk);
        [[maybe_unused]] const double eta2 = a.p[0] * a.p[0], teta = 2.0 * a.p[0];
        for (int s = s0; s < s1; ++s) {
            const double sw = a.sweep ? a.sweep[s] : 0.0;
            double ar[NC], ai[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                ar[c] = 0.0;
                ai[c] = 0.0;
            }
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                double vr[MAXC], vi[MAXC];
                if constexpr (adjG) {
                    double gr[9], gi[9];
                    adj_gloc_h3(adj[j], sw, a.p[0], gr, gi);
#pragma unroll
                    for (int c = 0; c < 9; ++c) {
                        vr[c] = gr[c];
                        vi[c] = gi[c];
                    }
                } else if constexpr (adjG4) {
                    double gr[16], gi[16];
                    adj_gloc_h4(adj4[j], sw, a.p[0], gr, gi);
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        vr[c] = gr[c];
                        vi[c] = gi[c];
                    }
                } else if constexpr (poly4) {
                    double tr, ti;
                    charpoly_trace_h4<FID != ABZ_F_DOS>(cp4[j], sw, a.p[0], eta2, teta, tr, ti);
                    vr[0] = (FID == ABZ_F_DOS) ? -ti * 0.31830988618379067153776752674503 : tr;
                    vi[0] = (FID == ABZ_F_DOS) ? 0.0 : ti;
                } else if constexpr (polyH) {
                    double tr, ti;
                    charpoly_trace_h<N, FID != ABZ_F_DOS>(cph[j], sw, a.p[0], eta2, teta, tr, ti);
                    vr[0] = (FID == ABZ_F_DOS) ? -ti * 0.31830988618379067153776752674503 : tr;
                    vi[0] = (FID == ABZ_F_DOS) ? 0.0 : ti;
                } else if constexpr (usePoly) {
                    double tr, ti;
                    charpoly_trace<N>(cp[j], sw, a.p[0], tr, ti);
                    vr[0] = (FID == ABZ_F_DOS) ? -ti * 0.31830988618379067153776752674503 : tr;
                    vi[0] = (FID == ABZ_F_DOS) ? 0.0 : ti;
                } else {
                    integrand_value<N, FID>(H[(usePoly0 || adjG || adjG4 || poly4) ? 0 : j], e[j], xk[j], a.d, a.p, sw, vr, vi);
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    ar[c] = fma(wk[j], vr[c], ar[c]);
                    ai[c] = fma(wk[j], vi[c], ai[c]);
                }
            }
            constexpr bool realValued = (FID == ABZ_F_DOS || FID == ABZ_F_DOS_EIG || FID == ABZ_F_ONE);
            if constexpr (RS::viaLds) {
#pragma unroll
                for (int c = 0; c < NC; ++c) T[lane * NC + c] = make_double2(ar[c], ai[c]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < RS::G * NC) {
                    const int c = lane % NC, g = lane / NC;
                    double2 acc = make_double2(0.0, 0.0);
                    for (int r = g; r < 64; r += RS::G) {
                        const double2 v = T[r * NC + c];
                        acc.x += v.x;
                        acc.y += v.y;
                    }
                    lds[((s - s0) * PW + wave * RS::G + g) * NC + c] = acc;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const double sr = wave_sum(ar[c]);
                    const double si = realValued ? 0.0 : wave_sum(ai[c]);
                    if (lane == 0) lds[((s - s0) * PW + wave) * NC + c] = make_double2(sr, si);
                }
            }
        }
        __syncthreads();
        const int cols = (s1 - s0) * NC;
        for (int t = threadIdx.x; t < cols; t += 256) {
            const int s = t / NC, c = t - s * NC;
            double2 acc = lds[(s * PW + 0) * NC + c];
            for (int w2 = 1; w2 < PW; ++w2) {
                const double2 v = lds[(s * PW + w2) * NC + c];
                acc.x += v.x;
                acc.y += v.y;
            }
            if (c < a.ncomp) partial[(int64_t)blockIdx.x * ((int64_t)a.n_sweep * a.ncomp) + (int64_t)(s0 + s) * a.ncomp + c] = acc;
        }
        __syncthreads();
    }
}

// out[col] = scale * sum_blocks partial[block][col]; one workgroup per column, fixed summation
// tree (reproducible run to run).
__global__ __launch_bounds__(256) void final_reduce_kernel(const double2* __restrict__ partial, int64_t nblocks,
                                                           int64_t ncols, double scale, double2* __restrict__ out) {
    __shared__ double2 sh[4];
    const int64_t col = blockIdx.x;
    double sr = 0.0, si = 0.0;
    for (int64_t b = threadIdx.x; b < nblocks; b += 256) {
        const double2 v = partial[b * ncols + col];
        sr += v.x;
        si += v.y;
    }
    sr = wave_sum(sr);
    si = wave_sum(si);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = make_double2(sr, si);
    __syncthreads();
    if (threadIdx.x == 0) {
        double ar = 0.0, ai = 0.0;
        for (int w = 0; w < 4; ++w) {
            ar += sh[w].x;
            ai += sh[w].y;
        }
        out[col] = make_double2(ar * scale, ai * scale);
    }
}

// ------------------------------------------------------------------------------------------
// DOS scans of 3-band rules (Phase B of a frequency sweep; ref: quadsum over the cached rule, src/fourier.jl:204-207, with
// the DOS integrand -Im tr inv((w + i eta) I - H(k)) / pi of aps_example/aps_example.jl:30).  One thread keeps the state of KT
// nodes -- three doubles each: (q, p1, p0) of the real characteristic cubic of H - (tr H / 3) I, or the three
// eigenvalues -- and walks the sweep.
//  MODE 0, Hermitian H:  z = x + i eta, x = w - q,  p(z) = z^3 + p1 z + p0,  u' = x^2 - 3 eta^2:
//        Re p = x (u' + p1) + p0,   Im p = eta B,   B = 3 u' + p1 + 8 eta^2,   Re p' = B - 2 eta^2,   Im p' = 6 x eta
//        Im (p'/p) = eta [6 x Re p - (B - 2 eta^2) B] / (Re p^2 + eta^2 B^2)
//    19 instructions per (node, value): the common factor -eta/pi goes into the final scale.
//  MODE 1, cached eigenvalues: sum_b 1/x_b, x_b = (w - e_b)^2 + eta^2, over ONE reciprocal: 16 instructions.
// Per swept value a lane's partial sum goes into a wave-private LDS tile [16 values][64 lanes]; every 16 values the
// tile is summed transposed (lane = (value, quarter of the lanes): 16 reads + 2 shuffles) -- 0.4 instructions per
// (node, value) where a shuffle reduction per value costs 30 / KT.  Pairs of nodes share one reciprocal.
// ------------------------------------------------------------------------------------------
constexpr int DOS3_C = 16;    // swept values per transposed reduction
constexpr int DOS3_TP = 66;   // row pitch of the tile in doubles
constexpr int DOS3_CH = 128;  // swept values per block-level pass (LDS partial rows)

template <int KT, int MODE>
__global__ __launch_bounds__(256, 3) void dos3_scan_kernel(ReduceArgs a, double three, double six, double2* __restrict__ partial) {
    extern __shared__ double lds_d[];  // [DOS3_CH][4 waves] | [4 waves][DOS3_C][DOS3_TP]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* const P = lds_d;
    double* const T = lds_d + DOS3_CH * 4 + wave * (DOS3_C * DOS3_TP);
    const int64_t base = (int64_t)blockIdx.x * (256 * KT);
    double c0[KT], c1[KT], c2[KT], wk[KT];
    const int LL = (MODE == 0 ? a.H : a.E).line_len;
    const int64_t tile = (MODE == 0 ? a.H : a.E).tile;
    int64_t vline = (base + threadIdx.x) / LL;
    unsigned vcol = (unsigned)((base + threadIdx.x) - vline * LL);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const int64_t k = base + threadIdx.x + 256 * j;
        const bool ok = k < a.nk;
        const int64_t voff = ok ? vline * tile + vcol : 0;
        {
            vcol += 256u;
            const unsigned qd = vcol / (unsigned)LL;
            vline += qd;
            vcol -= qd * (unsigned)LL;
        }
        wk[j] = ok ? (a.w ? a.w[k] : 1.0) : 0.0;
        if constexpr (MODE == 0) {
            const double* __restrict__ in = a.H.base + voff;
            const int64_t pp = a.H.pitch;
            CharPolyH cp;  // plane of Re H[r][c]: hplane (full or Hermitian-compact layout), Im the next one
            const int p01 = hplane<3>(a.H, 0, 1), p02 = hplane<3>(a.H, 0, 2), p12 = hplane<3>(a.H, 1, 2);
            charpoly_init_h3(in[0], in[hplane<3>(a.H, 1, 1) * pp], in[hplane<3>(a.H, 2, 2) * pp], in[p01 * pp], in[(p01 + 1) * pp],
                             in[p02 * pp], in[(p02 + 1) * pp], in[p12 * pp], in[(p12 + 1) * pp], cp);
            c0[j] = cp.q;
            c1[j] = cp.p1;
            c2[j] = cp.p0;
        } else {
            const double* __restrict__ ei = a.E.base + voff;
            c0[j] = ei[0];
            c1[j] = ei[a.E.pitch];
            c2[j] = ei[2 * (int64_t)a.E.pitch];
        }
    }
    const double eta = a.p[0], e2 = eta * eta;
    const double m3e2 = -3.0 * e2, p8e2 = 8.0 * e2, p6e2 = 6.0 * e2;
    // nodes of weight one (full grids, all of the block inside the rule): the weight multiplication is dropped
    const bool weighted = a.w != nullptr || base + 256 * KT > a.nk;
    const int sw_lo = (int)blockIdx.y * a.sweep_per_row;
    const int sw_hi = min(a.n_sweep, sw_lo + a.sweep_per_row);
    auto value = [&](auto wtag, double sw) -> double {
        constexpr bool W = decltype(wtag)::value;
        double acc = 0.0;
        double num[KT], den[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            if constexpr (MODE == 0) {
                const double x = sw - c0[j];
                const double up = fma(x, x, m3e2);
                const double A = up + c1[j];
                const double dr = fma(x, A, c2[j]);
                const double B0 = fma(three, up, c1[j]);
                const double B = B0 + p8e2;
                const double Bm = B0 + p6e2;
                const double m = Bm * B;
                const double t1 = x * dr;
                num[j] = fma(six, t1, -m);
                den[j] = fma(e2 * B, B, dr * dr);
            } else {
                const double d0 = sw - c0[j], d1 = sw - c1[j], d2 = sw - c2[j];
                const double x0 = fma(d0, d0, e2), x1 = fma(d1, d1, e2), x2 = fma(d2, d2, e2);
                const double x12 = x1 * x2;
                num[j] = fma(x0, x1 + x2, x12);
                den[j] = x0 * x12;
            }
            if constexpr (W) num[j] *= wk[j];
        }
        // one reciprocal per PAIR of nodes, n1/d1 + n2/d2 = (n1 d2 + n2 d1) / (d1 d2): v_rcp_f64 issues at a quarter of
        // the FMA rate and carries two Newton steps.  The denominators are |p(z)|^2 (or products of three Lorentzian
        // denominators): a product of two stays far inside the double range for any sane energy scale.
#pragma unroll
        for (int j = 0; j + 1 < KT; j += 2) {
            const double t = fma(num[j + 1], den[j], num[j] * den[j + 1]);
            acc = fma(t, fast_rcp(den[j] * den[j + 1]), acc);
        }
        if constexpr (KT & 1) acc = fma(num[KT - 1], fast_rcp(den[KT - 1]), acc);
        return acc;
    };
    auto fill = [&](auto wtag, int sb, int cnt) {  // partial sums of `cnt` swept values into the wave's tile
        double sw = a.sweep[sb];
        for (int i = 0; i < cnt; ++i) {
            const double sw_next = a.sweep[min(sb + i + 1, a.n_sweep - 1)];  // scalar load one value ahead
            T[i * DOS3_TP + lane] = value(wtag, sw);
            sw = sw_next;
        }
    };
    for (int s0 = sw_lo; s0 < sw_hi; s0 += DOS3_CH) {
        const int s1 = min(sw_hi, s0 + DOS3_CH);
        for (int sb = s0; sb < s1; sb += DOS3_C) {
            const int cnt = min(DOS3_C, s1 - sb);
            if (weighted)
                fill(std::true_type{}, sb, cnt);
            else
                fill(std::false_type{}, sb, cnt);
            wave_lds_sync();
            {
                const double* __restrict__ row = T + (lane & 15) * DOS3_TP + (lane >> 4) * 16;
                double sum = 0.0;  // rows >= cnt hold older values: summed, never written out
#pragma unroll
                for (int i = 0; i < 16; ++i) sum += row[i];
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                if (lane < cnt) P[(sb - s0 + lane) * 4 + wave] = sum;
            }
            wave_lds_sync();
        }
        __syncthreads();
        for (int t = threadIdx.x; t < s1 - s0; t += 256) {
            const double v = (P[t * 4] + P[t * 4 + 1]) + (P[t * 4 + 2] + P[t * 4 + 3]);
            partial[(int64_t)blockIdx.x * (int64_t)a.n_sweep + (s0 + t)] = make_double2(v, 0.0);
        }
        __syncthreads();
    }
}

// mode 0: ABZ_F_DOS on the H planes of a Hermitian rule; mode 1: ABZ_F_DOS_EIG on the eigenvalue planes
static int launch_dos3(abz_ctx* ctx, const ReduceSpec& rs, const ReduceArgs& a0, int mode) {
    ReduceArgs a = a0;
    // KT = 8 nodes per thread at 3 waves per SIMD (KT = 4 ... 7, and 6 or 4 at 4 waves per SIMD, measured the same to 3 % at
    // 150^3: the kernel is bound by VALU issue).  Small rules (one rank's slab of a k-sharded grid) split the SWEEP over
    // blockIdx.y as well, >= 16 values per row; long sweeps take two rows (256 values: -5 %, the tail of the last blocks).
    constexpr int kt = 8;
    int rows = 1;
    {
        const int force = abz_switch(SW_REDUCE_ROWS);  // per call (tests)
        const int max_rows = (int)cdiv(rs.n_sweep, 16);
        const int want = force > 0 ? force : std::max((int)cdiv(1024, cdiv(rs.nk, 256 * kt)), rs.n_sweep >= 192 ? 2 : 1);
        rows = std::max(1, std::min(want, max_rows));
        const int per = (int)cdiv(rs.n_sweep, rows);
        a.sweep_per_row = per;
        rows = (int)cdiv(rs.n_sweep, per);
    }
    const int64_t nblocks = cdiv(rs.nk, 256 * kt);
    const int64_t ncols = rs.n_sweep;
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(nblocks * ncols));
    if (rc) return rc;
    const size_t lds = sizeof(double) * ((size_t)DOS3_CH * 4 + (size_t)4 * DOS3_C * DOS3_TP);
    const double pi = 3.14159265358979323846;
    const double scale = rs.scale * (mode == 0 ? -rs.params[0] / pi : rs.params[0] / pi);
    if (mode == 0)
        hipLaunchKernelGGL((dos3_scan_kernel<kt, 0>), dim3((unsigned)nblocks, (unsigned)rows), dim3(256), lds, ctx->stream, a, 3.0, 6.0,
                           ctx->scratch[1].as<double2>());
    else
        hipLaunchKernelGGL((dos3_scan_kernel<kt, 1>), dim3((unsigned)nblocks, (unsigned)rows), dim3(256), lds, ctx->stream, a, 3.0, 6.0,
                           ctx->scratch[1].as<double2>());
    ABZ_HIP(hipGetLastError());
    hipLaunchKernelGGL(final_reduce_kernel, dim3((unsigned)ncols), dim3(256), 0, ctx->stream, ctx->scratch[1].as<double2>(),
                       nblocks, ncols, scale, rs.out_map_dev ? rs.out_map_dev : ctx->scratch[2].as<double2>());
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// Store-free PTR: rule(f, B) without materialising the rule.  Same work loop as eval_grid_kernel, but a
// unit's H(k) go straight into the integrand and the per-lane partial sums; nothing is written per node.
// For grids that are used once (an AutoPTR refinement step, a single omega) or do not fit in HBM
// (1000^3 k-points x 168 B = 168 GB): the kernel is bound by the m-loop, ~40 G k-points/s.
// NW sweep values per launch share the H(k) of a node (their sums live in registers).
// ------------------------------------------------------------------------------------------
struct SumArgs {
    int fid, nw, ncomp, d;
    double p[4];
    double sweep[8];
};

template <int N, int KPL, bool HERM, int FID, int NW>
__global__ __launch_bounds__(256, 2) void eval_sum_grid_kernel(EvalArgs a, SumArgs q, double2* __restrict__ partial) {
    extern __shared__ double2 lds_c[];  // [4 waves][2 buffers][MNN] | [npt] table
    constexpr int NC = NComp<FID>::template value<N>();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // the level-1 sets are PACKED Hermitian sets (packed_herm.h): this kernel serves Hermitian series only
    const int MNN = Pk<N>::size((a.M - 1) / 2);
    double2* const mybuf = lds_c + (size_t)wave * 2 * MNN;
    int fm = a.first % a.npt;
    if (fm < 0) fm += a.npt;
    const int npass = (a.npt + 64 * KPL - 1) / (64 * KPL);
    double2* const tab_l = lds_c + (size_t)4 * 2 * MNN;
    for (int i = threadIdx.x; i < a.npt; i += 256) tab_l[i] = a.tab[i];
    __syncthreads();
    double accr[NW][NC], acci[NW][NC];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            accr[w][c] = 0.0;
            acci[w][c] = 0.0;
        }
    }
    const int64_t lstride = (int64_t)gridDim.x * 4;
    int64_t line = (int64_t)blockIdx.x * 4 + wave;
    if (line < a.nlines) {
        const double2* __restrict__ src = a.src + line * MNN;
#pragma unroll
        for (int t = 0; t < EVAL_MAX_MNN / 64; ++t) {
            const int idx = lane + 64 * t;
            if (idx < MNN) mybuf[idx] = src[idx];
        }
    }
    int iz0[KPL], iw0[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int i1 = lane + 64 * j;
        iz0[j] = i1 < a.npt ? i1 : 0;
        iw0[j] = (int)(((unsigned)fm * (unsigned)iz0[j]) % (unsigned)a.npt);
    }
    [[maybe_unused]] const double eta2 = q.p[0] * q.p[0], teta = 2.0 * q.p[0];
    int cur = 0, pass = 0;
    while (line < a.nlines) {
        const bool last_pass = pass + 1 >= npass;
        const int64_t nline = last_pass ? line + lstride : line;
        const bool have_next = nline < a.nlines;
        double2 pre[EVAL_MAX_MNN / 64];
        {
            const double2* __restrict__ src = a.src + (have_next ? nline : line) * MNN;
#pragma unroll
            for (int t = 0; t < EVAL_MAX_MNN / 64; ++t) {
                const int idx = lane + 64 * t;
                pre[t] = src[idx < MNN ? idx : MNN - 1];
            }
        }
        wave_lds_sync();
        const int i0 = pass * (64 * KPL);
        CMat<N> H[KPL];
        auto mid_fn = [=]() {  // by value: by reference the prefetched registers made an 80-B scratch round trip per unit
            if (have_next) {
                double2* dst = mybuf + (size_t)(cur ^ 1) * MNN;
#pragma unroll
                for (int t = 0; t < EVAL_MAX_MNN / 64; ++t) {
                    const int idx = lane + 64 * t;
                    if (idx < MNN) dst[idx] = pre[t];
                }
            }
        };
        eval_unit_core<N, KPL, HERM, decltype(mid_fn)&, true>(a, mybuf + (size_t)cur * MNN, tab_l, fm, iz0, iw0, npass, i0, lane, mid_fn, H);
        // integrand at the unit's nodes
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int i1 = i0 + lane + 64 * j;
            if (i1 < a.npt) {
                double xk[ABZ_MAX_DIM] = {0.0, 0.0, 0.0};
                if constexpr (FID == ABZ_F_LINEAR_X) {
                    xk[0] = (double)i1 / (double)a.npt;
                    int64_t r = line + a.line0;  // (i2, i3) of the line, slab offset included
                    for (int t = 1; t < q.d; ++t) {
                        xk[t] = (double)(r % a.npt) / (double)a.npt;
                        r /= a.npt;
                    }
                }
                if constexpr (HERM && N == 3 && FID == ABZ_F_DOS) {
                    // the arithmetic of dos3_scan_kernel: numerator and denominator of Im p'/p in 13 instructions, one
                    // reciprocal per pair of swept values, the factor -eta/pi applied once per lane after the loop
                    CharPolyH cp;
                    charpoly_init_h3(H[j].re[0][0], H[j].re[1][1], H[j].re[2][2], H[j].re[0][1], H[j].im[0][1], H[j].re[0][2],
                                     H[j].im[0][2], H[j].re[1][2], H[j].im[1][2], cp);
                    double num[NW], den[NW];
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        num[w] = 0.0;
                        den[w] = 1.0;
                        if (w < q.nw) {
                            const double x = q.sweep[w] - cp.q;
                            const double up = fma(x, x, -3.0 * eta2);
                            const double dr = fma(x, up + cp.p1, cp.p0);
                            const double B0 = fma(3.0, up, cp.p1);
                            const double B = B0 + 8.0 * eta2;
                            num[w] = fma(6.0, x * dr, -((B0 + 6.0 * eta2) * B));
                            den[w] = fma(eta2 * B, B, dr * dr);
                        }
                    }
#pragma unroll
                    for (int w = 0; w + 1 < NW; w += 2) {
                        if (w + 1 < q.nw) {
                            const double r = fast_rcp(den[w] * den[w + 1]);
                            accr[w][0] = fma(num[w] * den[w + 1], r, accr[w][0]);
                            accr[w + 1][0] = fma(num[w + 1] * den[w], r, accr[w + 1][0]);
                        } else if (w < q.nw) {
                            accr[w][0] = fma(num[w], fast_rcp(den[w]), accr[w][0]);
                        }
                    }
                    if constexpr (NW & 1) {
                        if (NW - 1 < q.nw) accr[NW - 1][0] = fma(num[NW - 1], fast_rcp(den[NW - 1]), accr[NW - 1][0]);
                    }
                } else if constexpr (HERM && (N == 2 || N == 3) && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC)) {
                    CharPolyH cp;
                    if constexpr (N == 3)
                        charpoly_init_h3(H[j].re[0][0], H[j].re[1][1], H[j].re[2][2], H[j].re[0][1], H[j].im[0][1], H[j].re[0][2],
                                         H[j].im[0][2], H[j].re[1][2], H[j].im[1][2], cp);
                    else
                        charpoly_init_h2(H[j].re[0][0], H[j].re[1][1], H[j].re[0][1], H[j].im[0][1], cp);
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        if (w < q.nw) {
                            double tr, ti;
                            charpoly_trace_h<N, FID != ABZ_F_DOS>(cp, q.sweep[w], q.p[0], eta2, teta, tr, ti);
                            accr[w][0] += (FID == ABZ_F_DOS) ? -ti * 0.31830988618379067153776752674503 : tr;
                            acci[w][0] += (FID == ABZ_F_DOS) ? 0.0 : ti;
                        }
                    }
                } else if (N == 4 && HERM && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC) && q.nw >= 3) {
                    // 4 bands, several sweep values: Faddeev-LeVerrier polynomial once, ~36 flops per value
                    if constexpr (N == 4) {
                        CharPolyH4 cp;
                        charpoly_init_h4(H[j], cp);
#pragma unroll
                        for (int w = 0; w < NW; ++w) {
                            if (w < q.nw) {
                                double tr, ti;
                                charpoly_trace_h4<FID != ABZ_F_DOS>(cp, q.sweep[w], q.p[0], eta2, teta, tr, ti);
                                accr[w][0] += (FID == ABZ_F_DOS) ? -ti * 0.31830988618379067153776752674503 : tr;
                                acci[w][0] += (FID == ABZ_F_DOS) ? 0.0 : ti;
                            }
                        }
                    }
                } else {
                    double e[N];
                    if constexpr (FID == ABZ_F_DOS_EIG) {
                        herm_eig_values<N>(H[j], e);
                    }
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        if (w < q.nw) {
                            double vr[MAXC], vi[MAXC];
                            integrand_value<N, FID>(H[j], e, xk, q.d, q.p, q.sweep[w], vr, vi);
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                accr[w][c] += vr[c];
                                acci[w][c] += vi[c];
                            }
                        }
                    }
                }
            }
        }
        cur ^= 1;
        pass = last_pass ? 0 : pass + 1;
        line = nline;
    }
    if constexpr (HERM && N == 3 && FID == ABZ_F_DOS) {  // the common factor of the 3-band DOS terms (see the node loop)
        const double fac = -q.p[0] * 0.31830988618379067153776752674503;
#pragma unroll
        for (int w = 0; w < NW; ++w) accr[w][0] *= fac;
    }
    // block partial sums: wave shuffles, then the 4 waves through LDS (after everyone is done with it)
    __syncthreads();
    double2* red = lds_c;  // [4][NW * NC]
#pragma unroll
    for (int w = 0; w < NW; ++w) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double sr = wave_sum(accr[w][c]);
            const double si = wave_sum(acci[w][c]);
            if (lane == 0) red[wave * (NW * NC) + w * NC + c] = make_double2(sr, si);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NW * NC; t += 256) {
        const int w = t / NC, c = t - w * NC;
        double2 acc = red[t];
        for (int w2 = 1; w2 < 4; ++w2) {
            acc.x += red[w2 * (NW * NC) + t].x;
            acc.y += red[w2 * (NW * NC) + t].y;
        }
        if (w < q.nw && c < q.ncomp) partial[(int64_t)blockIdx.x * ((int64_t)q.nw * q.ncomp) + (int64_t)w * q.ncomp + c] = acc;
    }
}

int launch_final_reduce(abz_ctx* ctx, const double2* partial, int64_t nblocks, int64_t ncols, double scale, double2* out) {
    hipLaunchKernelGGL(final_reduce_kernel, dim3((unsigned)ncols), dim3(256), 0, ctx->stream, partial, nblocks, ncols, scale, out);
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

bool eval_sum_supported(int n, int M, int npt, int integrand, bool herm) {
    if (!herm || n < 1 || n > 4 || npt <= 128 || npt >= 65536) return false;
    const int mnn = M * n * n;
    if (mnn > EVAL_MAX_MNN || sizeof(double2) * (4 * 2 * (size_t)mnn + (size_t)npt) > 64 * 1024) return false;
    if ((integrand == ABZ_F_LINEAR || integrand == ABZ_F_LINEAR_X) && n != 1) return false;
    return integrand >= ABZ_F_ONE && integrand <= ABZ_F_DOS_EIG;
}

// out_reim [n_sweep][ncomp][2] (host): scale * sum over the nlines * npt nodes
int launch_eval_sum(abz_ctx* ctx, const SumSpec& ss, double* out_reim) {
    const int ncomp = integrand_ncomp(ss.integrand, ss.n, ss.d);
    const bool swept = ss.integrand == ABZ_F_DOS || ss.integrand == ABZ_F_TRGLOC || ss.integrand == ABZ_F_GLOC ||
                       ss.integrand == ABZ_F_DOS_EIG;
    const int ns = swept ? ss.n_sweep : 1;
    if (ss.integrand == ABZ_F_ONE) {  // the sum of ones
        out_reim[0] = ss.scale * (double)ss.nlines * (double)ss.npt;
        out_reim[1] = 0.0;
        return ABZ_OK;
    }
    EvalArgs a{};
    a.src = ss.src;
    a.tab = ss.tab;
    a.nlines = ss.nlines;
    a.nk = ss.nlines * ss.npt;
    a.line0 = ss.line0;
    a.M = ss.M;
    a.first = ss.first;
    a.npt = ss.npt;
    a.deriv = 0;
    a.herm = 1;
    a.nt = 0;
    a.inv_period = 1.0;
    int kpl = 2;
    {
        double best = 1e30;
        for (int k = 2; k <= 3; ++k) {
            const double cost = (double)(cdiv(ss.npt, 64 * k) * k) * (1.0 + 0.3 / k);
            if (cost < best - 1e-12) {
                best = cost;
                kpl = k;
            }
        }
    }
    const int mnn = (int)packed_row_elems(ss.n, ss.M);  // packed Hermitian level-1 sets (packed_herm.h)
    const size_t lds = sizeof(double2) * (4 * 2 * (size_t)mnn + (size_t)ss.npt);
    const int64_t blocks = std::min<int64_t>(cdiv(ss.nlines, 4), 256 * 8);
    // scalar integrands: up to 8 sweep values share the H(k) of a node (must match the NW of the dispatch below)
    const int nwmax = (ss.integrand == ABZ_F_GLOC || ss.integrand == ABZ_F_LINEAR_X) ? 1 : 8;
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(blocks * nwmax * ncomp));
    if (rc) return rc;
    if ((rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)(nwmax * ncomp)))) return rc;
    double2* partial = ctx->scratch[1].as<double2>();
    double2* outd = ctx->scratch[2].as<double2>();
    for (int s0 = 0; s0 < ns; s0 += nwmax) {
        SumArgs q;
        q.fid = ss.integrand;
        q.nw = std::min(nwmax, ns - s0);
        q.ncomp = ncomp;
        q.d = ss.d;
        for (int i = 0; i < 4; ++i) q.p[i] = ss.params[i];
        for (int w = 0; w < 8; ++w) q.sweep[w] = (swept && w < q.nw) ? ss.sweep_host[s0 + w] : 0.0;
        {
            ProfScope ps(ctx, ABZ_K_EVAL);
#define SUMK(NN, KK, FF, WW) \
    hipLaunchKernelGGL((eval_sum_grid_kernel<NN, KK, true, FF, WW>), dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a, q, partial)
#define SUMN(NN, FF, WW)   \
    if (kpl == 2) {        \
        SUMK(NN, 2, FF, WW); \
    } else {               \
        SUMK(NN, 3, FF, WW); \
    }
#define SUMF(FF, WW)                               \
    switch (ss.n) {                                \
        case 1: SUMN(1, FF, WW) break;             \
        case 2: SUMN(2, FF, WW) break;             \
        case 3: SUMN(3, FF, WW) break;             \
        default: SUMN(4, FF, WW) break;            \
    }
            switch (ss.integrand) {
                case ABZ_F_LINEAR: SUMN(1, ABZ_F_LINEAR, 8) break;
                case ABZ_F_LINEAR_X: SUMN(1, ABZ_F_LINEAR_X, 1) break;
                case ABZ_F_DOS: SUMF(ABZ_F_DOS, 8) break;
                case ABZ_F_TRGLOC: SUMF(ABZ_F_TRGLOC, 8) break;
                case ABZ_F_GLOC: SUMF(ABZ_F_GLOC, 1) break;
                case ABZ_F_DOS_EIG: SUMF(ABZ_F_DOS_EIG, 8) break;
                default: set_error("store-free sum: integrand %d", ss.integrand); return ABZ_ERR_UNSUPPORTED;
            }
#undef SUMF
#undef SUMN
#undef SUMK
            ABZ_HIP(hipGetLastError());
            const int64_t ncols = (int64_t)q.nw * ncomp;
            hipLaunchKernelGGL(final_reduce_kernel, dim3((unsigned)ncols), dim3(256), 0, ctx->stream, partial, blocks, ncols,
                               ss.scale, outd);
            ABZ_HIP(hipGetLastError());
        }
        ABZ_HIP(hipMemcpyAsync(out_reim + 2 * (size_t)s0 * ncomp, outd, sizeof(double2) * (size_t)(q.nw * ncomp),
                               hipMemcpyDeviceToHost, ctx->stream));
        ABZ_HIP(hipStreamSynchronize(ctx->stream));
    }
    return ABZ_OK;
}

// nodes per thread: the wave reduction per sweep value is amortised over KT nodes; the cheap
// per-node states (characteristic polynomial: 6 doubles, eigenvalues: n doubles) allow KT = 8
template <int N, int FID, bool HERM>
constexpr int reduce_kt_of() {
    if (HERM && N == 3 && FID == ABZ_F_GLOC) return 3;  // adjugate state: 21 doubles per node (4 would spill)
    if (HERM && N == 4 && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC)) return 8;  // 4 real coefficients per node
    if (FID == ABZ_F_GLOC || N >= 4) return 1;
    if (FID == ABZ_F_DOS_EIG || ((N == 2 || N == 3) && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC))) return 8;
    return 2;
}

// A block owns 256 KT nodes for the whole sweep, so a small rule leaves the chip under-filled: one rank's slab of a
// k-sharded 150^3 grid (4.3e5 nodes at 8 GPUs) is 209 blocks on 256 CUs and its 256-omega scan took 0.20 ms, a quarter of
// the full grid's 0.83 ms for an eighth of the work.  Rules with fewer than ~4 blocks per CU split the SWEEP over
// blockIdx.y as well (the per-node polynomial is set up once per row: ~60 flops against
// 27 per swept value).  KT = 2 instead of 8 was measured too: 4x the blocks but 0.227 ms -- the per-value wave reduction
// is then amortised over 2 nodes instead of 8.
template <int N, int FID, bool HERM>
static int launch_reduce_h(abz_ctx* ctx, const ReduceSpec& rs, const ReduceArgs& a0) {
    constexpr int KT = reduce_kt_of<N, FID, HERM>();
    constexpr int NC = NComp<FID>::template value<N>();
    const int64_t nblocks = cdiv(rs.nk, 256 * KT);
    ReduceArgs a = a0;
    const int64_t ncols = (int64_t)rs.n_sweep * a.ncomp;
    int rc = ctx->scratch[1].reserve(sizeof(double2) * (size_t)(nblocks * ncols));
    if (rc) return rc;
    using RS = ReduceShape<NC>;
    int rows = 1;
    {
        const int force = abz_switch(SW_REDUCE_ROWS);
        const int max_rows = (int)cdiv(rs.n_sweep, 16);  // at least 16 swept values per row: the node set-up stays < 15 %
        const int want = force > 0 ? force : (int)cdiv(1024, nblocks);
        rows = std::max(1, std::min(want, max_rows));
        const int per = (int)cdiv(rs.n_sweep, rows);
        a.sweep_per_row = per;
        rows = (int)cdiv(rs.n_sweep, per);
    }
    const size_t lds = sizeof(double2) * ((size_t)RS::chunk * RS::PW * NC + (size_t)4 * RS::tile);
    if (lds > 64 * 1024)
        ABZ_HIP(hipFuncSetAttribute((const void*)reduce_kernel<N, FID, KT, HERM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    hipLaunchKernelGGL((reduce_kernel<N, FID, KT, HERM>), dim3((unsigned)nblocks, (unsigned)rows), dim3(256), lds, ctx->stream, a,
                       ctx->scratch[1].as<double2>());
    ABZ_HIP(hipGetLastError());
    hipLaunchKernelGGL(final_reduce_kernel, dim3((unsigned)ncols), dim3(256), 0, ctx->stream, ctx->scratch[1].as<double2>(),
                       nblocks, ncols, rs.scale, rs.out_map_dev ? rs.out_map_dev : ctx->scratch[2].as<double2>());
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// Hermitian rules take the real-polynomial / adjugate paths where they exist
template <int N, int FID>
static int launch_reduce_t(abz_ctx* ctx, const ReduceSpec& rs, const ReduceArgs& a) {
    constexpr bool canH = ((N == 2 || N == 3 || N == 4) && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC)) || ((N == 3 || N == 4) && FID == ABZ_F_GLOC);
    if constexpr (N == 3 && (FID == ABZ_F_DOS || FID == ABZ_F_DOS_EIG)) {  // the sweep kernel of 3-band DOS scans
        if (abz_switch(SW_DOS3_SCAN) && (FID == ABZ_F_DOS_EIG || rs.herm)) return launch_dos3(ctx, rs, a, FID == ABZ_F_DOS ? 0 : 1);
    }
    if (canH && rs.herm) return launch_reduce_h<N, FID, canH>(ctx, rs, a);
    return launch_reduce_h<N, FID, false>(ctx, rs, a);
}

template <int FID>
static constexpr int reduce_kt(int n) {
    return (FID == ABZ_F_GLOC || n >= 4) ? 1 : 2;
}

int launch_reduce(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim) {
    if (rs.n > 4) return launch_gen_reduce(ctx, rs, out_reim);
    const int ncomp = integrand_ncomp(rs.integrand, rs.n, rs.d);
    if (ncomp < 0) {
        set_error("unknown integrand id %d", rs.integrand);
        return ABZ_ERR_ARG;
    }
    if ((rs.integrand == ABZ_F_LINEAR || rs.integrand == ABZ_F_LINEAR_X) && rs.n != 1) {
        set_error("ABZ_F_LINEAR(_X) needs a scalar (n = 1) series");
        return ABZ_ERR_ARG;
    }
    ReduceArgs a;
    a.H = rs.H;
    a.E = rs.E;
    a.w = rs.w;
    a.idx = rs.idx;
    a.k_offset = rs.k_offset;
    a.sweep = rs.sweep_dev;
    a.nk = rs.nk;
    a.d = rs.d;
    a.npt = rs.npt;
    a.n_sweep = rs.n_sweep;
    a.ncomp = ncomp;
    for (int i = 0; i < 4; ++i) a.p[i] = rs.params[i];
    const int64_t ncols = (int64_t)rs.n_sweep * ncomp;
    int rc = ctx->scratch[2].reserve(sizeof(double2) * (size_t)ncols);
    if (rc) return rc;
    double2* outd = ctx->scratch[2].as<double2>();
    {
        ProfScope ps(ctx, ABZ_K_REDUCE);
#define CASE(FID)                                                          \
    case FID:                                                              \
        switch (rs.n) {                                                    \
            case 1: rc = launch_reduce_t<1, FID>(ctx, rs, a); break;       \
            case 2: rc = launch_reduce_t<2, FID>(ctx, rs, a); break;       \
            case 3: rc = launch_reduce_t<3, FID>(ctx, rs, a); break;       \
            case 4: rc = launch_reduce_t<4, FID>(ctx, rs, a); break;       \
            default:                                                       \
                set_error("n = %d bands: only n <= 4 is built in this round", rs.n); \
                return ABZ_ERR_UNSUPPORTED;                                \
        }                                                                  \
        break;
        switch (rs.integrand) {
            CASE(ABZ_F_ONE)
            CASE(ABZ_F_LINEAR)
            CASE(ABZ_F_LINEAR_X)
            CASE(ABZ_F_DOS)
            CASE(ABZ_F_TRGLOC)
            CASE(ABZ_F_GLOC)
            CASE(ABZ_F_DOS_EIG)
        }
#undef CASE
        if (rc) return rc;
    }
    if (rs.out_dev) {  // the sums stay in HBM (they feed a collective on the same stream)
        ABZ_HIP(hipMemcpyAsync(rs.out_dev, outd, sizeof(double2) * (size_t)ncols, hipMemcpyDeviceToDevice, ctx->stream));
        return ABZ_OK;
    }
    if (rs.out_map_dev) {  // the last kernel wrote the sums into the pinned mailbox: one synchronisation, no copy call
        if (!rs.out_map_host) return ABZ_OK;  // ... which the caller does itself, after enqueueing more work
        ABZ_HIP(hipStreamSynchronize(ctx->stream));
        std::memcpy(out_reim, rs.out_map_host, sizeof(double2) * (size_t)ncols);
        return ABZ_OK;
    }
    ABZ_HIP(hipMemcpyAsync(out_reim, outd, sizeof(double2) * (size_t)ncols, hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// export planar -> AoS
// ------------------------------------------------------------------------------------------
// row_major_n > 0 (matrix planes, abz_eval_nodes with ABZ_WANT_H_ROW_MAJOR): component 2 (a n + b) + {re, im} of the output is the
// matrix element (a, b), i.e. plane 2 (a + n b) + {re, im}
__global__ void export_kernel(PlaneView v, int ncomp, int64_t nk, double* __restrict__ out, int row_major_n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nk * ncomp) return;
    const int64_t k = t / ncomp;
    int c = (int)(t - k * ncomp);
    if (row_major_n > 0) {
        const int im = c & 1, ab = c >> 1, a = ab / row_major_n, b = ab - a * row_major_n;
        c = 2 * (a + row_major_n * b) + im;
    }
    if (v.compact) {  // upper-triangle planes -> the full matrix in the reference's order, component 2 (a + n b) + {re, im}
        const int n = v.compact, im = c & 1, ab = c >> 1;
        const int a = ab % n, b = ab / n;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        double val = 0.0;
        if (!(im && a == b)) val = v.base[view_off(v, k) + (int64_t)(hi * hi + 2 * lo + im) * v.pitch];
        out[t] = (im && a > b) ? -val : val;
        return;
    }
    out[t] = v.base[view_off(v, k) + (int64_t)c * v.pitch];
}

int export_planes(abz_ctx* ctx, PlaneView v, int ncomp, int64_t nk, double* host_out, int row_major_n) {
    if (nk == 0) return ABZ_OK;
    const size_t bytes = sizeof(double) * (size_t)nk * ncomp;
    int rc = ctx->scratch[3].reserve(bytes);
    if (rc) return rc;
    double* stg = ctx->scratch[3].as<double>();
    hipLaunchKernelGGL(export_kernel, dim3((unsigned)cdiv(nk * ncomp, 256)), dim3(256), 0, ctx->stream, v, ncomp, nk, stg, row_major_n);
    ABZ_HIP(hipGetLastError());
    return stage_d2h(ctx, host_out, stg, bytes);  // large: through the pinned staging buffer
}

// ------------------------------------------------------------------------------------------
// IAI innermost nodes: series + integrand, one lane per node
// ------------------------------------------------------------------------------------------
// H(k) of a Hermitian series at ONE point from a PACKED coefficient set (packed_herm.h; LDS or constant-address-space
// pointer): z = e^{2 pi i x}, p = z^f by recurrence, one FMA group for +f and -f, the lower triangle mirrored.  The
// arithmetic of eval_unit_core's packed branch for a single node: the IAI kernels' series (half the terms of the
// 2 F + 1 loop, and no seed phase z^first).
// FC > 0: the number of frequency pairs is known at compile time (the loop unrolls and the coefficient reads of the later
// frequencies are issued while the earlier ones are summed); the same operations in the same order either way.
template <int N, class PTR, int FC = 0>
__device__ __forceinline__ void series_point_pk(PTR c1, int F_, double zr, double zi, CMat<N>& H) {
    const int F = FC > 0 ? FC : F_;
#pragma unroll
    for (int bb = 0; bb < N; ++bb) {
#pragma unroll
        for (int aa = 0; aa <= bb; ++aa) {
            const double cx = c1[Pk<N>::tri(aa, bb)].x, cy = c1[Pk<N>::tri(aa, bb)].y;
            H.re[aa][bb] = cx;
            H.im[aa][bb] = (aa == bb) ? 0.0 : cy;
        }
    }
    double pr = 1.0, pi = 0.0;
#pragma unroll
    for (int f = 1; f <= F; ++f) {
        // every multiply-add is spelled out: left to the compiler, `a * b + c * d` is contracted one way in one kernel and
        // the other way in the next, and the IAI kernels that share this function stop agreeing to the bit
        const double nr = fma(pr, zr, -(pi * zi));
        const double ni = fma(pr, zi, pi * zr);
        pr = nr;
        pi = ni;
        const int o = Pk<N>::blk(1) + (f - 1) * (N * N);
#pragma unroll
        for (int aa = 0; aa < N; ++aa) {
            const double dx = c1[o + aa].x, dy = c1[o + aa].y;
            H.re[aa][aa] = fma(dx, pr, H.re[aa][aa]);
            H.re[aa][aa] = fma(-dy, pi, H.re[aa][aa]);
        }
#pragma unroll
        for (int bb = 1; bb < N; ++bb) {
#pragma unroll
            for (int aa = 0; aa < bb; ++aa) {
                const int q = o + N + 2 * Pk<N>::pair(aa, bb);
                const double sx = c1[q].x, sy = c1[q].y, tx = c1[q + 1].x, ty = c1[q + 1].y;
                H.re[aa][bb] = fma(sx, pr, H.re[aa][bb]);
                H.re[aa][bb] = fma(-sy, pi, H.re[aa][bb]);
                H.im[aa][bb] = fma(tx, pi, H.im[aa][bb]);
                H.im[aa][bb] = fma(ty, pr, H.im[aa][bb]);
            }
        }
    }
#pragma unroll
    for (int bb = 0; bb < N; ++bb) {
#pragma unroll
        for (int aa = bb + 1; aa < N; ++aa) {
            H.re[aa][bb] = H.re[bb][aa];
            H.im[aa][bb] = -H.im[bb][aa];
        }
    }
}

struct NodeArgs {
    const double2* src;
    const int64_t* parents;
    const double* x;
    const double* tail;
    const double* sweep_arr;
    int64_t nnodes;
    int M, first, d, ncomp;
    int pk;  // the level-1 sets are packed Hermitian rows (packed_herm.h)
    double inv_period, sweep;
    double p[4];
};

template <int N, int FID>
__global__ __launch_bounds__(256) void node_integrand_kernel(NodeArgs a, double2* __restrict__ values) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.nnodes) return;
    const double xx = a.x[k] * a.inv_period;
    double zr, zi, wr, wi;
    sincospi(2.0 * xx, &zi, &zr);
    CMat<N> H;
    if (a.pk) {
        const int F = (a.M - 1) / 2;
        series_point_pk<N>(as_const(a.src + a.parents[k] * (int64_t)Pk<N>::size(F)), F, zr, zi, H);
    } else {
        cptr_t c1 = as_const(a.src + a.parents[k] * ((int64_t)a.M * N * N));
        sincospi(2.0 * ((double)a.first * xx), &wi, &wr);
        series_lane<N>(c1, a.M, a.first, zr, zi, wr, wi, false, H);
    }
    double e[N];
    if constexpr (FID == ABZ_F_DOS_EIG) {
        herm_eig_values<N>(H, e);
    }
    double xk[ABZ_MAX_DIM] = {0.0, 0.0, 0.0};
    if constexpr (FID == ABZ_F_LINEAR_X) {
        xk[0] = a.x[k];
        for (int j = 1; j < a.d; ++j) xk[j] = a.tail[k * (a.d - 1) + (j - 1)];
    }
    double vr[MAXC], vi[MAXC];
    integrand_value<N, FID>(H, e, xk, a.d, a.p, a.sweep_arr ? a.sweep_arr[k] : a.sweep, vr, vi);
    constexpr int NC = NComp<FID>::template value<N>();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < a.ncomp) values[k * a.ncomp + c] = make_double2(vr[c], vi[c]);
    }
}

int launch_node_integrand(abz_ctx* ctx, const NodeEvalSpec& ns, double2* values_dev) {
    if (ns.nnodes == 0) return ABZ_OK;
    if (ns.n > 4) {
        GenSpec gs;
        gs.n = ns.n;
        gs.M = ns.M;
        gs.first = ns.first;
        gs.npt = 0;
        gs.d = ns.d;
        gs.period = ns.period;
        gs.src = ns.src;
        gs.grid = false;
        gs.parents = ns.parents;
        gs.gi = nullptr;
        gs.x = ns.x;
        gs.tab = nullptr;
        gs.deriv = false;
        gs.panels15 = ns.panels15;
        gs.herm = ns.herm;
        gs.nnodes = ns.nnodes;
        gs.Hplanes = PlaneView();
        gs.Eplanes = PlaneView();
        gs.Haos = nullptr;
        gs.Eaos = nullptr;
        gs.integrand = ns.integrand;
        for (int i = 0; i < 4; ++i) gs.params[i] = ns.params[i];
        gs.sweep_dev = nullptr;
        gs.sweep0 = ns.sweep;
        gs.n_sweep = 1;
        gs.sweep_per_node = ns.sweep_arr;
        gs.values = values_dev;
        return launch_gen_nodes(ctx, gs);
    }
    const int ncomp = integrand_ncomp(ns.integrand, ns.n, ns.d);
    if (ncomp < 0) {
        set_error("unknown integrand id %d", ns.integrand);
        return ABZ_ERR_ARG;
    }
    if ((ns.integrand == ABZ_F_LINEAR || ns.integrand == ABZ_F_LINEAR_X) && ns.n != 1) {
        set_error("ABZ_F_LINEAR(_X) needs a scalar (n = 1) series");
        return ABZ_ERR_ARG;
    }
    NodeArgs a;
    a.src = ns.src;
    a.parents = ns.parents;
    a.x = ns.x;
    a.tail = ns.tail;
    a.nnodes = ns.nnodes;
    a.M = ns.M;
    a.first = ns.first;
    a.d = ns.d;
    a.ncomp = ncomp;
    a.pk = ns.packed ? 1 : 0;
    a.inv_period = 1.0 / ns.period;
    a.sweep = ns.sweep;
    a.sweep_arr = ns.sweep_arr;
    for (int i = 0; i < 4; ++i) a.p[i] = ns.params[i];
    ProfScope ps(ctx, ABZ_K_EVAL);
    const unsigned blocks = (unsigned)cdiv(ns.nnodes, 256);
#define CASE(FID)                                                                                                  \
    case FID:                                                                                                      \
        switch (ns.n) {                                                                                            \
            case 1: hipLaunchKernelGGL((node_integrand_kernel<1, FID>), dim3(blocks), dim3(256), 0, ctx->stream, a, values_dev); break; \
            case 2: hipLaunchKernelGGL((node_integrand_kernel<2, FID>), dim3(blocks), dim3(256), 0, ctx->stream, a, values_dev); break; \
            case 3: hipLaunchKernelGGL((node_integrand_kernel<3, FID>), dim3(blocks), dim3(256), 0, ctx->stream, a, values_dev); break; \
            case 4: hipLaunchKernelGGL((node_integrand_kernel<4, FID>), dim3(blocks), dim3(256), 0, ctx->stream, a, values_dev); break; \
            default:                                                                                               \
                set_error("n = %d bands: only n <= 4 is built in this round", ns.n);                               \
                return ABZ_ERR_UNSUPPORTED;                                                                        \
        }                                                                                                          \
        break;
    switch (ns.integrand) {
        CASE(ABZ_F_ONE)
        CASE(ABZ_F_LINEAR)
        CASE(ABZ_F_LINEAR_X)
        CASE(ABZ_F_DOS)
        CASE(ABZ_F_TRGLOC)
        CASE(ABZ_F_GLOC)
        CASE(ABZ_F_DOS_EIG)
    }
#undef CASE
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// IAI innermost level on the device: adaptive GK(7,15) per 1-D integral, one half-wave each
// ------------------------------------------------------------------------------------------
// H = sum_m c1[m] (w z^m) with the coefficient set in LDS (every lane of a group reads the same address);
// HERM: upper triangle only, mirrored.
template <int N, bool HERM>
__device__ __forceinline__ void series_lane_lds(const double2* c1, int M, double zr, double zi, double wr, double wi,
                                                CMat<N>& H) {
#pragma unroll
    for (int a = 0; a < N; ++a) {
#pragma unroll
        for (int b = 0; b < N; ++b) {
            H.re[a][b] = 0.0;
            H.im[a][b] = 0.0;
        }
    }
    double pr = wr, pi = wi;
    for (int m = 0; m < M; ++m) {
        const double2* cm = c1 + m * (N * N);
#pragma unroll
        for (int b = 0; b < N; ++b) {
#pragma unroll
            for (int a = 0; a < N; ++a) {
                if (HERM && a > b) continue;
                const double2 c = cm[a + N * b];
                H.re[a][b] = fma(c.x, pr, H.re[a][b]);
                H.re[a][b] = fma(-c.y, pi, H.re[a][b]);
                if (!(HERM && a == b)) {
                    H.im[a][b] = fma(c.x, pi, H.im[a][b]);
                    H.im[a][b] = fma(c.y, pr, H.im[a][b]);
                }
            }
        }
        const double nr = fma(pr, zr, -(pi * zi));
        const double ni = fma(pr, zi, pi * zr);
        pr = nr;
        pi = ni;
    }
    if constexpr (HERM) {
#pragma unroll
        for (int b = 0; b < N; ++b) {
#pragma unroll
            for (int a = b + 1; a < N; ++a) {
                H.re[a][b] = H.re[b][a];
                H.im[a][b] = -H.im[b][a];
            }
        }
    }
}

// LDS doubles of one integral of inner_adaptive_kernel: the adaptive state + its coefficient set
// (+ ABZ_INNER_MAXSEG doubles: adapt_step_pair's mirror of the heap's error estimates)
__host__ __device__ inline int inner_group_stride(int ncomp, int mnn) { return inner_group_doubles(ncomp) + ABZ_INNER_MAXSEG + 2 * mnn; }

struct InnerArgs {
    const double2* src;
    const int64_t* slot;
    const double* lo;
    const double* hi;
    const double* atol;
    const double* tail;
    const double* sweep_arr;
    int64_t nint, maxevals;
    int M, first, d, ncomp, has_rtol;
    int pk;    // the level-1 sets are packed Hermitian rows (packed_herm.h): folded series, no seed phase
    double sc[16];  // sincospi_poly's coefficients (kernel arguments: scalar operands of the FMAs)
    double inv_period, sweep, rtol_user;
    double p[4];
    double2* I_out;
    double* E_out;
    int64_t* nev_out;
    int* status_out;
};

template <int N, int FID, bool HERM>
__global__ __launch_bounds__(256) void inner_adaptive_kernel(InnerArgs a) {
    extern __shared__ double lds_in[];
    constexpr int MS = ABZ_INNER_MAXSEG;
    constexpr int NC = NComp<FID>::template value<N>();  // >= a.ncomp; 1 for the scalar integrands
    const int group = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int nc = a.ncomp;
    const int MNN = a.pk ? Pk<N>::size((a.M - 1) / 2) : a.M * N * N;
    double* g = lds_in + (size_t)group * inner_group_stride(nc, MNN);
    double2* const cl = reinterpret_cast<double2*>(g + inner_group_doubles(nc) + MS);  // this integral's coefficients
    double* seg_a = g;
    double* seg_b = seg_a + MS;
    double* seg_E = seg_b + MS;
    gkc* seg_I = reinterpret_cast<gkc*>(seg_E + MS);
    gkc* vals = seg_I + (size_t)MS * nc;
    int* heap = reinterpret_cast<int*>(vals + (size_t)30 * nc);
    double* ctl = reinterpret_cast<double*>(heap + MS);
    const int64_t gstride = (int64_t)gridDim.x * 8;
    for (int64_t q0 = (int64_t)blockIdx.x * 8; q0 < a.nint; q0 += gstride) {
        const int64_t q = q0 + group;
        const bool live = q < a.nint;  // the whole group shares it; both groups of a wave loop together
        // ---- lane 0 state
        AdaptStateT<NC> st;
        bool done = !live;
        double tailv[ABZ_MAX_DIM] = {0.0, 0.0, 0.0};
        double swq = a.sweep;
        if (live) {
            if (a.sweep_arr) swq = a.sweep_arr[q];
            // the set stays in LDS for the whole adaptive loop (the nodes of every round re-read it)
            const double2* __restrict__ src = a.src + a.slot[q] * (int64_t)MNN;
            for (int idx = l; idx < MNN; idx += 32) cl[idx] = src[idx];
            if (FID == ABZ_F_LINEAR_X && a.tail)
                for (int j = 0; j < a.d - 1; ++j) tailv[j] = a.tail[q * (a.d - 1) + j];
            if (l == 0) adapt_init(st, a.atol[q], a.has_rtol != 0, a.rtol_user, a.lo[q], a.hi[q], ctl);
        }
        while (true) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int grp_done = __shfl(done ? 1 : 0, (int)(threadIdx.x & 32u), 64);  // leader lane of this group
            // all lanes of the WAVE must agree to leave: ballot over both groups' leaders
            const unsigned long long alive = __ballot(!grp_done);
            if (alive == 0ull) break;
            if (!grp_done) {
                // ---- evaluate the nodes of the pending panels: lanes 0-14 panel 0, 16-30 panel 1
                const int np = (int)ctl[0];
                const int pnl = l >> 4, i = l & 15;
                if (pnl < np && i < 15) {
                    const double pa = ctl[1 + 2 * pnl], pb = ctl[2 + 2 * pnl];
                    const double x = gk15_node(pa, pb, i);
                    const double xx = x * a.inv_period;
                    double zr, zi, wr, wi;
                    CMat<N> H;
                    sincospi_poly(a.sc, 2.0 * xx, zi, zr);  // the same phases as inner_adaptive_wave_kernel: the two kernels agree to the bit
                    if (HERM && a.pk) {
                        series_point_pk<N>((const double2*)cl, (a.M - 1) / 2, zr, zi, H);
                    } else {
                        sincospi_poly(a.sc, 2.0 * ((double)a.first * xx), wi, wr);
                        series_lane_lds<N, HERM>(cl, a.M, zr, zi, wr, wi, H);
                    }
                    double e[N];
                    if constexpr (FID == ABZ_F_DOS_EIG) {
                        herm_eig_values<N>(H, e);
                    }
                    double xk[ABZ_MAX_DIM] = {x, tailv[0], tailv[1]};
                    double vr[MAXC], vi[MAXC];
                    if constexpr (HERM && (N == 2 || N == 3) && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC)) {
                        // Hermitian H: resolvent trace from the real characteristic polynomial
                        CharPolyH cp;
                        if constexpr (N == 3)
                            charpoly_init_h3(H.re[0][0], H.re[1][1], H.re[2][2], H.re[0][1], H.im[0][1], H.re[0][2], H.im[0][2],
                                             H.re[1][2], H.im[1][2], cp);
                        else
                            charpoly_init_h2(H.re[0][0], H.re[1][1], H.re[0][1], H.im[0][1], cp);
                        double tr, ti;
                        charpoly_trace_h<N, true>(cp, swq, a.p[0], a.p[0] * a.p[0], 2.0 * a.p[0], tr, ti);
                        vr[0] = (FID == ABZ_F_DOS) ? -ti * 0.31830988618379067153776752674503 : tr;
                        vi[0] = (FID == ABZ_F_DOS) ? 0.0 : ti;
                    } else {
                        integrand_value<N, FID>(H, e, xk, a.d, a.p, swq, vr, vi);
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if (c < nc) {
                            vals[(size_t)(pnl * 15 + i) * nc + c].re = vr[c];
                            vals[(size_t)(pnl * 15 + i) * nc + c].im = vi[c];
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (!grp_done && l == 0) {
                InnerOut out;
                out.I = a.I_out + q * nc;
                out.E = a.E_out + q;
                out.nev = a.nev_out + q;
                out.status = a.status_out + q;
                done = adapt_step<true, NC>(st, nc, seg_a, seg_b, seg_E, seg_I, vals, heap, ctl, a.maxevals, out);
            }
        }
    }
}

// ---- the same loop with the adaptive state in REGISTERS: one wavefront per integral (scalar integrands) ----
// inner_adaptive_kernel keeps an integral's segments and heap in LDS and lets one lane walk them: in-kernel stamps on the
// reference example's solve (SVO, eta = 0.01: launches of 16 ... 4096 integrals, i.e. a nearly empty chip where only the
// length of a round counts) gave 2 490 cycles for the 30 node values of a round and 5 090 for that serial step -- a chain of
// dependent LDS round trips (ctl, the fifteen values, the heap levels, the popped parent).  Here the state never leaves the
// wave's registers: lane j holds heap position j (error, segment slot) and segment slot j (a, b, I); everything the step
// decides is wave-uniform, so heap levels are v_readlane with scalar indices (a few cycles each), writes are compare-selects by the
// lane index (a v_writelane needs its value and lane in scalar registers: three times the instructions) and the branches
// are scalar.  Lanes 0 and 1 apply the GK rule to the two pending panels, whose fifteen values they pull from
// their owners' registers (ds_bpermute, no LDS storage).  The arithmetic and its order are those of adapt_step: identical
// (I, E), identical heap decisions (DataStructures.jl percolate semantics), identical counts.  The segment store is the
// wave (lanes 0 ... ABZ_INNER_MAXSEG - 1, the LDS kernel's capacity: an integral that needs more is redone by the host loop,
// whose node kernel evaluates the integrand by another formula -- the same integrals must take that road in both kernels).
// value of `v` in lane C of this lane's 16-lane row (v_mov_b64_dpp row_newbcast:C, gfx90a+)
template <int C>
__device__ __forceinline__ double row16_bcast(double v) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass of hipcc only knows the 32-bit signature of the builtin
    const long long x = __builtin_bit_cast(long long, v);
    return __builtin_bit_cast(double, (long long)__builtin_amdgcn_mov_dpp(x, 0x150 + C, 0xf, 0xf, false));
#else
    return v;
#endif
}
template <int... T>
__device__ __forceinline__ void gk_row_values(double vr, double vi, gkc (&rv)[15], std::integer_sequence<int, T...>) {
    ((rv[T].re = row16_bcast<T>(vr), rv[T].im = row16_bcast<T>(vi)), ...);
}
__device__ __forceinline__ double rl_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int rl_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

template <int N, int FID, bool HERM>
__global__ __launch_bounds__(256) void inner_adaptive_wave_kernel(InnerArgs a) {
    static_assert(NComp<FID>::template value<N>() == 1, "one complex value per node");
    extern __shared__ double lds_iw[];  // [4 waves][MNN] complex: the integrals' coefficient sets, nothing else
    constexpr int MS = ABZ_INNER_MAXSEG;  // like the LDS kernel: the same integrals go to the host loop whichever kernel runs
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int MNN = a.pk ? Pk<N>::size((a.M - 1) / 2) : a.M * N * N;
    double2* const cl = reinterpret_cast<double2*>(lds_iw) + (size_t)wave * MNN;
    for (int64_t q = (int64_t)blockIdx.x * 4 + wave; q < a.nint; q += (int64_t)gridDim.x * 4) {
        wave_lds_sync();  // the previous integral's readers of cl are done
        {
            const double2* __restrict__ src = a.src + a.slot[q] * (int64_t)MNN;
            for (int idx = l; idx < MNN; idx += 64) cl[idx] = src[idx];
        }
        wave_lds_sync();
        const double swq = a.sweep_arr ? a.sweep_arr[q] : a.sweep;
        // per-lane state: heap position l (error, segment slot); segment slot l (a, b, I) and, when the slot's panel had its
        // halves evaluated AHEAD of its pop, their GK sums (rdy)
        double hE = 0.0, sA = 0.0, sB = 0.0, sIr = 0.0, sIi = 0.0;
        double cE1 = 0.0, cI1r = 0.0, cI1i = 0.0, cE2 = 0.0, cI2r = 0.0, cI2i = 0.0;
        int hS = 0, rdy = 0;
        // wave-uniform state (AdaptStateT<1> + ctl of the LDS kernel)
        int nseg = 0, nheap = 0, popped = -1, status = 0;
        bool first = true;
        double E = 0.0, Ir = 0.0, Ii = 0.0, parE = 0.0, parIr = 0.0, parIi = 0.0;
        long long numevals = 0;
        const double at_in = a.atol[q];
        const double atol = at_in >= 0.0 ? at_in : 0.0;
        const double rtol = a.has_rtol ? a.rtol_user : ((at_in > 0.0) ? 0.0 : 1.4901161193847656e-08);  // sqrt(eps)
        int np = 1;
        double a1 = a.lo[q], b1 = a.hi[q], a2 = 0.0, b2 = 0.0;
        // A second panel per round (rows 2 and 3 of the wave, idle otherwise): the largest panel left in the heap once the
        // current one is out -- the next pop unless a half of the current panel overtakes it.  Its halves' sums wait in its
        // segment's lane; when the panel reaches the top its pop is replayed without a round of node evaluations, and if the
        // loop ends first they are dropped (numevals counts pops).  The pops, their order and every sum are those of the
        // one-panel-per-round loop.
        int qslot = -1;
        double qa = 0.0, qm = 0.0, qb = 0.0;
        bool finished = false;
        while (!finished) {
            // ---- the nodes of the pending panels: row r of the wave = panel r (rows 0, 1: the popped panel's halves; 2, 3: the
            // speculative one's)
            const int pnl = l >> 4, i = l & 15;
            const bool row_on = pnl < np || (pnl >= 2 && qslot >= 0);
            const double ra = pnl == 0 ? a1 : (pnl == 1 ? a2 : (pnl == 2 ? qa : qm));
            const double rb = pnl == 0 ? b1 : (pnl == 1 ? b2 : (pnl == 2 ? qm : qb));
            double vr = 0.0, vi = 0.0;
            if (row_on && i < 15) {
                const double x = gk15_node(ra, rb, i);
                const double xx = x * a.inv_period;
                double zr, zi, wr, wi;
                CMat<N> H;
                sincospi_poly(a.sc, 2.0 * xx, zi, zr);  // (the library routine: 880 of a round's 5 600 cycles; this one 1 ulp, a quarter of that)
                if (HERM && a.pk) {
                    const int F = (a.M - 1) / 2;
                    if (F == 5)  // 11 coefficients per variable (Wannier90 models on R in [-5, 5]^d, the reference's example)
                        series_point_pk<N, const double2*, 5>((const double2*)cl, 5, zr, zi, H);
                    else
                        series_point_pk<N>((const double2*)cl, F, zr, zi, H);
                } else {
                    sincospi_poly(a.sc, 2.0 * ((double)a.first * xx), wi, wr);
                    series_lane_lds<N, HERM>(cl, a.M, zr, zi, wr, wi, H);
                }
                double e[N];
                if constexpr (FID == ABZ_F_DOS_EIG) {
                    herm_eig_values<N>(H, e);
                }
                double xk[ABZ_MAX_DIM] = {x, 0.0, 0.0};
                double wr_[MAXC], wi_[MAXC];
                if constexpr (HERM && (N == 2 || N == 3) && (FID == ABZ_F_DOS || FID == ABZ_F_TRGLOC)) {
                    // Hermitian H: resolvent trace from the real characteristic polynomial
                    CharPolyH cp;
                    if constexpr (N == 3)
                        charpoly_init_h3(H.re[0][0], H.re[1][1], H.re[2][2], H.re[0][1], H.im[0][1], H.re[0][2], H.im[0][2],
                                         H.re[1][2], H.im[1][2], cp);
                    else
                        charpoly_init_h2(H.re[0][0], H.re[1][1], H.re[0][1], H.im[0][1], cp);
                    double tr, ti;
                    charpoly_trace_h<N, true>(cp, swq, a.p[0], a.p[0] * a.p[0], 2.0 * a.p[0], tr, ti);
                    wr_[0] = (FID == ABZ_F_DOS) ? -ti * 0.31830988618379067153776752674503 : tr;
                    wi_[0] = (FID == ABZ_F_DOS) ? 0.0 : ti;
                } else {
                    integrand_value<N, FID>(H, e, xk, a.d, a.p, swq, wr_, wi_);
                }
                vr = wr_[0];
                vi = wi_[0];
            }
            // ---- the GK rule: a panel's fifteen values sit in the first fifteen lanes of a 16-lane row, so every lane of the row
            // gets them by row broadcasts on the VALU (v_mov_b64_dpp row_newbcast: one instruction per value, no LDS crossbar)
            // and applies the rule of its row's panel
            gkc rv[15];
            gk_row_values(vr, vi, rv, std::make_integer_sequence<int, 15>());
            gkc Il;
            const double El = gk15_rule(rv, 1, ra, rb, &Il);
            const double E1 = rl_f64(El, 0), I1r = rl_f64(Il.re, 0), I1i = rl_f64(Il.im, 0);
            const double E2 = rl_f64(El, 16), I2r = rl_f64(Il.re, 16), I2i = rl_f64(Il.im, 16);
            if (qslot >= 0) {  // the speculative panel's halves: parked in its segment's lane
                const double Q1 = rl_f64(El, 32), Q1r = rl_f64(Il.re, 32), Q1i = rl_f64(Il.im, 32);
                const double Q2 = rl_f64(El, 48), Q2r = rl_f64(Il.re, 48), Q2i = rl_f64(Il.im, 48);
                const bool mine = l == qslot;
                cE1 = mine ? Q1 : cE1;
                cI1r = mine ? Q1r : cI1r;
                cI1i = mine ? Q1i : cI1i;
                cE2 = mine ? Q2 : cE2;
                cI2r = mine ? Q2r : cI2r;
                cI2i = mine ? Q2i : cI2i;
                rdy = mine ? 1 : rdy;
                qslot = -1;
            }
            // ---- the step, on wave-uniform values (adapt_step's operations in adapt_step's order): first the panel(s) just
            // evaluated, then every pop whose halves are already known
            double n1E = E1, n1r = I1r, n1i = I1i, n2E = E2, n2r = I2r, n2i = I2i;  // the halves being inserted
            double na1 = a1, nb1 = b1, na2 = a2, nb2 = b2;
            while (true) {
                int s1, s2 = -1;
                if (popped >= 0)
                    s1 = popped;  // the popped parent's slot is reused for the first child
                else
                    s1 = nseg++;
                if (s1 >= MS) {
                    status = 1;
                    s1 = MS - 1;
                }
                if (!first) {
                    s2 = nseg++;
                    if (s2 >= MS) {
                        status = 1;
                        s2 = MS - 1;
                    }
                }
                {
                    const bool m1 = l == s1;
                    sA = m1 ? na1 : sA;
                    sB = m1 ? nb1 : sB;
                    sIr = m1 ? n1r : sIr;
                    sIi = m1 ? n1i : sIi;
                    rdy = m1 ? 0 : rdy;
                }
                if (first) {
                    first = false;
                    Ir = n1r;
                    Ii = n1i;
                    E = n1E;
                    numevals = 15;
                    hS = l == 0 ? s1 : hS;
                    hE = l == 0 ? n1E : hE;
                    nheap = 1;
                } else {
                    {
                        const bool m2 = l == s2;
                        sA = m2 ? na2 : sA;
                        sB = m2 ? nb2 : sB;
                        sIr = m2 ? n2r : sIr;
                        sIi = m2 ? n2i : sIi;
                        rdy = m2 ? 0 : rdy;
                    }
                    {
#pragma clang fp contract(off)
                        Ir = ((Ir - parIr) + n1r) + n2r;
                        Ii = ((Ii - parIi) + n1i) + n2i;
                        E = ((E - parE) + n1E) + n2E;
                    }
                    for (int t = 0; t < 2; ++t) {  // heappush (percolate_up)
                        const int xs = t == 0 ? s1 : s2;
                        const double Ex = t == 0 ? n1E : n2E;
                        int h = nheap++;
                        while (h > 0) {
                            const int j = (h - 1) / 2;
                            const double Ej = rl_f64(hE, j);
                            if (!(Ej < Ex)) break;
                            {
                                const int hj = rl_i32(hS, j);
                                hS = l == h ? hj : hS;
                                hE = l == h ? Ej : hE;
                            }
                            h = j;
                        }
                        hS = l == h ? xs : hS;
                        hE = l == h ? Ex : hE;
                    }
                }
                double tol = atol;
                if (rtol != 0.0) {  // (rtol = 0: max(atol, 0 * |I|) = atol, no square root)
#pragma clang fp contract(off)
                    const double t1 = Ir * Ir, t2 = Ii * Ii;
                    const double t3 = t1 + t2;
                    const double nrm = sqrt(0.0 + t3);
                    tol = fmax(atol, rtol * nrm);
                }
                if (!(E > tol && numevals < a.maxevals && status == 0)) {
                    finished = true;
                    break;
                }
                // heappop: root out, last to root, percolate_down
                const int xs = rl_i32(hS, 0);
                parE = rl_f64(hE, 0);
                const int nh = --nheap;
                const int y = rl_i32(hS, nh);
                const double Ey = rl_f64(hE, nh);
                parIr = rl_f64(sIr, xs);
                parIi = rl_f64(sIi, xs);
                const double pa = rl_f64(sA, xs), pb = rl_f64(sB, xs);
                if (nh > 0) {
                    int h = 0;
                    while (true) {
                        const int lc = 2 * h + 1;
                        if (lc >= nh) break;
                        const int rc = lc + 1;
                        const double Elc = rl_f64(hE, lc);
                        bool left = true;
                        double Ej = Elc;
                        if (rc < nh) {
                            const double Erc = rl_f64(hE, rc);
                            left = Erc < Elc;
                            Ej = left ? Elc : Erc;
                        }
                        if (!(Ey < Ej)) break;
                        const int j = left ? lc : rc;
                        {
                            const int hj = rl_i32(hS, j);
                            hS = l == h ? hj : hS;
                            hE = l == h ? Ej : hE;
                        }
                        h = j;
                    }
                    hS = l == h ? y : hS;
                    hE = l == h ? Ey : hE;
                }
                popped = xs;
                numevals += 30;
                const double mid = (pa + pb) / 2;
                na1 = pa;
                nb1 = mid;
                na2 = mid;
                nb2 = pb;
                if (rl_i32(rdy, xs) != 0) {  // its halves were evaluated ahead: replay the pop at once
                    n1E = rl_f64(cE1, xs);
                    n1r = rl_f64(cI1r, xs);
                    n1i = rl_f64(cI1i, xs);
                    n2E = rl_f64(cE2, xs);
                    n2r = rl_f64(cI2r, xs);
                    n2i = rl_f64(cI2i, xs);
                    continue;
                }
                // the next round evaluates this panel's halves ...
                np = 2;
                a1 = pa;
                b1 = mid;
                a2 = mid;
                b2 = pb;
                // ... and those of the largest panel left: the likely next pop (its lanes cost the wave nothing)
                if (nheap > 0) {
                    const int top = rl_i32(hS, 0);
                    if (rl_i32(rdy, top) == 0) {
                        qslot = top;
                        qa = rl_f64(sA, top);
                        qb = rl_f64(sB, top);
                        qm = (qa + qb) / 2;
                    }
                }
                break;
            }
        }
        {  // re-sum over the heap in storage order (QuadGK does this after adapt)
#pragma clang fp contract(off)
            const int s0 = rl_i32(hS, 0);
            Ir = rl_f64(sIr, s0);
            Ii = rl_f64(sIi, s0);
            E = rl_f64(hE, 0);
            for (int h = 1; h < nheap; ++h) {
                const int sh = rl_i32(hS, h);
                Ir = Ir + rl_f64(sIr, sh);
                Ii = Ii + rl_f64(sIi, sh);
                E = E + rl_f64(hE, h);
            }
        }
        if (l == 0) {
            a.I_out[q] = make_double2(Ir, Ii);
            a.E_out[q] = E;
            a.nev_out[q] = numevals;
            a.status_out[q] = status;
        }
    }
}

// ---- panels of the level above the innermost one (abz_internal.h: PanelNodesSpec / PanelRuleSpec) ----
// One block per node of a panel: the node's coordinate (gk15_node), the limits / tolerance / swept value of the innermost
// integral beneath it (Lims::fix + range of iai_host.cpp, operation for operation), its M phases (phase_kernel's expression)
// and the contraction of its parent's coefficient set (contract_kernel's sum, term for term) -- three launches of the node
// path in one, fed by 40 B per PANEL that the kernel reads where the host wrote them (pinned, device-visible memory).
__global__ void panel_contract_kernel(PanelNodesSpec a, const double2* __restrict__ src, int64_t slot_elems, int M, int first,
                                      double inv_period, double2* __restrict__ out, int64_t Lrow) {
    extern __shared__ double2 pc_phs[];  // [M]
    const int64_t t = blockIdx.x;
    const int64_t p = t / 15;
    const int i = (int)(t - 15 * p);
    double x;
    {
#pragma clang fp contract(off)
        x = gk15_node(a.p_a[p], a.p_b[p], i);
        if (threadIdx.x == 0 && blockIdx.y == 0) {
            // Lims::fix(L, x) + range(1): CubicLimits (a0, b0); TetrahedralLimits s = x / a[L - 1], (0, a[0] * s)
            double lo = a.a0, hi = a.b0;
            if (a.lims_kind == ABZ_LIMS_TETRAHEDRAL) {
                const double sc = x / a.aL;
                lo = 0.0;
                hi = a.a0 * sc;
            }
            const double at = a.p_at[p];
            a.n_slot[t] = t;
            a.n_lo[t] = lo;
            a.n_hi[t] = hi;
            a.n_at[t] = at >= 0.0 ? at / (hi - lo) : -1.0;  // ref src/fourier.jl:479-480
            a.n_sw[t] = a.p_sw[p];
        }
    }
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        double c, sn;
        sincospi(2.0 * ((double)(first + m) * x * inv_period), &sn, &c);
        pc_phs[m] = make_double2(c, sn);
    }
    __syncthreads();
    const int64_t l = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    if (l >= Lrow) return;
    const double2* __restrict__ sp = src + a.p_slot[p] * slot_elems + l;
    double ar = 0.0, ai = 0.0;
    for (int m = 0; m < M; ++m) {
        const double2 c = sp[(int64_t)m * Lrow];
        const double2 ph = pc_phs[m];
        ar = fma(c.x, ph.x, ar);
        ar = fma(-c.y, ph.y, ar);
        ai = fma(c.x, ph.y, ai);
        ai = fma(c.y, ph.x, ai);
    }
    out[t * Lrow + l] = make_double2(ar, ai);
}

int launch_panel_contract(abz_ctx* ctx, const PanelNodesSpec& ps, const double2* src, int64_t slot_elems, int M, int first, double period,
                          double2* out, int64_t Lrow) {
    if (ps.npanels == 0) return ABZ_OK;
    ProfScope pf(ctx, ABZ_K_CONTRACT);
    const int bs = Lrow <= 64 ? 64 : (Lrow <= 128 ? 128 : 256);
    const int64_t gy = cdiv(Lrow, bs);
    if (gy > 65535) {
        set_error("panel_contract: row length %lld too large", (long long)Lrow);
        return ABZ_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(panel_contract_kernel, dim3((unsigned)(15 * ps.npanels), (unsigned)gy), dim3(bs), sizeof(double2) * (size_t)M, ctx->stream,
                       ps, src, slot_elems, M, first, 1.0 / period, out, Lrow);
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

__global__ __launch_bounds__(256) void panel_rule_kernel(PanelRuleSpec a) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.npanels) return;
    // the shared rule on the panel's fifteen innermost integrals (node-major, ncomp components each): what the host's
    // gk15_evalrule computes from the same numbers
    const gkc* fv = reinterpret_cast<const gkc*>(a.n_I + (size_t)(15 * p) * a.ncomp);
    a.p_E[p] = gk15_rule(fv, a.ncomp, a.p_a[p], a.p_b[p], reinterpret_cast<gkc*>(a.p_I + (size_t)p * a.ncomp));
    int64_t nev = 0;
    int st = 0;
    for (int i = 0; i < 15; ++i) {
        nev += a.n_nev[15 * p + i];
        st |= a.n_status[15 * p + i];
    }
    a.p_nev[p] = nev;
    a.p_status[p] = st;
}

int launch_panel_rule(abz_ctx* ctx, const PanelRuleSpec& ps) {
    if (ps.npanels == 0) return ABZ_OK;
    hipLaunchKernelGGL(panel_rule_kernel, dim3((unsigned)cdiv(ps.npanels, 256)), dim3(256), 0, ctx->stream, ps);
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

bool inner_adaptive_supported(int n, int M, int integrand) {
    const int nc = integrand_ncomp(integrand, n, 3);
    // + the coefficient set of every integral in flight
    return n >= 1 && n <= 4 && nc > 0 && nc <= MAXC && M * n * n <= EVAL_MAX_MNN &&
           sizeof(double) * (size_t)inner_group_stride(nc, M * n * n) * 8 <= 150 * 1024;
}

int launch_inner_adaptive(abz_ctx* ctx, const InnerSpec& is) {
    if (is.nint == 0) return ABZ_OK;
    const int ncomp = integrand_ncomp(is.integrand, is.n, is.d);
    if (ncomp < 0) {
        set_error("unknown integrand id %d", is.integrand);
        return ABZ_ERR_ARG;
    }
    if ((is.integrand == ABZ_F_LINEAR || is.integrand == ABZ_F_LINEAR_X) && is.n != 1) {
        set_error("ABZ_F_LINEAR(_X) needs a scalar (n = 1) series");
        return ABZ_ERR_ARG;
    }
    InnerArgs a;
    a.src = is.src;
    a.slot = is.slot;
    a.lo = is.lo;
    a.hi = is.hi;
    a.atol = is.atol;
    a.tail = is.tail;
    a.sweep_arr = is.sweep_arr;
    a.nint = is.nint;
    a.maxevals = is.maxevals;
    a.M = is.M;
    a.first = is.first;
    a.d = is.d;
    a.ncomp = ncomp;
    a.has_rtol = is.has_rtol ? 1 : 0;
    // (a polynomial sincospi for the node phases measured 31.3 against 32.0 ms on the SVO full-BZ solve: the phases are
    // not what bounds a round; the library routine stays)
    a.pk = (is.packed && is.herm) ? 1 : 0;
    for (int i = 0; i < 16; ++i) a.sc[i] = kSinCosPiCoef[i];
    a.inv_period = 1.0 / is.period;
    a.sweep = is.sweep;
    a.rtol_user = is.rtol_user;
    for (int i = 0; i < 4; ++i) a.p[i] = is.params[i];
    a.I_out = is.I_out;
    a.E_out = is.E_out;
    a.nev_out = is.nev_out;
    a.status_out = is.status_out;
    ProfScope ps(ctx, ABZ_K_EVAL);
    const int mnn = a.pk ? (int)packed_row_elems(is.n, is.M) : is.M * is.n * is.n;
    if (mnn > EVAL_MAX_MNN) {
        set_error("inner adaptive kernel: %d coefficients per line exceed the LDS budget", mnn);
        return ABZ_ERR_UNSUPPORTED;
    }
    // one complex value per node: the adaptive state lives in the registers of one wavefront per integral
    // (inner_adaptive_wave_kernel); ABZ_ADAPT_PAIR=0 (tests) and the multi-component integrands: the LDS kernel
    const bool wave_state = ncomp == 1 && abz_switch(SW_ADAPT_PAIR) != 0;
    const size_t lds = wave_state ? sizeof(double2) * (size_t)mnn * 4 : sizeof(double) * (size_t)inner_group_stride(ncomp, mnn) * 8;
    const unsigned blocks = (unsigned)std::min<int64_t>(cdiv(is.nint, wave_state ? 4 : 8), 256 * 16);
#define LAUNCH_INNER_LDS(NN, FID, HH)                                                                                 \
    {                                                                                                                 \
        if (lds > 48 * 1024)                                                                                          \
            ABZ_HIP(hipFuncSetAttribute((const void*)inner_adaptive_kernel<NN, FID, HH>,                              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
        hipLaunchKernelGGL((inner_adaptive_kernel<NN, FID, HH>), dim3(blocks), dim3(256), lds, ctx->stream, a);       \
    }
#define LAUNCH_INNER2(NN, FID, HH)                                                                                    \
    {                                                                                                                 \
        if constexpr (NComp<FID>::template value<NN>() == 1) {                                                        \
            if (wave_state) {                                                                                         \
                hipLaunchKernelGGL((inner_adaptive_wave_kernel<NN, FID, HH>), dim3(blocks), dim3(256), lds, ctx->stream, a); \
            } else                                                                                                    \
                LAUNCH_INNER_LDS(NN, FID, HH)                                                                         \
        } else                                                                                                        \
            LAUNCH_INNER_LDS(NN, FID, HH)                                                                             \
    }
#define LAUNCH_INNER(NN, FID)             \
    if (is.herm) {                        \
        LAUNCH_INNER2(NN, FID, true)      \
    } else {                              \
        LAUNCH_INNER2(NN, FID, false)     \
    }
#define CASE(FID)                                                                                                     \
    case FID:                                                                                                         \
        switch (is.n) {                                                                                               \
            case 1: LAUNCH_INNER(1, FID) break;                                                                       \
            case 2: LAUNCH_INNER(2, FID) break;                                                                       \
            case 3: LAUNCH_INNER(3, FID) break;                                                                       \
            case 4: LAUNCH_INNER(4, FID) break;                                                                       \
            default: set_error("inner adaptive kernel: n = %d not supported", is.n); return ABZ_ERR_UNSUPPORTED;      \
        }                                                                                                             \
        break;
    switch (is.integrand) {
        CASE(ABZ_F_ONE)
        CASE(ABZ_F_LINEAR)
        CASE(ABZ_F_LINEAR_X)
        CASE(ABZ_F_DOS)
        CASE(ABZ_F_TRGLOC)
        CASE(ABZ_F_GLOC)
        CASE(ABZ_F_DOS_EIG)
    }
#undef CASE
#undef LAUNCH_INNER
#undef LAUNCH_INNER2
#undef LAUNCH_INNER_LDS
    ABZ_HIP(hipGetLastError());
    return ABZ_OK;
}

}  // namespace abz
