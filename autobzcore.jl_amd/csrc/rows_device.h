// Device helpers of the row-layout kernels (NP = 8 / 16 / 32 lanes per node, lane r owning row r of a matrix in registers):
// coefficient staging and series rows, broadcasts / sums inside a node's lanes, Householder tridiagonalisation and the
// eigenvalues of the tridiagonal.  Shared by kernels_generic.hip (rule builds, sweeps, IAI panels) and kernels_ggr_rows.hip
// (the fused GGR build of 5...32 bands).  gfx950 only.
#pragma once
#include <utility>

#include "abz_internal.h"

namespace abz {

__device__ __forceinline__ int64_t view_off(const PlaneView& v, int64_t k) {
    const int64_t line = k / v.line_len;
    return line * v.tile + (k - line * v.line_len);
}

__device__ __forceinline__ double rsqrt_nr(double x) {  // 1/sqrt(x), x in the normal range: estimate + 2 Newton steps
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(fma(-hx * y, y, 0.5), y, y);
    y = fma(fma(-hx * y, y, 0.5), y, y);
    return y;
}

__device__ __forceinline__ double rcp_nr(double x) {  // 1/x, x in the normal range: estimate + 2 Newton steps
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// Eight LDS reads in flight, then ONE wait: the empty asm needs all eight values in registers at this
// point, so the reads are issued back to back instead of one `s_waitcnt` per read (what the scheduler
// produces on its own under this kernel's register pressure).
__device__ __forceinline__ void pin8(double2 (&u)[8]) {
    asm volatile(""
                 : "+v"(u[0].x), "+v"(u[0].y), "+v"(u[1].x), "+v"(u[1].y), "+v"(u[2].x), "+v"(u[2].y), "+v"(u[3].x),
                   "+v"(u[3].y), "+v"(u[4].x), "+v"(u[4].y), "+v"(u[5].x), "+v"(u[5].y), "+v"(u[6].x), "+v"(u[6].y),
                   "+v"(u[7].x), "+v"(u[7].y));
}

// Row r (lane r of the node's NP lanes) of inv((sw + i eta) I - H(x)) into ar/ai; `coef` = the staged
// coefficient set.
// PAD: the set is staged as [M][NP*NP] with zeros outside the n x n block, so every loop runs to NP with
// no condition on n (the padding block of A is the identity and stays decoupled: its columns are exact
// zeros in the real rows).  !PAD: layout [M][n*n], loops guarded by (uniform) comparisons with n.
// -H(x) row r from the staged set: phases w z^m from their seeds (pr, pi) and z
template <int NP, bool PAD>
__device__ __forceinline__ void panel_series_row(const double2* coef, int n, int M, double zr, double zi, double pr,
                                                 double pi, int r, double (&ar)[NP], double (&ai)[NP]) {
    const int ld = PAD ? NP : n;   // column stride of a staged block
    const int nn = ld * ld;
    const int rr = (PAD || r < n) ? r : n - 1;  // !PAD: padded rows read a valid row and are overwritten later
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        ar[j] = 0.0;
        ai[j] = 0.0;
    }
    for (int m = 0; m < M; ++m) {
        const double2* __restrict__ cm = coef + (size_t)m * nn + rr;
        if constexpr (PAD) {
#pragma unroll
            for (int j0 = 0; j0 < NP; j0 += 8) {
                double2 c[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = cm[ld * (j0 + j)];
                pin8(c);
#pragma unroll
                for (int j = 0; j < 8; ++j) {  // A = z I - H: accumulate -H
                    ar[j0 + j] = fma(-c[j].x, pr, ar[j0 + j]);
                    ar[j0 + j] = fma(c[j].y, pi, ar[j0 + j]);
                    ai[j0 + j] = fma(-c[j].x, pi, ai[j0 + j]);
                    ai[j0 + j] = fma(-c[j].y, pr, ai[j0 + j]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (j < n) {  // uniform
                    const double2 c = cm[ld * j];
                    ar[j] = fma(-c.x, pr, ar[j]);
                    ar[j] = fma(c.y, pi, ar[j]);
                    ai[j] = fma(-c.x, pi, ai[j]);
                    ai[j] = fma(-c.y, pr, ai[j]);
                }
            }
        }
        const double nr = pr * zr - pi * zi, ni = pr * zi + pi * zr;
        pr = nr;
        pi = ni;
    }
}

// The same for a CHUNK of the set (unpadded layout [mc][n*n], blocks m0 ... m0 + mc - 1 of the M): the row accumulates over
// the chunks of a set that does not fit the LDS whole, the phase (pr, pi) = w z^m0 is carried from chunk to chunk.
template <int NP>
__device__ __forceinline__ void panel_series_row_chunk(const double2* coef, int n, int mc, double zr, double zi, double& pr, double& pi, int r,
                                                       double (&ar)[NP], double (&ai)[NP]) {
    const int nn = n * n;
    const int rr = r < n ? r : n - 1;
    for (int m = 0; m < mc; ++m) {
        const double2* __restrict__ cm = coef + (size_t)m * nn + rr;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (j < n) {  // uniform
                const double2 c = cm[n * j];
                ar[j] = fma(-c.x, pr, ar[j]);
                ar[j] = fma(c.y, pi, ar[j]);
                ai[j] = fma(-c.x, pi, ai[j]);
                ai[j] = fma(-c.y, pr, ai[j]);
            }
        }
        const double nr = pr * zr - pi * zi, ni = pr * zi + pi * zr;
        pr = nr;
        pi = ni;
    }
}

// Broadcast of lane C of every 16-lane row on the VALU: ONE `v_mov_b64_dpp ... row_newbcast:C` per double (DPP on 64-bit
// operands exists for row_newbcast only, gfx90a+; the destination's previous value is undefined, so nothing initialises
// it -- the first version, `update_dpp(0, ...)` on the two halves, cost four instructions per double).
template <int C>
__device__ __forceinline__ double row16_bcast_dpp(double v) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass of hipcc only knows the 32-bit signature of the builtin
    const long long x = __builtin_bit_cast(long long, v);
    return __builtin_bit_cast(double, (long long)__builtin_amdgcn_mov_dpp(x, 0x150 + C, 0xf, 0xf, false));
#else
    return v;
#endif
}

// value of `v` in lane C of this lane's NP-lane group.  NP = 16: the DPP row broadcast above (config 5: 11.2 -> 10.5 s
// against carrying the pivot rows half by swizzle, half by two 32-bit DPP moves).  Other group sizes: `ds_swizzle_b32`,
// bit-mask mode: lane' = (lane & ~(NP-1)) | C inside each half wave -- the LDS crossbar without an LDS access, 2.2 clk
// per dword against 14 clk for every `ds_write_b128` of a row published through memory (tools/micro/ldstest.hip)
template <int NP, int C>
__device__ __forceinline__ double group_bcast(double v) {
    if constexpr (NP == 16) return row16_bcast_dpp<C>(v);
    if constexpr (NP == 8) {
        // two nodes share a DPP row: lanes 0...7 take lane C, lanes 8...15 lane C + 8 -- two `v_mov_b64_dpp row_newbcast`
        // with bank masks (a bank = four lanes of the row) on the VALU instead of two `ds_swizzle_b32` through the LDS
        // crossbar and their wait (round 5: the 5...8-band kernels cost 15x a 4-band one per node)
#if defined(__HIP_DEVICE_COMPILE__)
        const long long x = __builtin_bit_cast(long long, v);
        long long t = __builtin_amdgcn_update_dpp(x, x, 0x150 + C, 0xf, 0x3, false);
        t = __builtin_amdgcn_update_dpp(t, x, 0x150 + C + 8, 0xf, 0xc, false);
        return __builtin_bit_cast(double, t);
#else
        return v;
#endif
    }
    constexpr int pattern = ((32 - NP) & 0x1f) | (C << 5);  // and_mask | or_mask << 5, xor_mask 0
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pattern);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pattern);
    return __hiloint2double(hi, lo);
}

// stage one coefficient set [M][n*n] into LDS, zero-padded to [M][NP*NP] when PAD
template <int NP, bool PAD>
__device__ __forceinline__ void panel_stage(double2* coef, const double2* __restrict__ src, int n, int M) {
    if constexpr (PAD) {
        for (int t = threadIdx.x; t < M * NP * NP; t += blockDim.x) {
            const int m = t / (NP * NP), e = t - m * (NP * NP);
            const int rr = e % NP, j = e / NP;
            coef[t] = (rr < n && j < n) ? src[(size_t)m * n * n + rr + n * j] : make_double2(0.0, 0.0);
        }
    } else {
        for (int t = threadIdx.x; t < M * n * n; t += blockDim.x) coef[t] = src[t];
    }
}

// ---- Hermitian series, symmetric frequency range (first = -F, M = 2 F + 1): H(x)_rj = c_0,rj + sum_{f=1..F} [c_f,rj z^f +
// conj(c_f,jr) z^-f].  With s = c_f,rj + c_f,jr and t = c_f,rj - c_f,jr (staged instead of c_f and c_-f: the same LDS bytes)
//   H_rj += (s.x pr - s.y pi) + i (t.x pi + t.y pr),   p = z^f = (pr, pi)
// one FMA group serves +f and -f: half the series flops of panel_series_row, one sincospi per node instead of two.
// Layout [1 + 2 F][NP * NP]: block 0 = (sw + i eta) I - c_0 with identity rows in the padding (the integral's shift is
// constant over its whole adaptive loop: no per-node diagonal select), block 2 f - 1 = s_f, block 2 f = t_f, element
// (row rr, column j) at j * NP + rr.
template <int NP>
__device__ __forceinline__ void panel_stage_fold(double2* coef, const double2* __restrict__ src, int n, int M, double sw, double eta) {
    const int F = (M - 1) / 2, nn = n * n;
    for (int t = threadIdx.x; t < M * NP * NP; t += blockDim.x) {
        const int b = t / (NP * NP), e = t - b * (NP * NP);
        const int rr = e % NP, j = e / NP;
        double2 v = make_double2(0.0, 0.0);
        if (b == 0 && rr == j) v = rr < n ? make_double2(sw, eta) : make_double2(1.0, 0.0);
        if (rr < n && j < n) {
            if (b == 0) {
                const double2 c = src[(size_t)F * nn + rr + n * j];
                v = make_double2(v.x - c.x, v.y - c.y);
            } else {
                const int f = (b + 1) >> 1;
                const double2 c = src[(size_t)(F + f) * nn + rr + n * j], cp = src[(size_t)(F + f) * nn + j + n * rr];
                v = (b & 1) ? make_double2(c.x + cp.x, c.y + cp.y) : make_double2(c.x - cp.x, c.y - cp.y);
            }
        }
        coef[t] = v;
    }
}

// row r of (sw + i eta) I - H(x) from the folded set; (zr, zi) = e^{2 pi i x}
template <int NP>
__device__ __forceinline__ void panel_series_row_fold(const double2* coef, int M, double zr, double zi, int r, double (&ar)[NP],
                                                      double (&ai)[NP]) {
    const int F = (M - 1) / 2;
    constexpr int nn = NP * NP;
#pragma unroll
    for (int j0 = 0; j0 < NP; j0 += 8) {
        double2 c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = coef[r + NP * (j0 + j)];
        pin8(c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ar[j0 + j] = c[j].x;
            ai[j0 + j] = c[j].y;
        }
    }
    double pr = 1.0, pi = 0.0;
    for (int f = 1; f <= F; ++f) {
        const double nr = pr * zr - pi * zi, ni = pr * zi + pi * zr;
        pr = nr;
        pi = ni;
        const double2* __restrict__ sm = coef + (size_t)(2 * f - 1) * nn + r;
        const double2* __restrict__ tm = sm + nn;
#pragma unroll
        for (int j0 = 0; j0 < NP; j0 += 8) {
            // eight reads in flight at a time (32 VGPRs beside the 64 of the row: sixteen would spill it at 4 waves/SIMD)
            double2 sv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) sv[j] = sm[NP * (j0 + j)];
            pin8(sv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // A = z I - H: accumulate -H
                ar[j0 + j] = fma(-sv[j].x, pr, ar[j0 + j]);
                ar[j0 + j] = fma(sv[j].y, pi, ar[j0 + j]);
            }
            double2 tv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) tv[j] = tm[NP * (j0 + j)];
            pin8(tv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ai[j0 + j] = fma(-tv[j].x, pi, ai[j0 + j]);
                ai[j0 + j] = fma(-tv[j].y, pr, ai[j0 + j]);
            }
        }
    }
}

// out[j] = v of lane j of the node's lanes
template <int NP, int... J>
__device__ __forceinline__ void group_gather(double v, double (&out)[NP], std::integer_sequence<int, J...>) {
    ((out[J] = group_bcast<NP, J>(v)), ...);
}

// sum over the NP lanes of a node (every lane gets it)
// (8 and 16 lanes: mirror steps inside the 16-lane DPP row -- row_mirror i <-> 15 - i, row_half_mirror i <-> 7 - i, then the
// two quad permutations; two 32-bit DPP moves and an add per step on the VALU instead of two `ds_bpermute_b32` and their
// wait on the LDS crossbar.  The Householder steps below run three of these sums per column.)
template <int CTRL>
__device__ __forceinline__ double dpp_perm_f64(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
#else
    return v;
#endif
}
template <int NP>
__device__ __forceinline__ double group_sum(double v) {
    if constexpr (NP == 32) {  // two DPP rows: each row's sum as for 16 lanes, then the other row's through the crossbar
        v += dpp_perm_f64<0x140>(v);
        v += dpp_perm_f64<0x141>(v);
        v += dpp_perm_f64<0x1b>(v);
        v += dpp_perm_f64<0xb1>(v);
        return v + __shfl_xor(v, 16, 64);
    } else if constexpr (NP == 16 || NP == 8) {
        if constexpr (NP == 16) v += dpp_perm_f64<0x140>(v);  // row_mirror
        v += dpp_perm_f64<0x141>(v);                          // row_half_mirror
        v += dpp_perm_f64<0x1b>(v);                           // quad_perm [3, 2, 1, 0]
        v += dpp_perm_f64<0xb1>(v);                           // quad_perm [1, 0, 3, 2]
        return v;
    } else {
#pragma unroll
        for (int off = NP / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    }
}

// ------------------------------------------------------------------------------------------
// Eigenvalues only, 5..16 bands, row layout: Householder tridiagonalisation + Sturm bisection.
// The parallel-order Jacobi above costs ~65 k instructions per four 16 x 16 matrices (~9 sweeps x 15 steps x ~480) and
// bounded rule builds with eigenvalues at 25 M eigensolves/s; eigenvalues alone need ~1/15 of its flops:
//   (1) n - 2 Householder reflections H = I - beta v v^H, A <- H A H by the rank-2 form A - v q^H - q v^H with
//       p = beta A v, q = p - (beta/2)(v^H p) v.  Lane r owns row r: the column below the diagonal is one element per
//       lane, its norm and v^H p are 16-lane sums, v and q travel by `group_bcast`.  Rows <= k are dead after step k and
//       are not protected.  Only the diagonal d_k and |e_k|^2 = the column norms survive;
//   (2) lane r finds the r-th smallest eigenvalue of the real symmetric tridiagonal (d, |e|) by bisection on the Sturm
//       count (ratio form q_i = d_i - x - |e_{i-1}|^2 / q_{i-1} with the usual pivot guard), 48 halvings of the
//       Gershgorin interval: all 64 lanes busy, no cross-lane traffic, backward stable (eps ||A|| like LAPACK's
//       stebz), degenerate spectra included.
// ~13 k instructions per four matrices.  Eigenvector builds (GGR) keep the Jacobi.
// ------------------------------------------------------------------------------------------
// column j of the two j-loops of a Householder step (j is a template parameter: `group_bcast` patterns are immediates).
// 16 lanes: v_j and q_j are read from lane j INSIDE the FMAs (`v_fmac_f64_dpp ... row_newbcast:j`, see fmac_col_bcast
// above for the idiom and its wait states): 12 instructions per complex column instead of 4 broadcasts + 16, and no
// registers for the broadcast copies of v.
// 32 lanes (two DPP rows per node): after one exchange with the partner lane of the other row every lane holds v_c in `lo`
// and v_{c+16} in `hi` (c = its position in its row), so that column j reads v_j from lane j mod 16 of its OWN row out of
// `lo` (j < 16) or `hi` -- the same FMA-with-broadcast columns as with 16 lanes instead of two crossbar swizzles per value.
struct HhPair {
    double lor, loi, hir, hii;
};
template <int NP>
__device__ __forceinline__ HhPair hh_pair(int r, double vr, double vi) {
    HhPair h = {vr, vi, vr, vi};
    if constexpr (NP == 32) {
        const double xr = __shfl_xor(vr, 16, 64), xi = __shfl_xor(vi, 16, 64);
        const bool row0 = r < 16;
        h.lor = row0 ? vr : xr;
        h.loi = row0 ? vi : xi;
        h.hir = row0 ? xr : vr;
        h.hii = row0 ? xi : vi;
    }
    return h;
}
template <int NP, int J>
__device__ __forceinline__ void hh_col_p(const double (&ar)[NP], const double (&ai)[NP], double vr, double vi, const HhPair& vp,
                                         double (&vjr)[NP], double (&vji)[NP], double& pr, double& pi) {
    if constexpr (NP == 16 || NP == 32) {
        // p += A[r][J] v_J:  pr += vJr ar - vJi ai,  pi += vJi ar + vJr ai
        const double sr = (NP == 16) ? vr : (J < 16 ? vp.lor : vp.hir), si = (NP == 16) ? vi : (J < 16 ? vp.loi : vp.hii);
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %2, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %3, -%5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %2, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
            : "+v"(pr), "+v"(pi)
            : "v"(sr), "v"(si), "v"(ar[J]), "v"(ai[J]), "n"(J % 16));
    } else {
        vjr[J] = group_bcast<NP, J>(vr);
        vji[J] = group_bcast<NP, J>(vi);
        pr = fma(ar[J], vjr[J], pr);
        pr = fma(-ai[J], vji[J], pr);
        pi = fma(ar[J], vji[J], pi);
        pi = fma(ai[J], vjr[J], pi);
    }
}
template <int NP, int J>
__device__ __forceinline__ void hh_col_upd(double (&ar)[NP], double (&ai)[NP], double vr, double vi, double qr, double qi, const HhPair& vp,
                                           const HhPair& qp, const double (&vjr)[NP], const double (&vji)[NP]) {
    if constexpr (NP == 16 || NP == 32) {
        // A[r][J] -= v_r conj(q_J) + q_r conj(v_J):
        //   ar -= qJr vr + qJi vi + vJr qr + vJi qi,   ai -= qJr vi - qJi vr + vJr qi - vJi qr
        // (operands 6...9: the registers v_J and q_J are read from, lane J mod 16 of the row; 16 lanes: v and q themselves)
        const double svr = (NP == 16) ? vr : (J < 16 ? vp.lor : vp.hir), svi = (NP == 16) ? vi : (J < 16 ? vp.loi : vp.hii);
        const double sqr = (NP == 16) ? qr : (J < 16 ? qp.lor : qp.hir), sqi = (NP == 16) ? qi : (J < 16 ? qp.loi : qp.hii);
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %8, -%2 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %8, -%3 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %9, -%3 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %9, %2 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %6, -%4 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %6, -%5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %7, -%5 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %7, %4 row_newbcast:%10 row_mask:0xf bank_mask:0xf"
            : "+v"(ar[J]), "+v"(ai[J])
            : "v"(vr), "v"(vi), "v"(qr), "v"(qi), "v"(svr), "v"(svi), "v"(sqr), "v"(sqi), "n"(J % 16));
    } else {
        const double qjr = group_bcast<NP, J>(qr), qji = group_bcast<NP, J>(qi);
        ar[J] -= (vr * qjr + vi * qji) + (qr * vjr[J] + qi * vji[J]);
        ai[J] -= (vi * qjr - vr * qji) + (qi * vjr[J] - qr * vji[J]);
    }
}
template <int NP, int K, int... JJ>
__device__ __forceinline__ void hh_cols_p(const double (&ar)[NP], const double (&ai)[NP], double vr, double vi, const HhPair& vp,
                                          double (&vjr)[NP], double (&vji)[NP], double& pr, double& pi, std::integer_sequence<int, JJ...>) {
    (hh_col_p<NP, K + 1 + JJ>(ar, ai, vr, vi, vp, vjr, vji, pr, pi), ...);
}
template <int NP, int K, int... JJ>
__device__ __forceinline__ void hh_cols_upd(double (&ar)[NP], double (&ai)[NP], double vr, double vi, double qr, double qi, const HhPair& vp,
                                            const HhPair& qp, const double (&vjr)[NP], const double (&vji)[NP],
                                            std::integer_sequence<int, JJ...>) {
    (hh_col_upd<NP, K + 1 + JJ>(ar, ai, vr, vi, qr, qi, vp, qp, vjr, vji), ...);
}

// What a Householder step leaves behind for an eigenVECTOR computation (kernels_ggr_rows.hip); the eigenvalue kernels pass
// HhDrop, which keeps nothing (and costs nothing).
//   reflect<K>: step K reflected with v (this lane's component, zero in rows <= K) and beta; x1 = A[K+1][K] before the step
//               (uniform inside the node), a1sq = |x1|^2, sigma = the squared norm of the column: the new subdiagonal
//               element is -(x1 / |x1|) sqrt(sigma)  (-sqrt(sigma) when x1 = 0)
//   last<K>:    no reflection (K + 2 = n): the subdiagonal element is A[K+1][K] itself, held by lane K + 1 in (xr, xi)
struct HhDrop {
    template <int NP, int K>
    __device__ __forceinline__ void reflect(int, double, double, double, double, double, double, double) {}
    template <int NP, int K>
    __device__ __forceinline__ void last(int, double, double) {}
};

template <int NP, int K, class KEEP>
__device__ __forceinline__ void hh_step(int n, int r, double (&ar)[NP], double (&ai)[NP], double (&e2)[NP], KEEP& keep) {
    e2[K] = 0.0;
    if constexpr (K + 1 < NP) {
        if (K + 1 >= n) return;  // uniform
        const bool below = r > K && r < n;
        const double xr = below ? ar[K] : 0.0, xi = below ? ai[K] : 0.0;  // column K below the diagonal, one element per lane
        const double sigma = group_sum<NP>(xr * xr + xi * xi);
        e2[K] = sigma;
        if (K + 2 >= n) keep.template last<NP, K>(r, xr, xi);
        if constexpr (K + 2 < NP) {
            if (K + 2 >= n) return;  // the last off-diagonal: nothing left to eliminate (uniform)
            const double x1r = group_bcast<NP, K + 1>(xr), x1i = group_bcast<NP, K + 1>(xi);
            // v = x + phase ||x|| e_1 (no cancellation), beta = 2 / ||v||^2 = 1 / (||x|| (||x|| + |x_1|)); sigma = 0: beta = 0, a
            // no-op.  With s = ||x|| |x_1| = sqrt(sigma |x_1|^2):  v_1 = x_1 (1 + s / |x_1|^2),  1 / beta = sigma + s -- one
            // square root per step; x_1 = 0 (v_1 = ||x||) takes a second one under a uniform branch.
            // (the square root and the two reciprocals by estimate + Newton: these scalars sit on the critical path of a
            // step that has two waves per SIMD to hide behind; magnitudes outside [1e-140, 1e140] take the library routines)
            const double a1sq = x1r * x1r + x1i * x1i;
            const double y = sigma * a1sq;
            double sx, fac, beta;
            if (!__any(!(a1sq >= 1e-140 && sigma <= 1e140))) {
                sx = y * rsqrt_nr(y);
                fac = fma(sx, rcp_nr(a1sq), 1.0);
                beta = rcp_nr(sigma + sx);
            } else {
                sx = sqrt(y);
                fac = a1sq > 0.0 ? 1.0 + sx / a1sq : 0.0;
                const double den = sigma + sx;
                beta = den > 0.0 ? 1.0 / den : 0.0;
            }
            double v1r = x1r * fac, v1i = x1i * fac;
            if (__any(a1sq == 0.0 && sigma > 0.0)) {
                const double nrm = sqrt(sigma);
                v1r = a1sq == 0.0 ? nrm : v1r;
                v1i = a1sq == 0.0 ? 0.0 : v1i;
            }
            const double vr = (r == K + 1) ? v1r : xr, vi = (r == K + 1) ? v1i : xi;
            keep.template reflect<NP, K>(r, vr, vi, beta, x1r, x1i, a1sq, sigma);
            double vjr[NP], vji[NP];
            double pr = 0.0, pi = 0.0;  // p_r = beta sum_{j > K} A[r][j] v_j
            const HhPair vp = hh_pair<NP>(r, vr, vi);
            hh_cols_p<NP, K>(ar, ai, vr, vi, vp, vjr, vji, pr, pi, std::make_integer_sequence<int, NP - K - 1>());
            pr *= beta;
            pi *= beta;
            // kappa = (beta / 2) v^H p (real for Hermitian A up to rounding; the imaginary part is kept for the non-ideal case)
            const double kr = 0.5 * beta * group_sum<NP>(vr * pr + vi * pi);
            const double ki = 0.5 * beta * group_sum<NP>(vr * pi - vi * pr);
            const double qr = pr - (kr * vr - ki * vi), qi = pi - (kr * vi + ki * vr);
            // A[r][j] -= v_r conj(q_j) + q_r conj(v_j), j > K
            const HhPair qp = hh_pair<NP>(r, qr, qi);
            hh_cols_upd<NP, K>(ar, ai, vr, vi, qr, qi, vp, qp, vjr, vji, std::make_integer_sequence<int, NP - K - 1>());
        }
    }
}

template <int NP, class KEEP, int... K>
__device__ __forceinline__ void hh_steps_keep(int n, int r, double (&ar)[NP], double (&ai)[NP], double (&e2)[NP], KEEP& keep,
                                              std::integer_sequence<int, K...>) {
    (hh_step<NP, K>(n, r, ar, ai, e2, keep), ...);
}
template <int NP, int... K>
__device__ __forceinline__ void hh_steps(int n, int r, double (&ar)[NP], double (&ai)[NP], double (&e2)[NP],
                                         std::integer_sequence<int, K...> seq) {
    HhDrop drop;
    hh_steps_keep<NP>(n, r, ar, ai, e2, drop, seq);
}

template <int NP, int... J>
__device__ __forceinline__ void diag_gather(const double (&ar)[NP], double (&d)[NP], std::integer_sequence<int, J...>) {
    ((d[J] = group_bcast<NP, J>(ar[J])), ...);  // A[J][J] lives in lane J and is final after step J - 1
}

// lane r (r < n) returns the r-th smallest eigenvalue of the Hermitian matrix whose row r it holds (rows / columns >= n: padding)
template <int NP>
__device__ __forceinline__ double tri_eigval_bisect(int n, int r, const double (&d)[NP], const double (&e2)[NP]);
template <int NP>
__device__ __forceinline__ double rows_eigvals_tridiag(int n, int r, double (&ar)[NP], double (&ai)[NP]) {
    double e2[NP], d[NP];
    hh_steps<NP>(n, r, ar, ai, e2, std::make_integer_sequence<int, NP>());
    diag_gather<NP>(ar, d, std::make_integer_sequence<int, NP>());
    return tri_eigval_bisect<NP>(n, r, d, e2);
}

// lane r (r < n) returns the r-th smallest eigenvalue of the real symmetric tridiagonal (d, |e|^2) (every lane of the node
// holds all of it)
template <int NP>
__device__ __forceinline__ double tri_eigval_bisect(int n, int r, const double (&d)[NP], const double (&e2)[NP]) {
    // Gershgorin interval of the leading n x n tridiagonal block and the pivot guard
    double lo = d[0], hi = d[0], emax = 0.0, eprev = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < n) {
            const double en = (i + 1 < n) ? sqrt(e2[i]) : 0.0;
            lo = fmin(lo, d[i] - eprev - en);
            hi = fmax(hi, d[i] + eprev + en);
            emax = fmax(emax, e2[i]);
            eprev = en;
        }
    }
    const double span = fmax(fabs(lo), fabs(hi));
    lo -= 2.3e-16 * span + 4.9e-324;
    hi += 2.3e-16 * span + 4.9e-324;
    // Sturm counts WITHOUT divisions: the signs of the leading principal minors p_0 = 1, p_1 = d_0 - x,
    // p_i = (d_{i-1} - x) p_{i-1} - |e_{i-2}|^2 p_{i-2} change once per eigenvalue below x.  The matrix is scaled to unit
    // Gershgorin radius (|p_i| <= 3^16) and |e|^2 is kept above eps^2 (a perturbation of the spectrum below eps * span)
    // so that an exact zero of p_i -- x on an eigenvalue of a leading block, diagonal and decoupled matrices -- is
    // followed by p_{i+1} = -|e|^2 p_{i-1} with the right sign instead of a run of zeros.  3 f64 + 3 integer
    // instructions per step against a reciprocal, 6 FMAs and 5 selects for the quotient form q_i = d_i - x - |e|^2 / q_{i-1}.
    const double sc = span > 0.0 ? 1.0 / span : 1.0;
    double ds[NP], es[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        ds[i] = d[i] * sc;
        es[i] = fmax(e2[i] * sc * sc, 4.9e-32);
    }
    lo *= sc;
    hi *= sc;
    const int want = r < n ? r : n - 1;
    // count of eigenvalues below x and the value p_n(x) of the characteristic polynomial (free: the last minor).  The
    // signs of the minors are shifted into one word (v_alignbit_b32: one instruction per step) and their changes counted
    // at the end: 4 instructions per step (subtract, multiply, FMA, shift)
    const unsigned smask = n >= 32 ? 0x7fffffffu : ((1u << n) - 1u) >> 1;  // n - 1 neighbouring pairs of n signs
    auto sturm = [&](double x, int& cnt, double& pv) {
        double pm = 1.0, p = ds[0] - x;
        unsigned bits = (unsigned)__double2hiint(p) >> 31;
#pragma unroll
        for (int i = 1; i < NP; ++i) {
            if (i < n) {  // uniform
                const double pn = fma(ds[i] - x, p, -(es[i - 1] * pm));
                bits = __builtin_amdgcn_alignbit(bits, (unsigned)__double2hiint(pn), 31);  // (bits << 1) | sign
                pm = p;
                p = pn;
            }
        }
        // sign of p_1 (p_0 = 1 is positive) + the changes between neighbours
        cnt = (int)((bits >> (n - 1)) & 1u) + __popc((bits ^ (bits >> 1)) & smask);
        pv = p;
    };
    // Bisection on the count until this lane's eigenvalue is ALONE in its bracket (counts at the ends differ by one: p_n
    // changes sign across it), then interpolation on p_n -- inverse quadratic through the two ends and the end replaced
    // last, false position while there is no third point -- inside a bracket that the counts keep valid whatever the
    // rounding of p_n does.  Each interpolated point is pushed 0.4 tol towards the far end of the bracket, so that once
    // the iterates sit on the root the far end jumps next to it and the WIDTH of the bracket is the stopping test, as in
    // plain bisection (tol = what 48 halvings leave).  Three interpolated passes in a row that do not halve the bracket
    // are followed by a bisection pass; after 30 passes, and for clusters that never separate (degenerate levels), it is
    // bisection all the way like before.  The loop is wave-uniform: ~19 passes for random spectra instead of 48.
    const double tol = 7.1e-15;  // 2 * 2^-48: the width 48 halvings leave of a Gershgorin interval of (scaled) length 2
    double plo = 0.0, phi = 0.0, xo = 0.0, po = 0.0;
    int clo = 0, chi = n, slow = 0;
    bool klo = false, khi = false, ko = false;  // p_n known at the end / a third point is there
    for (int it = 0; it < 200; ++it) {
        const double w = hi - lo;
        const bool ip = klo && khi && chi - clo == 1 && ((__double2hiint(plo) ^ __double2hiint(phi)) < 0) && slow < 3 && it < 30;
        double x = 0.5 * (lo + hi);
        if (ip) {
            // false position, or (three distinct values) inverse quadratic interpolation: one division either way
            double num = lo * phi - hi * plo, den = phi - plo;
            if (ko && po != plo && po != phi) {
                const double dab = plo - phi, dac = plo - po, dbc = phi - po;
                num = (lo * phi * po) * dbc - (hi * plo * po) * dac + (xo * plo * phi) * dab;
                den = dab * dac * dbc;
            }
            double xs = num / den;
            if (!(xs > lo && xs < hi)) xs = (lo * phi - hi * plo) / (phi - plo);  // (rare: the parabola left the bracket)
            xs += (xs - lo < hi - xs) ? 0.4 * tol : -0.4 * tol;
            if (xs > lo && xs < hi) x = xs;
        }
        int cnt;
        double pv;
        sturm(x, cnt, pv);
        if (w > tol) {
            if (cnt > want) {
                xo = hi;
                po = phi;
                ko = khi;
                hi = x;
                chi = cnt;
                phi = pv;
                khi = true;
            } else {
                xo = lo;
                po = plo;
                ko = klo;
                lo = x;
                clo = cnt;
                plo = pv;
                klo = true;
            }
            slow = (ip && hi - lo > 0.5 * w) ? slow + 1 : 0;
        }
        if (!__any(hi - lo > tol)) break;
    }
    return 0.5 * (lo + hi) * span;
}

// Eigenvalues of a real symmetric tridiagonal (d, |e|^2), ONE LANE PER MATRIX: the root-free QR iteration of Pal, Walker
// and Kahan as LAPACK's dsterf runs it (see tri_eig_kernel in kernels_generic.hip).  The arrays of a lane live in LDS
// ([j][lane], rows n and n + 1 exist: the sweep reads ahead); on return ld[0 .. n-1][lane] holds the eigenvalues, unsorted.
// Returns the bottom index of the block that was still coupled when the iteration budget ran out (0: converged).
__device__ __forceinline__ int tri_qr_lane(double (*ld)[64], double (*le)[64], int n, int lane, double anorm2) {
    const double eps2 = 1.2325951644078309e-32;  // (2^-53)^2
    const double floor2 = eps2 * 1e-2 * anorm2;  // |e| <= 0.1 eps ||T||: a coupling that small moves no eigenvalue by more than that
    int L = n - 1;        // bottom of the block that is still coupled
    int budget = 30 * n;  // dsterf's iteration limit
    while (__any(L > 0 && budget > 0)) {
        if (L > 0 && budget > 0) {
            // deflate as far as it goes, THEN sweep: every pass of the wave's loop is a sweep for every lane that is not
            // finished (a pass that only deflated would sit out the other lanes' sweep)
            double dL = ld[L][lane], dm = ld[L - 1][lane], eb = le[L - 1][lane];
            while (eb <= eps2 * fabs(dL * dm) + floor2) {
                le[L - 1][lane] = 0.0;
                --L;
                if (L == 0) break;
                dL = dm;
                dm = ld[L - 1][lane];
                eb = le[L - 1][lane];
            }
            if (L > 0) {
                --budget;
                // shift: the eigenvalue of the bottom 2 x 2 closer to d_L
                double sg;
                if (!__any(!(eb >= 1e-140 && eb <= 1e140 && fabs(dm - dL) <= 1e30))) {  // (estimate + Newton, as in the sweep)
                    const double irte = rsqrt_nr(eb), rte = eb * irte;
                    const double s0 = 0.5 * (dm - dL) * irte;
                    const double w = fma(s0, s0, 1.0);
                    sg = dL - rte * rcp_nr(s0 + copysign(w * rsqrt_nr(w), s0));
                } else {
                    const double rte = sqrt(eb);
                    const double s0 = (dm - dL) / (2.0 * rte);
                    sg = dL - rte / (s0 + copysign(sqrt(fma(s0, s0, 1.0)), s0));
                }
                // The sweep: ~35 instructions per rotation on the common path -- ONE reciprocal (estimate + Newton) of r p
                // serves c = p / r, s = b / r and 1 / c = r / p; the next step's two LDS reads are issued before this step's
                // arithmetic, unconditionally (rows NP, NP + 1 of the arrays exist for that); LAPACK's special cases
                // (p = 0, r = 0, and anything near the ends of the double range) take a wave-uniform branch to the same
                // quantities by true divisions.
                double c = 1.0, sn = 0.0, gamma = ld[0][lane] - sg, pp = gamma * gamma;
                double bb = le[0][lane], alpha = ld[1][lane];
                for (int i = 0; i < L; ++i) {
                    const double bbn = le[i + 1][lane], alphan = ld[i + 2][lane];
                    const double r2 = pp + bb;
                    if (i != 0) le[i - 1][lane] = sn * r2;
                    const double oldgam = gamma;
                    if (!__any(!(pp >= 1e-140 && r2 <= 1e140))) {
                        const double t = rcp_nr(r2 * pp);
                        const double ppt = pp * t;
                        c = pp * ppt;
                        sn = bb * ppt;
                        gamma = c * (alpha - sg) - sn * oldgam;
                        pp = (gamma * gamma) * (r2 * (r2 * t));
                    } else {
                        const double oldc = c;
                        c = r2 != 0.0 ? pp / r2 : 1.0;
                        sn = r2 != 0.0 ? bb / r2 : 0.0;
                        gamma = c * (alpha - sg) - sn * oldgam;
                        pp = c != 0.0 ? (gamma * gamma) / c : oldc * bb;
                    }
                    ld[i][lane] = oldgam + (alpha - gamma);
                    bb = bbn;
                    alpha = alphan;
                }
                le[L - 1][lane] = sn * pp;
                ld[L][lane] = sg + gamma;
            }
        }
    }
    return L;
}

}  // namespace abz
