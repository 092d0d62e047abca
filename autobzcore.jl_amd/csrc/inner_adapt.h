// One step of QuadGK's globally adaptive loop for ONE 1-D integral, executed by a single lane of the
// group that owns the integral (shared by the n <= 4 and the generic-n inner kernels).  Semantics =
// iai_host.cpp / oracle auxquadgk scalar mode: DataStructures.jl heap (Base.Order.Reverse on E),
// numevals += 30 at pop time, re-sum over the heap in storage order at the end, shared gk15.h rule.
#pragma once
#include "abz_internal.h"
#include "gk15.h"

namespace abz {

constexpr int ADAPT_MAXC = 16;

// MAXC = 1: the scalar integrands' instantiation -- every per-component array is one register and no loop
// over components survives, so the lane that runs the step never touches scratch memory
template <int MAXC>
struct AdaptStateT {
    int nseg = 0, nheap = 0, popped = -1, status = 0;
    bool first = true;
    double E = 0.0, atol = 0.0, rtol = 0.0;
    long long numevals = 0;
    double Ir[MAXC], Ii[MAXC];
};
using AdaptState = AdaptStateT<ADAPT_MAXC>;

struct InnerOut {
    double2* I;
    double* E;
    int64_t* nev;
    int* status;
};

// ctl: [0] number of pending panels, [1..4] their (a, b) pairs
template <int MAXC>
__device__ __forceinline__ void adapt_init(AdaptStateT<MAXC>& st, double at, bool has_rtol, double rtol_user, double lo,
                                           double hi, double* ctl) {
    st = AdaptStateT<MAXC>();
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        st.Ir[c] = 0.0;
        st.Ii[c] = 0.0;
    }
    st.atol = at >= 0.0 ? at : 0.0;
    st.rtol = has_rtol ? rtol_user : ((at > 0.0) ? 0.0 : 1.4901161193847656e-08);  // sqrt(eps)
    ctl[0] = 1.0;
    ctl[1] = lo;
    ctl[2] = hi;
}

// Consumes the values of the pending panels (vals: [30][nc]), updates the heap and either schedules the
// next bisection in ctl (returns false) or writes the result (returns true).
// REGS: scalar integrands pull the panel values into registers first (worth it where the caller has the
// registers to spare: the n <= 4 kernel; the generic-n kernels keep their rows in registers instead).
// MS: capacity of the segment store (an integral that needs more reports status 1 and is redone by the host loop).
template <bool REGS = false, int MAXC = ADAPT_MAXC, int MS = ABZ_INNER_MAXSEG>
__device__ inline bool adapt_step(AdaptStateT<MAXC>& st, int nc_, double* seg_a, double* seg_b, double* seg_E, gkc* seg_I,
                                  const gkc* vals, int* heap, double* ctl, long long maxevals, const InnerOut& out) {
    const int nc = MAXC == 1 ? 1 : nc_;
    const int np = (int)ctl[0];
    int newseg[2] = {-1, -1};
    for (int pnl = 0; pnl < np; ++pnl) {
        int sl;  // the popped parent's slot is reused for the first child
        if (pnl == 0 && st.popped >= 0)
            sl = st.popped;
        else
            sl = st.nseg++;
        if (sl >= MS) {
            st.status = 1;
            sl = MS - 1;
        }
        newseg[pnl] = sl;
    }
    // scalar integrands: pull the panel values into registers with back-to-back LDS reads first -- the rule
    // below is executed by ONE lane, and reads interleaved with the stores of its results would each pay
    // the full LDS latency
    gkc rv[30];
    const bool fast = REGS && nc == 1;
    if (fast) {
#pragma unroll
        for (int i = 0; i < 30; ++i) rv[i] = vals[i < 15 * np ? i : 0];
    }
    if (st.first) {
        st.first = false;
        const int sl = newseg[0];
        seg_a[sl] = ctl[1];
        seg_b[sl] = ctl[2];
        if (fast) {
            gkc I1;
            seg_E[sl] = gk15_rule(rv, 1, ctl[1], ctl[2], &I1);
            seg_I[sl] = I1;
        } else
            seg_E[sl] = gk15_rule(vals, nc, seg_a[sl], seg_b[sl], seg_I + (size_t)sl * nc);
        for (int c = 0; c < nc; ++c) {
            st.Ir[c] = seg_I[(size_t)sl * nc + c].re;
            st.Ii[c] = seg_I[(size_t)sl * nc + c].im;
        }
        st.E = seg_E[sl];
        st.numevals = 15;
        heap[0] = sl;
        st.nheap = 1;
    } else {
        // I = (I - I_parent) + I_1 + I_2, E likewise; the parent's values are read before its slot is reused
        const int sp = st.popped;
        double pIr[MAXC], pIi[MAXC];
        for (int c = 0; c < nc; ++c) {
            pIr[c] = seg_I[(size_t)sp * nc + c].re;
            pIi[c] = seg_I[(size_t)sp * nc + c].im;
        }
        const double pE = seg_E[sp];
        const int s1 = newseg[0], s2 = newseg[1];
        seg_a[s1] = ctl[1];
        seg_b[s1] = ctl[2];
        seg_a[s2] = ctl[3];
        seg_b[s2] = ctl[4];
        if (fast) {
            gkc I1, I2;
            const double E1 = gk15_rule(rv, 1, ctl[1], ctl[2], &I1);
            const double E2 = gk15_rule(rv + 15, 1, ctl[3], ctl[4], &I2);
            seg_I[s1] = I1;
            seg_I[s2] = I2;
            seg_E[s1] = E1;
            seg_E[s2] = E2;
        } else {
            seg_E[s1] = gk15_rule(vals, nc, seg_a[s1], seg_b[s1], seg_I + (size_t)s1 * nc);
            seg_E[s2] = gk15_rule(vals + (size_t)15 * nc, nc, seg_a[s2], seg_b[s2], seg_I + (size_t)s2 * nc);
        }
        {
#pragma clang fp contract(off)
            for (int c = 0; c < nc; ++c) {
                st.Ir[c] = ((st.Ir[c] - pIr[c]) + seg_I[(size_t)s1 * nc + c].re) + seg_I[(size_t)s2 * nc + c].re;
                st.Ii[c] = ((st.Ii[c] - pIi[c]) + seg_I[(size_t)s1 * nc + c].im) + seg_I[(size_t)s2 * nc + c].im;
            }
            st.E = ((st.E - pE) + seg_E[s1]) + seg_E[s2];
        }
        for (int t = 0; t < 2; ++t) {  // heappush (percolate_up)
            const int x = t == 0 ? s1 : s2;
            int i = st.nheap++;
            while (i > 0) {
                const int j = (i - 1) / 2;
                if (!(seg_E[heap[j]] < seg_E[x])) break;
                heap[i] = heap[j];
                i = j;
            }
            heap[i] = x;
        }
    }
    double nrm = 0.0;
    {
#pragma clang fp contract(off)
        for (int c = 0; c < nc; ++c) {
            const double t1 = st.Ir[c] * st.Ir[c], t2 = st.Ii[c] * st.Ii[c];
            const double t3 = t1 + t2;
            nrm = nrm + t3;
        }
    }
    nrm = sqrt(nrm);
    const double tol = fmax(st.atol, st.rtol * nrm);
    if (st.E > tol && st.numevals < maxevals && st.status == 0) {
        // heappop: root out, last to root, percolate_down
        const int x = heap[0];
        const int y = heap[--st.nheap];
        if (st.nheap > 0) {
            int i = 0;
            while (true) {
                const int lc = 2 * i + 1;
                if (lc >= st.nheap) break;
                const int rc = lc + 1;
                const int j = (rc >= st.nheap || seg_E[heap[rc]] < seg_E[heap[lc]]) ? lc : rc;
                if (!(seg_E[y] < seg_E[heap[j]])) break;
                heap[i] = heap[j];
                i = j;
            }
            heap[i] = y;
        }
        st.popped = x;
        st.numevals += 30;
        const double pa = seg_a[x], pb = seg_b[x];
        const double mid = (pa + pb) / 2;
        ctl[0] = 2.0;
        ctl[1] = pa;
        ctl[2] = mid;
        ctl[3] = mid;
        ctl[4] = pb;
        return false;
    }
    {  // re-sum over the heap in storage order (QuadGK does this after adapt)
#pragma clang fp contract(off)
        for (int c = 0; c < nc; ++c) {
            st.Ir[c] = seg_I[(size_t)heap[0] * nc + c].re;
            st.Ii[c] = seg_I[(size_t)heap[0] * nc + c].im;
        }
        st.E = seg_E[heap[0]];
        for (int h = 1; h < st.nheap; ++h) {
            for (int c = 0; c < nc; ++c) {
                st.Ir[c] = st.Ir[c] + seg_I[(size_t)heap[h] * nc + c].re;
                st.Ii[c] = st.Ii[c] + seg_I[(size_t)heap[h] * nc + c].im;
            }
            st.E = st.E + seg_E[heap[h]];
        }
    }
    for (int c = 0; c < nc; ++c) out.I[c] = make_double2(st.Ir[c], st.Ii[c]);
    *out.E = st.E;
    *out.nev = st.numevals;
    *out.status = st.status;
    return true;
}

// The same step for a scalar integrand, restructured around the LDS latency that dominates it when ONE lane walks the
// data (the block-per-integral kernel: every other wave of the workgroup waits at the barrier meanwhile).  Called by
// lanes 0 and 1 of a wave: lane l applies the rule to pending panel l with its fifteen values pulled into registers
// by back-to-back reads; lane 0 then runs the heap.  `heapE[i]` mirrors seg_E[heap[i]], so a heap level costs one LDS
// round trip instead of the dependent pair heap[j] -> seg_E[heap[j]]; the popped parent's (I, E) are read at pop
// time.  Same arithmetic in the same order as adapt_step: identical (I, E), identical heap decisions.
// `pI*` / `pE`: the popped parent's values, carried by the caller between steps (lane 0's copy is the one used).
struct AdaptParent {
    double Ir = 0.0, Ii = 0.0, E = 0.0;
};

// `lane`: 0 or 1, the caller's position in its pair; `lane1`: the wave lane that plays lane 1 (the half-wave kernel of
// n <= 4 runs two integrals per wave: pairs (0, 1) and (32, 33)).
template <int MS>
__device__ inline bool adapt_step_pair(AdaptStateT<1>& st, AdaptParent& par, int lane, double* seg_a, double* seg_b, double* seg_E,
                                       gkc* seg_I, const gkc* vals, int* heap, double* heapE, double* ctl, long long maxevals,
                                       const InnerOut& out, int lane1 = 1) {
    const int np = (int)ctl[0];
    const double a1 = ctl[1], b1 = ctl[2], a2 = ctl[3], b2 = ctl[4];
    const int pl = lane < np ? lane : 0;  // the panel of this lane
    gkc rv[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) rv[i] = vals[15 * pl + i];
    gkc Il;
    const double El = gk15_rule(rv, 1, pl ? a2 : a1, pl ? b2 : b1, &Il);
    const double I2r = __shfl(Il.re, lane1, 64), I2i = __shfl(Il.im, lane1, 64), E2 = __shfl(El, lane1, 64);
    if (lane != 0) return false;
    int newseg[2] = {-1, -1};
    for (int pnl = 0; pnl < np; ++pnl) {
        int sl;  // the popped parent's slot is reused for the first child
        if (pnl == 0 && st.popped >= 0)
            sl = st.popped;
        else
            sl = st.nseg++;
        if (sl >= MS) {
            st.status = 1;
            sl = MS - 1;
        }
        newseg[pnl] = sl;
    }
    if (st.first) {
        st.first = false;
        const int sl = newseg[0];
        seg_a[sl] = a1;
        seg_b[sl] = b1;
        seg_E[sl] = El;
        seg_I[sl] = Il;
        st.Ir[0] = Il.re;
        st.Ii[0] = Il.im;
        st.E = El;
        st.numevals = 15;
        heap[0] = sl;
        heapE[0] = El;
        st.nheap = 1;
    } else {
        const int s1 = newseg[0], s2 = newseg[1];
        seg_a[s1] = a1;
        seg_b[s1] = b1;
        seg_a[s2] = a2;
        seg_b[s2] = b2;
        seg_I[s1] = Il;
        gkc I2;
        I2.re = I2r;
        I2.im = I2i;
        seg_I[s2] = I2;
        seg_E[s1] = El;
        seg_E[s2] = E2;
        {
#pragma clang fp contract(off)
            st.Ir[0] = ((st.Ir[0] - par.Ir) + Il.re) + I2r;
            st.Ii[0] = ((st.Ii[0] - par.Ii) + Il.im) + I2i;
            st.E = ((st.E - par.E) + El) + E2;
        }
        for (int t = 0; t < 2; ++t) {  // heappush (percolate_up)
            const int x = t == 0 ? s1 : s2;
            const double Ex = t == 0 ? El : E2;
            int i = st.nheap++;
            while (i > 0) {
                const int j = (i - 1) / 2;
                const double Ej = heapE[j];
                const int hj = heap[j];
                if (!(Ej < Ex)) break;
                heap[i] = hj;
                heapE[i] = Ej;
                i = j;
            }
            heap[i] = x;
            heapE[i] = Ex;
        }
    }
    double tol = st.atol;
    if (st.rtol != 0.0) {  // (rtol = 0: max(atol, 0 * |I|) = atol, no square root)
#pragma clang fp contract(off)
        const double t1 = st.Ir[0] * st.Ir[0], t2 = st.Ii[0] * st.Ii[0];
        const double t3 = t1 + t2;
        const double nrm = sqrt(0.0 + t3);
        tol = fmax(st.atol, st.rtol * nrm);
    }
    if (st.E > tol && st.numevals < maxevals && st.status == 0) {
        // heappop: root out, last to root, percolate_down
        const int x = heap[0];
        const int nh = --st.nheap;
        const int y = heap[nh];
        const double Ey = heapE[nh];
        const gkc Ix = seg_I[x];
        const double pa = seg_a[x], pb = seg_b[x];
        par.E = heapE[0];
        par.Ir = Ix.re;
        par.Ii = Ix.im;
        if (nh > 0) {
            int i = 0;
            while (true) {
                const int lc = 2 * i + 1;
                if (lc >= nh) break;
                const int rc = lc + 1;
                const int rcs = rc < nh ? rc : lc;  // a readable index when the right child does not exist
                const double El_ = heapE[lc], Er_ = heapE[rcs];
                const int hl = heap[lc], hr = heap[rcs];
                const bool left = rc >= nh || Er_ < El_;
                const double Ej = left ? El_ : Er_;
                if (!(Ey < Ej)) break;
                heap[i] = left ? hl : hr;
                heapE[i] = Ej;
                i = left ? lc : rc;
            }
            heap[i] = y;
            heapE[i] = Ey;
        }
        st.popped = x;
        st.numevals += 30;
        const double mid = (pa + pb) / 2;
        ctl[0] = 2.0;
        ctl[1] = pa;
        ctl[2] = mid;
        ctl[3] = mid;
        ctl[4] = pb;
        return false;
    }
    {  // re-sum over the heap in storage order (QuadGK does this after adapt)
#pragma clang fp contract(off)
        st.Ir[0] = seg_I[heap[0]].re;
        st.Ii[0] = seg_I[heap[0]].im;
        st.E = heapE[0];
        for (int h = 1; h < st.nheap; ++h) {
            st.Ir[0] = st.Ir[0] + seg_I[heap[h]].re;
            st.Ii[0] = st.Ii[0] + seg_I[heap[h]].im;
            st.E = st.E + heapE[h];
        }
    }
    out.I[0] = make_double2(st.Ir[0], st.Ii[0]);
    *out.E = st.E;
    *out.nev = st.numevals;
    *out.status = st.status;
    return true;
}

// LDS doubles of one integral in flight: seg_a, seg_b, seg_E | seg_I | vals[30] | heap | ctl
__host__ __device__ inline int inner_group_doubles(int ncomp, int ms = ABZ_INNER_MAXSEG) {
    return 3 * ms + 2 * ncomp * ms + 2 * ncomp * 30 + ms / 2 + 8;
}

}  // namespace abz
