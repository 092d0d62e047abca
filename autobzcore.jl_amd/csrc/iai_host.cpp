// IAI: iterated adaptive integration with the adaptive Gauss-Kronrod loops on the host and every
// batch of nodes on the GPU.
//
// Reference shape (src/fourier.jl:432-510, src/algorithms.jl:215-239): a depth-first recursion in
// which an outer GK(7,15) node x contracts the Fourier coefficients (workspace_contract!) and
// launches a complete inner adaptive integral with abstol/len.  Here the same 1-D integrals exist,
// with the same tolerances and the same scalar refinement rule (pop the worst panel, bisect), but
// all sibling integrals of one outer round advance in lockstep so that every round is one batched
// contraction / evaluation launch.  A 1-D integral's decisions depend only on its own node values,
// so its panel tree is the one the depth-first traversal builds.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <chrono>
#include <cmath>
#include <complex>
#include <deque>
#include <exception>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "abz_internal.h"
#include "gk15.h"

namespace abz {

typedef std::complex<double> cd;

// A small persistent pool for the host side of the driver: the per-integral bookkeeping of a round (GK sums, heaps,
// limits of the new integrals) is independent per integral, and in a parameter sweep it -- not the GPU -- bounds the
// solve (432-omega sweep of the reference's demo: 0.06 s of host work beside 0.05 s of kernels).  Static partition,
// deterministic results (every integral is handled by exactly one thread and sees only its own state).
class HostPool {
public:
    explicit HostPool(int nthreads) : n_(std::max(1, nthreads)) {
        for (int t = 1; t < n_; ++t) workers_.emplace_back([this, t] { loop(t); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    int size() const { return n_; }
    // fn(begin, end, tid) over [0, count) in n contiguous shares
    void run(int64_t count, const std::function<void(int64_t, int64_t, int)>& fn) {
        if (n_ == 1 || count < 2) {
            fn(0, count, 0);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            count_ = count;
            pending_.store(n_ - 1, std::memory_order_relaxed);
            ++gen_;
        }
        cv_.notify_all();
        share(0);
        while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
        if (failed_.load(std::memory_order_acquire)) {  // a share threw (std::bad_alloc ...): rethrow on the caller's thread,
            std::exception_ptr e;                         // where the entry point's catch-all turns it into a status
            {
                std::lock_guard<std::mutex> lk(m_);
                e = error_;
                error_ = nullptr;
                failed_.store(false, std::memory_order_relaxed);
            }
            if (e) std::rethrow_exception(e);
        }
    }

private:
    void share(int t) {
        const int64_t b = count_ * t / n_, e = count_ * (t + 1) / n_;
        try {
            if (e > b) (*fn_)(b, e, t);
        } catch (...) {  // never out of a worker thread (that is std::terminate)
            std::lock_guard<std::mutex> lk(m_);
            if (!error_) error_ = std::current_exception();
            failed_.store(true, std::memory_order_release);
        }
    }
    void loop(int t) {
        uint64_t seen = 0;
        while (true) {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            share(t);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    int n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    uint64_t gen_ = 0;
    bool stop_ = false;
    const std::function<void(int64_t, int64_t, int)>* fn_ = nullptr;
    int64_t count_ = 0;
    std::atomic<int> pending_{0};
    std::atomic<bool> failed_{false};
    std::exception_ptr error_;
};

void gk15_nodes(double a, double b, double* x) {
    for (int i = 0; i < 15; ++i) x[i] = gk15_node(a, b, i);
}

// QuadGK.evalrule for order 7 on one panel: fv [15][ncomp] in gk15_nodes order (shared with the device)
void gk15_evalrule(const cd* fv, int ncomp, double a, double b, cd* I, double* E) {
    *E = gk15_rule(reinterpret_cast<const gkc*>(fv), ncomp, a, b, reinterpret_cast<gkc*>(I));
}

struct Seg {
    double a, b, E;
    int64_t ioff;     // offset of I[ncomp] in the owner's value store
    int64_t nev = 0;  // integrand evaluations beneath this panel (counted when the panel enters the integral)
    // The two halves of the panel, evaluated AHEAD of the pop that consumes them (see solve_level):
    // 0 = not requested, 1 = in flight, 2 = ready.
    int state = 0;
    double E1 = 0.0, E2 = 0.0;
    int64_t ioff1 = 0, ioff2 = 0, nev1 = 0, nev2 = 0;
};

// Break points of a polytope along one coordinate: the distinct vertex coordinates (sqrt(eps)
// tolerances, first occurrence kept), ascending.  ref: get_segs, ext/SymmetryReduceBZExt.jl:15-31.
static void unique_sorted(const std::vector<double>& vals, std::vector<double>& out) {
    const double tol = 1.4901161193847656e-08;
    out.clear();
    for (double v : vals) {
        bool dup = false;
        for (double u : out)
            if (std::fabs(v - u) <= std::max(tol, tol * std::max(std::fabs(v), std::fabs(u)))) {
                dup = true;
                break;
            }
        if (!dup) out.push_back(v);
    }
    std::sort(out.begin(), out.end());
}

// Iterated limits.  CUBIC / TETRAHEDRAL: closed forms.  POLYHEDRAL (3 variables): convex polyhedron by its
// faces, packed [nv, x y z * nv] per face; fixing z gives a POLYGON (2 variables): vertices [x y] in order
// around the boundary; fixing y gives an interval (CUBIC).  The slicing arithmetic is the oracle's, with
// FMA contraction off, so that both sides place the same panels.
// ref: Polyhedron3 / Polygon2, pg_vert_from_zslice, xlim_from_yslice (ext/SymmetryReduceBZExt.jl:33-58,
// ext/ibzlims.jl:198-289).
struct Lims {
    int kind;
    double a[ABZ_MAX_DIM], b[ABZ_MAX_DIM];
    double s;
    // polytope data: owned by the arena of the solve (raw pointers: a Lims is copied for every kid of every
    // round, reference counting there cost a third of the sweep time of the reference's demo)
    const std::vector<double>* poly = nullptr;
    std::deque<std::vector<double>>* arena = nullptr;
    // end points of variable L without touching the heap (every kid of every round asks for them)
    bool range(int L, double& lo, double& hi) const {
        if (kind == ABZ_LIMS_CUBIC) {
            lo = a[L - 1];
            hi = b[L - 1];
            return true;
        }
        if (kind == ABZ_LIMS_TETRAHEDRAL) {
            lo = 0.0;
            hi = a[L - 1] * s;
            return true;
        }
        return false;  // polytopes: several break points, see segs()
    }
    void segs(int L, std::vector<double>& out) const {  // break points of variable L (1-based)
        out.clear();
        if (kind == ABZ_LIMS_CUBIC) {
            out.push_back(a[L - 1]);
            out.push_back(b[L - 1]);
        } else if (kind == ABZ_LIMS_TETRAHEDRAL) {
            out.push_back(0.0);
            out.push_back(a[L - 1] * s);
        } else if (kind == ABZ_LIMS_POLYHEDRAL) {
            std::vector<double> zs;
            const std::vector<double>& f = *poly;
            for (size_t i = 0; i < f.size();) {
                const int nv = (int)f[i++];
                for (int j = 0; j < nv; ++j, i += 3) zs.push_back(f[i + 2]);
            }
            unique_sorted(zs, out);
        } else {  // polygon
            std::vector<double> ys;
            const std::vector<double>& v = *poly;
            for (size_t i = 0; i + 1 < v.size(); i += 2) ys.push_back(v[i + 1]);
            unique_sorted(ys, out);
        }
    }
    Lims fix(int L, double x) const {  // limits of variables 1..L-1 once variable L is fixed
        if (kind <= ABZ_LIMS_TETRAHEDRAL) {  // the hot case, inlined into the kid loops
            Lims r = *this;
            if (kind == ABZ_LIMS_TETRAHEDRAL) r.s = x / a[L - 1];
            return r;
        }
        return fix_poly(L, x);
    }
    __attribute__((noinline)) Lims fix_poly(int L, double x) const {
#pragma clang fp contract(off)
        Lims r = *this;
        (void)L;
        if (kind == ABZ_LIMS_POLYHEDRAL) {
            const std::vector<double>& f = *poly;
            std::vector<double> pts;  // (x, y) pairs
            for (size_t i = 0; i < f.size();) {
                const int nv = (int)f[i++];
                const double* face = &f[i];
                for (int j = 0; j < nv; ++j) {
                    const double* p1 = face + 3 * j;
                    const double* p2 = face + 3 * ((j + 1) % nv);
                    const double z1 = p1[2], z2 = p2[2];
                    if ((z1 <= x && z2 >= x) || (z1 >= x && z2 <= x)) {
                        if (z2 == z1) continue;
                        const double t = (x - z1) / (z2 - z1);
                        const double omt = 1 - t;
                        const double qx = t * p2[0] + omt * p1[0];
                        const double qy = t * p2[1] + omt * p1[1];
                        bool dup = false;
                        for (size_t k = 0; k + 1 < pts.size() && !dup; k += 2) dup = pts[k] == qx && pts[k + 1] == qy;
                        if (!dup) {
                            pts.push_back(qx);
                            pts.push_back(qy);
                        }
                    }
                }
                i += (size_t)3 * nv;
            }
            const size_t np = pts.size() / 2;
            double cx = 0.0, cy = 0.0;
            for (size_t k = 0; k < np; ++k) {
                cx = cx + pts[2 * k];
                cy = cy + pts[2 * k + 1];
            }
            cx = cx / (double)np;
            cy = cy / (double)np;
            std::vector<std::pair<double, size_t>> ang(np);
            for (size_t k = 0; k < np; ++k) ang[k] = {std::atan2(pts[2 * k + 1] - cy, pts[2 * k] - cx), k};
            std::stable_sort(ang.begin(), ang.end(), [](const auto& u, const auto& v) { return u.first < v.first; });
            arena->emplace_back(2 * np);
            std::vector<double>& pg = arena->back();
            for (size_t k = 0; k < np; ++k) {
                pg[2 * k] = pts[2 * ang[k].second];
                pg[2 * k + 1] = pts[2 * ang[k].second + 1];
            }
            r.kind = ABZ_LIMS_POLYGON;
            r.poly = &pg;
        } else if (kind == ABZ_LIMS_POLYGON) {
            const std::vector<double>& v = *poly;
            const size_t nv = v.size() / 2;
            double lb = 0.0, ub = 0.0;
            int k = 0;
            for (size_t j = 0; j < nv && k < 2; ++j) {
                const size_t jp = (j + 1) % nv;
                const double y1 = v[2 * j + 1], y2 = v[2 * jp + 1];
                if ((y1 < x && y2 > x) || (y1 > x && y2 < x) || y1 == x) {
                    const double t = (y2 != y1) ? (x - y1) / (y2 - y1) : 0.0;
                    const double omt = 1 - t;
                    const double lim = t * v[2 * jp] + omt * v[2 * j];
                    if (++k == 1)
                        lb = ub = lim;
                    else {
                        ub = std::max(lb, lim);
                        lb = std::min(lb, lim);
                    }
                }
            }
            r.kind = ABZ_LIMS_CUBIC;
            r.poly = nullptr;
            r.a[0] = lb;
            r.b[0] = ub;
        }
        return r;
    }
};

// One adaptive 1-D integral (QuadGK do_quadgk/adapt state with DataStructures heap semantics).
struct Quad1D {
    int64_t slot = 0;  // coefficient set of this integral's series
    double sweep = 0.0;  // swept parameter of the solve this integral belongs to
    int root = 0;        // index of that solve (abz_iai_solve_many)
    double tail[ABZ_MAX_DIM] = {0, 0, 0};  // fixed outer coordinates x_{L+1}.. (tail[0] = x_{L+1})
    Lims lims;
    bool has_atol = false;
    double atol = 0.0, rtol = 0.0;
    std::vector<Seg> heap;
    std::vector<cd> store;  // segment integrals, ncomp each
    std::vector<cd> I;
    double E = 0.0;
    int64_t numevals = 0;  // nodes of THIS integral (what QuadGK's maxevals counts)
    int64_t fevals = 0;    // integrand evaluations of the whole subtree (EvalCounter)
    bool done = false, started = false;
    // pending work of the current round: panels to evaluate, and the parents they belong to --
    // heap positions of panels whose halves were requested ahead of their pop (scalar refinement), or the
    // panels popped together by the BatchIntegrand refinement rule
    std::vector<Seg> pend;
    std::vector<uint32_t> req;
    std::vector<Seg> popped;
};

static inline bool heap_lt(const Seg& x, const Seg& y) { return y.E < x.E; }  // lt(Reverse, x, y)

static void percolate_down(std::vector<Seg>& xs, size_t i, Seg x, size_t len) {
    while (true) {
        const size_t l = 2 * i + 1;
        if (l >= len) break;
        const size_t r = l + 1;
        const size_t j = (r >= len || heap_lt(xs[l], xs[r])) ? l : r;
        if (!heap_lt(xs[j], x)) break;
        xs[i] = xs[j];
        i = j;
    }
    xs[i] = x;
}
static void percolate_up(std::vector<Seg>& xs, size_t i, Seg x) {
    while (i > 0) {
        const size_t j = (i - 1) / 2;
        if (!heap_lt(x, xs[j])) break;
        xs[i] = xs[j];
        i = j;
    }
    xs[i] = x;
}
static void heapify(std::vector<Seg>& xs) {
    const size_t n = xs.size();
    for (size_t i = n / 2; i-- > 0;) percolate_down(xs, i, xs[i], n);
}
static void heap_push(std::vector<Seg>& xs, Seg x) {
    xs.push_back(x);
    percolate_up(xs, xs.size() - 1, x);
}
static Seg heap_pop(std::vector<Seg>& xs) {
    Seg x = xs[0];
    Seg y = xs.back();
    xs.pop_back();
    if (!xs.empty()) percolate_down(xs, 0, y, xs.size());
    return x;
}

static double vnorm(const std::vector<cd>& v) {
    double s = 0.0;
    for (auto& z : v) s += std::norm(z);
    return std::sqrt(s);
}

struct IaiDriver {
    abz_series* s;
    abz_ctx* ctx;
    int d, n, ncomp, integrand;
    double params[4];
    double sweep;
    bool has_rtol;
    double rtol_user;
    int64_t maxevals;
    int64_t max_batch = 0;  // 0: scalar refinement; > 0: BatchIntegrand refinement with this soft cap
    bool panels15 = true;  // eval_nodes is fed whole GK panels (solve_level); the node-list ABI entry clears it
    bool speculate = true;  // request the halves of every panel that is certain to be popped (solve_level)
    int64_t spec_cap_nodes = 131072;  // ... in rounds smaller than this
    int64_t pool_cap_bytes = (int64_t)4 << 30;  // contracted sets alive at once per level (ABZ_IAI_POOL_MB)
    int64_t launches = 0;   // innermost launches of this solve (diagnostics)
    // ABZ_IAI_STATS=1: innermost launches by size (log2 buckets): count, integrals, seconds
    bool stats = false;
    int64_t st_cnt[40] = {0}, st_int[40] = {0};
    int64_t st_rounds[ABZ_MAX_DIM + 1] = {0, 0, 0, 0};
    std::vector<int64_t> h_parents;
    std::vector<double> h_x, h_tail, h_sweep;
    std::vector<cd> h_values;

    double tol_r(const Quad1D& q) const {
        // quadgk defaults: rtol = (atol > 0 ? 0 : sqrt(eps)) when not given
        if (has_rtol) return rtol_user;
        return (q.has_atol && q.atol > 0) ? 0.0 : std::sqrt(2.220446049250313e-16);
    }

    bool device_inner = false;  // innermost adaptive loops on the GPU (scalar refinement, n <= 4)

    // Hermitian series of n <= 4 bands: the whole chain runs on PACKED coefficient rows (packed_herm.h; packing commutes with
    // the contraction of the outer variables): half the contraction work and level-1 sets of 51 instead of 99 numbers (SVO),
    // and the innermost kernels evaluate the folded series (half the terms, no seed phase)
    bool pk = false;
    int64_t set_elems(int level) const {  // numbers per level-`level` coefficient set in this solve's layout
        const int64_t row_full = (int64_t)s->dims[0] * s->n * s->n;
        return pk ? s->elems(level) / row_full * (int64_t)packed_row_elems(s->n, s->dims[0]) : s->elems(level);
    }
    const double2* top_coef() const { return pk ? s->coef_pk.as<double2>() : s->coef; }
    int contract_nodes(int L, int64_t nn, int64_t base_slot, int64_t off = 0, const int64_t* par = nullptr, const double* x = nullptr,
                       const int64_t* par_dev = nullptr, const double* x_dev = nullptr);
    int panels_enqueue(int L, int64_t c0p, int64_t cnp, int buf, const Lims& lims);
    int panels_collect(int64_t c0p, int64_t cnp, int buf, std::vector<int64_t>& redo);
    // panels of a round of the level above the innermost one (solve_level, panel mode): inputs gathered per round ...
    std::vector<int64_t> pp_slot;
    std::vector<double> pp_a, pp_b, pp_at, pp_sw;
    std::vector<uint32_t> pp_q;  // the integral a panel belongs to (index into the level's integrals)
    // ... and what comes back per panel: I_K s [ncomp], E, evaluations beneath it
    std::vector<cd> pI;
    std::vector<double> pE;
    std::vector<int64_t> pnev;
    int eval_nodes(int64_t nn);
    int solve_level(int L, std::vector<Quad1D>& quads);
    int flat_enqueue(int64_t nq, int buf);
    int flat_collect(int64_t nq, int buf, cd* vals, int64_t* nev, std::vector<int64_t>& redo);
    size_t flat_out_bytes(int64_t nq) const {
        return sizeof(double2) * (size_t)(nq * ncomp) + sizeof(double) * (size_t)nq + sizeof(int64_t) * (size_t)nq + sizeof(int) * (size_t)nq;
    }
    std::unique_ptr<HostPool> pool;  // ABZ_HOST_THREADS (default 8, capped by the hardware); nullptr: serial
    // run fn over [0, count): on the pool when the work is worth a wake-up and nothing shared is touched (polytope
    // limits allocate from the solve's arena)
    void par(int64_t count, bool safe, const std::function<void(int64_t, int64_t, int)>& fn) {
        if (pool && safe && count >= 2048)
            pool->run(count, fn);
        else
            fn(0, count, 0);
    }
    // a single solve sharded over the ranks of a process group (abz_iai_set_exchange)
    abz_exchange_fn ex_fn = nullptr;
    void* ex_user = nullptr;
    int ex_rank = 0, ex_world = 1;
    std::vector<int64_t> l_par, l_nev;
    std::vector<double> l_x;
    std::vector<uint32_t> l_q;
    std::vector<cd> l_vals;
    double st_exchange = 0.0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    double st_wait = 0.0, st_gather = 0.0, st_describe = 0.0, st_deliver = 0.0, st_kids = 0.0, st_total = 0.0;
    ~IaiDriver() {
        for (auto e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    std::vector<uint32_t> node_q;  // owner (index into the level's integrals) of every node of a round
    // structure-of-arrays description of the innermost integrals of a round (no per-integral objects)
    // (they live in a pinned host block laid out like the device staging buffer: one async copy per chunk; pageable
    // vectors of this size made the runtime pin and unpin them on every copy)
    int64_t* f_par = nullptr;   // [cn] parents of the chunk's contraction
    double* f_x = nullptr;      // [cn] its coordinates
    int64_t* f_slot = nullptr;
    double *f_lo = nullptr, *f_hi = nullptr, *f_at = nullptr, *f_sw = nullptr, *f_tl = nullptr;
    int pin_reserve(int which, size_t bytes);
    int flat_layout(int64_t cn, bool need_tail, int buf);
};

int IaiDriver::pin_reserve(int which, size_t bytes) {
    if (bytes <= s->iai_pin_cap[which]) return ABZ_OK;
    if (s->iai_pin[which]) {
        (void)hipStreamSynchronize(ctx->stream);  // kernels may still read / write the block in place
        (void)hipHostFree(s->iai_pin[which]);
    }
    s->iai_pin[which] = nullptr;
    s->iai_pin_cap[which] = 0;
    const size_t want = bytes + (bytes >> 2) + 4096;
    hipError_t e = hipHostMalloc(&s->iai_pin[which], want, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return ABZ_ERR_NOMEM;
    }
    s->iai_pin_cap[which] = want;
    s->iai_pin_dev[which] = nullptr;
    if (hipHostGetDevicePointer(&s->iai_pin_dev[which], s->iai_pin[which], 0) != hipSuccess) {
        (void)hipGetLastError();
        s->iai_pin_dev[which] = nullptr;  // the panel path then copies instead of reading / writing the block in place
    }
    return ABZ_OK;
}

// pinned input block of a chunk of cn innermost integrals: [par | x | slot | lo | hi | at | sw | tail]
int IaiDriver::flat_layout(int64_t cn, bool need_tail, int buf) {
    const size_t words = (size_t)cn * (7 + (need_tail ? (size_t)(d - 1) : 0));
    int rc = pin_reserve(buf, sizeof(double) * words);
    if (rc) return rc;
    char* b = static_cast<char*>(s->iai_pin[buf]);
    f_par = reinterpret_cast<int64_t*>(b);
    f_x = reinterpret_cast<double*>(b + sizeof(int64_t) * (size_t)cn);
    f_slot = reinterpret_cast<int64_t*>(f_x + cn);
    f_lo = reinterpret_cast<double*>(f_slot + cn);
    f_hi = f_lo + cn;
    f_at = f_hi + cn;
    f_sw = f_at + cn;
    f_tl = need_tail ? f_sw + cn : nullptr;
    return ABZ_OK;
}

// upload h_parents/h_x (or take them where they are on the device), contract level-L sets into level-(L-1) pool slots
// base_slot..base_slot+nn-1
int IaiDriver::contract_nodes(int L, int64_t nn, int64_t base_slot, int64_t off, const int64_t* par, const double* x,
                              const int64_t* par_dev, const double* x_dev) {
    const int M = s->dims[L - 1];
    const int64_t Lrow = set_elems(L - 1);
    int rc;
    if (!par_dev) {
        if ((rc = s->iai_io[0].reserve(sizeof(int64_t) * (size_t)nn))) return rc;
        if ((rc = s->iai_io[1].reserve(sizeof(double) * (size_t)nn))) return rc;
    }
    if ((rc = s->iai_io[4].reserve(sizeof(double2) * (size_t)(nn * M)))) return rc;
    if ((rc = s->iai_pool[L - 1].reserve(sizeof(double2) * (size_t)((base_slot + nn) * Lrow)))) return rc;
    if (!par_dev) {
        ABZ_HIP(hipMemcpyAsync(s->iai_io[0].p, par ? par : h_parents.data() + off, sizeof(int64_t) * (size_t)nn, hipMemcpyHostToDevice,
                               ctx->stream));
        ABZ_HIP(hipMemcpyAsync(s->iai_io[1].p, x ? x : h_x.data() + off, sizeof(double) * (size_t)nn, hipMemcpyHostToDevice, ctx->stream));
        par_dev = s->iai_io[0].as<int64_t>();
        x_dev = s->iai_io[1].as<double>();
    }
    PhaseSpec ps;
    ps.B = nn;
    ps.M = M;
    ps.first = s->first[L - 1];
    ps.gi = nullptr;
    ps.x = x_dev;
    ps.tab = nullptr;
    ps.npt = 0;
    ps.period = s->period[L - 1];
    ps.deriv = false;
    if ((rc = launch_phases(ctx, ps, s->iai_io[4].as<double2>()))) return rc;
    const double2* src = (L == d) ? top_coef() : s->iai_pool[L].as<double2>();
    double2* out = s->iai_pool[L - 1].as<double2>() + base_slot * Lrow;
    return launch_contract(ctx, src, set_elems(L), par_dev, 1, s->iai_io[4].as<double2>(), out, nn, Lrow, M);
}

// Panel mode of the level above the innermost one: panels [c0p, c0p + cnp) of the round (pp_*) go to the device as they
// are -- 40 B each -- and the device makes their nodes (panel_nodes_kernel), contracts the nodes' coefficient sets, runs
// the innermost adaptive loops and folds their results back into one (I, E, count, status) per panel (panel_rule_kernel):
// 16 ncomp + 20 B come back.  Enqueue only; panels_collect waits.
int IaiDriver::panels_enqueue(int L, int64_t c0p, int64_t cnp, int buf, const Lims& lims) {
    launches += 1;
    const int64_t cn = 15 * cnp;
    int rc;
    // pinned input block [slot | a | b | at | sw], one asynchronous copy
    const size_t in_bytes = sizeof(double) * 5 * (size_t)cnp;
    if ((rc = pin_reserve(buf, in_bytes))) return rc;
    {
        char* b = static_cast<char*>(s->iai_pin[buf]);
        std::memcpy(b, pp_slot.data() + c0p, sizeof(int64_t) * (size_t)cnp);
        double* f = reinterpret_cast<double*>(b) + cnp;
        std::memcpy(f, pp_a.data() + c0p, sizeof(double) * (size_t)cnp);
        std::memcpy(f + cnp, pp_b.data() + c0p, sizeof(double) * (size_t)cnp);
        std::memcpy(f + 2 * cnp, pp_at.data() + c0p, sizeof(double) * (size_t)cnp);
        std::memcpy(f + 3 * cnp, pp_sw.data() + c0p, sizeof(double) * (size_t)cnp);
    }
    // the kernel reads the panels where the host wrote them (pinned, device-visible: no copy operation in the stream)
    const char* in_dev = static_cast<const char*>(s->iai_pin_dev[buf]);
    if (!in_dev) {
        if ((rc = s->iai_io[2].reserve(in_bytes))) return rc;
        ABZ_HIP(hipMemcpyAsync(s->iai_io[2].p, s->iai_pin[buf], in_bytes, hipMemcpyHostToDevice, ctx->stream));
        in_dev = static_cast<const char*>(s->iai_io[2].p);
    }
    // [slot | lo | hi | at | sw] of the innermost loops, per node
    if ((rc = s->iai_io[0].reserve(sizeof(double) * 5 * (size_t)cn))) return rc;
    PanelNodesSpec pn;
    pn.npanels = cnp;
    pn.p_slot = reinterpret_cast<const int64_t*>(in_dev);
    pn.p_a = reinterpret_cast<const double*>(in_dev) + cnp;
    pn.p_b = pn.p_a + cnp;
    pn.p_at = pn.p_b + cnp;
    pn.p_sw = pn.p_at + cnp;
    pn.lims_kind = lims.kind;
    pn.a0 = lims.a[0];
    pn.b0 = lims.b[0];
    pn.aL = lims.a[L - 1];
    pn.n_slot = s->iai_io[0].as<int64_t>();
    pn.n_lo = s->iai_io[0].as<double>() + cn;
    pn.n_hi = pn.n_lo + cn;
    pn.n_at = pn.n_hi + cn;
    pn.n_sw = pn.n_at + cn;
    s->iai_used[L - 1] = 0;  // sets of the previous chunk are dead (stream order)
    {
        const int64_t Lrow = set_elems(L - 1);
        if ((rc = s->iai_pool[L - 1].reserve(sizeof(double2) * (size_t)(cn * Lrow)))) return rc;
        const double2* src = (L == d) ? top_coef() : s->iai_pool[L].as<double2>();
        if ((rc = launch_panel_contract(ctx, pn, src, set_elems(L), s->dims[L - 1], s->first[L - 1], s->period[L - 1],
                                        s->iai_pool[L - 1].as<double2>(), Lrow)))
            return rc;
    }
    // innermost loops: node outputs, then the panel outputs behind them
    const size_t node_out = flat_out_bytes(cn);
    const size_t node_out_al = (node_out + 15) / 16 * 16;
    const size_t pan_out = sizeof(double2) * (size_t)(cnp * ncomp) + sizeof(double) * (size_t)cnp + sizeof(int64_t) * (size_t)cnp + sizeof(int) * (size_t)cnp;
    if ((rc = s->iai_io[3].reserve(node_out_al + pan_out))) return rc;
    char* ob = static_cast<char*>(s->iai_io[3].p);
    InnerSpec is;
    is.n = n;
    is.d = d;
    is.M = s->dims[0];
    is.first = s->first[0];
    is.period = s->period[0];
    is.src = s->iai_pool[1].as<double2>();
    is.packed = pk;
    is.nint = cn;
    is.slot = pn.n_slot;
    is.lo = pn.n_lo;
    is.hi = pn.n_hi;
    is.atol = pn.n_at;
    is.tail = nullptr;
    is.integrand = integrand;
    for (int i = 0; i < 4; ++i) is.params[i] = params[i];
    is.sweep = sweep;
    is.sweep_arr = pn.n_sw;
    is.herm = s->hermitian;
    is.has_rtol = has_rtol;
    is.rtol_user = rtol_user;
    is.maxevals = maxevals;
    is.I_out = reinterpret_cast<double2*>(ob);
    is.E_out = reinterpret_cast<double*>(ob + sizeof(double2) * (size_t)(cn * ncomp));
    is.nev_out = reinterpret_cast<int64_t*>(is.E_out + cn);
    is.status_out = reinterpret_cast<int*>(is.nev_out + cn);
    if ((rc = (n > 4 ? launch_gen_inner_adaptive(ctx, is) : launch_inner_adaptive(ctx, is)))) return rc;
    PanelRuleSpec pr;
    pr.npanels = cnp;
    pr.ncomp = ncomp;
    pr.p_a = pn.p_a;
    pr.p_b = pn.p_b;
    pr.n_I = is.I_out;
    pr.n_nev = is.nev_out;
    pr.n_status = is.status_out;
    // ... written by the rule kernel straight into the pinned output block where that is device-visible
    if ((rc = pin_reserve(2 + buf, pan_out))) return rc;
    char* const pb_map = static_cast<char*>(s->iai_pin_dev[2 + buf]);
    char* pb = pb_map ? pb_map : ob + node_out_al;
    pr.p_I = reinterpret_cast<double2*>(pb);
    pr.p_E = reinterpret_cast<double*>(pb + sizeof(double2) * (size_t)(cnp * ncomp));
    pr.p_nev = reinterpret_cast<int64_t*>(pr.p_E + cnp);
    pr.p_status = reinterpret_cast<int*>(pr.p_nev + cnp);
    if ((rc = launch_panel_rule(ctx, pr))) return rc;
    if (!pb_map) ABZ_HIP(hipMemcpyAsync(s->iai_pin[2 + buf], pb, pan_out, hipMemcpyDeviceToHost, ctx->stream));
    if (!ev[buf]) ABZ_HIP(hipEventCreateWithFlags(&ev[buf], hipEventDisableTiming));
    ABZ_HIP(hipEventRecord(ev[buf], ctx->stream));
    if (stats) {
        int b = 0;
        while (((int64_t)1 << (b + 1)) <= cn) ++b;
        st_cnt[b] += 1;
        st_int[b] += cn;
    }
    return ABZ_OK;
}

// Wait for the panel chunk in `buf`; its results go to pI / pE / pnev at round positions c0p...; panels one of whose
// innermost integrals overflowed the device store are listed in `redo` (round positions)
int IaiDriver::panels_collect(int64_t c0p, int64_t cnp, int buf, std::vector<int64_t>& redo) {
    const auto t0 = std::chrono::steady_clock::now();
    ABZ_HIP(hipEventSynchronize(ev[buf]));
    if (stats) st_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const char* f_out = static_cast<const char*>(s->iai_pin[2 + buf]);
    const cd* hI = reinterpret_cast<const cd*>(f_out);
    const double* hE = reinterpret_cast<const double*>(f_out + sizeof(double2) * (size_t)(cnp * ncomp));
    const int64_t* hN = reinterpret_cast<const int64_t*>(hE + cnp);
    const int* hS = reinterpret_cast<const int*>(hN + cnp);
    std::memcpy(pI.data() + (size_t)(c0p * ncomp), hI, sizeof(cd) * (size_t)(cnp * ncomp));
    std::memcpy(pE.data() + (size_t)c0p, hE, sizeof(double) * (size_t)cnp);
    std::memcpy(pnev.data() + (size_t)c0p, hN, sizeof(int64_t) * (size_t)cnp);
    for (int64_t i = 0; i < cnp; ++i)
        if (hS[i] != 0) redo.push_back(c0p + i);
    return ABZ_OK;
}

// innermost: h_parents / h_x / h_tail -> h_values [nn][ncomp]
int IaiDriver::eval_nodes(int64_t nn) {
    int rc;
    if ((rc = s->iai_io[0].reserve(sizeof(int64_t) * (size_t)nn))) return rc;
    if ((rc = s->iai_io[1].reserve(sizeof(double) * (size_t)nn))) return rc;
    if ((rc = s->iai_io[3].reserve(sizeof(double2) * (size_t)(nn * ncomp)))) return rc;
    ABZ_HIP(hipMemcpyAsync(s->iai_io[0].p, h_parents.data(), sizeof(int64_t) * (size_t)nn, hipMemcpyHostToDevice,
                           ctx->stream));
    ABZ_HIP(hipMemcpyAsync(s->iai_io[1].p, h_x.data(), sizeof(double) * (size_t)nn, hipMemcpyHostToDevice, ctx->stream));
    const bool need_tail = integrand == ABZ_F_LINEAR_X && d > 1;
    if (need_tail) {
        if ((rc = s->iai_io[2].reserve(sizeof(double) * (size_t)(nn * (d - 1))))) return rc;
        ABZ_HIP(hipMemcpyAsync(s->iai_io[2].p, h_tail.data(), sizeof(double) * (size_t)(nn * (d - 1)),
                               hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = s->iai_io[5].reserve(sizeof(double) * (size_t)nn))) return rc;
    ABZ_HIP(hipMemcpyAsync(s->iai_io[5].p, h_sweep.data(), sizeof(double) * (size_t)nn, hipMemcpyHostToDevice, ctx->stream));
    NodeEvalSpec ns;
    ns.panels15 = panels15;
    ns.sweep_arr = s->iai_io[5].as<double>();
    ns.n = n;
    ns.d = d;
    ns.M = s->dims[0];
    ns.first = s->first[0];
    ns.period = s->period[0];
    ns.src = (d == 1) ? top_coef() : s->iai_pool[1].as<double2>();
    ns.packed = pk;
    ns.herm = s->hermitian;
    ns.parents = s->iai_io[0].as<int64_t>();
    ns.x = s->iai_io[1].as<double>();
    ns.tail = need_tail ? s->iai_io[2].as<double>() : nullptr;
    ns.nnodes = nn;
    ns.integrand = integrand;
    for (int i = 0; i < 4; ++i) ns.params[i] = params[i];
    ns.sweep = sweep;
    if ((rc = launch_node_integrand(ctx, ns, s->iai_io[3].as<double2>()))) return rc;
    h_values.resize((size_t)(nn * ncomp));
    ABZ_HIP(hipMemcpyAsync(h_values.data(), s->iai_io[3].p, sizeof(double2) * (size_t)(nn * ncomp),
                           hipMemcpyDeviceToHost, ctx->stream));
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    launches += 1;
    return ABZ_OK;
}

// All integrals of `kids` integrate variable 1: run their whole adaptive loops on the device
// (inner_adaptive_kernel); an integral that overflows the device segment store is redone on the host.
// The same from structure-of-arrays input (f_slot, f_lo, ...), results straight into vals [nq][ncomp]:
// a 432-solve sweep creates ~35 M innermost integrals, one heap-backed Quad1D each cost 4/5 of its time.
// Enqueue the innermost adaptive loops of the chunk described in pinned input block `buf` (flat_layout): input copy,
// kernel, output copy into pinned output block `buf`, event.  Nothing is waited for.
int IaiDriver::flat_enqueue(int64_t nq, int buf) {
    launches += 1;
    const bool need_tail = integrand == ABZ_F_LINEAR_X && d > 1;
    int rc;
    const size_t in_bytes = sizeof(int64_t) * (size_t)nq + sizeof(double) * (size_t)nq * 4;
    if ((rc = s->iai_io[0].reserve(in_bytes))) return rc;
    char* base = static_cast<char*>(s->iai_io[0].p);
    int64_t* d_slot = reinterpret_cast<int64_t*>(base);
    double* d_lo = reinterpret_cast<double*>(base + sizeof(int64_t) * (size_t)nq);
    double* d_hi = d_lo + nq;
    double* d_at = d_hi + nq;
    double* d_sw = d_at + nq;
    // [slot | lo | hi | at | sw] is contiguous in the pinned block, in the device block's order
    ABZ_HIP(hipMemcpyAsync(d_slot, f_slot, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    double* d_tail = nullptr;
    if (need_tail) {
        if ((rc = s->iai_io[2].reserve(sizeof(double) * (size_t)(nq * (d - 1))))) return rc;
        d_tail = s->iai_io[2].as<double>();
        ABZ_HIP(hipMemcpyAsync(d_tail, f_tl, sizeof(double) * (size_t)(nq * (d - 1)), hipMemcpyHostToDevice, ctx->stream));
    }
    const size_t out_bytes = flat_out_bytes(nq);
    if ((rc = s->iai_io[3].reserve(out_bytes))) return rc;
    char* ob = static_cast<char*>(s->iai_io[3].p);
    InnerSpec is;
    is.n = n;
    is.d = d;
    is.M = s->dims[0];
    is.first = s->first[0];
    is.period = s->period[0];
    is.src = (d == 1) ? top_coef() : s->iai_pool[1].as<double2>();
    is.packed = pk;
    is.nint = nq;
    is.slot = d_slot;
    is.lo = d_lo;
    is.hi = d_hi;
    is.atol = d_at;
    is.tail = d_tail;
    is.integrand = integrand;
    for (int i = 0; i < 4; ++i) is.params[i] = params[i];
    is.sweep = sweep;
    is.sweep_arr = d_sw;
    is.herm = s->hermitian;
    is.has_rtol = has_rtol;
    is.rtol_user = rtol_user;
    is.maxevals = maxevals;
    is.I_out = reinterpret_cast<double2*>(ob);
    is.E_out = reinterpret_cast<double*>(ob + sizeof(double2) * (size_t)(nq * ncomp));
    is.nev_out = reinterpret_cast<int64_t*>(is.E_out + nq);
    is.status_out = reinterpret_cast<int*>(is.nev_out + nq);
    if ((rc = (n > 4 ? launch_gen_inner_adaptive(ctx, is) : launch_inner_adaptive(ctx, is)))) return rc;
    if ((rc = pin_reserve(2 + buf, out_bytes))) return rc;
    ABZ_HIP(hipMemcpyAsync(s->iai_pin[2 + buf], ob, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (!ev[buf]) ABZ_HIP(hipEventCreateWithFlags(&ev[buf], hipEventDisableTiming));
    ABZ_HIP(hipEventRecord(ev[buf], ctx->stream));
    if (stats) {
        int b = 0;
        while (((int64_t)1 << (b + 1)) <= nq) ++b;
        st_cnt[b] += 1;
        st_int[b] += nq;
    }
    return ABZ_OK;
}

// Wait for chunk `buf` and take its results; integrals that overflowed the device segment store are listed in `redo`
// (chunk-local indices) for the host loop.
int IaiDriver::flat_collect(int64_t nq, int buf, cd* vals, int64_t* nev, std::vector<int64_t>& redo) {
    const auto t0 = std::chrono::steady_clock::now();
    ABZ_HIP(hipEventSynchronize(ev[buf]));
    if (stats) st_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const char* f_out = static_cast<const char*>(s->iai_pin[2 + buf]);
    const cd* hI = reinterpret_cast<const cd*>(f_out);
    const double* hE = reinterpret_cast<const double*>(f_out + sizeof(double2) * (size_t)(nq * ncomp));
    const int64_t* hN = reinterpret_cast<const int64_t*>(hE + nq);
    const int* hS = reinterpret_cast<const int*>(hN + nq);
    std::memcpy(vals, hI, sizeof(cd) * (size_t)(nq * ncomp));
    std::memcpy(nev, hN, sizeof(int64_t) * (size_t)nq);
    for (int64_t i = 0; i < nq; ++i)
        if (hS[i] != 0) redo.push_back(i);
    return ABZ_OK;
}

// Run every integral of `quads` (all integrate variable L) to completion.
//
// Refinement is QuadGK's scalar rule -- pop the panel with the largest error, bisect it, replace it by its halves --
// and every integral applies its pops strictly in that order, so values, errors, panel trees and evaluation counts are
// those of the depth-first reference.  What is NOT serialised is the evaluation of the halves: with a fixed tolerance
// (reltol = 0) a panel k is certain to be popped before the loop can stop as soon as the errors of the panels that
// will still be in the heap when k reaches its top -- k itself and every panel with a smaller error -- add up to more
// than the tolerance (error estimates are non-negative, so E_tot >= that sum > tol at that moment).  Every round
// therefore requests the halves of ALL such panels at once, for all sibling integrals, and replays the pops as their
// halves arrive.  A sweep of rounds is then as deep as the panel tree (tens) instead of as long as the pop sequence
// (10^5 for config 5 at abstol 1e-3, each with a host round trip), and a round carries 10^4..10^6 innermost integrals
// instead of ~100.  Contracted coefficient sets of a round are produced and consumed in chunks of `pool_cap_bytes`.
// With a relative tolerance in play, or a finite maxevals, only the top panel is requested (the round-1 behaviour);
// the BatchIntegrand refinement rule (max_batch > 0) pops its panels together as before.
int IaiDriver::solve_level(int L, std::vector<Quad1D>& quads) {
    std::vector<size_t> active;
    for (size_t i = 0; i < quads.size(); ++i) {
        Quad1D& q = quads[i];
        q.pend.clear();
        double lo1, hi1;
        if (q.lims.range(L, lo1, hi1)) {
            q.pend.push_back(Seg{lo1, hi1, 0.0, 0});
        } else {
            std::vector<double> sg;
            q.lims.segs(L, sg);
            for (size_t k = 0; k + 1 < sg.size(); ++k) q.pend.push_back(Seg{sg[k], sg[k + 1], 0.0, 0});
        }
        q.popped.clear();
        q.req.clear();
        q.heap.clear();
        q.store.clear();
        q.started = false;
        q.done = false;
        q.numevals = 0;
        q.fevals = 0;
        q.I.assign((size_t)ncomp, cd(0, 0));
        active.push_back(i);
    }
    const bool unlimited = maxevals >= ((int64_t)1 << 62);
    std::vector<cd> vals;        // [node][ncomp] of this round
    std::vector<int64_t> nev;    // [node]: integrand evaluations beneath the node
    std::vector<Quad1D> kids;    // inner integrals of a chunk (L - 1 > 1, or host-side innermost loops)
    std::vector<int64_t> act_off;  // first node of every active integral in this round's node list
    while (!active.empty()) {
        st_rounds[L] += 1;
        // ---- gather the nodes of all pending panels
        const auto tg0 = std::chrono::steady_clock::now();
        int64_t nn = 0;
        act_off.resize(active.size());
        for (size_t ai = 0; ai < active.size(); ++ai) {
            act_off[ai] = nn;
            nn += 15 * (int64_t)quads[active[ai]].pend.size();
        }
        const bool plain_lims = quads[active[0]].lims.kind <= ABZ_LIMS_TETRAHEDRAL;
        // Panel mode of the level above the innermost one (closed-form limits, the whole round on this GPU): the round's PANELS
        // go to the device, which makes their nodes, runs the innermost loops and returns one GK sum per panel
        // (panels_enqueue) -- the host never sees a node.  Values, errors and counts are those of the node path: same
        // gk15_node, same kernels, same gk15_rule, on the other side of the bus.
        const bool panel_mode = L == 2 && device_inner && plain_lims && nn > 0 && !(integrand == ABZ_F_LINEAR_X && d > 1) &&
                                !(ex_fn != nullptr && nn >= (int64_t)64 * ex_world) && abz_switch(SW_IAI_PANELS) != 0;
        if (panel_mode) {
            const int64_t npan = nn / 15;
            pp_slot.resize((size_t)npan);
            pp_a.resize((size_t)npan);
            pp_b.resize((size_t)npan);
            pp_at.resize((size_t)npan);
            pp_sw.resize((size_t)npan);
            pp_q.resize((size_t)npan);
            pI.resize((size_t)(npan * ncomp));
            pE.resize((size_t)npan);
            pnev.resize((size_t)npan);
            par((int64_t)active.size(), true, [&](int64_t b, int64_t e, int) {
                for (int64_t ai = b; ai < e; ++ai) {
                    const size_t qi = active[(size_t)ai];
                    const Quad1D& q = quads[qi];
                    int64_t pi = act_off[(size_t)ai] / 15;
                    for (size_t p = 0; p < q.pend.size(); ++p, ++pi) {
                        pp_slot[(size_t)pi] = q.slot;
                        pp_a[(size_t)pi] = q.pend[p].a;
                        pp_b[(size_t)pi] = q.pend[p].b;
                        pp_at[(size_t)pi] = q.has_atol ? q.atol : -1.0;
                        pp_sw[(size_t)pi] = q.sweep;
                        pp_q[(size_t)pi] = (uint32_t)qi;
                    }
                }
            });
        } else {
        h_parents.resize((size_t)nn);
        h_x.resize((size_t)nn);
        node_q.resize((size_t)nn);
        if (L == 1) h_sweep.resize((size_t)nn);
        if (d > 1) h_tail.resize((size_t)(nn * (d - 1)));
        par((int64_t)active.size(), true, [&](int64_t b, int64_t e, int) {
            double x15[15];
            for (int64_t ai = b; ai < e; ++ai) {
                const size_t qi = active[(size_t)ai];
                Quad1D& q = quads[qi];
                int64_t tt = act_off[(size_t)ai];
                for (size_t p = 0; p < q.pend.size(); ++p) {
                    gk15_nodes(q.pend[p].a, q.pend[p].b, x15);
                    for (int i = 0; i < 15; ++i, ++tt) {
                        h_parents[(size_t)tt] = q.slot;
                        h_x[(size_t)tt] = x15[i];
                        node_q[(size_t)tt] = (uint32_t)qi;
                        if (L == 1) h_sweep[(size_t)tt] = q.sweep;
                        if (d > 1 && L == 1)
                            for (int j = 0; j < d - 1; ++j) h_tail[(size_t)(tt * (d - 1) + j)] = q.tail[j];
                    }
                }
            }
        });
        vals.resize((size_t)(nn * ncomp));
        nev.resize((size_t)nn);
        }
        if (stats) st_gather += std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count();
        // requests ahead of the pops pay when a round is small (a single solve's stragglers); a round that fills the
        // chip anyway (a 432-omega sweep: 4e5 nodes per round) only pays their bookkeeping (+20 % host time measured)
        const bool spec_round = speculate && nn < spec_cap_nodes;
        bool panel_round = false;  // this round's GK sums were formed on the device (pI, pE, pnev per panel)
        // ---- evaluate them
        if (L == 1) {
            int rc = eval_nodes(nn);
            if (rc) return rc;
            std::copy(h_values.begin(), h_values.begin() + (size_t)(nn * ncomp), vals.begin());
            std::fill(nev.begin(), nev.end(), (int64_t)1);
        } else {
            // chunks: the level-(L-1) sets of a chunk are contracted, integrated over and then overwritten
            const int64_t set_bytes = (int64_t)sizeof(double2) * set_elems(L - 1);
            // (and of at most ~2.6e5 nodes: the per-chunk host arrays stay in cache, a launch still fills the chip)
            int64_t chunk = std::max<int64_t>(15, std::min<int64_t>(pool_cap_bytes / std::max<int64_t>(set_bytes, 1), 262140) / 15 * 15);
            const bool flat = (L - 1 == 1) && device_inner;
            const bool need_tail = integrand == ABZ_F_LINEAR_X && d > 1;
            if (flat) {
                // Two chunks in flight: while the GPU integrates chunk c the host describes chunk c + 1 (limits,
                // tolerances, slots) in the other pinned block.  Everything is ordered on one stream, so the device
                // buffers and the set pool need no second copy.
                std::vector<int64_t> redo_local, redo_nodes;
                // One solve on several GPUs: the round's innermost integrals are dealt to the ranks in
                // blocks of 64 nodes, every rank integrates its share on its own GPU, one all-gather of (value, count) per
                // node and round puts every rank back in the same state: all ranks run this driver redundantly and take
                // the same decisions (each integral is computed by exactly one GPU with the same kernel: the result
                // equals the single-GPU solve bit for bit).
                const int W = ex_world;
                const bool shard = ex_fn != nullptr && nn >= (int64_t)64 * W;
                auto owner = [W](int64_t tnode) { return (int)((tnode >> 6) % W); };
                if (shard) {
                    l_par.clear();
                    l_x.clear();
                    l_q.clear();
                    for (int64_t tn = 0; tn < nn; ++tn)
                        if (owner(tn) == ex_rank) {
                            l_par.push_back(h_parents[(size_t)tn]);
                            l_x.push_back(h_x[(size_t)tn]);
                            l_q.push_back(node_q[(size_t)tn]);
                        }
                    l_vals.resize(l_x.size() * (size_t)ncomp);
                    l_nev.resize(l_x.size());
                }
                // the nodes this rank integrates node by node (all of them, its share of a sharded solve, or -- panel mode --
                // the nodes of the few panels that have to be redone)
                int64_t N = shard ? (int64_t)l_x.size() : nn;
                const int64_t* P = shard ? l_par.data() : h_parents.data();
                const double* X = shard ? l_x.data() : h_x.data();
                const uint32_t* Q = shard ? l_q.data() : node_q.data();
                cd* V = shard ? l_vals.data() : vals.data();
                int64_t* NV = shard ? l_nev.data() : nev.data();
                auto describe_and_enqueue = [&](int64_t c0, int64_t cn, int buf) -> int {
                    const auto td0 = std::chrono::steady_clock::now();
                    struct Acc { double& a; std::chrono::steady_clock::time_point t0; bool on; ~Acc() { if (on) a += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } } acc_{st_describe, td0, stats};
                    int rc;
                    if ((rc = flat_layout(cn, need_tail, buf))) return rc;
                    std::memcpy(f_par, P + c0, sizeof(int64_t) * (size_t)cn);
                    std::memcpy(f_x, X + c0, sizeof(double) * (size_t)cn);
                    s->iai_used[L - 1] = 0;  // sets of the previous chunk are dead (stream order)
                    if ((rc = contract_nodes(L, cn, 0, 0, f_par, f_x))) return rc;
                    par(cn, plain_lims, [&](int64_t ub, int64_t ue, int) {
                        for (int64_t u = ub; u < ue; ++u) {
                            const Quad1D& q = quads[Q[(size_t)(c0 + u)]];
                            const double x = X[(size_t)(c0 + u)];
                            const Lims kl = q.lims.fix(L, x);
                            double lo1, hi1;
                            if (!kl.range(1, lo1, hi1)) {
                                std::vector<double> sg;
                                kl.segs(1, sg);
                                lo1 = sg.front();
                                hi1 = sg.back();  // the innermost slice of a convex domain is one interval
                            }
                            f_slot[(size_t)u] = u;
                            f_sw[(size_t)u] = q.sweep;
                            f_lo[(size_t)u] = lo1;
                            f_hi[(size_t)u] = hi1;
                            f_at[(size_t)u] = q.has_atol ? q.atol / (hi1 - lo1) : -1.0;  // ref src/fourier.jl:479-480
                            if (need_tail) {
                                f_tl[(size_t)(u * (d - 1))] = x;
                                for (int j = 1; j < d - 1; ++j) f_tl[(size_t)(u * (d - 1) + j)] = q.tail[j - 1];
                            }
                        }
                    });
                    return flat_enqueue(cn, buf);
                };
                // This rank's share.  In a sharded solve a failure here (a HIP error, out of memory) must not return before
                // the exchange below: the other ranks would wait in the collective for ever.  The status travels with the data.
                auto local_work = [&]() -> int {
                    const int64_t nchunks = (N + chunk - 1) / chunk;
                    for (int64_t ci = 0; ci <= nchunks; ++ci) {
                        if (ci < nchunks) {
                            const int64_t c0 = ci * chunk;
                            int rc = describe_and_enqueue(c0, std::min(chunk, N - c0), (int)(ci & 1));
                            if (rc) return rc;
                        }
                        if (ci > 0) {
                            const int64_t c0 = (ci - 1) * chunk, cn = std::min(chunk, N - c0);
                            redo_local.clear();
                            int rc = flat_collect(cn, (int)((ci - 1) & 1), V + (size_t)(c0 * ncomp), NV + (size_t)c0, redo_local);
                            if (rc) return rc;
                            for (int64_t u : redo_local) redo_nodes.push_back(c0 + u);
                        }
                    }
                    // integrals that overflowed the device segment store: their sets are contracted again (the pool has
                    // moved on) and the host loop integrates them
                    for (size_t r0 = 0; r0 < redo_nodes.size(); r0 += (size_t)chunk) {
                        const int64_t rn = (int64_t)std::min<size_t>((size_t)chunk, redo_nodes.size() - r0);
                        std::vector<int64_t> rp((size_t)rn);
                        std::vector<double> rx((size_t)rn);
                        kids.resize((size_t)rn);
                        for (int64_t u = 0; u < rn; ++u) {
                            const int64_t tn = redo_nodes[r0 + (size_t)u];
                            const Quad1D& q = quads[Q[(size_t)tn]];
                            Quad1D& k = kids[(size_t)u];
                            const double x = X[(size_t)tn];
                            rp[(size_t)u] = P[(size_t)tn];
                            rx[(size_t)u] = x;
                            k.slot = u;
                            k.sweep = q.sweep;
                            k.root = q.root;
                            k.tail[0] = x;
                            for (int j = 1; j < ABZ_MAX_DIM; ++j) k.tail[j] = q.tail[j - 1];
                            k.lims = q.lims.fix(L, x);
                            double lo1, hi1;
                            if (!k.lims.range(1, lo1, hi1)) {
                                std::vector<double> sg;
                                k.lims.segs(1, sg);
                                lo1 = sg.front();
                                hi1 = sg.back();
                            }
                            k.has_atol = q.has_atol;
                            k.atol = q.has_atol ? q.atol / (hi1 - lo1) : 0.0;
                        }
                        s->iai_used[L - 1] = 0;
                        int rc = contract_nodes(L, rn, 0, 0, rp.data(), rx.data());
                        if (rc) return rc;
                        std::vector<int64_t> keep_par;
                        std::vector<double> keep_x;
                        std::vector<uint32_t> keep_q;
                        keep_par.swap(h_parents);
                        keep_x.swap(h_x);
                        keep_q.swap(node_q);
                        const bool keep = device_inner;
                        device_inner = false;
                        rc = solve_level(1, kids);
                        device_inner = keep;
                        h_parents.swap(keep_par);
                        h_x.swap(keep_x);
                        node_q.swap(keep_q);
                        if (rc) return rc;
                        for (int64_t u = 0; u < rn; ++u) {
                            const int64_t tn = redo_nodes[r0 + (size_t)u];
                            for (int c = 0; c < ncomp; ++c) V[(size_t)(tn * ncomp + c)] = kids[(size_t)u].I[(size_t)c];
                            NV[(size_t)tn] = kids[(size_t)u].fevals;
                        }
                    }
                    return ABZ_OK;
                };
                int local_rc = ABZ_OK;
                if (panel_mode) {
                    const int64_t npan = nn / 15;
                    const Lims& lims0 = quads[active[0]].lims;  // closed-form limits: a, b are those of the solve
                    const int64_t pchunk = chunk / 15;
                    const int64_t nchunks = (npan + pchunk - 1) / pchunk;
                    std::vector<int64_t> redo_pan;
                    for (int64_t ci = 0; ci <= nchunks && local_rc == ABZ_OK; ++ci) {
                        if (ci < nchunks) {
                            const auto td0 = std::chrono::steady_clock::now();
                            local_rc = panels_enqueue(L, ci * pchunk, std::min(pchunk, npan - ci * pchunk), (int)(ci & 1), lims0);
                            if (stats) st_describe += std::chrono::duration<double>(std::chrono::steady_clock::now() - td0).count();
                        }
                        if (ci > 0 && local_rc == ABZ_OK)
                            local_rc = panels_collect((ci - 1) * pchunk, std::min(pchunk, npan - (ci - 1) * pchunk), (int)((ci - 1) & 1), redo_pan);
                    }
                    if (local_rc) return local_rc;
                    panel_round = true;
                    if (!redo_pan.empty()) {
                        // a panel with an innermost integral beyond the device's segment store: its fifteen nodes take the
                        // node path (whose own overflow handling sends them to the host loop), the host applies the rule
                        l_par.clear();
                        l_x.clear();
                        l_q.clear();
                        for (int64_t pi : redo_pan) {
                            double x15[15];
                            gk15_nodes(pp_a[(size_t)pi], pp_b[(size_t)pi], x15);
                            for (int i = 0; i < 15; ++i) {
                                l_par.push_back(pp_slot[(size_t)pi]);
                                l_x.push_back(x15[i]);
                                l_q.push_back(pp_q[(size_t)pi]);
                            }
                        }
                        l_vals.resize(l_x.size() * (size_t)ncomp);
                        l_nev.resize(l_x.size());
                        N = (int64_t)l_x.size();
                        P = l_par.data();
                        X = l_x.data();
                        Q = l_q.data();
                        V = l_vals.data();
                        NV = l_nev.data();
                        redo_nodes.clear();
                        local_rc = local_work();
                        if (local_rc) return local_rc;
                        for (size_t r = 0; r < redo_pan.size(); ++r) {
                            const int64_t pi = redo_pan[r];
                            gk15_evalrule(&l_vals[r * 15 * (size_t)ncomp], ncomp, pp_a[(size_t)pi], pp_b[(size_t)pi], &pI[(size_t)(pi * ncomp)], &pE[(size_t)pi]);
                            int64_t tot = 0;
                            for (int i = 0; i < 15; ++i) tot += l_nev[r * 15 + (size_t)i];
                            pnev[(size_t)pi] = tot;
                        }
                    }
                } else {
                    local_rc = local_work();
                }
                if (!shard && local_rc) return local_rc;
                if (shard) {
                    // all-gather: per rank `per` slots of (2 ncomp + 1) doubles -- values and evaluation counts of its nodes
                    int64_t per = 0;
                    for (int rr = 0; rr < W; ++rr) {
                        int64_t c = 0;
                        for (int64_t b0 = (int64_t)rr * 64; b0 < nn; b0 += (int64_t)64 * W) c += std::min<int64_t>(64, nn - b0);
                        per = std::max(per, c);
                    }
                    const int rec = 2 * ncomp + 1;
                    const int64_t seg = per * rec + 1;  // + one status word per rank: 0, or the ABZ_ERR_* code of its local failure
                    // pinned, so that a device-side all-gather (RCCL) stages it at PCIe rate; pageable if pinning itself fails
                    std::vector<double> ex_fallback;
                    double* ex_buf;
                    if (pin_reserve(4, sizeof(double) * (size_t)seg * (size_t)W) == ABZ_OK) {
                        ex_buf = static_cast<double*>(s->iai_pin[4]);
                    } else {
                        ex_fallback.assign((size_t)seg * (size_t)W, 0.0);
                        ex_buf = ex_fallback.data();
                    }
                    double* mine = ex_buf + (size_t)ex_rank * (size_t)seg;
                    for (int64_t i = 0; i < seg; ++i) mine[i] = 0.0;
                    if (local_rc == ABZ_OK) {
                        for (int64_t i = 0; i < N; ++i) {
                            for (int c = 0; c < ncomp; ++c) {
                                mine[i * rec + 2 * c] = V[(size_t)(i * ncomp + c)].real();
                                mine[i * rec + 2 * c + 1] = V[(size_t)(i * ncomp + c)].imag();
                            }
                            mine[i * rec + 2 * ncomp] = (double)NV[(size_t)i];  // exact below 2^53
                        }
                    }
                    mine[per * rec] = (double)local_rc;
                    const auto te0 = std::chrono::steady_clock::now();
                    // the callback itself must fail on every rank or on none (it is the caller's collective)
                    if (ex_fn(ex_user, ex_buf, seg) != 0) {
                        set_error("IAI: the exchange callback of a sharded solve failed");
                        return ABZ_ERR_HIP;
                    }
                    for (int rr = 0; rr < W; ++rr) {
                        const int code = (int)ex_buf[(size_t)rr * (size_t)seg + (size_t)(per * rec)];
                        if (code != 0) {  // every rank sees the same words and leaves with the same code
                            if (rr != ex_rank) set_error("IAI: rank %d of the sharded solve failed with code %d", rr, code);
                            return code;
                        }
                    }
                    if (stats) st_exchange += std::chrono::duration<double>(std::chrono::steady_clock::now() - te0).count();
                    std::vector<int64_t> pos((size_t)W, 0);
                    for (int64_t tn = 0; tn < nn; ++tn) {
                        const int rr = owner(tn);
                        const double* src = ex_buf + (size_t)rr * (size_t)seg + (size_t)(pos[(size_t)rr]++ * rec);
                        for (int c = 0; c < ncomp; ++c) vals[(size_t)(tn * ncomp + c)] = cd(src[2 * c], src[2 * c + 1]);
                        nev[(size_t)tn] = (int64_t)src[2 * ncomp];
                    }
                }
            } else
            for (int64_t c0 = 0; c0 < nn; c0 += chunk) {
                const int64_t cn = std::min(chunk, nn - c0);
                s->iai_used[L - 1] = 0;  // sets of the previous chunk are dead
                int rc;
                {
                    const auto tk0 = std::chrono::steady_clock::now();
                    if ((rc = contract_nodes(L, cn, 0, c0))) return rc;
                    kids.resize((size_t)cn);  // elements keep their vectors' capacity from earlier rounds (a fresh Quad1D costs six allocations)
                    for (int64_t u = 0; u < cn; ++u) {
                        const Quad1D& q = quads[node_q[(size_t)(c0 + u)]];
                        Quad1D& k = kids[(size_t)u];
                        const double x = h_x[(size_t)(c0 + u)];
                        k.slot = u;
                        k.sweep = q.sweep;
                        k.root = q.root;
                        k.tail[0] = x;
                        for (int j = 1; j < ABZ_MAX_DIM; ++j) k.tail[j] = q.tail[j - 1];
                        k.lims = q.lims.fix(L, x);
                        double lo1, hi1;
                        if (!k.lims.range(L - 1, lo1, hi1)) {
                            std::vector<double> sg;
                            k.lims.segs(L - 1, sg);
                            lo1 = sg.front();
                            hi1 = sg.back();
                        }
                        const double len = hi1 - lo1;  // ref: len = segs[end] - segs[1]
                        k.has_atol = q.has_atol;
                        k.atol = q.has_atol ? q.atol / len : 0.0;  // ref src/fourier.jl:479-480
                    }
                    // the recursion reuses the staging vectors of this round
                    std::vector<int64_t> keep_par;
                    std::vector<double> keep_x;
                    std::vector<uint32_t> keep_q;
                    keep_par.swap(h_parents);
                    keep_x.swap(h_x);
                    keep_q.swap(node_q);
                    if (stats) st_kids += std::chrono::duration<double>(std::chrono::steady_clock::now() - tk0).count();
                    rc = solve_level(L - 1, kids);
                    h_parents.swap(keep_par);
                    h_x.swap(keep_x);
                    node_q.swap(keep_q);
                    if (rc) return rc;
                    for (int64_t u = 0; u < cn; ++u) {
                        for (int c = 0; c < ncomp; ++c) vals[(size_t)((c0 + u) * ncomp + c)] = kids[(size_t)u].I[(size_t)c];
                        nev[(size_t)(c0 + u)] = kids[(size_t)u].fevals;
                    }
                }
            }
        }
        // ---- deliver: GK sums, then replay the pops whose halves are there, then the next requests
        const auto tv0 = std::chrono::steady_clock::now();
        std::vector<size_t> next;
        std::atomic<int> bad{0};
        double bad_a = 0.0, bad_b = 0.0;
        std::mutex bad_m;
        auto deliver = [&](size_t qi, int64_t t, std::vector<cd>& Iseg, std::vector<Seg>& got,
                           std::vector<std::pair<double, uint32_t>>& order, std::vector<double>& suffix) {
            Quad1D& q = quads[qi];
            const double rt = tol_r(q);
            const double at = q.has_atol ? q.atol : 0.0;
            got.resize(q.pend.size());
            for (size_t p = 0; p < q.pend.size(); ++p, t += 15) {
                Seg sg = q.pend[p];
                if (panel_round) {
                    for (int c = 0; c < ncomp; ++c) Iseg[(size_t)c] = pI[(size_t)((t / 15) * ncomp + c)];
                    sg.E = pE[(size_t)(t / 15)];
                } else {
                    gk15_evalrule(&vals[(size_t)(t * ncomp)], ncomp, sg.a, sg.b, Iseg.data(), &sg.E);
                }
                if (!std::isfinite(sg.E)) {
                    std::lock_guard<std::mutex> lk(bad_m);
                    if (!bad.exchange(1)) {
                        bad_a = sg.a;
                        bad_b = sg.b;
                    }
                    return;
                }
                sg.ioff = (int64_t)q.store.size();
                sg.nev = 0;
                if (panel_round)
                    sg.nev = pnev[(size_t)(t / 15)];
                else
                    for (int i = 0; i < 15; ++i) sg.nev += nev[(size_t)(t + i)];
                q.store.insert(q.store.end(), Iseg.begin(), Iseg.end());
                got[p] = sg;
            }
            if (!q.started) {
                // QuadGK's first pass: every initial segment once, I and E summed in segment order
                q.started = true;
                q.heap.assign(got.begin(), got.end());
                for (int c = 0; c < ncomp; ++c) q.I[(size_t)c] = q.store[(size_t)(got[0].ioff + c)];
                q.E = got[0].E;
                q.fevals = got[0].nev;
                for (size_t h = 1; h < got.size(); ++h) {
                    for (int c = 0; c < ncomp; ++c) q.I[(size_t)c] += q.store[(size_t)(got[h].ioff + c)];
                    q.E += got[h].E;
                    q.fevals += got[h].nev;
                }
                q.numevals = 15 * (int64_t)got.size();
                if (q.E <= std::max(at, rt * vnorm(q.I)) || q.numevals >= maxevals) {
                    q.done = true;
                    q.pend.clear();
                    return;
                }
                heapify(q.heap);
            } else if (max_batch > 0) {
                // BatchIntegrand refinement: children arrive in the order their parents were popped
                for (size_t k = 0; k < q.popped.size(); ++k) {
                    const Seg& par = q.popped[k];
                    const Seg& s1 = got[2 * k];
                    const Seg& s2 = got[2 * k + 1];
                    for (int c = 0; c < ncomp; ++c)
                        q.I[(size_t)c] = (q.I[(size_t)c] - q.store[(size_t)(par.ioff + c)]) +
                                         q.store[(size_t)(s1.ioff + c)] + q.store[(size_t)(s2.ioff + c)];
                    q.E = (q.E - par.E) + s1.E + s2.E;
                    q.fevals += s1.nev + s2.nev;
                    heap_push(q.heap, s1);
                    heap_push(q.heap, s2);
                }
            } else {
                // halves requested ahead: attach them to their parents (heap positions are stable between the
                // request and this point)
                for (size_t k = 0; k < q.req.size(); ++k) {
                    Seg& par = q.heap[q.req[k]];
                    par.state = 2;
                    par.E1 = got[2 * k].E;
                    par.ioff1 = got[2 * k].ioff;
                    par.nev1 = got[2 * k].nev;
                    par.E2 = got[2 * k + 1].E;
                    par.ioff2 = got[2 * k + 1].ioff;
                    par.nev2 = got[2 * k + 1].nev;
                }
            }
            q.pend.clear();
            q.popped.clear();
            q.req.clear();
            bool finished = false;
            double tol = std::max(at, rt * vnorm(q.I));
            if (max_batch <= 0) {
                // replay QuadGK's adapt loop for as long as the halves of the top panel are known
                while (true) {
                    tol = std::max(at, rt * vnorm(q.I));
                    if (!(q.E > tol && q.numevals < maxevals)) {
                        finished = true;
                        break;
                    }
                    if (q.heap[0].state != 2) break;
                    const Seg par = heap_pop(q.heap);
                    q.numevals += 30;
                    const double mid = (par.a + par.b) / 2;
                    Seg s1{par.a, mid, par.E1, par.ioff1};
                    Seg s2{mid, par.b, par.E2, par.ioff2};
                    s1.nev = par.nev1;
                    s2.nev = par.nev2;
                    for (int c = 0; c < ncomp; ++c)
                        q.I[(size_t)c] = (q.I[(size_t)c] - q.store[(size_t)(par.ioff + c)]) +
                                         q.store[(size_t)(s1.ioff + c)] + q.store[(size_t)(s2.ioff + c)];
                    q.E = (q.E - par.E) + s1.E + s2.E;
                    q.fevals += s1.nev + s2.nev;
                    heap_push(q.heap, s1);
                    heap_push(q.heap, s2);
                }
            } else {
                finished = !(q.E > tol && q.numevals < maxevals);
            }
            if (!finished) {
                if (max_batch <= 0) {
                    // the top panel is needed now; with a fixed tolerance so is every panel whose own error plus the
                    // errors of all smaller panels exceeds it (they are certain to be popped: see the header comment)
                    const size_t hn = q.heap.size();
                    const double need = tol * (1.0 + 1e-9);
                    // (cheap exit first: near convergence the panels below the top no longer add up to the tolerance)
                    if (spec_round && rt == 0.0 && unlimited && hn > 1 && q.E - q.heap[0].E > need) {
                        order.resize(hn);
                        for (size_t h = 0; h < hn; ++h) order[h] = {q.heap[h].E, (uint32_t)h};
                        std::sort(order.begin(), order.end(), [](const std::pair<double, uint32_t>& u, const std::pair<double, uint32_t>& v) {
                            return u.first > v.first || (u.first == v.first && u.second < v.second);
                        });
                        suffix.resize(hn);
                        double acc = 0.0;
                        for (size_t h = hn; h-- > 0;) {  // small errors first: S_k = sum_{j >= k} E_j
                            acc += order[h].first;
                            suffix[h] = acc;
                        }
                        for (size_t h = 0; h < hn && suffix[h] > need; ++h)
                            if (q.heap[order[h].second].state == 0) q.req.push_back(order[h].second);
                    }
                    if (q.heap[0].state == 0 && std::find(q.req.begin(), q.req.end(), 0u) == q.req.end()) q.req.push_back(0u);
                    for (uint32_t h : q.req) {
                        Seg& par = q.heap[h];
                        par.state = 1;
                        const double mid = (par.a + par.b) / 2;
                        q.pend.push_back(Seg{par.a, mid, 0.0, 0});
                        q.pend.push_back(Seg{mid, par.b, 0.0, 0});
                    }
                } else {
                    // BatchIntegrand refine: pop panels while the error of the REMAINING ones still
                    // exceeds the tolerance (auxquadgk batch mode, src/algorithms.jl:227-233)
                    while (!q.heap.empty() && 30 * ((int64_t)q.popped.size() + 1) <= max_batch && q.E > tol &&
                           q.numevals < maxevals) {
                        Seg sg = heap_pop(q.heap);
                        tol += sg.E;
                        q.numevals += 30;
                        q.popped.push_back(sg);
                    }
                    for (const Seg& par : q.popped) {
                        const double mid = (par.a + par.b) / 2;
                        q.pend.push_back(Seg{par.a, mid, 0.0, 0});
                        q.pend.push_back(Seg{mid, par.b, 0.0, 0});
                    }
                }
            } else {
                // re-sum over the heap in storage order (QuadGK does this after adapt)
                for (int c = 0; c < ncomp; ++c) q.I[(size_t)c] = q.store[(size_t)(q.heap[0].ioff + c)];
                q.E = q.heap[0].E;
                for (size_t h = 1; h < q.heap.size(); ++h) {
                    for (int c = 0; c < ncomp; ++c) q.I[(size_t)c] += q.store[(size_t)(q.heap[h].ioff + c)];
                    q.E += q.heap[h].E;
                }
                q.done = true;
            }
        };
        par((int64_t)active.size(), true, [&](int64_t b, int64_t e, int) {
            std::vector<cd> Iseg((size_t)ncomp);
            std::vector<Seg> got;
            std::vector<std::pair<double, uint32_t>> order;
            std::vector<double> suffix;
            for (int64_t ai = b; ai < e; ++ai) deliver(active[(size_t)ai], act_off[(size_t)ai], Iseg, got, order, suffix);
        });
        if (bad.load()) {
            set_error("IAI: integrand produced a non-finite value in (%g, %g) at level %d", bad_a, bad_b, L);
            return ABZ_ERR_ARG;
        }
        for (size_t qi : active)
            if (!quads[qi].done) next.push_back(qi);
        if (stats) st_deliver += std::chrono::duration<double>(std::chrono::steady_clock::now() - tv0).count();
        active.swap(next);
    }
    return ABZ_OK;
}

}  // namespace abz

using namespace abz;

extern "C" {

int abz_gk15_nodes(double a, double b, double* x15) try {
    ABZ_REQUIRE(x15, "null output");
    gk15_nodes(a, b, x15);
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_gk15_batch(const double* ab, const double* values_reim, int64_t npanels, int ncomp, double* I_reim, double* E) try {
    ABZ_REQUIRE(ab && values_reim && I_reim && E && npanels >= 0 && ncomp >= 1, "abz_gk15_batch: bad arguments");
    const cd* fv = reinterpret_cast<const cd*>(values_reim);
    cd* I = reinterpret_cast<cd*>(I_reim);
    for (int64_t p = 0; p < npanels; ++p)
        gk15_evalrule(fv + (size_t)p * 15 * ncomp, ncomp, ab[2 * p], ab[2 * p + 1], I + (size_t)p * ncomp, E + p);
    return ABZ_OK;
} ABZ_CATCH_ALL

static int iai_solve_lane(abz_series* s, int lims_kind, const double* lim_a, const double* lim_b, int integrand,
                          const double* params, int nparams, const double* sweeps, int n_sweep, double abstol, double reltol,
                          int64_t maxevals, int64_t max_batch, double* out_reim, double* err, int64_t* numevals,
                          double* panels, int64_t max_panels, int64_t* npanels) {
    ABZ_REQUIRE(s && s->ctx && !s->closed && !s->ctx->closed && lim_a && out_reim, "abz_iai_solve: null argument");
    ABZ_REQUIRE(n_sweep >= 1 && sweeps, "abz_iai_solve: at least one sweep value");
    ABZ_REQUIRE(lims_kind >= ABZ_LIMS_CUBIC && lims_kind <= ABZ_LIMS_POLYGON, "unknown limits kind %d", lims_kind);
    ABZ_REQUIRE(lims_kind == ABZ_LIMS_TETRAHEDRAL || lim_b, "these limits need lim_b");
    ABZ_REQUIRE(lims_kind != ABZ_LIMS_POLYHEDRAL || s->d == 3, "PolyhedralLimits are for 3 variables");
    ABZ_REQUIRE(lims_kind != ABZ_LIMS_POLYGON || s->d == 2, "PolygonLimits are for 2 variables");
    std::deque<std::vector<double>> arena;  // the polytope and every polygon sliced from it during this solve
    const std::vector<double>* poly = nullptr;
    if (lims_kind >= ABZ_LIMS_POLYHEDRAL) {
        const int64_t len = (int64_t)lim_b[0];
        ABZ_REQUIRE(len >= 6, "polytope description too short");
        arena.emplace_back(lim_a, lim_a + len);
        poly = &arena.back();
        if (lims_kind == ABZ_LIMS_POLYHEDRAL) {  // [nv, xyz * nv] per face
            int64_t i = 0;
            while (i < len) {
                const int64_t nv = (int64_t)lim_a[i];
                ABZ_REQUIRE(nv >= 3 && i + 1 + 3 * nv <= len, "malformed face list");
                i += 1 + 3 * nv;
            }
        } else {
            ABZ_REQUIRE(len % 2 == 0, "polygon vertices come as (x, y) pairs");
        }
    }
    ABZ_REQUIRE(nparams >= 0 && nparams <= 4, "nparams = %d not in 0..4", nparams);
    ABZ_HIP(hipSetDevice(s->ctx->device));
    IaiDriver drv;
    drv.s = s;
    drv.ctx = s->ctx;
    drv.d = s->d;
    drv.n = s->n;
    drv.integrand = integrand;
    drv.ncomp = integrand_ncomp(integrand, s->n, s->d);
    ABZ_REQUIRE(drv.ncomp > 0, "unknown integrand id %d", integrand);
    for (int i = 0; i < 4; ++i) drv.params[i] = (i < nparams && params) ? params[i] : 0.0;
    drv.sweep = sweeps[0];
    drv.has_rtol = reltol >= 0;
    drv.rtol_user = reltol;
    drv.maxevals = maxevals > 0 ? maxevals : (int64_t)1 << 62;
    drv.max_batch = max_batch;
    {
        drv.speculate = abz_switch(SW_IAI_SPECULATE) != 0;  // 0: one panel per integral per round (the round-1 driver)
        drv.ex_fn = s->ex_fn;
        drv.ex_user = s->ex_user;
        drv.ex_rank = s->ex_rank;
        drv.ex_world = s->ex_world;
        const int hw = (int)std::thread::hardware_concurrency();
        int nth = abz_switch(SW_HOST_THREADS);
        if (hw > 0) nth = std::min(nth, std::max(1, hw / 2));
        if (nth > 1 && n_sweep >= 8) drv.pool.reset(new HostPool(nth));  // single solves stay serial: their rounds are small
        drv.stats = abz_switch(SW_IAI_STATS) == 1;
        const int pool_mb = abz_switch(SW_IAI_POOL_MB);
        if (pool_mb > 0) drv.pool_cap_bytes = (int64_t)pool_mb << 20;
    }
    {
        // n > 4: the workgroup-per-integral kernel (coefficient set in LDS); ABZ_IAI_DEVICE_INNER=0 forces the host loop
        // at every level
        const bool ok = s->n > 4 ? gen_inner_panel_supported(s->n, s->dims[0], integrand, s->hermitian)
                                 : inner_adaptive_supported(s->n, s->dims[0], integrand);
        drv.device_inner = max_batch <= 0 && s->d >= 2 && ok && abz_switch(SW_IAI_DEVICE_INNER) != 0;
    }
    {
        // ABZ_IAI_PACKED=0: the full coefficient rows (per call: tests compare both)
        drv.pk = abz_switch(SW_IAI_PACKED) != 0 && s->n <= 4 && s->hermitian && (s->dims[0] & 1) && s->first[0] == -(s->dims[0] - 1) / 2;
        if (drv.pk) {
            int rcp = series_ensure_packed(s);
            if (rcp) return rcp;
        }
    }
    // one top-level integral per sweep value; they advance in lock-step like any other siblings
    std::vector<Quad1D> top((size_t)n_sweep);
    for (int r = 0; r < n_sweep; ++r) {
        Quad1D& q = top[(size_t)r];
        q.slot = 0;
        q.sweep = sweeps[r];
        q.root = r;
        q.lims.kind = lims_kind;
        q.lims.s = 1.0;
        q.lims.poly = poly;
        q.lims.arena = &arena;
        for (int j = 0; j < s->d; ++j) {
            q.lims.a[j] = lims_kind < ABZ_LIMS_POLYHEDRAL ? lim_a[j] : 0.0;
            q.lims.b[j] = (lims_kind < ABZ_LIMS_POLYHEDRAL && lim_b) ? lim_b[j] : 0.0;
        }
        q.has_atol = abstol >= 0;
        q.atol = abstol >= 0 ? abstol : 0.0;
    }
    const auto tt0 = std::chrono::steady_clock::now();
    int rc = drv.solve_level(s->d, top);
    if (rc) return rc;
    if (drv.stats) {
        fprintf(stderr, "[abz iai] host seconds: total %.3f | gather %.3f | describe+enqueue %.3f | deliver %.3f | kid setup %.3f | GPU wait %.3f\n",
                std::chrono::duration<double>(std::chrono::steady_clock::now() - tt0).count(), drv.st_gather, drv.st_describe, drv.st_deliver,
                drv.st_kids, drv.st_wait);
        if (drv.ex_world > 1) fprintf(stderr, "[abz iai] sharded over %d ranks (this is rank %d): exchange %.3f s\n", drv.ex_world, drv.ex_rank, drv.st_exchange);
        fprintf(stderr, "[abz iai] rounds per level:");
        for (int L = 1; L <= s->d; ++L) fprintf(stderr, " L%d=%lld", L, (long long)drv.st_rounds[L]);
        fprintf(stderr, "\n[abz iai] host waited %.3f s for the GPU; innermost launches by size: 2^b integrals | launches | integrals\n",
                drv.st_wait);
        for (int b = 0; b < 40; ++b)
            if (drv.st_cnt[b])
                fprintf(stderr, "[abz iai]   2^%-2d %9lld %12lld\n", b, (long long)drv.st_cnt[b], (long long)drv.st_int[b]);
    }
    for (int r = 0; r < n_sweep; ++r) {
        for (int c = 0; c < drv.ncomp; ++c) {
            out_reim[2 * ((size_t)r * drv.ncomp + c)] = top[(size_t)r].I[(size_t)c].real();
            out_reim[2 * ((size_t)r * drv.ncomp + c) + 1] = top[(size_t)r].I[(size_t)c].imag();
        }
        if (err) err[r] = top[(size_t)r].E;
        if (numevals) numevals[r] = top[(size_t)r].fevals;
    }
    if (npanels) *npanels = (int64_t)top[0].heap.size();
    if (panels) {  // panels of the first solve
        std::vector<std::pair<double, double>> pn;
        for (auto& sg : top[0].heap) pn.emplace_back(sg.a, sg.b);
        std::sort(pn.begin(), pn.end());
        for (int64_t i = 0; i < (int64_t)pn.size() && i < max_panels; ++i) {
            panels[2 * i] = pn[(size_t)i].first;
            panels[2 * i + 1] = pn[(size_t)i].second;
        }
    }
    return ABZ_OK;
}

// A sweep of independent solves is dealt round-robin to LANES: host threads that each drive a view of the series on a
// stream of its own, so one lane's host rounds (gather, heaps, delivery) run beside the other lanes' kernels -- a round of
// a solve cannot overlap its own kernels, the next panels depend on them.  Every solve makes the decisions it would make
// alone, so the split changes no result (tests/test_gpu_parity.py::test_iai_lanes_*).
int abz_iai_solve_many(abz_series* s, int lims_kind, const double* lim_a, const double* lim_b, int integrand,
                       const double* params, int nparams, const double* sweeps, int n_sweep, double abstol, double reltol,
                       int64_t maxevals, int64_t max_batch, double* out_reim, double* err, int64_t* numevals,
                       double* panels, int64_t max_panels, int64_t* npanels) try {
    int lanes = 1, ncomp = 0;
    if (s && s->ctx && !s->closed && !s->ctx->closed && sweeps && out_reim && lim_a && !panels && !s->ex_fn && !s->coef_borrowed &&
        (ncomp = integrand_ncomp(integrand, s->n, s->d)) > 0)
        lanes = std::max(1, std::min(abz_switch(SW_IAI_LANES), n_sweep / std::max(1, abz_switch(SW_IAI_LANE_MIN))));
    if (lanes > 1 && series_lane_views(s, lanes - 1) != ABZ_OK) lanes = 1;
    if (lanes == 1)
        return iai_solve_lane(s, lims_kind, lim_a, lim_b, integrand, params, nparams, sweeps, n_sweep, abstol, reltol, maxevals,
                              max_batch, out_reim, err, numevals, panels, max_panels, npanels);
    struct LaneJob {
        std::vector<double> sw, out, err;
        std::vector<int64_t> nev;
        int rc = ABZ_OK;
        std::string msg;
    };
    std::vector<LaneJob> jobs((size_t)lanes);
    for (int r = 0; r < n_sweep; ++r) jobs[(size_t)(r % lanes)].sw.push_back(sweeps[r]);
    auto run = [&](int j) {
        LaneJob& q = jobs[(size_t)j];
        const int m = (int)q.sw.size();
        try {
            q.out.resize((size_t)m * ncomp * 2);
            q.err.resize((size_t)m);
            q.nev.resize((size_t)m);
            q.rc = iai_solve_lane(j == 0 ? s : s->lanes[(size_t)j - 1], lims_kind, lim_a, lim_b, integrand, params, nparams, q.sw.data(), m,
                                  abstol, reltol, maxevals, max_batch, q.out.data(), q.err.data(), q.nev.data(), nullptr, 0,
                                  j == 0 ? npanels : nullptr);
        } catch (...) {  // a lane is a std::thread: an exception leaving it would be std::terminate
            q.rc = catch_status();
        }
        if (q.rc) {
            try {
                q.msg = abz_last_error();  // (the message is per thread)
            } catch (...) {
            }
        }
    };
    std::vector<std::thread> th;
    th.reserve((size_t)lanes);
    std::vector<int> inline_lanes{0};
    for (int j = 1; j < lanes; ++j) {
        try {
            th.emplace_back(run, j);
        } catch (...) {  // no thread to be had: that lane runs on the caller's thread after lane 0
            inline_lanes.push_back(j);
        }
    }
    for (int j : inline_lanes) run(j);
    for (auto& t : th) t.join();
    for (auto& q : jobs)
        if (q.rc) {
            set_error("%s", q.msg.c_str());
            return q.rc;
        }
    for (int r = 0; r < n_sweep; ++r) {
        const LaneJob& q = jobs[(size_t)(r % lanes)];
        const size_t i = (size_t)(r / lanes);
        std::copy(q.out.begin() + (ptrdiff_t)(i * ncomp * 2), q.out.begin() + (ptrdiff_t)((i + 1) * ncomp * 2), out_reim + (size_t)r * ncomp * 2);
        if (err) err[r] = q.err[i];
        if (numevals) numevals[r] = q.nev[i];
    }
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_iai_solve(abz_series* s, int lims_kind, const double* lim_a, const double* lim_b, int integrand,
                  const double* params, int nparams, double sweep, double abstol, double reltol, int64_t maxevals,
                  int64_t max_batch, double* out_reim, double* err, int64_t* numevals, double* panels,
                  int64_t max_panels, int64_t* npanels) try {
    return abz_iai_solve_many(s, lims_kind, lim_a, lim_b, integrand, params, nparams, &sweep, 1, abstol, reltol, maxevals,
                              max_batch, out_reim, err, numevals, panels, max_panels, npanels);
} ABZ_CATCH_ALL

// ---- building blocks for a host-language (Julia) adaptive loop -------------------------------
int abz_contract_nodes(abz_series* s, int src_level, const int64_t* parents, const double* x, int64_t nnodes,
                       int64_t* slots_out) try {
    ABZ_REQUIRE(s && s->ctx && !s->closed && !s->ctx->closed && parents && x && slots_out, "abz_contract_nodes: null argument");
    ABZ_REQUIRE(src_level >= 2 && src_level <= s->d, "src_level = %d must be in 2..d", src_level);
    ABZ_HIP(hipSetDevice(s->ctx->device));
    IaiDriver drv;
    drv.s = s;
    drv.ctx = s->ctx;
    drv.d = s->d;
    drv.n = s->n;
    drv.h_parents.assign(parents, parents + nnodes);
    drv.h_x.assign(x, x + nnodes);
    const int64_t base = s->iai_used[src_level - 1];
    // growing the pool must keep earlier slots: reserve with copy
    const int64_t Lrow = s->elems(src_level - 1);
    DevBuf& pool = s->iai_pool[src_level - 1];
    const size_t need = sizeof(double2) * (size_t)((base + nnodes) * Lrow);
    if (need > pool.cap && base > 0) {
        DevBuf bigger;
        int rc = bigger.reserve(need * 2);
        if (rc) return rc;
        // on the context's (non-blocking) stream, and complete before the old pool is freed
        ABZ_HIP(hipMemcpyAsync(bigger.p, pool.p, sizeof(double2) * (size_t)(base * Lrow), hipMemcpyDeviceToDevice,
                               s->ctx->stream));
        ABZ_HIP(hipStreamSynchronize(s->ctx->stream));
        pool.release();
        pool = bigger;
    }
    int rc = drv.contract_nodes(src_level, nnodes, base);
    if (rc) return rc;
    ABZ_HIP(hipStreamSynchronize(s->ctx->stream));
    for (int64_t i = 0; i < nnodes; ++i) slots_out[i] = base + i;
    s->iai_used[src_level - 1] = base + nnodes;
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_eval_line_nodes(abz_series* s, const int64_t* parents, const double* x, const double* tail, int64_t nnodes,
                        int integrand, const double* params, int nparams, double sweep, double* values_reim) try {
    ABZ_REQUIRE(s && s->ctx && !s->closed && !s->ctx->closed && parents && x && values_reim, "abz_eval_line_nodes: null argument");
    ABZ_REQUIRE(nparams >= 0 && nparams <= 4, "nparams = %d not in 0..4", nparams);
    ABZ_HIP(hipSetDevice(s->ctx->device));
    IaiDriver drv;
    drv.s = s;
    drv.ctx = s->ctx;
    drv.d = s->d;
    drv.n = s->n;
    drv.integrand = integrand;
    drv.ncomp = integrand_ncomp(integrand, s->n, s->d);
    ABZ_REQUIRE(drv.ncomp > 0, "unknown integrand id %d", integrand);
    ABZ_REQUIRE(integrand != ABZ_F_LINEAR_X || s->d == 1 || tail, "ABZ_F_LINEAR_X needs the outer coordinates (tail)");
    for (int i = 0; i < 4; ++i) drv.params[i] = (i < nparams && params) ? params[i] : 0.0;
    drv.sweep = sweep;
    drv.h_parents.assign(parents, parents + nnodes);
    drv.h_x.assign(x, x + nnodes);
    drv.h_sweep.assign((size_t)nnodes, sweep);
    drv.panels15 = false;  // arbitrary node list
    if (tail && s->d > 1) drv.h_tail.assign(tail, tail + nnodes * (s->d - 1));
    int rc = drv.eval_nodes(nnodes);
    if (rc) return rc;
    std::memcpy(values_reim, drv.h_values.data(), sizeof(double2) * (size_t)(nnodes * drv.ncomp));
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_iai_set_exchange(abz_series* s, abz_exchange_fn fn, void* user, int rank, int world) try {
    ABZ_REQUIRE(s && s->ctx && !s->closed, "abz_iai_set_exchange: null or destroyed series");
    ABZ_REQUIRE(fn == nullptr || (world >= 1 && rank >= 0 && rank < world), "rank %d of %d", rank, world);
    s->ex_fn = fn;  // world == 1 keeps the hook: a one-rank rehearsal of the exchange (the collective is then the identity)
    s->ex_user = user;
    s->ex_rank = s->ex_fn ? rank : 0;
    s->ex_world = s->ex_fn ? world : 1;
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_release_level(abz_series* s, int level) try {
    ABZ_REQUIRE(s && level >= 1 && level <= s->d, "abz_release_level: bad level");
    for (int L = 1; L < level && L <= ABZ_MAX_DIM; ++L) s->iai_used[L] = 0;
    return ABZ_OK;
} ABZ_CATCH_ALL

}  // extern "C"
