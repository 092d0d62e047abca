// Host side of the C ABI (include/abzhip.h): handles, rule construction plans, exports.
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <thread>

#include "abz_internal.h"

namespace abz {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

// The status of the exception in flight (called from the catch-all of every entry point: no C++ exception crosses the C ABI).
int catch_status() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        try {
            set_error("out of host memory (std::bad_alloc)");
        } catch (...) {
        }
        return ABZ_ERR_NOMEM;
    } catch (const std::exception& e) {
        try {
            set_error("internal error: %s", e.what());
        } catch (...) {
        }
        return ABZ_ERR_INTERNAL;
    } catch (...) {
        try {
            set_error("internal error: unknown C++ exception");
        } catch (...) {
        }
        return ABZ_ERR_INTERNAL;
    }
}

// ------------------------------------------------------------------------------------------
// Environment switches: the one place the library reads the environment (abz_internal.h: enum Switch).
// ------------------------------------------------------------------------------------------
namespace {
struct SwitchDef {
    const char* name;
    int dflt;
    const char* what;
};
const SwitchDef g_switches[SW_COUNT] = {
    {"ABZ_POOL_MB", 4096, "MB of freed device blocks the caching allocator keeps per device"},
    {"ABZ_DEBUG_TIMING", 0, "1: wall time of the rule-build phases on stderr (synchronises after each)"},
    {"ABZ_EVAL_PACKED", 1, "0: full-grid chains on plain coefficient sets instead of packed Hermitian sets"},
    {"ABZ_DOS3_SCAN", 1, "0: 3-band DOS sweeps through the generic reduce_kernel"},
    {"ABZ_REDUCE_ROWS", 0, "> 0: rows of blocks a sweep is split over in the scan kernels"},
    {"ABZ_ADAPT_PAIR", 1, "0: one-lane adaptive step in the device-side GK loops"},
    {"ABZ_GEN_SUM_TRI", 1, "0: 5...16-band sweeps with one inversion per swept value"},
    {"ABZ_IPANEL_FOLD", 1, "0: 16-lane panel kernel on the unfolded series"},
    {"ABZ_IPANEL_FMAC", 1, "0: 16-lane panel kernel with separate pivot-row broadcasts"},
    {"ABZ_GGR_FUSED", 1, "0: GGR build by eigenvectors + one velocity launch per variable"},
    {"ABZ_GGR_FUSE2", 1, "0: fused GGR build reads level-1 families instead of contracting variable 2 itself"},
    {"ABZ_GGR_UNIFORM", 1, "0: GGR scans search the energy window even in equispaced lists"},
    {"ABZ_IAI_SPECULATE", 1, "0: nested IAI driver requests one panel per integral and round"},
    {"ABZ_IAI_PACKED", 1, "0: IAI chains on plain coefficient sets"},
    {"ABZ_IAI_POOL_MB", 0, "> 0: MB per chunk of level sets in the IAI driver (default: sized from free memory)"},
    {"ABZ_IAI_DEVICE_INNER", 1, "0: innermost GK loops driven from the host, one launch per round"},
    {"ABZ_IAI_PANELS", 1, "0: the level above the innermost one ships nodes instead of panels (GK rule of that level on the host)"},
    {"ABZ_IAI_STATS", 0, "1: per-solve statistics of the IAI driver on stderr"},
    {"ABZ_HOST_THREADS", 8, "host threads for the per-integral bookkeeping of IAI sweeps (capped at half the cores)"},
    {"ABZ_AUTO_SWEEP_MAPPED", 1, "0: abz_autoptr_solve* uploads its swept values instead of letting the kernels read the pinned host copy"},
    {"ABZ_LANE_KERNELS", 1, "0: 5...8-band rule builds / store-free sums on full grids through the 8-lane row kernels (test: same values)"},
    {"ABZ_BIG_MFMA", 0, "1: 33...64 bands: the level-1 evaluation of full grid lines as a real GEMM on v_mfma_f64_16x16x4_f64 (test: same values)"},
    {"ABZ_BIG_CHUNK_MB", 0, "33...64 bands: MB of scratch for the matrices of a chunk of nodes (0: 256, GGR builds 2048; tests set 1 to walk many chunks)"},
    {"ABZ_BIG_TRI_WAVES", 0, "33...64 bands: waves per node of the Householder tridiagonalisation (1, 2, 4: the same reflectors, another summation order; 0: two up to 44 bands, one up to 48, four above)"},
    {"ABZ_EIG_FOLD", 1, "0: 5...16-band rule builds evaluate the full level-1 series of a Hermitian model instead of the folded one"},
    {"ABZ_EIG_SPLIT", 1, "0: eigenvalues of 5...16-band rules by bisection inside the grid kernel instead of the per-lane QR kernel"},
    {"ABZ_IAI_LANES", 4, "lanes (host thread + stream each) a sweep of independent IAI solves is split over; 1: off"},
    {"ABZ_IAI_LANE_MIN", 16, "solves per lane below which a sweep is not split further"},
};
}  // namespace

int abz_switch(Switch s) {
    const char* e = getenv(g_switches[s].name);
    return (e && *e) ? atoi(e) : g_switches[s].dflt;
}

// ------------------------------------------------------------------------------------------
// Caching device allocator.  hipMalloc / hipFree of MB-sized blocks cost milliseconds each (and hipFree
// synchronises the device); rule builds, symmetric-rule plans and the IAI pools allocate and free such
// blocks all the time (a cold symmetric 150^3 rule spent 20 of its 30 ms there).  Freed blocks are kept
// per device, up to ABZ_POOL_MB (default 4096) MB, and handed out again when a request fits within 2x.
// A block enters the cache only after the device is idle, so a new owner never races an old kernel.
// ------------------------------------------------------------------------------------------
namespace {
struct Block {
    void* p;
    size_t cap;
    int device;
};
std::mutex g_pool_mutex;
std::vector<Block> g_pool;
size_t g_pool_bytes = 0;
std::atomic<int64_t> g_live_bytes{0};   // handed out by dev_alloc and not yet returned (abz_mem_info)
std::atomic<int64_t> g_live_blocks{0};
size_t pool_limit() {
    static const size_t lim = (size_t)std::max(0, abz_switch(SW_POOL_MB)) << 20;
    return lim;
}
void pool_flush(int device) {  // give everything cached for `device` back to the driver
    for (size_t i = 0; i < g_pool.size();) {
        if (g_pool[i].device == device) {
            (void)hipFree(g_pool[i].p);
            g_pool_bytes -= g_pool[i].cap;
            g_pool[i] = g_pool.back();
            g_pool.pop_back();
        } else {
            ++i;
        }
    }
}
}  // namespace

int dev_alloc(void** out, size_t bytes, size_t* cap_out) {
    int device = 0;
    (void)hipGetDevice(&device);
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        size_t best = g_pool.size();
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].device == device && g_pool[i].cap >= bytes && g_pool[i].cap <= 2 * bytes + 4096 &&
                (best == g_pool.size() || g_pool[i].cap < g_pool[best].cap))
                best = i;
        if (best != g_pool.size()) {
            *out = g_pool[best].p;
            if (cap_out) *cap_out = g_pool[best].cap;
            g_pool_bytes -= g_pool[best].cap;
            g_live_bytes += (int64_t)g_pool[best].cap;
            g_live_blocks += 1;
            g_pool[best] = g_pool.back();
            g_pool.pop_back();
            return ABZ_OK;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {  // out of memory: drop the cache and try once more
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        pool_flush(device);
        (void)hipGetLastError();
        e = hipMalloc(out, bytes);
    }
    if (e != hipSuccess) {
        *out = nullptr;
        set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return ABZ_ERR_NOMEM;
    }
    if (cap_out) *cap_out = bytes;
    g_live_bytes += (int64_t)bytes;
    g_live_blocks += 1;
    return ABZ_OK;
}

void dev_free(void* p, size_t cap) {
    if (!p) return;
    g_live_bytes -= (int64_t)cap;
    g_live_blocks -= 1;
    int device = 0;
    (void)hipGetDevice(&device);  // = the owner's device: every destroy / release path selects it first (hipSetDevice)
    if (cap > 0 && cap <= pool_limit() / 4) {
        (void)hipDeviceSynchronize();  // nothing in flight may still touch the block when it is reused
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        if (g_pool_bytes + cap <= pool_limit()) {
            g_pool.push_back(Block{p, cap, device});
            g_pool_bytes += cap;
            return;
        }
    }
    (void)hipFree(p);
}

static int stage_reserve(abz_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->pin_cap) return ABZ_OK;
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    ctx->pin = nullptr;
    ctx->pin_cap = 0;
    const size_t want = std::max<size_t>(bytes, (size_t)8 << 20);
    hipError_t e = hipHostMalloc(&ctx->pin, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return ABZ_ERR_NOMEM;
    }
    ctx->pin_cap = want;
    return ABZ_OK;
}

int mbox_reserve(abz_ctx* ctx) {
    if (ctx->mbox) return ABZ_OK;
    const size_t cap = (size_t)128 << 10;
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return ABZ_ERR_NOMEM;
    }
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || !dp) {
        (void)hipGetLastError();
        (void)hipHostFree(p);
        return ABZ_ERR_NOMEM;
    }
    ctx->mbox = p;
    ctx->mbox_dev = dp;
    ctx->mbox_cap = cap;
    return ABZ_OK;
}

constexpr size_t STAGE_MIN = (size_t)256 << 10;   // smaller copies: the runtime's own staging is fine
constexpr size_t STAGE_MAX = (size_t)64 << 20;    // chunk size of big transfers

int stage_h2d(abz_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return ABZ_OK;
    if (bytes < STAGE_MIN) {
        ABZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        ABZ_HIP(hipStreamSynchronize(ctx->stream));
        return ABZ_OK;
    }
    int rc = stage_reserve(ctx, std::min(bytes, STAGE_MAX));
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += ctx->pin_cap) {
        const size_t n = std::min(ctx->pin_cap, bytes - off);
        std::memcpy(ctx->pin, (const char*)src + off, n);
        ABZ_HIP(hipMemcpyAsync((char*)dst + off, ctx->pin, n, hipMemcpyHostToDevice, ctx->stream));
        ABZ_HIP(hipStreamSynchronize(ctx->stream));
    }
    return ABZ_OK;
}

int stage_d2h(abz_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return ABZ_OK;
    if (bytes < STAGE_MIN) {
        ABZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        ABZ_HIP(hipStreamSynchronize(ctx->stream));
        return ABZ_OK;
    }
    int rc = stage_reserve(ctx, std::min(bytes, STAGE_MAX));
    if (rc) return rc;
    // two halves of the pinned buffer in turn: the DMA of a piece runs while the host copies the piece before it into the
    // caller's (pageable) array; that copy takes longer than the DMA and is dealt to up to four threads when a piece is large
    const size_t piece = std::max<size_t>((ctx->pin_cap / 2) & ~(size_t)4095, 4096);
    auto host_copy = [](char* d, const char* s_, size_t n) {
        const int nt = n >= ((size_t)8 << 20) ? 4 : 1;
        if (nt == 1) {
            std::memcpy(d, s_, n);
            return;
        }
        std::thread th[3];
        const size_t share = (n / nt + 63) & ~(size_t)63;
        for (int t = 1; t < nt; ++t) {
            const size_t o = std::min(n, share * t), m = std::min(n, share * (t + 1)) - o;
            th[t - 1] = std::thread([=] { std::memcpy(d + o, s_ + o, m); });
        }
        std::memcpy(d, s_, std::min(n, share));
        for (int t = 1; t < nt; ++t) th[t - 1].join();
    };
    size_t off = 0, prev_off = 0, prev_n = 0;
    int buf = 0;
    while (off < bytes || prev_n) {
        size_t n = 0;
        if (off < bytes) {
            n = std::min(piece, bytes - off);
            ABZ_HIP(hipMemcpyAsync((char*)ctx->pin + (size_t)buf * piece, (const char*)src + off, n, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (prev_n) host_copy((char*)dst + prev_off, (const char*)ctx->pin + (size_t)(buf ^ 1) * piece, prev_n);  // (its DMA was waited for below)
        if (n) ABZ_HIP(hipStreamSynchronize(ctx->stream));
        prev_off = off;
        prev_n = n;
        off += n;
        buf ^= 1;
    }
    return ABZ_OK;
}

int DevBuf::reserve(size_t bytes) {
    if (!view && bytes <= cap) return ABZ_OK;
    release();
    const size_t want = bytes + (bytes >> 2) + 256;  // grow geometrically
    int rc = dev_alloc(&p, want, &cap);
    if (rc) rc = dev_alloc(&p, bytes, &cap);
    if (rc) {
        p = nullptr;
        cap = 0;
    }
    return rc;
}

void DevBuf::release() {
    if (!view) dev_free(p, cap);
    p = nullptr;
    cap = 0;
    view = false;
}

ProfScope::ProfScope(abz_ctx* c, int kernel_id) : ctx(c), id(kernel_id) {
    if (!(ctx->prof & (1u << id))) return;
    auto get = [&]() {
        hipEvent_t e = nullptr;
        if (!ctx->event_pool.empty()) {
            e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else if (hipEventCreate(&e) != hipSuccess) {
            e = nullptr;
        }
        return e;
    };
    e0 = get();
    e1 = get();
    if (e0) (void)hipEventRecord(e0, ctx->stream);
}

ProfScope::~ProfScope() {
    if (!(ctx->prof & (1u << id)) || !e0 || !e1) return;
    (void)hipEventRecord(e1, ctx->stream);
    ctx->prof_slots[id].pending.emplace_back(e0, e1);
}

int prof_collect(abz_ctx* ctx) {
    for (int i = 0; i < ABZ_K_COUNT; ++i) {
        auto& sl = ctx->prof_slots[i];
        for (auto& pr : sl.pending) {
            ABZ_HIP(hipEventSynchronize(pr.second));
            float ms = 0.f;
            ABZ_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
            sl.ms += ms;
            sl.launches += 1;
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        sl.pending.clear();
    }
    return ABZ_OK;
}

template <class T>
static int upload(abz_ctx* ctx, DevBuf& buf, const T* host, size_t count) {
    int rc = buf.reserve(sizeof(T) * std::max<size_t>(count, 1));
    if (rc) return rc;
    if (count) return stage_h2d(ctx, buf.p, host, sizeof(T) * count);
    return ABZ_OK;
}

// ------------------------------------------------------------------------------------------
// Evaluation plan: which contracted coefficient sets are needed at every level.
//   level L (1 <= L < d) items: (grid index or coordinate of variable L+1, parent item at level L+1)
//   nodes: (grid index / coordinate of variable 1, parent item at level 1)
// Full grids are implicit (parent = item / npt, grid index = item % npt).
// ------------------------------------------------------------------------------------------
struct Plan {
    int d = 0;
    bool full = false;
    bool coords = false;  // explicit coordinates (x) instead of grid indices
    int npt = 0;
    int outer0 = 0, outer_n = 0;  // full grids: range of the outermost variable (a slab; whole grid = 0, npt)
    int64_t nk = 0;
    int64_t nitems[ABZ_MAX_DIM + 1] = {0, 0, 0, 0};  // per level 1..d-1
    std::vector<int32_t> gi[ABZ_MAX_DIM + 1];   // level L: grid index of variable L+1; [0]: nodes' i_1
    std::vector<double> xs[ABZ_MAX_DIM + 1];    // same with coordinates
    std::vector<int64_t> parent[ABZ_MAX_DIM + 1];  // level L item -> item at level L+1; [0]: node -> level-1 item
    std::vector<int64_t> run_start;  // irregular lists, d >= 2: first node of every level-1 item (+ nk at the end)
    int64_t nruns = 0;               // = number of level-1 items when the runs are known (host list above, or device tables)
};

static void plan_full(Plan& p, int d, int npt, int outer0, int outer_n) {
    p.d = d;
    p.full = true;
    p.npt = npt;
    p.outer0 = outer0;
    p.outer_n = outer_n;
    p.nk = outer_n;
    for (int j = 0; j + 1 < d; ++j) p.nk *= npt;
    for (int L = 1; L < d; ++L) {
        int64_t c = outer_n;
        for (int j = L; j + 1 < d; ++j) c *= npt;  // variables L+1..d-1, times the slab of variable d
        p.nitems[L] = c;
    }
}

// pts: either idx [nk][d] (int32 grid indices) or x [nk][d] (coords).  Consecutive nodes sharing the
// outer coordinates share the contracted sets (runs); no sorting is done here.
template <class T>
static void plan_runs(Plan& p, int d, int npt, const T* pts, int64_t nk, bool coords) {
    p.d = d;
    p.full = false;
    p.coords = coords;
    p.npt = npt;
    p.nk = nk;
    auto push = [&](int L, T v) {
        if (coords)
            p.xs[L].push_back((double)v);
        else
            p.gi[L].push_back((int32_t)v);
    };
    for (int L = 0; L < d; ++L) {
        p.gi[L].clear();
        p.xs[L].clear();
        p.parent[L].clear();
    }
    for (int64_t k = 0; k < nk; ++k) {
        const T* q = pts + k * d;
        // find the highest level whose tuple (q[L..d-1]) differs from the previous node's
        int newfrom = 0;  // levels >= newfrom... we create items for levels L where tuple changed
        if (k == 0) {
            newfrom = d - 1;
        } else {
            const T* r = pts + (k - 1) * d;
            newfrom = 0;
            for (int j = d - 1; j >= 1; --j) {
                if (q[j] != r[j]) {
                    newfrom = j;
                    break;
                }
            }
        }
        // create items top-down for levels L = newfrom .. 1 (level L item fixes variables L+1..d)
        for (int L = std::min(newfrom, d - 1); L >= 1; --L) {
            push(L, q[L]);
            const int64_t par = (L == d - 1) ? 0 : (int64_t)(coords ? p.xs[L + 1].size() : p.gi[L + 1].size()) - 1;
            p.parent[L].push_back(par);
        }
        push(0, q[0]);
        const int64_t par0 = (d == 1) ? 0 : (int64_t)(coords ? p.xs[1].size() : p.gi[1].size()) - 1;
        p.parent[0].push_back(par0);
    }
    for (int L = 1; L < d; ++L) p.nitems[L] = (int64_t)(coords ? p.xs[L].size() : p.gi[L].size());
    p.run_start.clear();
    if (d >= 2) {  // nodes of a level-1 item are consecutive: item i owns [run_start[i], run_start[i + 1])
        for (int64_t k = 0; k < nk; ++k)
            if (k == 0 || p.parent[0][(size_t)k] != p.parent[0][(size_t)k - 1]) p.run_start.push_back(k);
        p.run_start.push_back(nk);
    }
    p.nruns = p.run_start.empty() ? 0 : (int64_t)p.run_start.size() - 1;
}

struct PlanDev {
    DevBuf gi[ABZ_MAX_DIM + 1], xs[ABZ_MAX_DIM + 1], parent[ABZ_MAX_DIM + 1];
    DevBuf phg[ABZ_MAX_DIM + 1];  // full grids: phase table [npt][M_{L+1}] of the contraction at level L
    DevBuf runs;                  // irregular lists: run_start
    DevBuf arena;                 // symmetric rules: the copy of the cached tables that gi / parent / runs (and the rule's w, idx) point into
    void release() {
        runs.release();
        arena.release();
        for (int i = 0; i <= ABZ_MAX_DIM; ++i) {
            gi[i].release();
            xs[i].release();
            parent[i].release();
            phg[i].release();
        }
    }
};

static int plan_upload(abz_ctx* ctx, const Plan& p, PlanDev& pd) {
    if (p.full) return ABZ_OK;
    for (int L = 0; L < p.d; ++L) {
        int rc;
        if (p.coords)
            rc = upload(ctx, pd.xs[L], p.xs[L].data(), p.xs[L].size());
        else
            rc = upload(ctx, pd.gi[L], p.gi[L].data(), p.gi[L].size());
        if (rc) return rc;
        rc = upload(ctx, pd.parent[L], p.parent[L].data(), p.parent[L].size());
        if (rc) return rc;
    }
    if (!p.run_start.empty()) {
        int rc = upload(ctx, pd.runs, p.run_start.data(), p.run_start.size());
        if (rc) return rc;
    }
    return ABZ_OK;
}

// Build the level-1 coefficient sets for a plan.  deriv_dim (1-based, 0 = none) applies the
// derivative factor to that variable's phases.  Returns pointer to level-1 sets (or coef if d == 1).
// packed: the chain runs on the coefficients with the innermost variable packed (Hermitian series, packed_herm.h): rows of
// P = n (n + 1) / 2 + F n^2 numbers instead of M n^2 -- packing is linear and commutes with every contraction.
// the series' coefficients with the innermost variable packed (packed_herm.h), rebuilt after an update
int series_ensure_packed(abz_series* s) {
    if (s->coef_pk_valid) return ABZ_OK;
    const int64_t row_full = (int64_t)s->dims[0] * s->n * s->n;
    const int64_t row_len = (int64_t)packed_row_elems(s->n, s->dims[0]);
    const int64_t nrows = s->elems(s->d) / row_full;
    int rc = s->coef_pk.reserve(sizeof(double2) * (size_t)(nrows * row_len));
    if (rc) return rc;
    if ((rc = launch_pack_rows(s->ctx, s->n, s->dims[0], s->coef, nrows, s->coef_pk.as<double2>()))) return rc;
    s->coef_pk_valid = true;
    return ABZ_OK;
}

static int build_chain(abz_series* s, const Plan& p, const PlanDev& pd, const double2* tab, int deriv_dim,
                       const double2** level1, int last_level = 1, DevBuf* last_out = nullptr, bool packed = false) {
    abz_ctx* ctx = s->ctx;
    const int d = s->d;
    const int64_t row_full = (int64_t)s->dims[0] * s->n * s->n;
    const int64_t row_len = packed ? (int64_t)packed_row_elems(s->n, s->dims[0]) : row_full;
    auto elems_of = [&](int level) { return s->elems(level) / row_full * row_len; };  // numbers per level-`level` set
    if (packed) {
        int rc = series_ensure_packed(s);
        if (rc) return rc;
    }
    const double2* src = packed ? s->coef_pk.as<double2>() : s->coef;
    int64_t src_elems = elems_of(d);
    for (int L = d - 1; L >= last_level; --L) {  // last_level = 2: stop at the level-2 sets (fused last contraction)
        // contract variable L+1 (0-based dim index L)
        const int64_t B = p.nitems[L];
        const int M = s->dims[L];
        const int64_t Lrow = elems_of(L);
        // the sets of the last level go to `last_out` when given (several families alive at once: fused GGR build)
        DevBuf& ob = (L == last_level && last_out) ? *last_out : s->pool[L];
        int rc = ob.reserve(sizeof(double2) * (size_t)std::max<int64_t>(B * Lrow, 1));
        if (rc) return rc;
        double2* out = ob.as<double2>();
        // full grids: items of level L are (gi, parent) with gi fastest; gi runs over the slab for the
        // outermost variable and over the whole grid below it
        const int gbeg = (p.full && L == d - 1) ? p.outer0 : 0;
        const int gcnt = (p.full && L == d - 1) ? p.outer_n : p.npt;
        if (p.full && M <= ABZ_CONTRACT_GRID_MAXM && gcnt > 0 && B / gcnt <= 65535) {
            rc = launch_contract_grid(ctx, src, src_elems, B / gcnt, tab, out, Lrow, M, s->first[L], p.npt,
                                      deriv_dim == L + 1, gbeg, gcnt, pd.phg[L].p ? pd.phg[L].as<double2>() : nullptr);
            if (rc) return rc;
        } else {
            rc = ctx->scratch[0].reserve(sizeof(double2) * (size_t)std::max<int64_t>(B * M, 1));
            if (rc) return rc;
            double2* phs = ctx->scratch[0].as<double2>();
            PhaseSpec ps;
            ps.B = B;
            ps.M = M;
            ps.first = s->first[L];
            ps.gi = (p.full || p.coords) ? nullptr : pd.gi[L].as<int32_t>();
            ps.x = p.coords ? pd.xs[L].as<double>() : nullptr;
            ps.tab = tab;
            ps.npt = p.npt;
            ps.g0 = gbeg;
            ps.gcnt = gcnt;
            ps.period = s->period[L];
            ps.deriv = (deriv_dim == L + 1);
            rc = launch_phases(ctx, ps, phs);
            if (rc) return rc;
            const int64_t* parents = p.full ? nullptr : pd.parent[L].as<int64_t>();
            rc = launch_contract(ctx, src, src_elems, parents, p.full ? gcnt : 1, phs, out, B, Lrow, M);
            if (rc) return rc;
        }
        src = out;
        src_elems = Lrow;
    }
    *level1 = src;
    return ABZ_OK;
}

static int check_series(const abz_series* s) {
    if (!s || !s->ctx) {
        set_error("null series handle");
        return ABZ_ERR_ARG;
    }
    if (s->closed || s->ctx->closed) {
        set_error("series handle (or its context) was destroyed");
        return ABZ_ERR_ARG;
    }
    return ABZ_OK;
}

static int check_rule(const abz_rule* r) {
    if (!r || !r->plan || !r->s) {
        set_error("null rule handle");
        return ABZ_ERR_ARG;
    }
    if (r->s->closed || r->s->ctx->closed) {
        set_error("the rule's series (or context) was destroyed");
        return ABZ_ERR_ARG;
    }
    return ABZ_OK;
}

static void ctx_release(abz_ctx* ctx) {
    if (--ctx->refs > 0) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& b : ctx->scratch) b.release();
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->mbox) (void)hipHostFree(ctx->mbox);
    for (SymTables* c : ctx->sym_cache) {
        c->release();
        delete c;
    }
    ctx->sym_cache.clear();
    for (auto& e : ctx->phase_cache) e.second.release();
    ctx->phase_cache.clear();
    for (auto& sl : ctx->prof_slots)
        for (auto& pr : sl.pending) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static void series_release(abz_series* s) {
    if (--s->refs > 0) return;
    abz_ctx* ctx = s->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& b : s->pool) b.release();
    for (auto& b : s->iai_pool) b.release();
    for (auto& b : s->iai_io) b.release();
    for (auto& q : s->iai_pin)
        if (q) (void)hipHostFree(q);
    for (abz_series* v : s->lanes) {  // the views go with their series, and each one's context with it
        abz_ctx* vc = v->ctx;
        v->closed = true;
        series_release(v);
        (void)abz_ctx_destroy(vc);
        (void)hipSetDevice(ctx->device);
    }
    if (!s->coef_borrowed) dev_free(s->coef, s->coef_cap);
    s->coef_pk.release();
    s->auto_io.release();
    if (s->auto_pin) (void)hipHostFree(s->auto_pin);
    delete s;
    ctx_release(ctx);
}

}  // namespace abz

using namespace abz;

extern "C" {

const char* abz_last_error(void) { return g_err.c_str(); }
int abz_version(void) { return ABZ_VERSION; }

int abz_device_count(int* n) try {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (n) *n = (e == hipSuccess) ? c : 0;
    if (e != hipSuccess || c == 0) {
        set_error("no HIP device visible (%s): the product path has no CPU fallback",
                  e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return ABZ_ERR_NOGPU;
    }
    return ABZ_OK;
} ABZ_CATCH_ALL

static int ctx_create(int device, hipStream_t borrowed, bool borrow, abz_ctx** out);
int abz_ctx_create(int device, abz_ctx** out) try { return ctx_create(device, nullptr, false, out); } ABZ_CATCH_ALL
int abz_ctx_create_on_stream(int device, void* hip_stream, abz_ctx** out) try {
    return ctx_create(device, static_cast<hipStream_t>(hip_stream), true, out);
} ABZ_CATCH_ALL

static int ctx_create(int device, hipStream_t borrowed, bool borrow, abz_ctx** out) {
    ABZ_REQUIRE(out != nullptr, "abz_ctx_create: null out");
    *out = nullptr;
    int n = 0;
    int rc = abz_device_count(&n);
    if (rc) return rc;
    ABZ_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)", device, n);
    ABZ_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ABZ_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
        return ABZ_ERR_NOGPU;
    }
    abz_ctx* ctx = new abz_ctx();
    ctx->device = device;
    if (borrow) {
        ctx->stream = borrowed;
        ctx->owns_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
            return ABZ_ERR_HIP;
        }
    }
    // HIP loads a code object at the first launch of one of its kernels; the small one of the symmetric-rule tables
    // would otherwise load inside the first symmetric solve (0.6 ms of a 2.1 ms first AutoPTR solve on the cubic IBZ)
    preload_symptr_code();
    *out = ctx;
    return ABZ_OK;
}

int abz_ctx_destroy(abz_ctx* ctx) try {
    if (!ctx || ctx->closed) return ABZ_OK;
    ctx->closed = true;
    ctx_release(ctx);  // freed once the last series created on it is gone
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_ctx_sync(abz_ctx* ctx) try {
    ABZ_REQUIRE(ctx && !ctx->closed, "null or destroyed ctx");
    ABZ_HIP(hipStreamSynchronize(ctx->stream));
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_prof_enable(abz_ctx* ctx, int on) try {
    ABZ_REQUIRE(ctx, "null ctx");
    // on = 1: every kernel id; on > 1: bit mask (bit k+1 selects kernel id k), e.g. 1 << (ABZ_K_EVAL + 1)
    ctx->prof = on == 0 ? 0u : (on == 1 ? 0xffffffffu : ((unsigned)on >> 1));
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_mem_info(abz_ctx* ctx, int64_t* info) try {
    ABZ_REQUIRE(info, "abz_mem_info: null info");
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        info[1] = (int64_t)g_pool_bytes;
    }
    info[0] = g_live_bytes.load();
    info[2] = 0;
    info[3] = 0;
    info[4] = g_live_blocks.load();
    if (ctx) {
        for (const auto& b : ctx->scratch) info[2] += (int64_t)b.cap;
        info[3] = (int64_t)ctx->pin_cap;
    }
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_prof_reset(abz_ctx* ctx) try {
    ABZ_REQUIRE(ctx, "null ctx");
    int rc = prof_collect(ctx);
    for (auto& sl : ctx->prof_slots) {
        sl.ms = 0;
        sl.launches = 0;
    }
    return rc;
} ABZ_CATCH_ALL

int abz_prof_read(abz_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches) try {
    ABZ_REQUIRE(ctx && kernel_id >= 0 && kernel_id < ABZ_K_COUNT, "bad profile slot");
    int rc = prof_collect(ctx);
    if (rc) return rc;
    if (total_ms) *total_ms = ctx->prof_slots[kernel_id].ms;
    if (launches) *launches = ctx->prof_slots[kernel_id].launches;
    return ABZ_OK;
} ABZ_CATCH_ALL

// ---------------------------------------------------------------- series
static bool detect_hermitian(const abz_series* s, const double* coef_reim);

int abz_series_create(abz_ctx* ctx, const double* coef_reim, int d, const int32_t* dims, const int32_t* first,
                      const double* period, int n, abz_series** out) try {
    ABZ_REQUIRE(ctx && coef_reim && dims && first && period && out, "abz_series_create: null argument");
    ABZ_REQUIRE(!ctx->closed, "abz_series_create: the context was destroyed");
    ABZ_REQUIRE(d >= 1 && d <= ABZ_MAX_DIM, "series dimension d = %d not in 1..%d", d, ABZ_MAX_DIM);
    ABZ_REQUIRE(n >= 1 && n <= ABZ_MAX_BANDS, "n = %d bands not in 1..%d", n, ABZ_MAX_BANDS);
    *out = nullptr;
    ABZ_HIP(hipSetDevice(ctx->device));
    abz_series* s = new abz_series();
    s->ctx = ctx;
    s->d = d;
    s->n = n;
    for (int j = 0; j < d; ++j) {
        if (dims[j] < 1 || !(period[j] > 0)) {
            delete s;
            set_error("dims[%d] = %d / period = %g invalid", j, dims[j], period[j]);
            return ABZ_ERR_ARG;
        }
        s->dims[j] = dims[j];
        s->first[j] = first[j];
        s->period[j] = period[j];
    }
    const size_t bytes = sizeof(double2) * (size_t)s->elems(d);
    if (dev_alloc((void**)&s->coef, bytes, &s->coef_cap)) {  // through the library's allocator: abz_mem_info sees it
        delete s;
        return ABZ_ERR_NOMEM;
    }
    hipError_t e = hipMemcpy(s->coef, coef_reim, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        dev_free(s->coef, s->coef_cap);
        delete s;
        set_error("coefficient upload failed: %s", hipGetErrorString(e));
        return ABZ_ERR_HIP;
    }
    s->hermitian = detect_hermitian(s, coef_reim);
    ctx->refs += 1;
    *out = s;
    return ABZ_OK;
} ABZ_CATCH_ALL

// c(-R) == c(R)^dagger for every R (exact comparison) and symmetric frequency ranges
static bool detect_hermitian(const abz_series* s, const double* coef_reim) {
    const int d = s->d, n = s->n;
    for (int j = 0; j < d; ++j)
        if (s->first[j] != -(s->dims[j] - 1) - s->first[j]) return false;  // first = -(M-1)/2, M odd
    int64_t nR = 1;
    for (int j = 0; j < d; ++j) nR *= s->dims[j];
    const int nn = n * n;
    for (int64_t r = 0; r < nR; ++r) {
        int64_t rem = r, mr = 0, mul = 1;
        for (int j = 0; j < d; ++j) {
            const int64_t i = rem % s->dims[j];
            rem /= s->dims[j];
            mr += (s->dims[j] - 1 - i) * mul;
            mul *= s->dims[j];
        }
        if (mr < r) continue;
        const double* A = coef_reim + 2 * r * nn;
        const double* B = coef_reim + 2 * mr * nn;
        for (int b = 0; b < n; ++b)
            for (int a = 0; a < n; ++a) {
                // A[a,b] == conj(B[b,a]); blocks are column-major
                if (A[2 * (a + n * b)] != B[2 * (b + n * a)] || A[2 * (a + n * b) + 1] != -B[2 * (b + n * a) + 1])
                    return false;
            }
    }
    return true;
}

int abz_series_update(abz_series* s, const double* coef_reim) try {
    int rc = check_series(s);
    if (rc) return rc;
    ABZ_REQUIRE(coef_reim, "null coefficients");
    ABZ_HIP(hipSetDevice(s->ctx->device));
    ABZ_HIP(hipMemcpyAsync(s->coef, coef_reim, sizeof(double2) * (size_t)s->elems(s->d), hipMemcpyHostToDevice,
                           s->ctx->stream));
    ABZ_HIP(hipStreamSynchronize(s->ctx->stream));
    s->hermitian = detect_hermitian(s, coef_reim);
    s->coef_pk_valid = false;
    s->generation += 1;  // rules kept by the series refill themselves at their next use
    for (abz_series* v : s->lanes) {
        v->hermitian = s->hermitian;
        v->coef_pk_valid = false;
        v->generation += 1;
    }
    return ABZ_OK;
} ABZ_CATCH_ALL

}  // extern "C"

namespace abz {
int series_lane_views(abz_series* s, int count) {
    while ((int)s->lanes.size() < count) {
        abz_ctx* c = nullptr;
        int rc = abz_ctx_create(s->ctx->device, &c);
        if (rc) return rc;
        abz_series* v = new abz_series();
        v->ctx = c;
        v->d = s->d;
        v->n = s->n;
        for (int j = 0; j < ABZ_MAX_DIM; ++j) {
            v->dims[j] = s->dims[j];
            v->first[j] = s->first[j];
            v->period[j] = s->period[j];
        }
        v->hermitian = s->hermitian;
        v->coef = s->coef;
        v->coef_borrowed = true;
        c->refs += 1;
        s->lanes.push_back(v);
    }
    return ABZ_OK;
}
}  // namespace abz

extern "C" {

static void rule_free(abz_rule* r);
static void series_drop_kept_rules(abz_series* s) {
    if (s->kept_rules.empty()) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    for (auto& k : s->kept_rules) rule_free(k.r);  // they hold no reference on the series
    s->kept_rules.clear();
    s->summed_once.clear();
}

int abz_series_destroy(abz_series* s) try {
    if (!s || s->closed) return ABZ_OK;
    series_drop_kept_rules(s);
    s->closed = true;
    series_release(s);  // freed once the last rule built from it is gone
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_series_drop_rules(abz_series* s) try {
    int rc = check_series(s);
    if (rc) return rc;
    series_drop_kept_rules(s);
    return ABZ_OK;
} ABZ_CATCH_ALL

// ---------------------------------------------------------------- rules
namespace abz {
struct RulePlan {
    Plan plan;
    PlanDev pd;
    DevBuf tab;
    DevBuf tmpU, tmpD;  // eigenvector / derivative planes while velocities are built (unfused build)
    DevBuf fam[2];      // fused GGR build: coefficient sets with the derivative factor on variable 2 / 3
    // 33...64 bands, Hermitian rules with cached H(k): the tridiagonal form (d, |e|^2) of every node, 1 KB per node, filled by the
    // first DOS / tr G scan -- later scans of the same values skip the Householder pass, which is most of a scan
    DevBuf tri;
    int tri_state = 0;  // 0: not filled for the current values
};
}  // namespace abz

static void rule_free(abz_rule* r) {
    if (!r) return;
    dev_free(r->vals, r->vals_cap);
    if (!r->tables_view) {
        dev_free(r->w, r->w_cap);
        dev_free(r->idx, r->idx_cap);
    }
    if (r->plan) {
        RulePlan* rp = static_cast<RulePlan*>(r->plan);
        rp->pd.release();
        rp->tab.release();
        rp->tmpU.release();
        rp->tmpD.release();
        rp->fam[0].release();
        rp->fam[1].release();
        rp->tri.release();
        delete rp;
    }
    delete r;
}

int abz_rule_destroy(abz_rule* r) try {
    if (!r) return ABZ_OK;
    abz_series* s = r->s;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    rule_free(r);
    series_release(s);
    return ABZ_OK;
} ABZ_CATCH_ALL

// eigenvalues + velocities only, Hermitian series, 5...32 bands, full grids or node lists whose nodes come in runs per
// level-1 set: the fused build in the row layout (kernels_ggr_rows.hip).  ABZ_GGR_FUSED=0: the unfused build (tests compare)
static bool rule_ggr_rows(const abz_rule* r) {
    const abz_series* s = r->s;
    const RulePlan* rp = static_cast<const RulePlan*>(r->plan);
    if (!(r->want & ABZ_WANT_VEL) || !rp) return false;
    if (big_supported(s->n)) {  // 33...64 bands: the only velocity build there is; H, when wanted as well, by the plain build after it
        if (!r->full && rp->plan.coords) return false;  // (lists of grid nodes; explicit coordinates have no velocity build)
        return big_ggr_supported(s->n, s->d, s->dims[0], r->npt, s->hermitian);
    }
    if ((r->want & ABZ_WANT_H) || !abz_switch(SW_GGR_FUSED)) return false;
    if (!r->full && !(s->d >= 2 && rp->plan.nruns > 0 && !rp->plan.coords)) return false;
    return ggr_rows_supported(s->n, s->d, s->dims[0], r->npt, s->hermitian);
}

// eigenvalues + velocities only, Hermitian series: one of the fused GGR builds applies (n <= 4: kernels_ggr.hip)
static bool rule_ggr_fused(const abz_rule* r) {
    const abz_series* s = r->s;
    return ((r->want & ABZ_WANT_VEL) && !(r->want & ABZ_WANT_H) &&
            ggr_build_supported(s->n, s->d, s->dims[0], r->npt, s->hermitian)) ||
           rule_ggr_rows(r);
}

// launches only: contraction chain(s) + innermost evaluation (+ velocities)
static int rule_fill(abz_rule* r) {
    abz_series* s = r->s;
    abz_ctx* ctx = s->ctx;
    RulePlan* rp = static_cast<RulePlan*>(r->plan);
    const Plan& plan = rp->plan;
    const int d = s->d, n = s->n;
    const double2* tab = rp->tab.as<double2>();
    r->herm = s->hermitian;
    rp->tri_state = 0;  // new values: the cached tridiagonal forms are stale
    if (r->H.compact && !s->hermitian) {
        set_error("the rule keeps H(k) as an upper triangle (ABZ_WANT_H_COMPACT) and the series is no longer Hermitian: build a new rule");
        return ABZ_ERR_ARG;
    }
    bool vel_done = false;
    if (rule_ggr_rows(r)) {
        // 5...32 bands: H, every dH/dk_j, eigenvalues, eigenvectors and velocities of a node in the registers of its lanes
        GgrRowsSpec gs;
        gs.n = n;
        gs.d = d;
        gs.M = s->dims[0];
        gs.first = s->first[0];
        gs.npt = r->npt;
        gs.tab = tab;
        gs.E = r->E;
        gs.V = r->V;
        if (r->full) {
            gs.nlines = d == 1 ? 1 : plan.nitems[1];
        } else {
            gs.nlines = plan.nruns;
            gs.run_start = rp->pd.runs.as<int64_t>();  // (33...64 bands: only "a node list" is read from it)
            gs.gi = rp->pd.gi[0].as<int32_t>();
            gs.parents = d >= 2 ? rp->pd.parent[0].as<int64_t>() : nullptr;  // (one variable: one level-1 set)
            gs.nk = r->nk;
        }
        int rc;
        if ((rc = build_chain(s, plan, rp->pd, tab, 0, &gs.src[0]))) return rc;
        for (int j = 2; j <= d; ++j)
            if ((rc = build_chain(s, plan, rp->pd, tab, j, &gs.src[j - 1], 1, &rp->fam[j - 2]))) return rc;
        if (!big_supported(n)) return launch_ggr_rows(ctx, gs);
        if ((rc = launch_big_ggr(ctx, gs)) || !(r->want & ABZ_WANT_H)) return rc;
        vel_done = true;  // the matrices as well: the plain build below, without its eigenvalues
    }
    if (!vel_done && rule_ggr_fused(r)) {
        // Fused GGR build (kernels_ggr.hip): H, every dH/dk_j, the eigensolve and the velocities in one kernel; only
        // (e, v) reach HBM.  ref: src/dos_ggr.jl:14-44
        GgrBuildSpec gs;
        gs.n = n;
        gs.d = d;
        gs.M = s->dims[0];
        gs.first = s->first[0];
        gs.npt = r->npt;
        gs.tab = tab;
        gs.E = r->E;
        gs.V = r->V;
        gs.grid = r->full;
        int rc;
        if (r->full) {
            gs.nlines = d == 1 ? 1 : plan.nitems[1];
            gs.fuse = d >= 2 && ggr_build_can_fuse(n, d, s->dims[0], s->dims[1], r->npt);
            if (gs.fuse) {
                gs.M2 = s->dims[1];
                gs.first2 = s->first[1];
                gs.gbeg = (d == 2) ? plan.outer0 : 0;
                gs.gcnt = (d == 2) ? plan.outer_n : r->npt;
                if ((rc = build_chain(s, plan, rp->pd, tab, 0, &gs.src2[0], 2))) return rc;
                if (d == 3 && (rc = build_chain(s, plan, rp->pd, tab, 3, &gs.src2[1], 2, &rp->fam[1]))) return rc;
                const int64_t nparents = gs.nlines / std::max(gs.gcnt, 1);
                if ((rc = rp->fam[0].reserve(sizeof(double2) * ggr_build_pack2_elems(n, d, gs.M, gs.M2, nparents)))) return rc;
                gs.pack2 = rp->fam[0].as<double2>();
            } else {
                if ((rc = build_chain(s, plan, rp->pd, tab, 0, &gs.src[0]))) return rc;
                for (int j = 2; j <= d; ++j)
                    if ((rc = build_chain(s, plan, rp->pd, tab, j, &gs.src[j - 1], 1, &rp->fam[j - 2]))) return rc;
            }
        } else {
            gs.nk = r->nk;
            gs.parents = d == 1 ? nullptr : rp->pd.parent[0].as<int64_t>();
            gs.gi = rp->pd.gi[0].as<int32_t>();
            if ((rc = build_chain(s, plan, rp->pd, tab, 0, &gs.src[0]))) return rc;
            for (int j = 2; j <= d; ++j)
                if ((rc = build_chain(s, plan, rp->pd, tab, j, &gs.src[j - 1], 1, &rp->fam[j - 2]))) return rc;
        }
        return launch_ggr_build(ctx, gs);
    }
    // temporaries of a velocity build: eigenvectors and one derivative matrix, tiled like H alone
    PlaneView Uv, Dv;
    if ((r->want & ABZ_WANT_VEL) && !vel_done) {
        Uv.base = rp->tmpU.as<double>();
        Uv.pitch = r->H.row ? r->H.row : r->E.row;  // temporaries are tiled whatever the rule's layout
        Uv.row = Uv.pitch;
        Uv.line_len = r->E.line_len;
        Uv.tile = (int64_t)2 * n * n * Uv.pitch;
        Dv = Uv;
        Dv.base = rp->tmpD.as<double>();
    }
    // full grids of a Hermitian series (n <= 4, values / eigenvalues only): the chain and the grid kernel work on packed sets
    const bool packed_chain = r->full && s->hermitian && !(r->want & ABZ_WANT_VEL) && eval_packed_supported(n, s->dims[0], r->npt);
    auto run_eval = [&](const double2* level1, bool deriv, PlaneView Hout, PlaneView Eout, PlaneView Uout) -> int {
        EvalSpec es;
        es.n = n;
        es.M = s->dims[0];
        es.first = s->first[0];
        es.period = s->period[0];
        es.src = level1;
        es.grid = r->full;
        es.npt = r->npt;
        es.nlines = r->full ? (d == 1 ? 1 : plan.nitems[1]) : 0;
        es.tab = tab;
        es.nk = r->nk;
        es.parents = r->full ? nullptr : rp->pd.parent[0].as<int64_t>();
        es.gi = r->full ? nullptr : rp->pd.gi[0].as<int32_t>();
        if (!r->full && plan.nruns > 0 && !plan.coords) {  // runs of nodes per level-1 set (row kernels, n > 4)
            es.run_start = rp->pd.runs.as<int64_t>();
            es.nruns = plan.nruns;
        }
        es.x = nullptr;
        es.deriv = deriv;
        es.herm = s->hermitian;
        es.packed = packed_chain && !deriv && !Uout.base;
        es.H = Hout;
        es.E = Eout;
        es.U = Uout;
        return launch_eval(ctx, es);
    };
    const double2* level1 = nullptr;
    int rc;
    if ((rc = build_chain(s, plan, rp->pd, tab, 0, &level1, 1, nullptr, packed_chain))) return rc;
    if ((rc = run_eval(level1, false, r->H, vel_done ? PlaneView() : r->E, Uv))) return rc;
    if ((r->want & ABZ_WANT_VEL) && !vel_done) {
        // d/dx_1 reuses the level-1 sets; d/dx_j (j >= 2) rebuilds the chain with the derivative
        // factor on variable j (JacobianSeries, ref src/dos_ggr.jl:6-7)
        for (int j = 1; j <= d; ++j) {
            if (j >= 2)
                if ((rc = build_chain(s, plan, rp->pd, tab, j, &level1))) return rc;
            if ((rc = run_eval(level1, j == 1, Dv, PlaneView(), PlaneView()))) return rc;
            PlaneView Vj = r->V;
            Vj.base += (int64_t)(j - 1) * n * Vj.pitch;
            if ((rc = launch_velocity(ctx, n, Uv, Dv, Vj, r->nk))) return rc;
        }
    }
    return ABZ_OK;
}

#define RULE_TRY(expr)        \
    do {                      \
        int rc_ = (expr);     \
        if (rc_) {            \
            rule_free(r);     \
            return rc_;       \
        }                     \
    } while (0)
#define RULE_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            set_error("%s failed: %s", #call, hipGetErrorString(e_));                    \
            rule_free(r);                                                                \
            return e_ == hipErrorOutOfMemory ? ABZ_ERR_NOMEM : ABZ_ERR_HIP;              \
        }                                                                                \
    } while (0)

static int rule_build(abz_series* s, int npt, int64_t nirr, const int32_t* irr_idx, const int64_t* wsym, int want,
                      int outer0, int outer_n, abz_rule** out, const SymTables* st = nullptr, bool wait = true) {
    int rc = check_series(s);
    if (rc) return rc;
    ABZ_REQUIRE(out, "null out");
    *out = nullptr;
    ABZ_REQUIRE(npt >= 1, "npt = %d must be positive", npt);
    ABZ_REQUIRE((want & (ABZ_WANT_H | ABZ_WANT_EIG | ABZ_WANT_VEL)) != 0, "want = %d selects nothing", want);
    ABZ_REQUIRE((irr_idx == nullptr) == (wsym == nullptr), "irr_idx and wsym must be given together");
    ABZ_REQUIRE(!st || (st->d == s->d && st->npt == npt && st->nk > 0), "symmetric rule tables do not fit the series");
    if (st) nirr = st->nk;
    if (want & ABZ_WANT_VEL) want |= ABZ_WANT_EIG;
    abz_ctx* ctx = s->ctx;
    ABZ_HIP(hipSetDevice(ctx->device));
    const int d = s->d, n = s->n;
    abz_rule* r = new abz_rule();
    RulePlan* rp = new RulePlan();
    r->plan = rp;
    r->s = s;
    r->npt = npt;
    r->want = want;
    r->full = irr_idx == nullptr && st == nullptr;
    // ABZ_DEBUG_TIMING=1: wall time of the build phases on stderr (host plan, uploads, allocation, fill)
    const bool dbg = abz_switch(SW_DEBUG_TIMING) != 0;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tdbg = tnow();
    auto lap = [&](const char* what) {
        if (dbg) {
            (void)hipStreamSynchronize(ctx->stream);
            const double t = tnow();
            fprintf(stderr, "[abz] rule_build %-14s %8.3f ms\n", what, 1e3 * (t - tdbg));
            tdbg = t;
        }
    };
    Plan& plan = rp->plan;
    if (r->full) {
        plan_full(plan, d, npt, outer0, outer_n);
        r->k_offset = (int64_t)outer0;
        for (int j = 0; j + 1 < d; ++j) r->k_offset *= npt;
    } else if (st) {  // device-resident node list and plan (sym_tables_device): nothing is walked or uploaded here
        plan.d = d;
        plan.full = false;
        plan.coords = false;
        plan.npt = npt;
        plan.nk = st->nk;
        for (int L = 0; L <= ABZ_MAX_DIM; ++L) plan.nitems[L] = st->nitems[L];
        plan.nruns = d >= 2 ? st->nitems[1] : 0;
    } else {
        for (int64_t k = 0; k < nirr * d; ++k) {
            if (irr_idx[k] < 0 || irr_idx[k] >= npt) {
                set_error("irr_idx[%lld] = %d outside the grid", (long long)k, irr_idx[k]);
                rule_free(r);
                return ABZ_ERR_ARG;
            }
        }
        plan_runs<int32_t>(plan, d, npt, irr_idx, nirr, false);
    }
    lap("plan");
    r->nk = plan.nk;
    // tiles: a grid line (pitch = npt rounded up to 16 doubles = 128 B) or 64 nodes of an irregular list
    const int line_len = r->full ? npt : 64;
    const int pitch = (line_len + 15) / 16 * 16;  // whole 128-B lines: 64-B and 16-B quanta measured 25-40 % slower
    r->ntiles = std::max<int64_t>(1, (plan.nk + line_len - 1) / line_len);
    // Hermitian-compact H planes (abzhip.h): the upper triangle only
    // (1...4 bands: the grid / node kernels of kernels.hip; 5...16 bands: the row kernel of kernels_generic.hip, full grids and
    // node lists with runs -- the wave-per-node fallback writes the full layout only)
    const bool compact = (want & ABZ_WANT_H) && (want & ABZ_WANT_H_COMPACT) && s->hermitian &&
                         (n <= 4 || (!(want & ABZ_WANT_VEL) && (r->full || d >= 2) && gen_compact_supported(n, s->dims[0], npt)));
    if (!compact) r->want = (want &= ~ABZ_WANT_H_COMPACT);
    const int pH = (want & ABZ_WANT_H) ? (compact ? n * n : 2 * n * n) : 0;
    const int pE = (want & ABZ_WANT_EIG) ? n : 0;
    const int pV = (want & ABZ_WANT_VEL) ? d * n : 0;
    r->planes = pH + pE + pV;
    // layout of the value planes: tiles [line][plane][row] (a padded planar layout [plane][line][row] was built and measured
    // no better in the real kernel; PlaneView still carries both strides)
    const int64_t tile = (int64_t)r->planes * pitch;
    const int pstride = pitch;  // plane to plane
    if (st) {  // ONE device-to-device copy of the cached tables (a single block): the rule owns its plan like any other
        RULE_TRY(rp->pd.arena.reserve(st->arena_bytes));
        RULE_HIP(hipMemcpyAsync(rp->pd.arena.p, st->arena.p, st->arena_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        auto at = [&](const void* q) -> void* { return static_cast<char*>(rp->pd.arena.p) + (static_cast<const char*>(q) - static_cast<const char*>(st->arena.p)); };
        auto view = [&](DevBuf& b, const void* q) {
            b.release();
            b.p = at(q);
            b.view = true;
        };
        for (int L = 0; L < d; ++L) {
            view(rp->pd.gi[L], st->gi[L]);
            view(rp->pd.parent[L], st->parent[L]);
        }
        if (d >= 2) view(rp->pd.runs, st->runs);
        r->w = static_cast<double*>(at(st->w));
        r->idx = static_cast<int32_t*>(at(st->idx));
        r->tables_view = true;
    } else {
        RULE_TRY(plan_upload(ctx, plan, rp->pd));
    }
    lap("plan_upload");
    RULE_TRY(make_phase_table(ctx, npt, rp->tab));
    lap("phase_table");
    if (r->full) {  // per-level phase tables [npt][M] for the scalar-phase contraction kernel
        for (int L = 1; L < d; ++L) {
            const int M = s->dims[L];
            RULE_TRY(rp->pd.phg[L].reserve(sizeof(double2) * (size_t)npt * M));
            PhaseSpec ps;
            ps.B = npt;
            ps.M = M;
            ps.first = s->first[L];
            ps.gi = nullptr;
            ps.x = nullptr;
            ps.tab = rp->tab.as<double2>();
            ps.npt = npt;
            ps.g0 = 0;
            ps.gcnt = npt;
            ps.period = s->period[L];
            ps.deriv = false;
            RULE_TRY(launch_phases(ctx, ps, rp->pd.phg[L].as<double2>()));
        }
    }
    const size_t bytes = sizeof(double) * (size_t)(r->ntiles * (int64_t)r->planes * pitch);
    RULE_TRY(dev_alloc((void**)&r->vals, bytes, &r->vals_cap));
    // on the context's stream: it is non-blocking, a null-stream memset would not be ordered before the
    // fill kernels below (and could land on top of their results)
    // Irregular node lists: the last tile is partly empty and kernels that walk whole tiles multiply its slots by a zero
    // weight -- they must hold finite numbers.  Full grids have no partial tile (nk = lines x npt) and no kernel addresses
    // the padding columns npt .. pitch-1 of a row; the grid kernel of 1...4 bands writes them as filler anyway, so the
    // 346 MB memset of a 150^3 rule (0.045 ms, 0.8 ms at 400^3) is skipped there.  The other fill kernels (GGR builds,
    // 5...32 bands, velocity planes) leave the padding alone: the block comes from a recycling pool, and a client of
    // abz_rule_values_ptr sees the whole block, so it is zeroed once here (abzhip.h: padding is zero or filler, finite).
    const bool filler_written = r->full && n <= 4 && !(want & ABZ_WANT_VEL);
    if (!(r->full && (pitch == line_len || filler_written))) RULE_HIP(hipMemsetAsync(r->vals, 0, bytes, ctx->stream));
    auto mkview = [&](int plane0, bool present) {
        PlaneView v;
        if (present) {
            v.base = r->vals + (int64_t)plane0 * pstride;
            v.tile = tile;
            v.pitch = pstride;
            v.line_len = line_len;
            v.row = pitch;
        }
        return v;
    };
    r->H = mkview(0, pH > 0);
    r->H.compact = compact ? n : 0;
    r->E = mkview(pH, pE > 0);
    r->V = mkview(pH + pE, pV > 0);
    if (st) {
        // (w and idx came with the plan's copy of the tables)
    } else if (!r->full) {
        std::vector<double> wd(std::max<int64_t>(nirr, 1));
        for (int64_t k = 0; k < nirr; ++k) wd[k] = (double)wsym[k];
        RULE_TRY(dev_alloc((void**)&r->w, sizeof(double) * wd.size(), &r->w_cap));
        RULE_TRY(stage_h2d(ctx, r->w, wd.data(), sizeof(double) * (size_t)nirr));
        std::vector<int32_t> it((size_t)std::max<int64_t>(nirr * d, 1));
        for (int64_t k = 0; k < nirr; ++k)
            for (int j = 0; j < d; ++j) it[(size_t)j * nirr + k] = irr_idx[k * d + j];
        RULE_TRY(dev_alloc((void**)&r->idx, sizeof(int32_t) * it.size(), &r->idx_cap));
        RULE_TRY(stage_h2d(ctx, r->idx, it.data(), sizeof(int32_t) * (size_t)(nirr * d)));
    }
    if ((want & ABZ_WANT_VEL) && !rule_ggr_fused(r)) {
        const size_t tb = sizeof(double) * (size_t)(r->ntiles * 2 * n * n * pitch);  // tiled temporaries
        RULE_TRY(rp->tmpU.reserve(tb));
        RULE_TRY(rp->tmpD.reserve(tb));
    }
    lap("alloc+w/idx");
    RULE_TRY(rule_fill(r));
    // the C-ABI hands out a finished rule (a client may read its values from another stream); the library's own solve
    // loops (series_rule) keep going: the scan of the rule is ordered behind its fill on the context's stream
    if (wait || dbg) RULE_HIP(hipStreamSynchronize(ctx->stream));
    lap("fill");
    if (want & ABZ_WANT_VEL) {  // keep the big temporaries only while a rebuild needs them
        rp->tmpU.release();
        rp->tmpD.release();
    }
    s->refs += 1;
    *out = r;
    return ABZ_OK;
}
#undef RULE_TRY
#undef RULE_HIP

int abz_ptr_rule_build(abz_series* s, int npt, int64_t nirr, const int32_t* irr_idx, const int64_t* wsym, int want,
                       abz_rule** out) try {
    return rule_build(s, npt, nirr, irr_idx, wsym, want, 0, npt, out);
} ABZ_CATCH_ALL

static int rule_build_sym(abz_series* s, int npt, const int32_t* syms, int nsyms, int want, abz_rule** out, bool wait);
int abz_ptr_rule_build_sym(abz_series* s, int npt, const int32_t* syms, int nsyms, int want, abz_rule** out) try {
    return rule_build_sym(s, npt, syms, nsyms, want, out, true);
} ABZ_CATCH_ALL
static int rule_build_sym(abz_series* s, int npt, const int32_t* syms, int nsyms, int want, abz_rule** out, bool wait) {
    int rc = check_series(s);
    if (rc) return rc;
    ABZ_REQUIRE(out && syms && nsyms >= 1 && npt >= 1, "abz_ptr_rule_build_sym: bad arguments");
    *out = nullptr;
    abz_ctx* ctx = s->ctx;
    ABZ_HIP(hipSetDevice(ctx->device));
    const int d = s->d;
    const size_t nsy = (size_t)nsyms * d * d;
    SymTables* st = nullptr;
    for (size_t i = 0; i < ctx->sym_cache.size(); ++i) {
        SymTables* c = ctx->sym_cache[i];
        if (c->npt == npt && c->d == d && c->syms.size() == nsy && std::equal(c->syms.begin(), c->syms.end(), syms)) {
            st = c;
            ctx->sym_cache.erase(ctx->sym_cache.begin() + (long)i);
            break;
        }
    }
    if (!st) {
        st = new SymTables();
        rc = sym_tables_device(ctx, npt, d, syms, nsyms, *st);
        if (rc) {
            st->release();
            delete st;
            return rc;
        }
    }
    ctx->sym_cache.push_back(st);
    // a few grids stay cached (AutoPTR walks npt = 50, 100, 150, ...): 32 B per irreducible node
    size_t bytes = 0;
    for (const SymTables* c : ctx->sym_cache) bytes += (size_t)c->nk * 40;
    while (ctx->sym_cache.size() > 8 || (bytes > ((size_t)2 << 30) && ctx->sym_cache.size() > 1)) {
        SymTables* old = ctx->sym_cache.front();
        ctx->sym_cache.erase(ctx->sym_cache.begin());
        bytes -= (size_t)old->nk * 40;
        (void)hipStreamSynchronize(ctx->stream);
        old->release();
        delete old;
    }
    ABZ_REQUIRE(st->nk > 0, "the symmetry set leaves no node");
    return rule_build(s, npt, 0, nullptr, nullptr, want, 0, npt, out, st, wait);
}

int abz_ptr_rule_build_slab(abz_series* s, int npt, int outer_begin, int outer_end, int want, abz_rule** out) try {
    int rc = check_series(s);
    if (rc) return rc;
    ABZ_REQUIRE(s->d >= 2, "a slab needs at least two variables (d = %d)", s->d);
    ABZ_REQUIRE(0 <= outer_begin && outer_begin < outer_end && outer_end <= npt,
                "slab [%d, %d) outside the grid of %d points", outer_begin, outer_end, npt);
    return rule_build(s, npt, 0, nullptr, nullptr, want, outer_begin, outer_end - outer_begin, out);
} ABZ_CATCH_ALL

int abz_rule_rebuild(abz_rule* r) try {
    int rc0 = check_rule(r);
    if (rc0) return rc0;
    abz_ctx* ctx = r->s->ctx;
    ABZ_HIP(hipSetDevice(ctx->device));
    RulePlan* rp = static_cast<RulePlan*>(r->plan);
    if ((r->want & ABZ_WANT_VEL) && !rule_ggr_fused(r)) {
        const size_t tb = sizeof(double) * (size_t)(r->ntiles * 2 * r->s->n * r->s->n * r->E.row);
        int rc = rp->tmpU.reserve(tb);
        if (rc) return rc;
        if ((rc = rp->tmpD.reserve(tb))) return rc;
    }
    return rule_fill(r);
} ABZ_CATCH_ALL

int abz_rule_info(const abz_rule* r, int64_t* nk, int* n, int* d, int* npt, int* want) try {
    ABZ_REQUIRE(r, "null rule");
    if (nk) *nk = r->nk;
    if (n) *n = r->s->n;
    if (d) *d = r->s->d;
    if (npt) *npt = r->npt;
    if (want) *want = r->want;
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_rule_export(abz_rule* r, double* x, double* w, double* H, double* eig, double* vel) try {
    int rc0 = check_rule(r);
    if (rc0) return rc0;
    abz_ctx* ctx = r->s->ctx;
    ABZ_HIP(hipSetDevice(ctx->device));
    const int d = r->s->d, n = r->s->n;
    if (x || w) {
        std::vector<int32_t> it;
        std::vector<double> wd;
        if (!r->full) {
            it.resize((size_t)(r->nk * d));
            wd.resize((size_t)r->nk);
            ABZ_HIP(hipMemcpy(it.data(), r->idx, sizeof(int32_t) * it.size(), hipMemcpyDeviceToHost));
            ABZ_HIP(hipMemcpy(wd.data(), r->w, sizeof(double) * wd.size(), hipMemcpyDeviceToHost));
        }
        for (int64_t k = 0; k < r->nk; ++k) {
            int64_t rem = k + r->k_offset;
            for (int j = 0; j < d; ++j) {
                int gi;
                if (r->full) {
                    gi = (int)(rem % r->npt);
                    rem /= r->npt;
                } else {
                    gi = it[(size_t)j * r->nk + k];
                }
                if (x) x[k * d + j] = (double)gi / (double)r->npt;
            }
            if (w) w[k] = r->full ? 1.0 : wd[(size_t)k];
        }
    }
    if (H) {
        ABZ_REQUIRE(r->H.base, "rule holds no H(k) (want lacked ABZ_WANT_H)");
        int rc = export_planes(ctx, r->H, 2 * n * n, r->nk, H);
        if (rc) return rc;
    }
    if (eig) {
        ABZ_REQUIRE(r->E.base, "rule holds no eigenvalues (want lacked ABZ_WANT_EIG)");
        int rc = export_planes(ctx, r->E, n, r->nk, eig);
        if (rc) return rc;
    }
    if (vel) {
        ABZ_REQUIRE(r->V.base, "rule holds no velocities (want lacked ABZ_WANT_VEL)");
        int rc = export_planes(ctx, r->V, d * n, r->nk, vel);
        if (rc) return rc;
    }
    return ABZ_OK;
} ABZ_CATCH_ALL

static int rule_reduce(abz_rule* r, int integrand, const double* params, int nparams, const double* sweep, int n_sweep,
                       int nsyms, double* out_reim, bool device_io, double2* map_dev = nullptr);

int abz_rule_reduce(abz_rule* r, int integrand, const double* params, int nparams, const double* sweep, int n_sweep,
                    int nsyms, double* out_reim) try {
    return rule_reduce(r, integrand, params, nparams, sweep, n_sweep, nsyms, out_reim, false);
} ABZ_CATCH_ALL

int abz_rule_reduce_device(abz_rule* r, int integrand, const double* params, int nparams, const double* sweep_dev,
                           int n_sweep, int nsyms, double* out_dev_reim) try {
    return rule_reduce(r, integrand, params, nparams, sweep_dev, n_sweep, nsyms, out_dev_reim, true);
} ABZ_CATCH_ALL

int abz_rule_values_ptr(const abz_rule* r, void** base, int64_t* nbytes) try {
    int rc0 = check_rule(r);
    if (rc0) return rc0;
    if (base) *base = r->vals;
    if (nbytes) *nbytes = (int64_t)sizeof(double) * r->ntiles * r->planes * (r->H.base ? r->H.row : r->E.row);
    return ABZ_OK;
} ABZ_CATCH_ALL

// device_io: `sweep` and `out_reim` are device pointers, nothing is synchronised; with `map_dev` (n <= 4) the sums are
// written by the last kernel straight into that device-visible host address instead of `out_reim`
static int rule_reduce(abz_rule* r, int integrand, const double* params, int nparams, const double* sweep, int n_sweep,
                       int nsyms, double* out_reim, bool device_io, double2* map_dev) {
    int rc0 = check_rule(r);
    if (rc0) return rc0;
    ABZ_REQUIRE(out_reim || map_dev, "null out");
    ABZ_REQUIRE(nparams >= 0 && nparams <= 4, "nparams = %d not in 0..4", nparams);
    ABZ_REQUIRE(nsyms >= 1, "nsyms must be >= 1");
    abz_ctx* ctx = r->s->ctx;
    ABZ_HIP(hipSetDevice(ctx->device));
    const bool use_eig = integrand == ABZ_F_DOS_EIG;
    if (use_eig)
        ABZ_REQUIRE(r->E.base, "integrand needs cached eigenvalues: build the rule with ABZ_WANT_EIG");
    else if (integrand != ABZ_F_ONE)
        ABZ_REQUIRE(r->H.base, "integrand needs cached H(k): build the rule with ABZ_WANT_H");
    const bool swept = integrand == ABZ_F_DOS || integrand == ABZ_F_TRGLOC || integrand == ABZ_F_GLOC ||
                       integrand == ABZ_F_DOS_EIG;
    int ns = swept ? n_sweep : 1;
    ABZ_REQUIRE(ns >= 1 && (!swept || sweep), "sweep values required for integrand %d", integrand);
    const int need = (integrand == ABZ_F_LINEAR || integrand == ABZ_F_LINEAR_X) ? 2 : (swept ? 1 : 0);
    ABZ_REQUIRE(nparams >= need && (need == 0 || params), "integrand %d needs %d parameters", integrand, need);
    ReduceSpec rs;
    rs.n = r->s->n;
    rs.d = r->s->d;
    rs.npt = r->npt;
    rs.integrand = integrand;
    rs.H = r->H.base ? r->H : r->E;  // ABZ_F_ONE never dereferences it
    rs.E = r->E;
    rs.nk = r->nk;
    rs.w = r->w;
    rs.idx = r->idx;
    rs.k_offset = r->k_offset;
    rs.herm = r->herm;
    if (big_supported(rs.n) && r->herm && r->H.base && !r->H.compact && r->plan && (integrand == ABZ_F_DOS || integrand == ABZ_F_TRGLOC)) {
        RulePlan* rp = static_cast<RulePlan*>(r->plan);
        const int64_t tnk = (r->nk + 63) / 64 * 64;
        if (rp->tri.reserve(sizeof(double) * 2 * 64 * (size_t)tnk) == ABZ_OK) {  // (no room: the scan tridiagonalises on the fly)
            rs.tri_cache = rp->tri.as<double>();
            rs.tri_nk = tnk;
            rs.tri_state = &rp->tri_state;
        }
    }
    for (int i = 0; i < 4; ++i) rs.params[i] = (i < nparams) ? params[i] : 0.0;
    rs.n_sweep = ns;
    rs.sweep_dev = nullptr;
    if (swept && device_io) {
        rs.sweep_dev = sweep;
    } else if (swept) {
        // through the pinned mailbox: an asynchronous copy on the context's stream (a pageable source is staged by the
        // runtime and the small-copy path of upload() synchronises the stream before the first launch)
        const size_t sb = sizeof(double) * (size_t)ns;
        int rc = ctx->scratch[5].reserve(sb);
        if (rc) return rc;
        if (mbox_reserve(ctx) == ABZ_OK && sb <= ctx->mbox_cap / 2) {
            std::memcpy(ctx->mbox, sweep, sb);
            ABZ_HIP(hipMemcpyAsync(ctx->scratch[5].p, ctx->mbox, sb, hipMemcpyHostToDevice, ctx->stream));
        } else if ((rc = upload(ctx, ctx->scratch[5], sweep, (size_t)ns))) {
            return rc;
        }
        rs.sweep_dev = ctx->scratch[5].as<double>();
    }
    double vol = 1.0;
    for (int j = 0; j < rs.d; ++j) vol *= (double)r->npt;
    rs.scale = 1.0 / (vol * (double)nsyms);
    if (device_io && map_dev && rs.n <= 4)
        rs.out_map_dev = map_dev;
    else if (device_io)
        rs.out_dev = out_reim;
    if (!device_io && rs.n <= 4 && ctx->mbox) {  // sums land in the second half of the mailbox (zero copy)
        const int nc = integrand_ncomp(integrand, rs.n, rs.d);
        if (nc > 0 && sizeof(double2) * (size_t)ns * (size_t)nc <= ctx->mbox_cap / 2) {
            rs.out_map_dev = reinterpret_cast<double2*>(static_cast<char*>(ctx->mbox_dev) + ctx->mbox_cap / 2);
            rs.out_map_host = reinterpret_cast<const double2*>(static_cast<const char*>(ctx->mbox) + ctx->mbox_cap / 2);
        }
    }
    return launch_reduce(ctx, rs, device_io ? nullptr : out_reim);
}

int abz_ptr_sum(abz_series* s, int npt, int outer_begin, int outer_end, int integrand, const double* params, int nparams,
                const double* sweep, int n_sweep, int nsyms, double* out_reim) try {
    int rc = check_series(s);
    if (rc) return rc;
    ABZ_REQUIRE(out_reim, "null out");
    ABZ_REQUIRE(npt >= 1 && nsyms >= 1, "npt = %d, nsyms = %d", npt, nsyms);
    ABZ_REQUIRE(nparams >= 0 && nparams <= 4, "nparams = %d not in 0..4", nparams);
    const int d = s->d, n = s->n;
    ABZ_REQUIRE(0 <= outer_begin && outer_begin < outer_end && outer_end <= npt, "slab [%d, %d) outside the grid of %d points",
                outer_begin, outer_end, npt);
    ABZ_REQUIRE(d >= 2 || (outer_begin == 0 && outer_end == npt), "a slab needs at least two variables");
    // n <= 4 without a closed-form store-free kernel for this case (a series that is not Hermitian, a short grid line): the
    // inverse of every node like the larger matrices (kernels_big.hip), for the integrands it serves
    const bool inv_small = n <= 4 && !eval_sum_supported(n, s->dims[0], npt, integrand, s->hermitian) &&
                           (integrand == ABZ_F_DOS || integrand == ABZ_F_TRGLOC || integrand == ABZ_F_GLOC);
    const bool generic = n > 4 || inv_small;
    if (!inv_small && !(generic ? gen_sum_supported(n, s->dims[0], npt, integrand, s->hermitian)
                                : eval_sum_supported(n, s->dims[0], npt, integrand, s->hermitian))) {
        set_error("store-free sum not available for this series / grid / integrand (use a rule)");
        return ABZ_ERR_UNSUPPORTED;
    }
    const bool swept = integrand == ABZ_F_DOS || integrand == ABZ_F_TRGLOC || integrand == ABZ_F_GLOC || integrand == ABZ_F_DOS_EIG;
    ABZ_REQUIRE(!swept || (sweep && n_sweep >= 1), "sweep values required for integrand %d", integrand);
    const int need = (integrand == ABZ_F_LINEAR || integrand == ABZ_F_LINEAR_X) ? 2 : (swept ? 1 : 0);
    ABZ_REQUIRE(nparams >= need && (need == 0 || params), "integrand %d needs %d parameters", integrand, need);
    abz_ctx* ctx = s->ctx;
    ABZ_HIP(hipSetDevice(ctx->device));
    Plan plan;
    plan_full(plan, d, npt, d >= 2 ? outer_begin : 0, d >= 2 ? outer_end - outer_begin : npt);
    PlanDev pd;
    DevBuf tab;
    auto done = [&](int code) {
        pd.release();
        tab.release();
        return code;
    };
    if ((rc = make_phase_table(ctx, npt, tab))) return done(rc);
    for (int L = 1; L < d; ++L) {  // scalar-phase tables of the contraction levels
        const int M = s->dims[L];
        if ((rc = pd.phg[L].reserve(sizeof(double2) * (size_t)npt * M))) return done(rc);
        PhaseSpec ps;
        ps.B = npt;
        ps.M = M;
        ps.first = s->first[L];
        ps.gi = nullptr;
        ps.x = nullptr;
        ps.tab = tab.as<double2>();
        ps.npt = npt;
        ps.g0 = 0;
        ps.gcnt = npt;
        ps.period = s->period[L];
        ps.deriv = false;
        if ((rc = launch_phases(ctx, ps, pd.phg[L].as<double2>()))) return done(rc);
    }
    const double2* level1 = nullptr;
    // n <= 4: the store-free kernel takes packed Hermitian sets (eval_sum_supported requires a Hermitian series)
    if ((rc = build_chain(s, plan, pd, tab.as<double2>(), 0, &level1, 1, nullptr, !generic))) return done(rc);
    SumSpec ss;
    ss.n = n;
    ss.d = d;
    ss.M = s->dims[0];
    ss.first = s->first[0];
    ss.npt = npt;
    ss.src = level1;
    ss.tab = tab.as<double2>();
    ss.nlines = d == 1 ? 1 : plan.nitems[1];
    ss.line0 = (int64_t)(d >= 2 ? outer_begin : 0);
    for (int j = 0; j + 2 < d; ++j) ss.line0 *= npt;  // lines per outer index: npt^(d-2)
    ss.integrand = integrand;
    ss.herm = s->hermitian;
    ss.force_inverse = inv_small;
    ss.n_sweep = n_sweep;
    ss.sweep_host = sweep;
    for (int i = 0; i < 4; ++i) ss.params[i] = (i < nparams && params) ? params[i] : 0.0;
    double vol = 1.0;
    for (int j = 0; j < d; ++j) vol *= (double)npt;
    ss.scale = 1.0 / (vol * (double)nsyms);
    rc = inv_small ? launch_big_sum(ctx, ss, out_reim) : (generic ? launch_gen_sum(ctx, ss, out_reim) : launch_eval_sum(ctx, ss, out_reim));
    (void)hipStreamSynchronize(ctx->stream);
    return done(rc);
} ABZ_CATCH_ALL

// ---------------------------------------------------------------- whole AutoPTR solves
// The rule of grid `npt` kept by the series (built on first use, refilled after abz_series_update).  `keep` = false: a
// rule for one use; the caller frees it with rule_free.
static int series_rule(abz_series* s, int npt, const int32_t* syms, int nsyms, int want, bool keep, abz_rule** out, bool* owned) {
    const size_t nsy = syms ? (size_t)nsyms * s->d * s->d : 0;
    *owned = false;
    for (size_t i = 0; i < s->kept_rules.size(); ++i) {
        SeriesRule& k = s->kept_rules[i];
        if (k.npt != npt || (k.want & want) != want || k.syms.size() != nsy || !std::equal(k.syms.begin(), k.syms.end(), syms)) continue;
        if (k.generation != s->generation) {
            if (k.r->H.compact && !s->hermitian) {  // the series stopped being Hermitian: an upper-triangle rule cannot hold it
                rule_free(k.r);
                s->kept_rules.erase(s->kept_rules.begin() + (long)i);
                break;
            }
            int rc = rule_fill(k.r);
            if (rc) return rc;
            k.generation = s->generation;
        }
        k.stamp = ++s->kept_stamp;
        *out = k.r;
        return ABZ_OK;
    }
    abz_rule* r = nullptr;
    int rc = syms ? rule_build_sym(s, npt, syms, nsyms, want, &r, false) : rule_build(s, npt, 0, nullptr, nullptr, want, 0, npt, &r, nullptr, false);
    if (rc) return rc;
    s->refs -= 1;  // owned by the series or by the caller of this function: no reference cycle
    if (keep) {
        SeriesRule k;
        k.npt = npt;
        k.want = r->want;
        if (syms) k.syms.assign(syms, syms + nsy);
        k.r = r;
        k.generation = s->generation;
        k.stamp = ++s->kept_stamp;
        s->kept_rules.push_back(std::move(k));
    } else {
        *owned = true;
    }
    *out = r;
    return ABZ_OK;
}

static size_t rule_value_bytes(const abz_series* s, int npt, int64_t nk, int want) {
    const int n = s->n;
    const bool compact = (want & ABZ_WANT_H_COMPACT) && s->hermitian && (n <= 4 || gen_compact_supported(n, s->dims[0], npt));
    const size_t per = 8 * (size_t)(((want & ABZ_WANT_H) ? (compact ? n * n : 2 * n * n) : 0) + ((want & ABZ_WANT_EIG) ? n : 0));
    return per * (size_t)nk;
}

/* AutoSymPTR.autosymptr for the library's own integrands: I1 = rule(n0), I2 = rule(n0 + dn), err = norm(I2 - I1), refine
 * until err <= max(abstol, reltol norm(I2)) or numevals >= maxevals -- for n_sweep values of the swept parameter in
 * lock-step (every grid is built or found once and scanned for all solves that have not converged).
 * ref: src/algorithms.jl:418-432, src/fourier.jl:381-389, SURVEY A.2. */
int abz_autoptr_solve_many(abz_series* s, const int32_t* syms, int nsyms, int integrand, const double* params, int nparams,
                           const double* sweeps, int n_sweep, int n0, int dn, double abstol, double reltol, int64_t maxevals,
                           int keepmost, double value_factor, double* out_reim, double* err_out, int64_t* numevals_out,
                           int32_t* npt_out) try {
    int rc = check_series(s);
    if (rc) return rc;
    ABZ_REQUIRE(out_reim && n_sweep >= 1 && n0 >= 1 && dn >= 1, "abz_autoptr_solve_many: bad arguments (n0 = %d, dn = %d, n_sweep = %d)", n0, dn, n_sweep);
    ABZ_REQUIRE((syms == nullptr) == (nsyms <= 0) || (syms == nullptr && nsyms == 1), "syms and nsyms must be given together");
    ABZ_REQUIRE(nparams >= 0 && nparams <= 4, "nparams = %d not in 0..4", nparams);
    const int d = s->d, n = s->n;
    const int ncomp = integrand_ncomp(integrand, n, d);
    ABZ_REQUIRE(ncomp > 0, "unknown integrand id %d", integrand);
    const bool swept = integrand == ABZ_F_DOS || integrand == ABZ_F_TRGLOC || integrand == ABZ_F_GLOC || integrand == ABZ_F_DOS_EIG;
    ABZ_REQUIRE(!swept || sweeps, "sweep values required for integrand %d", integrand);
    ABZ_REQUIRE(swept || n_sweep == 1, "integrand %d has no swept parameter: n_sweep must be 1", integrand);
    abz_ctx* ctx = s->ctx;
    ABZ_HIP(hipSetDevice(ctx->device));
    const int ns_eff = syms ? nsyms : 1;
    if (maxevals <= 0) maxevals = (int64_t)1 << 62;
    double rtol, atol;
    if (abstol < 0 && reltol < 0) {
        rtol = std::sqrt(2.220446049250313e-16);
        atol = 0.0;
    } else {
        rtol = reltol < 0 ? 0.0 : reltol;
        atol = abstol < 0 ? 0.0 : abstol;
    }
    // rules of a Hermitian series keep the upper triangle of H(k): every built-in integrand reads those planes only
    int want = integrand == ABZ_F_DOS_EIG ? ABZ_WANT_EIG : ABZ_WANT_H;
    if ((want & ABZ_WANT_H) && s->hermitian) want |= ABZ_WANT_H_COMPACT;  // (ignored where no compact kernel exists)
    const size_t ncs = (size_t)ncomp;
    std::vector<double2> I1((size_t)n_sweep * ncs), I2((size_t)n_sweep * ncs), vals((size_t)n_sweep * ncs);
    std::vector<double> sw((size_t)n_sweep);
    std::vector<int> active((size_t)n_sweep);
    for (int i = 0; i < n_sweep; ++i) active[(size_t)i] = i;
    std::vector<int64_t> nev((size_t)n_sweep, 0);
    // staging: swept values of the active solves (one upload serves the first two grids) and, per grid in flight, the sums --
    // written by the scan's last kernel straight into the pinned mailbox where they fit (no copy call), else left in HBM and
    // fetched by one copy
    const size_t sw_bytes = sizeof(double) * (size_t)n_sweep, out_bytes = sizeof(double2) * (size_t)n_sweep * ncs;
    constexpr int NSLOT = 4;  // grids whose sums can be pending together
    if ((rc = s->auto_io.reserve(sw_bytes + NSLOT * out_bytes))) return rc;
    char* const io = static_cast<char*>(s->auto_io.p);
    // (a block of its own: the context's mailbox serves the calls made below -- abz_ptr_sum -- while sums are pending here)
    const size_t pin_need = sw_bytes + NSLOT * out_bytes;
    if (s->auto_pin_cap < pin_need) {
        (void)hipStreamSynchronize(ctx->stream);
        if (s->auto_pin) (void)hipHostFree(s->auto_pin);
        s->auto_pin = s->auto_pin_dev = nullptr;
        s->auto_pin_cap = 0;
        void *hp = nullptr, *dp = nullptr;
        const size_t cap = std::max<size_t>(pin_need, (size_t)16 << 10);
        if (hipHostMalloc(&hp, cap, hipHostMallocDefault) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp) {
            s->auto_pin = hp;
            s->auto_pin_dev = dp;
            s->auto_pin_cap = cap;
        } else {
            (void)hipGetLastError();
            if (hp) (void)hipHostFree(hp);
        }
    }
    const bool mb = s->auto_pin != nullptr;
    const bool mapped = mb && n <= 4;
    std::vector<char> host_io(mb ? 0 : pin_need);
    char* const hbase = mb ? static_cast<char*>(s->auto_pin) : host_io.data();
    double* const sw_h = reinterpret_cast<double*>(hbase);
    char* const out_h = hbase + sw_bytes;
    char* const out_map = mb ? static_cast<char*>(s->auto_pin_dev) + sw_bytes : nullptr;
    // few swept values: the kernels read them where the host wrote them (pinned, device-visible), no upload
    const bool sw_mapped = mapped && n_sweep <= 8 && abz_switch(SW_AUTO_SWEEP_MAPPED) != 0;
    double* const sw_d = sw_mapped ? reinterpret_cast<double*>(s->auto_pin_dev) : reinterpret_cast<double*>(io);
    size_t free_b = 0, total_b = 0;               // asked for when a rule has to be built
    const uint64_t call_stamp = s->kept_stamp + 1;  // rules stamped from here on serve this call
    bool sweeps_current = false;                    // sw_d holds the values of `active`

    // Value of the rule of grid `npt` for the active solves.  `slot` 0 / 1: so that the first two grids can be in flight
    // together.  Rule-based grids leave their sums pending (mailbox or HBM), store-free ones return them on the host.
    struct Pending {
        bool pending = false;
    };
    auto grid_value = [&](int npt, int gindex, int slot, const std::vector<int>& act, double2* host_vals, int64_t* nk_out, Pending* pend) -> int {
        const int na = (int)act.size();
        int64_t nk_full = 1;
        for (int j = 0; j < d; ++j) nk_full *= npt;
        bool have = false;
        for (const SeriesRule& k : s->kept_rules)
            have = have || (k.npt == npt && (k.want & want) == want && k.syms.size() == (syms ? (size_t)nsyms * d * d : 0) &&
                            std::equal(k.syms.begin(), k.syms.end(), syms));
        const bool sum_ok = !syms && (n > 4 ? gen_sum_supported(n, s->dims[0], npt, integrand, s->hermitian)
                                            : eval_sum_supported(n, s->dims[0], npt, integrand, s->hermitian));
        const size_t rbytes = rule_value_bytes(s, npt, nk_full, want);  // upper bound for symmetric rules
        if (!have) (void)hipMemGetInfo(&free_b, &total_b);  // (per grid: an earlier build of this call has taken its share)
        const bool fits = have || rbytes < free_b / 2;
        // kept: the first `keepmost` grids like the reference's cache, and any further one while the series' kept rules
        // stay below a quarter of the device memory (288 GB of HBM are there to keep rule values resident: a sweep or a
        // repeated solve then finds every grid of its sequence) -- least recently used rules of other grids make room
        bool keep = fits && gindex < keepmost;
        if (!have && fits && !keep) {
            size_t kept = 0;
            for (const SeriesRule& k : s->kept_rules) kept += (size_t)(sizeof(double) * k.r->ntiles * k.r->planes * (k.r->H.base ? k.r->H.row : k.r->E.row));
            const size_t budget = total_b / 4;
            while (kept + rbytes > budget && !s->kept_rules.empty()) {
                size_t victim = s->kept_rules.size();
                for (size_t i = 0; i < s->kept_rules.size(); ++i)
                    if (s->kept_rules[i].stamp < call_stamp && (victim == s->kept_rules.size() || s->kept_rules[i].stamp < s->kept_rules[victim].stamp)) victim = i;
                if (victim == s->kept_rules.size()) break;  // everything left serves this very call
                abz_rule* vr = s->kept_rules[victim].r;
                kept -= (size_t)(sizeof(double) * vr->ntiles * vr->planes * (vr->H.base ? vr->H.row : vr->E.row));
                (void)hipStreamSynchronize(ctx->stream);
                rule_free(vr);
                s->kept_rules.erase(s->kept_rules.begin() + (long)victim);
            }
            keep = kept + rbytes <= budget;
            // ... from its SECOND visit on where it can be summed on the fly: a one-off solve should not pay for storing it
            if (keep && sum_ok && na <= 8 && n <= 4 && std::find(s->summed_once.begin(), s->summed_once.end(), npt) == s->summed_once.end()) {
                keep = false;
                if (s->summed_once.size() < 64) s->summed_once.push_back(npt);
            }
        }
        // a grid used once: on the fly when few values share it (the store-free kernel takes 8 per pass), when it would not
        // fit, or above four bands (the panel kernels beat build + scan at any size)
        if (!have && !keep && sum_ok && (na <= 8 || !fits || n > 4)) {
            std::vector<double> swv((size_t)na);
            for (int i = 0; i < na; ++i) swv[(size_t)i] = swept ? sweeps[act[(size_t)i]] : 0.0;
            int rc2 = abz_ptr_sum(s, npt, 0, npt, integrand, params, nparams, swv.data(), na, ns_eff, reinterpret_cast<double*>(host_vals));
            if (rc2 == ABZ_OK) {
                *nk_out = nk_full;
                pend->pending = false;
                return ABZ_OK;
            }
            if (rc2 != ABZ_ERR_UNSUPPORTED) return rc2;
        }
        abz_rule* r = nullptr;
        bool owned = false;
        int rc2 = series_rule(s, npt, syms, nsyms, want, keep, &r, &owned);
        if (rc2) return rc2;
        *nk_out = r->nk;
        if (swept && !sweeps_current) {
            for (int i = 0; i < na; ++i) sw_h[i] = sweeps[act[(size_t)i]];
            hipError_t he = sw_mapped ? hipSuccess
                            : mb      ? hipMemcpyAsync(sw_d, sw_h, sizeof(double) * (size_t)na, hipMemcpyHostToDevice, ctx->stream)
                                      : hipMemcpy(sw_d, sw_h, sizeof(double) * (size_t)na, hipMemcpyHostToDevice);
            if (he != hipSuccess) {
                set_error("abz_autoptr_solve_many: upload of the swept values failed: %s", hipGetErrorString(he));
                rc2 = ABZ_ERR_HIP;
            }
            sweeps_current = true;
        }
        double* const out_d = reinterpret_cast<double*>(io + sw_bytes + (size_t)slot * out_bytes);
        double2* const map = mapped ? reinterpret_cast<double2*>(out_map + (size_t)slot * out_bytes) : nullptr;
        if (!rc2) rc2 = rule_reduce(r, integrand, params, nparams, sw_d, swept ? na : 1, ns_eff, out_d, true, map);
        if (owned) {  // a rule for this grid only: its blocks go back to the allocator once the stream has drained
            (void)hipStreamSynchronize(ctx->stream);
            rule_free(r);
        }
        if (rc2) return rc2;
        pend->pending = true;
        return ABZ_OK;
    };
    // sums of the pending slots to the host: one synchronisation (+ one copy when they were left in HBM)
    auto fetch = [&](int first_slot, int nslots) -> int {
        if (!mapped)
            ABZ_HIP(hipMemcpyAsync(out_h + (size_t)first_slot * out_bytes, io + sw_bytes + (size_t)first_slot * out_bytes,
                                   (size_t)nslots * out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        ABZ_HIP(hipStreamSynchronize(ctx->stream));
        return ABZ_OK;
    };
    auto host_out = [&](int slot) { return reinterpret_cast<const double2*>(out_h + (size_t)slot * out_bytes); };
    auto norm_of = [&](const double2* v) {
        double acc = 0.0;
        for (size_t c = 0; c < ncs; ++c) acc += v[c].x * v[c].x + v[c].y * v[c].y;
        return ncs == 1 ? std::hypot(v[0].x, v[0].y) : std::sqrt(acc);
    };

    // ---- the first two grids, in flight together when both are rule-based
    int npt = n0;
    Pending p0, p1;
    int64_t nk0 = 0, nk1 = 0;
    std::vector<double2> hv0((size_t)n_sweep * ncs), hv1((size_t)n_sweep * ncs);
    if ((rc = grid_value(npt, 0, 0, active, hv0.data(), &nk0, &p0))) return rc;
    npt += dn;
    if ((rc = grid_value(npt, 1, 1, active, hv1.data(), &nk1, &p1))) return rc;
    // A solve like the last one (same grid sequence, same integrand) that went BEYOND two grids, with those grids' rules
    // kept: their scans are launched now as well, so the whole solve waits for the device once (a cached three-grid solve
    // of config 3: 0.126 -> 0.10 ms).  If the solve stops earlier the extra sums are dropped and the hint shrinks.
    // (the key covers what decides how far a solve goes: the grid sequence, the integrand and its parameters, the symmetry
    // set, the tolerances -- a looser or differently parametrised call does not inherit the extra scans of a long one)
    uint64_t hint_key = ((uint64_t)(uint32_t)n0 << 40) ^ ((uint64_t)(uint32_t)dn << 20) ^ ((uint64_t)(uint32_t)integrand << 8) ^ (uint64_t)(uint32_t)ns_eff ^
                        ((uint64_t)(uint32_t)n_sweep << 52);
    {
        auto mix = [&](uint64_t v) { hint_key = (hint_key ^ v) * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull; };
        auto mixd = [&](double v) {
            uint64_t b;
            std::memcpy(&b, &v, sizeof b);
            mix(b);
        };
        mixd(atol);
        mixd(rtol);
        for (int i = 0; i < nparams && i < 4; ++i) mixd(params ? params[i] : 0.0);
        for (size_t i = 0; syms && i < (size_t)nsyms * d * d; ++i) mix((uint64_t)(uint32_t)syms[i]);
    }
    int nspec = 0;  // grids 2 ... 1 + nspec are pending in slots 2 ...
    int64_t nk_spec[NSLOT] = {0, 0, 0, 0};
    if (p0.pending && p1.pending && s->auto_hint_key == hint_key && s->auto_hint_grids > 2 && n_sweep <= 8) {
        const int want_g = std::min(s->auto_hint_grids, NSLOT);
        for (int g = 2; g < want_g; ++g) {
            const int np2 = n0 + g * dn;
            bool have = false;
            for (const SeriesRule& k : s->kept_rules)
                have = have || (k.npt == np2 && (k.want & want) == want && k.syms.size() == (syms ? (size_t)nsyms * d * d : 0) &&
                                std::equal(k.syms.begin(), k.syms.end(), syms));
            if (!have) break;
            Pending pg;
            if ((rc = grid_value(np2, g, g, active, nullptr, &nk_spec[g], &pg))) return rc;
            if (!pg.pending) return ABZ_ERR_UNSUPPORTED;  // (a kept rule is always scanned)
            nspec += 1;
        }
    }
    if (p0.pending || p1.pending) {
        if ((rc = fetch(p0.pending ? 0 : 1, (p0.pending && p1.pending) ? 2 + nspec : 1))) return rc;
        if (p0.pending) std::memcpy(hv0.data(), host_out(0), out_bytes);
        if (p1.pending) std::memcpy(hv1.data(), host_out(1), out_bytes);
    }
    for (int i = 0; i < n_sweep; ++i) {
        for (size_t c = 0; c < ncs; ++c) {
            const double2 a = hv0[(size_t)i * ncs + c], b = hv1[(size_t)i * ncs + c];
            I1[(size_t)i * ncs + c] = make_double2(a.x * value_factor, a.y * value_factor);
            I2[(size_t)i * ncs + c] = make_double2(b.x * value_factor, b.y * value_factor);
        }
        nev[(size_t)i] = nk0 + nk1;
    }
    int gindex = 1;
    while (true) {
        std::vector<int> next;
        for (int i : active) {
            double e2 = 0.0;
            double2 dv0 = make_double2(0.0, 0.0);
            for (size_t c = 0; c < ncs; ++c) {
                const double dx = I2[(size_t)i * ncs + c].x - I1[(size_t)i * ncs + c].x;
                const double dy = I2[(size_t)i * ncs + c].y - I1[(size_t)i * ncs + c].y;
                if (c == 0) dv0 = make_double2(dx, dy);
                e2 += dx * dx + dy * dy;
            }
            const double err = ncs == 1 ? std::hypot(dv0.x, dv0.y) : std::sqrt(e2);
            const bool done = err <= std::max(atol, rtol * norm_of(&I2[(size_t)i * ncs])) || nev[(size_t)i] >= maxevals || !std::isfinite(err);
            if (done) {
                std::memcpy(out_reim + 2 * (size_t)i * ncs, &I2[(size_t)i * ncs], sizeof(double2) * ncs);
                if (err_out) err_out[i] = err;
                if (numevals_out) numevals_out[i] = nev[(size_t)i];
                if (npt_out) npt_out[i] = npt;
            } else {
                next.push_back(i);
            }
        }
        active.swap(next);
        if (active.empty()) break;
        npt += dn;
        gindex += 1;
        Pending pp;
        int64_t nk = 0;
        if (gindex < 2 + nspec) {  // launched ahead, for every solve of the call: pick the active ones
            const char* pre = reinterpret_cast<const char*>(host_out(gindex));  // (8-byte aligned only: copied, not dereferenced)
            for (size_t a = 0; a < active.size(); ++a)
                std::memcpy(&vals[a * ncs], pre + sizeof(double2) * (size_t)active[a] * ncs, sizeof(double2) * ncs);
            nk = nk_spec[gindex];
        } else {
            sweeps_current = false;  // the active set shrank (or the staging was reused)
            if ((rc = grid_value(npt, gindex, 0, active, vals.data(), &nk, &pp))) return rc;
            if (pp.pending) {
                if ((rc = fetch(0, 1))) return rc;
                std::memcpy(vals.data(), host_out(0), sizeof(double2) * active.size() * ncs);
            }
        }
        for (size_t a = 0; a < active.size(); ++a) {
            const int i = active[a];
            for (size_t c = 0; c < ncs; ++c) {
                I1[(size_t)i * ncs + c] = I2[(size_t)i * ncs + c];
                I2[(size_t)i * ncs + c] = make_double2(vals[a * ncs + c].x * value_factor, vals[a * ncs + c].y * value_factor);
            }
            nev[(size_t)i] += nk;
        }
    }
    s->auto_hint_key = hint_key;
    s->auto_hint_grids = gindex + 1;
    return ABZ_OK;
} ABZ_CATCH_ALL

int abz_autoptr_solve(abz_series* s, const int32_t* syms, int nsyms, int integrand, const double* params, int nparams, double sweep,
                      int n0, int dn, double abstol, double reltol, int64_t maxevals, int keepmost, double value_factor,
                      double* out_reim, double* err_out, int64_t* numevals_out, int32_t* npt_out) try {
    return abz_autoptr_solve_many(s, syms, nsyms, integrand, params, nparams, &sweep, 1, n0, dn, abstol, reltol, maxevals, keepmost,
                                  value_factor, out_reim, err_out, numevals_out, npt_out);
} ABZ_CATCH_ALL

int abz_rule_ggr(abz_rule* r, const double* E, int nE, double* out) try {
    int rc0 = check_rule(r);
    if (rc0) return rc0;
    ABZ_REQUIRE(E && out && nE >= 1, "abz_rule_ggr: bad arguments");
    ABZ_REQUIRE(r->V.base && r->E.base, "GGR needs a rule built with ABZ_WANT_VEL");
    abz_ctx* ctx = r->s->ctx;
    ABZ_HIP(hipSetDevice(ctx->device));
    return launch_ggr(ctx, r->s->n, r->s->d, r->npt, r->E, r->V, r->w, r->nk, E, nE, out);
} ABZ_CATCH_ALL

// ---------------------------------------------------------------- arbitrary nodes
int abz_eval_nodes(abz_series* s, const double* k, int64_t nk, int want, double* H_out, double* eig_out) try {
    int rc = check_series(s);
    if (rc) return rc;
    ABZ_REQUIRE(k || nk == 0, "null nodes");
    ABZ_REQUIRE(nk >= 0, "negative node count");
    ABZ_REQUIRE(!(want & ABZ_WANT_VEL), "abz_eval_nodes: velocities are only available through PTR rules");
    ABZ_REQUIRE(!(want & ABZ_WANT_H) || H_out, "want H but H_out is null");
    ABZ_REQUIRE(!(want & ABZ_WANT_EIG) || eig_out, "want eigenvalues but eig_out is null");
    if (nk == 0) return ABZ_OK;
    abz_ctx* ctx = s->ctx;
    ABZ_HIP(hipSetDevice(ctx->device));
    const int d = s->d, n = s->n;
    // bound the contracted-set pools: chunk so the largest level stays under ~1 GiB
    int64_t per = s->elems(d > 1 ? d - 1 : 1) * (int64_t)sizeof(double2);
    int64_t chunk = std::max<int64_t>(256, std::min<int64_t>(nk, ((int64_t)1 << 30) / std::max<int64_t>(per, 1)));
    DevBuf Hd, Ed;
    PlanDev pd;
    int status = ABZ_OK;
    for (int64_t k0 = 0; k0 < nk && status == ABZ_OK; k0 += chunk) {
        const int64_t m = std::min(chunk, nk - k0);
        const int64_t ntl = (m + 63) / 64;
        PlaneView Hv, Ev;
        Plan plan;
        plan_runs<double>(plan, d, 0, k + k0 * d, m, true);
        if ((status = plan_upload(ctx, plan, pd))) break;
        const double2* level1 = nullptr;
        if ((status = build_chain(s, plan, pd, nullptr, 0, &level1))) break;
        if (want & ABZ_WANT_H) {
            if ((status = Hd.reserve(sizeof(double) * (size_t)ntl * 64 * 2 * n * n))) break;
            Hv.base = Hd.as<double>();
            Hv.pitch = 64;
            Hv.row = 64;
            Hv.line_len = 64;
            Hv.tile = (int64_t)2 * n * n * 64;
        }
        if (want & ABZ_WANT_EIG) {
            if ((status = Ed.reserve(sizeof(double) * (size_t)ntl * 64 * n))) break;
            Ev.base = Ed.as<double>();
            Ev.pitch = 64;
            Ev.row = 64;
            Ev.line_len = 64;
            Ev.tile = (int64_t)n * 64;
        }
        EvalSpec es;
        es.n = n;
        es.M = s->dims[0];
        es.first = s->first[0];
        es.period = s->period[0];
        es.src = level1;
        es.grid = false;
        es.npt = 0;
        es.nlines = 0;
        es.tab = nullptr;
        es.nk = m;
        es.parents = pd.parent[0].as<int64_t>();
        es.gi = nullptr;
        es.x = pd.xs[0].as<double>();
        es.deriv = false;
        es.herm = false;
        es.H = Hv;
        es.E = Ev;
        es.U = PlaneView();
        if ((status = launch_eval(ctx, es))) break;
        if (want & ABZ_WANT_H)
            if ((status = export_planes(ctx, Hv, 2 * n * n, m, H_out + k0 * 2 * n * n, (want & ABZ_WANT_H_ROW_MAJOR) ? n : 0))) break;
        if (want & ABZ_WANT_EIG)
            if ((status = export_planes(ctx, Ev, n, m, eig_out + k0 * n))) break;
    }
    Hd.release();
    Ed.release();
    pd.release();
    return status;
} ABZ_CATCH_ALL

}  // extern "C"
