// Internal declarations shared by the HIP translation units of libabzhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/abzhip.h"

namespace abz {

void set_error(const char* fmt, ...);
int catch_status() noexcept;  // status + abz_last_error message of the exception in flight

// Every extern "C" entry point is a function-try-block closed by this: std::bad_alloc -> ABZ_ERR_NOMEM, anything else ->
// ABZ_ERR_INTERNAL, the text in abz_last_error().
#define ABZ_CATCH_ALL \
    catch (...) { return abz::catch_status(); }

#define ABZ_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            abz::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                      \
            return ABZ_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define ABZ_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            abz::set_error(__VA_ARGS__);  \
            return ABZ_ERR_ARG;           \
        }                                 \
    } while (0)

// ---- environment switches -------------------------------------------------------------------------------------------
// Every switch the library reads, in one table (api.cpp: name, default, meaning; DESIGN.md section 11 repeats it).  They
// exist for two reasons only: a test compares two code paths that must agree (the switch selects the one that is not the
// default), or an operator sizes a resource.  Values are read per call -- tests flip them at run time.
enum Switch {
    SW_POOL_MB,          // ABZ_POOL_MB         cap of the caching device allocator
    SW_DEBUG_TIMING,     // ABZ_DEBUG_TIMING    wall time of the rule-build phases on stderr
    SW_EVAL_PACKED,      // ABZ_EVAL_PACKED     full-grid chains of Hermitian series on packed sets (packed_herm.h)
    SW_DOS3_SCAN,        // ABZ_DOS3_SCAN       3-band DOS sweeps through dos3_scan_kernel (0: the generic reduce_kernel)
    SW_REDUCE_ROWS,      // ABZ_REDUCE_ROWS     rows of blocks a sweep is split over (0: chosen by the launch)
    SW_ADAPT_PAIR,       // ABZ_ADAPT_PAIR      two-lane adaptive step of the device-side GK loops (0: one lane)
    SW_GEN_SUM_TRI,      // ABZ_GEN_SUM_TRI     5...16-band sweeps by tridiagonal resolvents (0: one inversion per value)
    SW_IPANEL_FOLD,      // ABZ_IPANEL_FOLD     16-lane panel kernel: folded series
    SW_IPANEL_FMAC,      // ABZ_IPANEL_FMAC     16-lane panel kernel: pivot broadcast inside v_fmac_f64_dpp
    SW_GGR_FUSED,        // ABZ_GGR_FUSED       one-kernel GGR build (0: eigenvectors + one velocity launch per variable)
    SW_GGR_FUSE2,        // ABZ_GGR_FUSE2       ... contracting variable 2 inside the kernel (0: level-1 families)
    SW_GGR_UNIFORM,      // ABZ_GGR_UNIFORM     GGR scans index equispaced energy lists by arithmetic
    SW_IAI_SPECULATE,    // ABZ_IAI_SPECULATE   requests ahead of the pops in the nested IAI driver
    SW_IAI_PACKED,       // ABZ_IAI_PACKED      IAI chains of Hermitian series (n <= 4) on packed rows
    SW_IAI_POOL_MB,      // ABZ_IAI_POOL_MB     size of a chunk of level sets in the IAI driver
    SW_IAI_DEVICE_INNER, // ABZ_IAI_DEVICE_INNER innermost adaptive loops on the device (0: host-driven rounds)
    SW_IAI_PANELS,       // ABZ_IAI_PANELS      level above the innermost: panels, not nodes, cross PCIe (0: nodes)
    SW_IAI_STATS,        // ABZ_IAI_STATS       per-solve statistics of the IAI driver on stderr
    SW_HOST_THREADS,     // ABZ_HOST_THREADS    host threads for the per-integral bookkeeping of IAI sweeps
    SW_AUTO_SWEEP_MAPPED,  // ABZ_AUTO_SWEEP_MAPPED  AutoPTR solves of <= 8 values: swept values read from pinned host memory
    SW_LANE_KERNELS,     // ABZ_LANE_KERNELS    5...8 bands on full grids: one node per lane (kernels_lane.hip) instead of the 8-lane row kernels
    SW_BIG_MFMA,         // ABZ_BIG_MFMA        33...64 bands: level-1 evaluation of grid lines as a real GEMM on v_mfma_f64_16x16x4_f64
    SW_BIG_CHUNK_MB,     // ABZ_BIG_CHUNK_MB    33...64 bands: scratch for the matrices of a chunk of nodes (0: 256 MB, GGR builds 2 GB)
    SW_BIG_TRI_WAVES,    // ABZ_BIG_TRI_WAVES   33...64 bands: waves per node of the Householder tridiagonalisation (1, 2 or 4)
    SW_EIG_FOLD,         // ABZ_EIG_FOLD        5...16-band rule builds of Hermitian series: folded level-1 series
    SW_EIG_SPLIT,        // ABZ_EIG_SPLIT       5...16-band eigenvalue builds: tridiagonal eigenvalues in a kernel of their own
    SW_IAI_LANES,        // ABZ_IAI_LANES       lanes (host thread + stream each) an IAI sweep is split over
    SW_IAI_LANE_MIN,     // ABZ_IAI_LANE_MIN    solves a lane needs before a sweep is split further
    SW_COUNT
};
int abz_switch(Switch s);  // the switch's integer value from the environment, or its default

// Device buffer that grows but never shrinks (scratch); freed with its owner.
// Host <-> device copies of arrays beyond a few hundred KB go through the context's pinned staging buffer:
// ROCm pins pageable memory on the fly for such copies, which costs milliseconds per call (5 MB of plan
// arrays took 15-27 ms; staged: < 2 ms).  Both calls return when the data has arrived.
int stage_h2d(abz_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int stage_d2h(abz_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
// the context's pinned, device-visible mailbox (ctx->mbox / mbox_dev / mbox_cap): small inputs and results of a call
int mbox_reserve(abz_ctx* ctx);
struct SymTables;
int sym_tables_device(abz_ctx* ctx, int npt, int d, const int32_t* syms, int nsyms, SymTables& out);
void preload_symptr_code();  // kernels_symptr.hip: force the lazy code-object load

// caching device allocator (api.cpp): blocks freed with dev_free are reused by later dev_alloc calls
int dev_alloc(void** out, size_t bytes, size_t* cap_out);
void dev_free(void* p, size_t cap);

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool view = false;  // p points into a block owned elsewhere (release only forgets it)
    int reserve(size_t bytes);
    void release();
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

// Tiled planar addressing of rule values: nodes are grouped in tiles of `line_len` consecutive nodes
// (a grid line, or 64 nodes of an irregular list); a tile stores all its planes back to back, each
// plane row `pitch` doubles long (pitch >= line_len, multiple of 16 = 128 B).  Element (node k,
// plane p) = base[(k / line_len) * tile + p * pitch + k % line_len].  One wave writes one tile: a
// single contiguous ~27 KB HBM region instead of 21 streams 27 MB apart (1.7x the write bandwidth,
// tools/micro/wtest.hip).
// The same formula carries the PADDED PLANAR layout of full-grid rules (ABZ_RULE_PLANAR): tile = row (the stride from one
// grid line to the next inside a plane), pitch = nlines * row (the stride from one plane to the next), so every plane is one
// dense array of padded rows and all resident waves write it as one front (tools/micro/placement.hip).  `row` is the padded
// row length either way (columns line_len .. row-1 of a row are padding).
struct PlaneView {
    double* base = nullptr;  // first plane of this array inside tile 0
    int64_t tile = 0;        // doubles from one line (tile) to the next
    int pitch = 0;           // doubles from one plane to the next
    int line_len = 1;
    int row = 0;             // padded row length (multiple of 16 doubles = 128 B)
    int compact = 0;         // H planes of a Hermitian rule stored as the upper triangle only (ABZ_WANT_H_COMPACT): = n
};

// Device-resident description of the nodes of a symmetric (irreducible-node) rule: grid indices, weights and the
// contraction plan (kernels_symptr.hip).  Integer tables: they do not depend on the series and are cached per context.
struct SymTables {
    int npt = 0, d = 0;
    int64_t nk = 0;
    int64_t nitems[ABZ_MAX_DIM + 1] = {0, 0, 0, 0};
    DevBuf arena;                    // one allocation; the tables below are views into it
    size_t arena_bytes = 0;          // its used length (a rule takes its own copy of the tables with ONE device copy)
    int32_t* idx = nullptr;          // [d][nk]
    double* w = nullptr;             // [nk]
    int32_t* gi[ABZ_MAX_DIM + 1] = {nullptr, nullptr, nullptr, nullptr};      // [0] i_1 of the nodes, [L] i_{L+1} of the level-L items
    int64_t* parent[ABZ_MAX_DIM + 1] = {nullptr, nullptr, nullptr, nullptr};  // [0] level-1 item of a node, [L] level-(L+1) item of a level-L item
    int64_t* runs = nullptr;         // [nitems[1] + 1]: first node of every level-1 item (d >= 2)
    std::vector<int32_t> syms;       // the key, with npt and d
    void release();
};

struct ProfSlot {
    double ms = 0.0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace abz

// Handle lifetimes: a series keeps its context alive and a rule keeps its series alive (reference counts),
// so the destroy calls may come in any order (finalizers of a garbage-collected host language do).  A destroy
// call marks the handle closed -- further API calls on it fail with ABZ_ERR_ARG while dependants still hold the
// object -- and the memory goes when the last dependant is destroyed.
struct abz_ctx {
    std::atomic<int> refs{1};  // finalizers of a host language may release from another thread than the one that creates
    bool closed = false;
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;  // false: borrowed from the caller (abz_ctx_create_on_stream)
    unsigned prof = 0;  // bit k set: record HIP events around launches of kernel id k
    abz::ProfSlot prof_slots[ABZ_K_COUNT];
    std::vector<hipEvent_t> event_pool;
    abz::DevBuf scratch[8];  // phases, partials, staging...
    void* pin = nullptr;     // pinned host staging buffer (hipHostMalloc), grown on demand
    size_t pin_cap = 0;
    // small pinned, device-visible mailbox: swept values go in through it without a synchronising pageable copy, and the
    // last kernel of a reduction writes its few sums straight into it (zero copy) -- one stream synchronisation per call
    std::vector<abz::SymTables*> sym_cache;  // most recently used last
    // phase tables e^{2 pi i j / npt} already on the device, by npt (they depend on nothing else): a rule build copies
    // its table device to device instead of 2 npt long-double sincos + a synchronising upload (0.03-0.07 ms per build)
    std::vector<std::pair<int, abz::DevBuf>> phase_cache;
    void* mbox = nullptr;
    void* mbox_dev = nullptr;  // the same memory as the device sees it
    size_t mbox_cap = 0;
};

namespace abz {
// a rule kept by its series for the whole-solve entry points (abz_autoptr_solve*): owned by the series (it holds no
// reference on it), refreshed when the coefficients change
struct SeriesRule {
    int npt = 0, want = 0;
    std::vector<int32_t> syms;  // [nsyms][d][d], empty: full grid
    abz_rule* r = nullptr;
    uint64_t generation = 0;    // the series' generation its values were filled at
    uint64_t stamp = 0;         // last use
};
}  // namespace abz

struct abz_series {
    std::atomic<int> refs{1};
    bool closed = false;
    abz_ctx* ctx = nullptr;
    int d = 0, n = 0;
    int dims[ABZ_MAX_DIM] = {1, 1, 1};
    int first[ABZ_MAX_DIM] = {0, 0, 0};
    double period[ABZ_MAX_DIM] = {1, 1, 1};
    bool hermitian = false;   // c(-R) == c(R)^dagger exactly  =>  H(k) Hermitian: half the Fourier work
    size_t coef_cap = 0;
    abz::DevBuf coef_pk;       // Hermitian series, n <= 4: the coefficients with the innermost variable packed (packed_herm.h)
    bool coef_pk_valid = false;
    double2* coef = nullptr;  // level d: [M_d]...[M_1][n*n] complex, i_1 fastest (Julia order)
    bool coef_borrowed = false;       // a lane view: `coef` is its parent's block
    std::vector<abz_series*> lanes;   // views of this series on contexts (streams) of their own: the sweep lanes of abz_iai_solve_many
    // pools of contracted coefficient sets: level j (1 <= j < d) holds (j)-dim series of
    // elems(j) = M_1*...*M_j*n*n complex numbers per slot.
    abz::DevBuf pool[ABZ_MAX_DIM + 1];      // rule builds / abz_eval_nodes
    abz::DevBuf iai_pool[ABZ_MAX_DIM + 1];  // IAI: contracted sets per level, slot-addressed
    int64_t iai_used[ABZ_MAX_DIM + 1] = {0, 0, 0, 0};
    abz::DevBuf iai_io[6];                  // parents / x / tail / values / phases staging
    abz_exchange_fn ex_fn = nullptr;  // a single IAI solve sharded over ranks: all-gather hook (abz_iai_set_exchange)
    void* ex_user = nullptr;
    int ex_rank = 0, ex_world = 1;
    void* iai_pin[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // pinned host blocks of the IAI driver: chunk inputs [0,1] / outputs [2,3] / exchange [4]
    size_t iai_pin_cap[5] = {0, 0, 0, 0, 0};
    void* iai_pin_dev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // device-visible addresses of the same blocks (zero-copy)
    uint64_t generation = 0;                  // bumped by abz_series_update
    std::vector<abz::SeriesRule> kept_rules;  // rules of abz_autoptr_solve*, most recently used last
    uint64_t kept_stamp = 0;
    std::vector<int> summed_once;             // full grids beyond `keepmost` that were summed on the fly once already
    uint64_t auto_hint_key = 0;               // the last abz_autoptr_solve* call's sequence / integrand ...
    int auto_hint_grids = 0;                  // ... and how many grids it needed (scans launched ahead next time)
    abz::DevBuf auto_io;                      // swept values in / sums out of abz_autoptr_solve*
    void* auto_pin = nullptr;                 // ... and their pinned, device-visible host block (zero-copy results)
    void* auto_pin_dev = nullptr;
    size_t auto_pin_cap = 0;
    int64_t elems(int level) const {
        int64_t e = (int64_t)n * n;
        for (int j = 0; j < level; ++j) e *= dims[j];
        return e;
    }
};

namespace abz {
// at least `count` lane views of s (same device coefficients, own context / stream / pools); owned by s
int series_lane_views(abz_series* s, int count);
}  // namespace abz

struct abz_rule {
    abz_series* s = nullptr;
    int npt = 0, want = 0;
    int64_t nk = 0;       // number of nodes
    int64_t ntiles = 0;   // tiles (grid lines, or 64-node groups of an irregular list)
    bool full = true;     // full grid (implicit nodes/weights) or explicit irregular list
    bool herm = false;    // values come from a Hermitian series (set by every fill): H(k) = H(k)^dagger
    int64_t k_offset = 0;  // full grids: flat grid index of node 0 (non-zero for a slab of the outermost variable)
    double* vals = nullptr;  // [ntiles][planes][pitch]: H planes 2*(a + n*b) + {re, im}, then E (n), then V (d*n)
    size_t vals_cap = 0, w_cap = 0, idx_cap = 0;  // block sizes as handed out by dev_alloc
    int planes = 0;
    abz::PlaneView H, E, V;  // views into vals (base == nullptr when absent)
    double* w = nullptr;   // [nk] weights (symmetric rules)
    int32_t* idx = nullptr;  // [d][nk] grid indices (symmetric rules)
    void* plan = nullptr;    // abz::RulePlan (api.cpp): contraction plan + phase table, device resident
    bool tables_view = false;  // w and idx point into the plan's copy of the symmetric-rule tables
};

namespace abz {

// RAII-ish profiling bracket around launches of one logical kernel.
struct ProfScope {
    abz_ctx* ctx;
    int id;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(abz_ctx* c, int kernel_id);
    ~ProfScope();
};
int prof_collect(abz_ctx* ctx);

// ---- launchers (kernels.hip) ------------------------------------------------------------
// phase table tab[i] = (cos, sin)(2 pi i / npt), computed on the host in long double
int make_phase_table(abz_ctx* ctx, int npt, DevBuf& buf);

struct PhaseSpec {
    // phs[b][m] for b < B, m < M.  Grid mode (x == nullptr): grid index gi[b] (or b % npt when
    // gi == nullptr), phase = tab[(freq * gi) mod npt].  Node mode: phase = exp(2 pi i freq x[b] / period).
    int64_t B;
    int M;
    int first;
    const int32_t* gi;
    const double* x;
    const double2* tab;
    int npt;
    int g0 = 0, gcnt = 0;  // gi == nullptr: grid index of item b is g0 + b % gcnt (gcnt = 0: npt)
    double period;
    bool deriv;  // multiply by i * 2 pi * freq  (d/dx_j * period_j, src/dos_ggr.jl:20,35)
};
int launch_phases(abz_ctx* ctx, const PhaseSpec& ps, double2* phs);

// out[b][l] = sum_m phs[b][m] * src[parent(b)][m*L + l];  parent(b) = parents ? parents[b] : b / per_parent
int launch_contract(abz_ctx* ctx, const double2* src, int64_t src_slot_elems, const int64_t* parents,
                    int64_t per_parent, const double2* phs, double2* out, int64_t B, int64_t L, int M);

// full-grid contraction: out[(parent*npt + gi)][l], parents 0..nparents-1, all gi, phases from tab
constexpr int ABZ_CONTRACT_GRID_MAXM = 16;
int launch_contract_grid(abz_ctx* ctx, const double2* src, int64_t src_slot_elems, int64_t nparents, const double2* tab,
                         double2* out, int64_t L, int M, int first, int npt, bool deriv, int gbeg, int gcnt,
                         const double2* phs_table = nullptr);  // phs_table [npt][M]: scalar-phase kernel (no derivative)

struct EvalSpec {
    int n;              // bands
    int M, first;       // innermost dim
    double period;
    const double2* src;  // level-1 coefficient sets, slot stride M*n*n
    // grid mode (full PTR grid): nodes k = line*npt + i1, slot = line
    bool grid;
    int npt;
    int64_t nlines;
    const double2* tab;
    // node mode: explicit parents (slot per node), and either grid index gi or coordinate x
    int64_t nk;
    const int64_t* run_start = nullptr;  // device [nruns + 1]: nodes of level-1 set i are [run_start[i], run_start[i + 1]) (grid-index lists)
    int64_t nruns = 0;
    const int64_t* parents;
    const int32_t* gi;
    const double* x;
    bool deriv;
    bool herm;  // series is Hermitian-symmetric and deriv is false: evaluate the upper triangle only
    bool packed = false;  // grid mode: `src` holds PACKED Hermitian level-1 sets (packed_herm.h), Pk<n>::size((M - 1) / 2) numbers per line
    // outputs (tiled planar views), base == nullptr when not wanted
    PlaneView H;
    PlaneView E;
    PlaneView U;  // eigenvector planes 2*(a + n*b) + {re, im}: component a of vector b
};
int launch_eval(abz_ctx* ctx, const EvalSpec& es);
bool eval_packed_supported(int n, int M, int npt);
// rows [nrows][M n n] of full coefficients (innermost variable fastest) -> packed rows [nrows][P] (packed_herm.h)
int launch_pack_rows(abz_ctx* ctx, int n, int M, const double2* src, int64_t nrows, double2* out);
size_t packed_row_elems(int n, int M);
int series_ensure_packed(abz_series* s);  // api.cpp: the packed copy of the coefficients (coef_pk) is current

int launch_eig_planes(abz_ctx* ctx, int n, PlaneView H, PlaneView E, PlaneView U, int64_t nk);
// V[b](k) = Re sum_{a,c} conj(U[a,b]) dH[a,c] U[c,b]   (Vj: view of the n planes of one direction)
int launch_velocity(abz_ctx* ctx, int n, PlaneView U, PlaneView dH, PlaneView Vj, int64_t nk);

struct ReduceSpec {
    int n, d, npt;
    int integrand;
    PlaneView H;
    PlaneView E;
    int64_t nk;
    const double* w;      // null: uniform weight 1
    const int32_t* idx;   // null: full grid (k -> grid indices implicitly)
    int64_t k_offset = 0; // full grid: flat index of node 0
    bool herm = false;    // the cached H(k) are exactly Hermitian (rule of a Hermitian series)
    double params[4];
    const double* sweep_dev;  // device [n_sweep]
    int n_sweep;
    double scale;
    double* tri_cache = nullptr;  // 33...64 bands: the rule's room for the tridiagonal forms of its nodes [2 x 64][tri_nk] ...
    int64_t tri_nk = 0;
    int* tri_state = nullptr;     // ... and whether it holds them (set by the scan that fills it)
    double* out_dev = nullptr;  // device [n_sweep][ncomp][2]: leave the result in HBM, no host synchronisation
    double2* out_map_dev = nullptr;         // host-io calls: device view of the pinned mailbox region the sums are written to ...
    const double2* out_map_host = nullptr;  // ... and its host view (read after the stream synchronisation; null with
                                            // out_map_dev set: the launch returns without synchronising, n <= 4)
};
int integrand_ncomp(int integrand, int n, int d);
// result: host out_reim [n_sweep][ncomp][2]
int launch_reduce(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim);

int launch_gen_velocity(abz_ctx* ctx, int n, PlaneView U, PlaneView dH, PlaneView Vj, int64_t nk);

// Fused GGR build (kernels_ggr.hip): eigenvalues + band velocities of every node in one kernel, Hermitian series,
// n <= 4.  ref: src/dos_ggr.jl:14-44.
struct GgrBuildSpec {
    int n, d;
    int M, first;   // variable 1
    int npt;
    const double2* tab;
    PlaneView E, V;  // V: d*n planes, plane j*n + b = velocity of band b along variable j+1
    // grid mode
    bool grid = true;
    int64_t nlines = 0;
    bool fuse = false;  // the wave contracts variable 2 itself from the level-2 sets src2 (ggr_build_can_fuse)
    const double2* src[3] = {nullptr, nullptr, nullptr};  // !fuse / node lists: level-1 families: plain, derivative on variable 2, on variable 3
    const double2* src2[2] = {nullptr, nullptr};          // fuse: level-2 sets: plain, derivative on variable 3
    double2* pack2 = nullptr;  // fuse: room for ggr_build_pack2_elems() complex numbers (the packed level-2 sets)
    int M2 = 0, first2 = 0, gbeg = 0, gcnt = 0;
    // node lists (symmetric rules)
    int64_t nk = 0;
    const int64_t* parents = nullptr;
    const int32_t* gi = nullptr;
};
bool ggr_build_supported(int n, int d, int M, int npt, bool herm);
bool ggr_build_can_fuse(int n, int d, int M, int M2, int npt);
size_t ggr_build_pack2_elems(int n, int d, int M, int M2, int64_t nparents);
int launch_ggr_build(abz_ctx* ctx, const GgrBuildSpec& gs);
// The same for 5...32 bands in the row layout (kernels_ggr_rows.hip): Householder + per-lane tridiagonal eigenvectors +
// back-transformation + quadratic forms, one kernel, only (e, v) stored.
struct GgrRowsSpec {
    int n, d;
    int M, first;  // variable 1
    int npt;
    const double2* tab;
    PlaneView E, V;
    int64_t nlines = 0;                  // level-1 sets: grid lines, or runs of a node list
    const int64_t* run_start = nullptr;  // node lists: first node of every run (+ nk at the end); nullptr: full grid lines
    const int32_t* gi = nullptr;         // node lists: grid index i_1 of every node
    const int64_t* parents = nullptr;    // node lists: the level-1 set of every node (33...64 bands, which do not walk runs)
    int64_t nk = 0;                      // node lists: number of nodes
    const double2* src[3] = {nullptr, nullptr, nullptr};  // level-1 families: plain, derivative on variable 2, on variable 3
};
bool ggr_rows_supported(int n, int d, int M, int npt, bool herm);
int launch_ggr_rows(abz_ctx* ctx, const GgrRowsSpec& gs);
// 33...64 bands (kernels_big.hip, kernels_big_vec.hip): the same build, one wave per node
bool big_ggr_supported(int n, int d, int M, int npt, bool herm);
int launch_big_ggr(abz_ctx* ctx, const GgrRowsSpec& gs);
int launch_big_vec(abz_ctx* ctx, const double* tri, int64_t tri_nk, const double2* keep, const double2* Dm, int64_t dstride, int64_t node0,
                   int64_t nnodes, int n, int d, PlaneView E, PlaneView V);
int launch_ggr(abz_ctx* ctx, int n, int d, int npt, PlaneView E, PlaneView V, const double* w, int64_t nk,
               const double* Es_host, int nE, double* out_host);

// tiled planar (ncomp planes of the view) -> AoS [nk][ncomp] on the host
int export_planes(abz_ctx* ctx, PlaneView v, int ncomp, int64_t nk, double* host_out, int row_major_n = 0);

// IAI innermost nodes: values[node][ncomp] complex
struct NodeEvalSpec {
    int n, d, M, first;
    double period;
    const double2* src;
    const int64_t* parents;  // device
    const double* x;         // device
    const double* tail;      // device [nnodes][d-1] or null
    int64_t nnodes;
    int integrand;
    double params[4];
    double sweep;
    const double* sweep_arr = nullptr;  // device [nnodes]: per-node sweep value (overrides `sweep`)
    bool panels15 = false;  // every aligned run of 15 nodes is one GK(7,15) panel (same parent)
    bool packed = false;  // `src` holds PACKED Hermitian level-1 sets (packed_herm.h), n <= 4
    bool herm = false;    // the series is Hermitian (33...64 bands: resolvent traces come from the tridiagonal of Hermitian(h))
};
int launch_node_integrand(abz_ctx* ctx, const NodeEvalSpec& ns, double2* values_dev);

// Innermost level of IAI entirely on the device: one half-wave (32 lanes) runs the globally adaptive
// GK(7,15) loop of ONE 1-D integral (QuadGK adapt, scalar refinement, DataStructures heap semantics,
// the shared gk15.h rule) -- 15 / 30 lanes evaluate the nodes of the new panels, lane 0 keeps the heap.
struct InnerSpec {
    int n, d, M, first;
    double period;
    const double2* src;      // level-1 coefficient sets
    int64_t nint;            // number of 1-D integrals
    const int64_t* slot;     // device [nint]
    const double* lo;        // device [nint]
    const double* hi;        // device [nint]
    const double* atol;      // device [nint], < 0: none
    const double* tail;      // device [nint][d-1] or null
    int integrand;
    double params[4];
    double sweep;
    const double* sweep_arr = nullptr;  // device [nint]: per-integral sweep value (overrides `sweep`)
    bool herm = false;  // the series is Hermitian: upper-triangle series, real characteristic polynomial
    bool packed = false;  // `src` holds PACKED Hermitian level-1 sets (packed_herm.h), n <= 4
    bool has_rtol;
    double rtol_user;
    int64_t maxevals;
    double2* I_out;          // device [nint][ncomp]
    double* E_out;           // device [nint]
    int64_t* nev_out;        // device [nint]
    int* status_out;         // device [nint]: 0 ok, 1 = segment store overflow (redo on the host)
};
constexpr int ABZ_INNER_MAXSEG = 48;
// workgroup-per-integral kernel (5..32 bands): the store is 44 B per segment beside a 53 KB coefficient set, so it
// can be deep -- at config 5's abstol = 1e-3 thousands of innermost integrals need more than 48 panels, and each
// of them fell back to the host-driven loop (30 of the solve's 40 s)
constexpr int ABZ_PANEL_MAXSEG = 384;
bool inner_adaptive_supported(int n, int M, int integrand);
int launch_inner_adaptive(abz_ctx* ctx, const InnerSpec& is);

// The level above the innermost one, panel by panel (iai_host.cpp: the host keeps that level's heaps and ships PANELS, not
// nodes): a panel (a, b) of a level-2 integral becomes its fifteen GK nodes -- coordinate and parent set for the contraction,
// limits / tolerance / swept value of the innermost integral beneath each node -- on the device, and the fifteen innermost
// results are folded back into the panel's (I_K s, E, evaluation count) by the shared gk15.h rule: 15 x less traffic over
// PCIe and 15 x fewer words for the host to touch per round.  Iterated limits in closed form only (CubicLimits,
// TetrahedralLimits): the arithmetic is Lims::fix / Lims::range of iai_host.cpp operation for operation.
struct PanelNodesSpec {
    int64_t npanels;
    const int64_t* p_slot;  // device-visible [npanels]: level-2 coefficient set of the panel's integral
    const double *p_a, *p_b, *p_at, *p_sw;  // device-visible [npanels]: panel limits, the integral's tolerance (< 0: none), swept value
    int lims_kind;          // ABZ_LIMS_CUBIC: innermost limits (a0, b0); ABZ_LIMS_TETRAHEDRAL: (0, a0 * (x / aL))
    double a0, b0, aL;
    int64_t* n_slot;        // device [15 npanels] out: level-1 set of the innermost integral beneath every node (= node index)
    double *n_lo, *n_hi, *n_at, *n_sw;
};
// nodes + phases + contraction of the level-L sets `src` (slot_elems numbers each, M coefficients of variable L starting at
// frequency `first`) into out[node][Lrow]
int launch_panel_contract(abz_ctx* ctx, const PanelNodesSpec& ps, const double2* src, int64_t slot_elems, int M, int first, double period,
                          double2* out, int64_t Lrow);
struct PanelRuleSpec {
    int64_t npanels;
    int ncomp;
    const double *p_a, *p_b;   // device [npanels]
    const double2* n_I;        // device [15 npanels][ncomp]: innermost integrals
    const int64_t* n_nev;      // device [15 npanels]
    const int* n_status;       // device [15 npanels]
    double2* p_I;              // device [npanels][ncomp] out
    double* p_E;               // device [npanels] out
    int64_t* p_nev;            // device [npanels] out: evaluations beneath the panel
    int* p_status;             // device [npanels] out: != 0 if any of its innermost integrals overflowed the device store
};
int launch_panel_rule(abz_ctx* ctx, const PanelRuleSpec& ps);
bool gen_inner_panel_supported(int n, int M, int integrand, bool herm);  // n > 4: one workgroup per 1-D integral, set in LDS
int launch_gen_inner_adaptive(abz_ctx* ctx, const InnerSpec& is);

// ---- generic n (5..32 bands): wave-per-node kernels (kernels_generic.hip)
struct GenSpec {
    int n, M, first, npt, d;
    double period;
    const double2* src;
    bool grid;               // node k = (line = k / npt, i1 = k % npt), slot = line
    const int64_t* parents;  // node mode
    const int64_t* run_start = nullptr;  // node mode with grid indices: runs of nodes per coefficient set (see EvalSpec)
    int64_t nruns = 0;
    const int32_t* gi;
    const double* x;
    const double2* tab;
    bool deriv;
    int64_t nnodes;
    PlaneView Hplanes;
    PlaneView Eplanes;
    PlaneView Uplanes;  // eigenvectors out (velocity builds)
    bool herm = false;  // the series is Hermitian: H(k) = H(k)^dagger to rounding
    double2* Haos;
    double* Eaos;
    int integrand;
    double params[4];
    const double* sweep_dev;
    double sweep0;
    int n_sweep;
    const double* sweep_per_node = nullptr;  // device [nnodes]: one sweep value per node (n_sweep = 1)
    double2* values;  // [node][n_sweep][ncomp] or null
    bool panels15 = false;  // nodes come as GK(7,15) panels: every aligned run of 15 shares its parent
};
int launch_gen_nodes(abz_ctx* ctx, const GenSpec& gs);

// store-free PTR sums (kernels.hip): rule(f, B) of a full grid / slab without materialising the rule
struct SumSpec {
    int n, d, M, first, npt;
    const double2* src;   // level-1 sets, one per line
    const double2* tab;
    int64_t nlines, line0;
    int integrand, n_sweep;
    const double* sweep_host;
    double params[4];
    double scale;
    bool herm = true;  // the series is Hermitian (a series that is not goes through the inverse of every node)
    bool force_inverse = false;  // n <= 4 without a closed-form store-free kernel for this case: the inverse of every node as well
};

bool eval_sum_supported(int n, int M, int npt, int integrand, bool herm);
int launch_eval_sum(abz_ctx* ctx, const SumSpec& ss, double* out_reim);
int launch_final_reduce(abz_ctx* ctx, const double2* partial, int64_t nblocks, int64_t ncols, double scale, double2* out);
// the same for 5..32 bands (kernels_generic.hip): resolvent-trace integrands, one workgroup per grid line
bool gen_sum_supported(int n, int M, int npt, int integrand, bool herm);
// 5...16 bands: can a rule of a Hermitian series keep H(k) as its upper triangle (ABZ_WANT_H_COMPACT)?  True when the row
// kernel gen_grid_eig_kernel fills it (and every scan of kernels_generic.hip reads either layout)
bool gen_compact_supported(int n, int M, int npt);
int launch_gen_sum(abz_ctx* ctx, const SumSpec& ss, double* out_reim);
// 33...64 bands (kernels_big.hip): wave-per-node Householder with the matrix in LDS, eigenvalues by bisection, resolvent
// traces from the tridiagonal
// 5...8 bands on full grids, one node per lane (kernels_lane.hip)
bool lane_grid_supported(const GenSpec& gs);
int launch_lane_grid(abz_ctx* ctx, const GenSpec& gs);
bool lane_sum_supported(int n, int M, int first, int npt, int integrand, int n_sweep);
int launch_lane_sum(abz_ctx* ctx, const SumSpec& ss, double* out_reim);
bool lane_scan_supported(const ReduceSpec& rs);
int launch_lane_scan(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim);
bool big_supported(int n);
bool big_inverse_wanted(int n, int integrand, bool herm);  // 5...64 bands: G / traces through big_inverse_kernel (scans, node values)
bool big_inverse_sum_wanted(int n, int integrand, bool herm);  // ... store-free sums
bool big_sum_supported(int n, int M, int npt, int integrand, bool herm);
int launch_big_nodes(abz_ctx* ctx, const GenSpec& gs);
int launch_big_sum(abz_ctx* ctx, const SumSpec& ss, double* out_reim);
int launch_big_reduce(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim);
int launch_gen_reduce(abz_ctx* ctx, const ReduceSpec& rs, double* out_reim);

}  // namespace abz
