/*
 * abzhip.h -- C ABI of libabzhip.so: the MI355X (gfx950) hot path behind AutoBZCore.jl's
 * integrand/algorithm dispatch boundary.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, and returns an int status
 * (0 = ok, negative = error; abz_last_error() gives the message).  No C++ exceptions cross the
 * boundary.  Host arrays are owned by the caller (Julia must GC.@preserve them for the ccall);
 * device memory lives behind opaque handles owned by the library.  Handles are NOT thread-safe:
 * use one abz_ctx (one HIP stream) per host thread, like the reference gives every thread its own
 * workspace and deep-copied solver (src/fourier.jl:60-86, src/interfaces.jl:213).  What MAY run
 * concurrently: calls on handles of different contexts; and the abz_*_destroy calls on any handle
 * beside other threads' work on the same context -- the context <- series <- rule reference counts
 * are atomic, so a finalizer thread of a garbage-collected host language may release handles while
 * another thread builds rules from the same series.  Everything else on one context is serial.
 *
 * Citations `ref:` are file:line into lxvm/AutoBZCore.jl v0.3.8 -- the reference interface each
 * entry point replaces.  The reference-side binding (Julia ccall shim) is in INTEGRATION.md and
 * julia/AutoBZCoreHIP.jl.
 *
 * Coefficient memory order is the reference's own (Julia, column-major): interleaved (re, im)
 * doubles; inside one coefficient the n x n block is column-major (row index fastest); blocks are
 * ordered with i_1 fastest ... i_d slowest, i.e. exactly `Array{SMatrix{n,n,ComplexF64},d}`
 * (aps_example/aps_example.jl:15-27).  A scalar series is n = 1.
 */
#ifndef ABZHIP_H
#define ABZHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABZ_VERSION 502

/* status codes */
#define ABZ_OK 0
#define ABZ_ERR_ARG (-1)     /* ref: ArgumentError sites src/fourier.jl:167,267,506 */
#define ABZ_ERR_HIP (-2)     /* a HIP runtime call failed */
#define ABZ_ERR_NOGPU (-3)   /* no usable gfx950 device: the product has NO CPU fallback */
#define ABZ_ERR_UNSUPPORTED (-4)
#define ABZ_ERR_NOMEM (-5)
#define ABZ_ERR_INTERNAL (-6) /* a C++ exception was caught at the boundary (none crosses it); text in abz_last_error() */

#define ABZ_MAX_DIM 3        /* BZ dimension d = 1..3 (ref tests: test/fourier.jl:10,43) */
#define ABZ_MAX_BANDS 64     /* n x n Hamiltonians up to n = 64 (33...64: kernels_big.hip, kernels_big_vec.hip) */

typedef struct abz_ctx abz_ctx;       /* device + stream + scratch + profiling */
typedef struct abz_series abz_series; /* device-resident Fourier coefficients (FourierSeries / FourierWorkspace) */
typedef struct abz_rule abz_rule;     /* device-resident cached rule values (FourierPTR / FourierMonkhorstPack / GGR data) */

/* what a rule / node evaluation materialises (bit mask) */
#define ABZ_WANT_H 1    /* series values H(k): FourierValue.s, ref src/fourier.jl:111-114 */
#define ABZ_WANT_EIG 2  /* ascending eigenvalues of Hermitian(H(k)) (upper triangle), ref src/dos_ggr.jl:19,34 */
#define ABZ_WANT_VEL 4  /* band velocities Re diag(U' dH/dk_j U) * t_j, ref src/dos_ggr.jl:20,35 (implies EIG) */
/* With ABZ_WANT_H, for a Hermitian series of n <= 16 bands: keep H(k) as its UPPER TRIANGLE only (what Hermitian(h) reads,
 * src/dos_ggr.jl:19,34) -- n^2 value planes instead of 2 n^2, e.g. 96 instead of 168 bytes per k-point for 3 bands + eig.
 * The lower triangle is the conjugate of the upper one bit for bit, so nothing is lost: abz_rule_export still returns
 * the full matrices and every built-in integrand reads the compact planes.  Ignored (full layout) for series that are
 * not Hermitian or have more than 16 bands; abz_rule_info reports the bit only when the rule really is compact.
 * Plane order inside a tile: Re H[a][b], Im H[a][b] for a < b at planes b^2 + 2a, b^2 + 2a + 1; H[b][b] at b^2 + 2b. */
#define ABZ_WANT_H_COMPACT 8
/* abz_eval_nodes only: H_out holds every matrix ROW-major (element (a, b) at 2 (a n + b) + {re, im}) instead of the reference's
 * column-major blocks -- for hosts whose arrays are row-major (the Python mirror: transposing 4 096 matrices of 32 x 32 on the
 * host took longer than evaluating them). */
#define ABZ_WANT_H_ROW_MAJOR 16

/* built-in device integrands f(FourierValue(k, H(k)), params...; sweep) -- the integrands that
 * appear in the reference's tests, docs and example (user closures cannot cross a C ABI; they get
 * H(k) batches through abz_rule_export / abz_eval_nodes instead).  Values are complex. */
#define ABZ_F_ONE 0        /* 1                                   ref test/brillouin.jl:38        ncomp 1  */
#define ABZ_F_LINEAR 1     /* a*s + b, scalar s = H[1,1]          ref test/fourier.jl:41          ncomp 1; params {a, b} */
#define ABZ_F_LINEAR_X 2   /* a*s*x .+ b                          ref test/fourier.jl:16          ncomp d; params {a, b} */
#define ABZ_F_DOS 3        /* -Im tr inv((w+i eta)I - H)/pi       ref aps_example/aps_example.jl:30  ncomp 1; params {eta}; sweep w */
#define ABZ_F_TRGLOC 4     /* tr inv((w+i eta)I - H)              ref docs/src/examples.md:21    ncomp 1; params {eta}; sweep w */
#define ABZ_F_GLOC 5       /* inv((w+i eta)I - H)  (n x n, col-major) ref docs/src/examples.md:90  ncomp n*n; params {eta}; sweep w */
#define ABZ_F_DOS_EIG 6    /* (eta/pi) sum_b 1/((w-e_b)^2+eta^2) from cached eigenvalues == ABZ_F_DOS  ncomp 1 */

/* iterated limits for IAI (ref: src/brillouin.jl:2-5,267,304) */
#define ABZ_LIMS_CUBIC 0        /* CubicLimits(a, b) */
#define ABZ_LIMS_TETRAHEDRAL 1  /* TetrahedralLimits(a): 0 <= x_1 <= ... <= x_d <= a_d (scaled) */
/* General convex irreducible zones (ext/SymmetryReduceBZExt.jl:33-58, ext/ibzlims.jl:198-289):
 * lim_a = packed polytope, lim_b[0] = number of doubles in lim_a.
 *   POLYHEDRAL (d = 3): per face [nv, x y z of its nv vertices in order around the face]; vertices shared
 *                       between faces must be bit-identical.
 *   POLYGON (d = 2):    vertices (x, y) in order around the boundary. */
#define ABZ_LIMS_POLYHEDRAL 2
#define ABZ_LIMS_POLYGON 3

/* profiled kernels (abz_prof_read) */
#define ABZ_K_CONTRACT 0   /* outer-dimension contraction (workspace_contract!)      */
#define ABZ_K_EVAL 1       /* innermost 1-D evaluation (+ fused eig)  (workspace_evaluate!) -- the Fourier-eval kernel */
#define ABZ_K_REDUCE 2     /* integrand scan + reduce over a cached rule (quadsum)   */
#define ABZ_K_GGR 3        /* GGR formula scan (sum_ggr)                             */
#define ABZ_K_EIG 4        /* stand-alone Hermitian eigensolve                        */
#define ABZ_K_GGRBUILD 5   /* fused GGR build: H, dH/dk, eig, velocities per node (get_ggr_data, ref src/dos_ggr.jl:14-44) */
#define ABZ_K_COUNT 8

/* The library's own view of its memory, for leak checks and capacity planning (no reference counterpart):
 * info[0] device bytes handed out by its allocator and not yet returned (all contexts: coefficients, contracted
 * sets, rule values, scratch), info[1] device bytes parked in its cache of freed blocks (ABZ_POOL_MB), info[2] the
 * part of info[0] that is `ctx`'s grow-only scratch, info[3] `ctx`'s pinned host staging bytes, info[4] live blocks.
 * ctx may be NULL (then info[2] = info[3] = 0). */
int abz_mem_info(abz_ctx* ctx, int64_t* info);

/* ---------------------------------------------------------------- library / context */
const char* abz_last_error(void);
int abz_version(void);
/* Number of visible HIP devices; ABZ_ERR_NOGPU (and *n = 0) if there is none. */
int abz_device_count(int* n);
/* One context per host thread / rank: binds `device`, creates its stream. */
int abz_ctx_create(int device, abz_ctx** out);
/* The same on a stream the caller owns (a `hipStream_t`, e.g. the current stream of the PyTorch harness):
 * the library's launches are then ordered with the caller's own work and with RCCL collectives enqueued on
 * that stream -- the sharded sweep needs no host synchronisation between its scan and its gather.  The
 * stream is borrowed: abz_ctx_destroy does not destroy it.  Replaces: nothing in the reference (it has no
 * device queue); the per-thread workspace rule of src/fourier.jl:60-86 still applies, one context per thread. */
int abz_ctx_create_on_stream(int device, void* hip_stream, abz_ctx** out);
int abz_ctx_destroy(abz_ctx* ctx);
int abz_ctx_sync(abz_ctx* ctx);
/* HIP-event timing of the library's own launches on the context's stream.  on = 0: off; 1: every
 * kernel id; otherwise a mask with bit (k+1) selecting ABZ_K_<k> (timing one kernel keeps the event
 * records out of the gaps between the others). */
int abz_prof_enable(abz_ctx* ctx, int on);
int abz_prof_reset(abz_ctx* ctx);
int abz_prof_read(abz_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);

/* ---------------------------------------------------------------- series
 * Replaces: FourierSeries(c; period, offset) + workspace_allocate_vec (src/fourier.jl:56-86).
 * dims[j] = M_j, first[j] = integer frequency of the first coefficient along dim j
 * (= 1 + offset_j of FourierSeriesEvaluators, or the OffsetArray's first axis value),
 * s(x) = sum_i c[i] exp(2 pi i sum_j (first_j + i_j) x_j / period_j), i_j = 0..M_j-1. */
int abz_series_create(abz_ctx* ctx, const double* coef_reim, int d, const int32_t* dims,
                      const int32_t* first, const double* period, int n, abz_series** out);
int abz_series_destroy(abz_series* s);
/* Replace the coefficients in place (same shape).  Replaces: mutating `h.c` / assigning cache.H
 * (test/dos.jl:122-129); rules built from the series become stale until abz_rule_rebuild. */
int abz_series_update(abz_series* s, const double* coef_reim);

/* ---------------------------------------------------------------- arbitrary nodes
 * Replaces: the fallback evaluator f.w(x) (src/fourier.jl:120-122) and the body of a
 * BatchIntegrand f!(y, x, p) (src/batch.jl:1-38) for FourierValue batches.
 * k is [nk][d] (x_1 first).  Outputs (host, nullable per `want`):
 *   H_out   [nk][n*n][2]  column-major blocks (reference layout)
 *   eig_out [nk][n]
 * Evaluation is hierarchical when nodes share outer coordinates; results do not depend on it. */
int abz_eval_nodes(abz_series* s, const double* k, int64_t nk, int want, double* H_out,
                   double* eig_out);

/* ---------------------------------------------------------------- PTR rules
 * Replaces: FourierPTR ctor + fourier_ptr! (src/fourier.jl:132-174), FourierMonkhorstPack ctor +
 * _fourier_symptr! (:216-277), nextrule (:315-321), and get_ggr_data (src/dos_ggr.jl:14-44).
 * Full grid:  irr_idx = wsym = NULL, nirr = 0  -> nodes are all npt^d grid points, i_1 fastest.
 * Symmetric:  irr_idx [nirr][d] 0-based grid indices in column-major order and their integer
 *             weights wsym [nirr] (from abz_symptr_rule).
 * The values stay resident in HBM (planar layout). */
int abz_ptr_rule_build(abz_series* s, int npt, int64_t nirr, const int32_t* irr_idx,
                       const int64_t* wsym, int want, abz_rule** out);

/* Symmetric rule built entirely on the device: orbit representatives, weights, the contraction plan of the node
 * list and the values, from the symmetry matrices syms [nsyms][d][d] (row-major integers, a group).  Same nodes,
 * weights and values as abz_symptr_rule + abz_ptr_rule_build, without the node list crossing PCIe twice; the integer
 * tables of a grid are cached per context (they do not depend on the series).
 * Replaces: the FourierMonkhorstPack constructor, which builds wsym / flags and fills the values in one go
 * (src/fourier.jl:265-277). */
int abz_ptr_rule_build_sym(abz_series* s, int npt, const int32_t* syms, int nsyms, int want,
                           abz_rule** out);

/* A slab of the full grid: only i_d in [outer_begin, outer_end) of the outermost variable (d >= 2),
 * i.e. one rank's share when a single solve is sharded over k (the recursion of fourier_ptr! is
 * independent per outer index, src/fourier.jl:148-164).  Weights and abz_rule_reduce's 1/npt^d
 * normalisation are those of the whole grid, so the partial sums of disjoint slabs add up to the
 * full rule value (one all-reduce).  A symmetric rule is sharded by passing a subset of
 * the irreducible nodes to abz_ptr_rule_build instead. */
int abz_ptr_rule_build_slab(abz_series* s, int npt, int outer_begin, int outer_end, int want,
                            abz_rule** out);
int abz_rule_destroy(abz_rule* r);
/* Re-evaluate every cached value of the rule from the series' current coefficients, in place and
 * without host synchronisation (launches only).  Replaces: the re-init of a stale cache,
 * solve!(::DOSCache) with isfresh (src/dos_interfaces.jl:104-109), and nextrule rebuilding a grid
 * (src/fourier.jl:315-321) when the buffers can be reused. */
int abz_rule_rebuild(abz_rule* r);
int abz_rule_info(const abz_rule* r, int64_t* nk, int* n, int* d, int* npt, int* want);
/* Copy rule contents to the host in the reference's layout (any pointer may be NULL):
 *   x [nk][d], w [nk] (1 on a full grid), H [nk][n*n][2], eig [nk][n], vel [nk][d][n]. */
int abz_rule_export(abz_rule* r, double* x, double* w, double* H, double* eig, double* vel);

/* Replaces: rule(f, B) = quadsum(AffineQuad(rule, B), f, vol/N) (src/fourier.jl:204-207,289-292)
 * for a built-in integrand, for n_sweep parameter values in one pass (batchsolve's omega sweep,
 * src/interfaces.jl:210-222, fused).  out_reim [n_sweep][ncomp][2] receives
 *   (sum_k w_k f(k, H(k); sweep_i)) / (npt^d * nsyms)        (nsyms = 1 on a full grid)
 * -- the caller applies |det B| and symmetrisation like do_solve_autobz (src/brillouin.jl:337-355). */
int abz_rule_reduce(abz_rule* r, int integrand, const double* params, int nparams,
                    const double* sweep, int n_sweep, int nsyms, double* out_reim);

/* abz_rule_reduce without the host round trip: `sweep_dev` [n_sweep] and `out_dev_reim`
 * [n_sweep][ncomp][2] are DEVICE pointers, the call only enqueues work on the context's stream and returns.
 * This is the per-rank leg of a sharded sweep / of a k-sharded solve: the partial sums stay in HBM and go
 * straight into the RCCL all_gather / all_reduce (src/interfaces.jl:210-222's fan-in). */
int abz_rule_reduce_device(abz_rule* r, int integrand, const double* params, int nparams,
                           const double* sweep_dev, int n_sweep, int nsyms, double* out_dev_reim);
/* Device address and size in bytes of the rule's value block (tiled planar layout; with
 * ABZ_WANT_H_COMPACT the H planes of a tile are the n^2 upper-triangle planes in the order given at that flag):
 * zero-copy views for a device-side harness, and the placement log of bench.py. */
int abz_rule_values_ptr(const abz_rule* r, void** base, int64_t* nbytes);

/* Whole AutoPTR solve(s) inside the library: the p-adaptive loop of AutoSymPTR.autosymptr on the library's own integrands.
 * Replaces: do_solve(::FourierIntegrand, ::Basis, p, ::AutoSymPTRJL, cacheval) (src/fourier.jl:385-389, src/algorithms.jl:
 * 418-432) with its rule family (src/fourier.jl:296-321): I1 = rule(n0), I2 = rule(n0 + dn), err = norm(I2 - I1); while
 * err > max(abstol, reltol norm(I2)) and numevals < maxevals: I1 = I2, I2 = rule(next npt).  n0 / dn are the INTEGERS of
 * MonkhorstPackRule (defaults 50 / 50).  abstol < 0 / reltol < 0 mean `nothing` (both nothing: reltol = sqrt(eps)).
 * syms [nsyms][d][d] row-major (NULL: the full grid): symmetric rules with integer weights, dvol = 1 / (npt^d nsyms)
 * (src/fourier.jl:289-292); every rule value is multiplied by `value_factor` before the error is formed -- nsyms for the
 * TrivialRep symmetrisation inside a SymmetricRule (src/brillouin.jl:127-130), 1 otherwise.  The caller applies |det B|
 * (abstol / |det B| in, I |det B| out: src/brillouin.jl:429-444).
 * Rules of the first `keepmost` grids (the reference's `keepmost` = 2) stay with the series and serve every later call
 * (they refill themselves after abz_series_update); a larger grid is summed on the fly where the store-free kernel applies
 * and few values share it, otherwise built, scanned and dropped.  The first two grids are in flight together: a converged
 * solve with cached rules costs two scan launches and ONE stream synchronisation.
 * The _many form solves for n_sweep values of the swept parameter in lock-step (batchsolve, src/interfaces.jl:210-222):
 * each solve makes exactly the decisions it would make alone; a grid is visited once for all solves still active.
 * out_reim [n_sweep][ncomp][2], err [n_sweep] (norm(I2 - I1), nullable), numevals [n_sweep] (sum of the node counts of the
 * rules used, nullable), npt_last [n_sweep] (nullable). */
int abz_autoptr_solve(abz_series* s, const int32_t* syms, int nsyms, int integrand, const double* params, int nparams,
                      double sweep, int n0, int dn, double abstol, double reltol, int64_t maxevals, int keepmost,
                      double value_factor, double* out_reim, double* err, int64_t* numevals, int32_t* npt_last);
int abz_autoptr_solve_many(abz_series* s, const int32_t* syms, int nsyms, int integrand, const double* params, int nparams,
                           const double* sweeps, int n_sweep, int n0, int dn, double abstol, double reltol,
                           int64_t maxevals, int keepmost, double value_factor, double* out_reim, double* err,
                           int64_t* numevals, int32_t* npt_last);
/* Drop the rules the series keeps for abz_autoptr_solve* (they are also dropped by abz_series_destroy). */
int abz_series_drop_rules(abz_series* s);

/* Store-free rule value: the same number as abz_rule_reduce on the full grid (or on the slab
 * [outer_begin, outer_end) of its outermost variable), computed without materialising H(k): the
 * Fourier evaluation feeds the integrand directly and only the sums leave the kernel.  For grids that are
 * used once (an AutoPTR refinement step) or do not fit in HBM (1000^3 k-points = 168 GB of rule values).
 * DOS, TRGLOC and GLOC: every series and grid up to ABZ_MAX_BANDS bands (Hermitian series: closed forms up to 4 bands on lines of
 * more than 128 points, the tridiagonal form of every node above; GLOC above 4 bands, series that are not Hermitian and short
 * lines: the inverse of every node, chunk by chunk).  Other integrands: Hermitian series of n <= 4 bands with npt > 128;
 * ABZ_ERR_UNSUPPORTED otherwise (build a rule instead).
 * Replaces: FourierPTR ctor + rule(f, B) back to back (src/fourier.jl:166-207). */
int abz_ptr_sum(abz_series* s, int npt, int outer_begin, int outer_end, int integrand,
                const double* params, int nparams, const double* sweep, int n_sweep, int nsyms,
                double* out_reim);

/* Replaces: sum_ggr / ggr_formula (src/dos_ggr.jl:58-104) for nE energies; rule must hold VEL. */
int abz_rule_ggr(abz_rule* r, const double* E, int nE, double* out);

/* Replaces: AutoSymPTR.symptr_rule as called at src/fourier.jl:271 (host, integer-exact).
 * syms [nsyms][d][d] row-major integer matrices acting on fractional coordinates.
 * First call with irr_idx = NULL to get *nirr; then with buffers irr_idx [nirr][d], wsym [nirr]. */
int abz_symptr_rule(int npt, int d, const int32_t* syms, int nsyms, int64_t* nirr,
                    int32_t* irr_idx, int64_t* wsym);
/* `syms` must be a group (closed, with the identity) as AutoSymPTR assumes: orbits are then equivalence
 * classes and "first member in column-major order" is well defined.
 * Same tables computed on the GPU (orbit kernel + order-preserving compaction): bit-identical output,
 * ~50x faster on large grids (the reference calls this step its likely bottleneck, src/fourier.jl:270). */
int abz_symptr_rule_device(abz_ctx* ctx, int npt, int d, const int32_t* syms, int nsyms,
                           int64_t* nirr, int32_t* irr_idx, int64_t* wsym);

/* ---------------------------------------------------------------- IAI building blocks + driver
 * Replaces: workspace_contract!(w, x) on a batch of nodes (src/fourier.jl:468,478):
 * contracts the outermost remaining variable of `src_level` coefficient sets.  The library keeps
 * contracted sets in its own device pool; `slots_out[i]` identifies the set made from parent slot
 * `parents[i]` (slot 0 at level d = the series itself) at coordinate x[i]. */
int abz_contract_nodes(abz_series* s, int src_level, const int64_t* parents, const double* x,
                       int64_t nnodes, int64_t* slots_out);
/* Replaces: workspace_evaluate!(w, x) + integrand at a batch of innermost nodes
 * (src/fourier.jl:445-446,454-455): values_reim [nnodes][ncomp][2].  tail [nnodes][d-1] gives the
 * outer coordinates (x_2..x_d) of each node's line (needed by ABZ_F_LINEAR_X only; may be NULL). */
int abz_eval_line_nodes(abz_series* s, const int64_t* parents, const double* x, const double* tail,
                        int64_t nnodes, int integrand, const double* params, int nparams,
                        double sweep, double* values_reim);
int abz_release_level(abz_series* s, int level); /* drop all contracted sets below `level` */

/* Whole IAI solve with the adaptive GK(7,15) loops on the host and every node batch on the GPU.
 * Replaces: do_solve(::FourierIntegrand, lims, p, ::NestedQuad, cacheval) (src/fourier.jl:493-510)
 * with init_nest (:432-486) and AuxQuadGKJL (src/algorithms.jl:215-239).  Sibling 1-D integrals
 * advance in lockstep so each round is one batch; every 1-D integral makes exactly the scalar
 * refinement decisions of the reference (pop worst panel, bisect), so panel trees are identical.
 * lim_a/lim_b: CubicLimits a, b (len d) or TetrahedralLimits a (lim_b ignored).
 * abstol < 0 / reltol < 0 mean `nothing`.  max_batch = 0: scalar refinement (the reference's default
 * IAI()); max_batch > 0: the BatchIntegrand / NestedBatchIntegrand refinement at every level (pop
 * panels while the error of the remaining ones exceeds the tolerance, 2*15*popped <= max_batch;
 * ref src/batch.jl:41-77, src/fourier.jl:441-473).  out_reim [ncomp][2]; err = the outermost GK
 * error estimate; panels (nullable, [max_panels][2]) receives the outermost integral's final panels. */
int abz_iai_solve(abz_series* s, int lims_kind, const double* lim_a, const double* lim_b,
                  int integrand, const double* params, int nparams, double sweep, double abstol,
                  double reltol, int64_t maxevals, int64_t max_batch, double* out_reim, double* err,
                  int64_t* numevals, double* panels, int64_t max_panels, int64_t* npanels);

/* The same for n_sweep values of the swept parameter at once (batchsolve over omega with IAI,
 * src/interfaces.jl:210-222): the solves are independent adaptive integrals that advance in lock-step,
 * so every contraction / evaluation launch carries the nodes of all of them.  Each solve makes exactly
 * the decisions it would make alone.  out_reim [n_sweep][ncomp][2], err [n_sweep], numevals [n_sweep];
 * panels: outermost panels of the first solve. */
int abz_iai_solve_many(abz_series* s, int lims_kind, const double* lim_a, const double* lim_b,
                       int integrand, const double* params, int nparams, const double* sweeps,
                       int n_sweep, double abstol, double reltol, int64_t maxevals, int64_t max_batch,
                       double* out_reim, double* err, int64_t* numevals, double* panels,
                       int64_t max_panels, int64_t* npanels);

/* ONE IAI solve on several GPUs (disjoint sets of the inner integrals of every refinement round).  All
 * ranks call abz_iai_solve(_many) with the same arguments; every round's innermost integrals are dealt to the ranks in
 * blocks of 64 nodes, each rank integrates its share on its own GPU, and `fn(user, buf, per_rank)` all-gathers the results:
 * `buf` holds world * per_rank doubles, segment `rank` is filled on entry, all segments must be filled on return (RCCL
 * / MPI / gloo: the library does not link a communication layer).  Every rank returns the same value, bit-identical to the
 * single-GPU solve.  fn = NULL switches it off (world = 1 with a hook is allowed: a one-rank rehearsal of the transport).
 * Errors: a rank whose share fails locally (HIP error, out of memory) still joins the collective -- every segment ends in a
 * status word -- and ALL ranks return that rank's ABZ_ERR_* code after the exchange; nobody is left waiting.  `fn` itself
 * must fail on every rank or on none (it is the caller's collective; return non-zero on all ranks to abort the solve).
 * Replaces: nothing in the reference (its parallelism is threads over parameters, src/interfaces.jl:210-222). */
typedef int (*abz_exchange_fn)(void* user, double* buf, int64_t per_rank);
int abz_iai_set_exchange(abz_series* s, abz_exchange_fn fn, void* user, int rank, int world);

/* Replaces: QuadGK.evalrule on a batch of panels (reached from src/algorithms.jl:227-233):
 * values [npanels][15][ncomp][2] in gk node order -> I_reim [npanels][ncomp][2], E [npanels]
 * with I = I_K * h, E = ||I_K - I_G|| * h (2-norm over components). */
int abz_gk15_nodes(double a, double b, double* x15);
int abz_gk15_batch(const double* ab, const double* values_reim, int64_t npanels, int ncomp,
                   double* I_reim, double* E);

#ifdef __cplusplus
}
#endif
#endif /* ABZHIP_H */
