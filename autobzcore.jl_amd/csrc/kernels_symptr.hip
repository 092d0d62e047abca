// Symmetric-PTR orbit tables on the GPU ("symptr_rule ... is not a fast or parallel
// algorithm", src/fourier.jl:270).  Same integer results as symptr_host.cpp, bit for bit:
//   flag kernel     one thread per grid point: image of the point under every symmetry (integer
//                   matrices acting mod npt); the point is the orbit's representative iff it is the
//                   smallest image in column-major order; its weight = number of DISTINCT images.
//   compaction      order-preserving stream compaction of the representatives (block counts ->
//                   exclusive scan -> scatter), so the list comes out in column-major order.
#include "abz_internal.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

namespace abz {

struct SymArgs {
    int npt, d, nsyms;
    int small;  // every |S v| < 2^31: 32-bit arithmetic (a 64-bit modulo costs ~4x more)
    int perm;   // every matrix is a signed permutation (cubic / inversion groups in the lattice basis): no modulo at all
    int group;  // the set is closed under multiplication (a group): orbit size = nsyms / |stabiliser|
    int64_t N;
    int S[48 * 9];  // up to 48 symmetries of a 3-d lattice, row-major
};

__device__ __forceinline__ int64_t sym_image(const SymArgs& a, const int* v, int s) {
    int64_t img = 0, mul = 1;
    if (a.perm) {  // row r has one entry +-1, in column c: the image coordinate is v[c] or (npt - v[c]) mod npt
        for (int r = 0; r < a.d; ++r) {
            int t = 0;
            for (int c = 0; c < a.d; ++c) {
                const int e = a.S[(s * a.d + r) * a.d + c];
                t = e > 0 ? v[c] : (e < 0 ? (v[c] == 0 ? 0 : a.npt - v[c]) : t);
            }
            img += (int64_t)t * mul;
            mul *= a.npt;
        }
        return img;
    }
    for (int r = 0; r < a.d; ++r) {
        int64_t t;
        if (a.small) {
            int t32 = 0;
            for (int c = 0; c < a.d; ++c) t32 += a.S[(s * a.d + r) * a.d + c] * v[c];
            t32 %= a.npt;
            if (t32 < 0) t32 += a.npt;
            t = t32;
        } else {
            t = 0;
            for (int c = 0; c < a.d; ++c) t += (int64_t)a.S[(s * a.d + r) * a.d + c] * v[c];
            t %= a.npt;
            if (t < 0) t += a.npt;
        }
        img += t * mul;
        mul *= a.npt;
    }
    return img;
}

__global__ __launch_bounds__(256) void symptr_flag_kernel(SymArgs a, int* __restrict__ wflag) {
    const int64_t lin = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= a.N) return;
    int v[3] = {0, 0, 0};
    int64_t r = lin;
    for (int j = 0; j < a.d; ++j) {
        v[j] = (int)(r % a.npt);
        r /= a.npt;
    }
    bool rep = true;
    for (int s = 0; s < a.nsyms && rep; ++s) rep = sym_image(a, v, s) >= lin;
    int w = 0;
    if (rep) {
        // distinct images (the set may or may not contain the point itself, like the host version)
        for (int s = 0; s < a.nsyms; ++s) {
            const int64_t is = sym_image(a, v, s);
            bool dup = false;
            for (int t = 0; t < s && !dup; ++t) dup = sym_image(a, v, t) == is;
            w += dup ? 0 : 1;
        }
    }
    wflag[lin] = w;
}

__global__ __launch_bounds__(256) void compact_count_kernel(const int* __restrict__ wflag, int64_t N, int* __restrict__ counts) {
    __shared__ int sh[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool f = i < N && wflag[i] != 0;
    const unsigned long long b = __ballot(f);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void compact_scatter_kernel(const int* __restrict__ wflag, int64_t N,
                                                              const int64_t* __restrict__ offsets, int npt, int d,
                                                              int32_t* __restrict__ idx, int64_t* __restrict__ w) {
    __shared__ int sh[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int wv = i < N ? wflag[i] : 0;
    const bool f = wv != 0;
    const unsigned long long b = __ballot(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = __popcll(b);
    __syncthreads();
    int before = 0;
    for (int t = 0; t < wave; ++t) before += sh[t];
    before += __popcll(b & ((1ull << lane) - 1ull));
    if (f) {
        const int64_t o = offsets[blockIdx.x] + before;
        int64_t r = i;
        for (int j = 0; j < d; ++j) {
            idx[o * d + j] = (int32_t)(r % npt);
            r /= npt;
        }
        w[o] = wv;
    }
}

int symptr_device(abz_ctx* ctx, int npt, int d, const int32_t* syms, int nsyms, std::vector<int32_t>& idx,
                  std::vector<int64_t>& w) {
    if (nsyms > 48 || d > 3) {
        set_error("symptr_device: at most 48 symmetries of a <= 3-d lattice");
        return ABZ_ERR_UNSUPPORTED;
    }
    SymArgs a;
    a.npt = npt;
    a.d = d;
    a.nsyms = nsyms;
    a.N = 1;
    for (int j = 0; j < d; ++j) a.N *= npt;
    int64_t smax = 1;
    for (int i = 0; i < nsyms * d * d; ++i) {
        a.S[i] = syms[i];
        smax = std::max<int64_t>(smax, std::llabs((long long)syms[i]));
    }
    a.small = (smax * d * (int64_t)npt < ((int64_t)1 << 30)) ? 1 : 0;
    a.perm = 1;
    a.group = 0;
    for (int sidx = 0; sidx < nsyms && a.perm; ++sidx)
        for (int r = 0; r < d && a.perm; ++r) {
            int nz = 0;
            for (int c = 0; c < d; ++c) {
                const int e = syms[(sidx * d + r) * d + c];
                if (e != 0) nz += (e == 1 || e == -1) ? 1 : 2;
            }
            if (nz != 1) a.perm = 0;
        }
    const int64_t nb = (a.N + 255) / 256;
    DevBuf flag, counts, offs, didx, dw;
    int rc;
    if ((rc = flag.reserve(sizeof(int) * (size_t)a.N))) return rc;
    if ((rc = counts.reserve(sizeof(int) * (size_t)nb))) return rc;
    if ((rc = offs.reserve(sizeof(int64_t) * (size_t)nb))) return rc;
    auto cleanup = [&]() {
        flag.release();
        counts.release();
        offs.release();
        didx.release();
        dw.release();
    };
    hipLaunchKernelGGL(symptr_flag_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, a, flag.as<int>());
    hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, flag.as<int>(), a.N, counts.as<int>());
    std::vector<int> hc((size_t)nb);
    hipError_t e = hipMemcpyAsync(hc.data(), counts.p, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        cleanup();
        set_error("symptr_device: %s", hipGetErrorString(e));
        return ABZ_ERR_HIP;
    }
    std::vector<int64_t> ho((size_t)nb);
    int64_t tot = 0;
    for (int64_t b = 0; b < nb; ++b) {
        ho[(size_t)b] = tot;
        tot += hc[(size_t)b];
    }
    idx.resize((size_t)(tot * d));
    w.resize((size_t)tot);
    if (tot > 0) {
        if ((rc = didx.reserve(sizeof(int32_t) * (size_t)(tot * d))) || (rc = dw.reserve(sizeof(int64_t) * (size_t)tot))) {
            cleanup();
            return rc;
        }
        e = hipMemcpyAsync(offs.p, ho.data(), sizeof(int64_t) * (size_t)nb, hipMemcpyHostToDevice, ctx->stream);
        hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)nb), dim3(256), 0, ctx->stream, flag.as<int>(), a.N,
                           offs.as<int64_t>(), npt, d, didx.as<int32_t>(), dw.as<int64_t>());
        if (e != hipSuccess) {
            cleanup();
            set_error("symptr_device: %s", hipGetErrorString(e));
            return ABZ_ERR_HIP;
        }
        // MB-sized results: through the pinned staging buffer
        if ((rc = stage_d2h(ctx, idx.data(), didx.p, sizeof(int32_t) * idx.size())) ||
            (rc = stage_d2h(ctx, w.data(), dw.p, sizeof(int64_t) * w.size()))) {
            cleanup();
            return rc;
        }
    }
    cleanup();
    return ABZ_OK;
}


// ------------------------------------------------------------------------------------------
// The same tables LEFT ON THE DEVICE, together with the contraction plan of the node list, for
// abz_ptr_rule_build_sym (ref: the FourierMonkhorstPack constructor builds wsym / flags and fills its values
// in one go, src/fourier.jl:265-277).  The host sees three integers (node and item counts).
//   rep kernel      representative flags only (the early-exit test)
//   line counts     one wave per grid line: representatives per line -> scans give every line's first node
//                   (= run_start of its level-1 coefficient set), its rank among the lines that hold nodes
//                   (= the level-1 item = parent of its nodes), likewise planes -> level-2 items
//   scatter         one wave per line: nodes in column-major order with i_1, parent, grid indices
//   weights         one thread per NODE (dense: the O(nsyms^2) distinct-image count no longer idles the 47 of 48
//                   lanes of a wave that are no representatives)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sym_rep_kernel(SymArgs a, uint8_t* __restrict__ rep) {
    const int64_t lin = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= a.N) return;
    int v[3] = {0, 0, 0};
    int64_t r = lin;
    for (int j = 0; j < a.d; ++j) {
        v[j] = (int)(r % a.npt);
        r /= a.npt;
    }
    bool isrep = true;
    for (int s = 0; s < a.nsyms && isrep; ++s) isrep = sym_image(a, v, s) >= lin;
    rep[lin] = isrep ? 1 : 0;
}

__global__ __launch_bounds__(256) void sym_line_count_kernel(const uint8_t* __restrict__ rep, int npt, int64_t nlines,
                                                             int* __restrict__ cnt) {
    const int64_t line = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (line >= nlines) return;
    int c = 0;
    for (int i0 = 0; i0 < npt; i0 += 64) {
        const int i = i0 + lane;
        c += __popcll(__ballot(i < npt && rep[line * npt + i] != 0));
    }
    if (lane == 0) cnt[line] = c;
}

// cnt2[plane] = number of its lines that hold nodes
__global__ __launch_bounds__(256) void sym_plane_count_kernel(const int* __restrict__ cnt1, int npt, int64_t nplanes,
                                                              int* __restrict__ cnt2) {
    const int64_t pl = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pl >= nplanes) return;
    int c = 0;
    for (int i = 0; i < npt; ++i) c += cnt1[pl * npt + i] != 0 ? 1 : 0;
    cnt2[pl] = c;
}

// exclusive scan of in[i] (NONZERO: of in[i] != 0) by one block; *total = the sum
template <bool NONZERO>
__global__ __launch_bounds__(1024) void sym_scan_kernel(const int* __restrict__ in, int64_t n, int64_t* __restrict__ out,
                                                        int64_t* __restrict__ total) {
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = (int64_t)t * per, e = b + per < n ? b + per : n;
    int64_t sum = 0;
    for (int64_t i = b; i < e; ++i) sum += NONZERO ? (in[i] != 0 ? 1 : 0) : in[i];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t add = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int64_t run = part[t] - sum;
    for (int64_t i = b; i < e; ++i) {
        out[i] = run;
        run += NONZERO ? (in[i] != 0 ? 1 : 0) : in[i];
    }
    if (t == 1023) *total = part[1023];
}

struct SymScatterArgs {
    const uint8_t* rep;
    const int* cnt1;
    const int64_t* lineoff;  // first node of every line
    const int64_t* rank1;    // level-1 item of every line (d >= 2)
    const int64_t* rank2;    // level-2 item of every plane (d == 3)
    int npt, d;
    int64_t nlines, nk, nitems1;
    int32_t* idx;            // [d][nk]
    int32_t* gi0;
    int64_t* parent0;
    int32_t* gi1;
    int64_t* parent1;
    int64_t* runs;           // [nitems1 + 1]
};

__global__ __launch_bounds__(256) void sym_scatter_kernel(SymScatterArgs a) {
    const int64_t line = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (line >= a.nlines) return;
    if (line == 0 && lane == 0 && a.d >= 2) a.runs[a.nitems1] = a.nk;
    if (a.cnt1[line] == 0) return;
    const int i2 = a.d >= 2 ? (int)(line % a.npt) : 0;
    const int i3 = a.d >= 3 ? (int)(line / a.npt) : 0;
    const int64_t item = a.d >= 2 ? a.rank1[line] : 0;
    int64_t o = a.lineoff[line];
    if (lane == 0 && a.d >= 2) {
        a.gi1[item] = i2;
        a.parent1[item] = a.d >= 3 ? a.rank2[line / a.npt] : 0;
        a.runs[item] = o;
    }
    for (int i0 = 0; i0 < a.npt; i0 += 64) {
        const int i = i0 + lane;
        const bool f = i < a.npt && a.rep[line * a.npt + i] != 0;
        const unsigned long long b = __ballot(f);
        if (f) {
            const int64_t k = o + __popcll(b & ((1ull << lane) - 1ull));
            a.idx[k] = i;
            if (a.d >= 2) a.idx[a.nk + k] = i2;
            if (a.d >= 3) a.idx[2 * a.nk + k] = i3;
            a.gi0[k] = i;
            a.parent0[k] = item;
        }
        o += __popcll(b);
    }
}

__global__ __launch_bounds__(256) void sym_items2_kernel(const int* __restrict__ cnt2, const int64_t* __restrict__ rank2,
                                                         int64_t nplanes, int32_t* __restrict__ gi2, int64_t* __restrict__ parent2) {
    const int64_t pl = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pl >= nplanes || cnt2[pl] == 0) return;
    gi2[rank2[pl]] = (int32_t)pl;  // the top level: the plane index is i_3
    parent2[rank2[pl]] = 0;
}

// weight of a node = number of DISTINCT images under the symmetry set (like the host version).  The images are computed
// once into LDS, then compared pairwise (recomputing them inside the pair loop made this kernel, one wave per SIMD and
// 1176 image evaluations per lane, the longest step of the whole table build).
constexpr int SYMW_THREADS = 128;
__global__ __launch_bounds__(SYMW_THREADS) void sym_weight_kernel(SymArgs a, const int32_t* __restrict__ idx, int64_t nk,
                                                                   double* __restrict__ w) {
    extern __shared__ int64_t simg[];  // [nsyms][SYMW_THREADS]
    const int64_t k = (int64_t)blockIdx.x * SYMW_THREADS + threadIdx.x;
    if (k >= nk) return;
    int v[3] = {0, 0, 0};
    for (int j = 0; j < a.d; ++j) v[j] = idx[j * nk + k];
    if (a.group) {
        // orbit-stabiliser: the images of a node under a GROUP fall into nsyms / |stabiliser| classes of equal size, so the
        // number of distinct images needs no pairwise comparison (48 image evaluations instead of 48 + 1128 dependent
        // LDS reads: 0.12 -> 0.01 ms per grid, the same integers)
        int64_t self = 0, mul = 1;
        for (int j = 0; j < a.d; ++j) {
            self += (int64_t)v[j] * mul;
            mul *= a.npt;
        }
        int stab = 0;
        for (int s = 0; s < a.nsyms; ++s) stab += sym_image(a, v, s) == self ? 1 : 0;
        w[k] = (double)(a.nsyms / (stab > 0 ? stab : 1));
        return;
    }
    for (int s = 0; s < a.nsyms; ++s) simg[s * SYMW_THREADS + threadIdx.x] = sym_image(a, v, s);
    int cnt = 0;
    for (int s = 0; s < a.nsyms; ++s) {
        const int64_t is = simg[s * SYMW_THREADS + threadIdx.x];
        bool dup = false;
        for (int t = 0; t < s && !dup; ++t) dup = simg[t * SYMW_THREADS + threadIdx.x] == is;
        cnt += dup ? 0 : 1;
    }
    w[k] = (double)cnt;
}

void SymTables::release() { arena.release(); }

// load this file's code object now (abz_ctx_create) instead of inside the first symmetric solve: 0.6 ms of a cold 2.1 ms
void preload_symptr_code() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)sym_rep_kernel);
}


// Is the set of integer matrices closed under multiplication (then it is a group: finite, invertible)?  Checked once per
// symmetry set (the last one is remembered): 48^2 products looked up in a sorted list of matrix keys.
static bool syms_closed(int d, const int32_t* syms, int nsyms) {
    static thread_local std::vector<int32_t> last;
    static thread_local int last_d = 0;
    static thread_local bool last_ok = false;
    const size_t n = (size_t)nsyms * d * d;
    if (last_d == d && last.size() == n && std::equal(last.begin(), last.end(), syms)) return last_ok;
    auto key = [&](const int64_t* m, bool& ok) {
        uint64_t k = 0;
        for (int i = 0; i < d * d; ++i) {
            if (m[i] < -3 || m[i] > 3) ok = false;  // lattice-basis point-group matrices have entries in -2..2
            k = k * 7 + (uint64_t)(m[i] + 3);
        }
        return k;
    };
    bool ok = true;
    std::vector<uint64_t> keys((size_t)nsyms);
    int64_t m[9];
    for (int s = 0; s < nsyms; ++s) {
        for (int i = 0; i < d * d; ++i) m[i] = syms[(size_t)s * d * d + i];
        keys[(size_t)s] = key(m, ok);
    }
    std::sort(keys.begin(), keys.end());
    if (std::adjacent_find(keys.begin(), keys.end()) != keys.end()) ok = false;  // duplicates: count images the long way
    for (int s = 0; s < nsyms && ok; ++s)
        for (int t = 0; t < nsyms && ok; ++t) {
            for (int r = 0; r < d; ++r)
                for (int c = 0; c < d; ++c) {
                    int64_t acc = 0;
                    for (int q = 0; q < d; ++q) acc += (int64_t)syms[((size_t)s * d + r) * d + q] * syms[((size_t)t * d + q) * d + c];
                    m[r * d + c] = acc;
                }
            bool inrange = true;
            const uint64_t k = key(m, inrange);
            if (!inrange || !std::binary_search(keys.begin(), keys.end(), k)) ok = false;
        }
    last.assign(syms, syms + n);
    last_d = d;
    last_ok = ok;
    return ok;
}

int sym_tables_device(abz_ctx* ctx, int npt, int d, const int32_t* syms, int nsyms, SymTables& st) {
    if (nsyms > 48 || d > 3 || d < 1) {
        set_error("symmetric rule tables: at most 48 symmetries of a <= 3-d lattice");
        return ABZ_ERR_UNSUPPORTED;
    }
    SymArgs a;
    a.npt = npt;
    a.d = d;
    a.nsyms = nsyms;
    a.N = 1;
    for (int j = 0; j < d; ++j) a.N *= npt;
    int64_t smax = 1;
    for (int i = 0; i < nsyms * d * d; ++i) {
        a.S[i] = syms[i];
        smax = std::max<int64_t>(smax, std::llabs((long long)syms[i]));
    }
    a.small = (smax * d * (int64_t)npt < ((int64_t)1 << 30)) ? 1 : 0;
    a.perm = 1;
    for (int sidx = 0; sidx < nsyms && a.perm; ++sidx)
        for (int r = 0; r < d && a.perm; ++r) {
            int nz = 0;
            for (int c = 0; c < d; ++c) {
                const int e = syms[(sidx * d + r) * d + c];
                if (e != 0) nz += (e == 1 || e == -1) ? 1 : 2;
            }
            if (nz != 1) a.perm = 0;
        }
    a.group = syms_closed(d, syms, nsyms) ? 1 : 0;
    const int64_t nlines = a.N / npt, nplanes = d >= 3 ? nlines / npt : 0;
    // one block for all temporaries (every allocation and every return to the allocator costs a driver call or a
    // device synchronisation: 16 of them per grid were most of a first symmetric solve)
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_rep = 0, o_cnt1 = o_rep + up((size_t)a.N), o_off = o_cnt1 + up(sizeof(int) * (size_t)nlines),
                 o_r1 = o_off + up(sizeof(int64_t) * (size_t)nlines), o_cnt2 = o_r1 + up(sizeof(int64_t) * (size_t)nlines),
                 o_r2 = o_cnt2 + up(sizeof(int) * (size_t)std::max<int64_t>(nplanes, 1)),
                 o_end = o_r2 + up(sizeof(int64_t) * (size_t)std::max<int64_t>(nplanes, 1));
    DevBuf tmp;
    auto cleanup = [&]() { tmp.release(); };
    int rc;
    if ((rc = tmp.reserve(o_end)) || (rc = mbox_reserve(ctx))) {
        cleanup();
        return rc;
    }
    char* const tb = static_cast<char*>(tmp.p);
    uint8_t* const rep = reinterpret_cast<uint8_t*>(tb + o_rep);
    int* const cnt1 = reinterpret_cast<int*>(tb + o_cnt1);
    int64_t* const lineoff = reinterpret_cast<int64_t*>(tb + o_off);
    int64_t* const rank1 = reinterpret_cast<int64_t*>(tb + o_r1);
    int* const cnt2 = reinterpret_cast<int*>(tb + o_cnt2);
    int64_t* const rank2 = reinterpret_cast<int64_t*>(tb + o_r2);
    // the three totals land in the pinned mailbox (zero copy): one stream synchronisation, no copy call
    int64_t* tot_dev = reinterpret_cast<int64_t*>(static_cast<char*>(ctx->mbox_dev) + ctx->mbox_cap / 2);
    volatile int64_t* tot_host = reinterpret_cast<volatile int64_t*>(static_cast<char*>(ctx->mbox) + ctx->mbox_cap / 2);
    hipStream_t st_ = ctx->stream;
    const bool dbg = abz_switch(SW_DEBUG_TIMING) != 0;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tdbg = tnow();
    auto lap = [&](const char* what) {
        if (dbg) {
            (void)hipStreamSynchronize(st_);
            const double t = tnow();
            fprintf(stderr, "[abz] sym_tables %-12s %8.3f ms\n", what, 1e3 * (t - tdbg));
            tdbg = t;
        }
    };
    lap("alloc");
    hipLaunchKernelGGL(sym_rep_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, st_, a, rep);
    lap("rep");
    hipLaunchKernelGGL(sym_line_count_kernel, dim3((unsigned)((nlines + 3) / 4)), dim3(256), 0, st_, rep, npt, nlines, cnt1);
    hipLaunchKernelGGL(sym_scan_kernel<false>, dim3(1), dim3(1024), 0, st_, cnt1, nlines, lineoff, tot_dev);
    hipLaunchKernelGGL(sym_scan_kernel<true>, dim3(1), dim3(1024), 0, st_, cnt1, nlines, rank1, tot_dev + 1);
    if (d >= 3) {
        hipLaunchKernelGGL(sym_plane_count_kernel, dim3((unsigned)((nplanes + 255) / 256)), dim3(256), 0, st_, cnt1, npt, nplanes, cnt2);
        hipLaunchKernelGGL(sym_scan_kernel<true>, dim3(1), dim3(1024), 0, st_, cnt2, nplanes, rank2, tot_dev + 2);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st_);
    if (e != hipSuccess) {
        cleanup();
        set_error("symmetric rule tables: %s", hipGetErrorString(e));
        return ABZ_ERR_HIP;
    }
    lap("counts+scans");
    st.release();
    st.npt = npt;
    st.d = d;
    st.nk = tot_host[0];
    for (int i = 0; i <= ABZ_MAX_DIM; ++i) st.nitems[i] = 0;
    if (d >= 2) st.nitems[1] = tot_host[1];
    if (d >= 3) st.nitems[2] = tot_host[2];
    const int64_t nk = st.nk;
    if (nk > 0) {
        const int64_t n1 = st.nitems[1], n2 = st.nitems[2];
        const size_t p_idx = 0, p_w = p_idx + up(sizeof(int32_t) * (size_t)(nk * d)), p_g0 = p_w + up(sizeof(double) * (size_t)nk),
                     p_p0 = p_g0 + up(sizeof(int32_t) * (size_t)nk), p_g1 = p_p0 + up(sizeof(int64_t) * (size_t)nk),
                     p_p1 = p_g1 + up(sizeof(int32_t) * (size_t)n1), p_ru = p_p1 + up(sizeof(int64_t) * (size_t)n1),
                     p_g2 = p_ru + up(sizeof(int64_t) * (size_t)(n1 + 1)), p_p2 = p_g2 + up(sizeof(int32_t) * (size_t)n2),
                     p_end = p_p2 + up(sizeof(int64_t) * (size_t)n2);
        if ((rc = st.arena.reserve(p_end))) {
            cleanup();
            st.release();
            return rc;
        }
        st.arena_bytes = p_end;
        char* const ab = static_cast<char*>(st.arena.p);
        st.idx = reinterpret_cast<int32_t*>(ab + p_idx);
        st.w = reinterpret_cast<double*>(ab + p_w);
        st.gi[0] = reinterpret_cast<int32_t*>(ab + p_g0);
        st.parent[0] = reinterpret_cast<int64_t*>(ab + p_p0);
        st.gi[1] = reinterpret_cast<int32_t*>(ab + p_g1);
        st.parent[1] = reinterpret_cast<int64_t*>(ab + p_p1);
        st.runs = reinterpret_cast<int64_t*>(ab + p_ru);
        st.gi[2] = reinterpret_cast<int32_t*>(ab + p_g2);
        st.parent[2] = reinterpret_cast<int64_t*>(ab + p_p2);
        SymScatterArgs sa;
        sa.rep = rep;
        sa.cnt1 = cnt1;
        sa.lineoff = lineoff;
        sa.rank1 = rank1;
        sa.rank2 = rank2;
        sa.npt = npt;
        sa.d = d;
        sa.nlines = nlines;
        sa.nk = nk;
        sa.nitems1 = st.nitems[1];
        sa.idx = st.idx;
        sa.gi0 = st.gi[0];
        sa.parent0 = st.parent[0];
        sa.gi1 = st.gi[1];
        sa.parent1 = st.parent[1];
        sa.runs = st.runs;
        lap("alloc out");
        hipLaunchKernelGGL(sym_scatter_kernel, dim3((unsigned)((nlines + 3) / 4)), dim3(256), 0, st_, sa);
        if (d >= 3)
            hipLaunchKernelGGL(sym_items2_kernel, dim3((unsigned)((nplanes + 255) / 256)), dim3(256), 0, st_, cnt2, rank2, nplanes,
                               st.gi[2], st.parent[2]);
        lap("scatter");
        hipLaunchKernelGGL(sym_weight_kernel, dim3((unsigned)((nk + SYMW_THREADS - 1) / SYMW_THREADS)), dim3(SYMW_THREADS),
                           sizeof(int64_t) * (size_t)nsyms * SYMW_THREADS, st_, a, st.idx, nk, st.w);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st_);  // the temporaries below go back to the allocator
        if (e != hipSuccess) {
            cleanup();
            st.release();
            set_error("symmetric rule tables: %s", hipGetErrorString(e));
            return ABZ_ERR_HIP;
        }
    }
    lap("weights");
    st.syms.assign(syms, syms + (size_t)nsyms * d * d);
    cleanup();
    lap("free tmp");
    return ABZ_OK;
}

}  // namespace abz

extern "C" int abz_symptr_rule_device(abz_ctx* ctx, int npt, int d, const int32_t* syms, int nsyms, int64_t* nirr,
                                      int32_t* irr_idx, int64_t* wsym) try {
    ABZ_REQUIRE(ctx, "null ctx");
    ABZ_REQUIRE(npt >= 1 && d >= 1 && d <= ABZ_MAX_DIM, "symptr_rule: npt = %d, d = %d invalid", npt, d);
    ABZ_REQUIRE(syms && nsyms >= 1 && nirr, "symptr_rule: null argument");
    ABZ_REQUIRE((irr_idx == nullptr) == (wsym == nullptr), "irr_idx and wsym must be given together");
    ABZ_HIP(hipSetDevice(ctx->device));
    // results of the last call are kept so that the size query and the fill share one device pass
    static thread_local std::vector<int32_t> kidx;
    static thread_local std::vector<int64_t> kw;
    static thread_local std::vector<int32_t> ksyms;
    static thread_local int knpt = 0, kd = 0;
    const size_t nsy = (size_t)nsyms * d * d;
    const bool hit = knpt == npt && kd == d && ksyms.size() == nsy && std::equal(ksyms.begin(), ksyms.end(), syms);
    if (!hit) {
        int rc = abz::symptr_device(ctx, npt, d, syms, nsyms, kidx, kw);
        if (rc) return rc;
        knpt = npt;
        kd = d;
        ksyms.assign(syms, syms + nsy);
    }
    const int64_t n = (int64_t)kw.size();
    if (irr_idx) {
        ABZ_REQUIRE(*nirr >= n, "symptr_rule: buffers hold %lld nodes, need %lld", (long long)*nirr, (long long)n);
        std::copy(kidx.begin(), kidx.end(), irr_idx);
        std::copy(kw.begin(), kw.end(), wsym);
    }
    *nirr = n;
    return ABZ_OK;
} ABZ_CATCH_ALL
