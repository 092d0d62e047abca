// Symmetric-PTR orbit tables on the GPU ("symptr_rule ... is not a fast or parallel
// algorithm", src/fourier.jl:270).  Same integer results as symptr_host.cpp, bit for bit:
//   flag kernel     one thread per grid point: image of the point under every symmetry (integer
//                   matrices acting mod npt); the point is the orbit's representative iff it is the
//                   smallest image in column-major order; its weight = number of DISTINCT images.
//   compaction      order-preserving stream compaction of the representatives (block counts ->
//                   exclusive scan -> scatter), so the list comes out in column-major order.
#include "abz_internal.h"

namespace abz {

struct SymArgs {
    int npt, d, nsyms;
    int small;  // every |S v| < 2^31: 32-bit arithmetic (a 64-bit modulo costs ~4x more)
    int64_t N;
    int S[48 * 9];  // up to 48 symmetries of a 3-d lattice, row-major
};

__device__ __forceinline__ int64_t sym_image(const SymArgs& a, const int* v, int s) {
    int64_t img = 0, mul = 1;
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

}  // namespace abz

extern "C" int abz_symptr_rule_device(abz_ctx* ctx, int npt, int d, const int32_t* syms, int nsyms, int64_t* nirr,
                                      int32_t* irr_idx, int64_t* wsym) {
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
}
