// Integer-exact symmetric-PTR tables on the host.
// Replaces AutoSymPTR.symptr_rule as called at src/fourier.jl:271: for every orbit of the npt^d
// grid under `syms` (integer matrices acting on fractional coordinates, mod 1) the first member in
// column-major order (i_1 fastest) is the irreducible node and carries the orbit size as weight.
#include <algorithm>
#include <cstring>
#include <vector>

#include "abz_internal.h"

namespace {
struct SymptrCache {
    int npt = 0, d = 0;
    std::vector<int32_t> syms;
    std::vector<int32_t> idx;
    std::vector<int64_t> w;
};
thread_local SymptrCache g_cache;
}  // namespace

extern "C" int abz_symptr_rule(int npt, int d, const int32_t* syms, int nsyms, int64_t* nirr, int32_t* irr_idx,
                               int64_t* wsym) try {
    ABZ_REQUIRE(npt >= 1 && d >= 1 && d <= ABZ_MAX_DIM, "symptr_rule: npt = %d, d = %d invalid", npt, d);
    ABZ_REQUIRE(syms && nsyms >= 1 && nirr, "symptr_rule: null argument");
    ABZ_REQUIRE((irr_idx == nullptr) == (wsym == nullptr), "irr_idx and wsym must be given together");
    SymptrCache& c = g_cache;
    const size_t nsy = (size_t)nsyms * d * d;
    const bool hit = c.npt == npt && c.d == d && c.syms.size() == nsy && std::memcmp(c.syms.data(), syms, nsy * 4) == 0;
    if (!hit) {
        int64_t N = 1;
        for (int j = 0; j < d; ++j) N *= npt;
        std::vector<uint8_t> visited((size_t)N, 0);
        c.idx.clear();
        c.w.clear();
        std::vector<int64_t> orbit((size_t)nsyms);
        int v[ABZ_MAX_DIM] = {0, 0, 0};
        for (int64_t lin = 0; lin < N; ++lin) {
            if (!visited[(size_t)lin]) {
                int64_t r = lin;
                for (int j = 0; j < d; ++j) {
                    v[j] = (int)(r % npt);
                    r /= npt;
                }
                for (int sidx = 0; sidx < nsyms; ++sidx) {
                    const int32_t* S = syms + (size_t)sidx * d * d;
                    int64_t img = 0, mul = 1;
                    for (int a = 0; a < d; ++a) {
                        int64_t t = 0;
                        for (int b = 0; b < d; ++b) t += (int64_t)S[a * d + b] * v[b];
                        t %= npt;
                        if (t < 0) t += npt;
                        img += t * mul;
                        mul *= npt;
                    }
                    orbit[(size_t)sidx] = img;
                }
                std::sort(orbit.begin(), orbit.end());
                const int64_t distinct = std::unique(orbit.begin(), orbit.end()) - orbit.begin();
                for (int64_t t = 0; t < distinct; ++t) visited[(size_t)orbit[(size_t)t]] = 1;
                orbit.resize((size_t)nsyms);
                if (!visited[(size_t)lin]) {  // syms without the identity: the point itself is its own class member
                    visited[(size_t)lin] = 1;
                }
                for (int j = 0; j < d; ++j) c.idx.push_back(v[j]);
                c.w.push_back(distinct);
            }
        }
        c.npt = npt;
        c.d = d;
        c.syms.assign(syms, syms + nsy);
    }
    const int64_t n = (int64_t)c.w.size();
    if (irr_idx) {
        ABZ_REQUIRE(*nirr >= n, "symptr_rule: buffers hold %lld nodes, need %lld", (long long)*nirr, (long long)n);
        std::memcpy(irr_idx, c.idx.data(), sizeof(int32_t) * c.idx.size());
        std::memcpy(wsym, c.w.data(), sizeof(int64_t) * c.w.size());
    }
    *nirr = n;
    return ABZ_OK;
} ABZ_CATCH_ALL
