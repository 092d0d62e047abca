// Packed coefficient sets of a Hermitian series along its innermost variable (shared by the grid-evaluation and the GGR
// kernels).  A Hermitian series (c(-R) = c(R)^dagger, symmetric frequency range first = -F, M = 2 F + 1) has level-1
// coefficients with c1[-f] = c1[f]^dagger, so only frequencies f >= 0 are kept: c1[0] (upper triangle) and per f > 0
//     dd_a = 2 c1[f]_aa,   s_ab = c1[f]_ab + c1[f]_ba,   t_ab = c1[f]_ab - c1[f]_ba   (a < b)
// with which  H_ab += (s.x pr - s.y pi) + i (t.x pi + t.y pr),  H_aa += dd.x pr - dd.y pi,  p = z^f = (pr, pi):
// one FMA group serves +f and -f.  The combinations are linear in the coefficients, so packing commutes with the
// contraction of the outer variables: the whole chain can run on packed rows of Pk<N>::size(F) instead of M n^2 numbers.
#pragma once
#include <hip/hip_runtime.h>

namespace abz {

template <int N>
struct Pk {
    static constexpr int NT0 = N * (N + 1) / 2;                                   // elements of the f = 0 block
    __host__ __device__ static constexpr int tri(int a, int b) { return b * (b + 1) / 2 + a; }  // c1[0]_ab, a <= b
    __host__ __device__ static constexpr int blk(int f) { return NT0 + (f - 1) * N * N; }        // first element of block f >= 1
    __host__ __device__ static constexpr int dd(int f, int a) { return blk(f) + a; }
    __host__ __device__ static constexpr int pair(int a, int b) { return b * (b - 1) / 2 + a; }  // a < b
    __host__ __device__ static constexpr int ss(int f, int a, int b) { return blk(f) + N + 2 * pair(a, b); }
    __host__ __device__ static constexpr int tt(int f, int a, int b) { return ss(f, a, b) + 1; }
    __host__ __device__ static constexpr int size(int F) { return NT0 + F * N * N; }
};

// packed element e of the set whose full coefficients are c[m][a + N b] (m = 0 .. 2F, frequency m - F)
template <int N>
__device__ __forceinline__ double2 pk_from_full(const double2* __restrict__ c, int F, int e) {
    constexpr int NN = N * N;
    if (e < Pk<N>::NT0) {
        int b = 0;
        while ((b + 1) * (b + 2) / 2 <= e) ++b;
        const int a = e - b * (b + 1) / 2;
        return c[F * NN + a + N * b];
    }
    const int r = e - Pk<N>::NT0;
    const int f = r / NN + 1, q = r - (f - 1) * NN;
    const double2* __restrict__ cf = c + (F + f) * NN;
    if (q < N) {
        const double2 v = cf[q + N * q];
        return make_double2(2.0 * v.x, 2.0 * v.y);
    }
    const int pi = (q - N) >> 1;
    int b = 1;
    while ((b + 1) * b / 2 <= pi) ++b;
    const int a = pi - b * (b - 1) / 2;
    const double2 u = cf[a + N * b], v = cf[b + N * a];
    return ((q - N) & 1) ? make_double2(u.x - v.x, u.y - v.y) : make_double2(u.x + v.x, u.y + v.y);
}

}  // namespace abz
