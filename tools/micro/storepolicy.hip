// Cache-policy bits of global stores on gfx950 for the tiled store pattern of the Fourier-eval epilogue:
// plain, nt, sc0, sc1, sc0 sc1, nt sc1, nt sc0 sc1 (inline asm), 4096 workgroups, 605 MB.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__device__ __forceinline__ void st(double* q, double v) {
    if (MODE == 0) *q = v;
    if (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(q), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(q), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory");
    if (MODE == 5) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(q), "v"(v) : "memory");
    if (MODE == 6) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(q), "v"(v) : "memory");
    if (MODE == 7) asm volatile("global_store_dwordx2 %0, %1, off sc0 nt" ::"v"(q), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void wkernel(double* out, int npt_pad, long nlines, int planes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long line = (long)blockIdx.x * 4 + wave; line < nlines; line += (long)gridDim.x * 4) {
        for (int j = 0; j < 3; ++j) {
            const int i1 = lane + 64 * j;
            if (i1 < npt_pad) {
                for (int p = 0; p < planes; ++p) {
                    const double v = (double)(line + p) + i1;
                    st<MODE>(out + (line * planes + p) * (long)npt_pad + i1, v);
                }
            }
        }
    }
}

int main() {
    const int npt_pad = 160, planes = 21;
    const long nlines = 22500;
    const size_t bytes = sizeof(double) * (size_t)planes * nlines * npt_pad;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[8] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0 nt"};
    for (int b = 0; b < 3; ++b) {
        double* d;
        CK(hipMalloc(&d, bytes));
        CK(hipMemset(d, 0, bytes));
        printf("buffer %d:", b);
        for (int mode = 0; mode < 8; ++mode) {
            float best = 1e9, ms;
            for (int rep = 0; rep < 10; ++rep) {
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: wkernel<0><<<4096, 256>>>(d, npt_pad, nlines, planes); break;
                    case 1: wkernel<1><<<4096, 256>>>(d, npt_pad, nlines, planes); break;
                    case 2: wkernel<2><<<4096, 256>>>(d, npt_pad, nlines, planes); break;
                    case 3: wkernel<3><<<4096, 256>>>(d, npt_pad, nlines, planes); break;
                    case 4: wkernel<4><<<4096, 256>>>(d, npt_pad, nlines, planes); break;
                    case 5: wkernel<5><<<4096, 256>>>(d, npt_pad, nlines, planes); break;
                    case 6: wkernel<6><<<4096, 256>>>(d, npt_pad, nlines, planes); break;
                    default: wkernel<7><<<4096, 256>>>(d, npt_pad, nlines, planes); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 1 && ms < best) best = ms;
            }
            printf("  %s %.4f", names[mode], best);
        }
        printf("\n");
    }
    return 0;
}
