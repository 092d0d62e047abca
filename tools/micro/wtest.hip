// Micro-benchmark: HBM write patterns of the Fourier-eval kernel's epilogue (no compute).
//   A: planar  [plane][line*npt + i1]            (current rule layout)
//   B: tiled   [line][plane][npt_pad] (pad written) 
//   C: AoS     [line*npt + i1][plane]
//   D: planar with 16-byte stores (2 nodes per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void wkernel(double* out, int npt, int npt_pad, long nlines, long stride, int planes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long line = (long)blockIdx.x * 4 + wave; line < nlines; line += (long)gridDim.x * 4) {
        if (MODE == 3) {
            for (int j = 0; j < 2; ++j) {
                const int i1 = 2 * (lane + 64 * j);
                if (i1 < npt) {
                    for (int p = 0; p < planes; ++p) {
                        double2 v = make_double2((double)(line + p), (double)i1);
                        *reinterpret_cast<double2*>(out + (long)p * stride + line * npt + i1) = v;
                    }
                }
            }
            continue;
        }
        for (int j = 0; j < 3; ++j) {
            const int i1 = lane + 64 * j;
            const int lim = (MODE == 1) ? npt_pad : npt;
            if (i1 < lim) {
                for (int p = 0; p < planes; ++p) {
                    const double v = (double)(line + p) + i1;
                    long addr;
                    if (MODE == 0) addr = (long)p * stride + line * npt + i1;
                    else if (MODE == 1) addr = (line * planes + p) * (long)npt_pad + i1;
                    else addr = (line * npt + i1) * (long)planes + p;
                    out[addr] = v;
                }
            }
        }
    }
}

int main() {
    const int npt = 150, npt_pad = 160, planes = 21;
    const long nlines = 22500, nk = nlines * npt;
    const long stride = (nk + 63) / 64 * 64;
    const size_t bytes = sizeof(double) * (size_t)planes * nlines * npt_pad + (1 << 20);
    double* d;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[4] = {"A planar 8B", "B tiled [line][plane][160]", "C AoS", "D planar 16B"};
    for (int blocks : {2048, 1024, 512}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) wkernel<0><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 1) wkernel<1><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 2) wkernel<2><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 3) wkernel<3><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double gb = (mode == 1 ? (double)nlines * npt_pad : (double)nk) * planes * 8 / 1e9;
            printf("blocks=%4d %-28s %.4f ms  %.2f TB/s (useful %.2f TB/s)\n", blocks, names[mode], best, gb / best, (double)nk * planes * 8 / 1e9 / best);
        }
    }
    return 0;
}
