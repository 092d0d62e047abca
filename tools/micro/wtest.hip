// Micro-benchmark: HBM write patterns of the Fourier-eval kernel's epilogue (no compute).
//   A: planar  [plane][line*npt + i1]            (current rule layout)
//   B: tiled   [line][plane][npt_pad] (pad written) 
//   C: AoS     [line*npt + i1][plane]
//   D: planar with 16-byte stores (2 nodes per lane)
//   E: tiles of 8 lines, NO padding  [line/8][plane][8*150]: a wave still owns one line, so every seam between two lines
//      (1200 B = 9 x 128 B + 48 B) puts two waves on one 128-B line            F: the same with non-temporal stores
//   G: B with non-temporal stores (the production pattern above the Infinity Cache)
//   H: tiles of 2 lines [line/2][plane][304] (1.3 % padding, one seam per row), non-temporal
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
        if (MODE == 8 || MODE == 9) {  // I / J: the tiled padded layout with 16-byte stores (two adjacent nodes per lane)
            for (int j = 0; j < 2; ++j) {
                const int i1 = 2 * (lane + 64 * j);
                if (i1 < npt_pad) {
                    for (int p = 0; p < planes; ++p) {
                        double2 v = make_double2((double)(line + p), (double)i1);
                        double2* q = reinterpret_cast<double2*>(out + (line * planes + p) * (long)npt_pad + i1);
                        if (MODE == 9) { __builtin_nontemporal_store(v.x, &q->x); __builtin_nontemporal_store(v.y, &q->y); }
                        else *q = v;
                    }
                }
            }
            continue;
        }
        for (int j = 0; j < 3; ++j) {
            const int i1 = lane + 64 * j;
            const int lim = (MODE == 1 || MODE == 6) ? npt_pad : npt;
            if (i1 < lim) {
                for (int p = 0; p < planes; ++p) {
                    const double v = (double)(line + p) + i1;
                    long addr;
                    if (MODE == 0) addr = (long)p * stride + line * npt + i1;
                    else if (MODE == 1 || MODE == 6) addr = (line * planes + p) * (long)npt_pad + i1;
                    else if (MODE == 4 || MODE == 5) addr = ((line / 8) * planes + p) * (long)(8 * npt) + (line % 8) * npt + i1;
                    else if (MODE == 7) addr = ((line / 2) * planes + p) * 304L + (line % 2) * npt + i1;
                    else addr = (line * npt + i1) * (long)planes + p;
                    if (MODE >= 5) __builtin_nontemporal_store(v, out + addr);
                    else out[addr] = v;
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
    const char* names[10] = {"A planar 8B", "B tiled [line][plane][160]", "C AoS", "D planar 16B", "E 8-line tiles no pad", "F 8-line tiles no pad NT",
                            "G tiled [line][plane][160] NT", "H 2-line tiles pitch 304 NT",
                            "I tiled [160] 16-B stores", "J tiled [160] 16-B stores NT"};
    for (int blocks : {8192, 4096, 2048, 1024, 512}) {
        for (int mode = 0; mode < 10; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) wkernel<0><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 1) wkernel<1><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 2) wkernel<2><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 3) wkernel<3><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 4) wkernel<4><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 5) wkernel<5><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 6) wkernel<6><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 7) wkernel<7><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 8) wkernel<8><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                if (mode == 9) wkernel<9><<<blocks, 256>>>(d, npt, npt_pad, nlines, stride, planes);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double gb = ((mode == 1 || mode == 6 || mode >= 8) ? (double)nlines * npt_pad : (double)nk) * planes * 8 / 1e9;
            printf("blocks=%4d %-28s %.4f ms  %.2f TB/s (useful %.2f TB/s)\n", blocks, names[mode], best, gb / best, (double)nk * planes * 8 / 1e9 / best);
        }
    }
    return 0;
}
