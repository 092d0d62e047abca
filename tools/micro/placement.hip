// Does the PLACEMENT of a 605 MB buffer change the bandwidth of streaming stores into it?  Several buffers are allocated
// one after the other (all kept), and for each the tiled store pattern of the Fourier-eval epilogue (wtest.hip mode B/G)
// and a plain hipMemsetAsync are timed.  tools/time_eval_blocks.py showed the eval kernel at 0.1128 ms on one buffer and
// 0.1239 ms on another of the same process.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// planar with padded rows: [plane][line][160] -- per plane all resident waves write ONE dense front (adjacent waves,
// adjacent 1280-B rows), 21 fronts in all, instead of 3000 independent 27-KB tile streams
template <bool NT>
__global__ __launch_bounds__(256) void pkernel(double* out, int npt_pad, long nlines, int planes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long stride = nlines * npt_pad;
    for (long line = (long)blockIdx.x * 4 + wave; line < nlines; line += (long)gridDim.x * 4) {
        for (int j = 0; j < 3; ++j) {
            const int i1 = lane + 64 * j;
            if (i1 < npt_pad) {
                for (int p = 0; p < planes; ++p) {
                    const double v = (double)(line + p) + i1;
                    double* q = out + p * stride + line * npt_pad + i1;
                    if (NT) __builtin_nontemporal_store(v, q); else *q = v;
                }
            }
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void wkernel(double* out, int npt_pad, long nlines, int planes) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long line = (long)blockIdx.x * 4 + wave; line < nlines; line += (long)gridDim.x * 4) {
        for (int j = 0; j < 3; ++j) {
            const int i1 = lane + 64 * j;
            if (i1 < npt_pad) {
                for (int p = 0; p < planes; ++p) {
                    const double v = (double)(line + p) + i1;
                    double* q = out + (line * planes + p) * (long)npt_pad + i1;
                    if (NT) __builtin_nontemporal_store(v, q); else *q = v;
                }
            }
        }
    }
}

int main(int argc, char** argv) {
    const int npt_pad = 160, planes = 21;
    const long nlines = 22500;
    const size_t bytes = sizeof(double) * (size_t)planes * nlines * npt_pad;
    const int nbuf = argc > 1 ? atoi(argv[1]) : 8;
    std::vector<double*> bufs;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int b = 0; b < nbuf; ++b) {
        double* d;
        CK(hipMalloc(&d, bytes));
        bufs.push_back(d);
        CK(hipMemset(d, 0, bytes));
        float t_k = 1e9, t_nt = 1e9, t_m = 1e9, t_p = 1e9, t_pn = 1e9, ms;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0));
            wkernel<false><<<4096, 256>>>(d, npt_pad, nlines, planes);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1 && ms < t_k) t_k = ms;
            CK(hipEventRecord(e0));
            wkernel<true><<<4096, 256>>>(d, npt_pad, nlines, planes);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1 && ms < t_nt) t_nt = ms;
            CK(hipEventRecord(e0));
            pkernel<false><<<4096, 256>>>(d, npt_pad, nlines, planes);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1 && ms < t_p) t_p = ms;
            CK(hipEventRecord(e0));
            pkernel<true><<<4096, 256>>>(d, npt_pad, nlines, planes);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1 && ms < t_pn) t_pn = ms;
            CK(hipEventRecord(e0));
            CK(hipMemsetAsync(d, 0, bytes));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1 && ms < t_m) t_m = ms;
        }
        printf("buffer %d at %p: tiled stores %.4f ms (%.2f TB/s)  non-temporal %.4f ms (%.2f TB/s)  memset %.4f ms (%.2f TB/s)  "
               "padded planar %.4f ms (%.2f TB/s)  padded planar NT %.4f ms (%.2f TB/s)\n", b, (void*)d,
               t_k, bytes / 1e9 / t_k, t_nt, bytes / 1e9 / t_nt, t_m, bytes / 1e9 / t_m, t_p, bytes / 1e9 / t_p, t_pn, bytes / 1e9 / t_pn);
    }
    return 0;
}
