// Micro-benchmark: f64 VALU issue, DPP broadcasts and f64 MFMA on gfx950 -- the inputs of the 16-band kernel design.
//   0: 64 x v_fma_f64 (16 independent accumulators)
//   1: 64 x v_fma_f64 + 16 x v_mov_b32 quad_perm:[0,0,0,0]      (pivot broadcast inside a 4-lane group)
//   2: 64 x v_fma_f64 + 16 x v_mov_b32 row_newbcast:3           (pivot broadcast inside a 16-lane row, gfx90a+)
//   3: 16 x v_mfma_f64_16x16x4_f64 (4 independent accumulators)
//   4: 16 x v_mfma_f64_16x16x4_f64 interleaved with 64 x v_fma_f64  (matrix + vector co-issue)
//   5: 64 x v_fma_f64 + 64 x v_mov_b32 quad_perm   (1 : 1, the 16-lanes-per-node ratio)
//   6: 64 x v_fmac_f64_dpp row_newbcast:3, independent accumulators (the broadcast folded into the FMA)
//   7: 16 x [s_nop 1 + 4 x v_fmac_f64_dpp] in the Gauss-Jordan column pattern (two accumulators per block, dependent)
//   8: 64 x v_fmac_f64 (VOP2, no DPP)
//   9: 64 x v_fma_f64 + 32 x v_mov_b64_dpp row_newbcast:3    (the round-2 pivot pattern: 2 moves per 4 FMAs)
//  10: 64 x v_rcp_f64 (independent)                          (the reciprocal of the DOS scan kernels)
//  11: 64 x v_fma_f64 + 8 x v_rcp_f64                         (the scan kernels' ratio before the reciprocals were paired)
// Reports shader clocks per body per wave for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

#define FMA(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b))
#define DPPQ(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "=v"(t[i]) : "v"(s[i]))
#define DPPR(i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t[i]) : "v"(s[i]))
#define MFMA(i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(m[i]) : "v"(a), "v"(b))
#define FMACD(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(a), "v"(b))
#define FMAC2(i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b))
#define COLB(i) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %1, -%3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %0, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[(2 * (i)) % 16]), "+v"(acc[(2 * (i) + 1) % 16]) : "v"(a), "v"(b))
#define MOV64D(i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(acc[i]))
#define RCP(i) asm volatile("v_rcp_f64 %0, %1" : "=v"(u[i]) : "v"(acc[i]))
#define R16(M) M(0); M(1); M(2); M(3); M(4); M(5); M(6); M(7); M(8); M(9); M(10); M(11); M(12); M(13); M(14); M(15)

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(double* out, int iters) {
    double acc[16];
    float s[16], t[16];
    d4 m[4];
    double u[16];
    for (int i = 0; i < 16; ++i) u[i] = 0.0;
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int i = 0; i < 16; ++i) { acc[i] = i; s[i] = (float)(i + threadIdx.x); t[i] = 0.f; }
    for (int i = 0; i < 4; ++i) m[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { R16(FMA); R16(FMA); R16(FMA); R16(FMA); }
        if (MODE == 1) { R16(FMA); R16(DPPQ); R16(FMA); R16(FMA); R16(FMA); }
        if (MODE == 2) { R16(FMA); R16(DPPR); R16(FMA); R16(FMA); R16(FMA); }
        if (MODE == 3) { MFMA(0); MFMA(1); MFMA(2); MFMA(3); MFMA(0); MFMA(1); MFMA(2); MFMA(3); MFMA(0); MFMA(1); MFMA(2); MFMA(3); MFMA(0); MFMA(1); MFMA(2); MFMA(3); }
        if (MODE == 4) {
            MFMA(0); FMA(0); FMA(1); FMA(2); FMA(3); MFMA(1); FMA(4); FMA(5); FMA(6); FMA(7); MFMA(2); FMA(8); FMA(9); FMA(10); FMA(11); MFMA(3); FMA(12); FMA(13); FMA(14); FMA(15);
            MFMA(0); FMA(0); FMA(1); FMA(2); FMA(3); MFMA(1); FMA(4); FMA(5); FMA(6); FMA(7); MFMA(2); FMA(8); FMA(9); FMA(10); FMA(11); MFMA(3); FMA(12); FMA(13); FMA(14); FMA(15);
            MFMA(0); FMA(0); FMA(1); FMA(2); FMA(3); MFMA(1); FMA(4); FMA(5); FMA(6); FMA(7); MFMA(2); FMA(8); FMA(9); FMA(10); FMA(11); MFMA(3); FMA(12); FMA(13); FMA(14); FMA(15);
            MFMA(0); FMA(0); FMA(1); FMA(2); FMA(3); MFMA(1); FMA(4); FMA(5); FMA(6); FMA(7); MFMA(2); FMA(8); FMA(9); FMA(10); FMA(11); MFMA(3); FMA(12); FMA(13); FMA(14); FMA(15);
        }
        if (MODE == 6) { R16(FMACD); R16(FMACD); R16(FMACD); R16(FMACD); }
        if (MODE == 7) { R16(COLB); }
        if (MODE == 8) { R16(FMAC2); R16(FMAC2); R16(FMAC2); R16(FMAC2); }
        if (MODE == 9) { R16(MOV64D); R16(FMA); R16(FMA); R16(MOV64D); R16(FMA); R16(FMA); }
        if (MODE == 10) { R16(RCP); R16(RCP); R16(RCP); R16(RCP); }
        if (MODE == 11) { R16(FMA); RCP(0); RCP(1); R16(FMA); RCP(2); RCP(3); R16(FMA); RCP(4); RCP(5); R16(FMA); RCP(6); RCP(7); }
        if (MODE == 5) { R16(FMA); R16(DPPQ); R16(FMA); R16(DPPQ); R16(FMA); R16(DPPQ); R16(FMA); R16(DPPQ); }
    }
    double r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i] + t[i] + u[i];
    for (int i = 0; i < 4; ++i) r += m[i].x + m[i].y + m[i].z + m[i].w;
    if (out) out[blockIdx.x * NT + threadIdx.x] = r;
}

template <int MODE, int NT>
int run(const char* name, int wg_per_cu, double ghz) {
    const int iters = 20000;
    double* d;
    CK(hipMalloc(&d, sizeof(double) * 256 * 8 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, NT>), dim3(256 * wg_per_cu), dim3(NT), 0, 0, d, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, NT>), dim3(256 * wg_per_cu), dim3(NT), 0, 0, d, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves_per_simd = (double)NT / 64 * wg_per_cu / 4;
    const double clk_per_body_per_simd = ms * 1e-3 * ghz * 1e9 / iters;   // all resident waves of a SIMD together
    printf("%d %-58s %4.1f waves/SIMD: %8.3f ms  %8.1f clk per body per SIMD (all its waves)  = %6.1f clk per body per wave-slot\n", MODE, name,
           waves_per_simd, ms, clk_per_body_per_simd, clk_per_body_per_simd / waves_per_simd);
    CK(hipFree(d));
    return 0;
}

int main() {
    int clk = 0;
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    const double ghz = clk * 1e-6;
    printf("clock %.3f GHz (nominal; the chip may hold less under load)\n", ghz);
#define ALL(NT, W)                                                                  \
    run<0, NT>("64 v_fma_f64", W, ghz);                                              \
    run<1, NT>("64 v_fma_f64 + 16 v_mov_b32 dpp quad_perm", W, ghz);                 \
    run<2, NT>("64 v_fma_f64 + 16 v_mov_b32 dpp row_newbcast", W, ghz);              \
    run<5, NT>("64 v_fma_f64 + 64 v_mov_b32 dpp quad_perm", W, ghz);                 \
    run<3, NT>("16 v_mfma_f64_16x16x4_f64", W, ghz);                                 \
    run<4, NT>("16 v_mfma_f64_16x16x4_f64 interleaved with 64 v_fma_f64", W, ghz);  \
    run<6, NT>("64 v_fmac_f64_dpp row_newbcast", W, ghz);                             \
    run<7, NT>("16 x (s_nop 1 + 4 dependent v_fmac_f64_dpp)", W, ghz);                \
    run<8, NT>("64 v_fmac_f64 (VOP2)", W, ghz);                                       \
    run<9, NT>("64 v_fma_f64 + 32 v_mov_b64_dpp row_newbcast", W, ghz);               \
    run<10, NT>("64 v_rcp_f64", W, ghz);                                              \
    run<11, NT>("64 v_fma_f64 + 8 v_rcp_f64", W, ghz);
    ALL(256, 1)
    ALL(256, 2)
    ALL(256, 4)
    return 0;
}
