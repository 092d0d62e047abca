// Micro-benchmark: LDS issue cost of the access patterns of the generic-n row kernels (kernels_generic.hip).
//   0: ds_read_b128, 64 distinct contiguous addresses
//   1: ds_read_b128, one address per 16-lane group (pivot-row read: 4 node slots per wave)
//   2: ds_read_b128, lanes r = 0..15 contiguous, the 4 groups identical (coefficient read)
//   3: ds_write_b128 by one lane of each 16-lane group (pivot-row publish), exec-masked
//   4: ds_read_b64,  one address per 16-lane group
//   5: ds_read_b128, one address per 16-lane group, group stride 512 B (no padding)
//   6: ds_write_b128, all 64 lanes, distinct contiguous      7: ds_write_b128, lane 0 only
//   8: ds_write_b128, lanes 0..15 only                       9: ds_write_b64 by one lane of each 16-lane group
//  10: ds_swizzle_b32 broadcasting one lane of each 16-lane group (no memory access)
// Reports LDS-pipeline clocks per wave instruction per CU (all waves of a CU share one LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

#define RD128(off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(v) : "v"(addr))
#define RD64(off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(w) : "v"(addr))
#define WR128(off) asm volatile("ds_write_b128 %0, %1 offset:" #off : : "v"(addr), "v"(v))
#define WR64(off) asm volatile("ds_write_b64 %0, %1 offset:" #off : : "v"(addr), "v"(w))
#define SWZ(off) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(BITMASK_PERM, \"0000p\")" : "=v"(x) : "v"(x))
#define X16(M) M(0); M(16); M(32); M(48); M(64); M(80); M(96); M(112); M(128); M(144); M(160); M(176); M(192); M(208); M(224); M(240)
#define X16W(M) M(0); M(1024); M(2048); M(3072); M(4096); M(5120); M(6144); M(7168); M(8192); M(9216); M(10240); M(11264); M(12288); M(13312); M(14336); M(15360)

template <int MODE>
__global__ __launch_bounds__(256) void lds_kernel(double* out, int iters) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (double)i;
    __syncthreads();
    unsigned addr;
    if (MODE == 0) addr = lane * 16;
    else if (MODE == 1 || MODE == 3 || MODE == 4) addr = wave * 4096 + (lane / 16) * 576;
    else if (MODE == 2) addr = (lane % 16) * 16;
    else if (MODE == 6) addr = lane * 16;
    else if (MODE == 7 || MODE == 8) addr = wave * 4096 + lane * 16;
    else if (MODE == 9) addr = wave * 4096 + (lane / 16) * 576;
    else addr = wave * 4096 + (lane / 16) * 512;
    d2 v = {1.0, 2.0};
    double w = 0.0;
    float x = (float)lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { X16W(RD128); }
        else if (MODE == 1 || MODE == 5) { X16(RD128); }
        else if (MODE == 2) { X16W(RD128); }
        else if (MODE == 3) { if (lane % 16 == (it & 15)) { X16(WR128); } }
        else if (MODE == 4) { X16(RD64); }
        else if (MODE == 6) { X16W(WR128); }
        else if (MODE == 7) { if (lane == 0) { X16(WR128); } }
        else if (MODE == 8) { if (lane < 16) { X16W(WR128); } }
        else if (MODE == 9) { if (lane % 16 == (it & 15)) { X16(WR64); } }
        else { X16(SWZ); }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    if (out) out[blockIdx.x * 256 + threadIdx.x] = v.x + w + x;
}

template <int MODE>
int run(const char* name, int blocks_per_cu, double ghz) {
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(lds_kernel<MODE>, dim3(blocks), dim3(256), 32768, 0, nullptr, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double instr_per_cu = (double)iters * 16 * 4 * blocks_per_cu;
    printf("%-58s %d waves/CU: %8.3f ms  %6.2f clk per wave-instruction per CU\n", name, 4 * blocks_per_cu, best,
           best * 1e-3 * ghz * 1e9 / instr_per_cu);
    return 0;
}

int main() {
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz * 1e-6;
    printf("clock %.3f GHz\n", ghz);
    for (int bpc : {2, 4}) {
        if (run<0>("0 ds_read_b128 distinct contiguous", bpc, ghz)) return 1;
        if (run<1>("1 ds_read_b128 one address per 16 lanes (stride 576 B)", bpc, ghz)) return 1;
        if (run<5>("5 ds_read_b128 one address per 16 lanes (stride 512 B)", bpc, ghz)) return 1;
        if (run<2>("2 ds_read_b128 16 contiguous x 4 identical groups", bpc, ghz)) return 1;
        if (run<3>("3 ds_write_b128 one lane per 16 (exec-masked)", bpc, ghz)) return 1;
        if (run<4>("4 ds_read_b64 one address per 16 lanes", bpc, ghz)) return 1;
        if (run<6>("6 ds_write_b128 all lanes distinct", bpc, ghz)) return 1;
        if (run<7>("7 ds_write_b128 lane 0 only", bpc, ghz)) return 1;
        if (run<8>("8 ds_write_b128 lanes 0..15 only", bpc, ghz)) return 1;
        if (run<9>("9 ds_write_b64 one lane per 16 (exec-masked)", bpc, ghz)) return 1;
        if (run<10>("10 ds_swizzle_b32 broadcast within 16 lanes", bpc, ghz)) return 1;
    }
    return 0;
}
