// dev probe: what a SIMD gets done next to its v_mfma_f32_16x16x32_bf16 stream.  One workgroup of 4 or 8 waves (one or two per
// SIMD); each wave issues 300000 x {MFMA on rotating accumulators, NV fp32 VALU ops [, an LDS write / a transposing LDS read]};
// prints WALL-CLOCK ns per MFMA of a SIMD (events; s_memtime ticks beside them: their rate changes with the configuration --
// 2393 MHz with one wave per SIMD, 1641 MHz with two -- so ticks only compare within one).  Also: where the waves of a
// workgroup sit (HW_ID), and the chip-wide MFMA rate against waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)1.f; }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    unsigned addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 4096;   // (8 waves x 4 KiB)
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 r = {0, 0};
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[v & 7]));
            if (MODE == 1) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(r) : "memory");
            if (MODE == 2) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int m = 0; m < 3; ++m) s += acc[m][0];
    sink[threadIdx.x] = s + r.x;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
template <int NV, int MODE> void run(unsigned long long* d, float* sink, int threads = 256) {
    const int iters = 100000;
    unsigned long long h = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, MODE>), dim3(1), dim3(threads), 0, 0, d, sink, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NV, MODE>), dim3(1), dim3(threads), 0, 0, d, sink, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double per = 3.0 * iters * (threads / 256);         // MFMAs of one SIMD
    printf("%d wave(s) per SIMD, %s%d VALU per MFMA: %6.2f ns (%5.1f ticks) per MFMA of the SIMD\n", threads / 256,
           MODE == 1 ? "1 LDS write + " : MODE == 2 ? "1 transposing LDS read + " : "", NV, ms * 1e6 / per, (double)h / per);
    hipEventDestroy(e0); hipEventDestroy(e1);
}
// tick calibration + chip-wide MFMA rate: `grid` workgroups of `threads`, 3 x iters MFMAs per wave, wall clock by events
__global__ __launch_bounds__(512) void kcal(unsigned long long* out, float* sink, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)1.f; }
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    sink[(blockIdx.x * blockDim.x + threadIdx.x) & 1023] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
void cal(unsigned long long* d, float* sink, int grid, int threads) {
    const int iters = 100000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kcal, dim3(grid), dim3(threads), 0, 0, d, sink, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kcal, dim3(grid), dim3(threads), 0, 0, d, sink, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double mf = 4.0 * iters, flop = mf * 16 * 16 * 32 * 2 * (threads / 64) * (double)grid;
    printf("grid %4d x %3d threads: %8.3f ms wall, %12llu ticks (%.1f MHz tick), %.1f ticks = %.2f ns per MFMA of a wave, %.1f TFLOP/s\n",
           grid, threads, ms, h, h / (ms * 1e3), h / mf, ms * 1e6 / mf, flop / (ms * 1e-3) / 1e12);
}
// where the waves of a workgroup sit: HW_ID (gfx9: wave [3:0], SIMD [5:4], pipe [7:6], CU [11:8], SH [12], SE [15:13])
__global__ void kwhere(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
void where(int threads, int grid) {
    unsigned* d; hipMalloc(&d, grid * 16 * 4);
    hipMemset(d, 0xff, grid * 16 * 4);
    hipLaunchKernelGGL(kwhere, dim3(grid), dim3(threads), 0, 0, d);
    unsigned h[16 * 4];
    hipMemcpy(h, d, sizeof(unsigned) * 16 * (grid < 4 ? grid : 4), hipMemcpyDeviceToHost);
    for (int b = 0; b < (grid < 4 ? grid : 4); ++b) {
        printf("workgroup %d of %d x %4d threads: (SE.CU.SIMD.slot)", b, grid, threads);
        for (int w = 0; w < threads / 64; ++w) printf("  %u.%u.%u.%u", (h[b * 16 + w] >> 13) & 7, (h[b * 16 + w] >> 8) & 15, (h[b * 16 + w] >> 4) & 3, h[b * 16 + w] & 15);
        printf("\n");
    }
    hipFree(d);
}
int main() {
    where(128, 1); where(256, 2); where(512, 2); where(1024, 1);
    unsigned long long* d; float* sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 4096);
    cal(d, sink, 1, 256); cal(d, sink, 1, 512); cal(d, sink, 256, 256); cal(d, sink, 256, 512); cal(d, sink, 512, 512); cal(d, sink, 1024, 256);
    run<0, 0>(d, sink); run<1, 0>(d, sink); run<2, 0>(d, sink); run<3, 0>(d, sink); run<4, 0>(d, sink); run<6, 0>(d, sink); run<8, 0>(d, sink);
    run<0, 1>(d, sink); run<2, 1>(d, sink); run<0, 2>(d, sink); run<2, 2>(d, sink);
    run<0, 0>(d, sink, 512); run<1, 0>(d, sink, 512); run<2, 0>(d, sink, 512); run<3, 0>(d, sink, 512); run<4, 0>(d, sink, 512); run<6, 0>(d, sink, 512); run<8, 0>(d, sink, 512);
    run<0, 1>(d, sink, 512); run<2, 1>(d, sink, 512); run<0, 2>(d, sink, 512); run<2, 2>(d, sink, 512);
    return 0;
}
