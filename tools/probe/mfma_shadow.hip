// dev probe: how many VALU instructions of the SAME wave fit under one v_mfma_f32_16x16x32_bf16 (16 pipe cycles) for free?
// One workgroup of 4 waves (one per SIMD); each wave issues 3000 x {MFMA on rotating accumulators, NV fp32 VALU ops};
// prints s_memtime ticks per MFMA for NV = 0..8, and the same with an LDS write / a transposing LDS read per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink) {
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
    for (int it = 0; it < 1000; ++it) {
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
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<NV, MODE>), dim3(1), dim3(threads), 0, 0, d, sink);
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    }
    printf("%d waves per SIMD, mode %d, VALU per MFMA %d: %6.1f ticks per MFMA of the SIMD\n", threads / 256, MODE, NV, (double)h / 3000.0 / (threads / 256));
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
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 4096);
    cal(d, sink, 1, 256); cal(d, sink, 1, 512); cal(d, sink, 256, 256); cal(d, sink, 256, 512); cal(d, sink, 512, 512); cal(d, sink, 1024, 256);
    run<0, 0>(d, sink); run<1, 0>(d, sink); run<2, 0>(d, sink); run<3, 0>(d, sink); run<4, 0>(d, sink); run<6, 0>(d, sink); run<8, 0>(d, sink);
    run<0, 1>(d, sink); run<2, 1>(d, sink); run<0, 2>(d, sink); run<2, 2>(d, sink);
    run<0, 0>(d, sink, 512); run<1, 0>(d, sink, 512); run<2, 0>(d, sink, 512); run<3, 0>(d, sink, 512); run<4, 0>(d, sink, 512); run<6, 0>(d, sink, 512); run<8, 0>(d, sink, 512);
    run<0, 1>(d, sink, 512); run<2, 1>(d, sink, 512); run<0, 2>(d, sink, 512); run<2, 2>(d, sink, 512);
    return 0;
}
