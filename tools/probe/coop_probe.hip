// coop_probe.hip -- what does a COOPERATIVE launch (hipLaunchCooperativeKernel: the runtime guarantees that every workgroup of the grid
// is resident at once, or refuses) cost against an ordinary one, and can it be captured into a hipGraph?
// Why: the exchange-form IAF step takes a ticket so that a workgroup only ever waits for a workgroup that is RUNNING (HIP promises no
// dispatch order); the ticket's round trip to the memory side is ~3.3 k cycles of every launch's prologue.  If all 256 workgroups of a
// B = 32 launch are resident by contract, blockIdx can name the rows again and nobody needs a ticket.
// Kernel: 256 x 512 threads, 150 KB of dynamic LDS (one workgroup per CU, like the step), spins ~8 us.
// prints us per launch: ordinary / cooperative, eager and as a graph of 20 launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(512) void spin(float* out, long long ticks) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = (float)threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = sm[1];
}

int main() {
    int dev = 0, coop = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t lds = 150 * 1024;
    CK(hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 512, lds));
    printf("cooperative launch supported: %d; CUs %d; workgroups per CU at 150 KB / 512 threads: %d -> a cooperative grid may hold %d\n", coop, cus, occ, occ * cus);
    float* out; CK(hipMalloc(&out, 4096 * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long long ticks = 800;     // s_memtime ticks at 100 MHz: 8 us
    void* args[] = {&out, &ticks};
    const int grid = 256, N = 200;
    auto ordinary = [&]() { hipLaunchKernelGGL(spin, dim3(grid), dim3(512), lds, st, out, ticks); return hipGetLastError(); };
    auto cooperative = [&]() { return hipLaunchCooperativeKernel((const void*)spin, dim3(grid), dim3(512), args, (unsigned)lds, st); };
    for (int mode = 0; mode < 2; ++mode) {
        auto launch = [&]() { return mode ? cooperative() : ordinary(); };
        hipError_t e = launch();
        if (e != hipSuccess) { printf("%s launch refused: %s\n", mode ? "cooperative" : "ordinary", hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) CK(launch());
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-12s eager: %.2f us per launch\n", mode ? "cooperative" : "ordinary", ms * 1e3 / N);
        hipGraph_t g; hipGraphExec_t ge;
        e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            for (int i = 0; i < 20 && e == hipSuccess; ++i) e = launch();
            hipError_t e2 = hipStreamEndCapture(st, &g);
            if (e == hipSuccess) e = e2;
        }
        if (e != hipSuccess) { printf("%-12s capture refused: %s\n", mode ? "cooperative" : "ordinary", hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        if (e != hipSuccess) { printf("%-12s instantiate refused: %s\n", mode ? "cooperative" : "ordinary", hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-12s graph of 20: %.2f us per launch\n", mode ? "cooperative" : "ordinary", ms * 1e3 / 400);
    }
    return 0;
}
