// l2_prefetch.hip -- does data touched by ONE launch stay in the XCDs' L2s for the NEXT launch on the stream?
// Kernel `stream` is the access pattern of the one-launch IAF step at 8-pixel rows: 256 workgroups, each reads ALL of a 1.2 MB
// buffer (its stack's weight packs) once, 16 bytes per lane.  Launched round-robin over NBUF different buffers (every launch
// meets a buffer the L2s have not seen for NBUF - 1 launches), as one graph.  Variants:
//   0  plain                           1  same buffer every launch (L2-hot reference)
//   2  every launch also touches 1/32 of the NEXT buffer's 128-byte lines per workgroup, slice = (blockIdx / 8) % 32, at its start
//   3  ... at its end                  4  ... slice = position inside the XCD it really runs on (HW_REG_XCC_ID + a counter)
//   5  ... every workgroup of XCD-position 0 touches ALL lines (one workgroup per XCD does the whole prefetch)
// prints microseconds per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream(const f4* __restrict__ x, int n16, const char* nxt, int nlines, int mode, unsigned* ctr, float* out) {
    const int tid = threadIdx.x;
    int slice = (blockIdx.x >> 3) & 31;
    if (mode == 4 || mode == 5) {
        __shared__ int sl;
        if (tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
            sl = (int)(atomicAdd(ctr + 32 * (xcc & 7), 1u) & 31u);
        }
        __syncthreads();
        slice = sl;
    }
    auto touch = [&]() {
        if (mode < 2) return;
        if (mode == 5) {
            if (slice != 0) return;
            for (int k = tid; k < nlines; k += 256) { unsigned j; asm volatile("global_load_dword %0, %1, off" : "=v"(j) : "v"(nxt + (size_t)k * 128) : "memory"); }
            return;
        }
        const int share = (nlines + 31) >> 5;
        for (int k = tid; k < share; k += 256) {
            int line = slice * share + k;
            line = line < nlines ? line : nlines - 1;
            unsigned j;
            asm volatile("global_load_dword %0, %1, off" : "=v"(j) : "v"(nxt + (size_t)line * 128) : "memory");
        }
    };
    if (mode == 2 || mode == 4 || mode == 5) touch();
    f4 a = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < n16; i += 256) a += x[i];
    if (mode == 3) touch();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.x + a.y + a.z + a.w == 12345.678f) out[blockIdx.x] = a.x;
}

int main(int argc, char** argv) {
    const int NBUF = argc > 1 ? atoi(argv[1]) : 10, REPS = 200;
    const size_t bytes = 1228800;                       // the bf16x3 packs of one stack (n_h = 160, depth_ar = 2)
    std::vector<char*> buf(NBUF);
    for (auto& b : buf) { CK(hipMalloc((void**)&b, bytes)); CK(hipMemset(b, 0, bytes)); }
    unsigned* ctr; CK(hipMalloc((void**)&ctr, 4096)); CK(hipMemset(ctr, 0, 4096));
    float* out; CK(hipMalloc((void**)&out, 4096));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode <= 5; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < NBUF; ++i) {
            const char* cur = mode == 1 ? buf[0] : buf[i];
            const char* nxt = buf[(i + 1) % NBUF];
            hipLaunchKernelGGL(stream, dim3(256), dim3(256), 0, st, (const f4*)cur, (int)(bytes / 16), nxt, (int)(bytes / 128), mode, ctr, out);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d: %.2f us per launch (%d buffers of %.2f MB in turn)\n", mode, ms * 1e3 / (REPS * NBUF), NBUF, bytes / 1e6);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
