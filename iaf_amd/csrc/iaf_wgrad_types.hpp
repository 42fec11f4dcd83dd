// iaf_wgrad_types.hpp -- launch descriptor of the weight-gradient GEMM kernels (iaf_kernels_backward.hpp: fp32 MFMA;
// iaf_wgrad_bf3.hip: bf16x3 on the bf16 matrix cores), shared between their translation units.
#pragma once
#include <hip/hip_runtime.h>
#ifndef MAXTAPS
#define MAXTAPS 9
#endif

struct WgradP {
    const float* x;      // [P][cin]
    const float* dy;     // [P][cout]
    float* part;         // [nrange][ntaps][cin][cout]
    int B, H, W, HW, P, cin, cout, nrange, px_per_range;
    int ntaps;           // 5 (masked) or 9 (plain)
    int tap_dh[MAXTAPS], tap_dw[MAXTAPS];
    const unsigned short* tapmask;   // [P]: bit (dh+1)*3+(dw+1) set when pixel p's neighbour (dh,dw) lies inside its image
    int gx, gz;          // workgroups per pixel range: gx = ntaps * ceil(cin/32) operand blocks, gz output-channel blocks
    // iaf_wgrad_bf3.hip only: the taps grouped into rows of equal dh (gx = ngroups * cin/32 there)
    int ngroups, grp_n[3], grp_dh[3], grp_tap[3][3];
#ifdef IAF_WSTAMP
    unsigned long long* dbg;     // [grid][4] phase ticks (tools/probe/wgrad_probe.hip)
#endif
};

// Workgroup -> (operand block x, pixel range, output block z), 1-D grid, the workgroups of a pixel range adjacent.
// (Dealing whole pixel ranges to each XCD -- workgroup i runs on XCD i % 8, each XCD has its own 4 MB L2 -- so that an
// L2 only ever sees 1/8 of the pixels was measured and changed nothing: the operand streams are not L2-capacity bound.)
__device__ __forceinline__ void wgrad_decode(const WgradP& p, int& x, int& range, int& z) {
    const int per_range = p.gx * p.gz;
    const int v = blockIdx.x;
    range = v / per_range;
    const int rem = v - range * per_range;
    z = rem / p.gx;
    x = rem - z * p.gx;
}


// The same GEMM on the bf16 matrix cores (iaf_wgrad_bf3.hip): ncob = output tiles per workgroup (iaf_wgrad_bf3_ncob: 4, 10, 12 or
// 14 dividing cout / 16, cin a multiple of 32; 0 = this conv is not covered); P % 8 == 0, px_per_range % 32 == 0, the tap rows
// filled in; grid = p.gx * p.gz * p.nrange workgroups of 256 threads.  Returns a hipError_t / IAF status.
extern "C" int iaf_wgrad_bf3_ncob(int cin, int cout);
extern "C" int iaf_launch_wgrad_bf3(const WgradP* p, int ncob, hipStream_t st);
