// dev probe: what ds_read_b64_tr_b16 (gfx950) returns to each lane.  LDS is filled with u16 values = their own element
// index; lane l passes byte address addr(l); prints the four u16 each lane receives, for two address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short* out, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (pattern == 0) addr = 8u * l;                                            // lane l -> chunk l (4 consecutive u16)
    else if (pattern == 1) addr = 8u * (l & 3) + 64u * ((l >> 2) & 3) + 1024u * (l >> 4);   // 4 rows of 64 B per 16 lanes
    else addr = 2u * ((l & 15) * 0) + 32u * ((l & 15) >> 2) + 8u * (l & 3) + 512u * (l >> 4);  // 4 rows of 32 B ([4][16] block)
    unsigned base = (unsigned)(size_t)lds;   // LDS offset (low 32 bits of the generic pointer are the LDS address on amdgcn)
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(base + addr) : "memory");
    out[l * 4 + 0] = r.x & 0xffff; out[l * 4 + 1] = r.x >> 16; out[l * 4 + 2] = r.y & 0xffff; out[l * 4 + 3] = r.y >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int pat = 0; pat < 3; ++pat) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pat);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l & 3) == 3 ? "\n" : "   ");
    }
    return 0;
}
