// iaf_wgrad_bf3.hip -- weight gradient of a 3x3 conv on the bf16 matrix cores (bf16x3 split products, fp32-grade):
//     dW[tap][ci][co] = sum_p X[p + shift(tap)][ci] * dY[p][co]           (what TF autodiff derives for layers.py:63-64)
// as the GEMM  D[ci][co] += A[ci][k = pixel] B[k = pixel][co]  on pixel-major X [P][cin] and dY [P][cout] -- the same problem
// iaf_wgrad_wide_kernel (iaf_kernels_backward.hpp) solves with v_mfma_f32_16x16x4_f32 straight from global memory.
//
// Why that kernel's structure does not carry over: v_mfma_f32_16x16x32_bf16 wants 8 consecutive K = PIXELS per lane for one
// channel, and in pixel-major memory those are 8 elements a whole row apart.  Here a K block of 32 pixels is staged in LDS
// pixel-major as it arrives (16-byte loads of 4 channels of one pixel, split into the three bf16 planes, 8-byte LDS writes)
// and the MFMA operands are read with gfx950's transposing LDS read: ds_read_b64_tr_b16 hands each lane of a 16-lane group
// one COLUMN of the [4 pixels][16 channels] block the group's 16 addresses span (probed on the hardware, tools/probe/
// tr_probe.hip: lane i supplies the 8-byte chunk (row i >> 2, columns 4 (i & 3) .. +3) and receives column i of rows 0..3).
//
// Workgroup = (tap, 32 input channels, pixel range, NCOB output tiles), 4 waves; wave w owns input tile (w & 1) and output
// tiles [(w >> 1) NCOB/2, +NCOB/2): no cross-wave reduction.  Two LDS stages: one barrier per K block; the next block's
// global loads are in flight while the current one is multiplied.  Border pixels (the neighbour of tap (dh, dw) outside the
// image) and pixels past the range are staged as zeros.  Output: the same partial layout [range][tap][cin][cout] as the fp32
// kernels, summed by the same reduce launch.
#include "iaf_conv_bf3.hpp"
#include "iaf_hip.h"
#include "iaf_wgrad_types.hpp"

typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ wu32x2 lds_read_tr16_b64(unsigned addr) {
    wu32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
    return r;
}

// s_waitcnt lgkmcnt(N) that the six registers of a fragment set pass THROUGH: the compiler does not know that the outputs of
// the load asm above are not there yet when it "returns", and nothing but a data dependency keeps a register-only consumer
// (the MFMA) behind a bare wait statement (the first version of this kernel multiplied fragments still in flight)
template <int N>
__device__ __forceinline__ void lds_wait_frag(wu32x2 (&f)[3][2]) {
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1])
                 : "n"(N)
                 : "memory");
}

template <int NCOB>
__global__ __launch_bounds__(256) void iaf_wgrad_bf3_kernel(WgradP p) {
    static_assert(NCOB % 2 == 0, "two waves per input tile split the output tiles");
    constexpr int KB = 32;                                   // pixels per K block = one MFMA K
    constexpr int NCO = 16 * NCOB, UW = NCOB / 2;
    constexpr int XROW = 32 * 2 + 16, YROW = NCO * 2 + 16;   // LDS row strides in bytes (+16: rows 4 apart land on different banks)
    constexpr int XPL = KB * XROW, YPL = KB * YROW;          // one plane
    constexpr int STAGE = 3 * (XPL + YPL);
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, range, bz;
    wgrad_decode(p, bx, range, bz);
    const int tap = bx % p.ntaps, cip = bx / p.ntaps;
    const int ci0 = cip * 32, cob = bz * NCO;
    const int dh = p.tap_dh[tap], dw = p.tap_dw[tap];
    const int tapbit = (dh + 1) * 3 + (dw + 1), shift = dh * p.W + dw;
    const int r0 = range * p.px_per_range;
    const int r1 = min(p.P, r0 + p.px_per_range);
    const int nkb = (r1 - r0 + KB - 1) / KB;

    // ---- staging roles: X block = 32 px x 32 ch = 256 16-byte items (one per thread); dY block = 32 px x NCO ch = 32 * 4 NCOB
    // items, NCOB / 2 per thread
    constexpr int YI = NCOB / 2;
    const int xpx = tid >> 3, xq = tid & 7;
    f32x4 xv, yv[YI];
    bool xok;
    bool yok[YI];
    auto load_block = [&](int kb) {
        const int pb = r0 + kb * KB;
        {
            const int pp = pb + xpx;
            const bool in = pp < r1;
            const int pc = in ? pp : r1 - 1;
            xok = in && ((p.tapmask[pc] >> tapbit) & 1);
            int ps = pc + shift;
            ps = ps < 0 ? 0 : (ps >= p.P ? p.P - 1 : ps);                      // a valid address; the value is dropped when !xok
            xv = *(const f32x4*)(p.x + (size_t)ps * p.cin + ci0 + 4 * xq);
        }
#pragma unroll
        for (int i = 0; i < YI; ++i) {
            const int it = tid + 256 * i;
            const int px = it / (4 * NCOB), q = it - px * (4 * NCOB);
            const int pp = pb + px;
            yok[i] = pp < r1;
            const int pc = yok[i] ? pp : r1 - 1;
            yv[i] = *(const f32x4*)(p.dy + (size_t)pc * p.cout + cob + 4 * q);
        }
    };
    auto store_block = [&](int buf) {
        char* xs = wsm + buf * STAGE;
        char* ys = xs + 3 * XPL;
        {
            f32x4 v = xok ? xv : f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned h0, m0, l0, h1, m1, l1;
            bf3_split2(f32x2{v[0], v[1]}, h0, m0, l0);
            bf3_split2(f32x2{v[2], v[3]}, h1, m1, l1);
            char* b = xs + xpx * XROW + xq * 8;
            *(u32x2*)(b) = u32x2{h0, h1};
            *(u32x2*)(b + XPL) = u32x2{m0, m1};
            *(u32x2*)(b + 2 * XPL) = u32x2{l0, l1};
        }
#pragma unroll
        for (int i = 0; i < YI; ++i) {
            const int it = tid + 256 * i;
            const int px = it / (4 * NCOB), q = it - px * (4 * NCOB);
            f32x4 v = yok[i] ? yv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            unsigned h0, m0, l0, h1, m1, l1;
            bf3_split2(f32x2{v[0], v[1]}, h0, m0, l0);
            bf3_split2(f32x2{v[2], v[3]}, h1, m1, l1);
            char* b = ys + px * YROW + q * 8;
            *(u32x2*)(b) = u32x2{h0, h1};
            *(u32x2*)(b + YPL) = u32x2{m0, m1};
            *(u32x2*)(b + 2 * YPL) = u32x2{l0, l1};
        }
    };

    // ---- MFMA roles
    const int ct = wave & 1, j0 = (wave >> 1) * UW;
    const int kg = lane >> 4, i16 = lane & 15;
    // address of this lane's 8-byte chunk inside the [4 px][16 ch] block of its 16-lane group: row i16 >> 2, columns 4 (i16 & 3)
    const unsigned lds0 = (unsigned)(size_t)wsm;
    const unsigned xa = lds0 + (8 * kg + (i16 >> 2)) * XROW + (16 * ct + 4 * (i16 & 3)) * 2;
    const unsigned ya = lds0 + 3 * XPL + (8 * kg + (i16 >> 2)) * YROW + (16 * j0 + 4 * (i16 & 3)) * 2;
    f32x4 acc[UW];
#pragma unroll
    for (int j = 0; j < UW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_block(0);
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        store_block(buf);
        if (kb + 1 < nkb) load_block(kb + 1);                // in flight while this block is multiplied
        __syncthreads();                                     // stage `buf` complete; the other stage's readers passed the previous barrier
        const unsigned so = buf * STAGE;
        // operand fragments through the transposing read; the output tiles' fragments are requested one tile ahead of their
        // MFMAs (two register sets, counted lgkmcnt: the six reads of tile j + 1 stay in flight while tile j multiplies)
        wu32x2 af[3][2], bfr[2][3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            af[pl][0] = lds_read_tr16_b64(xa + so + pl * XPL);
            af[pl][1] = lds_read_tr16_b64(xa + so + pl * XPL + 4 * XROW);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            bfr[0][pl][0] = lds_read_tr16_b64(ya + so + pl * YPL);
            bfr[0][pl][1] = lds_read_tr16_b64(ya + so + pl * YPL + 4 * YROW);
        }
        typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
#define IAF_WFRAG(F, PL) __builtin_bit_cast(bf16x8, wu32x4{F[PL][0].x, F[PL][0].y, F[PL][1].x, F[PL][1].y})
        static_for<UW>([&](auto j_c) {
            constexpr int j = decltype(j_c)::value;
            if constexpr (j + 1 < UW) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    bfr[(j + 1) & 1][pl][0] = lds_read_tr16_b64(ya + so + pl * YPL + 32 * (j + 1));
                    bfr[(j + 1) & 1][pl][1] = lds_read_tr16_b64(ya + so + pl * YPL + 32 * (j + 1) + 4 * YROW);
                }
                lds_wait_frag<6>(bfr[j & 1]);
            } else {
                lds_wait_frag<0>(bfr[j & 1]);
            }
            if constexpr (j == 0) lds_wait_frag<(UW > 1 ? 6 : 0)>(af);      // (requested before the first output tile's: already there)
            const bf16x8 ah = IAF_WFRAG(af, 0), am = IAF_WFRAG(af, 1), al = IAF_WFRAG(af, 2);
            const bf16x8 bh = IAF_WFRAG(bfr[j & 1], 0), bm = IAF_WFRAG(bfr[j & 1], 1), bl = IAF_WFRAG(bfr[j & 1], 2);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[j], 0, 0, 0);
        });
#undef IAF_WFRAG
    }
    // D[m = ci 4 (lane >> 4) + r][n = co lane & 15]
    float* dst = p.part + (((size_t)range * p.ntaps + tap) * p.cin + ci0 + 16 * ct + 4 * kg) * p.cout + cob + 16 * j0 + i16;
#pragma unroll
    for (int j = 0; j < UW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(size_t)r * p.cout + 16 * j] = acc[j][r];
}

template <int NCOB>
static int launch_t(const WgradP& p, hipStream_t st) {
    constexpr int NCO = 16 * NCOB;
    const size_t lds = (size_t)2 * 3 * (32 * (32 * 2 + 16) + 32 * (NCO * 2 + 16));
    static bool raised = false;
    if (!raised && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)iaf_wgrad_bf3_kernel<NCOB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        raised = true;
    }
    hipLaunchKernelGGL(iaf_wgrad_bf3_kernel<NCOB>, dim3((unsigned)(p.gx * p.gz * p.nrange)), dim3(256), lds, st, p);
    return (int)hipGetLastError();
}

extern "C" int iaf_launch_wgrad_bf3(const WgradP* p, int ncob, hipStream_t st) {
    if (!p) return IAF_ERR_NULL;
    if (p->cin % 32 != 0 || ncob <= 0 || p->cout % (16 * ncob) != 0) return IAF_ERR_UNSUPPORTED;
    switch (ncob) {
        case 4: return launch_t<4>(*p, st);
        case 10: return launch_t<10>(*p, st);
        case 12: return launch_t<12>(*p, st);
        case 14: return launch_t<14>(*p, st);
    }
    return IAF_ERR_UNSUPPORTED;
}
