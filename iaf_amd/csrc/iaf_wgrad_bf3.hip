// iaf_wgrad_bf3.hip -- weight gradient of a 3x3 conv on the bf16 matrix cores (bf16x3 split products, fp32-grade):
//     dW[tap][ci][co] = sum_p X[p + shift(tap)][ci] * dY[p][co]           (what TF autodiff derives for layers.py:63-64)
// as the GEMM  D[ci][co] += A[ci][k = pixel] B[k = pixel][co]  on pixel-major X [P][cin] and dY [P][cout] -- the same problem
// iaf_wgrad_wide_kernel (iaf_kernels_backward.hpp) solves with v_mfma_f32_16x16x4_f32 straight from global memory.
//
// What bounds that kernel, and the first version of this one (one tap per workgroup, 40 / 127 / 150 us on the layer's convs
// against 38 / 87 / 104 us, profiles/r03/experiments/ab_wgrad.txt): every (tap, 32 input channels) workgroup streams the
// whole dY of its pixel range again -- 360 workgroups x 1024 pixels x 1 KiB = 377 MB out of L2 / MALL for the 160 -> 224 conv,
// 3.6 TB/s at 104 us -- and multiplies for 0.4 us between loads that take longer than that to arrive.  Here a workgroup owns
// a ROW of taps (the up to three (dh, dw) with the same dh): it stages dY once for them, X once with one halo pixel each
// side (the neighbours (dh, -1), (dh, 0), (dh, +1) of 32 consecutive pixels are 34 consecutive pixels), and has three times
// the MFMA work per K block to cover the next block's loads.
//
// Operands: v_mfma_f32_16x16x32_bf16 wants 8 consecutive K = PIXELS per lane for one channel, and in pixel-major memory
// those are 8 elements a whole row apart.  A K block of 32 pixels is staged in LDS pixel-major as it arrives (16-byte loads
// of 4 channels of one pixel, split into the three bf16 planes, 8-byte LDS writes) and the operands are read with gfx950's
// transposing LDS read: ds_read_b64_tr_b16 hands each lane of a 16-lane group one COLUMN of the [4 pixels][16 channels]
// block the group's 16 addresses span (probed on the hardware, tools/probe/tr_probe.hip: lane i supplies the 8-byte chunk
// (row i >> 2, columns 4 (i & 3) .. +3) and receives column i of rows 0..3).  Every lane gives its own address, so the
// one-pixel shift between the taps of a row is free.  Border pixels (the neighbour (dh, dw) of pixel p outside its image) are
// zeroed in the A fragments: lane (kg, .) holds pixels 8 kg .. 8 kg + 7, their tapmask bits become four AND masks per tap.
// Pixels past the range are staged as zero rows of dY.
//
// Workgroup = (tap row, 32 input channels, pixel range, NCOB output tiles), 4 waves; wave w owns input tile (w & 1) and
// output tiles [(w >> 1) NCOB/2, +NCOB/2) for every tap of the row: no cross-wave reduction.  Two LDS stages: one barrier
// per K block.  Output: the same partial layout [range][tap][cin][cout] as the fp32 kernels, summed by the same reduce launch.
#include "iaf_conv_bf3.hpp"
#include "iaf_hip.h"
#include "iaf_wgrad_types.hpp"

typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ wu32x2 lds_read_tr16_b64(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    wu32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
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

// IAF_WSTAMP (tools/probe/wgrad_probe.hip only): phase times of wave 0, summed over the K blocks, in s_memtime ticks
#ifdef IAF_WSTAMP
__device__ __forceinline__ unsigned long long wg_now() {
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define IAF_WST(k) do { const unsigned long long n_ = wg_now(); wst[k] += n_ - wlast; wlast = n_; } while (0)
#else
#define IAF_WST(k) do { } while (0)
#endif

template <int NCOB> struct WgGeom {
    static constexpr int KB = 32;                                   // pixels per K block = one MFMA K
    static constexpr int XR = KB + 2;                               // staged X rows: the dw = -1 .. +1 neighbours of KB pixels
    static constexpr int NCO = 16 * NCOB, UW = NCOB / 2;
    static constexpr int XROW = 32 * 2 + 16, YROW = NCO * 2 + 16;   // LDS row strides in bytes (+16: rows 4 apart on different banks)
    static constexpr int XPL = XR * XROW, YPL = KB * YROW;          // one plane
    static constexpr int STAGE = 3 * (XPL + YPL);
};

// TG = taps in this workgroup's row
template <int NCOB, int TG>
__device__ __forceinline__ void wgrad_bf3_row(const WgradP& p, char* wsm, int grp, int cip, int range, int bz) {
    using G = WgGeom<NCOB>;
    constexpr int KB = G::KB, NCO = G::NCO, UW = G::UW, XROW = G::XROW, YROW = G::YROW, XPL = G::XPL, YPL = G::YPL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = cip * 32, cob = bz * NCO;
    const int dh = p.grp_dh[grp];
    int tap[TG], dw[TG], tapbit[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        tap[t] = p.grp_tap[grp][t];
        dw[t] = p.tap_dw[tap[t]];
        tapbit[t] = (dh + 1) * 3 + (dw[t] + 1);
    }
    const int r0 = range * p.px_per_range;
    const int r1 = min(p.P, r0 + p.px_per_range);
    const int nkb = (r1 - r0 + KB - 1) / KB;
    const int kg = lane >> 4, i16 = lane & 15;

    // ---- staging roles.  Thread (row = tid >> 3, c = tid & 7) owns pixel row `row` of the K block: 16-byte chunk c of its 32 X
    // channels and chunks c, c + 8, ... of its NCO dY channels (4 NCOB / 8 = UW of them): every address of an item is the
    // thread's base plus a constant.  X has two halo rows more (34 = the dw = -1 .. +1 neighbours of 32 pixels): threads 0..15.
    // Addresses are uniform base + 32-bit byte offset, advanced by a constant per K block and clamped into the tensor (the
    // launcher checks that the tensors stay below 2 GiB): a neighbour outside the tensor has its tapmask bit clear, a dY row
    // past the range is staged as zeros -- any valid address will do for them.
    const int row = tid >> 3, c8 = tid & 7;
    f32x4 xv, xv2, yv[UW];
    wu32x4 mm;
    const int xstep = KB * p.cin * 4, ystep = KB * p.cout * 4;
    const int xlim = (p.P * p.cin - 4) * 4, ylim = (p.P * p.cout - 4 - 32 * (UW - 1)) * 4;
    const int xoff0 = ((r0 + dh * p.W - 1 + row) * p.cin + ci0 + 4 * c8) * 4;             // staged row r = source pixel pb + dh W - 1 + r
    const int hoff0 = ((r0 + dh * p.W - 1 + KB + (row & 1)) * p.cin + ci0 + 4 * c8) * 4;  // halo rows: every thread loads, 0..15 store
    const int yoff0 = ((r0 + row) * p.cout + cob + 4 * c8) * 4;
    auto ldx_at = [&](int off) {
        off = off < 0 ? 0 : (off > xlim ? xlim : off);
        return *(const f32x4*)((const char*)p.x + (unsigned)off);
    };
    auto load_block = [&](int kb) {
        xv = ldx_at(xoff0 + kb * xstep);
        xv2 = ldx_at(hoff0 + kb * xstep);
        int yo = yoff0 + kb * ystep;
        yo = yo > ylim ? ylim : yo;
        const char* yp = (const char*)p.dy + (unsigned)yo;
#pragma unroll
        for (int i = 0; i < UW; ++i) yv[i] = *(const f32x4*)(yp + 128 * i);
        const int pm = r0 + kb * KB + 8 * kg;                      // P % 8 == 0 (checked by the launcher): all 8 or none
        const wu32x4 v = *(const wu32x4*)((const char*)p.tapmask + (unsigned)(2 * (pm < p.P ? pm : p.P - 8)));
        mm = pm < p.P ? v : wu32x4{0u, 0u, 0u, 0u};
    };
    auto split_store = [&](char* b, int plane, f32x4 v) {
        unsigned h0, m0, l0, h1, m1, l1;
        bf3_split2(f32x2{v[0], v[1]}, h0, m0, l0);
        bf3_split2(f32x2{v[2], v[3]}, h1, m1, l1);
        *(u32x2*)(b) = u32x2{h0, h1};
        *(u32x2*)(b + plane) = u32x2{m0, m1};
        *(u32x2*)(b + 2 * plane) = u32x2{l0, l1};
    };
    char* const xs = wsm + row * XROW + c8 * 8;
    char* const ys = wsm + 3 * XPL + row * YROW + c8 * 8;
    auto store_block = [&](int kb) {
        split_store(xs, XPL, xv);
        if (tid < 16) split_store(xs + KB * XROW, XPL, xv2);
        if (r0 + (kb + 1) * KB <= r1) {                            // (uniform) a block inside the range: no selects
#pragma unroll
            for (int i = 0; i < UW; ++i) split_store(ys + 64 * i, YPL, yv[i]);
        } else {
            const bool ok = r0 + kb * KB + row < r1;               // rows past the range: zeros
#pragma unroll
            for (int i = 0; i < UW; ++i) split_store(ys + 64 * i, YPL, ok ? yv[i] : f32x4{0.f, 0.f, 0.f, 0.f});
        }
    };

    // ---- MFMA roles
    const int ct = wave & 1, j0 = (wave >> 1) * UW;
    // address of this lane's 8-byte chunk inside the [4 px][16 ch] block of its 16-lane group: row i16 >> 2, columns 4 (i16 & 3)
    const unsigned lds0 = (unsigned)(size_t)wsm;
    unsigned xa[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t)
        xa[t] = lds0 + (8 * kg + (i16 >> 2) + 1 + dw[t]) * XROW + (16 * ct + 4 * (i16 & 3)) * 2;
    const unsigned ya = lds0 + 3 * XPL + (8 * kg + (i16 >> 2)) * YROW + (16 * j0 + 4 * (i16 & 3)) * 2;
    f32x4 acc[TG][UW];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int j = 0; j < UW; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifdef IAF_WSTAMP
    unsigned long long wst[4] = {0, 0, 0, 0}, wlast = wg_now();
#endif
    // ---- K loop.  ONE LDS stage and a split phase / MFMA phase per block: what overlaps them is the OTHER workgroup of the CU
    // (the stage and the registers are sized for two per CU).  Measured in wall-clock time per MFMA of a SIMD (tools/probe/
    // mfma_shadow.hip, profiles/r03/experiments/mfma_shadow.txt): one wave alone issues an MFMA every 9.4 ns in a rolled loop
    // (7.3 ns = the pipe's rate, 2.3 PFLOP/s over 1024 SIMDs, unrolled); each VALU instruction it issues between two MFMAs adds
    // 1.1 - 1.6 ns (they do not hide under its own MFMAs), a transposing LDS read 8 ns, an LDS write 13 ns.  With a second wave
    // on the SIMD the read costs 1 ns, the write 3.6 ns, (MFMA + 2 VALU) 7 - 10 ns instead of 11.6.  A software pipeline inside
    // one wave (the next block's split interleaved with this block's MFMAs, two LDS stages, one workgroup per CU) measured
    // 46.4 us where the plain two-phase loop took 48.2 (160 -> 224 conv, B = 32, 16 x 16); on the 160 -> 448 conv this loop takes
    // 87 us with one workgroup per CU and 71.5 us with two (the fp32 MFMA kernel: 104 us).
    // The next block's loads are issued before the MFMA phase and land under it.
    if (nkb > 0) load_block(0);
    for (int kb = 0; kb < nkb; ++kb) {
        IAF_WST(3);
        store_block(kb);
        // border masks of this block's A fragments (before the next block's loads overwrite mm): dword d of a fragment holds
        // pixels 8 kg + 2 d (low half) and + 2 d + 1 (high half) = the two 16-bit tapmask entries of mm[d]
        unsigned mk[TG][4];
#pragma unroll
        for (int t = 0; t < TG; ++t)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int lo = (int)(mm[d] << (31 - tapbit[t])) >> 31;
                const int hi = (int)(mm[d] << (15 - tapbit[t])) >> 31;
                mk[t][d] = ((unsigned)lo & 0xFFFFu) | ((unsigned)hi & 0xFFFF0000u);
            }
        load_block(kb + 1);                                  // (clamped past the end: loaded, never staged)
        IAF_WST(0);
        __syncthreads();                                     // the stage is complete
        IAF_WST(1);
        // operand fragments through the transposing read; the output tiles' fragments are requested one tile ahead of their
        // MFMAs (two register sets, counted lgkmcnt: the six reads of tile j + 1 stay in flight while tile j multiplies)
        wu32x2 af[TG][3][2], bfr[2][3][2];
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const unsigned a = xa[t];
            af[t][0][0] = lds_read_tr16_b64<0>(a);
            af[t][0][1] = lds_read_tr16_b64<4 * XROW>(a);
            af[t][1][0] = lds_read_tr16_b64<XPL>(a);
            af[t][1][1] = lds_read_tr16_b64<XPL + 4 * XROW>(a);
            af[t][2][0] = lds_read_tr16_b64<2 * XPL>(a);
            af[t][2][1] = lds_read_tr16_b64<2 * XPL + 4 * XROW>(a);
        }
        auto read_b = [&](auto j_c, wu32x2 (&f)[3][2]) {
            constexpr int o = 32 * decltype(j_c)::value;
            f[0][0] = lds_read_tr16_b64<o>(ya);
            f[0][1] = lds_read_tr16_b64<o + 4 * YROW>(ya);
            f[1][0] = lds_read_tr16_b64<o + YPL>(ya);
            f[1][1] = lds_read_tr16_b64<o + YPL + 4 * YROW>(ya);
            f[2][0] = lds_read_tr16_b64<o + 2 * YPL>(ya);
            f[2][1] = lds_read_tr16_b64<o + 2 * YPL + 4 * YROW>(ya);
        };
        read_b(std::integral_constant<int, 0>{}, bfr[0]);
#define IAF_WFRAG(F, PL) __builtin_bit_cast(bf16x8, wu32x4{F[PL][0].x, F[PL][0].y, F[PL][1].x, F[PL][1].y})
        static_for<UW>([&](auto j_c) {
            constexpr int j = decltype(j_c)::value;
            if constexpr (j + 1 < UW) {
                read_b(std::integral_constant<int, j + 1>{}, bfr[(j + 1) & 1]);
                lds_wait_frag<6>(bfr[j & 1]);
            } else {
                lds_wait_frag<0>(bfr[j & 1]);
            }
            if constexpr (j == 0) {                          // (requested before the first output tile's: they are there too)
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    lds_wait_frag<(UW > 1 ? 6 : 0)>(af[t]);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        af[t][pl][0].x &= mk[t][0]; af[t][pl][0].y &= mk[t][1];
                        af[t][pl][1].x &= mk[t][2]; af[t][pl][1].y &= mk[t][3];
                    }
                }
            }
            const bf16x8 bh = IAF_WFRAG(bfr[j & 1], 0), bm = IAF_WFRAG(bfr[j & 1], 1), bl = IAF_WFRAG(bfr[j & 1], 2);
            // the six part products, small terms first; the taps' accumulators alternate so that no MFMA waits for the one before
#define IAF_WPP(AP, BV)                                                                                        \
            _Pragma("unroll") for (int t = 0; t < TG; ++t)                                                     \
                acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(IAF_WFRAG(af[t], AP), BV, acc[t][j], 0, 0, 0);
            IAF_WPP(2, bh) IAF_WPP(0, bl) IAF_WPP(1, bm) IAF_WPP(1, bh) IAF_WPP(0, bm) IAF_WPP(0, bh)
#undef IAF_WPP
        });
#undef IAF_WFRAG
        IAF_WST(2);
        __syncthreads();                                     // everyone is done reading the stage
    }
#ifdef IAF_WSTAMP
    if (p.dbg && tid == 0)
        for (int k = 0; k < 4; ++k) p.dbg[(size_t)blockIdx.x * 4 + k] = wst[k];
#endif
    // D[m = ci 4 (lane >> 4) + r][n = co lane & 15]
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        float* dst = p.part + (((size_t)range * p.ntaps + tap[t]) * p.cin + ci0 + 16 * ct + 4 * kg) * p.cout + cob + 16 * j0 + i16;
#pragma unroll
        for (int j = 0; j < UW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)r * p.cout + 16 * j] = acc[t][j][r];
    }
}

template <int NCOB>
__global__ __launch_bounds__(256, 2) void iaf_wgrad_bf3_kernel(WgradP p) {
    static_assert(NCOB % 2 == 0, "two waves per input tile split the output tiles");
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    int bx, range, bz;
    wgrad_decode(p, bx, range, bz);
    const int grp = bx % p.ngroups, cip = bx / p.ngroups;
    switch (p.grp_n[grp]) {                                  // (uniform over the workgroup)
        case 3: wgrad_bf3_row<NCOB, 3>(p, wsm, grp, cip, range, bz); break;
        case 2: wgrad_bf3_row<NCOB, 2>(p, wsm, grp, cip, range, bz); break;
        default: wgrad_bf3_row<NCOB, 1>(p, wsm, grp, cip, range, bz); break;
    }
}

template <int NCOB>
static int launch_t(const WgradP& p, hipStream_t st) {
    const size_t lds = (size_t)WgGeom<NCOB>::STAGE;
    static bool raised = false;
    if (!raised && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)iaf_wgrad_bf3_kernel<NCOB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        raised = true;
    }
    hipLaunchKernelGGL(iaf_wgrad_bf3_kernel<NCOB>, dim3((unsigned)(p.gx * p.gz * p.nrange)), dim3(256), lds, st, p);
    return (int)hipGetLastError();
}

extern "C" int iaf_wgrad_bf3_ncob(int cin, int cout) {
    if (cin <= 0 || cout <= 0 || cin % 32 != 0 || cout % 16 != 0) return 0;
    static const int cand[4] = {14, 12, 10, 4};
    for (int c : cand)
        if ((cout / 16) % c == 0) return c;
    return 0;
}

extern "C" int iaf_launch_wgrad_bf3(const WgradP* p, int ncob, hipStream_t st) {
    if (!p) return IAF_ERR_NULL;
    if (p->cin % 32 != 0 || ncob <= 0 || p->cout % (16 * ncob) != 0 || p->P % 8 != 0 || p->px_per_range % 32 != 0 ||
        (long long)p->P * (p->cin > p->cout ? p->cin : p->cout) * 4 >= (1LL << 31) ||
        p->ngroups <= 0 || p->gx != p->ngroups * (p->cin / 32))
        return IAF_ERR_UNSUPPORTED;
    switch (ncob) {
        case 4: return launch_t<4>(*p, st);
        case 10: return launch_t<10>(*p, st);
        case 12: return launch_t<12>(*p, st);
        case 14: return launch_t<14>(*p, st);
    }
    return IAF_ERR_UNSUPPORTED;
}
