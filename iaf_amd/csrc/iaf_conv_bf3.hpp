// iaf_conv_bf3.hpp -- the masked 3x3 conv of the IAF step on the bf16 matrix cores at fp32 accuracy ("bf16x3").
//
// Why: the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, iaf_conv_kernel.hpp) runs at the fp32 VECTOR rate, 1/16 of the bf16
// MFMA rate.  Every fp32 operand is the exact sum of three bf16 numbers, x = xh + xm + xl (8 significand bits each,
// |xm| <= 2^-9|x|, |xl| <= 2^-18|x|), so a product of two fp32 numbers is the sum of nine bf16 products; the three
// smallest (m*l, l*m, l*l <= 2^-27 relative) are below fp32 resolution and are dropped.  The remaining SIX products
//     w_h x_h + w_h x_m + w_m x_h + w_h x_l + w_l x_h + w_m x_m
// are accumulated in the fp32 accumulator of v_mfma_f32_16x16x32_bf16: 6 MFMAs of 16 cycles per K = 32 instead of 8 of
// 32 cycles -- 2.7x the matrix-core throughput.  Products of bf16 pairs are exact in fp32, and the accumulation is fp32
// like the exact path's, so the result carries fp32-grade error (measured against an fp64 evaluation: not larger than the
// fp32 MFMA chain's, tests/test_hip_bf3.py and docs/LAB_NOTEBOOK_r01-r03.md 4.7).  Masked weights are exactly zero in all three planes, so
// the autoregressive structure stays bit-exact.
//
// Data flow (differs from the fp32 kernel because the operand rate per MFMA cycle is ~4x higher):
//   weights      prep kernel writes three bf16 planes in fragment order  [step = (c_in pair of 32, tap)][co tile][plane]
//                [lane 64][8 bf16]: one (step, tile, plane) fragment = 1 KiB = one global_load_dwordx4 per wave.
//   activations  staged once per workgroup into LDS as three bf16 planes per pixel slot ([slot][plane][c_in], +16 B pad:
//                slot stride = 8*odd dwords, ds_read_b128 conflict-free); split fp32 -> 3 x bf16 while staging.
//   tiling       workgroup = PXT x WCO x KS waves (WCO co groups share one staged activation tile).  A wave owns PPW pixel tiles (16 px) x NT co tiles (16 ch) -- every weight
//                fragment it fetches is used by PPW pixel tiles (register blocking along pixels: the weight stream into a
//                CU, not the MFMA pipe, is the limit at BASELINE batch sizes) -- for ONE slice of the K steps; the KS
//                waves of a group split the steps of the same tile (nothing shared, no barrier in the K loop) and
//                exchange partial sums through LDS at the end.
//   K loop       steps (pair, tap) with a register ring of RD+1 step slots; refills are clamped to the wave's last step
//                so every body is branch-free straight-line code (counted s_waitcnt).
// Epilogues: the same fused epilogues as the fp32 kernel (bias + context + ELU -> pixel-major scratch; output pair ->
// affine transform + log-det term / posterior KL elements), shared code in iaf_conv_epilogue.hpp.
#pragma once
#include "iaf_conv_epilogue.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x = h + m + l exactly up to 2^-27|x|; each part a bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32)
__device__ __forceinline__ void bf3_split2(f32x2 x, unsigned& h, unsigned& m, unsigned& l) {
    const bf16x2 hb = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(hb, f32x2);
    const bf16x2 mb = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(mb, f32x2);
    const bf16x2 lb = __builtin_convertvector(r2, bf16x2);
    h = __builtin_bit_cast(unsigned, hb);
    m = __builtin_bit_cast(unsigned, mb);
    l = __builtin_bit_cast(unsigned, lb);
}

// 4 consecutive channels of one pixel slot -> the three planes of the LDS tile (8 bytes each)
__device__ __forceinline__ void bf3_store4(char* smem, int slot, int q, f32x4 v, int s16, int cin8) {
    unsigned h0, m0, l0, h1, m1, l1;
    bf3_split2(f32x2{v[0], v[1]}, h0, m0, l0);
    bf3_split2(f32x2{v[2], v[3]}, h1, m1, l1);
    char* base = smem + ((size_t)slot * s16 << 4) + q * 8;
    *(u32x2*)(base) = u32x2{h0, h1};
    *(u32x2*)(base + ((size_t)cin8 << 4)) = u32x2{m0, m1};
    *(u32x2*)(base + ((size_t)cin8 << 5)) = u32x2{l0, l1};
}

// ---- "f16x2": TWO fp16 planes per fp32 operand (round 6; iaf_step_fused.hpp F16) -------------------------------------------
// x = h + l 2^-11 with h = fp16(x) (round to nearest even), l = fp16((x - h) 2^11): x - h is exact in fp32 and at most half an ulp of h,
// so l sits in the binade of x or below -- in fp16's NORMAL range wherever h is (a plain fp16(x - h) would fall into the
// subnormals from |x| < 2^-3 on and lose its bits: tests/studies/f16_split_study.py).  22 significand bits + sign handling; the
// product of two such operands is  h h' + (h l' + l h') 2^-11  up to 2^-22 relative: THREE products on v_mfma_f32_16x16x32_f16 (each
// exact in fp32: 11 + 11 bits), the two cross products in an accumulator of their own that is scaled once, where the sums meet.
// Range: |x| > 65504 has no fp16 -- h = inf, the outputs carry inf / NaN, and the kernels raise a host-visible word (StepP::rng_err) on
// which the stack returns to the bf16x3 kernels, whose planes have fp32's exponent range.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define IAF_F16_MAX 65504.0f
__device__ __forceinline__ void f16s_split2(f32x2 x, unsigned& h, unsigned& l) {
    const f16x2 hb = __builtin_convertvector(x, f16x2);
    const f32x2 r = (x - __builtin_convertvector(hb, f32x2)) * 2048.0f;
    const f16x2 lb = __builtin_convertvector(r, f16x2);
    h = __builtin_bit_cast(unsigned, hb);
    l = __builtin_bit_cast(unsigned, lb);
}
// 4 consecutive channels of one pixel slot -> the two planes of the LDS tile (8 bytes each)
__device__ __forceinline__ void f16s_store4(char* smem, int slot, int q, f32x4 v, int s16, int cin8) {
    unsigned h0, l0, h1, l1;
    f16s_split2(f32x2{v[0], v[1]}, h0, l0);
    f16s_split2(f32x2{v[2], v[3]}, h1, l1);
    char* base = smem + ((size_t)slot * s16 << 4) + q * 8;
    *(u32x2*)(base) = u32x2{h0, h1};
    *(u32x2*)(base + ((size_t)cin8 << 4)) = u32x2{l0, l1};
}

// NTP_: taps of the filter -- the 5 live ones of a MADE-masked conv, or all 9 of a plain conv2d (EPI_PLAIN: the convs
// around the IAF step, tf_train.py:36,41,53,93; halo on both sides of the pixel tile)
// S2: the two strided convs of the downsampling IAFLayer at their minimal work, on the 9-tap pack as it is:
//   1  conv2d(stride 2, SAME) (tf_train.py:33,36; layers.py:31-64): out[i,j] = sum_{di,dj} x[2i+di, 2j+dj] V[di,dj].  With the four phase
//      images x_ab[i][j] = x[2i+a][2j+b] staged side by side in LDS, tap (di,dj) is a unit-stride tap (di>>1, dj>>1) into phase
//      (di&1, dj&1): nine steps per c_in pair as for a stride-1 conv, on a quarter of the pixels.  p.H, p.W = the OUTPUT grid.
//   2  deconv2d(stride 2, SAME) (tf_train.py:89-91; layers.py:83-112) = the stride-1 conv of the zero-inserted input with the
//      rotated filter (iaf_kernels_resample.hpp): output phase (a,b) = pixels (2i+a, 2j+b) only meets taps with di = a (mod 2)...
//      precisely di in {0,2} (rows i-1, i) when a = 0, di = 1 (row i) when a = 1 -- 4, 2, 2, 1 taps for the four phases, nine in
//      all.  blockIdx.z = the phase; the staged tile is the ordinary low-resolution one.  p.H, p.W = the INPUT grid.
// F16 = 1 (round 6): the operands as TWO fp16 planes and three part-products per step ("f16x2", see f16s_split2 above and
// iaf_step_fused.hpp) -- the forward plain convs (EPI_PLAIN, NCHW input: activations O(1), weight-normed filters); the data gradients keep
// the bf16 planes (a gradient tensor's magnitudes have no business with fp16's exponent range).  p.wp = the two-plane pack.
template <int NT, int PPW, int PXT, int KS, int INMODE, int EPI, int WCO = 1, int NTP_ = NTAPS, int S2 = 0, int F16 = 0>
__global__ __launch_bounds__(64 * PXT * KS * WCO) void iaf_conv_bf3_kernel(ConvP p) {
    static_assert(!F16 || INMODE != IN_FUSED0, "the fused first layer exists on bf16 planes only");
    // F16 with a pixel-major input = the DATA GRADIENT on two fp16 planes (round 6): a gradient tensor has no business with fp16's exponent
    // range, so the workgroup scales the tile it stages by a power of two of its own -- the largest magnitude of the tile lands in
    // [2^13, 2^14) -- and its sums by the inverse behind the K loop: nothing can leave fp16's range (no range word), an element's error is
    // 2^-22 of itself down to 2^-14 of the tile's largest and 2^-36 of that largest below (fp16's subnormals), which is what the gradient
    // tests hold (errors relative to the tensor's largest element).
    constexpr bool DG16 = F16 && INMODE == IN_PIXMAJOR;
    [[maybe_unused]] float dg_scale = 1.0f, dg_inv = 1.0f;
    extern __shared__ __attribute__((aligned(16))) f32x4 smem4[];
    char* smem = (char*)smem4;
    constexpr int NTP = NTP_;
    constexpr int NPL = F16 ? 2 : 3;                             // planes per operand
    constexpr int TM = 16 * PPW * PXT;
    constexpr int NTHREADS = 64 * PXT * KS * WCO;   // WCO co groups share ONE staged activation tile
#ifndef IAF_F16_PLAIN_RD
#define IAF_F16_PLAIN_RD 1
#endif
    // ring look-ahead in steps (a step is PPW*NT*6 MFMAs of 16 cycles; on the fp16 planes PPW*NT*3: IAF_F16_PLAIN_RD, dev knob)
    constexpr int RD = F16 ? IAF_F16_PLAIN_RD : (PPW >= 2) ? 1 : 2;
    constexpr int U = RD + 1;                                   // ring slots
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wave % PXT, cw = (wave / PXT) % WCO, kh = wave / (PXT * WCO);
    const int grp = cw * PXT + pw;         // the KS waves with the same (pixel block, co group) exchange partial sums
    const int P0 = blockIdx.x * TM;
    const int cot0 = (blockIdx.y * WCO + cw) * NT;
    asm volatile("" ::"s"(p.x), "s"(p.wp), "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.HW), "s"(p.P), "s"(p.cin), "s"(p.cout),
                 "s"(p.nchunk), "s"(p.ncot), "s"(p.nslot), "s"(p.mode), "s"(p.halo_before));
    asm volatile("" ::"s"(p.tap_dh[0]), "s"(p.tap_dh[1]), "s"(p.tap_dh[2]), "s"(p.tap_dh[3]), "s"(p.tap_dh[4]), "s"(p.tap_dw[0]),
                 "s"(p.tap_dw[1]), "s"(p.tap_dw[2]), "s"(p.tap_dw[3]), "s"(p.tap_dw[4]));
    if constexpr (NTP > NTAPS)
        asm volatile("" ::"s"(p.tap_dh[5]), "s"(p.tap_dh[6]), "s"(p.tap_dh[7]), "s"(p.tap_dh[8]), "s"(p.tap_dw[5]), "s"(p.tap_dw[6]),
                     "s"(p.tap_dw[7]), "s"(p.tap_dw[8]));
    if constexpr (EPI == EPI_PLAIN) asm volatile("" ::"s"(p.x2), "s"(p.res), "s"(p.c_split), "s"(p.in_elu), "s"(p.nsplit));
    if constexpr (EPI == EPI_DGRAD) asm volatile("" ::"s"(p.res), "s"(p.nsplit), "s"(p.qm), "s"(p.ql));
    asm volatile("" ::"s"(p.bias), "s"(p.ctx), "s"(p.ctx2), "s"(p.y), "s"(p.zin), "s"(p.out0), "s"(p.out1), "s"(p.border));
    if constexpr (INMODE == IN_POSTERIOR || EPI == EPI_OUT)
        asm volatile("" ::"s"(p.qm), "s"(p.ql), "s"(p.rm), "s"(p.rl), "s"(p.pm), "s"(p.pl), "s"(p.eps), "s"(p.kl_elem));
    const int HW = p.HW, W = p.W;
    // S2 = 2: this workgroup's output phase and its tap list: tap k of the list = (di, dj) = (a ? 1 : 2 (k >> lgb), b ? 1 : 2 (k & nb-1))
    const int ph_a = S2 == 2 ? (int)(blockIdx.z >> 1) : 0, ph_b = S2 == 2 ? (int)(blockIdx.z & 1) : 0;
    const int lgb = ph_b ? 0 : 1, lgt = (ph_a ? 0 : 1) + lgb;      // log2 of the list's length
    auto step_pair = [&](int sc) __attribute__((always_inline)) -> int { return S2 == 2 ? sc >> lgt : sc / NTP; };
    auto step_tap = [&](int sc, int pair) __attribute__((always_inline)) -> int {                 // the tap of the 9-tap pack step sc multiplies
        if constexpr (S2 == 2) {
            const int k = sc & ((1 << lgt) - 1);
            const int di = ph_a ? 1 : 2 * (k >> lgb), dj = ph_b ? 1 : 2 * (k & ((1 << lgb) - 1));
            return di * 3 + dj;
        } else {
            return sc - pair * NTP;
        }
    };
    const int cin8 = p.cin >> 3;           // one plane of one slot, in 16-byte units
    const int s16 = NPL * cin8 + 2;        // slot stride in 16-byte units (the planes + 32 B pad)
    // 4 channels of a pixel slot -> the planes of an LDS tile; F16: the largest magnitude this lane has split (NaNs pass fmaxf by)
    [[maybe_unused]] float rngmax = 0.f;
    auto split4 = [&](char* region, int slot, int q, f32x4 v, int s16_, int c8_) __attribute__((always_inline)) {
        if constexpr (F16) {
            rngmax = fmaxf(fmaxf(rngmax, fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
            f16s_store4(region, slot, q, v, s16_, c8_);
        } else {
            bf3_store4(region, slot, q, v, s16_, c8_);
        }
    };
    const int npair = p.nchunk >> 1;       // c_in / 32
#define IAF_BSTAMP(k) do { if (p.dbg && tid == 0) p.dbg[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
    IAF_BSTAMP(0);

    // ================= prologue (1): activation tile loads ============================================================
    const int nq = p.cin >> 2;                     // 4-channel items per pixel
    const int nitems = p.nslot * nq;
    constexpr int SU = (INMODE == IN_PIXMAJOR) ? 16 : 4;
    f32x4 sv[SU];
    const int Pbase = P0 - p.halo_before;
    const int flo = Pbase < 0 ? -Pbase * nq : 0;
    const long long rem = (long long)(p.P - Pbase) * nq;
    const int fhi = rem < nitems ? (int)rem : nitems;
    if (INMODE == IN_PIXMAJOR) {
        const f32x4* src = (const f32x4*)p.x + (long long)Pbase * nq;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int f = tid + u * NTHREADS;
            const int fc = f < flo ? flo : (f < fhi ? f : fhi - 1);
            sv[u] = src[fc];
            if (f < flo || f >= fhi) sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // IN_NCHW (z, or [elu](concat(x, x2)) of a plain conv).  A thread stages ONE pixel slot per pass (lane = slot: coalesced along
    // pixels) and the channel quads q = wave, wave + NWV, ... of it: the index arithmetic (a division by reciprocal, the source
    // select of the concat) is per slot, a quad costs one pointer add; all loads of a pass are issued before the arithmetic, and the
    // first pass's loads are issued HERE, in front of the weight ring's (vmcnt counts in order: a wait for a staging load issued
    // behind the ring loads would be a wait for the ring as well).  Round 4 measured (tools/conv_stamps.py,
    // profiles/r04/experiments/plain_conv_stamps.txt) that none of this moved the launch time of the bf16x3 kernels -- two workgroups shared a
    // CU at NT = 2 and the weight stream bound the launch; on the fp16 planes the searched shapes are 512-thread workgroups, one per CU, and the
    // staging is exposed (round 6: tail_fold below, profiles/r06/experiments/ab_plain_conv_tail_slots_same_box.txt).
    constexpr int NWV = NTHREADS / 64;
    constexpr int SQ = 10;                                         // quads in flight per thread (c_in <= 40 NWV: every c_in the packs allow at 8 waves)
    f32x4 nb_v[SQ];
    int nb_slot = -1, nb_q0 = 0, nb_nq = 0;
    bool nb_ok = false;
    // quads [qb, qb + SQ) of this thread's list, slot pass sp
    auto nchw_issue = [&](int sp, int qb) __attribute__((always_inline)) {
        const int sl = sp * 64 + lane;
        const bool in = sl < p.nslot;
        const int Pg = Pbase + sl;
        nb_ok = in && Pg >= 0 && Pg < p.P;
        int b, ppx;
        fast_divmod(nb_ok ? Pg : 0, HW, 1.0f / (float)HW, b, ppx);
        const bool two = (EPI == EPI_PLAIN) && p.x2 != nullptr;
        const int c1 = two ? p.c_split : p.cin;                    // channels of the first source
        const float* s1 = p.x + (size_t)b * c1 * HW + ppx;
        const float* s2 = two ? p.x2 + (size_t)b * (p.cin - c1) * HW + ppx : s1;
        nb_slot = in ? sl : -1; nb_q0 = wave + qb * NWV;
#pragma unroll
        for (int u = 0; u < SQ; ++u) {
            const int q = nb_q0 + u * NWV;                         // this thread's u-th quad of the batch (wave-uniform)
            if (q < nq) {                                          // (a scalar branch)
                const bool second = two && 4 * q >= c1;
                const float* src = (second ? s2 + (size_t)(4 * q - c1) * HW : s1 + (size_t)(4 * q) * HW);
#pragma unroll
                for (int r = 0; r < 4; ++r) nb_v[u][r] = src[(size_t)r * HW];
            }
        }
    };
    auto nchw_finish = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < SQ; ++u) {
            const int q = nb_q0 + u * NWV;
            if (q >= nq) continue;
            f32x4 v = nb_v[u];
            if (EPI == EPI_PLAIN && p.in_elu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = elu_f(v[r]);
            }
            if (!nb_ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (nb_slot >= 0) split4(smem, nb_slot, q, v, s16, cin8);
        }
    };
    // The slots past the first 64 (16-pixel rows at 32 pixels per workgroup: 66 slots) used to be a second slot pass with TWO live lanes and a
    // whole load round trip of its own; where they are few -- (nslot - 64) nq items <= one per thread -- they travel with the first pass
    // instead, one (slot, quad) item per thread (round 6; profiles/r06/plain_conv_stamps.txt: staging is 10-18 k of a workgroup's 35 k cycles)
    f32x4 tl_v = f32x4{0.f, 0.f, 0.f, 0.f};
    int tl_slot = -1, tl_q = 0;
    bool tl_ok = false;
    const int ntail = p.nslot > 64 ? p.nslot - 64 : 0;
#ifndef IAF_EXP_TAILFOLD
#define IAF_EXP_TAILFOLD 1
#endif
    const bool tail_fold = IAF_EXP_TAILFOLD && (INMODE == IN_NCHW && S2 != 1) && ntail > 0 && ntail * nq <= NTHREADS;
    auto tail_issue = [&]() __attribute__((always_inline)) {
        const bool valid = tid < ntail * nq;
        int q, t;
        fast_divmod(valid ? tid : 0, ntail, 1.0f / (float)ntail, q, t);
        const int sl = 64 + t, Pg = Pbase + sl;
        tl_ok = valid && Pg >= 0 && Pg < p.P;
        int b, ppx;
        fast_divmod(tl_ok ? Pg : 0, HW, 1.0f / (float)HW, b, ppx);
        const bool two = (EPI == EPI_PLAIN) && p.x2 != nullptr;
        const int c1 = two ? p.c_split : p.cin;
        const bool second = two && 4 * q >= c1;
        const float* src = second ? p.x2 + ((size_t)b * (p.cin - c1) + (4 * q - c1)) * HW + ppx : p.x + ((size_t)b * c1 + 4 * q) * HW + ppx;
#pragma unroll
        for (int r = 0; r < 4; ++r) tl_v[r] = src[(size_t)r * HW];
        tl_slot = valid ? sl : -1; tl_q = q;
    };
    auto tail_finish = [&]() __attribute__((always_inline)) {
        f32x4 v = tl_v;
        if (EPI == EPI_PLAIN && p.in_elu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = elu_f(v[r]);
        }
        if (!tl_ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tl_slot >= 0) split4(smem, tl_slot, tl_q, v, s16, cin8);
    };
    if constexpr (INMODE == IN_NCHW && S2 != 1) {
        nchw_issue(0, 0);
        if (tail_fold) tail_issue();
    }

    // ================= prologue (2): weight ring ======================================================================
    // step s = pair * 5 + tap; wave kh owns steps [s0, s1).  One step = NT tiles x 3 planes x 1 KiB, contiguous.
    const int S = S2 == 2 ? npair << lgt : npair * NTP;
    const int s0 = (kh * S) / KS, s1 = ((kh + 1) * S) / KS;
    const size_t wstep = (size_t)p.ncot * NPL * 64;                             // f32x4 per step
    const f32x4* wbase = (const f32x4*)p.wp + (size_t)cot0 * NPL * 64 + lane;  // this wave's tiles, this lane's 16 bytes
    // ragged co groups (round 6: c_out / 16 need not be a multiple of NT x WCO -- down_conv1's 28 tiles in three workgroups of 2 x 5 per
    // pixel block instead of seven of 2 x 2): a tile past the layer's last one reads the last one's fragments (a valid address; its
    // sums are computed and dropped -- the epilogue's items of such tiles are inactive)
    int toff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ta = cot0 + t < p.ncot ? cot0 + t : p.ncot - 1;
        toff[t] = (ta - cot0) * NPL * 64;
    }
    f32x4 wr[U][NT][NPL];
    // fragments [lo, hi) of step s -> ring slot I (a step's refill is issued in PPW parts, one per pixel-tile group of
    // MFMAs, so that the loads sit BETWEEN the MFMAs instead of in a cluster that starves the pipe)
    auto load_part = [&](auto slot_c, auto lo_c, auto hi_c, int s) __attribute__((always_inline)) {
        constexpr int I = decltype(slot_c)::value, LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        int sc = s < s1 ? s : s1 - 1;                                           // clamped: branch-free, redundant at the tail
        // S2 = 2: a one-tap phase of a narrow conv has fewer steps than wave groups (c_in = 64: S = 2 < KS = 4) -- the group with the
        // empty range [0, 0) must not clamp to step -1, one step IN FRONT of the pack (read, never used: a fault where that page is not mapped)
        if constexpr (S2 == 2) sc = sc < 0 ? 0 : sc;
        int ws = sc;
        if constexpr (S2 == 2) { const int pr = step_pair(sc); ws = pr * NTP + step_tap(sc, pr); }
        const f32x4* q = wbase + (size_t)ws * wstep;
#pragma unroll
        for (int f = LO; f < HI; ++f) wr[I][f / NPL][f % NPL] = q[toff[f / NPL] + (f % NPL) * 64];
    };
    auto load_step = [&](auto slot_c, int s) __attribute__((always_inline)) {
        load_part(slot_c, std::integral_constant<int, 0>{}, std::integral_constant<int, NT * NPL>{}, s);
    };
    static_for<RD>([&](auto i) { load_step(i, s0 + decltype(i)::value); });
    IAF_BSTAMP(1);

    // ================= prologue (3): per-lane geometry ================================================================
    const int pl = lane & 15, kk = lane >> 4;
    // x operand address of (pixel tile q, tap t) = xbase[q] + toff(t) when the tap lands inside the image, else the all-zero
    // slot; toff is wave-uniform.  (Kept as base + validity bits, selected arithmetically per step: a per-tap address
    // ARRAY indexed by the runtime tap would be placed in scratch memory.)
    int xbase[PPW];
    unsigned xvalid[PPW];
    const int zaddr = p.nslot * s16 + kk;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int tl = (pw * PPW + q) * 16 + pl;      // pixel index inside the workgroup tile
        const int Pl = P0 + tl;
        const bool pvalid = Pl < p.P;
        int bimg, pp, h, w;
        fast_divmod(Pl, HW, 1.0f / (float)HW, bimg, pp);
        fast_divmod(pp, W, 1.0f / (float)W, h, w);
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            const int dh = p.tap_dh[t], dw = p.tap_dw[t];
            const bool v = pvalid && (h + dh >= 0) && (h + dh < p.H) && (w + dw >= 0) && (w + dw < W);
            m |= (v ? 1u : 0u) << t;
        }
        xvalid[q] = m;
        xbase[q] = (tl + p.halo_before) * s16 + kk;
    }

    // NCHW source (z, or the posterior sample computed on the fly) -> three bf16 planes of an LDS tile of `nsl` slots
    // NCHW source (z, or the posterior sample computed on the fly) -> three bf16 planes of an LDS tile of `nsl` slots (the fused first
    // layer's z tile and the posterior sample; a plain NCHW input goes through nchw_issue / nchw_finish above)
    auto stage_nchw = [&](char* region, int nsl, int cin_, int s16_, int cin8_, const float* xsrc, bool posterior) {
        const int nit = nsl * (cin_ >> 2);
        // (index arithmetic by float reciprocal: an integer division is ~40 instructions on this ISA)
        const float rnsl = 1.0f / (float)nsl, rHW = 1.0f / (float)HW;
        for (int base = tid; base < nit; base += 4 * NTHREADS) {
            int dq[4], dsl[4];
            f32x4 v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NTHREADS;
                dsl[u] = -1; dq[u] = 0;
                v4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < nit) {
                    int q, sl;                                                 // slot fastest: coalesced along pixels
                    fast_divmod(idx, nsl, rnsl, q, sl);
                    const int Pg = Pbase + sl;
                    dsl[u] = sl; dq[u] = q;
                    if (Pg >= 0 && Pg < p.P) {
                        int b, ppx;
                        fast_divmod(Pg, HW, rHW, b, ppx);
                        const size_t gb = ((size_t)b * cin_ + 4 * q) * HW + ppx;
                        if constexpr (EPI == EPI_PLAIN) {     // input = [elu](concat(x[:, :c_split], x2))  (tf_train.py:36,52,87-88)
                            if (p.x2 && 4 * q >= p.c_split) {
                                const size_t g2 = ((size_t)b * (cin_ - p.c_split) + (4 * q - p.c_split)) * HW + ppx;
#pragma unroll
                                for (int r = 0; r < 4; ++r) v4[u][r] = p.x2[g2 + (size_t)r * HW];
                            } else {
                                const size_t g1 = p.x2 ? ((size_t)b * p.c_split + 4 * q) * HW + ppx : gb;
#pragma unroll
                                for (int r = 0; r < 4; ++r) v4[u][r] = xsrc[g1 + (size_t)r * HW];
                            }
                            if (p.in_elu) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v4[u][r] = elu_f(v4[u][r]);
                            }
                        } else if (!posterior) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v4[u][r] = xsrc[gb + (size_t)r * HW];
                        } else {   // z0 = (qm+rm) + exp(ql+rl) * eps   (tf_train.py:57,63; distributions.py:21)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const size_t i = gb + (size_t)r * HW;
                                v4[u][r] = (p.qm[i] + p.rm[i]) + __expf(0.5f * (2.f * (p.ql[i] + p.rl[i]))) * p.eps[i];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dsl[u] >= 0) split4(region, dsl[u], dq[u], v4[u], s16_, cin8_);
        }
    };

    // S2 = 1: the four phase images of [elu](x), x [B,c_in,2H,2W]: slot s2_pb[2a+b] + l holds x[:, 2i+a, 2j+b] of output pixel P0 + l
    auto stage_s2d = [&]() {
        const int nq_ = p.cin >> 2, nsl = p.nslot, nit = nsl * nq_;
        const int W2 = 2 * W;
        const size_t HW4 = 4 * (size_t)HW;
        const float rnsl = 1.0f / (float)nsl, rHW = 1.0f / (float)HW, rW = 1.0f / (float)W;
        for (int base = tid; base < nit; base += 4 * NTHREADS) {
            int dq[4], dsl[4];
            f32x4 v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NTHREADS;
                dsl[u] = -1; dq[u] = 0;
                v4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < nit) {
                    int q, sl;                                                 // slot fastest: coalesced along pixels
                    fast_divmod(idx, nsl, rnsl, q, sl);
                    const int ph = (sl >= p.s2_pb[1] ? 1 : 0) + (sl >= p.s2_pb[2] ? 1 : 0) + (sl >= p.s2_pb[3] ? 1 : 0);
                    const int pb = ph == 0 ? 0 : (ph == 1 ? p.s2_pb[1] : (ph == 2 ? p.s2_pb[2] : p.s2_pb[3]));
                    const int Pg = P0 + (sl - pb);
                    dsl[u] = sl; dq[u] = q;
                    if (Pg < p.P) {
                        int b, ppx, i, j;
                        fast_divmod(Pg, HW, rHW, b, ppx);
                        fast_divmod(ppx, W, rW, i, j);
                        const size_t gb = ((size_t)b * p.cin + 4 * q) * HW4 + (size_t)(2 * i + (ph >> 1)) * W2 + 2 * j + (ph & 1);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v4[u][r] = p.x[gb + (size_t)r * HW4];
                        if (p.in_elu) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v4[u][r] = elu_f(v4[u][r]);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dsl[u] >= 0) split4(smem, dsl[u], dq[u], v4[u], s16, cin8);
        }
    };

    // IN_FUSED0: the stack's first masked conv (c_in = 32: five steps) is computed HERE instead of in a launch of its
    // own -- its output, the hidden activation this kernel's K loop consumes, goes straight into the LDS tile and never
    // touches HBM.  z is staged with a double halo (nslot + W + 1 slots); every wave computes all pixel tiles of the
    // tile (+ halo) for its own co tiles (tiles w, w + NW, ...: distinct weights per wave, held in registers), applies
    // bias + context + ELU (layers.py:163-165) and writes the three bf16 planes.  TF statement only (no halo before).
    auto fused_layer0 = [&]() {
        constexpr int NW = PXT * WCO * KS;
        constexpr int NTL = (NW >= 8) ? 2 : 3;              // co tiles of the fused layer per wave (c_h <= 16*NW*NTL)
        const int z8 = p.f_cin >> 3, z16 = 3 * z8 + 2;
        const int nslot0 = p.nslot + W + 1;
        char* zreg = smem + ((size_t)(p.nslot + 1) * s16 << 4);
        f32x4* zreg4 = (f32x4*)zreg;
        const int ncot0 = p.nchunk;                         // the fused layer's c_out = this layer's c_in
        const int npt = (p.nslot + 15) >> 4;                // pixel tiles of the hidden tile (+ halo)
        // co tiles of this wave: tile `wave` for every pixel tile, and further tiles w + NW, w + 2 NW ...  With 8 waves
        // and at most 2 tiles beyond the first 8 (c_h = 160: tiles 8, 9) those are dealt out per (tile, pixel tile) --
        // wave w takes (8 + w/4, pixel tile w%4) -- so every wave multiplies 5 units instead of waves 0, 1 doing 8
        const bool balanced = (NW == 8) && (ncot0 - NW <= 2) && (npt <= 4);
        int tile_of[NTL];
#pragma unroll
        for (int j = 0; j < NTL; ++j) tile_of[j] = (balanced && j == 1) ? NW + (wave >> 2) : wave + j * NW;
        // its weights: [tap 5][ncot0][plane 3][lane][8 bf16] (one c_in pair), all of them in registers
        f32x4 wf[NTAPS][NTL][3];
        const f32x4* fw = (const f32x4*)p.f_wp + lane;
        f32x4 bi[NTL];
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int tc = tile_of[j] < ncot0 ? tile_of[j] : ncot0 - 1;
#pragma unroll
            for (int tp = 0; tp < NTAPS; ++tp)
#pragma unroll
                for (int pn = 0; pn < 3; ++pn) wf[tp][j][pn] = fw[(((size_t)tp * ncot0 + tc) * 3 + pn) * 64];
            bi[j] = *(const f32x4*)(p.f_bias + tc * 16 + 4 * kk);
        }
        for (int i = tid; i < z16; i += NTHREADS) zreg4[(size_t)nslot0 * z16 + i] = f32x4{0.f, 0.f, 0.f, 0.f};
        stage_nchw(zreg, nslot0, p.f_cin, z16, z8, p.f_x, p.f_x == nullptr);
        const int zzero = nslot0 * z16 + kk;
        // context (+ second context) of this lane's 4 channels of every co tile at pixel tile q: fetched one pixel
        // tile AHEAD of the MFMAs that need it
        auto load_ctx = [&](int q, f32x4* cx) {
            const int sl = q * 16 + pl, Pl = P0 + sl;
            const bool live = q < npt && sl < p.nslot && Pl < p.P;
            int bimg, pp;
            fast_divmod(Pl, HW, 1.0f / (float)HW, bimg, pp);
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                cx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool act = tile_of[j] < ncot0 && (!(balanced && j == 1) || q == (wave & 3));
                if (act && live && p.f_ctx) {
                    const size_t cb = ((size_t)bimg * p.cin + tile_of[j] * 16 + 4 * kk) * HW + pp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) cx[j][r] = p.f_ctx[cb + (size_t)r * HW];
                    if (p.f_ctx2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) cx[j][r] += p.f_ctx2[cb + (size_t)r * HW];
                    }
                }
            }
        };
        f32x4 cxn[NTL];
        load_ctx(0, cxn);
        __syncthreads();
        for (int q = 0; q < npt; ++q) {
            const int sl = q * 16 + pl;                      // slot of the hidden tile this lane's pixel column is
            const int Pl = P0 + sl;
            const bool live = sl < p.nslot && Pl < p.P;
            int bimg, pp, h, w;
            fast_divmod(Pl, HW, 1.0f / (float)HW, bimg, pp);
            fast_divmod(pp, W, 1.0f / (float)W, h, w);
            f32x4 cx[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j) cx[j] = cxn[j];
            load_ctx(q + 1, cxn);
            const bool act1 = !balanced || q == (wave & 3);   // is the second tile slot at work on this pixel tile?
            f32x4 a0[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j) a0[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tp = 0; tp < NTAPS; ++tp) {
                const int dh = p.tap_dh[tp], dw = p.tap_dw[tp];
                const bool v = live && (h + dh >= 0) && (h + dh < p.H) && (w + dw >= 0) && (w + dw < W);
                const int za = zzero + ((v ? -1 : 0) & ((sl + dh * W + dw) * z16 + kk - zzero));
                const bf16x8 xh = __builtin_bit_cast(bf16x8, zreg4[za]);
                const bf16x8 xm = __builtin_bit_cast(bf16x8, zreg4[za + z8]);
                const bf16x8 xl = __builtin_bit_cast(bf16x8, zreg4[za + 2 * z8]);
#define IAF_BF3_PROD0(J, WP, XV) a0[J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[tp][J][WP]), XV, a0[J], 0, 0, 0);
#define IAF_BF3_SIX(J) IAF_BF3_PROD0(J, 2, xh) IAF_BF3_PROD0(J, 0, xl) IAF_BF3_PROD0(J, 1, xm) IAF_BF3_PROD0(J, 1, xh) IAF_BF3_PROD0(J, 0, xm) IAF_BF3_PROD0(J, 0, xh)
                IAF_BF3_SIX(0)
                if constexpr (NTL > 1) { if (act1) { IAF_BF3_SIX(1) } }
                if constexpr (NTL > 2) { IAF_BF3_SIX(2) }
#undef IAF_BF3_SIX
#undef IAF_BF3_PROD0
            }
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const bool act = tile_of[j] < ncot0 && (j != 1 || act1);
                if (act && sl < p.nslot) {
                    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};             // pixels past the end of the batch: a zero row
                    if (live) {
                        v = a0[j] + bi[j] + cx[j];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = elu_f(v[r]);
                    }
                    split4(smem, sl, tile_of[j] * 4 + kk, v, s16, cin8);
                }
            }
        }
    };

    // ================= prologue (4): tile -> LDS, split into three bf16 planes ========================================
    {
        f32x4* zslot = smem4 + (size_t)p.nslot * s16;
        for (int i = tid; i < s16; i += NTHREADS) zslot[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (INMODE == IN_PIXMAJOR) {
            const float rnq = 1.0f / (float)nq;
            const f32x4* src = (const f32x4*)p.x + (long long)Pbase * nq;
            if constexpr (DG16) {
                // the tile's largest magnitude: the items in registers, then one pass of loads over the rest (they are loaded again below: L2)
                float m = 0.f;
                auto amax4 = [&](f32x4 v) __attribute__((always_inline)) {
                    m = fmaxf(fmaxf(m, fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
                };
#pragma unroll
                for (int u = 0; u < SU; ++u) amax4(sv[u]);
                for (int f = tid + SU * NTHREADS; f < nitems; f += NTHREADS)
                    if (f >= flo && f < fhi) amax4(src[f]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                __shared__ float dg_wmax[NTHREADS / 64];
                if (lane == 0) dg_wmax[wave] = m;
                __syncthreads();
                m = dg_wmax[0];
#pragma unroll
                for (int w = 1; w < NTHREADS / 64; ++w) m = fmaxf(m, dg_wmax[w]);
                // scale = 2^k, k = 13 - e, e = the largest magnitude's exponent
                const int eb = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);            // 0 (zero / subnormal) .. 255 (inf / NaN)
                int k = 13 - (eb - 127);
                k = k < -100 ? -100 : (k > 100 ? 100 : k);                                       // (both factors normal, exact inverses of each other)
                dg_scale = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
                dg_inv = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int f = tid + u * NTHREADS;
                if (f < nitems) {
                    const int sl = (int)(((float)f + 0.5f) * rnq);
                    split4(smem, sl, f - sl * nq, DG16 ? sv[u] * dg_scale : sv[u], s16, cin8);
                }
            }
            for (int f = tid + SU * NTHREADS; f < nitems; f += NTHREADS) {
                const int sl = (int)(((float)f + 0.5f) * rnq);
                const f32x4 v = (f >= flo && f < fhi) ? src[f] : f32x4{0.f, 0.f, 0.f, 0.f};
                split4(smem, sl, f - sl * nq, DG16 ? v * dg_scale : v, s16, cin8);
            }
        } else if (S2 == 1) {
            stage_s2d();
        } else if (INMODE == IN_NCHW) {
            nchw_finish();                                         // (pass 0, first SQ quads: issued in prologue (1))
            if (tail_fold) tail_finish();                          // (... and the slots past 64 with them)
            const int nqt = (nq + NWV - 1) / NWV;                  // quads per thread and slot
            for (int qb = SQ; qb < nqt; qb += SQ) { nchw_issue(0, qb); nchw_finish(); }
            if (!tail_fold)
                for (int sp = 1; sp * 64 < p.nslot; ++sp)
                    for (int qb = 0; qb < nqt; qb += SQ) { nchw_issue(sp, qb); nchw_finish(); }
        } else if (INMODE == IN_POSTERIOR) {
            stage_nchw(smem, p.nslot, p.cin, s16, cin8, p.x, true);
        }
    }
    if constexpr (INMODE == IN_FUSED0) { IAF_BSTAMP(6); fused_layer0(); IAF_BSTAMP(7); }
    __syncthreads();
    IAF_BSTAMP(2);

    // ================= K loop =========================================================================================
    f32x4 acc[PPW][NT];
    [[maybe_unused]] f32x4 accx[F16 ? PPW : 1][F16 ? NT : 1];   // F16: the cross products (hi lo' + lo' hi), scaled by 2^-11 behind the K loop
#pragma unroll
    for (int q = 0; q < PPW; ++q)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (F16) accx[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    if constexpr (F16) {        // an operand beyond fp16's largest finite number went into the planes: the host reads this word at its next call
        if (__any(rngmax > IAF_F16_MAX) && lane == 0 && p.rng_err)
            __hip_atomic_fetch_or(p.rng_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }

    // x operand address of (pixel tile q, step s): tap and pair are wave-uniform
    auto xaddr = [&](auto q_c, int s) __attribute__((always_inline)) -> int {
        constexpr int q = decltype(q_c)::value;
        const int sc = s < s1 ? s : s1 - 1;
        const int pair = step_pair(sc), tap = step_tap(sc, pair);
        int toff;
        if constexpr (S2 == 1) {
            int to = p.tap_off[0];
            static_for<NTP - 1>([&](auto t_c) {
                constexpr int t = decltype(t_c)::value + 1;
                to = tap == t ? p.tap_off[t] : to;
            });
            toff = to * s16;
        } else {
            int dh = p.tap_dh[0], dw = p.tap_dw[0];                      // scalar selects on kernel arguments
            static_for<NTP - 1>([&](auto t_c) {
                constexpr int t = decltype(t_c)::value + 1;
                dh = tap == t ? p.tap_dh[t] : dh; dw = tap == t ? p.tap_dw[t] : dw;
            });
            toff = (dh * W + dw) * s16;
        }
        const int mask = -(int)((xvalid[q] >> tap) & 1u);              // all ones when the tap is inside the image
        return zaddr + pair * 4 + (mask & (xbase[q] + toff - zaddr));  // branch-free: a ?: here becomes control flow
    };
    f32x4 xn[3];      // the NEXT (step, pixel tile)'s three planes, in flight while the current one is multiplied
    {
        const int a = xaddr(std::integral_constant<int, 0>{}, s0);
        xn[0] = smem4[a]; xn[1] = smem4[a + cin8];
        if constexpr (NPL == 3) xn[2] = smem4[a + 2 * cin8];
    }
    auto step_body = [&](auto slot_c, int s) __attribute__((always_inline)) {
        constexpr int I = decltype(slot_c)::value;
        static_for<PPW>([&](auto q_c) {
            constexpr int q = decltype(q_c)::value;
            // refill (this group's share) of the slot consumed RD steps from now; the last group of a multi-group step issues
            // none, so that every fragment has at least one group of MFMAs (~500 cycles) to arrive
            constexpr int NG = PPW > 1 ? PPW - 1 : 1;
            constexpr int LO = q < NG ? (q * NT * NPL) / NG : NT * NPL, HI = q < NG ? ((q + 1) * NT * NPL) / NG : NT * NPL;
            load_part(std::integral_constant<int, (I + RD) % U>{}, std::integral_constant<int, LO>{},
                      std::integral_constant<int, HI>{}, s + RD);
            f32x4 xr[3];
            xr[0] = xn[0]; xr[1] = xn[1];
            if constexpr (NPL == 3) xr[2] = xn[2];
            {   // prefetch the next pixel tile of this step, or tile 0 of the next step
                const int a = (q + 1 < PPW) ? xaddr(std::integral_constant<int, (q + 1) % PPW>{}, s)
                                            : xaddr(std::integral_constant<int, 0>{}, s + 1);
                xn[0] = smem4[a]; xn[1] = smem4[a + cin8];
                if constexpr (NPL == 3) xn[2] = smem4[a + 2 * cin8];
            }
            if constexpr (F16) {
                // three products per co tile: hi x lo' and lo' x hi into the cross accumulator, hi x hi into the main one (between them)
#define IAF_F16_PROD(ACC, WP, XP)                                                                                  \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                 \
        ACC[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wr[I][t][WP]), __builtin_bit_cast(f16x8, xr[XP]), ACC[q][t], 0, 0, 0);
                IAF_F16_PROD(accx, 0, 1)
                IAF_F16_PROD(acc, 0, 0)
                IAF_F16_PROD(accx, 1, 0)
#undef IAF_F16_PROD
            } else {
            const bf16x8 xh = __builtin_bit_cast(bf16x8, xr[0]);
            const bf16x8 xm = __builtin_bit_cast(bf16x8, xr[1]);
            const bf16x8 xl = __builtin_bit_cast(bf16x8, xr[2]);
            // six products per co tile, interleaved over the tiles so that consecutive MFMAs hit different accumulators;
            // the small terms first
#define IAF_BF3_PROD(WP, XV)                                                                                      \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                 \
        acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wr[I][t][WP]), XV, acc[q][t], 0, 0, 0);
            IAF_BF3_PROD(2, xh)   // w_l x_h
            IAF_BF3_PROD(0, xl)   // w_h x_l
            IAF_BF3_PROD(1, xm)   // w_m x_m
            IAF_BF3_PROD(1, xh)   // w_m x_h
            IAF_BF3_PROD(0, xm)   // w_h x_m
            IAF_BF3_PROD(0, xh)   // w_h x_h
#undef IAF_BF3_PROD
            }
            sched_interleave<(F16 ? 3 : 6) * NT, NPL, 0, HI - LO>();
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    {
        int s = s0;
        for (; s + U <= s1; s += U)
            static_for<U>([&](auto i) { step_body(i, s + decltype(i)::value); });
        const int r = s1 - s;                       // 0 .. U-1 steps left; chunk at s sits in ring slot 0
        static_for<U>([&](auto r_c) {
            constexpr int R = decltype(r_c)::value;
            if (R > 0 && r == R) static_for<R>([&](auto i) { step_body(i, s + decltype(i)::value); });
        });
    }

    if constexpr (F16) {
#pragma unroll
        for (int q = 0; q < PPW; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[q][t] += accx[q][t] * (1.0f / 2048.0f);                              // the cross products carry lo' = lo 2^11
                if constexpr (DG16) acc[q][t] *= dg_inv;                                // (the staged tile was scaled by 2^k)
            }
    }
    IAF_BSTAMP(3);
    // ================= split-K exchange through LDS + epilogue ========================================================
    // item = (pixel tile q of this wave group, epilogue unit u); the KS waves of a group share the items round-robin
    constexpr bool ONE_TILE_UNITS = (EPI != EPI_OUT);
    constexpr int TPU = ONE_TILE_UNITS ? 1 : 2;
    constexpr int NUNIT = NT / TPU;
    constexpr int NITEM = PPW * NUNIT;
    constexpr int NMY = (NITEM + KS - 1) / KS;
    EpiGeom g[NMY];
    EpiOps ops[NMY];
#pragma unroll
    for (int i = 0; i < NMY; ++i) {
        const int item = kh + i * KS;
        const int q = item / NUNIT, u = item - q * NUNIT;
        g[i] = epi_geom(p, P0 + (pw * PPW + q) * 16 + pl, kk, item < NITEM && cot0 + u * TPU < p.ncot);
        if constexpr (S2 == 2) g[i].up = 4 + 2 * ph_a + ph_b;
        epi_load<EPI>(p, g[i], cot0 + u * TPU, ops[i]);
    }
    f32x4 val[NMY][TPU];
    if constexpr (KS > 1) {
        __syncthreads();                                       // every wave is done reading the activation tile
        f32x4* red = smem4;                                    // [pw][kh][q][t][64 lanes] x 16 bytes
        f32x4* wbuf = red + (size_t)((grp * KS + kh) * PPW * NT) * 64 + lane;
#pragma unroll
        for (int q = 0; q < PPW; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) wbuf[(q * NT + t) * 64] = acc[q][t];
        __syncthreads();
        const f32x4* rbuf = red + (size_t)(grp * KS * PPW * NT) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NMY; ++i) {
            const int item = kh + i * KS;
            const int q = item / NUNIT, u = item - q * NUNIT;
#pragma unroll
            for (int e = 0; e < TPU; ++e) {
                f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
                if (item < NITEM) {
                    const int t = u * TPU + e;
#pragma unroll
                    for (int k = 0; k < KS; ++k) sum += rbuf[(size_t)((k * PPW + q) * NT + t) * 64];
                }
                val[i][e] = sum;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NMY; ++i)          // KS == 1: item i = (q, u) with static indices
#pragma unroll
            for (int e = 0; e < TPU; ++e) val[i][e] = acc[i / NUNIT][(i % NUNIT) * TPU + e];
    }
    IAF_BSTAMP(4);
#pragma unroll
    for (int i = 0; i < NMY; ++i) {
        const int item = kh + i * KS;
        const int u = item % NUNIT;
        epi_apply<EPI, NTP>(p, g[i], cot0 + u * TPU, val[i][0], val[i][TPU - 1], ops[i]);
    }
    IAF_BSTAMP(5);
}
