// iaf_kernels_backward.hpp -- backward kernels: affine/log-det, border-bit table, MFMA weight gradient + reduce, weight-norm backward (masked and plain), pixel-major staging, posterior-block elementwise.
// Part of the single translation unit iaf_engine.hip (included there, in order; not a standalone header).
#pragma once

// ---------------------------------------------------------------------------------------------
// backward of the IAF step (what TF autodiff derives from tf_train.py:69-72 + layers.py:52-64,158-166)
// ---------------------------------------------------------------------------------------------
// (1) affine + log-det:  z_new = (z - 0.1 m_raw) e^{-0.1 s_raw},  logsd = 0.1 s_raw
//       d m_raw = -0.1 dz_new e^{-logsd};   d s_raw = 0.1 (dlogsd - dz_new z_new)
//     written pixel-major in the PACKED channel order of the output GEMM (tiles m0,s0,m1,s1,...), plus a pixel-major
//     copy of z (operand of the first conv's weight gradient).
__global__ __launch_bounds__(256) void iaf_bwd_affine_kernel(const float* __restrict__ z, const float* __restrict__ z_new,
                                                            const float* __restrict__ logsd, const float* __restrict__ dzn,
                                                            const float* __restrict__ dls, float* __restrict__ dy3,
                                                            float* __restrict__ zpm, int n_z, int HW, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // i runs pixel-major: channel fastest (coalesced writes); NCHW reads are strided but the tensors are tiny
        const int c = (int)(i % n_z);
        const long long P = i / n_z;
        const long long b = P / HW, pp = P - b * HW;
        const size_t src = ((size_t)b * n_z + c) * HW + pp;
        const float g = dzn[src], ls = logsd[src];
        const float dm = -0.1f * g * __expf(-ls);
        const float ds = 0.1f * (dls[src] - g * z_new[src]);
        const size_t row = (size_t)P * (2 * n_z) + (size_t)(c >> 4) * 32 + (c & 15);
        dy3[row] = dm;
        dy3[row + 16] = ds;
        zpm[(size_t)P * n_z + c] = z[src];
    }
}

// (2) weight gradient of one masked conv:  dW[tap][ci][co] = sum_p X[p + shift(tap)][ci] * dY[p][co]
//     as MFMA GEMM  D[ci][co] += A[ci][k = pixel] B[k = pixel][co]  on pixel-major X [P][cin] and dY [P][cout].
//     Workgroup = (tap, pair of ci tiles, pixel range); its 4 waves split the range, reduce through LDS and write one
//     partial [krange][tap][cin][cout]; the partials are summed by iaf_wn_bwd_kernel.  Operands are dword loads
//     straight from L1/L2 (A: 2 per K step, B: NCOT per K step for 2*NCOT MFMAs).
#include "iaf_wgrad_types.hpp"

// border table of the weight gradient: the 9 in-image bits of every pixel, computed once per backward instead of two
// integer divisions per K step per lane (which cost as much issue time as the MFMAs of the step)
__global__ __launch_bounds__(256) void iaf_tapmask_kernel(unsigned short* __restrict__ mask, int H, int W, int P) {
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= P) return;
    const int pp = px % (H * W);
    const int h = pp / W, w = pp - h * W;
    unsigned m = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int dh = t / 3 - 1, dw = t % 3 - 1;
        if (h + dh >= 0 && h + dh < H && w + dw >= 0 && w + dw < W) m |= 1u << t;
    }
    mask[px] = (unsigned short)m;
}

template <int NCOT, int D>
__global__ __launch_bounds__(256, 2) void iaf_wgrad_kernel(WgradP p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, range, bz;
    wgrad_decode(p, bx, range, bz);
    const int tap = bx % p.ntaps;
    const int cip = bx / p.ntaps;                  // pair of ci tiles
    const int cob = bz * NCOT * 16;                // this workgroup's first packed output channel
    const int ci0 = cip * 32;
    const int nci = (p.cin - ci0 >= 32) ? 2 : 1;   // c_in = 16 has a single tile
    const int i15 = lane & 15, ks = lane >> 4;
    const int dh = p.tap_dh[tap], dw = p.tap_dw[tap];
    const int tapbit = (dh + 1) * 3 + (dw + 1), shift = dh * p.W + dw;
    f32x4 acc[2][NCOT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < NCOT; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r0 = range * p.px_per_range;
    const int r1 = min(p.P, r0 + p.px_per_range);
    // wave w takes K steps w, w+4, ... of the range (4 pixels each).  With two waves per SIMD nothing but the wave's own
    // prefetch hides the global-load latency (several hundred ns under load, i.e. more than one K step of MFMAs): the raw
    // operands of the next D steps sit in a register ring.  Per step: the border selects of the oldest slot (its loads were
    // issued D steps ago), then the MFMAs with the loads of step k+D into that slot spread between them.  Loads are
    // unconditional and their addresses do not depend on loaded data (the shifted pixel is clamped into the tensor, the
    // border bit only feeds the select): a load behind a load, a branch around a load or a select right behind it would put
    // the wave to sleep before its MFMAs are issued.
    const int start = r0 + 4 * wave;
    const int nstep = (r1 - start + 15) / 16;              // steps of this wave (<= 0: nothing to do)
    float av[2], bv[NCOT];                                 // operands of the current step
    float ra[D][2], rb[D][NCOT];                           // raw loads of the steps in flight
    unsigned rm[D];
    bool rv[D];
    const int xoff = ci0 + i15, xoff1 = ci0 + i15 + (nci == 2 ? 16 : 0);
    auto issue = [&](int slot, int pb) {
        const int pk = pb + ks;                            // this lane's pixel for both operands
        rv[slot] = pk < r1;
        const int pkc = rv[slot] ? pk : r1 - 1;
        int xp = pkc + shift;
        xp = xp < 0 ? 0 : (xp >= p.P ? p.P - 1 : xp);
        rm[slot] = p.tapmask[pkc];
        const float* xr = p.x + (size_t)xp * p.cin;
        const float* dr = p.dy + (size_t)pkc * p.cout + cob + i15;
        ra[slot][0] = xr[xoff];
        ra[slot][1] = xr[xoff1];
#pragma unroll
        for (int t = 0; t < NCOT; ++t) rb[slot][t] = dr[t * 16];
    };
    auto take = [&](int slot) {
        const bool nxv = rv[slot] && ((rm[slot] >> tapbit) & 1u);
        av[0] = nxv ? ra[slot][0] : 0.f;
        av[1] = (nxv && nci == 2) ? ra[slot][1] : 0.f;
#pragma unroll
        for (int t = 0; t < NCOT; ++t) bv[t] = rv[slot] ? rb[slot][t] : 0.f;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {       // slot by slot, as the loop issues them: its counted waits (vmcnt) assume that order
        issue(d, start + 16 * d);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int k = 0; k < nstep; k += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            take(d);            // (a trailing partial group runs on clamped addresses and zero operands)
            __builtin_amdgcn_sched_barrier(0);
            issue(d, start + 16 * (k + d + D));            // beyond the range: clamped address, masked to zero
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int t = 0; t < NCOT; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[t], acc[a][t], 0, 0, 0);
            static_for<2 * NCOT>([&](auto m_c) {            // address arithmetic and loads spread between the MFMAs (see (2'))
                constexpr int m = decltype(m_c)::value;
                constexpr int NLD = NCOT + 3, FIRST = (2 * NCOT) / 4, SPAN = 2 * NCOT - FIRST;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if constexpr (m >= FIRST && ((m - FIRST) * NLD) / SPAN != ((m - FIRST + 1) * NLD) / SPAN)
                    __builtin_amdgcn_sched_group_barrier(0x020, ((m - FIRST + 1) * NLD) / SPAN - ((m - FIRST) * NLD) / SPAN, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // reduce the 4 waves through LDS, one ci tile (a) at a time: [wave][t][r][lane] -- half the LDS of doing both at once,
    // which is what lets two workgroups share a CU (two waves per SIMD hide each other's load latency)
    // D layout: lane holds D[row = 4*(l>>4)+r][col = l&15] = (ci = tile*16 + 4*ks + r, co = t*16 + i15)
    float* out = p.part + (((size_t)range * p.ntaps + tap) * p.cin) * p.cout;
    float* mine = wsm + (size_t)wave * (NCOT * 4 * 64) + lane;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (a) __syncthreads();
#pragma unroll
        for (int t = 0; t < NCOT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(t * 4 + r) * 64] = acc[a][t][r];
        __syncthreads();
        if (a < nci)
            for (int e = wave; e < NCOT * 4; e += 4) {            // (t, r) pairs spread over the waves
                const int t = e >> 2, r = e & 3;
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) sum += wsm[(size_t)k * (NCOT * 4 * 64) + (size_t)e * 64 + lane];
                out[(size_t)(ci0 + a * 16 + 4 * ks + r) * p.cout + cob + t * 16 + i15] = sum;
            }
    }
}

// (2') the same GEMM with WIDE operand loads.  A lane's 8-byte load of X brings channels (2 i15, 2 i15 + 1) of its pixel and
//      its BW*4-byte loads of dY bring BW consecutive output channels: element j of a load feeds MFMA tile j, whose rows /
//      columns are therefore the channels {base + width*i + j} -- a permutation of the 16x16 tiles that only the write-out
//      needs to know about.  One load instruction per 2 (X) or BW (dY) tiles instead of one per tile: the K loop of (2) is
//      bound by the number of vector-memory instructions per MFMA, not by their latency (a deeper ring changes nothing).
//      Workgroup = (tap, 32 input channels, pixel range, NB*BW output tiles); needs cin % 16 == 0, cout % (16 BW NB) == 0.
template <int BW, int NB, int D>
__global__ __launch_bounds__(256, 2) void iaf_wgrad_wide_kernel(WgradP p) {
    constexpr int NBT = BW * NB;
    typedef float fA __attribute__((ext_vector_type(2)));
    typedef float fB __attribute__((ext_vector_type(BW)));
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, range, bz;
    wgrad_decode(p, bx, range, bz);
    const int tap = bx % p.ntaps;
    const int cip = bx / p.ntaps;
    const int cob = bz * NBT * 16;
    const int ci0 = cip * 32;
    const int i15 = lane & 15, ks = lane >> 4;
    const bool civ = ci0 + 2 * i15 + 1 < p.cin;        // cin % 32 == 16: the last block's lanes i15 >= 8 hold nothing
    const int dh = p.tap_dh[tap], dw = p.tap_dw[tap];
    const int tapbit = (dh + 1) * 3 + (dw + 1), shift = dh * p.W + dw;
    f32x4 acc[2][NBT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < NBT; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r0 = range * p.px_per_range;
    const int r1 = min(p.P, r0 + p.px_per_range);
    const int start = r0 + 4 * wave;
    const int nstep = (r1 - start + 15) / 16;
    float av[2], bv[NBT];
    fA ra[D];
    fB rb[D][NB];
    unsigned rm[D];
    bool rv[D];
    // addresses as uniform base + 32-bit byte offset (host guarantees the tensors are < 4 GiB): a handful of full-rate
    // VALU instructions per step, which the scheduler is told to spread -- with the loads -- between the MFMAs of the
    // step (an in-order wave that issues its 30 VALU + 5 VMEM instructions in a block leaves its SIMD's MFMA pipe to
    // the other resident wave for ~250 of every ~1000 cycles; measured, the two did not fill each other's gaps).
    const unsigned cin4 = (unsigned)p.cin * 4u, cout4 = (unsigned)p.cout * 4u;
    const unsigned xlane = (unsigned)(civ ? ci0 + 2 * i15 : ci0) * 4u, dlane = (unsigned)(cob + BW * i15) * 4u;
    const char* xb = (const char*)p.x;
    const char* db = (const char*)p.dy;
    const char* mb = (const char*)p.tapmask;
    auto issue = [&](int slot, int pb) {
        const int pk = pb + ks;
        rv[slot] = pk < r1;
        const int pkc = min(pk, r1 - 1);
        const int xp = max(0, min(pkc + shift, p.P - 1));
        rm[slot] = *(const unsigned short*)(mb + (unsigned)pkc * 2u);
        ra[slot] = *(const fA*)(xb + (__umul24((unsigned)xp, cin4) + xlane));
        const unsigned doff = __umul24((unsigned)pkc, cout4) + dlane;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) rb[slot][nb] = *(const fB*)(db + (doff + (unsigned)(nb * 16 * BW * 4)));
    };
    auto take = [&](int slot) {
        const bool nxv = civ && rv[slot] && ((rm[slot] >> tapbit) & 1u);
        av[0] = nxv ? ra[slot][0] : 0.f;
        av[1] = nxv ? ra[slot][1] : 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < BW; ++e) bv[nb * BW + e] = rv[slot] ? rb[slot][nb][e] : 0.f;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {       // slot by slot, as the loop issues them: its counted waits (vmcnt) assume that order
        issue(d, start + 16 * d);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int k = 0; k < nstep; k += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            take(d);            // (a trailing partial group runs on clamped addresses and zero operands)
            __builtin_amdgcn_sched_barrier(0);
            issue(d, start + 16 * (k + d + D));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int t = 0; t < NBT; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[t], acc[a][t], 0, 0, 0);
            // one MFMA, then up to two of the step's address instructions; the loads from the second quarter on
            static_for<2 * NBT>([&](auto m_c) {
                constexpr int m = decltype(m_c)::value;
                constexpr int NLD = NB + 2, FIRST = (2 * NBT) / 4, SPAN = 2 * NBT - FIRST;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if constexpr (m >= FIRST && ((m - FIRST) * NLD) / SPAN != ((m - FIRST + 1) * NLD) / SPAN)
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // reduce the 4 waves through LDS as in (2); tile (a, t = nb*BW + e) holds
    //   rows ci = ci0 + 2 (4 ks + r) + a,   columns co = cob + nb*16*BW + BW*i15 + e
    float* out = p.part + (((size_t)range * p.ntaps + tap) * p.cin) * p.cout;
    float* mine = wsm + (size_t)wave * (NBT * 4 * 64) + lane;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (a) __syncthreads();
#pragma unroll
        for (int t = 0; t < NBT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(t * 4 + r) * 64] = acc[a][t][r];
        __syncthreads();
        for (int e = wave; e < NBT * 4; e += 4) {
            const int t = e >> 2, r = e & 3;
            const int ci = ci0 + 2 * (4 * ks + r) + a;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += wsm[(size_t)k * (NBT * 4 * 64) + (size_t)e * 64 + lane];
            if (ci < p.cin) out[(size_t)ci * p.cout + cob + (t / BW) * 16 * BW + BW * i15 + (t % BW)] = sum;
        }
    }
}

// part[0][i] + part[1][i] + ... in THAT order (bit for bit what a plain loop gives), the loads of eight ranges issued before their adds:
// a loop over a run-time count is not pipelined by the compiler, and 17 dependent trips to memory were the 8 us of this launch
__device__ __forceinline__ f32x4 sum_ranges(const f32x4* __restrict__ part, size_t n4, size_t i, int nrange) {
    f32x4 a = part[i];
    int k = 1;
    for (; k + 8 <= nrange; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * n4 + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; k + 2 <= nrange; k += 2) {
        const f32x4 v0 = part[(size_t)k * n4 + i], v1 = part[(size_t)(k + 1) * n4 + i];
        a += v0; a += v1;
    }
    if (k < nrange) a += part[(size_t)k * n4 + i];
    return a;
}

// (3a) sum the wgrad partials over the pixel ranges (fully parallel, 16-byte accesses) and, in extra workgroups of the
//      same launch, column-sum dY over pixel slabs for the bias gradient:
//        dW[i] = sum_k part[k][i]            blocks [0, nblk_w)
//        dbp[r][co] = sum_{p in slab r} dY[p][co]   blocks [nblk_w, nblk_w + nslab)
__global__ __launch_bounds__(256) void iaf_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int nrange,
                                                              size_t n4, int nblk_w, const float* __restrict__ dy,
                                                              float* __restrict__ dbp, int P, int cout, int px_per_slab) {
    if ((int)blockIdx.x < nblk_w) {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)nblk_w * blockDim.x) {
            ((f32x4*)dW)[i] = sum_ranges((const f32x4*)part, n4, i, nrange);
        }
    } else {
        const int slab = blockIdx.x - nblk_w;
        const int p0 = slab * px_per_slab, p1 = min(P, p0 + px_per_slab);
        for (int co = threadIdx.x; co < cout; co += blockDim.x) {
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int pix = p0;
            for (; pix + 8 <= p1; pix += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += dy[(size_t)(pix + u) * cout + co];     // 8 independent loads in flight
            }
            for (; pix < p1; ++pix) a[0] += dy[(size_t)pix * cout + co];
            dbp[(size_t)slab * cout + co] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        }
    }
}

// (3a') the same for ALL GEMM layers of a stack in one launch (three 5 us launches per stack otherwise): block ranges per
//       layer for the partial sums, then nslab blocks per layer for the dY column sums.
struct ReduceLayer {
    const float* part; float* dW; size_t n4; int nrange; int blk_begin;
    const float* dy; float* dbp; int cout;
    float* dbrd;      // Theano statement: [nslab][4][cout] sums of dY over the pixels whose tap t = 1..4 falls outside (or NULL)
};
struct ReduceArgs {
    ReduceLayer L[MAX_GEMM_LAYERS]; int n, nblk_total, nslab, P, px_per_slab;
    const unsigned short* tapmask; int brd_bit[NTAPS - 1];    // in-image bit of taps 1..4 in the border table
};

__global__ __launch_bounds__(256) void iaf_wgrad_reduce_multi_kernel(ReduceArgs a) {
    if ((int)blockIdx.x < a.nblk_total) {
        int li = 0;
        for (int i = 1; i < a.n; ++i)
            if ((int)blockIdx.x >= a.L[i].blk_begin) li = i;
        const float* part = a.L[li].part;
        float* dW = a.L[li].dW;
        const size_t n4 = a.L[li].n4;
        const int nrange = a.L[li].nrange;
        const int nblk = ((li + 1 < a.n) ? a.L[li + 1].blk_begin : a.nblk_total) - a.L[li].blk_begin;
        const int blk = blockIdx.x - a.L[li].blk_begin;
        for (size_t i = blk * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)nblk * blockDim.x) {
            ((f32x4*)dW)[i] = sum_ranges((const f32x4*)part, n4, i, nrange);
        }
    } else {
        const int idx = blockIdx.x - a.nblk_total;
        const int li = idx / a.nslab, slab = idx - li * a.nslab;
        const float* dy = a.L[li].dy;
        float* dbp = a.L[li].dbp;
        const int cout = a.L[li].cout;
        const int p0 = slab * a.px_per_slab, p1 = min(a.P, p0 + a.px_per_slab);
        for (int co = threadIdx.x; co < cout; co += blockDim.x) {
            float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int pix = p0;
            for (; pix + 8 <= p1; pix += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) s8[u] += dy[(size_t)(pix + u) * cout + co];
            }
            for (; pix < p1; ++pix) s8[0] += dy[(size_t)pix * cout + co];
            dbp[(size_t)slab * cout + co] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        }
        // gradient of the border-indicator channel's weights (conv.py:71-83): that channel is 1 exactly where a tap
        // leaves the image, so d border[t][co] = sum of dY[p][co] over the pixels p whose tap t falls outside
        float* dbrd = a.L[li].dbrd;
        if (dbrd) {
            for (int co = threadIdx.x; co < cout; co += blockDim.x) {
                float sb[NTAPS - 1] = {0.f, 0.f, 0.f, 0.f};
                for (int pix = p0; pix < p1; ++pix) {
                    const unsigned m = a.tapmask[pix];
                    if (m == 0x1ffu) continue;                       // interior pixel (uniform over the workgroup)
                    const float v = dy[(size_t)pix * cout + co];
#pragma unroll
                    for (int t = 0; t < NTAPS - 1; ++t) sb[t] += ((m >> a.brd_bit[t]) & 1u) ? 0.f : v;
                }
#pragma unroll
                for (int t = 0; t < NTAPS - 1; ++t) dbrd[((size_t)slab * (NTAPS - 1) + t) * cout + co] = sb[t];
            }
        }
    }
}

// (3b) push the weight gradient through mask + weight-norm (layers.py:57,60):
//       w = e u,  u = v / n,  v = mask V,  n = ||v||_o,  e = exp(g)
//       dg = sum dW w ;  dv = (e / n) (dW - u (sum dW u)) ;  dV = mask dv ;  db = sum_p dY
//     One workgroup per 16 output channels (same thread map as the prep kernel), all of a thread's loads in one batch.
struct WnBwdLayer {
    const float* V; const float* g;      // reference variables of THIS conv (HWIO V)
    const float* dW;                     // reduced effective-weight gradient [NTAPS][cin][cout_packed]
    const float* dbp;                    // [nslab][cout_packed] column sums of dY
    float* dV; float* dg; float* db;     // outputs: HWIO [3][3][cin][cout], [cout], [cout]
    int cin, cout, cout_packed, nslab, zerodiag, pack_stride, pack_off;   // packed channel of o: (o/16)*pack_stride*16 + pack_off*16 + o%16
    // Theano statement (variant != IAF_VARIANT_TF): V = w OIHW [cout][cin+1][3][3], g = s, dV / dg the same shapes;
    // dbrd = [nslab][4][cout_packed] partial gradients of the border channel's taps 1..4
    const float* dbrd; int variant;
    int skip;            // batched launch: this conv finished its own weight-norm backward already (a deconv2d)
};

template <int NCH>
__device__ __forceinline__ void wn_bwd_tile(const WnBwdLayer& L, int tile, float (*red)[16][17], float* s_n, float* s_dot) {
    const int oo = threadIdx.x & 15, cs = threadIdx.x >> 4;
    const int o = tile * 16 + oo;
    const int op = (o >> 4) * L.pack_stride * 16 + L.pack_off * 16 + (o & 15);   // packed channel index
    const int n_in = L.cin, n_out = L.cout;
    float v[NTAPS][NCH], dw[NTAPS][NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int ci = cs + 16 * it;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int kh = (t == 0 || t == 1) ? 1 : 2;
            const int kw = (t == 0) ? 1 : (t == 1 ? 2 : t - 2);
            v[t][it] = L.V[((size_t)(kh * 3 + kw) * n_in + ci) * n_out + o];
            dw[t][it] = L.dW[((size_t)t * n_in + ci) * L.cout_packed + op];
        }
    }
    float dbs = 0.f;
    for (int r = cs; r < L.nslab; r += 16) dbs += L.dbp[(size_t)r * L.cout_packed + op];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        if (!made_live(cs + 16 * it, o, n_in, n_out, L.zerodiag)) { v[0][it] = 0.f; dw[0][it] = 0.f; }
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) { ss += v[t][it] * v[t][it]; dot += dw[t][it] * v[t][it]; }
    }
    red[0][cs][oo] = ss; red[1][cs][oo] = dot; red[2][cs][oo] = dbs;
    __syncthreads();
    if (cs == 0) {
        float a = 0.f, b = 0.f, c = 0.f;
        for (int i = 0; i < 16; ++i) { a += red[0][i][oo]; b += red[1][i][oo]; c += red[2][i][oo]; }
        const float n = sqrtf(fmaxf(a, 1e-12f));
        const float e = expf(L.g[o]);
        s_n[oo] = n;
        s_dot[oo] = b / n;                 // sum dW u
        L.dg[o] = e * b / n;               // sum dW w
        L.db[o] = c;
    }
    __syncthreads();
    const float n = s_n[oo], du = s_dot[oo], e = expf(L.g[o]);
    // dV over all 9 taps: the 4 dead taps and the masked centre entries are exact zeros
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int ci = cs + 16 * it;
        const bool cl = made_live(ci, o, n_in, n_out, L.zerodiag);
#pragma unroll
        for (int kk9 = 0; kk9 < 9; ++kk9) {
            const int kh = kk9 / 3, kw = kk9 % 3;
            const int t = (kh == 1 && kw == 1) ? 0 : (kh == 1 && kw == 2) ? 1 : (kh == 2) ? 2 + kw : -1;
            float outv = 0.f;
            if (t > 0 || (t == 0 && cl)) outv = (e / n) * (dw[t < 0 ? 0 : t][it] - (v[t < 0 ? 0 : t][it] / n) * du);
            L.dV[((size_t)kk9 * n_in + ci) * n_out + o] = outv;
        }
    }
}

// (3b') the same through the Theano statement of the weights (graphy/nodes/ar.py:266-281,312-317; see prep_tile_theano):
//       k = mask w (centre tap of the first rows zeroed, ar.py:268-276),  n = ||k||_o over (c_in + border channel, taps),
//       W = e k / (n + 1e-8),  e = exp(3 s)
//       ds = 3 sum dW W ;  dk = (e / (n + 1e-8)) (dW - (k / n) (sum dW k) / (n + 1e-8)) ;  dw = mask dk ;  db = sum_p dY
//     The border channel's taps 1..4 are weights like any other (their dW comes from the reduce kernel's dbrd sums); with
//     flipmask its centre tap is live in the norm only (it multiplies zeros inside the image): dW = 0 there, dk != 0.
template <int NCH>
__device__ __forceinline__ void wn_bwd_tile_theano(const WnBwdLayer& L, int tile, float (*red)[16][17], float* s_n, float* s_dot) {
    const int oo = threadIdx.x & 15, cs = threadIdx.x >> 4;
    const int o = tile * 16 + oo;
    const int op = (o >> 4) * L.pack_stride * 16 + L.pack_off * 16 + (o & 15);
    const int n_in = L.cin, n_out = L.cout;
    const bool flip = (L.variant == IAF_VARIANT_THEANO_FLIPMASK);
    const float* wo = L.V + (size_t)o * (n_in + 1) * 9;
    float* dwo = L.dV + (size_t)o * (n_in + 1) * 9;
    float v[NTAPS][NCH], dw[NTAPS][NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int ci = cs + 16 * it;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int kh = (t == 0 || t == 1) ? 1 : 2;
            const int kw = (t == 0) ? 1 : (t == 1 ? 2 : t - 2);
            v[t][it] = wo[(size_t)ci * 9 + (flip ? (2 - kh) * 3 + (2 - kw) : kh * 3 + kw)];
            dw[t][it] = L.dW[((size_t)t * n_in + ci) * L.cout_packed + op];
        }
    }
    float dbs = 0.f, dbr[NTAPS - 1] = {0.f, 0.f, 0.f, 0.f};
    for (int r = cs; r < L.nslab; r += 16) {
        dbs += L.dbp[(size_t)r * L.cout_packed + op];
#pragma unroll
        for (int t = 0; t < NTAPS - 1; ++t) dbr[t] += L.dbrd[((size_t)r * (NTAPS - 1) + t) * L.cout_packed + op];
    }
    const int k0 = (n_out >= n_in) ? n_out / n_in : 1;
    const bool row_zeroed = L.zerodiag && o < k0;
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int ci = cs + 16 * it;
        const bool live = flip ? (ci >= 1 && made_live(n_in - ci, n_out - 1 - o, n_in, n_out, L.zerodiag))
                               : made_live(ci, o, n_in, n_out, L.zerodiag);
        if (!live || row_zeroed) { v[0][it] = 0.f; dw[0][it] = 0.f; }
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) { ss += v[t][it] * v[t][it]; dot += dw[t][it] * v[t][it]; }
    }
    red[0][cs][oo] = ss; red[1][cs][oo] = dot; red[2][cs][oo] = dbs;
#pragma unroll
    for (int t = 0; t < NTAPS - 1; ++t) red[3 + t][cs][oo] = dbr[t];
    __syncthreads();
    // border channel (thread cs == 0): weights of taps 1..4, the flipped mask's centre entry
    float wb[NTAPS - 1], dwb[NTAPS - 1], wc = 0.f;
    const bool centre_in_norm = flip && !row_zeroed && made_live(0, n_out - 1 - o, n_in, n_out, L.zerodiag);
    if (cs == 0) {
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int t = 0; t < NTAPS - 1; ++t) dwb[t] = 0.f;
        for (int i = 0; i < 16; ++i) {
            a += red[0][i][oo]; b += red[1][i][oo]; c += red[2][i][oo];
#pragma unroll
            for (int t = 0; t < NTAPS - 1; ++t) dwb[t] += red[3 + t][i][oo];
        }
#pragma unroll
        for (int t = 1; t < NTAPS; ++t) {
            const int kh = (t == 1) ? 1 : 2;
            const int kw = (t == 1) ? 2 : t - 2;
            wb[t - 1] = wo[(size_t)n_in * 9 + (flip ? (2 - kh) * 3 + (2 - kw) : kh * 3 + kw)];
            a += wb[t - 1] * wb[t - 1];
            b += dwb[t - 1] * wb[t - 1];
        }
        if (centre_in_norm) { wc = wo[(size_t)n_in * 9 + 4]; a += wc * wc; }
        const float n = sqrtf(a), ne = n + 1e-8f;
        const float e = expf(3.0f * L.g[o]);
        s_n[oo] = n;
        s_dot[oo] = b / ne;                     // sum dW k / (n + eps)
        L.dg[o] = 3.0f * e * b / ne;            // 3 sum dW W
        L.db[o] = c;
    }
    __syncthreads();
    const float n = s_n[oo], du = s_dot[oo], ne = n + 1e-8f;
    const float en = expf(3.0f * L.g[o]) / ne, rn = n > 0.f ? 1.0f / n : 0.f;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int ci = cs + 16 * it;
        const bool cl = !row_zeroed && (flip ? (ci >= 1 && made_live(n_in - ci, n_out - 1 - o, n_in, n_out, L.zerodiag))
                                             : made_live(ci, o, n_in, n_out, L.zerodiag));
#pragma unroll
        for (int pos = 0; pos < 9; ++pos) {
            const int q = flip ? 8 - pos : pos;       // filter position in the unflipped frame
            const int kh = q / 3, kw = q % 3;
            const int t = (kh == 1 && kw == 1) ? 0 : (kh == 1 && kw == 2) ? 1 : (kh == 2) ? 2 + kw : -1;
            float outv = 0.f;
            if (t > 0 || (t == 0 && cl)) outv = en * (dw[t < 0 ? 0 : t][it] - (v[t < 0 ? 0 : t][it] * rn) * du);
            dwo[(size_t)ci * 9 + pos] = outv;
        }
    }
    if (cs == 0) {
#pragma unroll
        for (int pos = 0; pos < 9; ++pos) {
            const int q = flip ? 8 - pos : pos;
            const int kh = q / 3, kw = q % 3;
            const int t = (kh == 1 && kw == 1) ? 0 : (kh == 1 && kw == 2) ? 1 : (kh == 2) ? 2 + kw : -1;
            float outv = 0.f;
            if (t > 0) outv = en * (dwb[t - 1] - (wb[t - 1] * rn) * du);
            else if (t == 0 && centre_in_norm) outv = en * (0.f - (wc * rn) * du);
            dwo[(size_t)n_in * 9 + pos] = outv;
        }
    }
}

template <int DUMMY = 0>
__device__ __forceinline__ void wn_bwd_dispatch(const WnBwdLayer& L, int tile, float (*red)[16][17], float* s_n, float* s_dot) {
    if (L.variant != IAF_VARIANT_TF) {
        switch (L.cin >> 4) {
            case 1: wn_bwd_tile_theano<1>(L, tile, red, s_n, s_dot); break;
            case 2: wn_bwd_tile_theano<2>(L, tile, red, s_n, s_dot); break;
            case 3: wn_bwd_tile_theano<3>(L, tile, red, s_n, s_dot); break;
            case 4: wn_bwd_tile_theano<4>(L, tile, red, s_n, s_dot); break;
            case 5: wn_bwd_tile_theano<5>(L, tile, red, s_n, s_dot); break;
            case 6: wn_bwd_tile_theano<6>(L, tile, red, s_n, s_dot); break;
            case 7: wn_bwd_tile_theano<7>(L, tile, red, s_n, s_dot); break;
            case 8: wn_bwd_tile_theano<8>(L, tile, red, s_n, s_dot); break;
            case 9: wn_bwd_tile_theano<9>(L, tile, red, s_n, s_dot); break;
            case 10: wn_bwd_tile_theano<10>(L, tile, red, s_n, s_dot); break;
            case 11: wn_bwd_tile_theano<11>(L, tile, red, s_n, s_dot); break;
            case 12: wn_bwd_tile_theano<12>(L, tile, red, s_n, s_dot); break;
            case 13: wn_bwd_tile_theano<13>(L, tile, red, s_n, s_dot); break;
            case 14: wn_bwd_tile_theano<14>(L, tile, red, s_n, s_dot); break;
            case 15: wn_bwd_tile_theano<15>(L, tile, red, s_n, s_dot); break;
            case 16: wn_bwd_tile_theano<16>(L, tile, red, s_n, s_dot); break;
        }
        return;
    }
    switch (L.cin >> 4) {
        case 1: wn_bwd_tile<1>(L, tile, red, s_n, s_dot); break;
        case 2: wn_bwd_tile<2>(L, tile, red, s_n, s_dot); break;
        case 3: wn_bwd_tile<3>(L, tile, red, s_n, s_dot); break;
        case 4: wn_bwd_tile<4>(L, tile, red, s_n, s_dot); break;
        case 5: wn_bwd_tile<5>(L, tile, red, s_n, s_dot); break;
        case 6: wn_bwd_tile<6>(L, tile, red, s_n, s_dot); break;
        case 7: wn_bwd_tile<7>(L, tile, red, s_n, s_dot); break;
        case 8: wn_bwd_tile<8>(L, tile, red, s_n, s_dot); break;
        case 9: wn_bwd_tile<9>(L, tile, red, s_n, s_dot); break;
        case 10: wn_bwd_tile<10>(L, tile, red, s_n, s_dot); break;
        case 11: wn_bwd_tile<11>(L, tile, red, s_n, s_dot); break;
        case 12: wn_bwd_tile<12>(L, tile, red, s_n, s_dot); break;
        case 13: wn_bwd_tile<13>(L, tile, red, s_n, s_dot); break;
        case 14: wn_bwd_tile<14>(L, tile, red, s_n, s_dot); break;
        case 15: wn_bwd_tile<15>(L, tile, red, s_n, s_dot); break;
        case 16: wn_bwd_tile<16>(L, tile, red, s_n, s_dot); break;
    }
}

struct WnBwdArgs {
    WnBwdLayer L[MAX_GEMM_LAYERS + 1];   // one entry per conv (the output pair counts twice)
    int tile_begin[MAX_GEMM_LAYERS + 2];
    int n;
};

// every conv of a stack in one launch: workgroup -> (conv, 16-channel output tile)
__global__ __launch_bounds__(256) void iaf_wn_bwd_kernel(WnBwdArgs a) {
    __shared__ float red[3 + NTAPS - 1][16][17];
    __shared__ float s_n[16], s_dot[16];
    int li = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.tile_begin[i]) li = i;
    wn_bwd_dispatch(a.L[li], blockIdx.x - a.tile_begin[li], red, s_n, s_dot);
}

// the same for the convs of MANY stacks (a whole model) in one launch: descriptors in device memory.  A stack's own launch
// has 24 workgroups for 256 CUs and is bound by one CU's address unit (~21 us); batched, the model's 480 tiles cost about
// the same as one stack alone.
__global__ __launch_bounds__(256) void iaf_wn_bwd_batch_kernel(const WnBwdLayer* __restrict__ layers, const int* __restrict__ tile2layer,
                                                              const int* __restrict__ tile_begin) {
    __shared__ float red[3 + NTAPS - 1][16][17];
    __shared__ float s_n[16], s_dot[16];
    const int li = tile2layer[blockIdx.x];
    const WnBwdLayer L = layers[li];
    wn_bwd_dispatch(L, blockIdx.x - tile_begin[li], red, s_n, s_dot);
}

// plain convs: NCHW -> pixel-major staging of the backward operands.  dst[P][C] = scale * act(concat_k src_k)[b, c, p]
// (act = ELU when elu is set).  Up to MAXSPLIT sources; boundaries are multiples of 4.  A workgroup moves a (64 pixels x 64 channels)
// tile: the reads are coalesced along pixels (a wave = 64 consecutive pixels of one channel, 16 loads in flight per thread), the tile
// turns in LDS, and the writes run along channels -- 16 lanes cover the 256 contiguous bytes of a pixel (round 5; before, a thread
// wrote the 16 bytes it had read, 64 lanes = 64 pixels = 64 separate 16-byte pieces C floats apart).  blockIdx.z picks one of up to
// two tensors: a conv's two operands (dY and [elu](x)) are packed by ONE launch.
struct PackP {
    const float* src[MAXSPLIT]; int end[MAXSPLIT]; int nsrc;
    float* dst; int C, HW, P; float scale; int elu;
    float* colsum;      // or NULL: [ceil(P / 64)][C] column sums of the packed tile (dY: the bias gradient's partials, one row per workgroup)
};
struct PackP2 { PackP t[2]; };
__global__ __launch_bounds__(256) void iaf_pack_pixmajor_kernel(PackP2 pp) {
    __shared__ float tile[64][65];
    PackP p = pp.t[0];                                       // (a uniform select: indexing the by-value block at run time would copy it to scratch)
    if (blockIdx.z) p = pp.t[1];
    const int cb = blockIdx.y * 64;
    if (cb >= p.C) return;                                   // (the other tensor of the launch has more channels)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int px0 = blockIdx.x * 64, px = px0 + lane;
    if (px < p.P) {
        const int b = px / p.HW, pix = px - b * p.HW;
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = cb + 4 * (wv + 4 * u);
            if (c < p.C) {                                   // (wave-uniform)
                const float* sk = p.src[0];                  // source k of the concat and its channel range [c0, c1): static indices
                int c0 = 0, c1 = p.end[0];
#pragma unroll
                for (int k = 1; k < MAXSPLIT; ++k)
                    if (k < p.nsrc && c >= p.end[k - 1]) { sk = p.src[k]; c0 = p.end[k - 1]; c1 = p.end[k]; }
                const float* s = sk + ((size_t)b * (c1 - c0) + (c - c0)) * p.HW + pix;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[u][r] = s[(size_t)r * p.HW];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = wv + 4 * u;
            if (cb + 4 * q < p.C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = v[u][r];
                    if (p.elu) t = elu_f(t);
                    tile[lane][4 * q + r] = t * p.scale;
                }
            }
        }
    }
    __syncthreads();
    if (p.colsum && threadIdx.x < 64 && cb + (int)threadIdx.x < p.C) {       // (the first wave; the others go on to the stores)
        const int n = p.P - px0 < 64 ? p.P - px0 : 64, t = threadIdx.x;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = 0;
        for (; r + 4 <= n; r += 4) { a0 += tile[r][t]; a1 += tile[r + 1][t]; a2 += tile[r + 2][t]; a3 += tile[r + 3][t]; }
        for (; r < n; ++r) a0 += tile[r][t];
        p.colsum[(size_t)blockIdx.x * p.C + cb + t] = (a0 + a1) + (a2 + a3);
    }
    const int ql = threadIdx.x & 15, c = cb + 4 * ql;
    if (c >= p.C) return;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int pl = pass * 16 + (threadIdx.x >> 4);
        if (px0 + pl < p.P) {
            const f32x4 o = {tile[pl][4 * ql], tile[pl][4 * ql + 1], tile[pl][4 * ql + 2], tile[pl][4 * ql + 3]};
            *(f32x4*)(p.dst + (size_t)(px0 + pl) * p.C + c) = o;
        }
    }
}

// weight-norm backward of a plain (unmasked, 9-tap) conv, layers.py:60:  w = e u, u = V/n, n = ||V||_o, e = exp(g)
//   dg = sum dW w;  dV = (e/n)(dW - u (sum dW u));  db = sum_p dY.   Same thread map as wn_bwd_tile; own kernel (18*NCH
//   live registers per thread would halve the occupancy of the masked one).
template <int NCH>
__device__ __forceinline__ void wn_bwd_plain_tile(const WnBwdLayer& L, int tile, float (*red)[16][17], float* s_n, float* s_dot) {
    const int oo = threadIdx.x & 15, cs = threadIdx.x >> 4;
    const int o = tile * 16 + oo;
    const int n_in = L.cin, n_out = L.cout;
    float v[MAXTAPS][NCH], dw[MAXTAPS][NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int ci = cs + 16 * it;
#pragma unroll
        for (int t = 0; t < MAXTAPS; ++t) {
            v[t][it] = L.V[((size_t)t * n_in + ci) * n_out + o];
            dw[t][it] = L.dW[((size_t)t * n_in + ci) * L.cout_packed + o];
        }
    }
    float dbs = 0.f;
    for (int r = cs; r < L.nslab; r += 16) dbs += L.dbp[(size_t)r * L.cout_packed + o];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int it = 0; it < NCH; ++it)
#pragma unroll
        for (int t = 0; t < MAXTAPS; ++t) { ss += v[t][it] * v[t][it]; dot += dw[t][it] * v[t][it]; }
    red[0][cs][oo] = ss; red[1][cs][oo] = dot; red[2][cs][oo] = dbs;
    __syncthreads();
    if (cs == 0) {
        float a = 0.f, b = 0.f, c = 0.f;
        for (int i = 0; i < 16; ++i) { a += red[0][i][oo]; b += red[1][i][oo]; c += red[2][i][oo]; }
        const float n = sqrtf(fmaxf(a, 1e-12f));
        s_n[oo] = n;
        s_dot[oo] = b / n;
        L.dg[o] = expf(L.g[o]) * b / n;
        L.db[o] = c;
    }
    __syncthreads();
    const float n = s_n[oo], du = s_dot[oo], e = expf(L.g[o]);
#pragma unroll
    for (int it = 0; it < NCH; ++it)
#pragma unroll
        for (int t = 0; t < MAXTAPS; ++t)
            L.dV[((size_t)t * n_in + cs + 16 * it) * n_out + o] = (e / n) * (dw[t][it] - (v[t][it] / n) * du);
}

__global__ __launch_bounds__(256) void iaf_wn_bwd_plain_kernel(WnBwdLayer L) {
    __shared__ float red[3][16][17];
    __shared__ float s_n[16], s_dot[16];
    switch (L.cin >> 4) {
        case 1: wn_bwd_plain_tile<1>(L, blockIdx.x, red, s_n, s_dot); break;
        case 2: wn_bwd_plain_tile<2>(L, blockIdx.x, red, s_n, s_dot); break;
        case 3: wn_bwd_plain_tile<3>(L, blockIdx.x, red, s_n, s_dot); break;
        case 4: wn_bwd_plain_tile<4>(L, blockIdx.x, red, s_n, s_dot); break;
        case 5: wn_bwd_plain_tile<5>(L, blockIdx.x, red, s_n, s_dot); break;
        case 6: wn_bwd_plain_tile<6>(L, blockIdx.x, red, s_n, s_dot); break;
        case 7: wn_bwd_plain_tile<7>(L, blockIdx.x, red, s_n, s_dot); break;
        case 8: wn_bwd_plain_tile<8>(L, blockIdx.x, red, s_n, s_dot); break;
        case 9: wn_bwd_plain_tile<9>(L, blockIdx.x, red, s_n, s_dot); break;
        case 10: wn_bwd_plain_tile<10>(L, blockIdx.x, red, s_n, s_dot); break;
        case 11: wn_bwd_plain_tile<11>(L, blockIdx.x, red, s_n, s_dot); break;
        case 12: wn_bwd_plain_tile<12>(L, blockIdx.x, red, s_n, s_dot); break;
        case 13: wn_bwd_plain_tile<13>(L, blockIdx.x, red, s_n, s_dot); break;
        case 14: wn_bwd_plain_tile<14>(L, blockIdx.x, red, s_n, s_dot); break;
        case 15: wn_bwd_plain_tile<15>(L, blockIdx.x, red, s_n, s_dot); break;
        case 16: wn_bwd_plain_tile<16>(L, blockIdx.x, red, s_n, s_dot); break;
    }
}

__global__ __launch_bounds__(256) void iaf_wn_bwd_plain_batch_kernel(const WnBwdLayer* __restrict__ layers,
                                                                    const int* __restrict__ tile2layer,
                                                                    const int* __restrict__ tile_begin) {
    __shared__ float red[3][16][17];
    __shared__ float s_n[16], s_dot[16];
    const int li = tile2layer[blockIdx.x];
    const WnBwdLayer L = layers[li];
    if (L.skip) return;
    const int tile = blockIdx.x - tile_begin[li];
    switch (L.cin >> 4) {
        case 1: wn_bwd_plain_tile<1>(L, tile, red, s_n, s_dot); break;
        case 2: wn_bwd_plain_tile<2>(L, tile, red, s_n, s_dot); break;
        case 3: wn_bwd_plain_tile<3>(L, tile, red, s_n, s_dot); break;
        case 4: wn_bwd_plain_tile<4>(L, tile, red, s_n, s_dot); break;
        case 5: wn_bwd_plain_tile<5>(L, tile, red, s_n, s_dot); break;
        case 6: wn_bwd_plain_tile<6>(L, tile, red, s_n, s_dot); break;
        case 7: wn_bwd_plain_tile<7>(L, tile, red, s_n, s_dot); break;
        case 8: wn_bwd_plain_tile<8>(L, tile, red, s_n, s_dot); break;
        case 9: wn_bwd_plain_tile<9>(L, tile, red, s_n, s_dot); break;
        case 10: wn_bwd_plain_tile<10>(L, tile, red, s_n, s_dot); break;
        case 11: wn_bwd_plain_tile<11>(L, tile, red, s_n, s_dot); break;
        case 12: wn_bwd_plain_tile<12>(L, tile, red, s_n, s_dot); break;
        case 13: wn_bwd_plain_tile<13>(L, tile, red, s_n, s_dot); break;
        case 14: wn_bwd_plain_tile<14>(L, tile, red, s_n, s_dot); break;
        case 15: wn_bwd_plain_tile<15>(L, tile, red, s_n, s_dot); break;
        case 16: wn_bwd_plain_tile<16>(L, tile, red, s_n, s_dot); break;
    }
}

// (4) posterior block backward, elementwise parts (tf_train.py:56-85 differentiated):
//   pre : dkl[b,c,:,:] = G[b,c]  (free bits: G = gate[c] (sum_b' dkl_obj[b']) / B, gate = [mean_b S[b,c] > kl_min];
//                                  kl_min <= 0: G = dkl_obj[b]);
//         z0 = mean + e^{lq} eps;  d z_tot = dz + dkl (z - pm) e^{-2 pl};  d pm = -dkl (z - pm) e^{-2 pl};
//         d pl = dkl (1 - (z - pm)^2 e^{-2 pl});   core inputs: dz_new := dz_tot, dlogsd := dkl  (logqs += s)
//   post: d mean = dz0;  d lq = dz0 (z0 - mean) - dkl     (d logq0/d mean = 0 and d logq0/d lq = -1 after the
//         reparametrisation paths cancel analytically)
// gate[c] in {0,1} comes from the forward (iaf_kl_finish_kernel); the common factor (sum_b dkl_obj[b]) / B is rebuilt per
// workgroup from the B-vector (a separate one-block launch used to do both).
__global__ __launch_bounds__(256) void iaf_post_bwd_pre_kernel(const float* qm, const float* ql, const float* rm, const float* rl,
                                                              const float* pm, const float* pl, const float* eps, const float* z,
                                                              const float* dz, const float* gate, const float* dkl_obj, float kl_min,
                                                              float* z0, float* dzt, float* dkl, float* dpm, float* dpl, int B, int Z,
                                                              int HW, size_t n) {
    __shared__ float s_part[4];
    float gsum = 0.f;
    if (kl_min > 0.f) {
        float a = 0.f;
        for (int b = threadIdx.x; b < B; b += 256) a += dkl_obj[b];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = a;
        __syncthreads();
        gsum = ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) / (float)B;
    }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = i / HW;
        const int c = (int)(bc % Z);
        const size_t b = bc / Z;
        const float g = (kl_min > 0.f) ? gate[c] * gsum : dkl_obj[b];
        const float mean = qm[i] + rm[i], lq = ql[i] + rl[i];
        z0[i] = mean + __expf(0.5f * (2.f * lq)) * eps[i];
        const float e2 = __expf(-2.f * pl[i]);
        const float d = z[i] - pm[i];
        const float gz = dz ? dz[i] : 0.f;
        dzt[i] = gz + g * d * e2;
        dkl[i] = g;
        dpm[i] = -g * d * e2;
        dpl[i] = g * (1.f - d * d * e2);
    }
}

__global__ __launch_bounds__(256) void iaf_post_bwd_post_kernel(const float* qm, const float* rm, const float* z0, const float* dz0,
                                                               const float* dkl, float* dmean, float* dlq, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = dz0[i];
        dmean[i] = g;
        dlq[i] = g * (z0[i] - (qm[i] + rm[i])) - dkl[i];
    }
}
