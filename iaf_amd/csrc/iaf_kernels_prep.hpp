// iaf_kernels_prep.hpp -- weight prep kernels: MADE mask rule, weight norm, repack into MFMA fragment order (masked 5-tap, Theano, plain 9-tap).
// Part of the single translation unit iaf_engine.hip (included there, in order; not a standalone header).
#pragma once

// ---------------------------------------------------------------------------------------------
// MADE channel mask rule, tf_utils/layers.py:115-131 (Python-2 integer division)
// ---------------------------------------------------------------------------------------------
__host__ __device__ static inline bool made_live(int i, int o, int n_in, int n_out, int zerodiag) {
    if (n_out >= n_in) {
        const int k = n_out / n_in;
        const int grp = o / k;                       // out-group index == highest visible input
        return zerodiag ? (i < grp) : (i <= grp);
    }
    const int k = n_in / n_out;
    return zerodiag ? (i < o * k) : (i < (o + 1) * k);
}

// ---------------------------------------------------------------------------------------------
// weight prep kernel: one workgroup per packed 16-channel output tile
// ---------------------------------------------------------------------------------------------

// Work split: one workgroup per (packed 16-channel output tile); thread (oo = tid&15, cs = tid>>4) owns output
// channel o = tile*16+oo and input channels ci = cs, cs+16, ...  All of its 5*NCH filter taps are fetched in ONE
// batch of independent, branch-free loads, kept in registers for the second pass.  NCH (= n_in/16) is a template
// parameter so that exactly the needed loads are issued.
struct PrepLayer {
    const float* V[2];
    const float* g[2];
    const float* b[2];
    float* wp;       // packed weights [chunk][tap][cot][64][4]
    float* bias;     // packed bias [ncot*16]
    float* border;   // Theano variant: [4][ncot*16] normalised weights of the border-indicator channel (taps 1..4)
    float* wpt;      // training: TRANSPOSED pack [chunk over packed c_out][tap][c_in tile][64][4] for dX = W^T dY (or NULL)
    void* wp3;       // bf16x3 pack (iaf_conv_bf3.hpp): [c_in pair of 32][tap][cot][plane h/m/l][lane 64][8 bf16] (or NULL)
    int cin, cout_each, ncot, nchunk, zerodiag, npair, tile_begin, variant;
    void* wp2;       // two-plane fp16 pack ("f16x2", iaf_step_fused.hpp F16): [c_in pair of 32][tap][cot][plane hi / lo 2^11][lane 64][8 fp16] (or NULL)
    unsigned* rng_err;   // ... host-visible word raised when a weight lies beyond fp16's largest finite number (or NULL)
};
struct PrepArgs {
    PrepLayer L[MAX_GEMM_LAYERS];
    int nlayers;
};

// The kernels below read their layer descriptor IN PLACE (a uniform address: scalar loads, pointers stay in SGPRs) and
// select the member of the two-element arrays with ?:.  Round 2's first form took a private copy and indexed the arrays at
// run time, which the compiler lowered to vector loads of the 120-byte struct + a scratch copy (128 B per thread; 1216 B in
// iaf_prep_kernel) with every pointer in VGPRs: 9 us of the 29 us batched launch (A/B on one box, 456.9 vs 465.6 us/step).
#define PREP_PICK(arr, which) ((which) ? (arr)[1] : (arr)[0])

// filter position (kh,kw) of live tap t: the 5 MADE-live taps (centre, right, then the row below), or all 9 row-major
template <int NTP> __device__ __forceinline__ int tap_kh(int t) { return NTP == 9 ? t / 3 : ((t == 0 || t == 1) ? 1 : 2); }
template <int NTP> __device__ __forceinline__ int tap_kw(int t) { return NTP == 9 ? t % 3 : ((t == 0) ? 1 : (t == 1 ? 2 : t - 2)); }

// The bf16x3 pack of one 16-channel output tile (iaf_conv_bf3.hpp): [c_in pair of 32][tap][cot][plane h/m/l][lane 64]
// [8 bf16].  A second pass over the weights with the FRAGMENT's thread mapping -- thread = (lane = kk*16+oo, quarter)
// owns the 8 consecutive input channels of one lane of one (pair, tap) fragment -- so that every store is a full 16
// bytes per lane, 1 KiB contiguous per wave (per-element 2-byte stores from the first pass's mapping doubled the prep
// kernel's time).  Split in two: the LOADS (mask applied) are issued together with the first pass's, before the norm
// reduction they do not depend on; the scale, the three-way split and the stores follow once s_scale[] is known.
#define PREP_BF3_UPQ(NCH) ((((NCH) / 2) * NTAPS + 3) / 4)      // (pair, tap) units per quarter of the workgroup
#define PREP_BF3_UPQ_T(NCH, NTP) ((((NCH) / 2) * (NTP) + 3) / 4)
template <int NCH, int NTP = NTAPS>
__device__ __forceinline__ void prep_bf3_load(const PrepLayer& L, int gt, float (*w)[8]) {
    const int which = (L.npair == 2) ? (gt & 1) : 0;
    const int src_tile = (L.npair == 2) ? (gt >> 1) : gt;
    const float* __restrict__ V = PREP_PICK(L.V, which);
    const int lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const int oo = lane & 15, kk = lane >> 4;
    const int o = src_tile * 16 + oo;
    const int n_out = L.cout_each, n_in = L.cin;
    const bool theano = (L.variant == IAF_VARIANT_THEANO || L.variant == IAF_VARIANT_THEANO_FLIPMASK);
    const bool flip = (L.variant == IAF_VARIANT_THEANO_FLIPMASK);
    const int k0 = (n_out >= n_in) ? n_out / n_in : 1;
    const bool row_zeroed = theano && L.zerodiag && o < k0;         // ar.py:268-276 (see prep_tile_theano)
    constexpr int NUNIT = (NCH / 2) * NTP;
#pragma unroll
    for (int i = 0; i < PREP_BF3_UPQ_T(NCH, NTP); ++i) {
        const int uu = quarter + 4 * i;
        const int u = uu < NUNIT ? uu : NUNIT - 1;                  // clamped: the surplus slot is loaded, never stored
        const int pair = u / NTP, t = u - pair * NTP;
        const int kh = tap_kh<NTP>(t), kw = tap_kw<NTP>(t);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = pair * 32 + 8 * kk + e;
            float x;
            bool live = true;
            if (NTP == MAXTAPS) {                                    // plain conv2d: every tap, no mask
                x = V[((size_t)(kh * 3 + kw) * n_in + ci) * n_out + o];
            } else if (theano) {
                x = V[((size_t)o * (n_in + 1) + ci) * 9 + (flip ? (2 - kh) * 3 + (2 - kw) : kh * 3 + kw)];
                if (t == 0)
                    live = !row_zeroed && (flip ? (ci >= 1 && made_live(n_in - ci, n_out - 1 - o, n_in, n_out, L.zerodiag))
                                                : made_live(ci, o, n_in, n_out, L.zerodiag));
            } else {
                x = V[((size_t)(kh * 3 + kw) * n_in + ci) * n_out + o];
                if (t == 0) live = made_live(ci, o, n_in, n_out, L.zerodiag);
            }
            w[i][e] = live ? x : 0.f;
        }
    }
}

template <int NCH, int NTP = NTAPS>
__device__ __forceinline__ void prep_bf3_store(const PrepLayer& L, int gt, const float (*w)[8], const float* s_scale) {
    typedef __bf16 pb16x2 __attribute__((ext_vector_type(2)));
    typedef float pf32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const float scale = s_scale[lane & 15];
    constexpr int NUNIT = (NCH / 2) * NTP;
#pragma unroll
    for (int i = 0; i < PREP_BF3_UPQ_T(NCH, NTP); ++i) {
        const int u = quarter + 4 * i;
        if (u >= NUNIT) continue;
        const int pair = u / NTP, t = u - pair * NTP;
        pu32x4 ph, pm, pl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const pf32x2 x = {w[i][2 * k] * scale, w[i][2 * k + 1] * scale};
            const pb16x2 hb = __builtin_convertvector(x, pb16x2);
            const pf32x2 r1 = x - __builtin_convertvector(hb, pf32x2);
            const pb16x2 mb = __builtin_convertvector(r1, pb16x2);
            const pf32x2 r2 = r1 - __builtin_convertvector(mb, pf32x2);
            const pb16x2 lb = __builtin_convertvector(r2, pb16x2);
            ph[k] = __builtin_bit_cast(unsigned, hb); pm[k] = __builtin_bit_cast(unsigned, mb); pl[k] = __builtin_bit_cast(unsigned, lb);
        }
        pu32x4* q = (pu32x4*)L.wp3 + (((size_t)(pair * NTP + t) * L.ncot + gt) * 3) * 64 + lane;
        q[0] = ph; q[64] = pm; q[128] = pl;
    }
}

// the two-plane fp16 pack of the same tile from the same registers: x = hi + lo' 2^-11 (iaf_conv_bf3.hpp, f16s_split2)
template <int NCH, int NTP = NTAPS>
__device__ __forceinline__ void prep_f16_store(const PrepLayer& L, int gt, const float (*w)[8], const float* s_scale) {
    typedef _Float16 ph16x2 __attribute__((ext_vector_type(2)));
    typedef float pf32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, quarter = threadIdx.x >> 6;
    const float scale = s_scale[lane & 15];
    constexpr int NUNIT = (NCH / 2) * NTP;
    float big = 0.f;
#pragma unroll
    for (int i = 0; i < PREP_BF3_UPQ_T(NCH, NTP); ++i) {
        const int u = quarter + 4 * i;
        if (u >= NUNIT) continue;
        const int pair = u / NTP, t = u - pair * NTP;
        pu32x4 ph, pl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const pf32x2 x = {w[i][2 * k] * scale, w[i][2 * k + 1] * scale};
            big = fmaxf(big, fmaxf(__builtin_fabsf(x[0]), __builtin_fabsf(x[1])));
            const ph16x2 hb = __builtin_convertvector(x, ph16x2);
            const pf32x2 r = (x - __builtin_convertvector(hb, pf32x2)) * 2048.0f;
            const ph16x2 lb = __builtin_convertvector(r, ph16x2);
            ph[k] = __builtin_bit_cast(unsigned, hb); pl[k] = __builtin_bit_cast(unsigned, lb);
        }
        pu32x4* q = (pu32x4*)L.wp2 + (((size_t)(pair * NTP + t) * L.ncot + gt) * 2) * 64 + lane;
        q[0] = ph; q[64] = pl;
    }
    if (big > 65504.0f && L.rng_err) __hip_atomic_fetch_or(L.rng_err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Round 6 (the batched prep at 15.5 us against a ~5 us HBM floor).  A TF-statement layer that keeps ONLY split packs
// (bf16x3 and / or two-plane fp16: every stack of the headline run) and whose tile fits the launch's dynamic LDS: the 16 output channels of
// a tile are 64 contiguous bytes of every V row [kh][kw][c_in][n_out] -- four lanes x 16 bytes -- so a wave instruction fetches 16 whole
// rows (1 KiB, against 4 x 64 B of 4-byte items in the fragment's own thread mapping: 15 loads per thread at n_in = 160 instead of 56),
// the tile goes through LDS once, unscaled, [row = tap n_in + c_in][16 o] with one pad row per 8 (the fragment's lanes (kk, oo) then read
// 8 rows 16-byte-strided from four different bank groups: conflict-free), and the sum of squares is taken on the way (wave shuffles +
// 4 x 16 partials).  The fragment registers come back in prep_bf3_load's mapping, so the split + store code is the same.
template <int NCH>
__device__ __forceinline__ void prep_tile_fast(const PrepLayer& L, int gt, float (*red)[17], float* s_scale, float* tile) {
    typedef float pf32x4 __attribute__((ext_vector_type(4)));
    constexpr int NIN = 16 * NCH, NP = (NIN + 63) / 64;
    const int which = (L.npair == 2) ? (gt & 1) : 0;
    const int src_tile = (L.npair == 2) ? (gt >> 1) : gt;
    const float* __restrict__ V = PREP_PICK(L.V, which);
    const int tid = threadIdx.x, c4 = tid & 3, r = tid >> 2;
    const int n_out = L.cout_each, n_in = L.cin;
    const int o4 = src_tile * 16 + 4 * c4;
    pf32x4 v[NTAPS][NP];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int ci = r + 64 * j;
            const int cic = (NIN % 64 == 0 || ci < NIN) ? ci : 0;
            v[t][j] = *(const pf32x4*)(V + ((size_t)(tap_kh<NTAPS>(t) * 3 + tap_kw<NTAPS>(t)) * n_in + cic) * n_out + o4);
        }
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int ci = r + 64 * j;
            if (NIN % 64 != 0 && ci >= NIN) continue;
            pf32x4 x = v[t][j];
            if (t == 0) {                                           // centre tap: channel MADE mask (layers.py:57)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!made_live(ci, o4 + e, n_in, n_out, L.zerodiag)) x[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) ss[e] += x[e] * x[e];
            const int row = t * NIN + ci;
            *(pf32x4*)(tile + (size_t)(row + (row >> 3)) * 16 + 4 * c4) = x;
        }
    // lanes of a wave with the same c4 (lane bits 2..5) hold partial sums of the same four output channels
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = ss[e];
        a += __shfl_xor(a, 4); a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        ss[e] = a;
    }
    if ((tid & 63) < 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tid >> 6][4 * c4 + e] = ss[e];
    }
    __syncthreads();
    if (tid < 16) {
        const int o = src_tile * 16 + tid;
        const float tot = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        s_scale[tid] = expf(PREP_PICK(L.g, which)[o]) / sqrtf(fmaxf(tot, 1e-12f));      // w = exp(g) * v / sqrt(max(sum v^2, 1e-12))
        L.bias[gt * 16 + tid] = PREP_PICK(L.b, which)[o];
    }
    float w3[PREP_BF3_UPQ(NCH)][8];
    {
        const int lane = tid & 63, quarter = tid >> 6, oo = lane & 15, kk = lane >> 4;
        constexpr int NUNIT = (NCH / 2) * NTAPS;
#pragma unroll
        for (int i = 0; i < PREP_BF3_UPQ(NCH); ++i) {
            const int uu = quarter + 4 * i;
            const int u = uu < NUNIT ? uu : NUNIT - 1;              // (the surplus slot: read, never stored)
            const int pair = u / NTAPS, t = u - pair * NTAPS;
            const int row0 = t * NIN + pair * 32 + 8 * kk;          // a multiple of 8: its 8 rows share one pad offset
            const float* src = tile + (size_t)(row0 + (row0 >> 3)) * 16 + oo;
#pragma unroll
            for (int e = 0; e < 8; ++e) w3[i][e] = src[16 * e];
        }
    }
    __syncthreads();
    if (L.wp3) prep_bf3_store<NCH, NTAPS>(L, gt, w3, s_scale);
    if (L.wp2) prep_f16_store<NCH, NTAPS>(L, gt, w3, s_scale);
}

template <int NCH, int NTP = NTAPS>
__device__ __forceinline__ void prep_tile(const PrepLayer& L, int gt, float (*red)[17], float* s_scale, float* fast_tile = nullptr) {
    const int which = (L.npair == 2) ? (gt & 1) : 0;     // output pair: even tiles = mean, odd = logsd
    const int src_tile = (L.npair == 2) ? (gt >> 1) : gt;
    const float* __restrict__ V = PREP_PICK(L.V, which);
    const int oo = threadIdx.x & 15, cs = threadIdx.x >> 4;
    const int o = src_tile * 16 + oo;
    const int n_out = L.cout_each, n_in = L.cin;
    const float gval = PREP_PICK(L.g, which)[o], bval = PREP_PICK(L.b, which)[o];

    constexpr bool BF3 = (NCH % 2 == 0);
    if constexpr (BF3 && NTP == NTAPS) {
        // split packs only, TF statement, the tile fits the launch's LDS: whole-row loads through LDS (prep_tile_fast)
        if (fast_tile && (L.wp3 || L.wp2) && !L.wp && !L.wpt) { prep_tile_fast<NCH>(L, gt, red, s_scale, fast_tile); return; }
    }
    float w3[BF3 ? PREP_BF3_UPQ_T(NCH, NTP) : 1][8];
    if constexpr (BF3) { if (L.wp3 || L.wp2) prep_bf3_load<NCH, NTP>(L, gt, w3); }
    // pass 1: fetch + mask (layers.py:57), sum of squares over (taps, c_in) (layers.py:60).  A thread owns QUADS of four
    // consecutive input channels, quad q = cs + 16 i (ci = 4q + jj): the four channels of a quad are the four floats a lane
    // of the MFMA B fragment holds, so pass 2 writes them as ONE 16-byte store and a wave as 1 KiB contiguous (one channel
    // per thread meant four dword stores 16 B apart).  Measured neutral on the batched launch (round 2).  Round 3 measured a
    // ONE-pass variant (a thread owning (8 input channels, tap) units, every weight fetched once, both packs written from the
    // same registers): slower on the whole step (0.4861 vs 0.4825 ms, 5 repeats each way) -- removed.
    constexpr int NQI = (NCH + 3) / 4;
    float v[NTP][4 * NQI];
    // Round 4: a stack that keeps ONLY the bf16x3 pack (iaf_stack_set_packs; no fp32 pack, no transposed pack) has every weight of
    // the tile in w3 already -- masked, one (pair, tap) unit of 8 input channels per thread and slot -- so the sum of squares comes
    // from those registers and the second fetch of the tile (60 of the 116 strided 4-byte loads per thread at n_in = 160) is gone.
    const bool bf3_only = BF3 && (L.wp3 || L.wp2) && !L.wp && !L.wpt;
    if (bf3_only) {
        if constexpr (BF3) {
            constexpr int NUNIT = (NCH / 2) * NTP;
            float ssb = 0.f;
#pragma unroll
            for (int i = 0; i < PREP_BF3_UPQ_T(NCH, NTP); ++i) {
                if ((int)(threadIdx.x >> 6) + 4 * i >= NUNIT) continue;          // (the clamped surplus slot: loaded, not part of the tile)
#pragma unroll
                for (int e = 0; e < 8; ++e) ssb += w3[i][e] * w3[i][e];
            }
            red[cs][oo] = ssb;                                    // (cs = thread / 16, oo = thread % 16 = the lane's output channel in both mappings)
            __syncthreads();
            if (cs == 0) {
                float tot = 0.f;
                for (int i = 0; i < 16; ++i) tot += red[i][oo];
                s_scale[oo] = expf(gval) / sqrtf(fmaxf(tot, 1e-12f));
                L.bias[gt * 16 + oo] = bval;
            }
            __syncthreads();
            if (L.wp3) prep_bf3_store<NCH, NTP>(L, gt, w3, s_scale);
            if (L.wp2) prep_f16_store<NCH, NTP>(L, gt, w3, s_scale);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NQI; ++i) {
        const int q = cs + 16 * i;
        const bool have = (NCH % 4 == 0) || q < 4 * NCH;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int ci = have ? 4 * q + jj : jj;
#pragma unroll
            for (int t = 0; t < NTP; ++t) {
                const float x = V[((size_t)(tap_kh<NTP>(t) * 3 + tap_kw<NTP>(t)) * n_in + ci) * n_out + o];
                v[t][4 * i + jj] = have ? x : 0.f;
            }
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NQI; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int ci = 4 * (cs + 16 * i) + jj;
            if (NTP == NTAPS && !made_live(ci, o, n_in, n_out, L.zerodiag)) v[0][4 * i + jj] = 0.f;   // centre tap: channel MADE mask
#pragma unroll
            for (int t = 0; t < NTP; ++t) ss += v[t][4 * i + jj] * v[t][4 * i + jj];
        }
    red[cs][oo] = ss;
    __syncthreads();
    if (cs == 0) {
        float tot = 0.f;
        for (int i = 0; i < 16; ++i) tot += red[i][oo];
        // w = exp(g) * v / sqrt(max(sum v^2, 1e-12))
        s_scale[oo] = expf(gval) / sqrtf(fmaxf(tot, 1e-12f));
        L.bias[gt * 16 + oo] = bval;
    }
    __syncthreads();
    const float scale = s_scale[oo];
    // pass 2: write fragment-ordered weights.  lane = kk*16+oo of chunk `it` holds channels it*16+4kk+{0..3} = quad q with
    // it = q >> 2, kk = q & 3; a wave's four quads are the four kk of one chunk.
    typedef float pf32x4 __attribute__((ext_vector_type(4)));
    if (L.wp)                  // (NULL: the stack only feeds bf16x3 kernels, iaf_stack_set_packs)
#pragma unroll
    for (int i = 0; i < NQI; ++i) {
        const int q = cs + 16 * i;
        if ((NCH % 4 != 0) && q >= 4 * NCH) continue;
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            const pf32x4 w4 = {v[t][4 * i] * scale, v[t][4 * i + 1] * scale, v[t][4 * i + 2] * scale, v[t][4 * i + 3] * scale};
            ((pf32x4*)L.wp)[((((size_t)(q >> 2) * NTP + t) * L.ncot + gt) * 64 + (q & 3) * 16 + oo)] = w4;
        }
    }
    if (L.wpt) {   // dgrad operand: K runs over the packed output channels (chunk = gt), N over input tiles (q >> 2)
#pragma unroll
        for (int i = 0; i < NQI; ++i) {
            const int q = cs + 16 * i;
            if ((NCH % 4 != 0) && q >= 4 * NCH) continue;
#pragma unroll
            for (int t = 0; t < NTP; ++t)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    L.wpt[((((size_t)gt * NTP + t) * NCH + (q >> 2)) * 64 + (oo >> 2) * 16 + 4 * (q & 3) + jj) * 4 + (oo & 3)] =
                        v[t][4 * i + jj] * scale;
        }
    }
    if constexpr (BF3) {
        if (L.wp3) prep_bf3_store<NCH, NTP>(L, gt, w3, s_scale);
        if (L.wp2) prep_f16_store<NCH, NTP>(L, gt, w3, s_scale);
    }
}

// Theano statement of the same weights (graphy/nodes/ar.py:243-330, l2norm=True, logscale=True, pad_channel=True):
//   w is OIHW [n_out][n_in+1][3][3] (last input channel = border indicator, graphy/nodes/conv.py:71-83),
//   kerns = mask*w;  kerns /= (sqrt(sum_{i,h,w} kerns^2) + 1e-8);  kerns *= exp(3*s)          (ar.py:312-317, 279-281)
//   the conv is a TRUE convolution (dnn_conv conv_mode='conv'), so filter position (kh,kw) meets the input at
//   (dh,dw) = (1-kh, 1-kw): the same 5 live filter positions as the TF statement, looking left/above.
// L.V = w, L.g = s, L.b = b.  The border channel never enters the GEMM: its 4 non-centre taps go to L.border and are
// added by the conv epilogue where a tap falls outside the image (its centre tap is always masked).
template <int NCH>
__device__ __forceinline__ void prep_tile_theano(const PrepLayer& L, int gt, float (*red)[17], float* s_scale) {
    const int which = (L.npair == 2) ? (gt & 1) : 0;
    const int src_tile = (L.npair == 2) ? (gt >> 1) : gt;
    const float* __restrict__ Wt = PREP_PICK(L.V, which);
    const int oo = threadIdx.x & 15, cs = threadIdx.x >> 4;
    const int o = src_tile * 16 + oo;
    const int n_out = L.cout_each, n_in = L.cin;
    // flipmask=True (ar.py:263-264): mask = mask[::-1,::-1,::-1,::-1] over [n_out, n_in+1, 3, 3].  The spatial flip moves
    // the live filter positions to the mirrored ones (tap t then reads filter position (2-kh, 2-kw) and -- the conv being
    // a true convolution -- looks right/below like the TF statement; the launch picks that geometry); the channel flips
    // turn the centre-tap rule into  live'(i, o) = live(n_in - i, n_out-1-o)  over the n_in+1 channels INCLUDING the
    // border-indicator channel (index n_in): real channel 0 maps to the (always masked) border column, and the border
    // channel's own centre tap maps to column 0 -- it multiplies zeros inside the image but it does enter the l2 norm.
    const bool flip = (L.variant == IAF_VARIANT_THEANO_FLIPMASK);
    const float sval = PREP_PICK(L.g, which)[o], bval = PREP_PICK(L.b, which)[o];
    const float* wo = Wt + (size_t)o * (n_in + 1) * 9;
    constexpr bool BF3 = (NCH % 2 == 0);
    float w3[BF3 ? PREP_BF3_UPQ(NCH) : 1][8];
    if constexpr (BF3) { if (L.wp3) prep_bf3_load<NCH>(L, gt, w3); }
    float v[NTAPS][NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int ci = cs + 16 * it;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int kh = (t == 0 || t == 1) ? 1 : 2;
            const int kw = (t == 0) ? 1 : (t == 1 ? 2 : t - 2);
            v[t][it] = wo[(size_t)ci * 9 + (flip ? (2 - kh) * 3 + (2 - kw) : kh * 3 + kw)];
        }
    }
    float wb[NTAPS - 1];   // border channel, taps 1..4 (thread cs == 0 accounts for it in the norm)
#pragma unroll
    for (int t = 1; t < NTAPS; ++t) {
        const int kh = (t == 1) ? 1 : 2;
        const int kw = (t == 1) ? 2 : t - 2;
        wb[t - 1] = wo[(size_t)n_in * 9 + (flip ? (2 - kh) * 3 + (2 - kw) : kh * 3 + kw)];
    }
    // l2normalize() first zeroes the centre tap of output rows [0, n_out/n_in) (or row 0) when zerodiagonal
    // (ar.py:268-276): a no-op for the plain mask (those rows see nothing), live weights for the flipped one
    const int k0 = (n_out >= n_in) ? n_out / n_in : 1;
    const bool row_zeroed = L.zerodiag && o < k0;
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int ci = cs + 16 * it;
        const bool live = flip ? (ci >= 1 && made_live(n_in - ci, n_out - 1 - o, n_in, n_out, L.zerodiag))
                               : made_live(ci, o, n_in, n_out, L.zerodiag);   // ar.py:249-262 == the TF rule
        if (!live || row_zeroed) v[0][it] = 0.f;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) ss += v[t][it] * v[t][it];
    }
    if (cs == 0) {
#pragma unroll
        for (int t = 0; t < NTAPS - 1; ++t) ss += wb[t] * wb[t];
        if (flip && !row_zeroed && made_live(0, n_out - 1 - o, n_in, n_out, L.zerodiag)) {
            const float wc = wo[(size_t)n_in * 9 + 4];       // border channel, centre tap: in the norm only
            ss += wc * wc;
        }
    }
    red[cs][oo] = ss;
    __syncthreads();
    if (cs == 0) {
        float tot = 0.f;
        for (int i = 0; i < 16; ++i) tot += red[i][oo];
        const float sc = expf(3.0f * sval) / (sqrtf(tot) + 1e-8f);
        s_scale[oo] = sc;
        L.bias[gt * 16 + oo] = bval;
#pragma unroll
        for (int t = 0; t < NTAPS - 1; ++t) L.border[(size_t)t * (L.ncot * 16) + gt * 16 + oo] = wb[t] * sc;
    }
    __syncthreads();
    const float scale = s_scale[oo];
    const int kk = cs >> 2, jj = cs & 3;
    if (L.wp)
#pragma unroll
    for (int it = 0; it < NCH; ++it)
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
            L.wp[((((size_t)it * NTAPS + t) * L.ncot + gt) * 64 + kk * 16 + oo) * 4 + jj] = v[t][it] * scale;
    if (L.wpt) {   // dgrad operand, as in prep_tile (the border channel carries no data gradient)
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int t = 0; t < NTAPS; ++t)
                L.wpt[((((size_t)gt * NTAPS + t) * NCH + it) * 64 + (oo >> 2) * 16 + cs) * 4 + (oo & 3)] = v[t][it] * scale;
    }
    if constexpr (BF3) { if (L.wp3) prep_bf3_store<NCH>(L, gt, w3, s_scale); }
}

#define PREP_MAXI 16   // n_in <= 256
#define PREP_PLAIN9 100   // PrepLayer.variant of a plain (unmasked, 9-tap) TF conv2d
// fast_tile / fast_floats: the launch's dynamic LDS (prep_tile_fast) and its size in floats (0: none)
template <int DUMMY = 0>
__device__ __forceinline__ void prep_dispatch(const PrepLayer& L, int gt, float (*red)[17], float* s_scale, float* fast_tile = nullptr,
                                              unsigned fast_floats = 0) {
    if ((size_t)NTAPS * L.cin * 18 > fast_floats) fast_tile = nullptr;
    if (L.variant == IAF_VARIANT_THEANO || L.variant == IAF_VARIANT_THEANO_FLIPMASK) {
        switch (L.nchunk) {
            case 1: prep_tile_theano<1>(L, gt, red, s_scale); break;
            case 2: prep_tile_theano<2>(L, gt, red, s_scale); break;
            case 3: prep_tile_theano<3>(L, gt, red, s_scale); break;
            case 4: prep_tile_theano<4>(L, gt, red, s_scale); break;
            case 5: prep_tile_theano<5>(L, gt, red, s_scale); break;
            case 6: prep_tile_theano<6>(L, gt, red, s_scale); break;
            case 7: prep_tile_theano<7>(L, gt, red, s_scale); break;
            case 8: prep_tile_theano<8>(L, gt, red, s_scale); break;
            case 9: prep_tile_theano<9>(L, gt, red, s_scale); break;
            case 10: prep_tile_theano<10>(L, gt, red, s_scale); break;
            case 11: prep_tile_theano<11>(L, gt, red, s_scale); break;
            case 12: prep_tile_theano<12>(L, gt, red, s_scale); break;
            case 13: prep_tile_theano<13>(L, gt, red, s_scale); break;
            case 14: prep_tile_theano<14>(L, gt, red, s_scale); break;
            case 15: prep_tile_theano<15>(L, gt, red, s_scale); break;
            case 16: prep_tile_theano<16>(L, gt, red, s_scale); break;
        }
        return;
    }
    switch (L.nchunk) {
        case 1: prep_tile<1>(L, gt, red, s_scale, fast_tile); break;
        case 2: prep_tile<2>(L, gt, red, s_scale, fast_tile); break;
        case 3: prep_tile<3>(L, gt, red, s_scale, fast_tile); break;
        case 4: prep_tile<4>(L, gt, red, s_scale, fast_tile); break;
        case 5: prep_tile<5>(L, gt, red, s_scale, fast_tile); break;
        case 6: prep_tile<6>(L, gt, red, s_scale, fast_tile); break;
        case 7: prep_tile<7>(L, gt, red, s_scale, fast_tile); break;
        case 8: prep_tile<8>(L, gt, red, s_scale, fast_tile); break;
        case 9: prep_tile<9>(L, gt, red, s_scale, fast_tile); break;
        case 10: prep_tile<10>(L, gt, red, s_scale, fast_tile); break;
        case 11: prep_tile<11>(L, gt, red, s_scale, fast_tile); break;
        case 12: prep_tile<12>(L, gt, red, s_scale, fast_tile); break;
        case 13: prep_tile<13>(L, gt, red, s_scale, fast_tile); break;
        case 14: prep_tile<14>(L, gt, red, s_scale, fast_tile); break;
        case 15: prep_tile<15>(L, gt, red, s_scale, fast_tile); break;
        case 16: prep_tile<16>(L, gt, red, s_scale, fast_tile); break;
    }
}

// many stacks in one launch: descriptors live in device memory; tile2layer maps a workgroup to its GEMM layer
// xcdpair != 0 (grid padded to a multiple of 16): tiles 2 m and 2 m + 1 -- the two 64-byte halves of the 128-byte lines of V rows
// [c_in][n_out] -- go to workgroups blockIdx and blockIdx + 8, which the dispatcher places on ONE XCD (round robin over 8): the line is
// fetched into one L2 once instead of into two
__device__ __forceinline__ int prep_virtual_tile(int xcdpair) {
    const int b = blockIdx.x;
    if (!xcdpair) return b;
    const int xcd = b & 7, slot = b >> 3;
    return ((((slot >> 1) << 3) + xcd) << 1) + (slot & 1);
}
__global__ __launch_bounds__(256) void iaf_prep_batch_kernel(const PrepLayer* __restrict__ layers,
                                                            const int* __restrict__ tile2layer, int ntiles, int xcdpair, unsigned fast_floats) {
    __shared__ float red[16][17];
    __shared__ float s_scale[16];
    extern __shared__ __attribute__((aligned(16))) float prep_dyn_lds[];
    const int v = prep_virtual_tile(xcdpair);
    if (v >= ntiles) return;
    const PrepLayer& L = layers[__builtin_amdgcn_readfirstlane(tile2layer[v])];
    prep_dispatch(L, v - L.tile_begin, red, s_scale, prep_dyn_lds, fast_floats);
}

// plain (unmasked, 9-tap) convs: their own kernel so that the 9-tap register footprint does not tax the masked prep.
// tile2layer == NULL: a single layer.
__global__ __launch_bounds__(256) void iaf_prep_plain_kernel(const PrepLayer* __restrict__ layers,
                                                            const int* __restrict__ tile2layer) {
    __shared__ float red[16][17];
    __shared__ float s_scale[16];
    const PrepLayer& L = layers[tile2layer ? __builtin_amdgcn_readfirstlane(tile2layer[blockIdx.x]) : 0];
    const int gt = blockIdx.x - L.tile_begin;
    switch (L.nchunk) {
        case 1: prep_tile<1, MAXTAPS>(L, gt, red, s_scale); break;
        case 2: prep_tile<2, MAXTAPS>(L, gt, red, s_scale); break;
        case 3: prep_tile<3, MAXTAPS>(L, gt, red, s_scale); break;
        case 4: prep_tile<4, MAXTAPS>(L, gt, red, s_scale); break;
        case 5: prep_tile<5, MAXTAPS>(L, gt, red, s_scale); break;
        case 6: prep_tile<6, MAXTAPS>(L, gt, red, s_scale); break;
        case 7: prep_tile<7, MAXTAPS>(L, gt, red, s_scale); break;
        case 8: prep_tile<8, MAXTAPS>(L, gt, red, s_scale); break;
        case 9: prep_tile<9, MAXTAPS>(L, gt, red, s_scale); break;
        case 10: prep_tile<10, MAXTAPS>(L, gt, red, s_scale); break;
        case 11: prep_tile<11, MAXTAPS>(L, gt, red, s_scale); break;
        case 12: prep_tile<12, MAXTAPS>(L, gt, red, s_scale); break;
        case 13: prep_tile<13, MAXTAPS>(L, gt, red, s_scale); break;
        case 14: prep_tile<14, MAXTAPS>(L, gt, red, s_scale); break;
        case 15: prep_tile<15, MAXTAPS>(L, gt, red, s_scale); break;
        case 16: prep_tile<16, MAXTAPS>(L, gt, red, s_scale); break;
    }
}

__global__ __launch_bounds__(256) void iaf_prep_kernel(PrepArgs a, unsigned fast_floats) {
    __shared__ float red[16][17];
    __shared__ float s_scale[16];
    extern __shared__ __attribute__((aligned(16))) float prep_dyn_lds[];
    PrepLayer L = a.L[0];          // (selected with uniform compares: indexing the by-value block at run time copies it to scratch)
#pragma unroll
    for (int i = 1; i < MAX_GEMM_LAYERS; ++i)
        if (i < a.nlayers && (int)blockIdx.x >= a.L[i].tile_begin) L = a.L[i];
    prep_dispatch(L, blockIdx.x - L.tile_begin, red, s_scale, prep_dyn_lds, fast_floats);
}

// ---------------------------------------------------------------------------------------------
// training: the TRANSPOSED bf16x3 pack (data gradient dX = W^T dY on the bf16 matrix cores, iaf_conv_bf3.hpp with EPI_DGRAD)
// ---------------------------------------------------------------------------------------------
// Written from the transposed fp32 pack `wpt` the prep kernels above leave behind in training mode:
//   wpt  [K chunk = packed c_out / 16][tap][N tile = c_in / 16][lane = kk * 16 + n][4]   = W^T[n][16 chunk + 4 kk + r]
//   wpt3 [K pair  = packed c_out / 32][tap][N tile][plane h/m/l][lane = kk3 * 16 + n][8 bf16] = W^T[n][32 pair + 8 kk3 + e]
// i.e. the fragment of K pair P, lane (kk3, n) is the two 16-byte rows (kk = 2 (kk3 & 1), +1) of chunk 2 P + (kk3 >> 1):
// an elementwise re-layout + three-way split, one thread per (fragment, lane).  Up to PACKT3_MAX layers per launch.
#define PACKT3_MAX 16
struct PackT3Layer { const float* src; void* dst; int ntp, nct, begin; void* dst2; };   // dst2 (or NULL): the same fragments as two fp16 planes (hi, lo 2^11)
struct PackT3Args { PackT3Layer L[PACKT3_MAX]; int n, total; };

__global__ __launch_bounds__(256) void iaf_pack_t3_kernel(PackT3Args a) {
    typedef float pf32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 pb16x2 __attribute__((ext_vector_type(2)));
    typedef float pf32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= a.total) return;
    PackT3Layer L = a.L[0];
#pragma unroll
    for (int i = 1; i < PACKT3_MAX; ++i)
        if (i < a.n && item >= a.L[i].begin) L = a.L[i];
    const int local = item - L.begin, lane3 = local & 63, frag = local >> 6;
    const int cit = frag % L.nct, pt = frag / L.nct, t = pt % L.ntp, P = pt / L.ntp;
    const int kk3 = lane3 >> 4, n = lane3 & 15;
    const pf32x4* src4 = (const pf32x4*)L.src + ((size_t)((2 * P + (kk3 >> 1)) * L.ntp + t) * L.nct + cit) * 64;
    const pf32x4 a0 = src4[(2 * (kk3 & 1)) * 16 + n], a1 = src4[(2 * (kk3 & 1) + 1) * 16 + n];
    const float w[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    pu32x4 ph, pm, pl;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const pf32x2 x = {w[2 * k], w[2 * k + 1]};
        const pb16x2 hb = __builtin_convertvector(x, pb16x2);
        const pf32x2 r1 = x - __builtin_convertvector(hb, pf32x2);
        const pb16x2 mb = __builtin_convertvector(r1, pb16x2);
        const pf32x2 r2 = r1 - __builtin_convertvector(mb, pf32x2);
        const pb16x2 lb = __builtin_convertvector(r2, pb16x2);
        ph[k] = __builtin_bit_cast(unsigned, hb); pm[k] = __builtin_bit_cast(unsigned, mb); pl[k] = __builtin_bit_cast(unsigned, lb);
    }
    pu32x4* q = (pu32x4*)L.dst + (((size_t)(P * L.ntp + t) * L.nct + cit) * 3) * 64 + lane3;
    q[0] = ph; q[64] = pm; q[128] = pl;
    if (L.dst2) {       // the data gradient on two fp16 planes (iaf_conv_bf3.hpp DG16): weights' magnitudes were checked by the forward pack's prep
        typedef _Float16 ph16x2 __attribute__((ext_vector_type(2)));
        pu32x4 fh, fl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const pf32x2 x = {w[2 * k], w[2 * k + 1]};
            const ph16x2 hb = __builtin_convertvector(x, ph16x2);
            const pf32x2 r = (x - __builtin_convertvector(hb, pf32x2)) * 2048.0f;
            const ph16x2 lb = __builtin_convertvector(r, ph16x2);
            fh[k] = __builtin_bit_cast(unsigned, hb); fl[k] = __builtin_bit_cast(unsigned, lb);
        }
        pu32x4* q2 = (pu32x4*)L.dst2 + (((size_t)(P * L.ntp + t) * L.nct + cit) * 2) * 64 + lane3;
        q2[0] = fh; q2[64] = fl;
    }
}
