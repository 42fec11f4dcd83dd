// iaf_engine.hip -- MI355X (gfx950 / CDNA4) IAF posterior engine: kernels + C ABI (include/iaf_hip.h).
//
// What is computed (reference file:line, relative to the reference tree):
//   weight prep   get_conv_ar_mask + weight-norm          tf_utils/layers.py:134-141, 56-60
//   masked conv   ar_conv2d / ar_multiconv2d               tf_utils/layers.py:144-166, 63-64
//   IAF step      m,s=0.1*out; z=(z-m)/exp(s); logqs+=s    tf_train.py:69-72
//   posterior     sample/logqs/logps/KL/free bits          tf_train.py:56-85, tf_utils/distributions.py:5-25
//   IW bound      logsumexp / compute_lowerbound           tf_utils/distributions.py:35-62
//
// Design (DESIGN.md has the long form).  The MADE mask is exploited as structured sparsity, not
// as a scan: 4 of the 9 taps are dead and are never touched; the centre tap is block-triangular and
// its dead 16x16 blocks are skipped.  Each masked conv is an implicit GEMM on the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32):  D[co][pixel] += W[co][k] * X[k][pixel],  k = (tap, c_in).
//   * X tile: pixel-major [slot][c_in (+8 pad)] in LDS, staged once per workgroup with a one-sided
//     halo (live taps only look right/below); tap shifts are per-lane LDS addresses, image borders
//     are a dedicated all-zero slot -- no predication in the main loop.
//   * W: repacked by the prep kernel into MFMA fragment order [chunk][tap][co_tile][lane][4] so a
//     wave fetches one contiguous 1 KiB global_load_dwordx4 per 4 MFMAs; waves never synchronise
//     inside the K loop.
//   * K order inside a 16-channel chunk is permuted (lane k-slot kk owns channels 4kk..4kk+3) so both
//     operands are 16-byte loads.
//   * epilogues fuse bias, context add, ELU, the 0.1 scaling, the affine transform and the
//     log-det term; hidden activations live in a pixel-major scratch that the next conv stages
//     with straight 16-byte copies.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_set>

#include "iaf_hip.h"

#include "iaf_conv_kernel.hpp"

#define IAF_ABI_VERSION 2   // 2: + iaf_conv3x3_*
#define MAX_GEMM_LAYERS 10   // depth_ar <= 9 hidden + 1 output pair

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
    int cin, cout_each, ncot, nchunk, zerodiag, npair, tile_begin, variant;
};
struct PrepArgs {
    PrepLayer L[MAX_GEMM_LAYERS];
    int nlayers;
};

// filter position (kh,kw) of live tap t: the 5 MADE-live taps (centre, right, then the row below), or all 9 row-major
template <int NTP> __device__ __forceinline__ int tap_kh(int t) { return NTP == 9 ? t / 3 : ((t == 0 || t == 1) ? 1 : 2); }
template <int NTP> __device__ __forceinline__ int tap_kw(int t) { return NTP == 9 ? t % 3 : ((t == 0) ? 1 : (t == 1 ? 2 : t - 2)); }

template <int NCH, int NTP = NTAPS>
__device__ __forceinline__ void prep_tile(const PrepLayer& L, int gt, float (*red)[17], float* s_scale) {
    const int which = (L.npair == 2) ? (gt & 1) : 0;     // output pair: even tiles = mean, odd = logsd
    const int src_tile = (L.npair == 2) ? (gt >> 1) : gt;
    const float* __restrict__ V = L.V[which];
    const int oo = threadIdx.x & 15, cs = threadIdx.x >> 4;
    const int o = src_tile * 16 + oo;
    const int n_out = L.cout_each, n_in = L.cin;
    const float gval = L.g[which][o], bval = L.b[which][o];

    // pass 1: fetch + mask (layers.py:57), sum of squares over (taps, c_in) (layers.py:60)
    float v[NTP][NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int ci = cs + 16 * it;
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
            v[t][it] = V[((size_t)(tap_kh<NTP>(t) * 3 + tap_kw<NTP>(t)) * n_in + ci) * n_out + o];
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        if (NTP == NTAPS && !made_live(cs + 16 * it, o, n_in, n_out, L.zerodiag)) v[0][it] = 0.f;   // centre tap: channel MADE mask
#pragma unroll
        for (int t = 0; t < NTP; ++t) ss += v[t][it] * v[t][it];
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
    // pass 2: write fragment-ordered weights.  lane = kk*16+oo holds channels chunk*16+4kk+{0..3};
    // ci = cs + 16*it  ->  chunk = it, kk = cs>>2, jj = cs&3: a wave writes 256 contiguous bytes.
    const int kk = cs >> 2, jj = cs & 3;
#pragma unroll
    for (int it = 0; it < NCH; ++it)
#pragma unroll
        for (int t = 0; t < NTP; ++t)
            L.wp[((((size_t)it * NTP + t) * L.ncot + gt) * 64 + kk * 16 + oo) * 4 + jj] = v[t][it] * scale;
    if (L.wpt) {   // dgrad operand: K runs over the packed output channels (chunk = gt), N over input tiles (it)
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int t = 0; t < NTP; ++t)
                L.wpt[((((size_t)gt * NTP + t) * NCH + it) * 64 + (oo >> 2) * 16 + cs) * 4 + (oo & 3)] = v[t][it] * scale;
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
    const float* __restrict__ Wt = L.V[which];
    const int oo = threadIdx.x & 15, cs = threadIdx.x >> 4;
    const int o = src_tile * 16 + oo;
    const int n_out = L.cout_each, n_in = L.cin;
    const float sval = L.g[which][o], bval = L.b[which][o];
    const float* wo = Wt + (size_t)o * (n_in + 1) * 9;
    float v[NTAPS][NCH];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int ci = cs + 16 * it;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int kh = (t == 0 || t == 1) ? 1 : 2;
            const int kw = (t == 0) ? 1 : (t == 1 ? 2 : t - 2);
            v[t][it] = wo[(size_t)ci * 9 + kh * 3 + kw];
        }
    }
    float wb[NTAPS - 1];   // border channel, taps 1..4 (thread cs == 0 accounts for it in the norm)
#pragma unroll
    for (int t = 1; t < NTAPS; ++t) {
        const int kh = (t == 1) ? 1 : 2;
        const int kw = (t == 1) ? 2 : t - 2;
        wb[t - 1] = wo[(size_t)n_in * 9 + kh * 3 + kw];
    }
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        if (!made_live(cs + 16 * it, o, n_in, n_out, L.zerodiag)) v[0][it] = 0.f;   // ar.py:249-262 == the TF rule
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) ss += v[t][it] * v[t][it];
    }
    if (cs == 0) {
#pragma unroll
        for (int t = 0; t < NTAPS - 1; ++t) ss += wb[t] * wb[t];
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
#pragma unroll
    for (int it = 0; it < NCH; ++it)
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
            L.wp[((((size_t)it * NTAPS + t) * L.ncot + gt) * 64 + kk * 16 + oo) * 4 + jj] = v[t][it] * scale;
}

#define PREP_MAXI 16   // n_in <= 256
#define PREP_PLAIN9 100   // PrepLayer.variant of a plain (unmasked, 9-tap) TF conv2d
template <int DUMMY = 0>
__device__ __forceinline__ void prep_dispatch(const PrepLayer& L, int gt, float (*red)[17], float* s_scale) {
    if (L.variant == IAF_VARIANT_THEANO) {
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
        case 1: prep_tile<1>(L, gt, red, s_scale); break;
        case 2: prep_tile<2>(L, gt, red, s_scale); break;
        case 3: prep_tile<3>(L, gt, red, s_scale); break;
        case 4: prep_tile<4>(L, gt, red, s_scale); break;
        case 5: prep_tile<5>(L, gt, red, s_scale); break;
        case 6: prep_tile<6>(L, gt, red, s_scale); break;
        case 7: prep_tile<7>(L, gt, red, s_scale); break;
        case 8: prep_tile<8>(L, gt, red, s_scale); break;
        case 9: prep_tile<9>(L, gt, red, s_scale); break;
        case 10: prep_tile<10>(L, gt, red, s_scale); break;
        case 11: prep_tile<11>(L, gt, red, s_scale); break;
        case 12: prep_tile<12>(L, gt, red, s_scale); break;
        case 13: prep_tile<13>(L, gt, red, s_scale); break;
        case 14: prep_tile<14>(L, gt, red, s_scale); break;
        case 15: prep_tile<15>(L, gt, red, s_scale); break;
        case 16: prep_tile<16>(L, gt, red, s_scale); break;
    }
}

// many stacks in one launch: descriptors live in device memory; tile2layer maps a workgroup to its GEMM layer
__global__ __launch_bounds__(256) void iaf_prep_batch_kernel(const PrepLayer* __restrict__ layers,
                                                            const int* __restrict__ tile2layer) {
    __shared__ float red[16][17];
    __shared__ float s_scale[16];
    const PrepLayer L = layers[tile2layer[blockIdx.x]];
    prep_dispatch(L, blockIdx.x - L.tile_begin, red, s_scale);
}

// plain (unmasked, 9-tap) convs: their own kernel so that the 9-tap register footprint does not tax the masked prep.
// tile2layer == NULL: a single layer.
__global__ __launch_bounds__(256) void iaf_prep_plain_kernel(const PrepLayer* __restrict__ layers,
                                                            const int* __restrict__ tile2layer) {
    __shared__ float red[16][17];
    __shared__ float s_scale[16];
    const PrepLayer L = layers[tile2layer ? tile2layer[blockIdx.x] : 0];
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

__global__ __launch_bounds__(256) void iaf_prep_kernel(PrepArgs a) {
    __shared__ float red[16][17];
    __shared__ float s_scale[16];
    int li = 0;
    for (int i = 1; i < a.nlayers; ++i)
        if ((int)blockIdx.x >= a.L[i].tile_begin) li = i;
    const PrepLayer& L = a.L[li];
    prep_dispatch(L, blockIdx.x - L.tile_begin, red, s_scale);
}

// ---------------------------------------------------------------------------------------------
// KL / free-bits reduction, tf_train.py:77-85.  kl_elem [B,Z,H,W] -> kl_cost[B], kl_obj[B].
// One workgroup: deterministic tree order.  S[b,c] = sum_{H,W} kl.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iaf_kl_rowsum_kernel(const float* kl, float* S, int rows, int HW) {
    // one wave per (b,c) row
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float a = 0.f;
    for (int i = lane; i < HW; i += 64) a += kl[(size_t)row * HW + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
    if (lane == 0) S[row] = a;
}

__global__ __launch_bounds__(256) void iaf_kl_finish_kernel(const float* S, float* kl_obj, float* kl_cost, int B, int Z,
                                                           float kl_min) {
    // S is tiny ([B, Z]); stage it through LDS in one coalesced sweep instead of B*Z dependent global loads
    __shared__ float sh[8192];
    __shared__ float part[256];
    __shared__ float s_fb;
    const int tid = threadIdx.x, n = B * Z;
    const bool in_lds = n <= 8192;
    if (in_lds) {
        for (int i = tid; i < n; i += 256) sh[i] = S[i];
        __syncthreads();
    }
    const float* src = in_lds ? sh : S;
    if (kl_min > 0.f) {
        // kl_ave[c] = max(mean_b S[b,c], kl_min); kl_obj[b] = sum_c kl_ave[c]   (tf_train.py:79-82)
        float a = 0.f;
        for (int c = tid; c < Z; c += 256) {
            float m = 0.f;
            for (int b = 0; b < B; ++b) m += src[(size_t)b * Z + c];
            a += fmaxf(m / (float)B, kl_min);
        }
        part[tid] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) part[tid] += part[tid + o];
            __syncthreads();
        }
        if (tid == 0) s_fb = part[0];
        __syncthreads();
    }
    for (int b = tid; b < B; b += 256) {
        float a = 0.f;
        for (int c = 0; c < Z; ++c) a += src[(size_t)b * Z + c];
        kl_cost[b] = a;                                        // tf_train.py:85
        kl_obj[b] = (kl_min > 0.f) ? s_fb : a;                 // tf_train.py:82 / 84
    }
}

// ---------------------------------------------------------------------------------------------
// elementwise distributions (tf_utils/distributions.py)
// ---------------------------------------------------------------------------------------------
// max |a - b| over n elements -> *out (float bits; non-negative floats order like unsigned ints).  NaN counts as +inf.
__global__ __launch_bounds__(256) void iaf_maxdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                         unsigned* out) {
    float m = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = fabsf(a[i] - b[i]);
        m = (d > m || d != d) ? (d != d ? __builtin_inff() : d) : m;
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

__global__ void iaf_gauss_sample_kernel(const float* mean, const float* logvar, const float* noise, float* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = mean[i] + expf(0.5f * logvar[i]) * noise[i];
}
__global__ void iaf_gauss_logps_kernel(const float* mean, const float* logvar, const float* sample, float* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = sample[i] - mean[i];
        out[i] = -0.5f * (1.8378770664093453f + logvar[i] + d * d / expf(logvar[i]));
    }
}

// Adamax (tf_utils/adamax.py:40-56; NB the reference's slot naming: "v" = first moment, "m" = infinity norm) fused
// with the 1/N gradient averaging of average_grads (tf_utils/common.py:86) and the EMA of the parameters
// (tf_train.py:157-158, decay 0.999).  One pass over flat fp32 buffers: 5 reads + 4 writes per element, HBM-bound.
__global__ __launch_bounds__(256) void iaf_adamax_ema_kernel(float* __restrict__ var, const float* __restrict__ grad,
                                                            float* __restrict__ slot_m, float* __restrict__ slot_v,
                                                            float* __restrict__ ema, size_t n4, size_t n, float lr, float beta1,
                                                            float beta2, float eps, float ema_decay, float grad_scale) {
    auto upd = [&](float& w, float g, float& m, float& v, float& e) {
        g *= grad_scale;
        v = beta1 * v + (1.f - beta1) * g;                    // adamax.py:50
        m = fmaxf(beta2 * m + eps, fabsf(g));                 // adamax.py:52
        w -= lr * (v / m);                                    // adamax.py:53-55
        e -= (1.f - ema_decay) * (e - w);                     // ExponentialMovingAverage.apply
    };
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 w = ((f32x4*)var)[i], g = ((const f32x4*)grad)[i], m = ((f32x4*)slot_m)[i], v = ((f32x4*)slot_v)[i];
        f32x4 e = ema ? ((f32x4*)ema)[i] : w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float wr_ = w[r], mr = m[r], vr = v[r], er = e[r];
            upd(wr_, g[r], mr, vr, er);
            w[r] = wr_; m[r] = mr; v[r] = vr; e[r] = er;
        }
        ((f32x4*)var)[i] = w; ((f32x4*)slot_m)[i] = m; ((f32x4*)slot_v)[i] = v;
        if (ema) ((f32x4*)ema)[i] = e;
    }
    for (size_t i = 4 * n4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {   // tail
        float e = ema ? ema[i] : var[i];
        upd(var[i], grad[i], slot_m[i], slot_v[i], e);
        if (ema) ema[i] = e;
    }
}

// streaming logsumexp over k importance weights per image: one wave per image
__global__ __launch_bounds__(256) void iaf_lb_update_kernel(float* run_max, float* run_sum, const float* log_pxz,
                                                           const float* sum_kl, int n, int kc) {
    const int img = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (img >= n) return;
    const float* a = log_pxz + (size_t)img * kc;
    const float* b = sum_kl + (size_t)img * kc;
    float m = -INFINITY;
    for (int i = lane; i < kc; i += 64) m = fmaxf(m, a[i] - b[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float old_m = run_max[img];
    const float new_m = fmaxf(old_m, m);
    float s = 0.f;
    for (int i = lane; i < kc; i += 64) s += expf((a[i] - b[i]) - new_m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const float old_s = run_sum[img];
        run_sum[img] = (old_m == -INFINITY ? 0.f : old_s * expf(old_m - new_m)) + s;
        run_max[img] = new_m;
    }
}
__global__ void iaf_lb_init_kernel(float* run_max, float* run_sum, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { run_max[i] = -INFINITY; run_sum[i] = 0.f; }
}
__global__ void iaf_lb_finalize_kernel(const float* run_max, const float* run_sum, float* out, int n, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -(-logf((float)k) + run_max[i] + logf(run_sum[i]));   // distributions.py:62
}
__global__ void iaf_lb_k1_kernel(const float* log_pxz, const float* sum_kl, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sum_kl[i] - log_pxz[i];                                 // distributions.py:57
}

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
struct WgradP {
    const float* x;      // [P][cin]
    const float* dy;     // [P][cout]
    float* part;         // [nrange][ntaps][cin][cout]
    int B, H, W, HW, P, cin, cout, nrange, px_per_range;
    int ntaps;           // 5 (masked) or 9 (plain)
    int tap_dh[MAXTAPS], tap_dw[MAXTAPS];
    const unsigned short* tapmask;   // [P]: bit (dh+1)*3+(dw+1) set when pixel p's neighbour (dh,dw) lies inside its image
};

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

template <int NCOT>
__global__ __launch_bounds__(256, 2) void iaf_wgrad_kernel(WgradP p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tap = blockIdx.x % p.ntaps;
    const int cip = blockIdx.x / p.ntaps;          // pair of ci tiles
    const int range = blockIdx.y;
    const int cob = blockIdx.z * NCOT * 16;        // this workgroup's first packed output channel
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
    // wave w takes K steps w, w+4, ... of the range (4 pixels each).  The operands of step k+1 are requested before the
    // MFMAs of step k issue: with one wave per SIMD nothing else hides the ~1 us global-load latency.
    // Software pipeline, one K step deep: the loads of step k+1 are issued, THEN the MFMAs of step k run, THEN the loaded
    // values are masked (select) into the operand registers.  Loads are unconditional (addresses clamped into the
    // tensors, zeroing by select): a branch around a load, or a select right behind it, makes the wave wait for the
    // load before the MFMAs are issued and the prefetch is for nothing (sched_barrier pins the three phases).
    const int start = r0 + 4 * wave;
    const int nstep = (r1 - start + 15) / 16;              // steps of this wave (<= 0: nothing to do)
    float av[2], bv[NCOT];                                 // operands of the current step
    float ra[2], rb[NCOT];                                 // raw loads of the next step
    bool nxv = false, npv = false;
    auto issue = [&](int pb) {
        const int pk = pb + ks;                            // this lane's pixel for both operands
        npv = pk < r1;
        const int pkc = npv ? pk : r1 - 1;
        nxv = npv && ((p.tapmask[pkc] >> tapbit) & 1);
        const long long xp = nxv ? (long long)pkc + shift : (long long)pkc;
        const float* xr = p.x + xp * p.cin + ci0 + i15;
        const float* dr = p.dy + (size_t)pkc * p.cout + cob + i15;
        ra[0] = xr[0];
        ra[1] = xr[nci == 2 ? 16 : 0];
#pragma unroll
        for (int t = 0; t < NCOT; ++t) rb[t] = dr[t * 16];
    };
    auto take = [&]() {
        av[0] = nxv ? ra[0] : 0.f;
        av[1] = (nxv && nci == 2) ? ra[1] : 0.f;
#pragma unroll
        for (int t = 0; t < NCOT; ++t) bv[t] = npv ? rb[t] : 0.f;
    };
    issue(start);
    take();
    for (int k = 0; k < nstep; ++k) {
        issue(start + 16 * (k + 1));                       // beyond the range: clamped address, masked to zero
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int t = 0; t < NCOT; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[t], acc[a][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        take();
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

// (3a) sum the wgrad partials over the pixel ranges (fully parallel, 16-byte accesses) and, in extra workgroups of the
//      same launch, column-sum dY over pixel slabs for the bias gradient:
//        dW[i] = sum_k part[k][i]            blocks [0, nblk_w)
//        dbp[r][co] = sum_{p in slab r} dY[p][co]   blocks [nblk_w, nblk_w + nslab)
__global__ __launch_bounds__(256) void iaf_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int nrange,
                                                              size_t n4, int nblk_w, const float* __restrict__ dy,
                                                              float* __restrict__ dbp, int P, int cout, int px_per_slab) {
    if ((int)blockIdx.x < nblk_w) {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)nblk_w * blockDim.x) {
            f32x4 a = ((const f32x4*)part)[i];
            for (int k = 1; k < nrange; ++k) a += ((const f32x4*)part)[(size_t)k * n4 + i];
            ((f32x4*)dW)[i] = a;
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

struct WnBwdArgs {
    WnBwdLayer L[MAX_GEMM_LAYERS + 1];   // one entry per conv (the output pair counts twice)
    int tile_begin[MAX_GEMM_LAYERS + 2];
    int n;
};

// every conv of a stack in one launch: workgroup -> (conv, 16-channel output tile)
__global__ __launch_bounds__(256) void iaf_wn_bwd_kernel(WnBwdArgs a) {
    __shared__ float red[3][16][17];
    __shared__ float s_n[16], s_dot[16];
    int li = 0;
    for (int i = 1; i < a.n; ++i)
        if ((int)blockIdx.x >= a.tile_begin[i]) li = i;
    const WnBwdLayer& L = a.L[li];
    const int tile = blockIdx.x - a.tile_begin[li];
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

// plain convs: NCHW -> pixel-major staging of the backward operands.  dst[P][C] = scale * act(concat_k src_k)[b, c, p]
// (act = ELU when elu is set).  Up to MAXSPLIT sources; boundaries are multiples of 4.  A thread owns 4 channels of a pixel:
// reads are coalesced along pixels (64 lanes = 64 consecutive pixels), the write is one 16-byte store.
struct PackP {
    const float* src[MAXSPLIT]; int end[MAXSPLIT]; int nsrc;
    float* dst; int C, HW, P; float scale; int elu;
};
__global__ __launch_bounds__(256) void iaf_pack_pixmajor_kernel(PackP p) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int c = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4;
    if (px >= p.P || c >= p.C) return;
    int k = 0;
    while (k + 1 < p.nsrc && c >= p.end[k]) ++k;
    const int c0 = k ? p.end[k - 1] : 0, ck = p.end[k] - c0;
    const int b = px / p.HW, pp = px - b * p.HW;
    const float* s = p.src[k] + ((size_t)b * ck + (c - c0)) * p.HW + pp;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = s[(size_t)r * p.HW];
        if (p.elu) t = elu_f(t);
        v[r] = t * p.scale;
    }
    *(f32x4*)(p.dst + (size_t)px * p.C + c) = v;
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

// (4) posterior block backward, elementwise parts (tf_train.py:56-85 differentiated):
//   pre : dkl[b,c,:,:] = G[b,c]  (free bits: G = (sum_b' dkl_obj[b']) / B where mean_b S[b,c] > kl_min, else 0;
//                                  kl_min <= 0: G = dkl_obj[b]);
//         z0 = mean + e^{lq} eps;  d z_tot = dz + dkl (z - pm) e^{-2 pl};  d pm = -dkl (z - pm) e^{-2 pl};
//         d pl = dkl (1 - (z - pm)^2 e^{-2 pl});   core inputs: dz_new := dz_tot, dlogsd := dkl  (logqs += s)
//   post: d mean = dz0;  d lq = dz0 (z0 - mean) - dkl     (d logq0/d mean = 0 and d logq0/d lq = -1 after the
//         reparametrisation paths cancel analytically)
__global__ __launch_bounds__(256) void iaf_post_bwd_gate_kernel(const float* __restrict__ S, const float* __restrict__ dkl_obj,
                                                               float* __restrict__ Gc, int B, int Z, float kl_min) {
    // one block; Gc[c] = gate(c) * sum_b dkl_obj[b] / B
    __shared__ float s_sum;
    if (threadIdx.x == 0) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dkl_obj[b];
        s_sum = a / (float)B;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Z; c += blockDim.x) {
        float m = 0.f;
        for (int b = 0; b < B; ++b) m += S[(size_t)b * Z + c];
        Gc[c] = (m / (float)B > kl_min) ? s_sum : 0.f;
    }
}

__global__ __launch_bounds__(256) void iaf_post_bwd_pre_kernel(const float* qm, const float* ql, const float* rm, const float* rl,
                                                              const float* pm, const float* pl, const float* eps, const float* z,
                                                              const float* dz, const float* Gc, const float* dkl_obj, float kl_min,
                                                              float* z0, float* dzt, float* dkl, float* dpm, float* dpl, int Z,
                                                              int HW, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = i / HW;
        const int c = (int)(bc % Z);
        const size_t b = bc / Z;
        const float g = (kl_min > 0.f) ? Gc[c] : dkl_obj[b];
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

// ---------------------------------------------------------------------------------------------
// data-dependent init, tf_utils/layers.py:45-51: per-channel moments of x_init = conv(x, l2norm(mask*V)) over (N,H,W),
//   scale = init_scale / sqrt(var + 1e-10);  g = log(scale)/3;  b = -mean*scale;  y = scale*(x_init - mean)
// One workgroup per channel; two passes (mean, then centred variance) in a fixed tree order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void iaf_datainit_kernel(const float* x, const float* __restrict__ add, float* y,
                                                          float* __restrict__ g, float* __restrict__ b, int B, int C, int HW,
                                                          float init_scale) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int n = B * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const int bb = i / HW; s += x[((size_t)bb * C + c) * HW + (i - bb * HW)]; }
    const float mean = block_sum_256(s, red) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int bb = i / HW;
        const float d = x[((size_t)bb * C + c) * HW + (i - bb * HW)] - mean;
        q += d * d;
    }
    const float var = block_sum_256(q, red) / (float)n;          // tf.nn.moments: biased
    const float scale = init_scale / sqrtf(var + 1e-10f);
    if (threadIdx.x == 0) { g[c] = logf(scale) / 3.0f; b[c] = -mean * scale; }
    if (y)
        for (int i = threadIdx.x; i < n; i += 256) {
            const int bb = i / HW;
            const size_t o = ((size_t)bb * C + c) * HW + (i - bb * HW);
            y[o] = scale * (x[o] - mean) + (add ? add[o] : 0.f);
        }
}

// discretized logistic log-likelihood, tf_utils/distributions.py:28-32 (call site tf_train.py:210): one workgroup per
// batch row, out[b] = sum log(sigmoid(s + binsize/scale) - sigmoid(s) + 1e-7), s = (floor(x/binsize)*binsize - mean)/scale
__global__ __launch_bounds__(256) void iaf_disc_logistic_kernel(const float* __restrict__ mean, const float* __restrict__ logscale,
                                                               int scalar_scale, const float* __restrict__ sample,
                                                               float* __restrict__ out, size_t n, float binsize) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * n;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 256) {
        const float scale = expf(scalar_scale ? logscale[0] : logscale[base + i]);
        const float s = (floorf(sample[base + i] / binsize) * binsize - mean[base + i]) / scale;
        // sigmoid(s+d) - sigmoid(s), evaluated on the side where both terms are small (sigmoid(t) = 1 - sigmoid(-t)):
        // the literal fp32 form cancels to ~1e-7 absolute in the upper tail, the size of the +1e-7 floor itself
        const float d = binsize / scale;
        float diff;
        if (s > 0.f) {
            const float e0 = expf(-s), e1 = expf(-(s + d));
            diff = e0 / (1.0f + e0) - e1 / (1.0f + e1);
        } else {
            diff = 1.0f / (1.0f + expf(-(s + d))) - 1.0f / (1.0f + expf(-s));
        }
        acc += logf(diff + 1e-7f);
    }
    const float tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------------------------
// GENERIC FALLBACK: direct (VALU) masked conv for channel counts the MFMA path does not cover (not multiples of 16,
// or > 256).  Same arithmetic, same fused epilogues, NCHW everywhere, one thread per output element.  Orders of
// magnitude slower than the MFMA path -- it exists so that every shape the reference accepts (layers.py:116 only asks
// that n_h and n_z divide each other) runs through the same C ABI; tests hold it to the reference's tiny golden cases.
// ---------------------------------------------------------------------------------------------
struct GenPrepLayer {
    const float* V[2]; const float* g[2]; const float* b[2];
    float* w;        // effective weights [NTAPS][cin][cout_total]
    float* bias;     // [cout_total]
    int cin, cout_each, npair, zerodiag, ch_begin;
    int ntaps;       // 5 = MADE-masked (default when 0), 9 = all nine filter positions stored
    int mask9;       // ntaps == 9 only: 0 = unmasked conv2d, 1 = ar_conv2d mask (dead taps stored as zeros)
};
struct GenPrepArgs { GenPrepLayer L[MAX_GEMM_LAYERS]; int nlayers; };

__global__ __launch_bounds__(256) void iaf_generic_prep_kernel(GenPrepArgs a) {
    __shared__ float red[256];
    int li = 0;
    for (int i = 1; i < a.nlayers; ++i)
        if ((int)blockIdx.x >= a.L[i].ch_begin) li = i;
    const GenPrepLayer& L = a.L[li];
    const int oc = blockIdx.x - L.ch_begin;              // channel inside this GEMM layer: [mean channels | logsd channels]
    const int which = oc / L.cout_each, o = oc - which * L.cout_each;
    const float* V = L.V[which];
    const int n_in = L.cin, n_out = L.cout_each, ctot = L.cout_each * L.npair;
    const bool full = (L.ntaps == MAXTAPS);
    const int ntaps = full ? MAXTAPS : NTAPS;
    float ss = 0.f;
    for (int e = threadIdx.x; e < ntaps * n_in; e += 256) {
        const int t = e / n_in, ci = e - t * n_in;
        const int kh = full ? t / 3 : ((t == 0 || t == 1) ? 1 : 2), kw = full ? t % 3 : ((t == 0) ? 1 : (t == 1 ? 2 : t - 2));
        const bool live = full ? (!L.mask9 || kh == 2 || (kh == 1 && kw == 2) ||
                                  (kh == 1 && kw == 1 && made_live(ci, o, n_in, n_out, L.zerodiag)))   // layers.py:134-141
                               : ((t != 0) || made_live(ci, o, n_in, n_out, L.zerodiag));
        const float v = live ? V[((size_t)(kh * 3 + kw) * n_in + ci) * n_out + o] : 0.f;
        ss += v * v;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    const float scale = expf(L.g[which][o]) / sqrtf(fmaxf(red[0], 1e-12f));      // layers.py:60
    for (int e = threadIdx.x; e < ntaps * n_in; e += 256) {
        const int t = e / n_in, ci = e - t * n_in;
        const int kh = full ? t / 3 : ((t == 0 || t == 1) ? 1 : 2), kw = full ? t % 3 : ((t == 0) ? 1 : (t == 1 ? 2 : t - 2));
        const bool live = full ? (!L.mask9 || kh == 2 || (kh == 1 && kw == 2) ||
                                  (kh == 1 && kw == 1 && made_live(ci, o, n_in, n_out, L.zerodiag)))   // layers.py:134-141
                               : ((t != 0) || made_live(ci, o, n_in, n_out, L.zerodiag));
        L.w[((size_t)t * n_in + ci) * ctot + oc] = live ? V[((size_t)(kh * 3 + kw) * n_in + ci) * n_out + o] * scale : 0.f;
    }
    if (threadIdx.x == 0) L.bias[oc] = L.b[which][o];
}

struct GenConvP {
    const float* x;       // NCHW input (NULL in posterior mode for the first layer)
    const float* w; const float* bias;
    const float* ctx; const float* ctx2;
    float* y;             // hidden output, NCHW
    const float* zin; float* out0; float* out1; float* kl_elem;
    const float* qm; const float* ql; const float* rm; const float* rl; const float* pm; const float* pl; const float* eps;
    int B, H, W, cin, cout, nz, mode, is_out, posterior_in;
};

__device__ __forceinline__ float gen_x(const GenConvP& p, int b, int ci, int hh, int ww) {
    const size_t i = (((size_t)b * p.cin + ci) * p.H + hh) * p.W + ww;
    if (!p.posterior_in) return p.x[i];
    return (p.qm[i] + p.rm[i]) + __expf(0.5f * (2.f * (p.ql[i] + p.rl[i]))) * p.eps[i];      // tf_train.py:57,63
}

__global__ __launch_bounds__(256) void iaf_generic_conv_kernel(GenConvP p) {
    const int tap_dh[NTAPS] = {0, 0, 1, 1, 1}, tap_dw[NTAPS] = {0, 1, -1, 0, 1};
    const int nout = p.is_out ? p.nz : p.cout;
    const size_t total = (size_t)p.B * nout * p.H * p.W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ww = (int)(i % p.W);
        const int hh = (int)((i / p.W) % p.H);
        const int co = (int)((i / ((size_t)p.W * p.H)) % nout);
        const int b = (int)(i / ((size_t)p.W * p.H * nout));
        float a0 = 0.f, a1 = 0.f;
        for (int t = 0; t < NTAPS; ++t) {
            const int h2 = hh + tap_dh[t], w2 = ww + tap_dw[t];
            if (h2 < 0 || h2 >= p.H || w2 < 0 || w2 >= p.W) continue;
            const float* wt = p.w + (size_t)t * p.cin * p.cout;
            for (int ci = 0; ci < p.cin; ++ci) {
                const float xv = gen_x(p, b, ci, h2, w2);
                a0 = fmaf(xv, wt[(size_t)ci * p.cout + co], a0);
                if (p.is_out) a1 = fmaf(xv, wt[(size_t)ci * p.cout + p.nz + co], a1);
            }
        }
        if (!p.is_out) {
            float v = a0 + p.bias[co];
            if (p.ctx) { v += p.ctx[i]; if (p.ctx2) v += p.ctx2[i]; }
            p.y[i] = elu_f(v);
            continue;
        }
        const float m_raw = a0 + p.bias[co], s_raw = a1 + p.bias[p.nz + co];
        if (p.mode == MODE_RAW) { p.out0[i] = m_raw; p.out1[i] = s_raw; continue; }
        const float m = m_raw * 0.1f, sgm = s_raw * 0.1f;
        if (p.mode == MODE_IAF) { p.out0[i] = (p.zin[i] - m) / __expf(sgm); p.out1[i] = sgm; continue; }
        if (p.mode == MODE_INVERSE) { p.out0[i] = p.zin[i] * __expf(sgm) + m; p.out1[i] = sgm; continue; }
        const float mean = p.qm[i] + p.rm[i], logvar = 2.f * (p.ql[i] + p.rl[i]);
        const float z0 = mean + __expf(0.5f * logvar) * p.eps[i];
        const float d0 = z0 - mean;
        float logqs = -0.5f * (1.8378770664093453f + logvar + d0 * d0 / __expf(logvar)) + sgm;
        const float z = (z0 - m) / __expf(sgm);
        const float plv = 2.f * p.pl[i], d1 = z - p.pm[i];
        const float logps = -0.5f * (1.8378770664093453f + plv + d1 * d1 / __expf(plv));
        p.out0[i] = z;
        if (p.out1) p.out1[i] = sgm;
        p.kl_elem[i] = logqs - logps;
    }
}

// plain 3x3 SAME conv, generic channel counts: y = conv(elu?(concat(x, x2)), w) + b  [-> res + 0.1*y], output channels
// scattered to the split tensors.  One thread per output element.
struct GenPlainP {
    const float* x; const float* x2; const float* w; const float* bias; const float* res;
    int B, H, W, cin, cout, c_split, in_elu, nsplit;
    int split_end[MAXSPLIT]; float* split_ptr[MAXSPLIT];
};

__global__ __launch_bounds__(256) void iaf_generic_conv3x3_kernel(GenPlainP p) {
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * p.cout * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ww = (int)(i % p.W);
        const int hh = (int)((i / p.W) % p.H);
        const int co = (int)((i / HW) % p.cout);
        const int b = (int)(i / (HW * p.cout));
        float acc = 0.f;
        for (int t = 0; t < MAXTAPS; ++t) {
            const int h2 = hh + t / 3 - 1, w2 = ww + t % 3 - 1;
            if (h2 < 0 || h2 >= p.H || w2 < 0 || w2 >= p.W) continue;
            const float* wt = p.w + (size_t)t * p.cin * p.cout;
            for (int ci = 0; ci < p.cin; ++ci) {
                float xv;
                if (p.x2 && ci >= p.c_split) xv = p.x2[(((size_t)b * (p.cin - p.c_split) + (ci - p.c_split)) * p.H + h2) * p.W + w2];
                else xv = p.x[(((size_t)b * (p.x2 ? p.c_split : p.cin) + ci) * p.H + h2) * p.W + w2];
                if (p.in_elu) xv = elu_f(xv);
                acc = fmaf(xv, wt[(size_t)ci * p.cout + co], acc);
            }
        }
        const float v = acc + p.bias[co];
        int k = 0;
        while (k + 1 < p.nsplit && co >= p.split_end[k]) ++k;
        const int c0 = k ? p.split_end[k - 1] : 0;
        const size_t o = (((size_t)b * (p.split_end[k] - c0) + (co - c0)) * p.H + hh) * p.W + ww;
        p.split_ptr[k][o] = p.res ? p.res[i] + 0.1f * v : v;
    }
}

// ---------------------------------------------------------------------------------------------
// host side: stack object
// ---------------------------------------------------------------------------------------------
static dim3 ew_grid(size_t n);

struct GemmLayer {
    int cin, cout;        // cout = GEMM N (output pair: 2*n_z)
    int nchunk, ncot;
    int zerodiag, npair;
    float* wp = nullptr;
    float* bias = nullptr;
    float* border = nullptr;   // Theano variant only
    float* wpt = nullptr;      // transposed pack for dgrad (allocated by iaf_stack_set_training)
    int* lim = nullptr;
    // launch shape: fixed by iaf_stack_set_tuning (user_tuned) or chosen per problem size by auto_shape()
    int nt, pxt, wco, ks;
    bool user_tuned = false;
    bool full3x3 = false;      // plain 9-tap conv (iaf_conv3x3): halo on both sides of the pixel tile
    double live_macs_per_px, dense_macs_per_px;
};

struct iaf_stack {
    int n_z, n_h, depth_ar, variant;
    int nlayers;          // depth_ar + 1
    GemmLayer L[MAX_GEMM_LAYERS];
    GemmLayer T[MAX_GEMM_LAYERS];   // transposed problems (dX = W^T dY) of the same layers; valid when training
    bool training = false;
    bool generic = false;     // channel counts outside the MFMA path: direct-conv fallback kernels
    bool prepared;
    size_t weight_bytes;  // raw V/g/b bytes of the stack (for the algorithmic byte count)
    // optional per-launch event timing of one layer
    unsigned long long* dbg = nullptr; int dbg_layer = -1;
    int prof_layer = -1, prof_cap = 0, prof_n = 0;
    hipEvent_t* prof_start = nullptr;
    hipEvent_t* prof_stop = nullptr;
};

#define HIP_TRY(expr)                               \
    do {                                            \
        hipError_t _e = (expr);                     \
        if (_e != hipSuccess) return (int)_e;       \
    } while (0)

// one translation unit per launch shape (iaf_conv_inst.hip, compiled with -DIAF_PXT/-DIAF_WCO/-DIAF_KS)
#define IAF_DECL_SHAPE(P, W, K) extern "C" conv_fn_t iaf_pick_conv_##P##_##W##_##K(int nt, int inmode, int epi);
IAF_DECL_SHAPE(4, 1, 1)
IAF_DECL_SHAPE(4, 1, 2)
IAF_DECL_SHAPE(2, 2, 1)
IAF_DECL_SHAPE(2, 2, 2)
IAF_DECL_SHAPE(2, 1, 2)
IAF_DECL_SHAPE(2, 1, 4)
IAF_DECL_SHAPE(1, 1, 4)
IAF_DECL_SHAPE(1, 2, 2)

// the launch shapes that are compiled: (pxt, wco, ks)
static conv_fn_t pick_kernel(int nt, int pxt, int wco, int ks, int inmode, int epi) {
    if (nt < 1 || nt > 5) return nullptr;
    if (pxt == 4 && wco == 1 && ks == 1) return iaf_pick_conv_4_1_1(nt, inmode, epi);
    if (pxt == 4 && wco == 1 && ks == 2) return iaf_pick_conv_4_1_2(nt, inmode, epi);
    if (pxt == 2 && wco == 2 && ks == 1) return iaf_pick_conv_2_2_1(nt, inmode, epi);
    if (pxt == 2 && wco == 2 && ks == 2) return iaf_pick_conv_2_2_2(nt, inmode, epi);
    if (pxt == 2 && wco == 1 && ks == 2) return iaf_pick_conv_2_1_2(nt, inmode, epi);
    if (pxt == 2 && wco == 1 && ks == 4) return iaf_pick_conv_2_1_4(nt, inmode, epi);
    if (pxt == 1 && wco == 1 && ks == 4) return iaf_pick_conv_1_1_4(nt, inmode, epi);
    if (pxt == 1 && wco == 2 && ks == 2) return iaf_pick_conv_1_2_2(nt, inmode, epi);
    return nullptr;
}

static size_t conv_lds_bytes(const GemmLayer& L, int W) {
    const int tm = 16 * L.pxt, nslot = tm + (L.full3x3 ? 2 : 1) * (W + 1), cp = L.cin + 8;
    size_t fl = (size_t)(nslot + 1) * cp;
    size_t wbuf = 0, red = 0;
#if defined(IAF_SHARED_W) && IAF_SHARED_W
    if (L.pxt > 1) wbuf = (size_t)L.wco * L.ks * 2 * NTAPS * L.nt * 256;   // shared weights: 2 chunk buffers per wave group
#endif
    if (L.ks > 1) red = (size_t)L.ks * L.pxt * L.wco * L.nt * 256;          // split-K exchange (aliases the weight buffers)
    fl += wbuf > red ? wbuf : red;
    return fl * sizeof(float);
}

static void default_tuning(GemmLayer& L, bool is_out) {
    if (is_out) {
        L.nt = 2; L.pxt = 2; L.wco = (L.ncot >= 4 && (L.ncot / 2) % 2 == 0) ? 2 : 1; L.ks = (L.wco == 2) ? 1 : 2;
        return;
    }
    int nt = 1;
    for (int c = 5; c >= 1; --c)
        if (L.ncot % c == 0) { nt = c; break; }
    L.nt = nt; L.pxt = 4; L.wco = 1; L.ks = 1;
}

extern "C" int iaf_abi_version(void) { return IAF_ABI_VERSION; }

extern "C" const char* iaf_error_string(int code) {
    switch (code) {
        case IAF_OK: return "ok";
        case IAF_ERR_NULL: return "null pointer argument";
        case IAF_ERR_SHAPE: return "bad shape";
        case IAF_ERR_NOT_MULTIPLE: return "n_h must be a multiple of n_z or vice versa";
        case IAF_ERR_NOT_PREPARED: return "iaf_stack_prepare has not been called";
        case IAF_ERR_WORKSPACE: return "workspace too small or misaligned";
        case IAF_ERR_UNSUPPORTED: return "not covered by the gfx950 kernels (channels must be multiples of 16 and <= 256; launch shape must fit 160 KiB of LDS)";
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

extern "C" int iaf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

static void count_macs(GemmLayer& L, int n_in, int n_out_each, int zerodiag, int npair) {
    double centre = 0;
    for (int i = 0; i < n_in; ++i)
        for (int o = 0; o < n_out_each; ++o) centre += made_live(i, o, n_in, n_out_each, zerodiag) ? 1 : 0;
    L.live_macs_per_px = npair * (4.0 * n_in * n_out_each + centre);
    L.dense_macs_per_px = npair * 9.0 * n_in * n_out_each;
}

extern "C" int iaf_stack_create(iaf_stack_t** out, int n_z, int n_h, int depth_ar, int variant) {
    if (!out) return IAF_ERR_NULL;
    *out = nullptr;
    if (n_z <= 0 || n_h <= 0 || depth_ar < 0 || depth_ar > MAX_GEMM_LAYERS - 1) return IAF_ERR_SHAPE;
    if (variant != IAF_VARIANT_TF && variant != IAF_VARIANT_THEANO) return IAF_ERR_UNSUPPORTED;
    if (depth_ar > 0 && !(n_z % n_h == 0 || n_h % n_z == 0)) return IAF_ERR_NOT_MULTIPLE;   // layers.py:116
    const bool generic = (n_z % 16 != 0 || (depth_ar > 0 && n_h % 16 != 0) || n_z > 16 * PREP_MAXI ||
                          (depth_ar > 0 && n_h > 16 * PREP_MAXI));
    if (generic && variant != IAF_VARIANT_TF) return IAF_ERR_UNSUPPORTED;
    iaf_stack* s = new (std::nothrow) iaf_stack();
    if (!s) return (int)hipErrorOutOfMemory;
    s->generic = generic;
    s->n_z = n_z; s->n_h = n_h; s->depth_ar = depth_ar; s->variant = variant;
    s->nlayers = depth_ar + 1;
    s->prepared = false;
    s->weight_bytes = 0;
    int cin = n_z;
    for (int l = 0; l < s->nlayers; ++l) {
        GemmLayer& L = s->L[l];
        const bool is_out = (l == depth_ar);
        const int each = is_out ? n_z : n_h;
        L.cin = cin;
        L.npair = is_out ? 2 : 1;
        L.cout = each * L.npair;
        L.zerodiag = is_out ? 1 : 0;          // layers.py:162 (False) / 166 (True)
        L.nchunk = (cin + 15) / 16;
        L.ncot = (L.cout + 15) / 16;
        default_tuning(L, is_out);
        count_macs(L, cin, each, L.zerodiag, L.npair);
        s->weight_bytes += (size_t)L.npair * (9 * (size_t)(cin + (variant == IAF_VARIANT_THEANO ? 1 : 0)) * each + 2 * (size_t)each) * sizeof(float);
        int rc;
        const size_t wfloats = generic ? (size_t)NTAPS * cin * L.cout : (size_t)L.nchunk * NTAPS * L.ncot * 256;
        if ((rc = (int)hipMalloc(&L.wp, wfloats * sizeof(float))) != 0 ||
            (rc = (int)hipMalloc(&L.bias, (size_t)L.cout * sizeof(float))) != 0 ||
            (variant == IAF_VARIANT_THEANO && (rc = (int)hipMalloc(&L.border, (size_t)4 * L.cout * sizeof(float))) != 0) ||
            (rc = (int)hipMalloc(&L.lim, (size_t)L.ncot * sizeof(int))) != 0) {
            iaf_stack_destroy(s);
            return rc;
        }
        // live-chunk limit of the centre tap per packed co-tile (the mask is monotone in c_in)
        int limh[1024];
        for (int gt = 0; gt < L.ncot && !generic; ++gt) {
            const int src = is_out ? (gt >> 1) : gt;
            int maxci = -1;
            for (int oo = 0; oo < 16; ++oo)
                for (int i = 0; i < cin; ++i)
                    if (made_live(i, src * 16 + oo, cin, each, L.zerodiag) && i > maxci) maxci = i;
            limh[gt] = (maxci + 16) / 16;    // ceil((maxci+1)/16); 0 if nothing live
        }
        if (!generic && (rc = (int)hipMemcpy(L.lim, limh, L.ncot * sizeof(int), hipMemcpyHostToDevice)) != 0) {
            iaf_stack_destroy(s);
            return rc;
        }
        cin = each;
    }
    *out = s;
    return IAF_OK;
}

static void prof_free(iaf_stack* s) {
    for (int i = 0; i < s->prof_cap; ++i) {
        if (s->prof_start) (void)hipEventDestroy(s->prof_start[i]);
        if (s->prof_stop) (void)hipEventDestroy(s->prof_stop[i]);
    }
    free(s->prof_start); free(s->prof_stop);
    s->prof_start = s->prof_stop = nullptr;
    s->prof_cap = s->prof_n = 0; s->prof_layer = -1;
}

extern "C" int iaf_stack_set_debug(iaf_stack_t* s, int layer, void* buf) {
    if (!s) return IAF_ERR_NULL;
    s->dbg_layer = layer; s->dbg = (unsigned long long*)buf;
    return IAF_OK;
}

extern "C" int iaf_stack_profile_enable(iaf_stack_t* s, int layer, int max_samples) {
    if (!s) return IAF_ERR_NULL;
    prof_free(s);
    if (layer < 0) return IAF_OK;
    if (layer >= s->nlayers || max_samples <= 0 || max_samples > (1 << 20)) return IAF_ERR_SHAPE;
    s->prof_start = (hipEvent_t*)calloc(max_samples, sizeof(hipEvent_t));
    s->prof_stop = (hipEvent_t*)calloc(max_samples, sizeof(hipEvent_t));
    if (!s->prof_start || !s->prof_stop) { prof_free(s); return (int)hipErrorOutOfMemory; }
    for (int i = 0; i < max_samples; ++i) {
        HIP_TRY(hipEventCreate(&s->prof_start[i]));
        HIP_TRY(hipEventCreate(&s->prof_stop[i]));
        s->prof_cap = i + 1;
    }
    s->prof_layer = layer;
    return IAF_OK;
}

extern "C" int iaf_stack_profile_read(iaf_stack_t* s, float* ms_out, int capacity, int* n_out) {
    if (!s || !ms_out || !n_out) return IAF_ERR_NULL;
    int n = s->prof_n < capacity ? s->prof_n : capacity;
    for (int i = 0; i < n; ++i) {
        HIP_TRY(hipEventSynchronize(s->prof_stop[i]));
        HIP_TRY(hipEventElapsedTime(&ms_out[i], s->prof_start[i], s->prof_stop[i]));
    }
    *n_out = n;
    s->prof_n = 0;
    return IAF_OK;
}

extern "C" int iaf_stack_destroy(iaf_stack_t* s) {
    if (!s) return IAF_ERR_NULL;
    prof_free(s);
    for (int l = 0; l < s->nlayers; ++l) {
        if (s->L[l].wp) (void)hipFree(s->L[l].wp);
        if (s->L[l].bias) (void)hipFree(s->L[l].bias);
        if (s->L[l].border) (void)hipFree(s->L[l].border);
        if (s->L[l].wpt) (void)hipFree(s->L[l].wpt);
        if (s->L[l].lim) (void)hipFree(s->L[l].lim);
    }
    delete s;
    return IAF_OK;
}

extern "C" int iaf_stack_set_tuning(iaf_stack_t* s, int layer, int nt, int pxt, int wco, int ks) {
    if (!s) return IAF_ERR_NULL;
    if (layer < 0 || layer >= s->nlayers) return IAF_ERR_SHAPE;
    if (s->generic) return IAF_ERR_UNSUPPORTED;
    GemmLayer& L = s->L[layer];
    const bool is_out = (layer == s->depth_ar);
    if (nt < 1 || L.ncot % (nt * wco) != 0) return IAF_ERR_UNSUPPORTED;
    if (is_out && (nt % 2 != 0)) return IAF_ERR_UNSUPPORTED;
    if (!pick_kernel(nt, pxt, wco, ks, IN_PIXMAJOR, is_out ? EPI_OUT : EPI_HIDDEN)) return IAF_ERR_UNSUPPORTED;
    if (L.nchunk < ks) return IAF_ERR_UNSUPPORTED;
    L.nt = nt; L.pxt = pxt; L.wco = wco; L.ks = ks;
    L.user_tuned = true;
    return IAF_OK;
}

extern "C" int iaf_stack_prepare(iaf_stack_t* s, const float* const* V, const float* const* g, const float* const* b,
                                 void* stream) {
    if (!s || !V || !g || !b) return IAF_ERR_NULL;
    const int nconv = s->depth_ar + 2;
    for (int i = 0; i < nconv; ++i)
        if (!V[i] || !g[i] || !b[i]) return IAF_ERR_NULL;
    if (s->generic) {
        GenPrepArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.nlayers = s->nlayers;
        int ch = 0;
        for (int l = 0; l < s->nlayers; ++l) {
            const GemmLayer& L = s->L[l];
            GenPrepLayer& P = ga.L[l];
            P.V[0] = V[l]; P.g[0] = g[l]; P.b[0] = b[l];
            if (L.npair == 2) { P.V[1] = V[l + 1]; P.g[1] = g[l + 1]; P.b[1] = b[l + 1]; }
            P.w = L.wp; P.bias = L.bias; P.cin = L.cin; P.cout_each = L.cout / L.npair; P.npair = L.npair;
            P.zerodiag = L.zerodiag; P.ch_begin = ch;
            ch += L.cout;
        }
        hipLaunchKernelGGL(iaf_generic_prep_kernel, dim3(ch), dim3(256), 0, (hipStream_t)stream, ga);
        HIP_TRY(hipGetLastError());
        s->prepared = true;
        return IAF_OK;
    }
    PrepArgs a;
    memset(&a, 0, sizeof(a));
    a.nlayers = s->nlayers;
    int tiles = 0;
    for (int l = 0; l < s->nlayers; ++l) {
        const GemmLayer& L = s->L[l];
        PrepLayer& P = a.L[l];
        P.V[0] = V[l]; P.g[0] = g[l]; P.b[0] = b[l];
        if (L.npair == 2) { P.V[1] = V[l + 1]; P.g[1] = g[l + 1]; P.b[1] = b[l + 1]; }
        P.wp = L.wp; P.bias = L.bias; P.border = L.border; P.variant = s->variant; P.wpt = L.wpt;
        P.cin = L.cin; P.cout_each = L.cout / L.npair; P.ncot = L.ncot; P.nchunk = L.nchunk;
        P.zerodiag = L.zerodiag; P.npair = L.npair; P.tile_begin = tiles;
        tiles += L.ncot;
    }
    hipLaunchKernelGGL(iaf_prep_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    s->prepared = true;
    return IAF_OK;
}

// ---- batched prepare: all stacks of a model in ONE launch (weights of every layer are known at step start)
struct iaf_prep_batch {
    int n;
    iaf_stack** stacks;
    int nlayers_total, ntiles;
    PrepLayer* h_layers;   // pinned; read by the async copy (also when a captured graph replays)
    PrepLayer* d_layers;
    int* d_tile2layer;
};

extern "C" int iaf_prep_batch_destroy(iaf_prep_batch_t* b) {
    if (!b) return IAF_ERR_NULL;
    if (b->h_layers) (void)hipHostFree(b->h_layers);
    if (b->d_layers) (void)hipFree(b->d_layers);
    if (b->d_tile2layer) (void)hipFree(b->d_tile2layer);
    free(b->stacks);
    delete b;
    return IAF_OK;
}

extern "C" int iaf_prep_batch_create(iaf_prep_batch_t** out, iaf_stack_t* const* stacks, int n) {
    if (!out || !stacks) return IAF_ERR_NULL;
    *out = nullptr;
    if (n <= 0) return IAF_ERR_SHAPE;
    iaf_prep_batch* b = new (std::nothrow) iaf_prep_batch();
    if (!b) return (int)hipErrorOutOfMemory;
    memset(b, 0, sizeof(*b));
    b->n = n;
    b->stacks = (iaf_stack**)calloc(n, sizeof(iaf_stack*));
    int nl = 0, nt = 0;
    for (int i = 0; i < n; ++i) {
        if (!stacks[i]) { iaf_prep_batch_destroy(b); return IAF_ERR_NULL; }
        if (stacks[i]->generic) { iaf_prep_batch_destroy(b); return IAF_ERR_UNSUPPORTED; }
        b->stacks[i] = stacks[i];
        for (int l = 0; l < stacks[i]->nlayers; ++l) { nl++; nt += stacks[i]->L[l].ncot; }
    }
    b->nlayers_total = nl; b->ntiles = nt;
    int* t2l = (int*)malloc(sizeof(int) * nt);
    int rc;
    if ((rc = (int)hipHostMalloc((void**)&b->h_layers, sizeof(PrepLayer) * nl)) != 0 ||
        (rc = (int)hipMalloc((void**)&b->d_layers, sizeof(PrepLayer) * nl)) != 0 ||
        (rc = (int)hipMalloc((void**)&b->d_tile2layer, sizeof(int) * nt)) != 0) {
        free(t2l); iaf_prep_batch_destroy(b); return rc;
    }
    memset(b->h_layers, 0, sizeof(PrepLayer) * nl);
    int li = 0, tile = 0;
    for (int i = 0; i < n; ++i)
        for (int l = 0; l < stacks[i]->nlayers; ++l, ++li) {
            const GemmLayer& L = stacks[i]->L[l];
            PrepLayer& P = b->h_layers[li];
            P.wp = L.wp; P.bias = L.bias; P.border = L.border; P.variant = stacks[i]->variant; P.wpt = L.wpt;
            P.cin = L.cin; P.cout_each = L.cout / L.npair; P.ncot = L.ncot; P.nchunk = L.nchunk;
            P.zerodiag = L.zerodiag; P.npair = L.npair; P.tile_begin = tile;
            for (int t = 0; t < L.ncot; ++t) t2l[tile++] = li;
        }
    rc = (int)hipMemcpy(b->d_tile2layer, t2l, sizeof(int) * nt, hipMemcpyHostToDevice);
    free(t2l);
    if (rc) { iaf_prep_batch_destroy(b); return rc; }
    *out = b;
    return IAF_OK;
}

extern "C" int iaf_prep_batch_run(iaf_prep_batch_t* b, const float* const* V, const float* const* g,
                                  const float* const* bias, void* stream) {
    if (!b || !V || !g || !bias) return IAF_ERR_NULL;
    int li = 0, ci = 0;   // ci: running conv index over all stacks (depth_ar + 2 convs per stack)
    for (int i = 0; i < b->n; ++i) {
        const iaf_stack* s = b->stacks[i];
        for (int l = 0; l < s->nlayers; ++l, ++li) {
            PrepLayer& P = b->h_layers[li];
            const int np = s->L[l].npair;
            for (int e = 0; e < np; ++e) {
                if (!V[ci + l + e] || !g[ci + l + e] || !bias[ci + l + e]) return IAF_ERR_NULL;
                P.V[e] = V[ci + l + e]; P.g[e] = g[ci + l + e]; P.b[e] = bias[ci + l + e];
            }
        }
        ci += s->depth_ar + 2;
    }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(b->d_layers, b->h_layers, sizeof(PrepLayer) * b->nlayers_total, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(iaf_prep_batch_kernel, dim3(b->ntiles), dim3(256), 0, st, b->d_layers, b->d_tile2layer);
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < b->n; ++i) b->stacks[i]->prepared = true;
    return IAF_OK;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t iaf_stack_workspace_bytes(const iaf_stack_t* s, int B, int H, int W) {
    if (!s || B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t P = (size_t)B * H * W;
    // two ping-pong hidden buffers [P][n_h], kl elements [B,n_z,H,W], row sums [B*n_z]
    size_t bytes = 0;
    if (s->depth_ar > 0) bytes += 2 * align_up(P * s->n_h * sizeof(float), 256);
    bytes += align_up(P * s->n_z * sizeof(float), 256);
    bytes += align_up((size_t)B * s->n_z * sizeof(float), 256);
    return bytes;
}

struct Ws {
    float* hbuf[2];
    float* kl_elem;
    float* rowsum;
};

static int carve_ws(const iaf_stack_t* s, int B, int H, int W, void* ws, size_t ws_bytes, Ws* o) {
    if (!ws) return IAF_ERR_NULL;
    if (((uintptr_t)ws & 15) != 0 || ws_bytes < iaf_stack_workspace_bytes(s, B, H, W)) return IAF_ERR_WORKSPACE;
    const size_t P = (size_t)B * H * W;
    char* p = (char*)ws;
    o->hbuf[0] = o->hbuf[1] = nullptr;
    if (s->depth_ar > 0) {
        o->hbuf[0] = (float*)p; p += align_up(P * s->n_h * sizeof(float), 256);
        o->hbuf[1] = (float*)p; p += align_up(P * s->n_h * sizeof(float), 256);
    }
    o->kl_elem = (float*)p; p += align_up(P * s->n_z * sizeof(float), 256);
    o->rowsum = (float*)p;
    return IAF_OK;
}

// Launch-shape model.  At the BASELINE batch sizes the convs are latency-bound (a few thousand pixels for
// 256 CUs), so the shape is chosen to minimise the longest per-SIMD MFMA chain:
//   cycles(wave) = (5*nchunk/ks) steps * nt tiles * 4 MFMA * 32 cycles;  waves sharing a SIMD serialise;
//   T = rounds over the 256 CUs * (cycles(WG) + fixed prologue/epilogue cost).
// Ties go to fewer rounds, then less split-K, then bigger tiles (more operand reuse).
static const int k_shapes[][3] = {{4, 1, 1}, {2, 2, 1}, {4, 1, 2}, {2, 2, 2}, {2, 1, 2}, {1, 2, 2}, {2, 1, 4}, {1, 1, 4}};

static void auto_shape(GemmLayer& L, bool is_out, long long P, int W) {
    double best = 1e30;
    int bnt = 0, bs = -1;
    for (int si = 0; si < 8; ++si) {
        const int pxt = k_shapes[si][0], wco = k_shapes[si][1], ks = k_shapes[si][2];
        if (L.nchunk < ks) continue;
        for (int nt = 5; nt >= 1; --nt) {
            if (is_out && (nt & 1)) continue;
            if (L.ncot % (nt * wco) != 0) continue;
            GemmLayer t = L;
            t.nt = nt; t.pxt = pxt; t.wco = wco; t.ks = ks;
            if (conv_lds_bytes(t, W) > 160 * 1024) continue;
            const double wgs = (double)((P + 16 * pxt - 1) / (16 * pxt)) * (L.ncot / (nt * wco));
            const double rounds = ceil(wgs / 256.0);
            const double cyc_wave = ((L.full3x3 ? 9.0 : 5.0) * L.nchunk / ks) * nt * 128.0;
            const double waves = pxt * wco * ks;
            const double cyc_wg = cyc_wave * ceil(waves / 4.0);
            const double T = rounds * (cyc_wg + 6000.0) + 400.0 * (ks - 1) + 1e-3 * si - 1e-2 * nt;
            if (T < best) { best = T; bnt = nt; bs = si; }
        }
    }
    if (bs >= 0) { L.nt = bnt; L.pxt = k_shapes[bs][0]; L.wco = k_shapes[bs][1]; L.ks = k_shapes[bs][2]; }
}

// raise the dynamic-LDS cap once per kernel (never inside a stream capture: warm up first)
static int raise_lds_cap(conv_fn_t fn, size_t lds) {
    if (lds <= 48 * 1024) return 0;
    static std::mutex mu;
    static std::unordered_set<const void*> done;
    std::lock_guard<std::mutex> lk(mu);
    if (!done.count((const void*)fn)) {
        HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        done.insert((const void*)fn);
    }
    return 0;
}

// launches the conv kernel for GEMM descriptor L (forward layer, or a transposed descriptor for dgrad)
static int launch_gemm(const iaf_stack_t* s, GemmLayer& L, int epi, bool negate_taps, int prof_id, ConvP& p, int inmode,
                       hipStream_t st) {
    if (!L.user_tuned) auto_shape(L, epi == EPI_OUT, p.P, p.W);
    const int tm = 16 * L.pxt;
    conv_fn_t fn = nullptr;
    // a grid of at most one workgroup per CU has nothing to overlap a prologue with: use the double-depth weight ring
    if (epi == EPI_HIDDEN && inmode == IN_PIXMAJOR && ((p.P + tm - 1) / tm) * (L.ncot / (L.nt * L.wco)) <= 256)
        fn = pick_kernel(L.nt, L.pxt, L.wco, L.ks, inmode, EPI_HIDDEN_DEEP);
    if (!fn) fn = pick_kernel(L.nt, L.pxt, L.wco, L.ks, inmode, epi);
    if (!fn) return IAF_ERR_UNSUPPORTED;
    p.wp = L.wp; p.bias = L.bias; p.lim = L.lim;
    {   // tap geometry of the two statements of the operator (see ConvP); the data gradient runs the mirrored taps
        static const int tf_dh[NTAPS] = {0, 0, 1, 1, 1}, tf_dw[NTAPS] = {0, 1, -1, 0, 1};
        const int sgn = ((s->variant == IAF_VARIANT_THEANO) != negate_taps) ? -1 : 1;
        for (int t = 0; t < NTAPS; ++t) { p.tap_dh[t] = sgn * tf_dh[t]; p.tap_dw[t] = sgn * tf_dw[t]; }
        p.halo_before = (sgn < 0) ? p.W + 1 : 0;
        p.border = negate_taps ? nullptr : L.border;
    }
    p.cin = L.cin; p.cout = L.cout; p.nchunk = L.nchunk; p.ncot = L.ncot;
    p.cp = L.cin + 8;
    p.nslot = tm + p.W + 1;
    p.dbg = (prof_id >= 0 && s->dbg_layer == prof_id) ? s->dbg : nullptr;
    const size_t lds = conv_lds_bytes(L, p.W);
    if (lds > 160 * 1024) return IAF_ERR_UNSUPPORTED;
    { int rc = raise_lds_cap(fn, lds); if (rc) return rc; }
    dim3 grid((p.P + tm - 1) / tm, L.ncot / (L.nt * L.wco));
    const bool prof = (prof_id >= 0 && s->prof_layer == prof_id && s->prof_n < s->prof_cap);
    iaf_stack* ms = const_cast<iaf_stack*>(s);
    if (prof) HIP_TRY(hipEventRecord(ms->prof_start[ms->prof_n], st));
    p.gx = (int)grid.x;
    hipLaunchKernelGGL(fn, grid, dim3(64 * L.pxt * L.wco * L.ks), lds, st, p);
    if (prof) { HIP_TRY(hipEventRecord(ms->prof_stop[ms->prof_n], st)); ms->prof_n++; }
    return (int)hipGetLastError();
}

static int launch_conv(const iaf_stack_t* s, int layer, ConvP& p, int inmode, hipStream_t st) {
    return launch_gemm(s, const_cast<iaf_stack*>(s)->L[layer], layer == s->depth_ar ? EPI_OUT : EPI_HIDDEN, false, layer, p,
                       inmode, st);
}

static int check_dims(const iaf_stack_t* s, int B, int H, int W) {
    if (!s) return IAF_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return IAF_ERR_SHAPE;
    if ((long long)B * H * W > (1LL << 30) / 64) return IAF_ERR_SHAPE;
    if (!s->prepared) return IAF_ERR_NOT_PREPARED;
    return IAF_OK;
}

// the depth_ar hidden convs + the output pair as launch descriptors.  mode (in base) selects the final epilogue.
struct Launch { int layer; ConvP p; int inmode; };
static int build_stack(iaf_stack_t* s, ConvP base, int first_inmode, const float* ctx, const float* ctx2, const Ws& ws,
                       Launch* out) {
    const float* cur = base.x;
    int inmode = first_inmode, n = 0;
    for (int l = 0; l < s->depth_ar; ++l) {
        ConvP p = base;
        p.x = cur;
        p.ctx = (l == 0) ? ctx : nullptr;       // context only after the first conv (layers.py:163)
        p.ctx2 = (l == 0) ? ctx2 : nullptr;
        p.y = ws.hbuf[l & 1];
        out[n++] = Launch{l, p, inmode};
        cur = p.y;
        inmode = IN_PIXMAJOR;
    }
    ConvP p = base;
    p.x = cur;
    out[n++] = Launch{s->depth_ar, p, inmode};
    return n;
}

static int run_stack_generic(iaf_stack_t* s, const ConvP& base, int first_inmode, const float* ctx, const float* ctx2,
                             const Ws& ws, hipStream_t st) {
    const float* cur = base.x;
    for (int l = 0; l <= s->depth_ar; ++l) {
        const GemmLayer& L = s->L[l];
        GenConvP p;
        memset(&p, 0, sizeof(p));
        p.B = base.B; p.H = base.H; p.W = base.W; p.cin = L.cin; p.cout = L.cout; p.nz = s->n_z;
        p.w = L.wp; p.bias = L.bias;
        p.x = cur; p.posterior_in = (l == 0 && first_inmode == IN_POSTERIOR) ? 1 : 0;
        p.qm = base.qm; p.ql = base.ql; p.rm = base.rm; p.rl = base.rl; p.pm = base.pm; p.pl = base.pl; p.eps = base.eps;
        p.is_out = (l == s->depth_ar) ? 1 : 0;
        if (p.is_out) {
            p.mode = base.mode; p.zin = base.zin; p.out0 = base.out0; p.out1 = base.out1; p.kl_elem = base.kl_elem;
        } else {
            p.ctx = (l == 0) ? ctx : nullptr; p.ctx2 = (l == 0) ? ctx2 : nullptr;
            p.y = ws.hbuf[l & 1];
        }
        const size_t total = (size_t)p.B * (p.is_out ? p.nz : p.cout) * p.H * p.W;
        hipLaunchKernelGGL(iaf_generic_conv_kernel, ew_grid(total), dim3(256), 0, st, p);
        cur = p.y;
    }
    return (int)hipGetLastError();
}

static int run_stack(iaf_stack_t* s, ConvP base, int first_inmode, const float* ctx, const float* ctx2, const Ws& ws,
                     hipStream_t st) {
    if (s->generic) return run_stack_generic(s, base, first_inmode, ctx, ctx2, ws, st);
    Launch ls[MAX_GEMM_LAYERS];
    const int n = build_stack(s, base, first_inmode, ctx, ctx2, ws, ls);
    for (int i = 0; i < n; ++i) {
        int rc = launch_conv(s, ls[i].layer, ls[i].p, ls[i].inmode, st);
        if (rc) return rc;
    }
    return IAF_OK;
}

extern "C" int iaf_step_time_layer(iaf_stack_t* s, int layer, const float* z, const float* context, float* z_new,
                                   float* logsd, int B, int H, int W, void* workspace, size_t workspace_bytes, int reps,
                                   void* stream, float* avg_ms) {
    int rc = check_dims(s, B, H, W);
    if (rc) return rc;
    if (!z || !z_new || !logsd || !avg_ms || (s->depth_ar > 0 && !context)) return IAF_ERR_NULL;
    if (layer < 0 || layer >= s->nlayers || reps <= 0) return IAF_ERR_SHAPE;
    if (s->generic) return IAF_ERR_UNSUPPORTED;
    Ws ws;
    if ((rc = carve_ws(s, B, H, W, workspace, workspace_bytes, &ws))) return rc;
    hipStream_t st = (hipStream_t)stream;
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = B * H * W;
    p.x = z; p.zin = z; p.out0 = z_new; p.out1 = logsd; p.mode = MODE_IAF;
    Launch ls[MAX_GEMM_LAYERS];
    const int n = build_stack(s, p, IN_NCHW, context, nullptr, ws, ls);
    for (int i = 0; i < n; ++i)                      // one full step: every layer's input is valid scratch afterwards
        if ((rc = launch_conv(s, ls[i].layer, ls[i].p, ls[i].inmode, st))) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const int saved = s->prof_layer;
    s->prof_layer = -1;
    HIP_TRY(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
        if ((rc = launch_conv(s, ls[layer].layer, ls[layer].p, ls[layer].inmode, st))) break;
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    s->prof_layer = saved;
    *avg_ms = ms / (float)reps;
    return rc;
}

extern "C" int iaf_ar_multiconv2d_forward(iaf_stack_t* s, const float* z, const float* context, float* m_raw,
                                          float* s_raw, int B, int H, int W, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    int rc = check_dims(s, B, H, W);
    if (rc) return rc;
    if (!z || !m_raw || !s_raw || (s->depth_ar > 0 && !context)) return IAF_ERR_NULL;
    Ws ws;
    if ((rc = carve_ws(s, B, H, W, workspace, workspace_bytes, &ws))) return rc;
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = B * H * W;
    p.x = z; p.out0 = m_raw; p.out1 = s_raw; p.mode = MODE_RAW;
    return run_stack(s, p, IN_NCHW, context, nullptr, ws, (hipStream_t)stream);
}

extern "C" int iaf_step_forward(iaf_stack_t* s, const float* z, const float* context, float* z_new, float* logsd,
                                int B, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_dims(s, B, H, W);
    if (rc) return rc;
    if (!z || !z_new || !logsd || (s->depth_ar > 0 && !context)) return IAF_ERR_NULL;
    Ws ws;
    if ((rc = carve_ws(s, B, H, W, workspace, workspace_bytes, &ws))) return rc;
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = B * H * W;
    p.x = z; p.zin = z; p.out0 = z_new; p.out1 = logsd; p.mode = MODE_IAF;
    return run_stack(s, p, IN_NCHW, context, nullptr, ws, (hipStream_t)stream);
}

// Inverse of the IAF step: given the flow OUTPUT z, find z0 with (z0 - m(z0))/exp(s(z0)) = z.
// m, s at a position depend only on z0 at earlier positions of the autoregressive order, so the Jacobi iteration
//   z0 <- z * exp(s(z0)) + m(z0)
// fixes at least one more position per sweep (exact after at most H*W*n_z sweeps) and, because the reference scales
// m and s by 0.1, contracts to fp32 precision in a handful of sweeps; every sweep is one full-width run of the conv
// stack instead of H*W*n_z dependent scalar steps.
extern "C" int iaf_step_inverse(iaf_stack_t* s, const float* z, const float* context, float* z0, float* logsd, int B, int H,
                                int W, void* workspace, size_t workspace_bytes, int max_sweeps, float tol, int check_every,
                                void* stream, int* sweeps_done, float* residual) {
    int rc = check_dims(s, B, H, W);
    if (rc) return rc;
    if (!z || !z0 || !logsd || (s->depth_ar > 0 && !context)) return IAF_ERR_NULL;
    if (max_sweeps <= 0 || check_every < 0 || tol < 0.f) return IAF_ERR_SHAPE;
    Ws ws;
    if ((rc = carve_ws(s, B, H, W, workspace, workspace_bytes, &ws))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * s->n_z * H * W;
    // ping-pong between the caller's z0 and the workspace's kl_elem plane, arranged so that the last sweep lands in z0
    float* buf[2] = {z0, ws.kl_elem};
    unsigned* d_res = (unsigned*)ws.rowsum;               // [B*n_z] floats are free during the inverse: first word
    const bool checking = (tol > 0.f && check_every > 0);
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = B * H * W;
    p.zin = z; p.out1 = logsd; p.mode = MODE_INVERSE;
    const float* cur = z;                                 // initial guess z0 = z (m = 0, s = 0)
    int k = 0, done = 0;
    float res = -1.f;
    while (done < max_sweeps) {
        // choose the target so that parity works out if we stop at max_sweeps; an early stop may need one copy
        float* dst = buf[(max_sweeps - 1 - done) & 1];
        p.x = cur; p.out0 = dst;
        if ((rc = run_stack(s, p, IN_NCHW, context, nullptr, ws, st))) return rc;
        ++done;
        if (checking && (++k == check_every || done == max_sweeps)) {
            k = 0;
            HIP_TRY(hipMemsetAsync(d_res, 0, sizeof(unsigned), st));
            hipLaunchKernelGGL(iaf_maxdiff_kernel, ew_grid(n), dim3(256), 0, st, (const float*)dst, cur, n, d_res);
            unsigned bits = 0;
            HIP_TRY(hipMemcpyAsync(&bits, d_res, sizeof(unsigned), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            memcpy(&res, &bits, sizeof(float));
            if (res <= tol) { cur = dst; break; }
        }
        cur = dst;
    }
    if (cur != z0) HIP_TRY(hipMemcpyAsync(z0, cur, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (sweeps_done) *sweeps_done = done;
    if (residual) *residual = res;
    return IAF_OK;
}

extern "C" int iaf_posterior_block_forward(iaf_stack_t* s, const float* qz_mean, const float* qz_logsd,
                                           const float* rz_mean, const float* rz_logsd, const float* pz_mean,
                                           const float* pz_logsd, const float* up_context,
                                           const float* down_context, const float* eps, float kl_min, float* z_out,
                                           float* kl_obj, float* kl_cost, float* kl_elem, int B, int H, int W,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_dims(s, B, H, W);
    if (rc) return rc;
    if (!qz_mean || !qz_logsd || !rz_mean || !rz_logsd || !pz_mean || !pz_logsd || !eps || !z_out || !kl_obj ||
        !kl_cost)
        return IAF_ERR_NULL;
    if (s->depth_ar > 0 && (!up_context || !down_context)) return IAF_ERR_NULL;
    Ws ws;
    if ((rc = carve_ws(s, B, H, W, workspace, workspace_bytes, &ws))) return rc;
    hipStream_t st = (hipStream_t)stream;
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = B * H * W;
    p.qm = qz_mean; p.ql = qz_logsd; p.rm = rz_mean; p.rl = rz_logsd; p.pm = pz_mean; p.pl = pz_logsd; p.eps = eps;
    p.out0 = z_out; p.out1 = nullptr; p.kl_elem = kl_elem ? kl_elem : ws.kl_elem; p.mode = MODE_POSTERIOR;
    if ((rc = run_stack(s, p, IN_POSTERIOR, up_context, down_context, ws, st))) return rc;
    const int rows = B * s->n_z;
    hipLaunchKernelGGL(iaf_kl_rowsum_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, p.kl_elem, ws.rowsum, rows, H * W);
    hipLaunchKernelGGL(iaf_kl_finish_kernel, dim3(1), dim3(256), 0, st, ws.rowsum, kl_obj, kl_cost, B, s->n_z, kl_min);
    return (int)hipGetLastError();
}

static dim3 ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

extern "C" int iaf_gaussian_sample(const float* mean, const float* logvar, const float* noise, float* out, size_t n,
                                   void* stream) {
    if (!mean || !logvar || !noise || !out) return IAF_ERR_NULL;
    if (n == 0) return IAF_OK;
    hipLaunchKernelGGL(iaf_gauss_sample_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, mean, logvar, noise, out, n);
    return (int)hipGetLastError();
}

extern "C" int iaf_gaussian_logps(const float* mean, const float* logvar, const float* sample, float* out, size_t n,
                                  void* stream) {
    if (!mean || !logvar || !sample || !out) return IAF_ERR_NULL;
    if (n == 0) return IAF_OK;
    hipLaunchKernelGGL(iaf_gauss_logps_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, mean, logvar, sample, out, n);
    return (int)hipGetLastError();
}

extern "C" int iaf_lowerbound_stream_init(float* run_max, float* run_sum, int n, void* stream) {
    if (!run_max || !run_sum) return IAF_ERR_NULL;
    if (n <= 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_lb_init_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, run_max, run_sum, n);
    return (int)hipGetLastError();
}

extern "C" int iaf_lowerbound_stream_update(float* run_max, float* run_sum, const float* log_pxz, const float* sum_kl,
                                            int n, int k_chunk, void* stream) {
    if (!run_max || !run_sum || !log_pxz || !sum_kl) return IAF_ERR_NULL;
    if (n <= 0 || k_chunk <= 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_lb_update_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, run_max, run_sum,
                       log_pxz, sum_kl, n, k_chunk);
    return (int)hipGetLastError();
}

extern "C" int iaf_lowerbound_stream_finalize(const float* run_max, const float* run_sum, float* out, int n,
                                              int k_total, void* stream) {
    if (!run_max || !run_sum || !out) return IAF_ERR_NULL;
    if (n <= 0 || k_total <= 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_lb_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, run_max, run_sum,
                       out, n, k_total);
    return (int)hipGetLastError();
}

extern "C" int iaf_compute_lowerbound(const float* log_pxz, const float* sum_kl, float* out, int n, int k, void* stream) {
    if (!log_pxz || !sum_kl || !out) return IAF_ERR_NULL;
    if (n <= 0 || k <= 0) return IAF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (k == 1) {
        hipLaunchKernelGGL(iaf_lb_k1_kernel, dim3((n + 255) / 256), dim3(256), 0, st, log_pxz, sum_kl, out, n);
        return (int)hipGetLastError();
    }
    // single-chunk streaming pass with the state kept in `out` itself is not possible (two words
    // per image); k>1 callers provide state through the stream API.  For convenience allocate on
    // the stream-ordered pool.
    float* state = nullptr;
    HIP_TRY(hipMallocAsync((void**)&state, 2 * (size_t)n * sizeof(float), st));
    hipLaunchKernelGGL(iaf_lb_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, state, state + n, n);
    hipLaunchKernelGGL(iaf_lb_update_kernel, dim3((n + 3) / 4), dim3(256), 0, st, state, state + n, log_pxz, sum_kl, n, k);
    hipLaunchKernelGGL(iaf_lb_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, st, state, state + n, out, n, k);
    int rc = (int)hipGetLastError();
    HIP_TRY(hipFreeAsync(state, st));
    return rc;
}

// ---------------------------------------------------------------------------------------------
// training: forward that keeps the hidden activations + backward
// ---------------------------------------------------------------------------------------------
extern "C" int iaf_stack_set_training(iaf_stack_t* s, int on) {
    if (!s) return IAF_ERR_NULL;
    if (s->variant != IAF_VARIANT_TF || s->generic) return IAF_ERR_UNSUPPORTED;
    if (!on) { s->training = false; return IAF_OK; }
    for (int l = 0; l < s->nlayers; ++l) {
        GemmLayer& L = s->L[l];
        if (!L.wpt) HIP_TRY(hipMalloc(&L.wpt, (size_t)L.nchunk * NTAPS * L.ncot * 256 * sizeof(float)));
        GemmLayer& T = s->T[l];
        T = GemmLayer();
        T.cin = L.cout; T.cout = L.cin; T.nchunk = L.ncot; T.ncot = L.nchunk;
        T.zerodiag = L.zerodiag; T.npair = 1;
        T.wp = L.wpt; T.bias = nullptr; T.border = nullptr; T.lim = nullptr; T.wpt = nullptr;
        T.nt = 1; T.pxt = 4; T.wco = 1; T.ks = 1; T.user_tuned = false;
    }
    s->training = true;
    s->prepared = false;      // the transposed packs are written by the next prepare
    return IAF_OK;
}

// pixel ranges of the weight-gradient GEMM: as many as keep the grid within ONE round of 256 workgroups
// (grid.x = 5 taps * ceil(cin/32) ci pairs), at least 64 pixels each, at most 16 (the partial buffer is sized for 16)
static int wgrad_nrange(long long P, int cin = 160, int ntaps = NTAPS) {
    long long n = 512 / (ntaps * ((cin + 31) / 32));      // two workgroups per CU are resident
    if (n > P / 64) n = P / 64;
    if (n < 1) n = 1;
    if (n > 16) n = 16;
    return (int)n;
}

struct TrainWs {
    float* h[MAX_GEMM_LAYERS];
    float* da[2];
    float* dy3;
    float* zpm;
    float* part;
    float* dWeff[MAX_GEMM_LAYERS];   // per layer: reduced effective-weight gradient [NTAPS][cin][cout]
    float* dbp[MAX_GEMM_LAYERS];     // per layer: [<=256 slabs][cout] column sums of dY
    // posterior block: saved forward values and backward temporaries, all NCHW [P*n_z] unless noted
    float* logsd; float* klelem; float* z0; float* dzt; float* dkl; float* dz0;
    float* rowsum;   // [B*n_z]  (P*n_z floats reserved: B <= P)
    float* gate;     // [n_z]
    unsigned short* tapmask;   // [P] border bits for the weight gradient
};

static size_t train_ws_floats(const iaf_stack_t* s, long long P, TrainWs* o, float* base) {
    size_t off = 0;
    auto take = [&](size_t n) { float* q = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return q; };
    TrainWs t;
    for (int l = 0; l < s->depth_ar; ++l) t.h[l] = take((size_t)P * s->n_h);
    t.da[0] = take((size_t)P * s->n_h);
    t.da[1] = take((size_t)P * s->n_h);
    t.dy3 = take((size_t)P * 2 * s->n_z);
    t.zpm = take((size_t)P * s->n_z);
    size_t maxw = 0;
    for (int l = 0; l < s->nlayers; ++l) {
        const size_t w = (size_t)s->L[l].cin * s->L[l].cout;
        if (w > maxw) maxw = w;
    }
    t.part = take((size_t)16 * NTAPS * maxw);
    for (int l = 0; l < s->nlayers; ++l) {
        t.dWeff[l] = take((size_t)NTAPS * s->L[l].cin * s->L[l].cout);
        t.dbp[l] = take((size_t)256 * s->L[l].cout);
    }
    t.logsd = take((size_t)P * s->n_z); t.klelem = take((size_t)P * s->n_z); t.z0 = take((size_t)P * s->n_z);
    t.dzt = take((size_t)P * s->n_z); t.dkl = take((size_t)P * s->n_z); t.dz0 = take((size_t)P * s->n_z);
    t.rowsum = take((size_t)P * s->n_z);
    t.gate = take((size_t)s->n_z);
    t.tapmask = (unsigned short*)take(((size_t)P + 1) / 2);
    if (o) *o = t;
    return off;
}

extern "C" size_t iaf_stack_train_workspace_bytes(const iaf_stack_t* s, int B, int H, int W) {
    if (!s || B <= 0 || H <= 0 || W <= 0) return 0;
    return train_ws_floats(s, (long long)B * H * W, nullptr, nullptr) * sizeof(float);
}

extern "C" int iaf_step_forward_train(iaf_stack_t* s, const float* z, const float* context, float* z_new, float* logsd,
                                      int B, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_dims(s, B, H, W);
    if (rc) return rc;
    if (!s->training) return IAF_ERR_NOT_PREPARED;
    if (!z || !z_new || !logsd || !workspace || (s->depth_ar > 0 && !context)) return IAF_ERR_NULL;
    if (((uintptr_t)workspace & 15) != 0 || workspace_bytes < iaf_stack_train_workspace_bytes(s, B, H, W)) return IAF_ERR_WORKSPACE;
    TrainWs tw;
    train_ws_floats(s, (long long)B * H * W, &tw, (float*)workspace);
    hipStream_t st = (hipStream_t)stream;
    ConvP base;
    memset(&base, 0, sizeof(base));
    base.B = B; base.H = H; base.W = W; base.HW = H * W; base.P = B * H * W;
    base.zin = z; base.out0 = z_new; base.out1 = logsd; base.mode = MODE_IAF;
    const float* cur = z;
    int inmode = IN_NCHW;
    for (int l = 0; l < s->depth_ar; ++l) {       // like run_stack, but every hidden activation gets its own buffer
        ConvP p = base;
        p.mode = 0;
        p.x = cur;
        p.ctx = (l == 0) ? context : nullptr;
        p.y = tw.h[l];
        if ((rc = launch_conv(s, l, p, inmode, st))) return rc;
        cur = p.y;
        inmode = IN_PIXMAJOR;
    }
    ConvP p = base;
    p.x = cur;
    return launch_conv(s, s->depth_ar, p, inmode, st);
}

template <int NCOT>
static void launch_wgrad_t(const WgradP& p, dim3 grid, hipStream_t st) {
    const size_t lds = (size_t)4 * NCOT * 4 * 64 * sizeof(float);
    static bool attr_done = false;
    if (lds > 48 * 1024 && !attr_done) {
        (void)hipFuncSetAttribute((const void*)iaf_wgrad_kernel<NCOT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(iaf_wgrad_kernel<NCOT>, grid, dim3(256), lds, st, p);
}

static int launch_tapmask(unsigned short* mask, int B, int H, int W, hipStream_t st) {
    const int P = B * H * W;
    hipLaunchKernelGGL(iaf_tapmask_kernel, dim3((P + 255) / 256), dim3(256), 0, st, mask, H, W, P);
    return (int)hipGetLastError();
}

static int launch_wgrad(const iaf_stack_t* s, const GemmLayer& L, const float* x, const float* dy, float* part,
                        const unsigned short* tapmask, int B, int H, int W, hipStream_t st) {
    WgradP p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.dy = dy; p.part = part; p.tapmask = tapmask;
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = B * H * W;
    p.cin = L.cin; p.cout = L.cout;
    const int ntaps = L.full3x3 ? MAXTAPS : NTAPS;
    p.ntaps = ntaps;
    p.nrange = wgrad_nrange(p.P, L.cin, ntaps);
    p.px_per_range = (int)(((long long)p.P + p.nrange - 1) / p.nrange + 15) / 16 * 16;
    static const int tf_dh[NTAPS] = {0, 0, 1, 1, 1}, tf_dw[NTAPS] = {0, 1, -1, 0, 1};
    for (int t = 0; t < ntaps; ++t) {
        p.tap_dh[t] = L.full3x3 ? t / 3 - 1 : tf_dh[t];
        p.tap_dw[t] = L.full3x3 ? t % 3 - 1 : tf_dw[t];
    }
    static const int cand[] = {14, 12, 10, 8, 7, 6, 5, 4, 3, 2, 1};
    int ncot = 1;
    for (int c : cand)
        if (L.ncot % c == 0) { ncot = c; break; }
    dim3 grid(ntaps * ((L.cin + 31) / 32), p.nrange, L.ncot / ncot);
    switch (ncot) {
        case 14: launch_wgrad_t<14>(p, grid, st); break;
        case 12: launch_wgrad_t<12>(p, grid, st); break;
        case 7: launch_wgrad_t<7>(p, grid, st); break;
        case 10: launch_wgrad_t<10>(p, grid, st); break;
        case 8: launch_wgrad_t<8>(p, grid, st); break;
        case 6: launch_wgrad_t<6>(p, grid, st); break;
        case 5: launch_wgrad_t<5>(p, grid, st); break;
        case 4: launch_wgrad_t<4>(p, grid, st); break;
        case 3: launch_wgrad_t<3>(p, grid, st); break;
        case 2: launch_wgrad_t<2>(p, grid, st); break;
        default: launch_wgrad_t<1>(p, grid, st); break;
    }
    (void)s;
    return (int)hipGetLastError();
}

extern "C" int iaf_step_backward(iaf_stack_t* s, const float* z, const float* context, const float* z_new,
                                 const float* logsd, const float* dz_new, const float* dlogsd, float* dz, float* dcontext,
                                 const float* const* V, const float* const* g, float* const* dV, float* const* dg,
                                 float* const* db, int B, int H, int W, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    int rc = check_dims(s, B, H, W);
    if (rc) return rc;
    if (!s->training) return IAF_ERR_NOT_PREPARED;
    if (!z || !z_new || !logsd || !dz_new || !dlogsd || !dz || !V || !g || !dV || !dg || !db || !workspace) return IAF_ERR_NULL;
    if (s->depth_ar > 0 && (!context || !dcontext)) return IAF_ERR_NULL;
    for (int i = 0; i < s->depth_ar + 2; ++i)
        if (!V[i] || !g[i] || !dV[i] || !dg[i] || !db[i]) return IAF_ERR_NULL;
    if (((uintptr_t)workspace & 15) != 0 || workspace_bytes < iaf_stack_train_workspace_bytes(s, B, H, W)) return IAF_ERR_WORKSPACE;
    const int P = B * H * W, d = s->depth_ar;
    TrainWs tw;
    train_ws_floats(s, P, &tw, (float*)workspace);
    hipStream_t st = (hipStream_t)stream;

    // (1) affine + log-det backward -> packed pixel-major dY of the output GEMM, pixel-major copy of z
    {
        const long long total = (long long)P * s->n_z;
        long long blocks = (total + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(iaf_bwd_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, st, z, z_new, logsd, dz_new, dlogsd,
                           tw.dy3, tw.zpm, s->n_z, H * W, total);
    }
    ConvP base;
    memset(&base, 0, sizeof(base));
    base.B = B; base.H = H; base.W = W; base.HW = H * W; base.P = P;

    const int nslab = (P + 31) / 32 < 256 ? (P + 31) / 32 : 256;
    const int px_per_slab = (P + nslab - 1) / nslab;
    auto reduce = [&](int l, const float* dy) {     // partials -> dWeff[l], dY column sums -> dbp[l]
        const GemmLayer& L = s->L[l];
        const size_t n4 = (size_t)NTAPS * L.cin * L.cout / 4;
        int nblk = (int)((n4 + 255) / 256);
        if (nblk > 1024) nblk = 1024;
        hipLaunchKernelGGL(iaf_wgrad_reduce_kernel, dim3(nblk + nslab), dim3(256), 0, st, tw.part, tw.dWeff[l],
                           wgrad_nrange(P, L.cin), n4, nblk, dy, tw.dbp[l], P, L.cout, px_per_slab);
    };
    WnBwdArgs wa;
    memset(&wa, 0, sizeof(wa));
    auto wn_add = [&](int conv_index, int l, int n_out_each, int pack_stride, int pack_off) {
        const GemmLayer& L = s->L[l];
        WnBwdLayer& w = wa.L[wa.n];
        w.V = V[conv_index]; w.g = g[conv_index];
        w.dW = tw.dWeff[l]; w.dbp = tw.dbp[l];
        w.dV = dV[conv_index]; w.dg = dg[conv_index]; w.db = db[conv_index];
        w.cin = L.cin; w.cout = n_out_each; w.cout_packed = L.cout; w.nslab = nslab; w.zerodiag = L.zerodiag;
        w.pack_stride = pack_stride; w.pack_off = pack_off;
        wa.tile_begin[wa.n + 1] = wa.tile_begin[wa.n] + n_out_each / 16;
        wa.n++;
    };

    if ((rc = launch_tapmask(tw.tapmask, B, H, W, st))) return rc;
    // (2) walk the layers backwards: data gradient (same conv kernel on W^T, mirrored taps), then weight gradient
    const float* dy = tw.dy3;                       // gradient w.r.t. the output of layer l (packed pixel-major)
    for (int l = d; l >= 0; --l) {
        const float* x_in = (l == 0) ? tw.zpm : tw.h[l - 1];     // what layer l read in the forward pass
        ConvP p = base;
        p.x = dy;
        if (l == 0) {                               // dz = W_0^T dY + dz_new e^{-logsd}
            p.mode = MODE_DGRAD_Z;
            p.qm = dz_new; p.ql = logsd; p.out0 = dz;
        } else {                                    // d a_{l-1} = (W_l^T dY) elu'(h_{l-1})   (+ NCHW copy = d context)
            p.mode = MODE_DGRAD_ELU;
            p.zin = tw.h[l - 1];
            p.y = tw.da[(l - 1) & 1];
            p.out0 = (l - 1 == 0) ? dcontext : nullptr;
        }
        if ((rc = launch_gemm(s, s->T[l], EPI_DGRAD, true, -1, p, IN_PIXMAJOR, st))) return rc;
        if ((rc = launch_wgrad(s, s->L[l], x_in, dy, tw.part, tw.tapmask, B, H, W, st))) return rc;
        reduce(l, dy);
        if (l == d) {
            wn_add(d, l, s->n_z, 2, 0);       // layer_out_0 (mean tiles)
            wn_add(d + 1, l, s->n_z, 2, 1);   // layer_out_1 (logsd tiles)
        } else {
            wn_add(l, l, s->L[l].cout, 1, 0);
        }
        if (l > 0) dy = tw.da[(l - 1) & 1];
    }
    // (3) mask + weight-norm backward of every conv of the stack in one launch
    hipLaunchKernelGGL(iaf_wn_bwd_kernel, dim3(wa.tile_begin[wa.n]), dim3(256), 0, st, wa);
    return (int)hipGetLastError();
}

extern "C" int iaf_posterior_block_forward_train(iaf_stack_t* s, const float* qz_mean, const float* qz_logsd,
                                                 const float* rz_mean, const float* rz_logsd, const float* pz_mean,
                                                 const float* pz_logsd, const float* up_context, const float* down_context,
                                                 const float* eps, float kl_min, float* z_out, float* kl_obj, float* kl_cost,
                                                 int B, int H, int W, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_dims(s, B, H, W);
    if (rc) return rc;
    if (!s->training) return IAF_ERR_NOT_PREPARED;
    if (!qz_mean || !qz_logsd || !rz_mean || !rz_logsd || !pz_mean || !pz_logsd || !eps || !z_out || !kl_obj || !kl_cost ||
        !workspace)
        return IAF_ERR_NULL;
    if (s->depth_ar > 0 && (!up_context || !down_context)) return IAF_ERR_NULL;
    if (((uintptr_t)workspace & 15) != 0 || workspace_bytes < iaf_stack_train_workspace_bytes(s, B, H, W)) return IAF_ERR_WORKSPACE;
    TrainWs tw;
    train_ws_floats(s, (long long)B * H * W, &tw, (float*)workspace);
    hipStream_t st = (hipStream_t)stream;
    ConvP base;
    memset(&base, 0, sizeof(base));
    base.B = B; base.H = H; base.W = W; base.HW = H * W; base.P = B * H * W;
    base.qm = qz_mean; base.ql = qz_logsd; base.rm = rz_mean; base.rl = rz_logsd; base.pm = pz_mean; base.pl = pz_logsd;
    base.eps = eps;
    base.out0 = z_out; base.out1 = tw.logsd; base.kl_elem = tw.klelem; base.mode = MODE_POSTERIOR;
    const float* cur = nullptr;
    int inmode = IN_POSTERIOR;
    for (int l = 0; l < s->depth_ar; ++l) {
        ConvP p = base;
        p.x = cur;
        p.ctx = (l == 0) ? up_context : nullptr;
        p.ctx2 = (l == 0) ? down_context : nullptr;
        p.y = tw.h[l];
        if ((rc = launch_conv(s, l, p, inmode, st))) return rc;
        cur = p.y;
        inmode = IN_PIXMAJOR;
    }
    ConvP p = base;
    p.x = cur;
    if ((rc = launch_conv(s, s->depth_ar, p, inmode, st))) return rc;
    const int rows = B * s->n_z;
    hipLaunchKernelGGL(iaf_kl_rowsum_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, tw.klelem, tw.rowsum, rows, H * W);
    hipLaunchKernelGGL(iaf_kl_finish_kernel, dim3(1), dim3(256), 0, st, tw.rowsum, kl_obj, kl_cost, B, s->n_z, kl_min);
    return (int)hipGetLastError();
}

extern "C" int iaf_posterior_block_backward(iaf_stack_t* s, const float* qz_mean, const float* qz_logsd, const float* rz_mean,
                                            const float* rz_logsd, const float* pz_mean, const float* pz_logsd,
                                            const float* eps, float kl_min, const float* z, const float* dz,
                                            const float* dkl_obj, float* dmean, float* dlogsd_q, float* dpz_mean,
                                            float* dpz_logsd, float* dcontext, const float* const* V, const float* const* g,
                                            float* const* dV, float* const* dg, float* const* db, int B, int H, int W,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_dims(s, B, H, W);
    if (rc) return rc;
    if (!s->training) return IAF_ERR_NOT_PREPARED;
    if (!qz_mean || !qz_logsd || !rz_mean || !rz_logsd || !pz_mean || !pz_logsd || !eps || !z || !dkl_obj || !dmean ||
        !dlogsd_q || !dpz_mean || !dpz_logsd || !workspace)
        return IAF_ERR_NULL;
    if (((uintptr_t)workspace & 15) != 0 || workspace_bytes < iaf_stack_train_workspace_bytes(s, B, H, W)) return IAF_ERR_WORKSPACE;
    TrainWs tw;
    train_ws_floats(s, (long long)B * H * W, &tw, (float*)workspace);
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)B * s->n_z * H * W;
    if (kl_min > 0.f)
        hipLaunchKernelGGL(iaf_post_bwd_gate_kernel, dim3(1), dim3(256), 0, st, tw.rowsum, dkl_obj, tw.gate, B, s->n_z, kl_min);
    hipLaunchKernelGGL(iaf_post_bwd_pre_kernel, ew_grid(n), dim3(256), 0, st, qz_mean, qz_logsd, rz_mean, rz_logsd, pz_mean,
                       pz_logsd, eps, z, dz, tw.gate, dkl_obj, kl_min, tw.z0, tw.dzt, tw.dkl, dpz_mean, dpz_logsd, s->n_z, H * W,
                       n);
    if ((rc = (int)hipGetLastError())) return rc;
    // core: z := z0, (z_new, logsd) := saved forward values, incoming gradients := (dz_tot, dkl)
    if ((rc = iaf_step_backward(s, tw.z0, dcontext /* value unused */, z, tw.logsd, tw.dzt, tw.dkl, tw.dz0, dcontext, V, g, dV, dg,
                                db, B, H, W, workspace, workspace_bytes, stream)))
        return rc;
    hipLaunchKernelGGL(iaf_post_bwd_post_kernel, ew_grid(n), dim3(256), 0, st, qz_mean, rz_mean, tw.z0, tw.dz0, tw.dkl, dmean,
                       dlogsd_q, n);
    return (int)hipGetLastError();
}

extern "C" int iaf_adamax_ema_step(float* var, const float* grad, float* slot_m, float* slot_v, float* ema, size_t n, float lr,
                                   float beta1, float beta2, float eps, float ema_decay, float grad_scale, void* stream) {
    if (!var || !grad || !slot_m || !slot_v) return IAF_ERR_NULL;
    if (n == 0) return IAF_OK;
    const bool al = (((uintptr_t)var | (uintptr_t)grad | (uintptr_t)slot_m | (uintptr_t)slot_v | (uintptr_t)ema) & 15) == 0;
    const size_t n4 = al ? n / 4 : 0;
    hipLaunchKernelGGL(iaf_adamax_ema_kernel, ew_grid(n4 ? n4 : n), dim3(256), 0, (hipStream_t)stream, var, grad, slot_m, slot_v,
                       ema, n4, n, lr, beta1, beta2, eps, ema_decay, grad_scale);
    return (int)hipGetLastError();
}

extern "C" int iaf_layer_work(const iaf_stack_t* s, int layer, int B, int H, int W, double* live_flops,
                              double* dense_flops, double* bytes) {
    if (!s) return IAF_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || layer < 0 || layer >= s->nlayers) return IAF_ERR_SHAPE;
    const GemmLayer& L = s->L[layer];
    const double px = (double)B * H * W;
    if (live_flops) *live_flops = 2.0 * L.live_macs_per_px * px;
    if (dense_flops) *dense_flops = 2.0 * L.dense_macs_per_px * px;
    if (bytes) {
        double b = 4.0 * px * L.cin;                                     // activations in
        if (layer == s->depth_ar) b += 4.0 * px * (3.0 * s->n_z);         // z in, z_new out, logsd out
        else b += 4.0 * px * L.cout * (layer == 0 ? 2.0 : 1.0);           // hidden out (+ context in)
        b += 4.0 * ((double)L.nchunk * NTAPS * L.ncot * 256 + L.cout);    // packed weights + bias, once
        *bytes = b;
    }
    return IAF_OK;
}

extern "C" int iaf_step_work(const iaf_stack_t* s, int B, int H, int W, double* live_flops, double* dense_flops,
                             double* bytes) {
    if (!s) return IAF_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return IAF_ERR_SHAPE;
    const double px = (double)B * H * W;
    double live = 0, dense = 0;
    for (int l = 0; l < s->nlayers; ++l) { live += s->L[l].live_macs_per_px; dense += s->L[l].dense_macs_per_px; }
    if (live_flops) *live_flops = 2.0 * live * px;
    if (dense_flops) *dense_flops = 2.0 * dense * px;
    // fused algorithmic bytes: z in, context in, z out, s out + raw weights once
    if (bytes) *bytes = 4.0 * (3.0 * s->n_z + (s->depth_ar > 0 ? s->n_h : 0)) * px + (double)s->weight_bytes;
    return IAF_OK;
}

// ---------------------------------------------------------------------------------------------
// plain weight-normed 3x3 convs around the IAF step: up_conv1 / up_conv3 / down_conv1 / down_conv2
// (tf_train.py:36-44, 52-54, 87-94; operator tf_utils/layers.py:31-64 with mask=None, stride 1, pad SAME).
// Same implicit-GEMM kernel as the masked stack with all 9 taps live (template NTP = 9) and the EPI_PLAIN epilogue:
// ELU / channel concat fused into the input staging, channel split / residual fused into the store.
// ---------------------------------------------------------------------------------------------
struct iaf_conv3x3 {
    int n_in, n_out;
    int mask_mode;     // 0 plain conv2d, 1 ar_conv2d(zerodiagonal=False), 2 ar_conv2d(zerodiagonal=True)
    bool generic, prepared;
    GemmLayer L;
    PrepLayer* h_desc = nullptr;   // pinned staging of the prep descriptor
    PrepLayer* d_desc = nullptr;
    bool training = false;
    GemmLayer T;                   // transposed problem dX = W^T dY (valid when training)
};

extern "C" int iaf_conv3x3_destroy(iaf_conv3x3_t* c) {
    if (!c) return IAF_ERR_NULL;
    if (c->L.wp) (void)hipFree(c->L.wp);
    if (c->L.bias) (void)hipFree(c->L.bias);
    if (c->L.wpt) (void)hipFree(c->L.wpt);
    if (c->h_desc) (void)hipHostFree(c->h_desc);
    if (c->d_desc) (void)hipFree(c->d_desc);
    delete c;
    return IAF_OK;
}

static int conv3x3_create(iaf_conv3x3_t** out, int n_in, int n_out, int mask_mode);
extern "C" int iaf_conv3x3_create(iaf_conv3x3_t** out, int n_in, int n_out) { return conv3x3_create(out, n_in, n_out, 0); }
extern "C" int iaf_conv3x3_create_masked(iaf_conv3x3_t** out, int n_in, int n_out, int zerodiagonal) {
    if (n_in > 0 && n_out > 0 && !(n_in % n_out == 0 || n_out % n_in == 0)) return IAF_ERR_NOT_MULTIPLE;   // layers.py:116
    return conv3x3_create(out, n_in, n_out, zerodiagonal ? 2 : 1);
}

static int conv3x3_create(iaf_conv3x3_t** out, int n_in, int n_out, int mask_mode) {
    if (!out) return IAF_ERR_NULL;
    *out = nullptr;
    if (n_in <= 0 || n_out <= 0) return IAF_ERR_SHAPE;
    iaf_conv3x3* c = new (std::nothrow) iaf_conv3x3();
    if (!c) return (int)hipErrorOutOfMemory;
    c->n_in = n_in; c->n_out = n_out; c->prepared = false; c->mask_mode = mask_mode;
    c->generic = (n_in % 16 != 0 || n_out % 16 != 0 || n_in > 16 * PREP_MAXI);
    GemmLayer& L = c->L;
    L.cin = n_in; L.cout = n_out; L.npair = 1; L.zerodiag = (mask_mode == 2) ? 1 : 0; L.full3x3 = (mask_mode == 0);
    L.nchunk = (n_in + 15) / 16; L.ncot = (n_out + 15) / 16;
    default_tuning(L, false);
    L.live_macs_per_px = L.dense_macs_per_px = 9.0 * n_in * n_out;
    if (mask_mode) count_macs(L, n_in, n_out, L.zerodiag, 1);
    const size_t wfloats = c->generic ? (size_t)MAXTAPS * n_in * n_out : (size_t)L.nchunk * MAXTAPS * L.ncot * 256;
    int rc;
    if ((rc = (int)hipMalloc(&L.wp, wfloats * sizeof(float))) != 0 ||
        (rc = (int)hipMalloc(&L.bias, (size_t)L.ncot * 16 * sizeof(float))) != 0 ||
        (rc = (int)hipHostMalloc((void**)&c->h_desc, sizeof(PrepLayer))) != 0 ||
        (rc = (int)hipMalloc((void**)&c->d_desc, sizeof(PrepLayer))) != 0) {
        iaf_conv3x3_destroy(c);
        return rc;
    }
    *out = c;
    return IAF_OK;
}

extern "C" int iaf_conv3x3_prepare(iaf_conv3x3_t* c, const float* V, const float* g, const float* b, void* stream) {
    if (!c || !V || !g || !b) return IAF_ERR_NULL;
    const GemmLayer& L = c->L;
    if (c->generic) {
        GenPrepArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.nlayers = 1;
        GenPrepLayer& P = ga.L[0];
        P.V[0] = V; P.g[0] = g; P.b[0] = b; P.w = L.wp; P.bias = L.bias;
        P.cin = L.cin; P.cout_each = L.cout; P.npair = 1; P.zerodiag = L.zerodiag; P.ch_begin = 0; P.ntaps = MAXTAPS;
        P.mask9 = c->mask_mode ? 1 : 0;
        hipLaunchKernelGGL(iaf_generic_prep_kernel, dim3(L.cout), dim3(256), 0, (hipStream_t)stream, ga);
    } else if (c->mask_mode) {    // the masked prep of the stack, one layer
        PrepArgs a;
        memset(&a, 0, sizeof(a));
        a.nlayers = 1;
        PrepLayer& P = a.L[0];
        P.V[0] = V; P.g[0] = g; P.b[0] = b; P.wp = L.wp; P.bias = L.bias; P.variant = IAF_VARIANT_TF;
        P.cin = L.cin; P.cout_each = L.cout; P.ncot = L.ncot; P.nchunk = L.nchunk; P.zerodiag = L.zerodiag; P.npair = 1;
        hipLaunchKernelGGL(iaf_prep_kernel, dim3(L.ncot), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        PrepLayer& P = *c->h_desc;
        memset(&P, 0, sizeof(P));
        P.V[0] = V; P.g[0] = g; P.b[0] = b; P.wp = L.wp; P.bias = L.bias; P.variant = PREP_PLAIN9;
        P.wpt = c->training ? L.wpt : nullptr;
        P.cin = L.cin; P.cout_each = L.cout; P.ncot = L.ncot; P.nchunk = L.nchunk; P.npair = 1; P.tile_begin = 0;
        HIP_TRY(hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(PrepLayer), hipMemcpyHostToDevice, (hipStream_t)stream));
        hipLaunchKernelGGL(iaf_prep_plain_kernel, dim3(L.ncot), dim3(256), 0, (hipStream_t)stream, c->d_desc, (const int*)nullptr);
    }
    HIP_TRY(hipGetLastError());
    c->prepared = true;
    return IAF_OK;
}

// weight prep of many plain convs in one launch (the four convs of every IAFLayer of a model): descriptors in device
// memory, refreshed per run like iaf_prep_batch_run
struct iaf_conv3x3_prep_batch {
    int n, ntiles;
    iaf_conv3x3** convs;
    PrepLayer* h_layers;
    PrepLayer* d_layers;
    int* d_tile2layer;
};

extern "C" int iaf_conv3x3_prep_batch_destroy(iaf_conv3x3_prep_batch_t* b) {
    if (!b) return IAF_ERR_NULL;
    if (b->h_layers) (void)hipHostFree(b->h_layers);
    if (b->d_layers) (void)hipFree(b->d_layers);
    if (b->d_tile2layer) (void)hipFree(b->d_tile2layer);
    free(b->convs);
    delete b;
    return IAF_OK;
}

extern "C" int iaf_conv3x3_prep_batch_create(iaf_conv3x3_prep_batch_t** out, iaf_conv3x3_t* const* convs, int n) {
    if (!out || !convs) return IAF_ERR_NULL;
    *out = nullptr;
    if (n <= 0) return IAF_ERR_SHAPE;
    iaf_conv3x3_prep_batch* b = new (std::nothrow) iaf_conv3x3_prep_batch();
    if (!b) return (int)hipErrorOutOfMemory;
    memset(b, 0, sizeof(*b));
    b->n = n;
    b->convs = (iaf_conv3x3**)calloc(n, sizeof(iaf_conv3x3*));
    int nt = 0;
    for (int i = 0; i < n; ++i) {
        if (!convs[i]) { iaf_conv3x3_prep_batch_destroy(b); return IAF_ERR_NULL; }
        if (convs[i]->generic || convs[i]->mask_mode) { iaf_conv3x3_prep_batch_destroy(b); return IAF_ERR_UNSUPPORTED; }
        b->convs[i] = convs[i];
        nt += convs[i]->L.ncot;
    }
    b->ntiles = nt;
    int* t2l = (int*)malloc(sizeof(int) * nt);
    int rc;
    if ((rc = (int)hipHostMalloc((void**)&b->h_layers, sizeof(PrepLayer) * n)) != 0 ||
        (rc = (int)hipMalloc((void**)&b->d_layers, sizeof(PrepLayer) * n)) != 0 ||
        (rc = (int)hipMalloc((void**)&b->d_tile2layer, sizeof(int) * nt)) != 0) {
        free(t2l); iaf_conv3x3_prep_batch_destroy(b); return rc;
    }
    memset(b->h_layers, 0, sizeof(PrepLayer) * n);
    int tile = 0;
    for (int i = 0; i < n; ++i) {
        const GemmLayer& L = convs[i]->L;
        PrepLayer& P = b->h_layers[i];
        P.wp = L.wp; P.bias = L.bias; P.variant = PREP_PLAIN9;
        P.cin = L.cin; P.cout_each = L.cout; P.ncot = L.ncot; P.nchunk = L.nchunk; P.npair = 1; P.tile_begin = tile;
        for (int t = 0; t < L.ncot; ++t) t2l[tile++] = i;
    }
    rc = (int)hipMemcpy(b->d_tile2layer, t2l, sizeof(int) * nt, hipMemcpyHostToDevice);
    free(t2l);
    if (rc) { iaf_conv3x3_prep_batch_destroy(b); return rc; }
    *out = b;
    return IAF_OK;
}

extern "C" int iaf_conv3x3_prep_batch_run(iaf_conv3x3_prep_batch_t* b, const float* const* V, const float* const* g,
                                          const float* const* bias, void* stream) {
    if (!b || !V || !g || !bias) return IAF_ERR_NULL;
    for (int i = 0; i < b->n; ++i) {
        if (!V[i] || !g[i] || !bias[i]) return IAF_ERR_NULL;
        b->h_layers[i].V[0] = V[i]; b->h_layers[i].g[0] = g[i]; b->h_layers[i].b[0] = bias[i];
        b->h_layers[i].wpt = b->convs[i]->training ? b->convs[i]->L.wpt : nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(b->d_layers, b->h_layers, sizeof(PrepLayer) * b->n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(iaf_prep_plain_kernel, dim3(b->ntiles), dim3(256), 0, st, b->d_layers, b->d_tile2layer);
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < b->n; ++i) b->convs[i]->prepared = true;
    return IAF_OK;
}

extern "C" int iaf_conv3x3_set_tuning(iaf_conv3x3_t* c, int nt, int pxt, int wco, int ks) {
    if (!c) return IAF_ERR_NULL;
    GemmLayer& L = c->L;
    if (nt == 0) { L.user_tuned = false; return IAF_OK; }     // back to the automatic choice
    if (c->generic || !pick_kernel(nt, pxt, wco, ks, IN_NCHW, c->mask_mode ? EPI_PLAIN5 : EPI_PLAIN)) return IAF_ERR_UNSUPPORTED;
    if (L.ncot % (nt * wco) != 0 || L.nchunk < ks) return IAF_ERR_UNSUPPORTED;
    L.nt = nt; L.pxt = pxt; L.wco = wco; L.ks = ks; L.user_tuned = true;
    return IAF_OK;
}

// launch of the conv kernel for a plain / single masked conv descriptor (forward, or its transposed problem with the
// taps mirrored for the data gradient)
static int conv3x3_launch(GemmLayer& L, ConvP& p, int epi_sel, int inmode, bool masked, bool mirror, hipStream_t st) {
    if (!L.user_tuned) auto_shape(L, false, p.P, p.W);
    conv_fn_t fn = pick_kernel(L.nt, L.pxt, L.wco, L.ks, inmode, epi_sel);
    if (!fn) return IAF_ERR_UNSUPPORTED;
    const int tm = 16 * L.pxt, W = p.W, sgn = mirror ? -1 : 1;
    p.wp = L.wp; p.bias = L.bias; p.lim = nullptr;
    if (masked) {     // the 5 live taps of the MADE-masked filter: look right / below only
        static const int tf_dh[NTAPS] = {0, 0, 1, 1, 1}, tf_dw[NTAPS] = {0, 1, -1, 0, 1};
        for (int t = 0; t < NTAPS; ++t) { p.tap_dh[t] = sgn * tf_dh[t]; p.tap_dw[t] = sgn * tf_dw[t]; }
        p.halo_before = mirror ? W + 1 : 0;
        p.nslot = tm + W + 1;
    } else {
        for (int t = 0; t < MAXTAPS; ++t) { p.tap_dh[t] = sgn * (t / 3 - 1); p.tap_dw[t] = sgn * (t % 3 - 1); }   // cross-correlation, SAME
        p.halo_before = W + 1;
        p.nslot = tm + 2 * (W + 1);
    }
    p.cin = L.cin; p.cout = L.cout; p.nchunk = L.nchunk; p.ncot = L.ncot;
    p.cp = L.cin + 8;
    const size_t lds = conv_lds_bytes(L, W);
    if (lds > 160 * 1024) return IAF_ERR_UNSUPPORTED;
    int rc = raise_lds_cap(fn, lds);
    if (rc) return rc;
    dim3 grid((p.P + tm - 1) / tm, L.ncot / (L.nt * L.wco));
    p.gx = (int)grid.x;
    hipLaunchKernelGGL(fn, grid, dim3(64 * L.pxt * L.wco * L.ks), lds, st, p);
    return (int)hipGetLastError();
}

extern "C" int iaf_conv3x3_forward(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                                   const float* residual, float* const* outs, const int* out_channels, int n_outs, int B,
                                   int H, int W, void* stream) {
    if (!c || !x || !outs || !out_channels) return IAF_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || n_outs < 1 || n_outs > MAXSPLIT) return IAF_ERR_SHAPE;
    if ((long long)B * H * W > (1LL << 30) / 64) return IAF_ERR_SHAPE;
    if (!c->prepared) return IAF_ERR_NOT_PREPARED;
    if (x2 && (c_split <= 0 || c_split >= c->n_in)) return IAF_ERR_SHAPE;
    if (residual && n_outs != 1) return IAF_ERR_SHAPE;
    int tot = 0, ends[MAXSPLIT];
    for (int k = 0; k < n_outs; ++k) {
        if (!outs[k]) return IAF_ERR_NULL;
        if (out_channels[k] <= 0) return IAF_ERR_SHAPE;
        tot += out_channels[k];
        ends[k] = tot;
    }
    if (tot != c->n_out) return IAF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    GemmLayer& L = c->L;
    if (c->generic) {
        GenPlainP p;
        memset(&p, 0, sizeof(p));
        p.x = x; p.x2 = x2; p.w = L.wp; p.bias = L.bias; p.res = residual;
        p.B = B; p.H = H; p.W = W; p.cin = L.cin; p.cout = L.cout; p.c_split = c_split; p.in_elu = elu_input ? 1 : 0;
        p.nsplit = n_outs;
        for (int k = 0; k < n_outs; ++k) { p.split_end[k] = ends[k]; p.split_ptr[k] = outs[k]; }
        hipLaunchKernelGGL(iaf_generic_conv3x3_kernel, ew_grid((size_t)B * L.cout * H * W), dim3(256), 0, st, p);
        return (int)hipGetLastError();
    }
    // MFMA path: a lane owns 4 consecutive channels, so the concat point and the split points must be multiples of 4
    if (x2 && (c_split & 3)) return IAF_ERR_UNSUPPORTED;
    for (int k = 0; k < n_outs; ++k)
        if (ends[k] & 3) return IAF_ERR_UNSUPPORTED;
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = B * H * W;
    p.x = x; p.x2 = x2; p.c_split = c_split; p.in_elu = elu_input ? 1 : 0; p.res = residual;
    p.nsplit = n_outs;
    for (int k = 0; k < MAXSPLIT; ++k) {
        p.split_end[k] = ends[k < n_outs ? k : n_outs - 1];
        p.split_ptr[k] = outs[k < n_outs ? k : n_outs - 1];
    }
    return conv3x3_launch(L, p, c->mask_mode ? EPI_PLAIN5 : EPI_PLAIN, IN_NCHW, c->mask_mode != 0, false, st);
}

extern "C" int iaf_conv3x3_autotune(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                                    const float* residual, float* const* outs, const int* out_channels, int n_outs, int B,
                                    int H, int W, int reps, void* stream, int* best_shape, float* best_us) {
    if (!c) return IAF_ERR_NULL;
    if (c->generic) return IAF_OK;
    if (reps <= 0) reps = 20;
    hipStream_t st = (hipStream_t)stream;
    GemmLayer& L = c->L;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    float best = 1e30f;
    int bsh[4] = {L.nt, L.pxt, L.wco, L.ks};
    int rc = IAF_OK;
    for (int si = 0; si < 8 && rc == IAF_OK; ++si)
        for (int nt = 5; nt >= 1 && rc == IAF_OK; --nt) {
            const int pxt = k_shapes[si][0], wco = k_shapes[si][1], ks = k_shapes[si][2];
            if (L.ncot % (nt * wco) != 0 || L.nchunk < ks) continue;
            GemmLayer t = L;
            t.nt = nt; t.pxt = pxt; t.wco = wco; t.ks = ks;
            if (conv_lds_bytes(t, W) > 160 * 1024) continue;
            L.nt = nt; L.pxt = pxt; L.wco = wco; L.ks = ks; L.user_tuned = true;
            for (int r = 0; r < 3 && rc == IAF_OK; ++r)
                rc = iaf_conv3x3_forward(c, x, x2, c_split, elu_input, residual, outs, out_channels, n_outs, B, H, W, stream);
            if (rc == IAF_ERR_UNSUPPORTED) { rc = IAF_OK; continue; }
            if (rc) break;
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps && rc == IAF_OK; ++r)
                rc = iaf_conv3x3_forward(c, x, x2, c_split, elu_input, residual, outs, out_channels, n_outs, B, H, W, stream);
            (void)hipEventRecord(e1, st);
            if (rc) break;
            if ((rc = (int)hipEventSynchronize(e1)) != 0) break;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) { best = ms; bsh[0] = nt; bsh[1] = pxt; bsh[2] = wco; bsh[3] = ks; }
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    L.nt = bsh[0]; L.pxt = bsh[1]; L.wco = bsh[2]; L.ks = bsh[3]; L.user_tuned = true;
    if (best_shape) for (int i = 0; i < 4; ++i) best_shape[i] = bsh[i];
    if (best_us) *best_us = best * 1e3f / reps;
    return rc;
}

extern "C" int iaf_conv3x3_work(const iaf_conv3x3_t* c, int B, int H, int W, double* flops, double* bytes) {
    if (!c) return IAF_ERR_NULL;
    const double P = (double)B * H * W;
    if (flops) *flops = 2.0 * c->L.live_macs_per_px * P;
    // input + output activations once, raw V/g/b once
    if (bytes) *bytes = 4.0 * (P * (c->n_in + c->n_out) + 9.0 * c->n_in * c->n_out + 2.0 * c->n_out);
    return IAF_OK;
}

extern "C" int iaf_datainit_normalize(const float* x_init, const float* add, float* y, float* g, float* b, int B, int C,
                                      int HW, float init_scale, void* stream) {
    if (!x_init || !g || !b) return IAF_ERR_NULL;
    if (B <= 0 || C <= 0 || HW <= 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_datainit_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x_init, add, y, g, b, B, C, HW, init_scale);
    return (int)hipGetLastError();
}

extern "C" int iaf_discretized_logistic(const float* mean, const float* logscale, int logscale_is_scalar, const float* sample,
                                        float* out, int B, size_t n_per_row, float binsize, void* stream) {
    if (!mean || !logscale || !sample || !out) return IAF_ERR_NULL;
    if (B <= 0 || n_per_row == 0 || !(binsize > 0.f)) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_disc_logistic_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mean, logscale,
                       logscale_is_scalar ? 1 : 0, sample, out, n_per_row, binsize);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// backward of a plain conv (what TF's autodiff derives for layers.py:52-64 inside IAFLayer, tf_train.py:138):
//   forward   y = conv(a, w) + b,  a = act(concat(x, x2)),  w = exp(g) V / ||V||_o
//   given     dY (as up to 6 NCHW tensors -- the gradients of the split outputs -- times dy_scale)
//   computes  dX = [dx_residual +] act'(.) * (W^T dY)   -> NCHW, split like the forward concat
//             dV, dg (through the weight norm), db
// Passes: (1) pack dY and a pixel-major, (2) data gradient = the SAME conv kernel on the transposed packs with mirrored
// taps (EPI_DGRAD, 9 taps), (3) MFMA weight gradient over pixel ranges + reduce, (4) weight-norm backward.
// ---------------------------------------------------------------------------------------------
extern "C" int iaf_conv3x3_set_training(iaf_conv3x3_t* c, int on) {
    if (!c) return IAF_ERR_NULL;
    if (c->generic || c->mask_mode) return IAF_ERR_UNSUPPORTED;
    if (!on) { c->training = false; return IAF_OK; }
    GemmLayer& L = c->L;
    if (!L.wpt) HIP_TRY(hipMalloc(&L.wpt, (size_t)L.nchunk * MAXTAPS * L.ncot * 256 * sizeof(float)));
    GemmLayer& T = c->T;
    T = GemmLayer();
    T.cin = L.cout; T.cout = L.cin; T.nchunk = L.ncot; T.ncot = L.nchunk; T.zerodiag = 0; T.npair = 1; T.full3x3 = true;
    T.wp = L.wpt; T.nt = 1; T.pxt = 4; T.wco = 1; T.ks = 1; T.user_tuned = false;
    c->training = true;
    c->prepared = false;      // the transposed pack is written by the next prepare
    return IAF_OK;
}

struct ConvTrainWs { float* xe; float* dyc; float* part; float* dW; float* dbp; unsigned short* tapmask; };
static size_t conv3x3_train_ws_floats(const iaf_conv3x3* c, long long P, ConvTrainWs* o, float* base) {
    size_t off = 0;
    auto take = [&](size_t n) { float* q = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return q; };
    ConvTrainWs t;
    t.xe = take((size_t)P * c->n_in);
    t.dyc = take((size_t)P * c->n_out);
    t.part = take((size_t)16 * MAXTAPS * c->n_in * c->n_out);
    t.dW = take((size_t)MAXTAPS * c->n_in * c->n_out);
    t.dbp = take((size_t)256 * c->n_out);
    t.tapmask = (unsigned short*)take(((size_t)P + 1) / 2);
    if (o) *o = t;
    return off;
}

extern "C" size_t iaf_conv3x3_train_workspace_bytes(const iaf_conv3x3_t* c, int B, int H, int W) {
    if (!c || B <= 0 || H <= 0 || W <= 0) return 0;
    return conv3x3_train_ws_floats(c, (long long)B * H * W, nullptr, nullptr) * sizeof(float);
}

static int pack_pixmajor(const float* const* src, const int* chans, int n, float* dst, int C, int HW, int P, float scale,
                         int elu, hipStream_t st) {
    PackP p;
    memset(&p, 0, sizeof(p));
    int tot = 0;
    for (int k = 0; k < n; ++k) { tot += chans[k]; p.src[k] = src[k]; p.end[k] = tot; }
    p.nsrc = n; p.dst = dst; p.C = C; p.HW = HW; p.P = P; p.scale = scale; p.elu = elu;
    hipLaunchKernelGGL(iaf_pack_pixmajor_kernel, dim3((P + 63) / 64, (C + 15) / 16), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}

extern "C" int iaf_conv3x3_backward(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                                    const float* const* dys, const int* dy_channels, int n_dys, float dy_scale,
                                    float* const* dxs, const int* dx_channels, int n_dxs, const float* dx_residual,
                                    const float* V, const float* g, float* dV, float* dg, float* db, int B, int H, int W,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!c || !x || !dys || !dy_channels || !V || !g || !dV || !dg || !db || !workspace) return IAF_ERR_NULL;
    if (!c->training || !c->prepared) return IAF_ERR_NOT_PREPARED;
    if (B <= 0 || H <= 0 || W <= 0 || n_dys < 1 || n_dys > MAXSPLIT || n_dxs < 0 || n_dxs > MAXSPLIT) return IAF_ERR_SHAPE;
    if ((long long)B * H * W > (1LL << 30) / 64) return IAF_ERR_SHAPE;
    if (x2 && (c_split <= 0 || c_split >= c->n_in || (c_split & 3))) return IAF_ERR_SHAPE;
    if (n_dxs && (!dxs || !dx_channels)) return IAF_ERR_NULL;
    if (dx_residual && n_dxs != 1) return IAF_ERR_SHAPE;
    int tot = 0;
    for (int k = 0; k < n_dys; ++k) {
        if (!dys[k]) return IAF_ERR_NULL;
        if (dy_channels[k] <= 0 || (dy_channels[k] & 3)) return IAF_ERR_SHAPE;
        tot += dy_channels[k];
    }
    if (tot != c->n_out) return IAF_ERR_SHAPE;
    int dends[MAXSPLIT];
    tot = 0;
    for (int k = 0; k < n_dxs; ++k) {
        if (!dxs[k]) return IAF_ERR_NULL;
        if (dx_channels[k] <= 0 || (dx_channels[k] & 3)) return IAF_ERR_SHAPE;
        tot += dx_channels[k];
        dends[k] = tot;
    }
    if (n_dxs && tot != c->n_in) return IAF_ERR_SHAPE;
    if (((uintptr_t)workspace & 15) != 0 || workspace_bytes < iaf_conv3x3_train_workspace_bytes(c, B, H, W)) return IAF_ERR_WORKSPACE;
    const int P = B * H * W, HW = H * W;
    ConvTrainWs tw;
    conv3x3_train_ws_floats(c, P, &tw, (float*)workspace);
    hipStream_t st = (hipStream_t)stream;
    GemmLayer& L = c->L;
    int rc;
    // (1) operands, pixel-major
    if ((rc = pack_pixmajor(dys, dy_channels, n_dys, tw.dyc, c->n_out, HW, P, dy_scale, 0, st))) return rc;
    {
        const float* xs[2] = {x, x2};
        const int xc[2] = {x2 ? c_split : c->n_in, c->n_in - c_split};
        if ((rc = pack_pixmajor(xs, xc, x2 ? 2 : 1, tw.xe, c->n_in, HW, P, 1.0f, elu_input ? 1 : 0, st))) return rc;
    }
    // (2) data gradient
    if (n_dxs) {
        ConvP p;
        memset(&p, 0, sizeof(p));
        p.B = B; p.H = H; p.W = W; p.HW = HW; p.P = P;
        p.x = tw.dyc; p.mode = MODE_DGRAD_PLAIN;
        p.zin = elu_input ? tw.xe : nullptr;
        p.res = dx_residual;
        p.nsplit = n_dxs;
        for (int k = 0; k < MAXSPLIT; ++k) {
            p.split_end[k] = dends[k < n_dxs ? k : n_dxs - 1];
            p.split_ptr[k] = dxs[k < n_dxs ? k : n_dxs - 1];
        }
        if ((rc = conv3x3_launch(c->T, p, EPI_DGRAD9, IN_PIXMAJOR, false, true, st))) return rc;
    }
    // (3) weight gradient: partials over pixel ranges, then reduce (+ column sums of dY for db)
    if ((rc = launch_tapmask(tw.tapmask, B, H, W, st))) return rc;
    if ((rc = launch_wgrad(nullptr, L, tw.xe, tw.dyc, tw.part, tw.tapmask, B, H, W, st))) return rc;
    const int nslab = (P + 31) / 32 < 256 ? (P + 31) / 32 : 256;
    const int px_per_slab = (P + nslab - 1) / nslab;
    {
        const size_t n4 = (size_t)MAXTAPS * L.cin * L.cout / 4;
        int nblk = (int)((n4 + 255) / 256);
        if (nblk > 1024) nblk = 1024;
        hipLaunchKernelGGL(iaf_wgrad_reduce_kernel, dim3(nblk + nslab), dim3(256), 0, st, tw.part, tw.dW,
                           wgrad_nrange(P, L.cin, MAXTAPS), n4, nblk, (const float*)tw.dyc, tw.dbp, P, L.cout, px_per_slab);
    }
    // (4) through the weight norm
    WnBwdLayer w;
    memset(&w, 0, sizeof(w));
    w.V = V; w.g = g; w.dW = tw.dW; w.dbp = tw.dbp; w.dV = dV; w.dg = dg; w.db = db;
    w.cin = L.cin; w.cout = L.cout; w.cout_packed = L.cout; w.nslab = nslab; w.pack_stride = 1;
    hipLaunchKernelGGL(iaf_wn_bwd_plain_kernel, dim3(L.cout / 16), dim3(256), 0, st, w);
    return (int)hipGetLastError();
}
