// iaf_model_edge.hpp -- the two ends of the model around the IAFLayer stack, CVAE1._forward (tf_train.py:150-218): the image
// scaling (:153-154, 159), conv2d("x_enc", x, h_size, [5,5], [2,2]) (:183), the tiled h_top (:189-192), deconv2d("x_dec", elu(h), 3,
// [5,5]) + clip (:206-208) and the two scalar sums obj / loss (:211, 218).  Tiny channel counts on one side (3 image channels): not
// MFMA work -- direct convolutions (a few output channels per thread; the deconv by output phases with its input channels
// split over 16 thread slices), latency-bound, 10-20 microseconds each at the BASELINE batch.
// Part of the single translation unit iaf_engine.hip (included there; not a standalone header).
#pragma once
#include "iaf_conv_epilogue.hpp"     // fast_divmod

// out[(b k + s)][i] = clip((x[b][i] + 0.5) / 256, 0, 1) - 0.5  for s < k   (tf_train.py:153-154, repeat :159)
__global__ __launch_bounds__(256) void iaf_image_to_float_kernel(const unsigned char* __restrict__ x, float* __restrict__ out,
                                                                size_t n_per_image, size_t total, int k) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t row = e / n_per_image, i = e - row * n_per_image;
        const float v = ((float)x[(row / k) * n_per_image + i] + 0.5f) / 256.0f;
        out[e] = fminf(fmaxf(v, 0.0f), 1.0f) - 0.5f;
    }
}

// weight norm of a k x k filter (layers.py:56-60 conv2d, :104-106 deconv2d).  One workgroup per normalised channel:
//   conv    V [kh,kw,n_in,n_out]: w[.,.,.,o] = exp(g[o]) V[.,.,.,o] / ||V[.,.,.,o]||           (norm over kh,kw,n_in)
//   deconv  V [kh,kw,n_out,n_in]: w[.,.,o,c] = exp(g[o]) V[.,.,o,c] / ||V[.,.,.,c]||           (norm over kh,kw,n_OUT per input channel)
__global__ __launch_bounds__(256) void iaf_convk_weightnorm_kernel(const float* __restrict__ V, const float* __restrict__ g,
                                                                  float* __restrict__ w, int taps, int n_a, int n_b, int deconv) {
    // V is [taps][n_a][n_b]; the normalised channel is the LAST axis in both layouts (n_out for conv, n_in for deconv)
    __shared__ float red[256];
    const int ch = blockIdx.x;
    float ss = 0.f;
    for (int e = threadIdx.x; e < taps * n_a; e += 256) {
        const float v = V[(size_t)e * n_b + ch];
        ss += v * v;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float inv = 1.0f / sqrtf(fmaxf(red[0], 1e-12f));
    for (int e = threadIdx.x; e < taps * n_a; e += 256) {
        const int a = e % n_a;                                   // deconv: the output channel the gain belongs to
        const float gain = __expf(deconv ? g[a] : g[ch]);
        w[(size_t)e * n_b + ch] = gain * V[(size_t)e * n_b + ch] * inv;
    }
}

struct ConvKP {
    const float* x; const float* w; const float* b; float* y;
    int B, n_in, H, W, n_out, kh, kw, stride, OH, OW, pad_t, pad_l, elu;
    float clip_lo, clip_hi;
};

// y[b,o,oy,ox] = b[o] + sum_{a,c,ci} [elu](x[b,ci,oy s + a - pad_t, ox s + c - pad_l]) w[a,c,ci,o]   (tf.nn.conv2d SAME, NCHW)
// A thread owns one output pixel and CK_NO consecutive output channels (blockIdx.y = channel chunk): every x value it loads feeds
// CK_NO accumulators, the weights of a wave are uniform.  Out-of-image taps are multiplied by zero instead of branched around
// (clamped address), so the (column, channel) loop is straight-line code the compiler batches the loads of.
#define CK_NO 4
__global__ __launch_bounds__(256) void iaf_convk_forward_kernel(ConvKP p) {
    const unsigned npx = (unsigned)p.B * p.OH * p.OW;            // < 2^30 (checked by the host): 32-bit index arithmetic
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npx) return;
    const unsigned erow = e / (unsigned)p.OW;
    const int ox = (int)(e - erow * p.OW), bimg = (int)(erow / (unsigned)p.OH), oy = (int)(erow - (unsigned)bimg * p.OH);
    const int o0 = (int)blockIdx.y * CK_NO;
    float acc[CK_NO];
#pragma unroll
    for (int q = 0; q < CK_NO; ++q) acc[q] = (o0 + q < p.n_out) ? p.b[o0 + q] : 0.f;
    const size_t HW = (size_t)p.H * p.W;
    const int nt = p.kw * p.n_in;
    for (int a = 0; a < p.kh; ++a) {
        const int iy = oy * p.stride + a - p.pad_t;
        if (iy < 0 || iy >= p.H) continue;                       // (uniform over a row of output pixels)
        const float* xrow = p.x + ((size_t)bimg * p.n_in * p.H + iy) * p.W;
        // the weights of this row of taps are the same for every thread: a wave-uniform base (scalar loads)
        const float* wrow = p.w + __builtin_amdgcn_readfirstlane((a * nt) * p.n_out + o0);
        for (int c = 0; c < p.kw; ++c) {                         // (no integer divisions in here: ~40 instructions each on this ISA)
            const int ix = ox * p.stride + c - p.pad_l;
            const bool in = ix >= 0 && ix < p.W;
            const float* xs = xrow + (in ? ix : 0);
            const float* wc = wrow + c * p.n_in * p.n_out;
            if ((p.n_out & 3) == 0) {                             // four weights = one 16-byte load (o0 and n_out multiples of 4)
#pragma unroll 3
                for (int ci = 0; ci < p.n_in; ++ci) {
                    float v = xs[(size_t)ci * HW];
                    if (p.elu) v = elu_f(v);
                    v = in ? v : 0.f;
                    const f32x4 w4 = *(const f32x4*)(wc + ci * p.n_out);
#pragma unroll
                    for (int q = 0; q < CK_NO; ++q) acc[q] += v * w4[q];
                }
            } else {
#pragma unroll 3
                for (int ci = 0; ci < p.n_in; ++ci) {
                    float v = xs[(size_t)ci * HW];
                    if (p.elu) v = elu_f(v);
                    v = in ? v : 0.f;
#pragma unroll
                    for (int q = 0; q < CK_NO; ++q)
                        if (o0 + q < p.n_out) acc[q] += v * wc[ci * p.n_out + q];
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < CK_NO; ++q)
        if (o0 + q < p.n_out) p.y[(((size_t)bimg * p.n_out + o0 + q) * p.OH + oy) * p.OW + ox] = acc[q];
}

// conv2d_transpose(SAME, stride s) + b [+ clip] (layers.py:67-80, 108-111; tf_train.py:207-208): y [B,n_out,H s,W s];
//   y[b,o,Y,X] = b[o] + sum over taps (a,c) with (Y + pad_t - a) = s i, (X + pad_l - c) = s j inside the input of
//                [elu](x[b,ci,i,j]) w[a,c,o,ci]          (pad_t, pad_l: the SAME padding of the forward conv it transposes)
// By output phases: blockIdx.y = (Y mod s, X mod s), so every thread of a workgroup meets the same taps.  A workgroup owns 16
// consecutive input-grid pixels x 16 slices of the input channels (thread = pixel + 16 slice): the partial sums of the slices are
// added through LDS in a fixed order.  Up to DK_NO output channels per thread (blockIdx.z = channel chunk).
#define DK_NO 4
__global__ __launch_bounds__(256) void iaf_deconvk_forward_kernel(ConvKP p) {
    __shared__ float red[DK_NO][16][17];
    extern __shared__ float wl[];                                // this phase's weights [tap of the phase][DK_NO][n_in]
    const int s = p.stride, py = (int)blockIdx.y / s, px = (int)blockIdx.y % s;
    const int o0 = (int)blockIdx.z * DK_NO;
    const unsigned npx = (unsigned)p.B * p.H * p.W;              // < 2^30 (checked by the host): 32-bit index arithmetic
    const int pl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const unsigned e = blockIdx.x * 16 + pl;
    const bool live = e < npx;
    const unsigned ec = live ? e : npx - 1;
    const unsigned erow = ec / (unsigned)p.W;
    const int j0 = (int)(ec - erow * p.W), bimg = (int)(erow / (unsigned)p.H), i0 = (int)(erow - (unsigned)bimg * p.H);
    const int per = (p.n_in + 15) / 16, c_lo = sl * per, c_hi = (c_lo + per < p.n_in) ? c_lo + per : p.n_in;
    // taps of this phase: a = a0, a0 + s, ... with a0 = (py + pad_t) mod s (Y + pad_t - a = s (i0 + (py + pad_t - a) / s), exact)
    const int a0 = (py + p.pad_t) % s, c0 = (px + p.pad_l) % s;
    const int na = (p.kh - a0 + s - 1) / s, nc = (p.kw - c0 + s - 1) / s;
    const int wn = DK_NO * p.n_in;
    for (int ta = 0; ta < na; ++ta)                              // (nested loops instead of index divisions)
        for (int tc = 0; tc < nc; ++tc)
            for (int q = 0; q < DK_NO; ++q) {
                const float* src = p.w + ((size_t)((a0 + ta * s) * p.kw + c0 + tc * s) * p.n_out + o0 + q) * p.n_in;
                float* dst = wl + ((ta * nc + tc) * DK_NO + q) * p.n_in;
                for (int ci = threadIdx.x; ci < p.n_in; ci += 256) dst[ci] = (o0 + q < p.n_out) ? src[ci] : 0.f;
            }
    __syncthreads();
    float acc[DK_NO];
#pragma unroll
    for (int q = 0; q < DK_NO; ++q) acc[q] = 0.f;
    const size_t HW = (size_t)p.H * p.W;
    for (int ta = 0; ta < na; ++ta) {
        const int a = a0 + ta * s;
        const int i = i0 + (py + p.pad_t - a) / s;
        const bool iin = i >= 0 && i < p.H;
        for (int tc = 0; tc < nc; ++tc) {
            const int c = c0 + tc * s;
            const int j = j0 + (px + p.pad_l - c) / s;
            const bool in = iin && j >= 0 && j < p.W;
            const float* xs = p.x + ((size_t)bimg * p.n_in * p.H + (in ? i : 0)) * p.W + (in ? j : 0);
            const float* ws = wl + (ta * nc + tc) * wn;
#pragma unroll 5
            for (int ci = c_lo; ci < c_hi; ++ci) {
                float t = xs[(size_t)ci * HW];
                if (p.elu) t = elu_f(t);
                t = in ? t : 0.f;
#pragma unroll
                for (int q = 0; q < DK_NO; ++q) acc[q] += t * ws[q * p.n_in + ci];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < DK_NO; ++q) red[q][sl][pl] = acc[q];
    __syncthreads();
    if (sl < DK_NO && o0 + sl < p.n_out && live) {               // thread (pixel pl, output channel sl) adds the 16 slices in order
        float r = p.b[o0 + sl];
        for (int k = 0; k < 16; ++k) r += red[sl][k][pl];
        if (p.clip_lo < p.clip_hi) r = fminf(fmaxf(r, p.clip_lo), p.clip_hi);
        p.y[(((size_t)bimg * p.n_out + o0 + sl) * p.OH + i0 * s + py) * p.OW + j0 * s + px] = r;
    }
}

// out[b,c,:,:] = v[c]: tf.tile(reshape(h_top, [1,-1,1,1]), [data_size, 1, S, S]) (tf_train.py:190-192)
__global__ __launch_bounds__(256) void iaf_tile_channels_kernel(const float* __restrict__ v, float* __restrict__ out, size_t total,
                                                               int C, int HW) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) out[e] = v[(e / HW) % C];
}

// out[0] = sum_i (a[i] + sb * b[i]) in a fixed order (one workgroup): obj = reduce_sum(kl_obj - log_pxz) (tf_train.py:211),
// loss = reduce_sum(compute_lowerbound(...)) (:218)
__global__ __launch_bounds__(256) void iaf_sum_axpy_kernel(const float* __restrict__ a, const float* __restrict__ b, float sb,
                                                          float* __restrict__ out, int n) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += a[i] + (b ? sb * b[i] : 0.f);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// ---- C ABI -------------------------------------------------------------------------------------------------------------------
extern "C" int iaf_image_to_float(const unsigned char* x, float* out, int B, size_t n_per_image, int k, void* stream) {
    if (!x || !out) return IAF_ERR_NULL;
    if (B <= 0 || k <= 0 || n_per_image == 0) return IAF_ERR_SHAPE;
    const size_t total = (size_t)B * k * n_per_image;
    hipLaunchKernelGGL(iaf_image_to_float_kernel, ew_grid(total), dim3(256), 0, (hipStream_t)stream, x, out, n_per_image, total, k);
    return (int)hipGetLastError();
}

extern "C" int iaf_convk_weightnorm(const float* V, const float* g, float* w, int kh, int kw, int n_in, int n_out, int deconv,
                                    void* stream) {
    if (!V || !g || !w) return IAF_ERR_NULL;
    if (kh <= 0 || kw <= 0 || n_in <= 0 || n_out <= 0) return IAF_ERR_SHAPE;
    // conv: V [kh,kw,n_in,n_out] -> one workgroup per n_out; deconv: V [kh,kw,n_out,n_in] -> one per n_in
    hipLaunchKernelGGL(iaf_convk_weightnorm_kernel, dim3(deconv ? n_in : n_out), dim3(256), 0, (hipStream_t)stream, V, g, w, kh * kw,
                       deconv ? n_out : n_in, deconv ? n_in : n_out, deconv ? 1 : 0);
    return (int)hipGetLastError();
}

// TF "SAME": out = ceil(n / s), total padding max((out - 1) s + k - n, 0), the smaller half first
static void same_pad(int n, int k, int s, int* out, int* before) {
    *out = (n + s - 1) / s;
    int tot = (*out - 1) * s + k - n;
    if (tot < 0) tot = 0;
    *before = tot / 2;
}

extern "C" int iaf_convk_forward(const float* x, const float* w, const float* b, float* y, int B, int n_in, int H, int W, int n_out,
                                 int kh, int kw, int stride, int elu_input, void* stream) {
    if (!x || !w || !b || !y) return IAF_ERR_NULL;
    if (B <= 0 || n_in <= 0 || n_out <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return IAF_ERR_SHAPE;
    ConvKP p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.b = b; p.y = y; p.B = B; p.n_in = n_in; p.H = H; p.W = W; p.n_out = n_out; p.kh = kh; p.kw = kw;
    p.stride = stride; p.elu = elu_input ? 1 : 0;
    same_pad(H, kh, stride, &p.OH, &p.pad_t);
    same_pad(W, kw, stride, &p.OW, &p.pad_l);
    const size_t npx = (size_t)B * p.OH * p.OW;
    if (npx > (1u << 30)) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_convk_forward_kernel, dim3((unsigned)((npx + 255) / 256), (n_out + CK_NO - 1) / CK_NO), dim3(256), 0,
                       (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int iaf_deconvk_forward(const float* x, const float* w, const float* b, float* y, int B, int n_in, int H, int W, int n_out,
                                   int kh, int kw, int stride, int elu_input, float clip_lo, float clip_hi, void* stream) {
    if (!x || !w || !b || !y) return IAF_ERR_NULL;
    if (B <= 0 || n_in <= 0 || n_out <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return IAF_ERR_SHAPE;
    ConvKP p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.b = b; p.y = y; p.B = B; p.n_in = n_in; p.H = H; p.W = W; p.n_out = n_out; p.kh = kh; p.kw = kw;
    p.stride = stride; p.elu = elu_input ? 1 : 0; p.clip_lo = clip_lo; p.clip_hi = clip_hi;
    p.OH = H * stride; p.OW = W * stride;
    int o;
    same_pad(p.OH, kh, stride, &o, &p.pad_t);       // the forward conv this transposes maps [OH, OW] -> [H, W]
    same_pad(p.OW, kw, stride, &o, &p.pad_l);
    const size_t npx = (size_t)B * H * W;
    if (npx > (1u << 30) || stride > 8) return IAF_ERR_SHAPE;
    // dynamic LDS: the weights of one output phase, at most ceil(kh/s) ceil(kw/s) taps x DK_NO channels x n_in
    const size_t wl_bytes = (size_t)((kh + stride - 1) / stride) * ((kw + stride - 1) / stride) * DK_NO * n_in * sizeof(float);
    if (wl_bytes > 60 * 1024) return IAF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(iaf_deconvk_forward_kernel, dim3((unsigned)((npx + 15) / 16), stride * stride, (n_out + DK_NO - 1) / DK_NO),
                       dim3(256), wl_bytes, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int iaf_tile_channels(const float* v, float* out, int B, int C, int HW, void* stream) {
    if (!v || !out) return IAF_ERR_NULL;
    if (B <= 0 || C <= 0 || HW <= 0) return IAF_ERR_SHAPE;
    const size_t total = (size_t)B * C * HW;
    hipLaunchKernelGGL(iaf_tile_channels_kernel, ew_grid(total), dim3(256), 0, (hipStream_t)stream, v, out, total, C, HW);
    return (int)hipGetLastError();
}

extern "C" int iaf_sum_axpy(const float* a, const float* b, float sb, float* out, int n, void* stream) {
    if (!a || !out) return IAF_ERR_NULL;
    if (n <= 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_sum_axpy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, b, sb, out, n);
    return (int)hipGetLastError();
}

// ==================================================================================================================================
// Backward of the two ends: what TF autodiff derives for tf_train.py:183, 189-192, 206-211 in `opt.compute_gradients(obj)` (:128)
// ==================================================================================================================================
// d log_pxz / d mean and / d logscale of discretized_logistic (distributions.py:28-32), times `up` (= d obj / d log_pxz = -1,
// tf_train.py:211), with clip_by_value's gradient folded in (:208: the gradient passes where the value was not clipped).
//   s = (floor(x / b) b - mean) / scale,  t = s + b / scale,  P = sig(t) - sig(s) + 1e-7,  logp = log P
//   d logp / d mean = -(sig'(t) - sig'(s)) / (scale P);   d logp / d logscale = (-t sig'(t) + s sig'(s)) / P
// One workgroup per row; d_logscale_rows[b] = up * sum over the row (summed over rows by iaf_sum_axpy).
__global__ __launch_bounds__(256) void iaf_discretized_logistic_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ logscale,
                                                                          const float* __restrict__ sample, float up, float clip_lo,
                                                                          float clip_hi, float* __restrict__ d_mean,
                                                                          float* __restrict__ d_logscale_rows, size_t n_per_row,
                                                                          float binsize) {
    __shared__ float red[256];
    const float scale = __expf(logscale[0]), inv = 1.0f / scale;
    const size_t base = (size_t)blockIdx.x * n_per_row;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n_per_row; i += 256) {
        const float m = mean[base + i];
        const float s = (floorf(sample[base + i] / binsize) * binsize - m) * inv, t = s + binsize * inv;
        const float ss = 1.0f / (1.0f + __expf(-s)), st = 1.0f / (1.0f + __expf(-t));
        const float P = st - ss + 1e-7f;
        const float ds = ss * (1.0f - ss), dt = st * (1.0f - st);
        const bool pass = !(clip_lo < clip_hi) || (m > clip_lo && m < clip_hi);
        d_mean[base + i] = pass ? up * (-(dt - ds) * inv / P) : 0.f;
        acc += (-t * dt + s * ds) / P;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) d_logscale_rows[blockIdx.x] = up * red[0];
}

// dW[a][c][ci][o] = sum_{b,oy,ox} X[b,ci,oy s + a - pad_t, ox s + c - pad_l] DY[b,o,oy,ox]  -- the weight gradient of a strided SAME conv
// with few channels (n_small) on its large-grid side: x_enc (X = the image, DY = d h) and, with the roles of data and gradient
// exchanged, x_dec (X = d x_out, DY = elu(h): its filter [kh,kw,3,h] has the same layout).  Workgroup = (tap, ci, 16 channels o);
// thread = (o, one of 16 pixel lanes); the lanes' partial sums are added by shuffles in a fixed order.
struct ConvKWP {
    const float* x; const float* dy; float* dW;
    int B, n_small, H, W, n_big, OH, OW, kh, kw, stride, pad_t, pad_l, elu_x, elu_dy;
};
__global__ __launch_bounds__(256) void iaf_convk_wgrad_kernel(ConvKWP p) {
    const int tapci = blockIdx.x, tap = tapci / p.n_small, ci = tapci - tap * p.n_small;
    const int a = tap / p.kw, c = tap - a * p.kw;
    const int o = blockIdx.y * 16 + (threadIdx.x >> 4), ln = threadIdx.x & 15;
    const int OHW = p.OH * p.OW;
    const bool olive = o < p.n_big;
    const int oc = olive ? o : p.n_big - 1;
    float acc = 0.f;
    // rows of the small grid in the outer loops, its columns over the 16 lanes: no index divisions in the loop
    for (int b = 0; b < p.B; ++b) {
        const float* xb = p.x + ((size_t)b * p.n_small + ci) * p.H * p.W;
        const float* db = p.dy + ((size_t)b * p.n_big + oc) * OHW;
        for (int ox = ln; ox < p.OW; ox += 16) {
            const int ix = ox * p.stride + c - p.pad_l;
            const bool cin_ = ix >= 0 && ix < p.W;
#pragma unroll 8
            for (int oy = 0; oy < p.OH; ++oy) {                   // branch-free: the loads of eight rows are in flight together
                const int iy = oy * p.stride + a - p.pad_t;
                const bool in = cin_ && iy >= 0 && iy < p.H;
                float xv = xb[in ? (size_t)iy * p.W + ix : 0];
                if (p.elu_x) xv = elu_f(xv);
                xv = in ? xv : 0.f;
                float dv = db[oy * p.OW + ox];
                if (p.elu_dy) dv = elu_f(dv);
                acc += xv * dv;
            }
        }
    }
    acc += __shfl_xor(acc, 8, 16);
    acc += __shfl_xor(acc, 4, 16);
    acc += __shfl_xor(acc, 2, 16);
    acc += __shfl_xor(acc, 1, 16);
    if (ln == 0 && olive) p.dW[(size_t)tapci * p.n_big + o] = acc;
}

// Through the weight norm (layers.py:56-60 / 104-106), one workgroup per normalised channel `ch` (the last axis of V [taps][n_a][n_b]):
//   conv    w = e_ch V / n_ch:      dg[ch] = e_ch (dW . V) / n;   dV = (e_ch / n) (dW - V (dW . V) / n^2)
//   deconv  w = e_a V / n_ch (a = the n_a index): S = sum e_a dW V;  dV = e_a dW / n - V S / n^3;  dg[a] = sum_ch (partial[ch][a] = sum_taps dW w)
__global__ __launch_bounds__(256) void iaf_convk_weightnorm_bwd_kernel(const float* __restrict__ V, const float* __restrict__ g,
                                                                      const float* __restrict__ dW, float* __restrict__ dV,
                                                                      float* __restrict__ dg_or_partial, int taps, int n_a, int n_b,
                                                                      int deconv) {
    __shared__ float red[2][256];
    const int ch = blockIdx.x;
    float ss = 0.f, dot = 0.f;
    for (int e = threadIdx.x; e < taps * n_a; e += 256) {
        const float v = V[(size_t)e * n_b + ch], d = dW[(size_t)e * n_b + ch];
        const float gain = deconv ? __expf(g[e % n_a]) : 1.0f;
        ss += v * v;
        dot += d * gain * v;
    }
    red[0][threadIdx.x] = ss; red[1][threadIdx.x] = dot;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { red[0][threadIdx.x] += red[0][threadIdx.x + k]; red[1][threadIdx.x] += red[1][threadIdx.x + k]; }
        __syncthreads();
    }
    const float s2 = fmaxf(red[0][0], 1e-12f), n = sqrtf(s2), S = red[1][0];
    const float e_ch = deconv ? 1.0f : __expf(g[ch]);
    for (int e = threadIdx.x; e < taps * n_a; e += 256) {
        const float v = V[(size_t)e * n_b + ch], d = dW[(size_t)e * n_b + ch];
        const float gain = deconv ? __expf(g[e % n_a]) : e_ch;
        dV[(size_t)e * n_b + ch] = deconv ? gain * d / n - v * S / (n * s2) : (gain / n) * (d - v * S / s2);
    }
    if (!deconv) {
        if (threadIdx.x == 0) dg_or_partial[ch] = e_ch * S / n;
    } else {
        __syncthreads();
        for (int a = 0; a < n_a; ++a) {                             // (n_a = the few output channels of x_dec)
            float t = 0.f;
            for (int tp = threadIdx.x; tp < taps; tp += 256) {
                const size_t e = (size_t)tp * n_a + a;
                t += dW[e * n_b + ch] * __expf(g[a]) * V[e * n_b + ch] / n;
            }
            red[0][threadIdx.x] = t;
            __syncthreads();
            for (int k = 128; k > 0; k >>= 1) {
                if ((int)threadIdx.x < k) red[0][threadIdx.x] += red[0][threadIdx.x + k];
                __syncthreads();
            }
            if (threadIdx.x == 0) dg_or_partial[(size_t)ch * n_a + a] = red[0][0];
            __syncthreads();
        }
    }
}

// out[c] = sum_{b,p} x[b,c,p]: a conv's bias gradient, and d h_top (the adjoint of tf.tile, tf_train.py:190-192)
__global__ __launch_bounds__(256) void iaf_channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C, int HW) {
    __shared__ float red[256];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b)
        for (int i = threadIdx.x; i < HW; i += 256) s += x[((size_t)b * C + c) * HW + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
}

// out = g * elu'(h), elu'(h) = h > 0 ? 1 : exp(h)  (the ELU in front of x_dec, tf_train.py:206)
__global__ __launch_bounds__(256) void iaf_mul_elu_grad_kernel(const float* __restrict__ g, const float* __restrict__ h,
                                                              float* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = g[i] * (h[i] > 0.f ? 1.0f : __expf(h[i]));
}

extern "C" int iaf_discretized_logistic_backward(const float* mean, const float* logscale, const float* sample, float up, float clip_lo,
                                                 float clip_hi, float* d_mean, float* d_logscale_rows, int B, size_t n_per_row,
                                                 float binsize, void* stream) {
    if (!mean || !logscale || !sample || !d_mean || !d_logscale_rows) return IAF_ERR_NULL;
    if (B <= 0 || n_per_row == 0 || !(binsize > 0.f)) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_discretized_logistic_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mean, logscale, sample, up, clip_lo,
                       clip_hi, d_mean, d_logscale_rows, n_per_row, binsize);
    return (int)hipGetLastError();
}

extern "C" int iaf_convk_wgrad(const float* x, const float* dy, float* dW, int B, int n_small, int H, int W, int n_big, int kh, int kw,
                               int stride, int elu_x, int elu_dy, void* stream) {
    if (!x || !dy || !dW) return IAF_ERR_NULL;
    if (B <= 0 || n_small <= 0 || n_big <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return IAF_ERR_SHAPE;
    ConvKWP p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.dy = dy; p.dW = dW; p.B = B; p.n_small = n_small; p.H = H; p.W = W; p.n_big = n_big; p.kh = kh; p.kw = kw;
    p.stride = stride; p.elu_x = elu_x ? 1 : 0; p.elu_dy = elu_dy ? 1 : 0;
    same_pad(H, kh, stride, &p.OH, &p.pad_t);
    same_pad(W, kw, stride, &p.OW, &p.pad_l);
    if ((long long)B * p.OH * p.OW > (1LL << 24)) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_convk_wgrad_kernel, dim3(kh * kw * n_small, (n_big + 15) / 16), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// dV (V's layout), dg [n_out]; deconv: scratch = n_in * n_out floats
extern "C" int iaf_convk_weightnorm_backward(const float* V, const float* g, const float* dW, float* dV, float* dg, float* scratch,
                                             int kh, int kw, int n_in, int n_out, int deconv, void* stream) {
    if (!V || !g || !dW || !dV || !dg || (deconv && !scratch)) return IAF_ERR_NULL;
    if (kh <= 0 || kw <= 0 || n_in <= 0 || n_out <= 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_convk_weightnorm_bwd_kernel, dim3(deconv ? n_in : n_out), dim3(256), 0, (hipStream_t)stream, V, g, dW, dV,
                       deconv ? scratch : dg, kh * kw, deconv ? n_out : n_in, deconv ? n_in : n_out, deconv ? 1 : 0);
    HIP_TRY(hipGetLastError());
    if (deconv) return iaf_colsum(scratch, dg, n_in, n_out, stream);
    return IAF_OK;
}

extern "C" int iaf_channel_sum(const float* x, float* out, int B, int C, int HW, void* stream) {
    if (!x || !out) return IAF_ERR_NULL;
    if (B <= 0 || C <= 0 || HW <= 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_channel_sum_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, out, B, C, HW);
    return (int)hipGetLastError();
}

extern "C" int iaf_mul_elu_grad(const float* g, const float* h, float* out, size_t n, void* stream) {
    if (!g || !h || !out) return IAF_ERR_NULL;
    if (n == 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_mul_elu_grad_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, g, h, out, n);
    return (int)hipGetLastError();
}

// ---- elementwise pieces of the data-dependent init pass (mode "init", tf_train.py:60-61, 70-71, 94, 208) -----------------------------
__global__ __launch_bounds__(256) void iaf_axpby_kernel(const float* __restrict__ a, float sa, const float* __restrict__ b, float sb,
                                                       float* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = sa * a[i] + sb * b[i];
}
// z' = (z - scale m) / exp(scale s): the IAF update from the raw outputs of ar_multiconv2d (tf_train.py:70-71 with scale = 0.1)
__global__ __launch_bounds__(256) void iaf_affine_kernel(const float* __restrict__ z, const float* __restrict__ m, const float* __restrict__ s,
                                                        float scale, float* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (z[i] - scale * m[i]) * __expf(-scale * s[i]);
}
__global__ __launch_bounds__(256) void iaf_clip_kernel(const float* __restrict__ x, float lo, float hi, float* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = fminf(fmaxf(x[i], lo), hi);
}
extern "C" int iaf_axpby(const float* a, float sa, const float* b, float sb, float* out, size_t n, void* stream) {
    if (!a || !b || !out) return IAF_ERR_NULL;
    if (n == 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_axpby_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, a, sa, b, sb, out, n);
    return (int)hipGetLastError();
}
extern "C" int iaf_affine_transform(const float* z, const float* m, const float* s, float scale, float* out, size_t n, void* stream) {
    if (!z || !m || !s || !out) return IAF_ERR_NULL;
    if (n == 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_affine_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, z, m, s, scale, out, n);
    return (int)hipGetLastError();
}
extern "C" int iaf_clip(const float* x, float lo, float hi, float* out, size_t n, void* stream) {
    if (!x || !out) return IAF_ERR_NULL;
    if (n == 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_clip_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, x, lo, hi, out, n);
    return (int)hipGetLastError();
}
