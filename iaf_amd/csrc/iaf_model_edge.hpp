// iaf_model_edge.hpp -- the two ends of the model around the IAFLayer stack, CVAE1._forward (tf_train.py:150-218): the image
// scaling (:153-154, 159), conv2d("x_enc", x, h_size, [5,5], [2,2]) (:183), the tiled h_top (:189-192), deconv2d("x_dec", elu(h), 3,
// [5,5]) + clip (:206-208) and the two scalar sums obj / loss (:211, 218).  Tiny channel counts on one side (3 image channels): not
// MFMA work -- direct convolutions, one output element per thread, HBM/L2-bound and a few microseconds each at the BASELINE batch.
// Part of the single translation unit iaf_engine.hip (included there; not a standalone header).
#pragma once

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
__global__ __launch_bounds__(256) void iaf_convk_forward_kernel(ConvKP p) {
    const size_t total = (size_t)p.B * p.n_out * p.OH * p.OW;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int ox = (int)(e % p.OW), oy = (int)((e / p.OW) % p.OH), o = (int)((e / ((size_t)p.OW * p.OH)) % p.n_out);
        const int bimg = (int)(e / ((size_t)p.OW * p.OH * p.n_out));
        float acc = p.b[o];
        for (int a = 0; a < p.kh; ++a) {
            const int iy = oy * p.stride + a - p.pad_t;
            if (iy < 0 || iy >= p.H) continue;
            for (int c = 0; c < p.kw; ++c) {
                const int ix = ox * p.stride + c - p.pad_l;
                if (ix < 0 || ix >= p.W) continue;
                const float* xs = p.x + ((size_t)bimg * p.n_in * p.H + iy) * p.W + ix;
                const float* ws = p.w + ((size_t)(a * p.kw + c) * p.n_in) * p.n_out + o;
                for (int ci = 0; ci < p.n_in; ++ci) {
                    float v = xs[(size_t)ci * p.H * p.W];
                    if (p.elu) v = elu_f(v);
                    acc += v * ws[(size_t)ci * p.n_out];
                }
            }
        }
        p.y[e] = acc;
    }
}

// conv2d_transpose(SAME, stride s) + b [+ clip] (layers.py:67-80, 108-111; tf_train.py:207-208): y [B,n_out,H s,W s];
//   y[b,o,Y,X] = b[o] + sum over taps (a,c) with (Y + pad_t - a) = s i, (X + pad_l - c) = s j inside the input of
//                [elu](x[b,ci,i,j]) w[a,c,o,ci]          (pad_t, pad_l: the SAME padding of the forward conv it transposes)
__global__ __launch_bounds__(256) void iaf_deconvk_forward_kernel(ConvKP p) {
    const size_t total = (size_t)p.B * p.n_out * p.OH * p.OW;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int X = (int)(e % p.OW), Y = (int)((e / p.OW) % p.OH), o = (int)((e / ((size_t)p.OW * p.OH)) % p.n_out);
        const int bimg = (int)(e / ((size_t)p.OW * p.OH * p.n_out));
        float acc = p.b[o];
        for (int a = 0; a < p.kh; ++a) {
            const int ty = Y + p.pad_t - a;
            if (ty < 0 || ty % p.stride != 0 || ty / p.stride >= p.H) continue;
            const int i = ty / p.stride;
            for (int c = 0; c < p.kw; ++c) {
                const int tx = X + p.pad_l - c;
                if (tx < 0 || tx % p.stride != 0 || tx / p.stride >= p.W) continue;
                const int j = tx / p.stride;
                const float* xs = p.x + ((size_t)bimg * p.n_in * p.H + i) * p.W + j;
                const float* ws = p.w + ((size_t)(a * p.kw + c) * p.n_out + o) * p.n_in;
                for (int ci = 0; ci < p.n_in; ++ci) {
                    float v = xs[(size_t)ci * p.H * p.W];
                    if (p.elu) v = elu_f(v);
                    acc += v * ws[ci];
                }
            }
        }
        if (p.clip_lo < p.clip_hi) acc = fminf(fmaxf(acc, p.clip_lo), p.clip_hi);
        p.y[e] = acc;
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
    hipLaunchKernelGGL(iaf_convk_forward_kernel, ew_grid((size_t)B * n_out * p.OH * p.OW), dim3(256), 0, (hipStream_t)stream, p);
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
    hipLaunchKernelGGL(iaf_deconvk_forward_kernel, ew_grid((size_t)B * n_out * p.OH * p.OW), dim3(256), 0, (hipStream_t)stream, p);
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
