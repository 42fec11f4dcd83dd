// iaf_kernels_resample.hpp -- what the DOWNSAMPLING IAFLayer needs on top of the stride-1 convs (tf_train.py:33,42-43,
// 89-91): 2x resampling in both directions and the weight prep of deconv2d (tf_utils/layers.py:67-112, 169-175).
// Part of the single translation unit iaf_engine.hip (included there, in order; not a standalone header).
//
// The two strided ops are expressed through the stride-1 3x3 conv kernel:
//   conv2d(stride 2, SAME)      out[i,j] = c[2i+1, 2j+1] of the stride-1 SAME conv c (TF pads bottom/right only when
//                               (H-1)*2+3-H = 1 is odd, so window (i,j) is centred on input (2i+1, 2j+1))
//   deconv2d(stride 2, SAME)    = stride-1 SAME cross-correlation of the zero-inserted input u[2i+1,2j+1] = x[i,j] with
//                               the 180-degree-rotated filter:  out[y,x,o] = sum x[(y-a)/2,(x-b)/2,c] f[a,b,o,c]
// Both cost 4x the minimal MFMA work, for ONE layer per resolution change (1 of 20 at the BASELINE config).
#pragma once

#define IAF_RESAMPLE_DOWN_EVEN 0   // dst[i,j] = src[2i, 2j]       == resize_nearest_neighbor(x, 0.5) (layers.py:169-175)
#define IAF_RESAMPLE_DOWN_ODD 1    // dst[i,j] = src[2i+1, 2j+1]   the outputs a stride-2 SAME 3x3 conv keeps
#define IAF_RESAMPLE_UP_NEAREST 2  // dst[y,x] = src[y/2, x/2]     == resize_nearest_neighbor(x, 2)
#define IAF_RESAMPLE_UP_ZERO_ODD 3 // dst[2i+1,2j+1] = src[i,j], 0 elsewhere: the zero-inserted input of conv2d_transpose
// ... and the adjoints the backward pass needs (UP_ZERO_ODD is DOWN_ODD's, and the other way round):
#define IAF_RESAMPLE_UP_ZERO_EVEN 4 // dst[2i,2j] = src[i,j], 0 elsewhere: adjoint of DOWN_EVEN
#define IAF_RESAMPLE_DOWN_SUM4 5    // dst[i,j] = sum of src[2i..2i+1, 2j..2j+1]: adjoint of UP_NEAREST

// H, W: size of the SMALLER of the two tensors
__global__ __launch_bounds__(256) void iaf_resample2_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n_dst,
                                                           int H, int W, int mode) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dst; i += stride) {
        if (mode == IAF_RESAMPLE_DOWN_EVEN || mode == IAF_RESAMPLE_DOWN_ODD || mode == IAF_RESAMPLE_DOWN_SUM4) {
            const int x = (int)(i % W), y = (int)((i / W) % H);
            const size_t plane = i / ((size_t)W * H);
            const int o = (mode == IAF_RESAMPLE_DOWN_ODD) ? 1 : 0;
            const float* q = src + (plane * 2 * H + 2 * y + o) * 2 * W + 2 * x + o;
            dst[i] = (mode == IAF_RESAMPLE_DOWN_SUM4) ? (q[0] + q[1]) + (q[2 * W] + q[2 * W + 1]) : q[0];
        } else {
            const int W2 = 2 * W, H2 = 2 * H;
            const int x = (int)(i % W2), y = (int)((i / W2) % H2);
            const size_t plane = i / ((size_t)W2 * H2);
            const float v = src[(plane * H + (y >> 1)) * W + (x >> 1)];
            const bool keep = mode == IAF_RESAMPLE_UP_NEAREST || (mode == IAF_RESAMPLE_UP_ZERO_ODD && (y & 1) && (x & 1)) ||
                              (mode == IAF_RESAMPLE_UP_ZERO_EVEN && !(y & 1) && !(x & 1));
            dst[i] = keep ? v : 0.f;
        }
    }
}

// deconv2d weight norm (layers.py:104): l2_normalize(V, [0,1,2]) with V [3,3,n_out,n_in] -> one norm per INPUT channel
__global__ __launch_bounds__(256) void iaf_deconv_norm_kernel(const float* __restrict__ V, float* __restrict__ inv_norm,
                                                             int n_in, int n_out) {
    __shared__ float red[256];
    const int ci = blockIdx.x;
    float ss = 0.f;
    for (int e = threadIdx.x; e < 9 * n_out; e += 256) {
        const float v = V[(size_t)e * n_in + ci];
        ss += v * v;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) inv_norm[ci] = 1.0f / sqrtf(fmaxf(red[0], 1e-12f));
}

// w[a][b][o][c] = exp(g[o]) * V[a][b][o][c] * inv_norm[c], stored for the stride-1 kernel at filter position
// (2-a, 2-b) (180-degree rotation): MFMA fragment order [chunk][tap 9][cot][lane][4], or the direct-conv fallback's
// [tap][c_in][c_out] when `generic`
__global__ __launch_bounds__(256) void iaf_deconv_pack_kernel(const float* __restrict__ V, const float* __restrict__ g,
                                                             const float* __restrict__ b, const float* __restrict__ inv_norm,
                                                             float* __restrict__ wp, float* __restrict__ bias, int n_in, int n_out,
                                                             int ncot, int generic, float* __restrict__ wpt) {
    const size_t total = (size_t)9 * n_out * n_in;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int ci = (int)(e % n_in), o = (int)((e / n_in) % n_out), ab = (int)(e / ((size_t)n_in * n_out));
        const int a = ab / 3, bb = ab % 3;
        const float w = __expf(g[o]) * V[e] * inv_norm[ci];
        const int t = (2 - a) * 3 + (2 - bb);
        if (generic) {
            wp[((size_t)t * n_in + ci) * n_out + o] = w;
        } else {
            const int chunk = ci >> 4, kk = (ci >> 2) & 3, jj = ci & 3, gt = o >> 4, oo = o & 15;
            wp[((((size_t)chunk * 9 + t) * ncot + gt) * 64 + kk * 16 + oo) * 4 + jj] = w;
            // training: the transposed pack of the data gradient, laid out as prep_tile writes it
            if (wpt) wpt[((((size_t)gt * 9 + t) * (n_in >> 4) + chunk) * 64 + (oo >> 2) * 16 + (ci & 15)) * 4 + (oo & 3)] = w;
        }
    }
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n_out; e += stride) bias[e] = b[e];
}

// ---- backward of the deconv2d weight norm ---------------------------------------------------------------------------
//   w[a,b,o,c] = e_o V[a,b,o,c] / n_c,  e_o = exp(g[o]),  n_c = ||V[:,:,:,c]||  (layers.py:104: one norm per INPUT channel)
//   dw[a,b,o,c] = dW_eff[tap (2-a, 2-b)][c][o]      (the conv kernel multiplied the 180-degree-rotated filter)
//   dg[o] = sum_{a,b,c} dw w ;  dV = e_o dw / n_c - V S_c / n_c^3 ,  S_c = sum_{a,b,o} dw e_o V ;  db[o] = sum_p dY
// (1) per input channel: 1/n_c and S_c
__global__ __launch_bounds__(256) void iaf_deconv_bwd_channel_kernel(const float* __restrict__ V, const float* __restrict__ g,
                                                                    const float* __restrict__ dW, float* __restrict__ inv_norm,
                                                                    float* __restrict__ S, int n_in, int n_out) {
    __shared__ float red[2][256];
    const int ci = blockIdx.x;
    float ss = 0.f, sd = 0.f;
    for (int e = threadIdx.x; e < 9 * n_out; e += 256) {
        const int ab = e / n_out, o = e - ab * n_out;
        const float v = V[(size_t)e * n_in + ci];
        const int t = (2 - ab / 3) * 3 + (2 - ab % 3);
        ss += v * v;
        sd += dW[((size_t)t * n_in + ci) * n_out + o] * __expf(g[o]) * v;
    }
    red[0][threadIdx.x] = ss; red[1][threadIdx.x] = sd;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { inv_norm[ci] = 1.0f / sqrtf(fmaxf(red[0][0], 1e-12f)); S[ci] = red[1][0]; }
}
// (2) per output channel: dg, db; and every element of dV
__global__ __launch_bounds__(256) void iaf_deconv_bwd_apply_kernel(const float* __restrict__ V, const float* __restrict__ g,
                                                                  const float* __restrict__ dW, const float* __restrict__ inv_norm,
                                                                  const float* __restrict__ S, const float* __restrict__ dbp, int nslab,
                                                                  float* __restrict__ dV, float* __restrict__ dg, float* __restrict__ db,
                                                                  int n_in, int n_out) {
    __shared__ float red[256];
    const int o = blockIdx.x;
    const float e_o = __expf(g[o]);
    float acc = 0.f;
    for (int e = threadIdx.x; e < 9 * n_in; e += 256) {
        const int ab = e / n_in, ci = e - ab * n_in;
        const size_t idx = ((size_t)ab * n_out + o) * n_in + ci;
        const int t = (2 - ab / 3) * 3 + (2 - ab % 3);
        const float v = V[idx], in = inv_norm[ci];
        const float dw = dW[((size_t)t * n_in + ci) * n_out + o];
        acc += dw * e_o * v * in;
        dV[idx] = e_o * dw * in - v * in * in * in * S[ci];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dg[o] = red[0];
        float s = 0.f;
        for (int r = 0; r < nslab; ++r) s += dbp[(size_t)r * n_out + o];
        db[o] = s;
    }
}
