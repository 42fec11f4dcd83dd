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

// H, W: size of the SMALLER of the two tensors
__global__ __launch_bounds__(256) void iaf_resample2_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n_dst,
                                                           int H, int W, int mode) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dst; i += stride) {
        if (mode == IAF_RESAMPLE_DOWN_EVEN || mode == IAF_RESAMPLE_DOWN_ODD) {
            const int x = (int)(i % W), y = (int)((i / W) % H);
            const size_t plane = i / ((size_t)W * H);
            const int o = (mode == IAF_RESAMPLE_DOWN_ODD) ? 1 : 0;
            dst[i] = src[(plane * 2 * H + 2 * y + o) * 2 * W + 2 * x + o];
        } else {
            const int W2 = 2 * W, H2 = 2 * H;
            const int x = (int)(i % W2), y = (int)((i / W2) % H2);
            const size_t plane = i / ((size_t)W2 * H2);
            const float v = src[(plane * H + (y >> 1)) * W + (x >> 1)];
            dst[i] = (mode == IAF_RESAMPLE_UP_NEAREST || ((y & 1) && (x & 1))) ? v : 0.f;
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
                                                             int ncot, int generic) {
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
        }
    }
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n_out; e += stride) bias[e] = b[e];
}
