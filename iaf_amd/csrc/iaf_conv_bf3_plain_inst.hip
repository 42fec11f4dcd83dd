// iaf_conv_bf3_plain_inst.hip -- instantiates iaf_conv_bf3_kernel with all 9 taps for the plain weight-normed conv2d around
// the IAF step (up_conv1/3, down_conv1/2, tf_train.py:36,41,53,93): NCHW input with the graph's elu / concat, EPI_PLAIN
// epilogue (bias, split store, residual), ONE launch shape per translation unit.  Built by iaf_amd/build.py.
#include "iaf_conv_bf3.hpp"

#ifndef IAF_WCO
#define IAF_WCO 1
#endif
#ifndef IAF_PPW
#error "compile with -DIAF_PPW=.. -DIAF_PXT=.. -DIAF_KS=.."
#endif

#define IAF_CAT_(a, b, c, d, e) a##b##_##c##_##d##_##e
#define IAF_CAT(a, b, c, d, e) IAF_CAT_(a, b, c, d, e)

// (WCO = 3: 768 threads, three waves per SIMD, 170 registers -- NT = 5 does not fit and is not instantiated)
// epi: EPI_PLAIN (forward), or EPI_DGRAD -- the data gradient of the same conv: dY pixel-major, transposed bf16x3 pack,
// mirrored taps (iaf_conv3x3_backward)
extern "C" conv_fn_t IAF_CAT(iaf_pick_bf3p_, IAF_PPW, IAF_PXT, IAF_KS, IAF_WCO)(int nt, int epi) {
    if (epi == EPI_DGRAD) {
        switch (nt) {
            case 2: return iaf_conv_bf3_kernel<2, IAF_PPW, IAF_PXT, IAF_KS, IN_PIXMAJOR, EPI_DGRAD, IAF_WCO, MAXTAPS>;
            case 4: return iaf_conv_bf3_kernel<4, IAF_PPW, IAF_PXT, IAF_KS, IN_PIXMAJOR, EPI_DGRAD, IAF_WCO, MAXTAPS>;
#if IAF_WCO < 3
            case 5: return iaf_conv_bf3_kernel<5, IAF_PPW, IAF_PXT, IAF_KS, IN_PIXMAJOR, EPI_DGRAD, IAF_WCO, MAXTAPS>;
#endif
        }
        return nullptr;
    }
    switch (nt) {
        case 2: return iaf_conv_bf3_kernel<2, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS>;
        case 4: return iaf_conv_bf3_kernel<4, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS>;
#if IAF_WCO < 3
        case 5: return iaf_conv_bf3_kernel<5, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS>;
#endif
    }
    return nullptr;
}

// the downsampling layer's strided convs at their minimal work (iaf_conv_bf3.hpp, S2): s2 = 1 conv2d stride 2, 2 deconv2d by phases
extern "C" conv_fn_t IAF_CAT(iaf_pick_bf3s_, IAF_PPW, IAF_PXT, IAF_KS, IAF_WCO)(int nt, int s2) {
    if (s2 == 1) {
        switch (nt) {
            case 2: return iaf_conv_bf3_kernel<2, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS, 1>;
            case 4: return iaf_conv_bf3_kernel<4, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS, 1>;
#if IAF_WCO < 3
            case 5: return iaf_conv_bf3_kernel<5, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS, 1>;
#endif
        }
    } else if (s2 == 2) {
        switch (nt) {
            case 2: return iaf_conv_bf3_kernel<2, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS, 2>;
            case 4: return iaf_conv_bf3_kernel<4, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS, 2>;
#if IAF_WCO < 3
            case 5: return iaf_conv_bf3_kernel<5, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS, 2>;
#endif
        }
    }
    return nullptr;
}

// the forward plain conv on two fp16 planes (iaf_conv_bf3.hpp F16; IAF_PRECISION_F16X2): p.wp = the two-plane pack
extern "C" conv_fn_t IAF_CAT(iaf_pick_bf3p16_, IAF_PPW, IAF_PXT, IAF_KS, IAF_WCO)(int nt) {
    switch (nt) {
        case 2: return iaf_conv_bf3_kernel<2, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS, 0, 1>;
        case 4: return iaf_conv_bf3_kernel<4, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS, 0, 1>;
#if IAF_WCO < 3
        case 5: return iaf_conv_bf3_kernel<5, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_PLAIN, IAF_WCO, MAXTAPS, 0, 1>;
#endif
    }
    return nullptr;
}

// ... and its data gradient on two fp16 planes with a tile-local scale (iaf_conv_bf3.hpp DG16): dY pixel-major, the transposed two-plane pack
extern "C" conv_fn_t IAF_CAT(iaf_pick_bf3p16d_, IAF_PPW, IAF_PXT, IAF_KS, IAF_WCO)(int nt) {
    switch (nt) {
        case 2: return iaf_conv_bf3_kernel<2, IAF_PPW, IAF_PXT, IAF_KS, IN_PIXMAJOR, EPI_DGRAD, IAF_WCO, MAXTAPS, 0, 1>;
        case 4: return iaf_conv_bf3_kernel<4, IAF_PPW, IAF_PXT, IAF_KS, IN_PIXMAJOR, EPI_DGRAD, IAF_WCO, MAXTAPS, 0, 1>;
#if IAF_WCO < 3
        case 5: return iaf_conv_bf3_kernel<5, IAF_PPW, IAF_PXT, IAF_KS, IN_PIXMAJOR, EPI_DGRAD, IAF_WCO, MAXTAPS, 0, 1>;
#endif
    }
    return nullptr;
}
