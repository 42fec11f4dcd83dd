// iaf_conv_inst.hip -- instantiates iaf_conv_kernel for ONE launch shape (IAF_PXT, IAF_WCO, IAF_KS) and every
// (co-tiles per wave, input mode, epilogue).  Built once per shape by iaf_amd/build.py, in parallel.
#include "iaf_conv_kernel.hpp"

#ifndef IAF_PXT
#error "compile with -DIAF_PXT=.. -DIAF_WCO=.. -DIAF_KS=.."
#endif

template <int NT>
static conv_fn_t pick_mode(int inmode, int epi) {
    if (epi == EPI_PLAIN5) {  // one masked conv on its own (ar_conv2d, layers.py:144-154): raw NCHW output
        if (inmode == IN_NCHW) return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_NCHW, EPI_PLAIN, NTAPS>;
        return nullptr;
    }
    if (epi == EPI_PLAIN) {   // plain 9-tap weight-normed conv (context producers / consumers): NCHW in, NCHW out
        if (inmode == IN_NCHW) return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_NCHW, EPI_PLAIN, MAXTAPS>;
        return nullptr;
    }
    if (epi == EPI_DGRAD9) {  // data gradient of a plain 9-tap conv: pixel-major dY in, NCHW (split) out
        if (inmode == IN_PIXMAJOR) return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_PIXMAJOR, EPI_DGRAD, MAXTAPS>;
        return nullptr;
    }
    if (epi == EPI_DGRAD) {   // data gradient: always reads pixel-major scratch
        if (inmode == IN_PIXMAJOR) return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_PIXMAJOR, EPI_DGRAD>;
        return nullptr;
    }
    if (epi == EPI_HIDDEN_DEEP) {   // double-depth ring: only the shape / tile counts the single-round case uses
#if IAF_PXT == 4 && IAF_WCO == 1 && IAF_KS == 1
        if constexpr (NT >= 4)
            if (inmode == IN_PIXMAJOR) return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_PIXMAJOR, EPI_HIDDEN, NTAPS, true>;
#endif
        return nullptr;
    }
    if (epi == EPI_HIDDEN) {
        if (inmode == IN_PIXMAJOR) return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_PIXMAJOR, EPI_HIDDEN>;
        if (inmode == IN_NCHW) return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_NCHW, EPI_HIDDEN>;
        return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_POSTERIOR, EPI_HIDDEN>;
    }
    if constexpr (NT % 2 == 0) {
        if (inmode == IN_PIXMAJOR) return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_PIXMAJOR, EPI_OUT>;
        if (inmode == IN_NCHW) return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_NCHW, EPI_OUT>;
        return iaf_conv_kernel<NT, IAF_PXT, IAF_WCO, IAF_KS, IN_POSTERIOR, EPI_OUT>;
    }
    return nullptr;
}

#define IAF_CAT_(a, b, c, d) a##b##_##c##_##d
#define IAF_CAT(a, b, c, d) IAF_CAT_(a, b, c, d)

extern "C" conv_fn_t IAF_CAT(iaf_pick_conv_, IAF_PXT, IAF_WCO, IAF_KS)(int nt, int inmode, int epi) {
    switch (nt) {
        case 1: return pick_mode<1>(inmode, epi);
        case 2: return pick_mode<2>(inmode, epi);
        case 3: return pick_mode<3>(inmode, epi);
        case 4: return pick_mode<4>(inmode, epi);
        case 5: return pick_mode<5>(inmode, epi);
    }
    return nullptr;
}
