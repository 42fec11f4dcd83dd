// iaf_conv_bf3_inst.hip -- instantiates iaf_conv_bf3_kernel for ONE launch shape (IAF_PPW pixel tiles per wave, IAF_PXT
// waves along pixels, IAF_KS K-slice waves) and every (co tiles per wave, input mode, epilogue) the stack uses.
// Built once per shape by iaf_amd/build.py, in parallel.
#include "iaf_conv_bf3.hpp"

#ifndef IAF_WCO
#define IAF_WCO 1
#endif
#ifndef IAF_PPW
#error "compile with -DIAF_PPW=.. -DIAF_PXT=.. -DIAF_KS=.."
#endif

template <int NT>
static conv_fn_t pick_mode_bf3(int inmode, int epi) {
    if (epi == EPI_HIDDEN) {
        if (inmode == IN_PIXMAJOR) return iaf_conv_bf3_kernel<NT, IAF_PPW, IAF_PXT, IAF_KS, IN_PIXMAJOR, EPI_HIDDEN, IAF_WCO>;
        // (the first layer fused into this one's prologue, iaf_stack_set_fuse_first: not with two co groups per workgroup -- those three
        //  instantiations spilled 10-68 VGPRs, and the form has lost to separate launches at every size measured on MI355X anyway)
        if constexpr (IAF_WCO == 1) {
            if (inmode == IN_FUSED0) return iaf_conv_bf3_kernel<NT, IAF_PPW, IAF_PXT, IAF_KS, IN_FUSED0, EPI_HIDDEN, IAF_WCO>;
        }
        if (inmode == IN_NCHW) return iaf_conv_bf3_kernel<NT, IAF_PPW, IAF_PXT, IAF_KS, IN_NCHW, EPI_HIDDEN, IAF_WCO>;
        if (inmode == IN_POSTERIOR) return iaf_conv_bf3_kernel<NT, IAF_PPW, IAF_PXT, IAF_KS, IN_POSTERIOR, EPI_HIDDEN, IAF_WCO>;
        return nullptr;
    }
    if (epi == EPI_DGRAD)     // data gradient of a masked conv: dY pixel-major, transposed bf16x3 pack, mirrored taps
        return inmode == IN_PIXMAJOR ? iaf_conv_bf3_kernel<NT, IAF_PPW, IAF_PXT, IAF_KS, IN_PIXMAJOR, EPI_DGRAD, IAF_WCO> : nullptr;
    if (epi == EPI_OUT) {     // the output pair always reads the last hidden layer (depth_ar = 0 stays on the fp32 kernel)
        if constexpr (NT % 2 == 0) {
            if (inmode == IN_PIXMAJOR) return iaf_conv_bf3_kernel<NT, IAF_PPW, IAF_PXT, IAF_KS, IN_PIXMAJOR, EPI_OUT, IAF_WCO>;
            if constexpr (IAF_WCO == 1) {
                if (inmode == IN_FUSED0) return iaf_conv_bf3_kernel<NT, IAF_PPW, IAF_PXT, IAF_KS, IN_FUSED0, EPI_OUT, IAF_WCO>;
            }
        }
        return nullptr;
    }
    return nullptr;
}

#define IAF_CAT_(a, b, c, d, e) a##b##_##c##_##d##_##e
#define IAF_CAT(a, b, c, d, e) IAF_CAT_(a, b, c, d, e)

extern "C" conv_fn_t IAF_CAT(iaf_pick_bf3_, IAF_PPW, IAF_PXT, IAF_KS, IAF_WCO)(int nt, int inmode, int epi) {
    switch (nt) {
        case 2: return pick_mode_bf3<2>(inmode, epi);
        case 4: return pick_mode_bf3<4>(inmode, epi);
        case 5: return pick_mode_bf3<5>(inmode, epi);
    }
    return nullptr;
}
