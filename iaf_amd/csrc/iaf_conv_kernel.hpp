// iaf_conv_kernel.hpp -- the masked 3x3 conv implicit-GEMM kernel template (see iaf_engine.hip for the
// design notes).  Included by iaf_engine.hip (types only) and by iaf_conv_inst.hip (instantiations).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

#define NTAPS 5        // live taps of a MADE-masked 3x3 filter
#define MAXSPLIT 6     // down_conv1 feeds 6 tensors (tf_train.py:54)
#define MAXTAPS 9      // a full (unmasked) 3x3 filter: the context producers / consumers (up_conv1, down_conv1, ...)
// The 5 live taps are the filter positions (kh,kw) = (1,1) (1,2) (2,0) (2,1) (2,2), centre first; their (dh,dw)
// relative to the output pixel come from ConvP (TF: (0,0)(0,1)(1,-1)(1,0)(1,1); Theano: the negated set).

#define EPI_HIDDEN 0   // y = elu(acc + bias [+ ctx (+ ctx2)])  -> pixel-major scratch
#define EPI_OUT 1      // output pair (mean, logsd) -> NCHW; mode selects raw / IAF step / posterior
#define EPI_DGRAD 2    // data gradient (transposed packs, mirrored taps): mode selects MODE_DGRAD_ELU / MODE_DGRAD_Z
#define EPI_PLAIN5 4   // host-side selector only: EPI_PLAIN with the 5 masked taps (a single ar_conv2d)
#define EPI_DGRAD9 5   // host-side selector only: EPI_DGRAD with all 9 taps (data gradient of a plain conv)
#define EPI_HIDDEN_DEEP 6   // host-side selector only: EPI_HIDDEN, pixel-major input, double-depth weight ring
#define EPI_PLAIN 3    // y = acc + bias [-> res + 0.1*y]  -> NCHW   (plain weight-normed conv2d, layers.py:63-64; tf_train.py:44,94)

#define MODE_RAW 0        // out0 = m_raw, out1 = s_raw                         (layers.py:166)
#define MODE_IAF 1        // out0 = (z-0.1m)/exp(0.1s), out1 = 0.1s             (tf_train.py:70-72)
#define MODE_POSTERIOR 2  // MODE_IAF on z0 rebuilt from the posterior inputs, plus kl elements
#define MODE_DGRAD_PLAIN 6 // EPI_DGRAD of a plain conv: dx = res + elu'(xe) * (W^T dY) -> NCHW through the split table
#define MODE_INVERSE 5    // one Jacobi sweep of the inverse flow: out0 = zin*exp(0.1s) + 0.1m, out1 = 0.1s (x = current z0 estimate)
// backward (data gradient) modes of EPI_DGRAD: the same kernel run on the transposed packed weights with the tap
// table negated computes dX = W^T * dY; the epilogue applies what autodiff applies next.  (Its own compile-time
// epilogue: as runtime branches inside EPI_HIDDEN these modes cost the forward kernels ~1800 cycles.)
#define MODE_DGRAD_ELU 3  // y = acc * elu'(a) with elu'(a) = (h > 0 ? 1 : h + 1), h = saved activation (p.zin, pixel-major);
                          // optional NCHW copy to p.out0 (d context, layers.py:163-164)
#define MODE_DGRAD_Z 4    // out0[NCHW] = acc + dz_new * exp(-logsd)   (p.qm = dz_new, p.ql = logsd; tf_train.py:71)

#define IN_PIXMAJOR 0     // x is [P][c_in] scratch written by a previous EPI_HIDDEN
#define IN_NCHW 1         // x is an NCHW tensor (z)
#define IN_POSTERIOR 2    // x = z0 = (qm+rm) + exp(ql+rl)*eps computed on the fly (tf_train.py:57,63)
#define IN_FUSED0 4       // bf16x3 kernels only: the stack's first masked conv (ConvP.f_*) is computed in this kernel's prologue,
                          // straight into the LDS tile the K loop reads (iaf_conv_bf3.hpp)

struct ConvP {
    // Field order = order of first use: the kernel pulls the whole descriptor into SGPRs in one batch of scalar loads
    // at its first instruction (pin_descriptor), so the kernarg cache lines miss in parallel instead of one after another.
    const float* x;       // IN_PIXMAJOR / IN_NCHW input
    const float* wp;      // packed weights [chunk][tap][co_tile][lane 64][4]
    int B, H, W, HW, P;   // P = B*H*W
    int cin, cout;        // GEMM K channels, GEMM N (EPI_OUT: 2*n_z)
    int nchunk, ncot;
    int cp;               // padded channel stride of the LDS tile (floats), == cin + 8
    int nslot;            // staged pixel slots = TM + W + 1 (one-sided halo; both sides for the full 3x3)
    int mode;
    int halo_before;      // staged slots start at pixel P0 - halo_before
    int in_elu;           // EPI_PLAIN: ELU on the staged input (tf_train.py:35,40,52,88)
    int gx;               // gridDim.x (read from here: the hidden kernarg lives in another, cold, cache line)
    // tap geometry (runtime so that both statements of the operator share the instantiations):
    //   TF      (tf_utils/layers.py, cross-correlation): taps look right/below, halo after the tile
    //   Theano  (graphy/nodes/ar.py + dnn_conv conv_mode='conv', flipped kernel): taps look left/above, halo before
    int tap_dh[MAXTAPS], tap_dw[MAXTAPS];
    const float* bias;    // packed bias
    const int* lim;       // per packed co-tile: number of live 16-channel chunks of the centre tap (unused for now)
    const float* ctx;     // EPI_HIDDEN: optional NCHW context  [B,cout,H,W]
    const float* ctx2;    // EPI_HIDDEN: optional second context (up_context + down_context)
    float* y;             // EPI_HIDDEN output, pixel-major [P][cout]
    const float* zin;     // EPI_OUT MODE_IAF: z  [B,n_z,H,W]
    float* out0;          // EPI_OUT: z_new / m_raw
    float* out1;          // EPI_OUT: logsd / s_raw
    const float* border;  // Theano pad_channel: [4][cout packed] weight of the border-indicator channel per non-centre tap
    unsigned long long* dbg;   // dev tool: per-workgroup s_memtime stamps [grid][8] (NULL in production)
    // posterior inputs (IN_POSTERIOR staging and MODE_POSTERIOR epilogue), all [B,n_z,H,W]
    const float* qm; const float* ql; const float* rm; const float* rl; const float* pm; const float* pl;
    const float* eps;
    float* kl_elem;       // MODE_POSTERIOR: logqs - logps [B,n_z,H,W]
    // EPI_PLAIN extras: input = elu(concat(x[:, :c_split], x2)) (tf_train.py:36,52,87-88), residual for the output
    const float* x2; const float* res; int c_split;
    // EPI_PLAIN output: the channel split of tf_train.py:37,54 fused into the store.  Channels [split_end[k-1], split_end[k])
    // go to the contiguous NCHW tensor split_ptr[k]; boundaries are multiples of 4 (one lane's 4 channels never straddle).
    int nsplit; int split_end[MAXSPLIT]; float* split_ptr[MAXSPLIT];
    int lds_bytes;        // dynamic LDS of this launch (read only by the -DIAF_EXP_POISON_LDS soak build)
    // IN_FUSED0: the fused first layer -- its bf16x3 pack / bias, its NCHW input z (NULL: the posterior sample from
    // qm/rm/ql/rl/eps), its c_in (32) and the context(s) added to its output (layers.py:163-164; tf_train.py:58)
    const void* f_wp; const float* f_bias; const float* f_x; int f_cin; const float* f_ctx; const float* f_ctx2;
    // bf16x3 kernels with S2 = 1 (conv2d stride 2, tf_train.py:33,36: H, W = the OUTPUT grid, x is [B,c_in,2H,2W]): the staged tile
    // is four phase tiles x_ab[i][j] = x[2i+a][2j+b]; s2_pb[2a+b] = first slot of phase (a,b), tap_off[t] = slot offset of tap t
    // = s2_pb[phase of t] + (di>>1) W + (dj>>1), tap_dh / tap_dw = (di>>1, dj>>1) for the border test
    int s2_pb[4]; int tap_off[MAXTAPS];
    // MODE_INVERSE sweeps chained without the host (iaf_step_inverse): a device word that is non-zero once the fixed point has been
    // reached -- the one-launch step kernel returns at once then (the layer-by-layer kernels ignore it: a sweep at the fixed point
    // changes nothing); NULL otherwise.  Host-side field only: read by launch_fused_step.
    const unsigned* inv_done;
    // two-plane fp16 kernels (iaf_conv_bf3.hpp F16): host-visible word raised when an operand beyond fp16's largest finite number was staged
    unsigned* rng_err;
};

// Pin the order "MFMAs with memory instructions spread evenly between them" inside the current scheduling region:
// NMF MFMAs, then NRD ds_reads, NWR ds_writes, NLDG global loads, each memory op after its share of the MFMAs.
// (An LDS/VMEM instruction occupies the wave's issue port for ~25 cycles, an MFMA keeps the pipe busy for 32: a
// cluster of memory instructions starves the pipe, one between every few MFMAs is free.)
template <int NMF, int NRD, int NWR, int NLDG>
__device__ __forceinline__ void sched_interleave() {
    constexpr int NMEM = NRD + NWR + NLDG;
    static_for<NMEM>([&](auto m_c) {
        constexpr int m = decltype(m_c)::value;
        constexpr int prev = (m * NMF) / (NMEM + 1);
        constexpr int upto = ((m + 1) * NMF) / (NMEM + 1);
        if constexpr (upto > prev) __builtin_amdgcn_sched_group_barrier(0x008, upto - prev, 0);
        if constexpr (m < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);              // DS read
        else if constexpr (m < NRD + NWR) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
        else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                // VMEM read
    });
    constexpr int last = (NMEM * NMF) / (NMEM + 1);
    if constexpr (NMF > last) __builtin_amdgcn_sched_group_barrier(0x008, NMF - last, 0);
}

// ELU with the hardware exponential: exp(v) - 1 for v <= 0 (what TF's fp32 kernel evaluates); abs error < 1e-7
__device__ __forceinline__ float elu_f(float v) { return v > 0.f ? v : __expf(v) - 1.0f; }

// Tiling (all compile time):
//   workgroup = PXT pixel tiles (16 px each) x WCO co-groups x KS K-slices, one wave each;
//   a wave owns NT co-tiles (16 channels each) of one pixel tile for its K slice.
// MFMA roles (v_mfma_f32_16x16x4_f32, D[i][j] += A[i][k] B[k][j]):
//   A = weights  : lane l holds W[co = tile*16 + (l&15)][k-slot l>>4]
//   B = activations: lane l holds X[k-slot l>>4][pixel l&15]
//   D            : lane l holds D[co = tile*16 + 4*(l>>4) + r][pixel l&15], r = 0..3
// K order inside a 16-channel chunk is permuted: k-slot kk owns channels 4kk..4kk+3, MFMA j of the chunk
// consumes channel 4kk+j, so each operand is ONE 16-byte load per lane per 4 MFMAs.
template <int NT, int PXT, int WCO, int KS, int INMODE, int EPI, int NTP = NTAPS, bool DEEP = false>
__global__ __launch_bounds__(64 * PXT * WCO * KS) void iaf_conv_kernel(ConvP p) {
    extern __shared__ __attribute__((aligned(16))) f32x4 smem4[];
    constexpr int TM = 16 * PXT;
    constexpr int NTHREADS = 64 * PXT * WCO * KS;
    constexpr int WPK = PXT * WCO;   // waves per K slice
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: keeps tile/K-slice indices and
                                                                 // the weight base pointer in SGPRs (saddr loads, no 64-bit VALU)
    const int pw = wave % PXT, cw = (wave / PXT) % WCO, kh = wave / WPK;
    const int P0 = blockIdx.x * TM;
    const int cot0 = (blockIdx.y * WCO + cw) * NT;
    // Pull the descriptor into SGPRs NOW, as one batch: left to itself the compiler loads each kernarg field right
    // before its first use, and the prologue becomes a chain of ~10 dependent scalar-cache misses (measured: each extra
    // cold kernarg line costs 0.15-0.25 us of a 6-22 us kernel).
    asm volatile("" ::"s"(p.x), "s"(p.wp), "s"(p.B), "s"(p.H), "s"(p.W), "s"(p.HW), "s"(p.P), "s"(p.cin), "s"(p.cout),
                 "s"(p.nchunk), "s"(p.ncot), "s"(p.cp), "s"(p.nslot), "s"(p.mode), "s"(p.halo_before), "s"(p.gx));
    asm volatile("" ::"s"(p.tap_dh[0]), "s"(p.tap_dh[1]), "s"(p.tap_dh[2]), "s"(p.tap_dh[3]), "s"(p.tap_dh[4]), "s"(p.tap_dw[0]),
                 "s"(p.tap_dw[1]), "s"(p.tap_dw[2]), "s"(p.tap_dw[3]), "s"(p.tap_dw[4]));
    if constexpr (NTP == MAXTAPS)
        asm volatile("" ::"s"(p.tap_dh[5]), "s"(p.tap_dh[6]), "s"(p.tap_dh[7]), "s"(p.tap_dh[8]), "s"(p.tap_dw[5]), "s"(p.tap_dw[6]),
                     "s"(p.tap_dw[7]), "s"(p.tap_dw[8]));
    asm volatile("" ::"s"(p.bias), "s"(p.ctx), "s"(p.ctx2), "s"(p.y), "s"(p.zin), "s"(p.out0), "s"(p.out1), "s"(p.border),
                 "s"(p.dbg));
    if constexpr (INMODE == IN_POSTERIOR || EPI == EPI_OUT)
        asm volatile("" ::"s"(p.qm), "s"(p.ql), "s"(p.rm), "s"(p.rl), "s"(p.pm), "s"(p.pl), "s"(p.eps), "s"(p.kl_elem));
    if constexpr (EPI == EPI_PLAIN || (EPI == EPI_DGRAD && NTP == MAXTAPS))
        asm volatile("" ::"s"(p.x2), "s"(p.res), "s"(p.c_split), "s"(p.in_elu), "s"(p.nsplit));
    const int HW = p.HW, W = p.W;
    const int cp4 = p.cp >> 2;       // LDS row stride in 16-byte units
#define IAF_STAMP(k) do { if (p.dbg && tid == 0) p.dbg[((size_t)blockIdx.y * p.gx + blockIdx.x) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
    IAF_STAMP(0);
#ifdef IAF_EXP_POISON_LDS
    // soak build (tools/soak.py): every LDS word starts as a signalling pattern, so a read of a word this launch never
    // wrote -- left-over data of the previous workgroup on the CU in a production build -- turns the output into NaN
    for (int i = tid; i < (p.lds_bytes >> 4); i += NTHREADS) smem4[i] = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
    __syncthreads();
#endif

    // ================= prologue, ordered by latency: (1) tile loads, (2) weight ring, (3) index math ==========
    // (1) activation tile: slots [P0, P0+nslot) x cin.  For the pixel-major scratch the tile is ONE contiguous run
    // of global memory, so the loads need no index math at all and are issued first, all in one batch.
    const int nq = p.cin >> 2;                     // 16-byte items per pixel
    const int nitems = p.nslot * nq;
    constexpr int SU = (INMODE == IN_PIXMAJOR) ? 16 : 4;    // items in flight per thread
    f32x4 sv[SU];
    const int Pbase = P0 - p.halo_before;          // global pixel of slot 0 (negative for the first tile of the Theano variant)
    const int flo = Pbase < 0 ? -Pbase * nq : 0;                       // items whose pixel exists: [flo, fhi)
    const long long rem = (long long)(p.P - Pbase) * nq;
    const int fhi = rem < nitems ? (int)rem : nitems;
    if (INMODE == IN_PIXMAJOR) {
        const f32x4* src = (const f32x4*)p.x + (long long)Pbase * nq;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int f = tid + u * NTHREADS;
            const int fc = f < flo ? flo : (f < fhi ? f : fhi - 1);    // clamped: branch-free; out-of-range items zeroed
            sv[u] = src[fc];
            if (f < flo || f >= fhi) sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // (2) weight ring.  A step is one (chunk, tap): NT x 1 KiB global_load_dwordx4 feeding 4*NT MFMAs.  Weights
    // come from L2 at ~1 us latency while a step is only 128*NT cycles, so R = NTP*RCH steps are kept in flight in
    // registers.  Step s lives in ring slot s % R; while step s computes, the
    // slot consumed by step s-1 is refilled with step s+R-1, the loads interleaved between the MFMAs.  Hence the
    // prologue fetches R-1 steps.  Loop bodies are straight-line (static slots/taps, no branches) so that hipcc
    // emits COUNTED s_waitcnt vmcnt(N) and the ring really stays in flight.
    // Depth: one chunk (5 or 9 steps) for NT >= 4, two for NT = 2..3, four for NT = 1 -- about 180 VGPRs at NT = 5, so
    // two workgroups share a CU and one's prologue/epilogue overlaps the other's K loop (measured: 3-20 % faster on every
    // BASELINE config than a ring twice as deep with one workgroup per CU).  DEEP doubles it: only worth it when the whole
    // grid is a single round of workgroups (nothing to overlap with), i.e. the 160->160 conv at B = 32, 16x16.
    constexpr int RCH_BASE = (NTP == NTAPS) ? ((NT >= 4) ? 1 : (NT >= 2 ? 2 : 4)) : ((NT >= 2) ? 1 : 2);
    constexpr int RCH = DEEP ? 2 * RCH_BASE : RCH_BASE;
    constexpr int R = RCH * NTP;
    const size_t wstep = (size_t)p.ncot * 64;   // f32x4 per (chunk,tap) step
    const int c_begin = (kh * p.nchunk) / KS, c_end = ((kh + 1) * p.nchunk) / KS;
    const bool ramp = DEEP && RCH == 2 && (c_begin + 2 * RCH <= c_end);   // see the prologue / RAMP body
    const f32x4* wbase = (const f32x4*)p.wp + (size_t)cot0 * 64;   // wave-uniform; lane offset added per load
    const unsigned ulane = (unsigned)lane;
    // ---- weight delivery: the per-wave register ring above is what ships.  An ablation variant for PXT > 1 (the PXT
    //  waves of a group need the SAME weight tiles) is kept behind -DIAF_SHARED_W=1: each 1 KiB tile fetched ONCE per
    //  group, parked in a double-buffered LDS chunk buffer and read by all waves with ds_read_b128 (the CU's address unit
    //  handles one 1 KiB wave-load per ~16 cycles, and sharing cuts that traffic PXT-fold).  Measured on the 160->160
    //  conv at B=32 16x16 (cycles per K step, 640 = MFMA-bound): ring 712, shared 768, both with the memory instructions
    //  interleaved between the MFMAs (873 / 867 with them clustered) -- the extra LDS reads cost more issue slots than
    //  the saved global loads.
#if defined(IAF_SHARED_W) && IAF_SHARED_W
    constexpr bool SHARED_W = (PXT > 1);
#else
    constexpr bool SHARED_W = false;
#endif
    constexpr int NTILE_CH = NTP * NT;                       // weight tiles per chunk
    constexpr int NLD = (NTILE_CH + PXT - 1) / PXT;            // tiles fetched per wave per chunk
    f32x4 wr[SHARED_W ? 1 : R][NT];
    f32x4 sr[SHARED_W ? NLD : 1];
    f32x4* wlds = smem4 + (size_t)(p.nslot + 1) * cp4 + (size_t)(wave / PXT) * (2 * NTILE_CH * 64);   // this group's 2 buffers
    auto issue_stage = [&](int chunk) {       // my share of chunk's tiles -> registers (clamped index: branch-free)
        const f32x4* q = wbase + (size_t)chunk * NTP * wstep;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int f = pw + i * PXT;
            f = f < NTILE_CH ? f : NTILE_CH - 1;
            const int tp = f / NT, t = f - tp * NT;
            sr[i] = (q + (size_t)tp * wstep + t * 64)[ulane];
        }
    };
    auto write_stage = [&](int buf) {         // registers -> LDS chunk buffer `buf`
        f32x4* wb = wlds + buf * (NTILE_CH * 64) + lane;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int f = pw + i * PXT;
            f = f < NTILE_CH ? f : NTILE_CH - 1;
            wb[f * 64] = sr[i];
        }
    };
    auto ring_prologue = [&](auto i) {
        constexpr int I = decltype(i)::value;
        if (c_begin + I < c_end) {
            const f32x4* q = wbase + (size_t)(c_begin + I) * NTP * wstep;
#pragma unroll
            for (int tp = 0; tp < NTP; ++tp) {
                if (I == RCH - 1 && tp == NTP - 1) continue;    // slot R-1 is filled by step 0
#pragma unroll
                for (int t = 0; t < NT; ++t) wr[I * NTP + tp][t] = (q + (size_t)tp * wstep)[ulane + t * 64];
            }
        }
    };
    if constexpr (SHARED_W) {
        if (c_begin < c_end) issue_stage(c_begin);
    } else {
        ring_prologue(std::integral_constant<int, 0>{});   // chunk 0 now; chunks 1..RCH-1 after the tile is staged
    }
    IAF_STAMP(1);

    // (3) per-lane geometry: MFMA B operand lane = (pixel l&15, k-slot l>>4)
    const int pl = lane & 15, kk = lane >> 4;
    const int Pl = P0 + pw * 16 + pl;
    const bool pvalid = Pl < p.P;
    const int bimg = Pl / HW, pp = Pl - bimg * HW;
    const int h = pp / W, w = pp - h * W;
    int xa[NTP];   // 16-byte offset into the LDS tile of this lane's 4 channels for each tap (chunk 0)
    unsigned outside = 0;   // bit t: tap t falls outside the image for this lane's pixel (Theano border channel)
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
        const int dh = p.tap_dh[t], dw = p.tap_dw[t];
        const bool v = pvalid && (h + dh >= 0) && (h + dh < p.H) && (w + dw >= 0) && (w + dw < W);
        const int slot = pw * 16 + pl + dh * W + dw + p.halo_before;
        xa[t] = (v ? slot : p.nslot) * cp4 + kk;     // image borders read the all-zero slot
        outside |= (v ? 0u : 1u) << t;
    }

    // (4) tile -> LDS
    {
        f32x4* zslot = smem4 + (size_t)p.nslot * cp4;
        for (int i = tid; i < cp4; i += NTHREADS) zslot[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (INMODE == IN_PIXMAJOR) {
            const float rnq = 1.0f / (float)nq;
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int f = tid + u * NTHREADS;
                if (f < nitems) {
                    const int sl = (int)(((float)f + 0.5f) * rnq);
                    smem4[sl * cp4 + (f - sl * nq)] = sv[u];
                }
            }
            // tiles larger than SU*NTHREADS items (c_in > 192 at TM = 64): plain extra rounds
            const f32x4* src = (const f32x4*)p.x + (long long)Pbase * nq;
            for (int f = tid + SU * NTHREADS; f < nitems; f += NTHREADS) {
                const int sl = (int)(((float)f + 0.5f) * rnq);
                smem4[sl * cp4 + (f - sl * nq)] = (f >= flo && f < fhi) ? src[f] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
            for (int base = tid; base < nitems; base += SU * NTHREADS) {
                int dst[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int idx = base + u * NTHREADS;
                    dst[u] = -1;
                    sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (idx < nitems) {
                        const int q = idx / p.nslot, sl = idx - q * p.nslot;   // slot fastest: coalesced along pixels
                        const int Pg = Pbase + sl;
                        dst[u] = sl * cp4 + q;
                        if (Pg >= 0 && Pg < p.P) {
                            const int b = Pg / HW, ppx = Pg - b * HW;
                            const size_t gb = ((size_t)b * p.cin + 4 * q) * HW + ppx;
                            if (INMODE == IN_NCHW) {
                                if (EPI == EPI_PLAIN && p.x2 && 4 * q >= p.c_split) {        // second tensor of a channel concat
                                    const size_t g2 = ((size_t)b * (p.cin - p.c_split) + (4 * q - p.c_split)) * HW + ppx;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) sv[u][r] = p.x2[g2 + (size_t)r * HW];
                                } else {
                                    const size_t g1 = (EPI == EPI_PLAIN && p.x2) ? ((size_t)b * p.c_split + 4 * q) * HW + ppx : gb;
#pragma unroll
                                    for (int r = 0; r < 4; ++r) sv[u][r] = p.x[g1 + (size_t)r * HW];
                                }
                                if (EPI == EPI_PLAIN && p.in_elu) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) sv[u][r] = elu_f(sv[u][r]);
                                }
                            } else {   // z0 = (qm+rm) + exp(0.5*2*(ql+rl)) * eps   (tf_train.py:57,63; distributions.py:21)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const size_t i = gb + (size_t)r * HW;
                                    sv[u][r] = (p.qm[i] + p.rm[i]) + __expf(0.5f * (2.f * (p.ql[i] + p.rl[i]))) * p.eps[i];
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < SU; ++u)
                    if (dst[u] >= 0) smem4[dst[u]] = sv[u];
            }
        }
    }
    if constexpr (SHARED_W) {
        if (c_begin < c_end) write_stage(0);
        if (c_begin + 1 < c_end) issue_stage(c_begin + 1);
    } else {
        // DEEP (two chunks): with enough chunks to run the steady loop, the second chunk's prologue loads are NOT issued
        // here -- 4*NT more 1 KiB loads per wave through an address unit that is the prologue's bottleneck -- but ride along
        // with the MFMAs of the first chunk (RAMP body below).
        if (!ramp) static_for<RCH - 1>([&](auto i) { ring_prologue(std::integral_constant<int, decltype(i)::value + 1>{}); });
    }
    __syncthreads();
    IAF_STAMP(2);

    // ================= epilogue work assignment ==========================================================
    // With split-K the KS waves of one (pixel tile, co group) share the epilogue: wave kh finishes units
    // u = kh, kh+KS, ...  (unit = one co-tile for EPI_HIDDEN, one (mean, logsd) tile pair for EPI_OUT).
    constexpr bool ONE_TILE_UNITS = (EPI != EPI_OUT);
    constexpr int NUNIT = ONE_TILE_UNITS ? NT : NT / 2;
    constexpr int NMY = (NUNIT + KS - 1) / KS;
    f32x4 pre0[NMY], pre1[NMY], pbias[NMY * (ONE_TILE_UNITS ? 1 : 2)];
    auto prefetch_epilogue = [&]() {     // operands that do not depend on the GEMM: bias, context, z
        if (!pvalid) return;
#pragma unroll
        for (int i = 0; i < NMY; ++i) {
            const int u = kh + i * KS;
            if (u >= NUNIT) continue;
            if (EPI == EPI_PLAIN) {
                pbias[i] = *(const f32x4*)(p.bias + (cot0 + u) * 16 + 4 * kk);
                if (p.res) {
                    const size_t cb = ((size_t)bimg * p.cout + (cot0 + u) * 16 + 4 * kk) * HW + pp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre0[i][r] = p.res[cb + (size_t)r * HW];
                }
            } else if (EPI == EPI_DGRAD) {
                if (p.mode == MODE_DGRAD_ELU) {
                    pre0[i] = *(const f32x4*)(p.zin + (size_t)Pl * p.cout + (cot0 + u) * 16 + 4 * kk);
                } else if (p.mode == MODE_DGRAD_PLAIN) {
                    if (p.zin) pre0[i] = *(const f32x4*)(p.zin + (size_t)Pl * p.cout + (cot0 + u) * 16 + 4 * kk);
                    if (p.res) {
                        const size_t cb = ((size_t)bimg * p.cout + (cot0 + u) * 16 + 4 * kk) * HW + pp;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pre1[i][r] = p.res[cb + (size_t)r * HW];
                    }
                } else {
                    const size_t cb = ((size_t)bimg * p.cout + (cot0 + u) * 16 + 4 * kk) * HW + pp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pre0[i][r] = p.qm[cb + (size_t)r * HW];
                        pre1[i][r] = p.ql[cb + (size_t)r * HW];
                    }
                }
            } else if (EPI == EPI_HIDDEN) {
                pbias[i] = *(const f32x4*)(p.bias + (cot0 + u) * 16 + 4 * kk);
                if (p.ctx) {
                    const size_t cb = ((size_t)bimg * p.cout + (cot0 + u) * 16 + 4 * kk) * HW + pp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre0[i][r] = p.ctx[cb + (size_t)r * HW];
                    if (p.ctx2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pre1[i][r] = p.ctx2[cb + (size_t)r * HW];
                    }
                }
            } else {
                pbias[2 * i] = *(const f32x4*)(p.bias + (cot0 + 2 * u) * 16 + 4 * kk);
                pbias[2 * i + 1] = *(const f32x4*)(p.bias + (cot0 + 2 * u + 1) * 16 + 4 * kk);
                if (p.mode == MODE_IAF || p.mode == MODE_INVERSE) {
                    const size_t zb = ((size_t)bimg * (p.cout >> 1) + ((cot0 + 2 * u) >> 1) * 16 + 4 * kk) * HW + pp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pre0[i][r] = p.zin[zb + (size_t)r * HW];
                }
            }
        }
    };

    // ================= K loop =============================================================================
    constexpr int NACC = (NT == 1) ? 2 : NT;   // a lone accumulator would serialise on the 40-cycle MFMA latency
    f32x4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (SHARED_W) {
        // ---- shared-weight K loop: one chunk per iteration, one barrier per chunk.
        //   LDS buffer (ci & 1) holds chunk ci; registers sr hold my share of chunk ci+1 (in flight since the middle
        //   of iteration ci-1).  Mid-iteration: sr -> other buffer (free: everyone left it at the last barrier),
        //   then fetch my share of chunk ci+2.
        const int nch = c_end - c_begin;
        // operands (x fragment + NT weight fragments) are double-buffered across taps: while tap tp's MFMAs run,
        // tap tp+1's ds_read_b128s are already in flight.  MODE: 0 = steady (park chunk+1, fetch chunk+2 in the
        // middle), 1 = park only, 2 = last chunk.
        f32x4 opx[2], opw[2][NT];
        auto read_ops = [&](int slot, int tp, int chunk, const f32x4* wb) {
            opx[slot] = smem4[xa[tp] + chunk * 4];
#pragma unroll
            for (int t = 0; t < NT; ++t) opw[slot][t] = wb[(tp * NT + t) * 64];
        };
        auto compute_chunk = [&](auto mode_c, int chunk, int buf) {
            constexpr int MODE = decltype(mode_c)::value;
            const f32x4* wb = wlds + buf * (NTILE_CH * 64) + lane;
            read_ops(0, 0, chunk, wb);
            static_for<NTP>([&](auto tp_c) {
                constexpr int tp = decltype(tp_c)::value;
                constexpr bool RD = (tp + 1 < NTP);
                constexpr bool WR = (tp == 2 && MODE <= 1);
                constexpr bool LDG = (tp == 2 && MODE == 0);
                if constexpr (RD) read_ops((tp + 1) & 1, tp + 1, chunk, wb);
                if constexpr (WR) write_stage(buf ^ 1);
                if constexpr (LDG) issue_stage(chunk + 2);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int a = (NT == 1) ? (j & 1) : t;
                        acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(opw[tp & 1][t][j], opx[tp & 1][j], acc[a], 0, 0, 0);
                    }
                sched_interleave<4 * NT, RD ? NT + 1 : 0, WR ? NLD : 0, LDG ? NLD : 0>();
                __builtin_amdgcn_sched_barrier(0);   // scheduling regions are per tap: nothing migrates across
            });
        };
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        using M2 = std::integral_constant<int, 2>;
        int ci = 0;
        for (; ci + 2 < nch; ++ci) {                 // steady state
            compute_chunk(M0{}, c_begin + ci, ci & 1);
            __syncthreads();
        }
        IAF_STAMP(3);
        prefetch_epilogue();
        if (ci + 1 < nch) {                          // second-to-last chunk: park the last one, nothing left to fetch
            compute_chunk(M1{}, c_begin + ci, ci & 1);
            __syncthreads();
            ++ci;
        }
        if (ci < nch) compute_chunk(M2{}, c_begin + ci, ci & 1);
    } else {
    // x operand: one ds_read_b128 per step, double-buffered one step ahead (xn is always the NEXT step's operand)
    f32x4 xn = smem4[xa[0] + c_begin * 4];
    auto chunk_body = [&](auto slot_c, auto refill_t4, auto refill_own, int chunk, auto ramp_c) {
        constexpr int I = decltype(slot_c)::value;
        constexpr bool RF_T4 = decltype(refill_t4)::value;    // step (chunk, 0) refills step (chunk+RCH-1, tap 4)
        constexpr bool RF_OWN = decltype(refill_own)::value;  // step (chunk, tp>=1) refills step (chunk+RCH, tp-1)
        constexpr bool RAMP = decltype(ramp_c)::value;        // first chunk of a DEEP ring: also fetch (chunk+1, tp) for tp < NTP-1
        static_for<NTP>([&](auto tp_c) {
            constexpr int tp = decltype(tp_c)::value;
            const f32x4 xv = xn;
            // next step: tap tp+1 of this chunk, or tap 0 of the next chunk (a read past the last chunk stays inside
            // the padded row of the LDS tile and is never used)
            xn = (tp + 1 < NTP) ? smem4[xa[(tp + 1) % NTP] + chunk * 4]
                                  : smem4[xa[0] + (chunk + 1 < c_end ? chunk + 1 : chunk) * 4];
            constexpr int PS = (I * NTP + tp + R - 1) % R;                  // slot consumed by the previous step
            constexpr bool rf = (tp == 0) ? RF_T4 : RF_OWN;
            const f32x4* q = wbase + ((size_t)chunk * NTP + tp + R - 1) * wstep;   // step s + R - 1
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int a = (NT == 1) ? (j & 1) : t;
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[I * NTP + tp][t][j], xv[j], acc[a], 0, 0, 0);
                }
            }
            if constexpr (rf) {   // refill of the slot the previous step consumed; spread between the MFMAs below
#pragma unroll
                for (int t = 0; t < NT; ++t) wr[PS][t] = q[ulane + t * 64];
            }
            constexpr bool rmp = RAMP && (tp + 1 < NTP);      // slot (1, NTP-1) is filled by step (chunk, 0)'s refill
            if constexpr (rmp) {
                const f32x4* q2 = wbase + ((size_t)(chunk + 1) * NTP + tp) * wstep;
#pragma unroll
                for (int t = 0; t < NT; ++t) wr[NTP + tp][t] = q2[ulane + t * 64];
            }
            sched_interleave<4 * NT, 1, 0, (rf ? NT : 0) + (rmp ? NT : 0)>();
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    {
        using T = std::true_type;
        using F = std::false_type;
        int c = c_begin;
        if constexpr (DEEP && RCH == 2) {
            if (ramp) {                                  // first ring revolution with the second chunk still arriving
                chunk_body(std::integral_constant<int, 0>{}, T{}, T{}, c, T{});
                chunk_body(std::integral_constant<int, 1>{}, T{}, T{}, c + 1, F{});
                c += RCH;
            }
        }
        for (; c + 2 * RCH <= c_end; c += RCH)       // steady state: every step refills
            static_for<RCH>([&](auto i) { chunk_body(i, T{}, T{}, c + decltype(i)::value, F{}); });
        IAF_STAMP(3);
        prefetch_epilogue();                         // >= one ring revolution of MFMA work left to hide it
        // drain: rem < 2*RCH chunks are left and chunk c sits in ring slot 0.  Which steps still refill depends only on
        // rem, so each value of rem gets its own STRAIGHT-LINE body (per-chunk `if`s here made hipcc emit vmcnt(0) and
        // expose a full memory latency -- and small problems spend their whole K loop in this code).
        const int rem = c_end - c;
        if constexpr (RCH <= 4) {
            static_for<2 * RCH>([&](auto rem_c) {
                constexpr int REM = decltype(rem_c)::value;
                if (rem == REM) {
                    static_for<REM>([&](auto i_c) {
                        constexpr int i = decltype(i_c)::value;
                        chunk_body(std::integral_constant<int, i % RCH>{}, std::bool_constant<(i + RCH - 1 < REM)>{},
                                   std::bool_constant<(i + RCH < REM)>{}, c + i, F{});
                    });
                }
            });
        } else {                                     // deep rings (NT == 1): code size wins, keep the generic form
            static_for<RCH>([&](auto i) {
                const int cc = c + decltype(i)::value;
                if (cc < c_end) {
                    if (cc + RCH < c_end) chunk_body(i, T{}, T{}, cc, F{});
                    else if (cc + RCH - 1 < c_end) chunk_body(i, T{}, F{}, cc, F{});
                    else chunk_body(i, F{}, F{}, cc, F{});
                }
            });
            static_for<RCH>([&](auto i) {
                const int cc = c + RCH + decltype(i)::value;
                if (cc < c_end) chunk_body(i, F{}, F{}, cc, F{});
            });
        }
    }
    }
    if (NT == 1) acc[0] += acc[1];
    IAF_STAMP(4);

    // ================= split-K reduction through LDS + epilogue ============================================
    // C/D layout of the 16x16 MFMA: lane holds D[row = 4*(l>>4)+r][col = l&15] = (co = tile*16 + 4*kk + r, pixel pl)
    f32x4 val[NMY * (ONE_TILE_UNITS ? 1 : 2)];
    if (KS > 1) {
        float* red = (float*)(smem4 + (size_t)(p.nslot + 1) * cp4);   // aliases the (now dead) shared-weight buffers
        if (SHARED_W) __syncthreads();                                // ... once every wave has left its K loop
        // layout [group = wave % WPK][k slice][tile][4][64 lanes]: conflict-free, every wave writes all its tiles
        float* wbuf = red + ((size_t)((wave % WPK) * KS + kh) * NT * 4) * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) wbuf[(t * 4 + j) * 64] = acc[t][j];
        __syncthreads();
        const float* rbuf = red + ((size_t)(wave % WPK) * KS * NT * 4) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NMY; ++i) {
            const int u = kh + i * KS;
            constexpr int TPU = ONE_TILE_UNITS ? 1 : 2;           // tiles per unit
#pragma unroll
            for (int e = 0; e < TPU; ++e) {
                f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
                if (u < NUNIT) {
                    const int t = u * TPU + e;
                    for (int k = 0; k < KS; ++k)
#pragma unroll
                        for (int j = 0; j < 4; ++j) sum[j] += rbuf[((size_t)(k * NT + t) * 4 + j) * 64];
                }
                val[i * TPU + e] = sum;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NT; ++i) val[i] = acc[i];
    }
    if (pvalid) {
#pragma unroll
        for (int i = 0; i < NMY; ++i) {
            const int u = kh + i * KS;
            if (u >= NUNIT) continue;
            if (EPI == EPI_PLAIN) {
                const int co = (cot0 + u) * 16 + 4 * kk;
                f32x4 v = val[i] + pbias[i];
                if (NTP == NTAPS && p.border) {   // a single Theano ar.conv2d (graphy/nodes/ar.py:200-375): border-indicator channel
#pragma unroll
                    for (int t = 1; t < NTP; ++t)
                        if (outside & (1u << t)) v += *(const f32x4*)(p.border + (size_t)(t - 1) * p.cout + co);
                }
                int c0 = 0, c1 = p.split_end[0];
                float* base = p.split_ptr[0];
#pragma unroll
                for (int q = 1; q < MAXSPLIT; ++q)      // static indices only: the descriptor stays in SGPRs
                    if (q < p.nsplit && co >= p.split_end[q - 1]) { c0 = p.split_end[q - 1]; c1 = p.split_end[q]; base = p.split_ptr[q]; }
                float* dst = base + ((size_t)bimg * (c1 - c0) + (co - c0)) * HW + pp;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(size_t)r * HW] = p.res ? pre0[i][r] + 0.1f * v[r] : v[r];
            } else if (EPI == EPI_DGRAD) {
                const int co = (cot0 + u) * 16 + 4 * kk;
                f32x4 v = val[i];
                if (p.mode == MODE_DGRAD_ELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= (pre0[i][r] > 0.f ? 1.f : pre0[i][r] + 1.f);
                    *(f32x4*)(p.y + (size_t)Pl * p.cout + co) = v;
                    if (p.out0) {
                        const size_t cb = ((size_t)bimg * p.cout + co) * HW + pp;
#pragma unroll
                        for (int r = 0; r < 4; ++r) p.out0[cb + (size_t)r * HW] = v[r];
                    }
                } else if (p.mode == MODE_DGRAD_PLAIN) {
                    if (p.zin) {       // the forward staged elu(x): xe > 0 <=> x > 0, elu'(x) = 1 or xe + 1
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] *= (pre0[i][r] > 0.f ? 1.f : pre0[i][r] + 1.f);
                    }
                    int c0 = 0, c1 = p.split_end[0];
                    float* base = p.split_ptr[0];
#pragma unroll
                    for (int q = 1; q < MAXSPLIT; ++q)
                        if (q < p.nsplit && co >= p.split_end[q - 1]) { c0 = p.split_end[q - 1]; c1 = p.split_end[q]; base = p.split_ptr[q]; }
                    float* dst = base + ((size_t)bimg * (c1 - c0) + (co - c0)) * HW + pp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(size_t)r * HW] = p.res ? pre1[i][r] + v[r] : v[r];
                } else {
                    const size_t cb = ((size_t)bimg * p.cout + co) * HW + pp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) p.out0[cb + (size_t)r * HW] = v[r] + pre0[i][r] * __expf(-pre1[i][r]);
                }
            } else if (EPI == EPI_HIDDEN) {
                const int co = (cot0 + u) * 16 + 4 * kk;
                f32x4 v = val[i] + pbias[i];
                if (NTP == NTAPS && p.border) {   // Theano pad_channel (conv.py:71-83, ar.py:229-233): taps that fall outside see a 1
#pragma unroll
                    for (int t = 1; t < NTP; ++t)
                        if (outside & (1u << t)) v += *(const f32x4*)(p.border + (size_t)(t - 1) * p.cout + co);
                }
                if (p.ctx) {   // x += context (layers.py:163-164); context = up_context + down_context (tf_train.py:58)
                    if (p.ctx2) v += (pre0[i] + pre1[i]);
                    else v += pre0[i];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = elu_f(v[r]);   // layers.py:165
                *(f32x4*)(p.y + (size_t)Pl * p.cout + co) = v;
            } else {
                const int nz = p.cout >> 1;
                const int gt = cot0 + 2 * u;              // packed tiles (gt, gt+1) = (mean, logsd) of channel group gt/2
                const int c0 = (gt >> 1) * 16 + 4 * kk;
                f32x4 bm = pbias[2 * i];
                f32x4 bs = pbias[2 * i + 1];
                if (NTP == NTAPS && p.border) {
#pragma unroll
                    for (int t = 1; t < NTP; ++t)
                        if (outside & (1u << t)) {
                            bm += *(const f32x4*)(p.border + (size_t)(t - 1) * p.cout + gt * 16 + 4 * kk);
                            bs += *(const f32x4*)(p.border + (size_t)(t - 1) * p.cout + (gt + 1) * 16 + 4 * kk);
                        }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t idx = ((size_t)bimg * nz + c0 + r) * HW + pp;
                    const float m_raw = val[2 * i][r] + bm[r];
                    const float s_raw = val[2 * i + 1][r] + bs[r];
                    if (p.mode == MODE_RAW) {
                        p.out0[idx] = m_raw;
                        p.out1[idx] = s_raw;
                    } else if (p.mode == MODE_IAF) {
                        const float m = m_raw * 0.1f, s = s_raw * 0.1f;        // tf_train.py:70
                        p.out0[idx] = (pre0[i][r] - m) / __expf(s);            // tf_train.py:71
                        p.out1[idx] = s;                                        // tf_train.py:72 (logqs += s)
                    } else if (p.mode == MODE_INVERSE) {
                        const float m = m_raw * 0.1f, s = s_raw * 0.1f;
                        p.out0[idx] = pre0[i][r] * __expf(s) + m;              // tf_train.py:71 solved for the input
                        p.out1[idx] = s;
                    } else {
                        const float m = m_raw * 0.1f, s = s_raw * 0.1f;
                        const float mean = p.qm[idx] + p.rm[idx];               // tf_train.py:57
                        const float logvar = 2.f * (p.ql[idx] + p.rl[idx]);
                        const float z0 = mean + __expf(0.5f * logvar) * p.eps[idx];                           // :63
                        const float d0 = z0 - mean;
                        float logqs = -0.5f * (1.8378770664093453f + logvar + d0 * d0 / __expf(logvar));     // :68
                        const float z = (z0 - m) / __expf(s);                                                 // :71
                        logqs += s;                                                                           // :72
                        const float plv = 2.f * p.pl[idx];                                                    // :56
                        const float d1 = z - p.pm[idx];
                        const float logps = -0.5f * (1.8378770664093453f + plv + d1 * d1 / __expf(plv));      // :73
                        p.out0[idx] = z;
                        if (p.out1) p.out1[idx] = s;
                        p.kl_elem[idx] = logqs - logps;                                                       // :75
                    }
                }
            }
        }
    }
    IAF_STAMP(5);
}

typedef void (*conv_fn_t)(ConvP);
