// conv_engine.hip.hpp — dense Conv1d as an implicit GEMM on the gfx950 f32 matrix cores.
//
//   out[b, co, t] = epi( bias[co] + sum_{ci,tap} W[co,ci,tap] * pro(x[b, ci, t + tap*dil - padL]) )
//
// GEMM view per utterance: M = Cout, N = T, K = Cin*taps.  v_mfma_f32_32x32x2_f32 computes a
// 32(co) x 32(t) tile per instruction with two k-values (an input-channel pair) — exact fp32 FMA
// chains, the arithmetic the reference's fp32 convs perform (only the summation order differs).
//
// Data movement: BOTH operands reach the matrix cores through LDS and are brought there by LDS-DMA
// (`global_load_lds`, no VGPR round trip), double-buffered per input-channel chunk:
//   A (weights)   : pre-packed at load time in MFMA lane order (model.cpp pack_conv); the
//                   [BM/32 blocks] x [K*CK/8 groups] x 1 KiB slab of a chunk is copied with 16-byte
//                   DMA pieces and read back with conflict-free lane-linear ds_read_b128.
//   B (activation): a [CK channels] x [BN + halo] tile, zero padding / sequence masking realised by
//                   redirecting out-of-range lanes to a zero page; read as 32-lane rows shifted per
//                   tap.  16-byte DMA pieces when rows are 16-byte aligned (T % 4 == 0), else 4-byte.
// Schedule per chunk: [wait own DMA] [barrier] [issue DMA for chunk+1] [MFMA over chunk from LDS].
//
// Measured facts this file is shaped by (tools/mfma_peak.hip, tools/conv_bench.py, MI355X):
//   * v_mfma_f32_32x32x2_f32 sustains 155 TFLOP/s alone, but it shares the SIMD's fp32 datapath with
//     VALU work: one ds_read2 + 6 VALU per 4 MFMAs already caps the loop at 130 TFLOP/s, whatever the
//     occupancy.  So the k-loop carries NO activation math: producers store pre-activated tensors
//     (epilogue `oslope`, optional second output), and the in-loop leaky-ReLU exists only as the ACT
//     template variant for callers that cannot arrange that.
//   * hipcc sinks LDS prefetches below the MFMA group and waits with lgkmcnt(0) -> LDS reads are
//     inline asm with our own waits, order pinned by sched_barrier.
//   * compiler-visible ds_reads are ordered against in-flight LDS-DMA by alias analysis; the two
//     pipeline stages are distinct __shared__ objects and the reads are asm, so the prefetch of the
//     next chunk is never drained early.
//
// One 256-thread workgroup = 4 wavefronts (64 lanes) arranged WM x WN, each owning MW x NW 32x32
// accumulator tiles.  Tile shapes (BM x BN): 128x128, 64x256, 32x512 for long sequences and 64x64,
// 32x128 for short ones (token-domain layers), chosen at pack time together with the chunk depth CK.
// Epilogues fuse bias, per-utterance conditioning bias, sequence mask, ReLU / leaky-ReLU, residual
// add, multi-receptive-field accumulation (/3), the coupling update, the transposed-conv pixel
// shuffle and an optional second (pre-activated) output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>

#include "sx_split.hip.hpp"

namespace vitsmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum : int {
    PRO_LRELU = 1,      // leaky-relu(slope) on the input (in-loop; prefer pre-activated inputs)
    PRO_MASK = 2,       // input * (t < len[b])
    EPI_RELU = 4,
    EPI_MASK = 8,       // value * (t < len[b])   (before the residual is added)
    EPI_RES = 16,       // + res[b,co,t]
    EPI_ACC = 32,       // out = out + value
    EPI_DIV = 64,       // out = value / div      (after EPI_ACC)
    EPI_COUPLING = 128,  // out = (out - value*mask) * mask   (modules.py:464, mean_only)
    // WN res_skip conv with its update folded in (modules.py:203-209): rows [0, wn_split) are the residual half,
    // out = (out + value * mask) [x = (x + res) * mask]; rows [wn_split, Cout) the skip half, out2 (+)= value * mask
    // (the sum of masked terms equals the reference's masked sum).  wn_split = 0: skip rows only (the last WN layer).
    EPI_WN = 256,
    EPI_WN_FIRST = 512,  // ... first WN layer: the skip rows are stored, not accumulated
    EPI_NO_PADFILL = 1 << 18,  // do not zero the columns between T and the row pitch (the output is a window into wider rows)
    DBG_NO_DMA = 1 << 16,  // ablation (tools/conv_bench.py): stop prefetching after the second chunk
    DBG_NO_EPI = 1 << 17   // ablation: skip the epilogue stores
};

struct ConvArgs {
    const float *x;
    int64_t x_bstride;  // floats between batch items of x
    int x_cstride;      // floats between channels of x (0 = T); > T means rows padded with ZEROS up to it
    int T;              // input (= virtual output) length
    const int *len;     // [B] valid lengths or nullptr
    const float *wp;    // packed weights
    const float *bias;  // [Cout] or nullptr
    const float *bias_b;  // per-batch bias [B][bias_b_stride] or nullptr
    int bias_b_stride;
    float *out;
    int64_t out_bstride;
    int out_cstride;      // floats between channels of out/out2/res (0 = T*ups); pad columns are written as 0
    float *out2;          // optional second output: leaky_relu(final value, oslope2); same layout as out
    const float *res;
    int64_t res_bstride;
    const float *zeros;   // >= 256 zero floats, 16-byte aligned (padding source for the DMA)
    int Cin, Cout, K, dil, padL, CK, nchunks, ups;
    int LW, padLa, xs_floats, stage_floats;  // filled by launch_conv
    unsigned magic;            // ceil(2^32 / LW)
    int flags;
    float slope, div;
    float oslope, oslope2;  // leaky-relu slope applied to the value stored in out / out2 (1 = none)
    int wn_split;           // EPI_WN: first skip row
    // optional third output: the value stored in `out` (rows [0, pl_rows)) once more as the two fp16 operand planes a
    // following split-operand conv reads (conv_sx_engine.hip.hpp: planes [3 slots][pl_rows/8][T][8], f16x3 format),
    // so that no separate split launch is needed; pl_rows % 32 == 0, ups == 1
    uint16_t *out_pl;
    int pl_rows;
    unsigned *peak;         // range-guard slots for those planes (SxArgs::peak), may be nullptr
};

template <int BYTES, int AUX = 0>
__device__ __forceinline__ void lds_dma(const void *gsrc, float *ldst) {
    // the size and cache-policy operands must be literals (not template-dependent): dispatch with if constexpr
    // (AUX = 2: `nt`, a streamed read that no other workgroup will ask for again)
    if constexpr (BYTES == 16 && AUX == 2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                         (__attribute__((address_space(3))) void *)ldst, 16, 0, 2);
    else if constexpr (BYTES == 16)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                         (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                         (__attribute__((address_space(3))) void *)ldst, 4, 0, 0);
}

__device__ __forceinline__ float lrelu_f(float v, float slope) {
    // leaky_relu(v) == med3(v, v*slope, +inf) for 0 < slope <= 1 (slope == 1: identity)
    return __builtin_amdgcn_fmed3f(v, v * slope, __builtin_inff());
}

// The two pipeline stages (x tile + A slab each) live in ONE dynamic LDS allocation whose size follows the
// layer (kernel width, chunk depth): narrow kernels take 2 x <= 26 KiB so three workgroups share a CU's
// 160 KiB, the widest (k = 11) take 2 x 38 KiB (two workgroups).  This is safe only because every LDS read of
// the main loop is inline asm: for compiler-visible reads hipcc could not prove the stages disjoint and would
// wait `vmcnt(0)` (drain the prefetch) in front of them.
// KS > 1 (token-domain layers): the workgroup's four waves SPLIT THE REDUCTION of one small output tile (MW x NW blocks,
// WM = WN = 1) instead of owning a block each: wave w takes the k-groups w, w + KS, .. of every chunk, the partial tiles
// meet in LDS and are summed in wave order (a fixed order: results do not depend on the batch), then wave e finishes
// block e.  Token-domain grids are short (B x 256 columns): as 64 x 64 tiles of four whole-K blocks the 768 -> 192, k = 3
// FFN conv of the encoder is 384 workgroups of 1152 dependent MFMAs per wave (1.5 per CU: 127 us at batch 32, 84 us at
// batch 1, 12 workgroups); as 32 x 64 tiles with the reduction split four ways it is 768 workgroups of 576.
template <int MW, int NW, int WM, int WN, int VEC, int ACT, int KS = 1>
__global__ __launch_bounds__(256, (MW * NW >= 4) ? 3 : 4) void conv_engine_kernel(ConvArgs a) {
    constexpr int BM = WM * MW * 32, BN = WN * NW * 32, MB = BM / 32;
    static_assert(WM * WN * KS == 4, "four waves per workgroup");
    static_assert(KS == 1 || MW * NW <= 4, "a wave per output block finishes the tile");
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    float *const stageP = lds_dyn;
    float *const stageQ = lds_dyn + a.stage_floats;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = KS > 1 ? 0 : wave / WN, wn = KS > 1 ? 0 : wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * BN;
    const int T = a.T;
    const int len_b = a.len ? a.len[b] : T;
    const int in_lim = (a.flags & PRO_MASK) ? (len_b < T ? len_b : T) : T;
    const int LW = a.LW, CK = a.CK, K = a.K;
    const int spc = (K * CK) >> 3;  // float4 groups per 32-row block per chunk
    const int XS = a.xs_floats;     // x tile floats (padded to the DMA piece); the A slab follows it
    const float *xb = a.x + (int64_t)b * a.x_bstride;
    const int mblk_base = blockIdx.y * MB;
    const int nxs = XS / (256 * VEC);
    // A slab of (m-tile, chunk): MB*spc KiB, CONTIGUOUS in the packed weights (model.cpp pack_conv)
    const float4 *aslab0 = reinterpret_cast<const float4 *>(a.wp) + (int64_t)blockIdx.y * a.nchunks * (MB * spc * 64) + lane;
    const int na = MB * spc;

    // Which element of the [CK rows][LW columns] x tile a lane fetches in DMA round s, and whether its column lies inside
    // the sequence, depends on the tile only; with every chunk full (Cin % CK == 0) the chunk just moves a scalar base.
    // So the per-lane element offsets are computed once (instead of a umulhi / 64-bit multiply chain in front of every
    // DMA: most layers on this engine have K loops too short to hide that behind their few MFMAs).
    constexpr int MAXS = 4;
    const bool fastx = nxs <= MAXS && a.Cin % CK == 0;
    int xoff[MAXS];
    bool xok[MAXS];
    if (fastx) {
#pragma unroll
        for (int s = 0; s < MAXS; s++) {
            const int e = (s * 256 + tid) * VEC;
            const int r = (int)__umulhi((unsigned)e, a.magic);
            const int c = e - r * LW;
            const int t = t0 - a.padLa + c;
            xok[s] = s < nxs && r < CK && t >= 0 && t < in_lim;
            xoff[s] = r * a.x_cstride + t;  // (launch_conv keeps CK * x_cstride inside 31 bits)
        }
    }

    auto issue = [&](int chunk, float *stage) {
        const float4 *src = aslab0 + (int64_t)chunk * (na * 64);
        float *ab = stage + XS;
        for (int i = wave; i < na; i += 4) lds_dma<16>(src + i * 64, ab + i * 256);
        if (fastx) {
            const float *xc = xb + (int64_t)chunk * CK * a.x_cstride;
#pragma unroll
            for (int s = 0; s < MAXS; s++) {
                if (s < nxs) {
                    const float *src_x = xok[s] ? xc + xoff[s] : a.zeros + lane * VEC;
                    lds_dma<VEC * 4>(src_x, stage + (s * 256 + wave * 64) * VEC);
                }
            }
            return;
        }
        // x tile: CK rows x LW columns, linear in LDS; out-of-range lanes read the zero page
        const int rows_valid = (a.Cin - chunk * CK) < CK ? (a.Cin - chunk * CK) : CK;
        const float *xc = xb + (int64_t)chunk * CK * a.x_cstride;
        for (int s = 0; s < nxs; s++) {
            const int e = (s * 256 + tid) * VEC;
            const int r = (int)__umulhi((unsigned)e, a.magic);
            const int c = e - r * LW;
            const int t = t0 - a.padLa + c;
            const bool ok = r < rows_valid && t >= 0 && t < in_lim;
            const float *src_x = ok ? xc + (int64_t)r * a.x_cstride + t : a.zeros + lane * VEC;
            lds_dma<VEC * 4>(src_x, stage + (s * 256 + wave * 64) * VEC);
        }
    };

    f32x16 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; m++)
#pragma unroll
        for (int n = 0; n < NW; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    const float slope = a.slope;
    const int g_per_tap = CK >> 3;
    const int col0 = (a.padLa - a.padL) + wn * (NW * 32) + l31;

    // ---- MFMA loop over one chunk, reading both operands from LDS with inline-asm ds_reads.
    // Order per k-step (pinned with sched_barrier): [wait for step s] [issue prefetch of step s+1]
    // [MW*NW MFMAs]; the prefetch latency hides behind the 256-cycle MFMA group.
    const uint32_t a_byte0 = (uint32_t)(XS + ((wm * MW) * spc * 64 + lane) * 4) * 4u;
    const int ngroups = K * g_per_tap;  // groups of 4 k-steps (8 input channels of one tap)
    const uint32_t lw2b = (uint32_t)(2 * LW) * 4u;
    auto compute = [&](const float *stage) {
        const uint32_t sbase = (uint32_t)(uintptr_t)stage;  // LDS byte offset of this stage
        auto read_a = [&](int m, int grp) {
            f32x4 r;
            const uint32_t ad = sbase + a_byte0 + (uint32_t)((m * spc + grp) * 1024);
            asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(ad) : "memory");
            return r;
        };
        auto read_b = [&](uint32_t ad, float *dst) {
            if constexpr (NW == 1) {
                float r;
                asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(ad) : "memory");
                dst[0] = r;
            } else {
                f32x2 r;
                asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(r) : "v"(ad) : "memory");
                dst[0] = r.x;
                dst[1] = r.y;
                if constexpr (NW == 4) {
                    f32x2 q;
                    asm volatile("ds_read2_b32 %0, %1 offset0:64 offset1:96" : "=v"(q) : "v"(ad) : "memory");
                    dst[2] = q.x;
                    dst[3] = q.y;
                }
            }
        };
        int tap = 0, g = 0;
        uint32_t bbyte = sbase + (uint32_t)(hi * LW + col0) * 4u;  // B(tap, g, j = 0, n = 0) of this lane
        f32x4 av_n[MW];
        float bv_n[NW];
#pragma unroll
        for (int m = 0; m < MW; m++) av_n[m] = read_a(m, 0);
        read_b(bbyte, bv_n);
        for (int gi = 0; gi < ngroups; gi++) {
            // next group's coordinates; the last group prefetches itself again (no branch around a read)
            int g2 = g + 1, tap2 = tap;
            if (g2 == g_per_tap) { g2 = 0; tap2 = tap + 1; }
            const bool more = gi + 1 < ngroups;
            const uint32_t bbyte2 = more ? sbase + (uint32_t)((g2 * 8 + hi) * LW + tap2 * a.dil + col0) * 4u : bbyte;
            const int gnext = more ? gi + 1 : gi;
            f32x4 av[MW];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float bv[NW];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments of step j have landed
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0) {
#pragma unroll
                    for (int m = 0; m < MW; m++) av[m] = av_n[m];
                }
#pragma unroll
                for (int n = 0; n < NW; n++) bv[n] = bv_n[n];
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0) {
#pragma unroll
                    for (int m = 0; m < MW; m++) av_n[m] = read_a(m, gnext);
                }
                read_b(j < 3 ? bbyte + (uint32_t)(j + 1) * lw2b : bbyte2, bv_n);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ACT) {
#pragma unroll
                    for (int n = 0; n < NW; n++) bv[n] = lrelu_f(bv[n], slope);
                }
#pragma unroll
                for (int m = 0; m < MW; m++) {
                    const float aval = j == 0 ? av[m].x : (j == 1 ? av[m].y : (j == 2 ? av[m].z : av[m].w));
#pragma unroll
                    for (int n = 0; n < NW; n++)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bv[n], acc[m][n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            g = g2;
            tap = tap2;
            bbyte = bbyte2;
        }
        // drain the self-prefetch of the last group before its destination registers die
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- the same loop for one wave's share of the k-groups (KS > 1): groups wave, wave + KS, ..
    auto compute_ks = [&](const float *stage) {
        const uint32_t sbase = (uint32_t)(uintptr_t)stage;
        auto read_a = [&](int m, int grp) {
            f32x4 r;
            const uint32_t ad = sbase + a_byte0 + (uint32_t)((m * spc + grp) * 1024);
            asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(ad) : "memory");
            return r;
        };
        auto read_b = [&](uint32_t ad, float *dst) {
            if constexpr (NW == 1) {
                float r;
                asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(ad) : "memory");
                dst[0] = r;
            } else {
                f32x2 r;
                asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(r) : "v"(ad) : "memory");
                dst[0] = r.x;
                dst[1] = r.y;
                if constexpr (NW == 4) {
                    f32x2 q;
                    asm volatile("ds_read2_b32 %0, %1 offset0:64 offset1:96" : "=v"(q) : "v"(ad) : "memory");
                    dst[2] = q.x;
                    dst[3] = q.y;
                }
            }
        };
        // group gi = (tap, g): g_per_tap = CK / 8 is a power of two
        const int gsh = g_per_tap == 4 ? 2 : (g_per_tap == 2 ? 1 : 0);
        auto baddr = [&](int gi) {
            const int tp = gi >> gsh, g = gi & (g_per_tap - 1);
            return sbase + (uint32_t)((g * 8 + hi) * LW + tp * a.dil + col0) * 4u;
        };
        int gi = wave;
        if (gi >= ngroups) return;  // (uniform per wave: a chunk with fewer groups than waves)
        uint32_t bbyte = baddr(gi);
        f32x4 av_n[MW];
        float bv_n[NW];
#pragma unroll
        for (int m = 0; m < MW; m++) av_n[m] = read_a(m, gi);
        read_b(bbyte, bv_n);
        for (; gi < ngroups; gi += KS) {
            const bool more = gi + KS < ngroups;
            const int gnext = more ? gi + KS : gi;
            const uint32_t bbyte2 = more ? baddr(gnext) : bbyte;
            f32x4 av[MW];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float bv[NW];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0) {
#pragma unroll
                    for (int m = 0; m < MW; m++) av[m] = av_n[m];
                }
#pragma unroll
                for (int n = 0; n < NW; n++) bv[n] = bv_n[n];
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0) {
#pragma unroll
                    for (int m = 0; m < MW; m++) av_n[m] = read_a(m, gnext);
                }
                read_b(j < 3 ? bbyte + (uint32_t)(j + 1) * lw2b : bbyte2, bv_n);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ACT) {
#pragma unroll
                    for (int n = 0; n < NW; n++) bv[n] = lrelu_f(bv[n], slope);
                }
#pragma unroll
                for (int m = 0; m < MW; m++) {
                    const float aval = j == 0 ? av[m].x : (j == 1 ? av[m].y : (j == 2 ? av[m].z : av[m].w));
#pragma unroll
                    for (int n = 0; n < NW; n++)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bv[n], acc[m][n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            bbyte = bbyte2;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- per-lane epilogue constants, computed up front so their loads hide behind the main loop.
    // C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    const int flags = a.flags;
    const int ups = a.ups;
    const int Tout = T * ups;
    float *ob = a.out + (int64_t)b * a.out_bstride;
    const float *rb = a.res ? a.res + (int64_t)b * a.res_bstride : a.zeros;
    const float *biasp = a.bias ? a.bias : a.zeros;
    const float *bbp = a.bias_b ? a.bias_b + (int64_t)b * a.bias_b_stride : a.zeros;
    const int bb_on = a.bias_b ? 1 : 0, b_on = a.bias ? 1 : 0;
    // KS > 1: after the reduction wave e finishes block e = (e / NW, e % NW) of the tile as a one-block wave
    constexpr int EM = KS > 1 ? 1 : MW, EN = KS > 1 ? 1 : NW;  // blocks per wave in the epilogue
    const bool e_live = KS == 1 || wave < MW * NW;
    const int e_m = (KS > 1 && e_live) ? wave / NW : 0, e_n = (KS > 1 && e_live) ? wave % NW : 0;
    const int mblk0 = mblk_base + wm * MW + e_m;
    // element offset of (co, t = 0) inside the utterance, recomputed where needed (2 VALU ops) rather than
    // kept in 32 registers across the main loop; -1 = row out of range
    auto row_off = [&](int m, int r) -> int {
        const int co = (mblk0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (co >= a.Cout) return -1;
        return ups == 1 ? co * a.out_cstride : (co / ups) * Tout + (co % ups);
    };
    // software pipeline over input-channel chunks, two stages, one barrier per chunk
    const int nchunks = a.nchunks;
    const bool dbg_nodma = a.flags & DBG_NO_DMA;
    issue(0, stageP);
    // (no exit from the middle of the unrolled pair: a mid-loop break makes hipcc copy the accumulators)
    auto run = [&](const float *stage) {
        if constexpr (KS > 1) compute_ks(stage);
        else compute(stage);
    };
    for (int chunk = 0; chunk + 1 < nchunks; chunk += 2) {
        __syncthreads();  // own DMA drained (vmcnt(0)) + everyone done reading stageQ
        if (!(dbg_nodma && chunk > 0)) issue(chunk + 1, stageQ);
        run(stageP);
        __syncthreads();
        if (chunk + 2 < nchunks && !dbg_nodma) issue(chunk + 2, stageP);
        run(stageQ);
    }
    if (nchunks & 1) {
        __syncthreads();
        run(stageP);
    }
    // ---- KS > 1: the four partial tiles meet in LDS (the stages are dead) and are summed in wave order; wave e then
    // owns block e in acc[0][0]
    if constexpr (KS > 1) {
        constexpr int NBLK = MW * NW;
        __syncthreads();
        float *part = lds_dyn;  // [wave][block][register][lane]
#pragma unroll
        for (int m = 0; m < MW; m++)
#pragma unroll
            for (int n = 0; n < NW; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) part[((wave * NBLK + m * NW + n) * 16 + r) * 64 + lane] = acc[m][n][r];
        __syncthreads();
        const int eb = e_live ? wave : 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float sum = part[((0 * NBLK + eb) * 16 + r) * 64 + lane];
#pragma unroll
            for (int w = 1; w < KS; w++) sum += part[((w * NBLK + eb) * 16 + r) * 64 + lane];
            acc[0][0][r] = sum;
        }
    }

    // ---- epilogue: branch-free per element (absent bias pointers read the zero page, flags become
    // multipliers / clamps; a per-element "load or not" branch makes hipcc wait vmcnt(0) per element).
    // The residual / accumulate operands of all NW tiles of a block row are requested together.
    if (flags & DBG_NO_EPI) {  // ablation only: keep the accumulators live, skip the stores
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < MW; m++)
#pragma unroll
            for (int n = 0; n < NW; n++) s += acc[m][n][0] + acc[m][n][7] + acc[m][n][15];
        if (s == 12345.678f) a.out[tid] = s;
        return;
    }
    float *ob2 = a.out2 ? a.out2 + (int64_t)b * a.out_bstride : nullptr;
    const bool is_wn = (flags & EPI_WN) != 0;
    uint16_t *plb = a.out_pl ? a.out_pl + (int64_t)b * 3 * a.pl_rows * T : nullptr;
    const int64_t plane_elems = (int64_t)a.pl_rows * T;
    float pk = 0.f;
    const float relu_floor = (flags & EPI_RELU) ? 0.f : -__builtin_inff();
    const float oslope = a.oslope, oslope2 = a.oslope2, div = a.div;
    // second operand added to the value: the residual tensor, or (EPI_ACC / EPI_COUPLING) the old output
    const float *addp = (flags & EPI_RES) ? rb : ob;
    const bool has_add = flags & (EPI_RES | EPI_ACC | EPI_COUPLING);
#pragma unroll
    for (int m = 0; m < EM; m++) {
        if (!e_live) break;  // (KS > 1: the waves beyond the tile's blocks only took part in the reduction)
        int orow_m[16];
        float brow_m[16];
        // EPI_WN: this 32-row block is either residual rows (-> out, added to it) or skip rows (-> out2)
        const bool skip_rows = is_wn && (mblk0 + m) * 32 >= a.wn_split;
        float *obm = skip_rows ? ob2 : ob;
        const float *addm = is_wn ? obm : addp;
        const bool has_add_m = is_wn ? !(skip_rows && (flags & EPI_WN_FIRST)) : has_add;
        const int co_shift = skip_rows ? a.wn_split : 0;
        const bool planes_m = plb && !skip_rows && (mblk0 + m) * 32 < a.pl_rows;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            orow_m[r] = row_off(m, r);
            if (is_wn && orow_m[r] >= 0) orow_m[r] -= co_shift * a.out_cstride;
            const int co = (mblk0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int cc = co < a.Cout ? co : 0;
            brow_m[r] = biasp[cc * b_on] + bbp[cc * bb_on];
        }
        constexpr int NB = EN >= 4 ? 2 : EN;  // tiles whose add-operands are in flight together (register budget)
        float ad[NB][16];
#pragma unroll
        for (int n = 0; n < EN; n++) {
            if (has_add_m && (n % NB) == 0) {
#pragma unroll
                for (int q = 0; q < NB; q++) {
                    const int tq = t0 + wn * (NW * 32) + (n + q + e_n) * 32 + l31;
                    const int ttq = tq * ups;
#pragma unroll
                    for (int r = 0; r < 16; r++) ad[q][r] = (orow_m[r] >= 0 && tq < T) ? addm[orow_m[r] + ttq] : 0.f;
                }
            }
            const int t = t0 + wn * (NW * 32) + (n + e_n) * 32 + l31;
            if (t >= T) {
                if (ups == 1 && t < a.out_cstride && !(flags & EPI_NO_PADFILL)) {  // row padding up to the pitch: zeros (see x_cstride)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        if (orow_m[r] >= 0) {
                            obm[orow_m[r] + t] = 0.f;
                            if (ob2 && !is_wn) ob2[orow_m[r] + t] = 0.f;
                        }
                }
                continue;
            }
            const float mk = (t < len_b) ? 1.f : 0.f;
            const float mk_sel = (flags & EPI_MASK) ? mk : 1.f;
            const int tt = t * ups;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = fmaxf(acc[m][n][r] + brow_m[r], relu_floor);
            if (flags & EPI_COUPLING) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (orow_m[r] >= 0) ob[orow_m[r] + tt] = (ad[n % NB][r] - v[r] * mk) * mk;
                continue;
            }
            if (has_add_m) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = v[r] * mk_sel + ad[n % NB][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] *= mk_sel;
            }
            if ((flags & EPI_RES) && (flags & EPI_ACC)) {  // residual AND accumulate: second fetch (MRF tail)
                float oo[16];
#pragma unroll
                for (int r = 0; r < 16; r++) oo[r] = orow_m[r] >= 0 ? ob[orow_m[r] + tt] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] += oo[r];
            }
            if (flags & EPI_DIV) {
#pragma unroll
                for (int r = 0; r < 16; r++) v[r] = v[r] / div;
            }
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (orow_m[r] >= 0) obm[orow_m[r] + tt] = lrelu_f(v[r], oslope);
            if (planes_m) {
                // register quad q = 4 consecutive channels 8q + 4*hi .. +3 of channel group (block * 4 + q): half a cell
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    unsigned wa[2], wb[2];
                    split2h_pair_pk(lrelu_f(v[4 * q], oslope), lrelu_f(v[4 * q + 1], oslope), wa[0], wa[1], pk);
                    split2h_pair_pk(lrelu_f(v[4 * q + 2], oslope), lrelu_f(v[4 * q + 3], oslope), wb[0], wb[1], pk);
                    const int64_t cell = ((int64_t)((mblk0 + m) * 4 + q) * T + t) * 8 + 4 * hi;
                    *reinterpret_cast<u32x2 *>(plb + cell) = u32x2{wa[0], wb[0]};
                    *reinterpret_cast<u32x2 *>(plb + plane_elems + cell) = u32x2{wa[1], wb[1]};
                }
            }
            if (ob2 && !is_wn) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (orow_m[r] >= 0) ob2[orow_m[r] + tt] = lrelu_f(v[r], oslope2);
            }
        }
    }
    if (a.peak) sx_publish_peak(a.peak, (int)(blockIdx.x + blockIdx.y + blockIdx.z), pk);  // (uniform; every thread arrives)
}

// tile configs: index -> (BM, BN)
//   0: 32x512   1: 64x256   2: 128x128   3: 64x64   4: 32x128   5: 32x64, the four waves split the reduction (KS = 4)
inline int conv_tile_m(int cfg) { return cfg == 2 ? 128 : ((cfg == 1 || cfg == 3) ? 64 : 32); }
inline int conv_tile_n(int cfg) {
    switch (cfg) {
        case 0: return 512;
        case 1: return 256;
        case 2: return 128;
        case 3: return 64;
        case 5: return 64;
        default: return 128;
    }
}

// largest pipeline stage (floats) a layer may use: 38 KiB (two workgroups per CU)
inline int conv_stage_floats(int cfg) {
    static const int cap3 = [] {
        const char *e = std::getenv("VITSMI_STAGE_CAP_SMALL");  // tuning experiments only (model.cpp stage_capacity)
        return e ? std::atoi(e) : 4864;
    }();
    if (cfg == 5) return 6144;
    return cfg <= 2 ? 9728 : cap3;
}

template <int MW, int NW, int WM, int WN, int VEC, int ACT, int KS = 1>
inline hipError_t launch_conv_k(const ConvArgs &a, dim3 grid, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};  // allow > 64 KiB of dynamic LDS, once per (instantiation, device)
    auto kern = conv_engine_kernel<MW, NW, WM, WN, VEC, ACT, KS>;
    if (hipError_t e = sx_allow_big_lds(reinterpret_cast<const void *>(kern), attr_done); e != hipSuccess) return e;
    if (g_launch_name_on)
        snprintf(g_launch_name, sizeof g_launch_name, "conv_engine_kernel<%d, %d, %d, %d, %d, %d, %d>", MW, NW, WM, WN, VEC, ACT, KS);
    kern<<<grid, 256, lds, stream>>>(a);
    return hipGetLastError();
}

template <int MW, int NW, int WM, int WN, int KS = 1>
inline hipError_t launch_conv_t(const ConvArgs &a, dim3 grid, bool vec4, bool act, size_t lds, hipStream_t stream) {
    if (vec4) return act ? launch_conv_k<MW, NW, WM, WN, 4, 1, KS>(a, grid, lds, stream) : launch_conv_k<MW, NW, WM, WN, 4, 0, KS>(a, grid, lds, stream);
    return act ? launch_conv_k<MW, NW, WM, WN, 1, 1, KS>(a, grid, lds, stream) : launch_conv_k<MW, NW, WM, WN, 1, 0, KS>(a, grid, lds, stream);
}

// Launch on `stream`; LW / padLa / xs_floats / magic are filled in here.
hipError_t launch_conv(ConvArgs a, int cfg, int B, hipStream_t stream);
#ifdef VITSMI_IMPL_CONV_F32  // (tu_conv_f32.hip: the f32 engine's instantiations are one translation unit)
hipError_t launch_conv(ConvArgs a, int cfg, int B, hipStream_t stream) {
    const int BN = conv_tile_n(cfg), BM = conv_tile_m(cfg);
    const int halo = (a.K - 1) * a.dil;
    // 16-byte DMA needs 16-byte aligned rows: T % 4 == 0, aligned base/batch stride, no ragged input mask
    if (a.x_cstride == 0) a.x_cstride = a.T;
    if (a.out_cstride == 0) a.out_cstride = a.T * a.ups;
    // (rows may be padded with zeros up to a pitch that is a multiple of 4: x_cstride)
    const bool vec4 = (a.x_cstride % 4 == 0) && (a.x_bstride % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0) &&
                      !(a.flags & PRO_MASK);
    if (vec4) {
        a.padLa = (a.padL + 3) & ~3;
        const int padRa = (halo - a.padL + 3) & ~3;
        a.LW = BN + a.padLa + padRa;
        a.xs_floats = (a.CK * a.LW + 1023) / 1024 * 1024;
    } else {
        a.padLa = a.padL;
        a.LW = BN + halo;
        a.xs_floats = (a.CK * a.LW + 255) / 256 * 256;
    }
    a.magic = (unsigned)((0x100000000ull + a.LW - 1) / a.LW);
    if (a.oslope == 0.f) a.oslope = 1.f;
    if (a.oslope2 == 0.f) a.oslope2 = 1.f;
    const bool act = (a.flags & PRO_LRELU) && a.slope != 1.f;
    dim3 grid((a.T + BN - 1) / BN, (a.Cout + BM - 1) / BM, B);
    if (grid.x == 0 || grid.y == 0 || B == 0) return hipSuccess;
    const size_t stage = (size_t)a.xs_floats + (size_t)(BM / 32) * (a.K * a.CK / 8) * 256;
    if (stage > (size_t)conv_stage_floats(cfg)) return hipErrorInvalidValue;  // pick_tiling guarantees this never fires
    a.stage_floats = (int)((stage + 63) / 64 * 64);
    size_t lds = 2 * (size_t)a.stage_floats * sizeof(float);
    if (cfg == 5 && lds < (size_t)4 * 2 * 16 * 64 * 4) lds = (size_t)4 * 2 * 16 * 64 * 4;  // the four partial 32 x 64 tiles
    if ((int64_t)a.Cout * (a.ups == 1 ? a.out_cstride : a.T) >= (int64_t)1 << 31) return hipErrorInvalidValue;  // 32-bit element offsets per utterance
    if ((int64_t)a.CK * a.x_cstride + a.T >= (int64_t)1 << 31) return hipErrorInvalidValue;  // ... and inside a chunk of the input
    switch (cfg) {
        case 0: return launch_conv_t<1, 4, 1, 4>(a, grid, vec4, act, lds, stream);
        case 1: return launch_conv_t<2, 2, 1, 4>(a, grid, vec4, act, lds, stream);
        case 2: return launch_conv_t<2, 2, 2, 2>(a, grid, vec4, act, lds, stream);
        case 3: return launch_conv_t<1, 1, 2, 2>(a, grid, vec4, act, lds, stream);
        case 5: return launch_conv_t<1, 2, 1, 1, 4>(a, grid, vec4, act, lds, stream);
        default: return launch_conv_t<1, 1, 1, 4>(a, grid, vec4, act, lds, stream);
    }
}
#endif  // VITSMI_IMPL_CONV_F32

}  // namespace vitsmi
