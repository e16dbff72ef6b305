// conv_sx_engine.hip.hpp — dense Conv1d as an implicit GEMM on the gfx950 16-bit matrix cores with fp32
// operands and results: every fp32 operand is carried as a few 16-bit PLANES whose products the MFMA forms
// exactly and accumulates in fp32 ("split operands", sx).  Two arithmetics (template parameter NP):
//   f16x3  (NP = 2, the default): two fp16 planes per operand, v ~ h0 + h1, and the three products
//          h0g0 + h0g1 + h1g0 per fp32 product; per-product error ~3 * 2^-24, i.e. one fp32 rounding.  The
//          range handling (per-tensor weight scale, low activation plane stored 2^11 up) is described at
//          split2h_pair below.  Ceiling 2516.6 / 3 = 839 TFLOP/s fp32-equivalent.
//   bf16x6 (NP = 6): three bf16 planes, v = p0 + p1 + p2 exactly (3 x 8 = 24 bits), and the six plane products
//          of combined order <= 2, w0x0 + w0x1 + w1x0 + w0x2 + w1x1 + w2x0; the dropped terms are below
//          3 * 2^-24 |w||x|.  Ceiling 2516.6 / 6 = 419 TFLOP/s (tools/mfma_bf16x6_probe.hip sustains 340-375 with
//          both operands streaming from LDS).
//   f16    (NP = 1): ONE fp16 plane per operand and one product: the reduced-precision vocoder of BASELINE config 4
//          ("bf16 vocoder": 16-bit storage, fp32 accumulation).  Activations are stored as the fp16 of the consumer's
//          leaky-ReLU only (2 bytes per element instead of raw fp32 + two planes = 8); a residual is recovered from that
//          plane by undoing the leaky-ReLU (exact); the multi-receptive-field sum stays fp32.  16x16x32 loop only.
// Both full-precision modes carry the error bound of an fp32 FMA chain (what the reference's fp32 convolutions
// and the f32 engine in conv_engine.hip.hpp compute); tests/test_gpu_parity.py checks them against float64.
// Why: the f32 matrix pipe peaks at 157 TFLOP/s (measured 155), the f16 / bf16 pipe at 2.5 PFLOP/s.
//
// Layouts (T = time steps of the tensor, C % 16 == 0 on inputs, C % 32 == 0 on outputs):
//   planes  16-bit [3 slots][C/8][T][8]   conv inputs; one 16-byte cell = 8 channels of one time step, which is
//                                 exactly one lane's B operand (8 k-values) of the MFMA (f16x3 uses 2 slots)
//   raw     fp32 [C/8][T][8]      residual stream (same cell structure, 32-byte cells)
//   weights 16-bit [m-tile][chunk of 16 ci][tap][32-row block][plane][lane][8]  (model.cpp pack_conv_sx; 3 planes,
//                                 f16x3: 2): MFMA A-operand lane order, one 1 KiB wave-load per (block row, plane)
// Tensors of <= 64 channels are HBM-bound layers: they are kept as raw only and the consuming conv (RAWIN
// instantiation) applies the leaky-ReLU and the split while it loads its tile.
// Pipeline: one step = one tap of one 16-channel chunk = 3*MW*NW MFMAs per wave (bf16x6: 6*MW*NW).
//   A (weights): global_load straight into registers (L2-resident, no LDS), two register sets, one step ahead.
//   B (x tile):  [planes][2 channel-group halves][BN + halo] cells in LDS, double-buffered per chunk, brought
//                by 16-byte LDS-DMA (f16x3: per-tile lane offsets and masks, padding cells zeroed once; else
//                out-of-range cells read a zero page) or, RAWIN, through registers; one barrier per chunk; B
//                fragments are re-read in two halves, half a step ahead.
//   Waits are counted (`vmcnt(n)` / `lgkmcnt(n)`, raw `s_barrier`): a `__syncthreads()` would drain the
//   prefetches in flight.
// Epilogue: bias, per-utterance bias, residual, multi-receptive-field accumulate and /n, leaky-ReLU,
// pixel shuffle of the transposed conv (virtual rows are r-major: row = r*Cr + co), then an fp32 raw
// store and/or a split into the planes the next conv reads; the common combinations are compile-time
// specialisations (EPI template parameter).
// What shaped it (tools/conv_bench.py --sx --prof, profiles/): LDS-DMA and global loads cost 60-180 issue
// cycles per 1 KiB wave-instruction next to a saturated matrix pipe, so the design minimises their count per
// MFMA (256-column tiles: one A load serves 4 block columns); a mid-loop `break` in the unrolled step pair
// makes hipcc copy all accumulators every step; at this load the chip runs at 1.7-1.9 GHz (power), so the
// matrix pipe's busy share (55-63 % on the 128-row tiles), not the nominal 2.4 GHz peak, is the honest gauge.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>

#include "conv_engine.hip.hpp"

#ifndef SX_CFG0_WIDE
#define SX_CFG0_WIDE 0  // 128 x 256 tile as four 32-row waves x 256 columns (half the weight bytes per workgroup)
#endif
#ifndef SX16_SPREAD
#define SX16_SPREAD 0  // (experiment) x-tile DMA rounds spread behind the quarters of the chunk-opening half-step
#endif
#ifndef SX16_PRIO
#define SX16_PRIO 0  // (experiment) raised wave priority around each quarter's MFMAs
#endif
#ifndef SX16_ABL
#define SX16_ABL 0  // timing-only ablations of the 16x16x32 loop (wrong results): 1 no MFMAs, 2 no B reads, 4 no chunk barrier
#endif
#ifndef SX16_NA_SMALL
#define SX16_NA_SMALL 2  // weight register sets of the 16x16x32 loop's tiles with 32 x 64 outputs per wave (see half_step); 3 = two
                         // half-steps of look-ahead: parity-tested, measured equal at batch 1 and 32 (the weight fetch is not what
                         // a step of these tiles waits for)
#endif
#ifndef SX16_WIDE
#define SX16_WIDE 1  // 16x16x32 loop, 128 x 256 tile as four waves of 32 rows x 256 columns: each wave streams its own weight rows (half
                     // the L2 -> CU weight bytes of the 2 x 2 arrangement, a whole step of lead), the x tile is read once per wave
                     // instead of twice per two waves (same LDS traffic): 3-4 % faster on every 128-row shape (r03i)
#endif
#ifndef SX_NOA
#define SX_NOA 0  // ablation: weights are not re-fetched after the first steps (wrong results, timing only)
#endif
#ifndef SX_XCD_GROUP_DEFAULT
// (same-box A/B, two interleaved rounds, profiles/r06_runs/xcd_group.txt: groups of 1 / 2 / 4 / 8 / 16 tiles - headline voice
// 140.7 / 140.9 / 141.2 / 140.8 / 140.6 M samples/s, default voice 806.8 / 811.4 / 812.9 / 813.7 / 811.9 M)
#define SX_XCD_GROUP_DEFAULT 4
#endif
namespace vitsmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum : int { SX_NO_RAW_STORE = 256 };  // out_raw is only the EPI_ACC operand, not a destination

// Ragged batches (the generator: models.py:348-368 has no mask, so the reference graph renders every utterance of a padded
// batch to the longest one's length).  Utterance b's tensors END at min(T, (len[b] + add) * mul) columns: a workgroup whose
// tile starts behind that end exits at once, columns behind it are zero padding for the loads and are not stored.  `add` is
// the generator's receptive field in frames (Model::gen_rf_frames), so every VALID sample (frame < len[b]) still sees exactly
// the values the padded rendering computes - the chunked renderer (vitsmi.hip render_chunks) relies on the same argument.
// len == nullptr: every utterance spans T (the padded rendering).
struct SxRagged {
    const int *len;       // valid frames per utterance (device), or nullptr
    int add, mul;         // frames of margin; columns per frame at this layer's input
};
__device__ __forceinline__ int sx_valid_cols(const SxRagged &r, int b, int T) {
    if (!r.len) return T;
    const int n = r.len[b];
    const long long v = n > 0 ? (long long)(n + r.add) * r.mul : 0;
    return v < T ? (int)v : T;
}

struct SxArgs {
    const u32x4 *xp;      // input planes, cells of 8 bf16
    int64_t x_bstride;    // cells between batch items (= 3 * Cin/8 * T)
    const float *xr;      // RAWIN kernels instead: fp32 raw input [Cin/8][T][8] per utterance ...
    float islope;         // ... with leaky_relu(islope) applied on the way in (1 = none)
    int T;                // input length
    const u32x4 *wp;      // packed weights
    int wshift;           // log2(packed tile height / this kernel's tile height): a 64- or 32-row kernel can run on weights
                          // packed for 128-row tiles (short grids: launch_conv_sx's pack_cfg)
    const float *bias;    // [virtual rows] or nullptr
    const float *bias_b;  // per-utterance bias [B][bias_b_stride] over real channels, or nullptr
    int bias_b_stride;
    float *out_raw;       // fp32 raw [Cr/8][T*ups][8] or nullptr; stores leaky_relu(value, oslope)
    int64_t raw_bstride;  // floats between batch items (out_raw and res)
    uint16_t *out_pl;     // planes [3][Cr/8][T*ups][8] or nullptr; stores split(leaky_relu(value, oslope2))
    int64_t pl_bstride;   // bf16 elements between batch items
    const float *res;     // fp32 raw residual or nullptr
    // f16 single-plane mode (NP = 1): the residual is read from the fp16 plane tensor that holds leaky_relu(residual,
    // 1 / res_unslope) (plane layout and batch stride of out_pl); res then stays nullptr
    const uint16_t *res_pl;
    float res_unslope;
    const float *zeros;   // >= 1 KiB of zeros, 16-byte aligned
    int Cin, Cout, Cr;    // Cout = virtual rows (Cr * ups)
    int K, dil, padL, nchunks, ups;
    int LW;               // x tile width in cells                       ( " )
    // SX_WN_RMW: valid lengths, second planar tensor, first row that goes to it, rows that also get planes
    const int *len;
    float *out_raw2;
    int row_split, pl_rows;
    int pl_of2;               // the planes are those of out_raw2's rows (the WN skip sum after its last layer) instead of out_raw's
    int64_t planar_bstride;   // batch stride of out_raw / res in the planar epilogue (0: row_split * T; the coupling's x1 rows
                              // live inside z: C * F)
    int RS;               // cells between the rows of an x stage (>= LW) ( " )
    int s16;              // the weights are packed for the 16x16x32 main loop (ConvDesc::s16; f16x3, plane input)
    unsigned magic;       // ceil(2^32 / LW)                             ( " )
    unsigned x_bytes;     // bytes of one x stage                        ( " )
    unsigned lds_bytes;   // dynamic LDS of the launch (two x stages)    ( " )
    int NT, MT, B;        // tiles along time / along rows, utterances   ( " )
    int flags;            // EPI_RES | EPI_ACC | EPI_DIV | DBG_*
    float div, oslope, oslope2;
    float wscale;         // f16 mode: 1 / (power-of-two scale the packed weights carry), applied to the accumulators
    unsigned long long *prof;  // PROF instantiation only: cycle counters [lgkm wait, vm wait, barrier, DMA issue, loads+MFMA, steps]
    // f16 mode range guard: 64 slots (float bits, atomicMax) that receive the largest |value| this launch splits into
    // fp16 planes - its plane outputs and, RAWIN, its inputs after the leaky-ReLU.  nullptr = not tracked (test hooks).
    // A peak above 65504 (or inf) means split2h_pair clamped: the run is then reported as out of range instead of
    // returning plausible-looking audio (vits_stats::f16_peak_max / f16_saturated, VITS_E_RANGE).
    unsigned *peak;
    SxRagged rag;         // per-utterance tensor ends (generator convs of a padded batch), see SxRagged
    // Transposed convs (ups > 1) of kernel 2 * ups in their dense 3-tap form: every output phase uses TWO of the three taps -
    // tap 2 is all zeros for the phases r with (r + zt_p) / ups == 0, tap 0 for the others (model.cpp pack_convT_sx).  zt_p >= 0:
    // the 16x16x32 loop's four-wave tile (a wave = one 32-row block = one phase) skips the MFMAs of its block's zero tap (a
    // third of the launch's matrix work; adding x * 0 changes no accumulator).  -1: no such structure.
    int zt_p = -1;
    // XCD grouping of the time tiles (sx_xcd_group_shift()): log2 of how many CONSECUTIVE (time tile, utterance) pairs one XCD
    // takes before the deal moves on to the next XCD.  0 = round robin tile by tile.
    int xgs = 0;
};

// Workgroup ids go round-robin over the 8 XCDs (blocks b and b + 8 share one, MI355X_MICROARCH.md "Workgroup dispatch"), each
// XCD with an L2 of its own.  Dealt tile by tile, two NEIGHBOURING time tiles - which share their halo columns, (K - 1) dil of
// 256, a fifth of a k = 11 dilation-5 tile - always sit on different XCDs and the halo crosses the fabric twice.  Dealt in
// groups of 2^xgs consecutive tiles per XCD, neighbours inside a group are consecutive slots of one XCD (running at the same
// time on CUs of that XCD): the halo's second reader finds the lines in that L2.  The groups stay small so that the deal still
// balances utterances of unequal length (ragged rendering) over the XCDs.  VITSMI_XCD_GROUP = 1 | 2 | 4 | 8 | 16 (tiles).
inline int sx_xcd_group_shift() {
    static const int v = [] {
        const char *e = std::getenv("VITSMI_XCD_GROUP");
        int g = e ? std::atoi(e) : SX_XCD_GROUP_DEFAULT, s = 0;
        while ((2 << s) <= g && s < 4) s++;
        return s;
    }();
    return v;
}
// tile index of (slot `tseq` of XCD `xcd`)
__device__ __forceinline__ int sx_xcd_tile(int tseq, int xcd, int xgs) {
    return ((tseq >> xgs) << (xgs + 3)) + (xcd << xgs) + (tseq & ((1 << xgs) - 1));
}



template <int OFF>
__device__ __forceinline__ u32x4 ds_read128(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}

// 16 bytes per lane from global memory at (uniform base + per-lane byte offset + OFF), asynchronous
template <int OFF>
__device__ __forceinline__ u32x4 global_read128(uint32_t voff, const void *sbase) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
    return r;
}

// Four (two) 16-byte loads per lane from ONE scalar base, 1 KiB apart, in one statement that opens with the five wait states
// a vector-memory instruction needs after a VALU instruction wrote its scalar base.  Under scalar-register pressure hipcc
// keeps uniform values in VGPR lanes and fetches them with v_readlane right in front of their use; its hazard recogniser
// inserts the wait states for the memory instructions IT emits, not for those inside inline asm: the load then goes out with
// a stale base (seen: conv_sx_pair16_kernel<.., PERSIST>, memory faults at wild addresses from its tile loop; DESIGN 5.1g
// hazard 5; phoonnx_amd.build.sgpr_vmem_hazards checks every kernel's ISA for it).
__device__ __forceinline__ void global_read128_x4(uint32_t voff, const void *sbase, u32x4 &r0, u32x4 &r1, u32x4 &r2, u32x4 &r3) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5 offset:0\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(voff), "s"(sbase)
                 : "memory");
}
__device__ __forceinline__ void global_read128_x2(uint32_t voff, const void *sbase, u32x4 &r0, u32x4 &r1) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3 offset:0\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(r0), "=&v"(r1)
                 : "v"(voff), "s"(sbase)
                 : "memory");
}

// (experiment switch) SX_X_NT = 1: the activation tiles - x tiles of every sx kernel, by LDS-DMA or through registers - are
// requested with the `nt` policy; the weight stream, which every workgroup re-reads from L2, never is
#ifndef SX_X_NT
#define SX_X_NT 0
#endif
constexpr int kSxXAux = SX_X_NT ? 2 : 0;
template <int OFF>
__device__ __forceinline__ u32x4 global_read128_x(uint32_t voff, const void *sbase) {
#if SX_X_NT
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(r) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
    return r;
#else
    return global_read128<OFF>(voff, sbase);
#endif
}

// ... at a per-lane address
template <int OFF>
__device__ __forceinline__ u32x4 global_read128_v(const void *p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ void ds_write128(uint32_t addr, u32x4 v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// v -> three bf16 planes (round-to-nearest-even at each step; the residuals are exact in fp32)
__device__ __forceinline__ void split3(float v, unsigned short &p0, unsigned short &p1, unsigned short &p2) {
    const __bf16 h0 = (__bf16)v;
    const float r1 = v - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;
    const __bf16 h2 = (__bf16)r2;
    p0 = __builtin_bit_cast(unsigned short, h0);
    p1 = __builtin_bit_cast(unsigned short, h1);
    p2 = __builtin_bit_cast(unsigned short, h2);
}

// The same split for a pair of values, in the packed form the planes are stored in (lo half = first value):
// one v_cvt_pk_bf16_f32 per plane and pair, 5.5 VALU operations per value in all.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3_pair(float x, float y, unsigned &w0, unsigned &w1, unsigned &w2) {
    w0 = cvt_pk_bf16(x, y);
    const float rx = x - __uint_as_float(w0 << 16), ry = y - __uint_as_float(w0 & 0xffff0000u);
    w1 = cvt_pk_bf16(rx, ry);
    const float sx = rx - __uint_as_float(w1 << 16), sy = ry - __uint_as_float(w1 & 0xffff0000u);
    w2 = cvt_pk_bf16(sx, sy);
}

__device__ __forceinline__ float f16_bits_to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }

// Epilogue description bits beyond EPI_RES / EPI_ACC / EPI_DIV (conv_engine.hip.hpp).  The first group is
// derived from the arguments by launch_conv_sx; a kernel instantiated with EPI >= 0 has the whole description
// as a compile-time constant (the common generator epilogues), EPI = -1 reads it at run time.
enum : int {
    SX_HAS_RAW = 1 << 9,    // out_raw is written
    SX_HAS_PL = 1 << 10,    // out_pl is written
    SX_RAW_ACT = 1 << 11,   // oslope != 1
    SX_PL_ACT = 1 << 12,    // oslope2 != 1
    SX_HAS_BIASB = 1 << 13, // per-utterance bias
    // RAWIN kernels whose residual IS their input tensor (ResBlock2: x = conv(lrelu(x)) + x, modules.py:355-364): the
    // residual is requested in the prologue, next to the first x tile that covers the same lines, and waits in
    // registers.  Requested in the epilogue it is a second HBM read of the tensor (measured: 1.40 GB read per launch
    // against a 0.71 GB tensor; the tile's lines are gone from the 4 MB L2 by then), requested here the two reads of a
    // line merge in the L2.
    SX_RES_EARLY = 1 << 14,
    // WN in-layer with the gate folded in (modules.py:193-199, commons.py:99-106): the rows were packed so that every
    // 64-row tile holds 32 tanh channels (rows 0-31: channels 32 m ..) and their 32 sigmoid partners (rows 32-63:
    // channels H + 32 m ..).  The sigmoid waves hand their values to the tanh waves through LDS; the output is
    // acts = tanh(a) * sigmoid(b) as a PLANAR fp32 tensor [H][T] (what the res_skip conv on the f32 engine reads).
    // out_raw = acts, raw_bstride = H * T, Cr = 2 H; bias_b indexes ORIGINAL rows.  64-row tiles only.
    SX_GATE = 1 << 15,
    // Planar epilogue (generic instantiation only): o = old + act(value) * mask goes to PLANAR fp32 tensors [rows][T] - rows
    // [0, row_split) to out_raw, rows [row_split, Cout) to out_raw2 (row index less row_split) - and the new rows
    // [0, pl_rows) once more as fp16 operand planes (out_pl; out_raw may then be absent).  old = the output itself
    // (EPI_ACC), the planar tensor res (EPI_RES; rows of out_raw only) or nothing; act = ReLU with EPI_RELU; mask =
    // t < len[b] with EPI_MASK.  Users: the flow's res_skip conv (modules.py:200-209: x = (x + res) * mask, skip += ..,
    // planes of x for the next in-layer) and the text encoder's 1 x 1 / FFN convs (attentions.py:66-75, 419-427), whose
    // neighbours (attention, LayerNorm) work on planar tensors.
    SX_WN_RMW = 1 << 19,  // (bits 16-18: DBG_NO_DMA, DBG_NO_EPI, EPI_NO_PADFILL of conv_engine.hip.hpp)
    SX_PLANAR_STORE2 = 1 << 20,  // planar epilogue with EPI_ACC: rows of out_raw2 are stored, not accumulated
    SX_PLANAR_COUPLING = 1 << 21,  // planar epilogue with EPI_ACC: o = (old - value * mask) * mask (modules.py:464, mean_only)
    // The residual (EPI_RES) is read from the PLANE tensor res_pl, which holds leaky_relu(residual, 1 / res_unslope) - one
    // fp16 plane (NP = 1: always) or the two planes h0, h1' of the f16x3 arithmetic (NP = 2: 22 bits, |x' - x| <= 2^-22 |x|) -
    // with the leaky-ReLU undone on the way in: a tensor that is both a conv input and a residual is then stored ONCE, as
    // operand planes, and no fp32 copy of the residual stream exists.
    SX_RES_PL = 1 << 22
};
constexpr int kSxEpiMask = EPI_RES | EPI_ACC | EPI_DIV | SX_HAS_RAW | SX_HAS_PL | SX_RAW_ACT | SX_PL_ACT | SX_HAS_BIASB | SX_RES_EARLY |
                           SX_GATE | SX_RES_PL;

// One 256-thread workgroup = 4 waves arranged WM x WN, each owning MW x NW 32x32 accumulator blocks.
// RAWIN: the input is the fp32 raw tensor itself; each x tile is loaded into registers, leaky-ReLU'd (islope) and
// split into the three bf16 planes on its way into LDS.  Used where the layer is HBM-bound (<= 64 channels):
// such tensors then exist only once, as 4-byte raw values, instead of raw + 6-byte planes.
// NP = arithmetic: 6 = six exact bf16 plane products; 2 = f16x3 (two fp16 planes, three products; the default); 1 = one
// fp16 plane, one product (VITSMI_GEN_PRECISION=f16, BASELINE config 4; 16x16x32 loop only).
// SH = MFMA shape of the main loop: 32 = v_mfma_f32_32x32x16 (one step = one tap of a 16-channel chunk), 16 =
// v_mfma_f32_16x16x32 (one step = one tap of a 32-channel chunk; f16x3 arithmetic, plane input only).  Same products,
// same accumulation order per output element up to the chunk grouping; the chip holds a higher clock on the 16x16x32
// shape under its power cap (tools/mfma_shape_probe.hip: 1.13-1.15x the FLOP/s at equal cycles per FLOP).
template <int MW, int NW, int WM, int WN, int EPI = -1, bool PROF = false, bool RAWIN = false, int NP = 6, int SH = 32>
__global__ __launch_bounds__(256, (MW * NW > 8) ? 1 : ((SH == 16 && MW * NW == 2 && !RAWIN) ? 3 : 2)) void conv_sx_kernel(SxArgs a) {
    constexpr int BM = MW * WM * 32, BN = NW * WN * 32, MB = BM / 32;
    constexpr bool S16 = SH == 16;
    static_assert(SH == 32 || (SH == 16 && (NP == 2 || NP == 1) && !RAWIN && !PROF), "16x16x32: fp16 arithmetics on plane inputs");
    static_assert(NP == 6 || NP == 1 || NP == 2, "arithmetic");
    static_assert(NP != 1 || SH == 16, "the single-plane mode exists on the 16x16x32 loop only");
    constexpr bool H1 = NP == 1;                           // one fp16 plane, one product
    constexpr bool F16 = NP == 2 || H1;                    // fp16 planes (two: three products)
    constexpr int NPROD = H1 ? 1 : (F16 ? 3 : NP);
    constexpr int NPL = NP == 6 ? 3 : (H1 ? 1 : 2);        // x planes read
    constexpr int NPLA = NPL;                              // weight planes read (f16x3: g0, g1; g0 * 2^-11 is made here)
    constexpr int NPW = H1 ? 1 : (F16 ? 2 : 3);            // weight planes packed per 32-row block
    static_assert(WM * WN == 4, "four waves per workgroup");
    static_assert(MW <= 2, "load_a addresses two block rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_sx[];  // two x stages
    const int tid = threadIdx.x, lane = tid & 63;
    // wave index as a SCALAR: the DMA bookkeeping (piece loops, vmcnt counts) then stays on the scalar unit
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware tile order (1-D grid).  Workgroup ids go round-robin over the 8 XCDs, each with its own L2: the
    // row tiles (mt) of one (time tile, utterance) get consecutive slots of the SAME XCD, so the x tile they all
    // read comes from HBM once and from that L2 for the others.
    const int wg_xcd = blockIdx.x & 7, wg_seq = blockIdx.x >> 3;
    const int mt = __builtin_amdgcn_readfirstlane(wg_seq % a.MT);  // (readfirstlane: keep these provably uniform)
    const int tile_nb = __builtin_amdgcn_readfirstlane(sx_xcd_tile(wg_seq / a.MT, wg_xcd, a.xgs));  // (time tile, utterance) index
    if (tile_nb >= a.NT * a.B) return;                 // padding workgroups of the last round (uniform exit)
    const int b = tile_nb / a.NT, t0 = (tile_nb - b * a.NT) * BN;
    const int T = a.T, LW = a.LW, K = a.K, CG = a.Cin >> 3;
    // TV: where this utterance's tensor ends (SxRagged; = T for a padded rendering).  T stays the row pitch of every tensor.
    const int TV = __builtin_amdgcn_readfirstlane(sx_valid_cols(a.rag, b, T));
    if (t0 >= TV) return;                              // (uniform exit) the whole tile lies behind the utterance's end
    const int RS = S16 ? a.RS : a.LW;          // cells between the rows of an x stage (16x16x32: rounded up, see launch_conv_sx)
    constexpr int XG = S16 ? 4 : 2;            // channel groups of 8 per chunk: rows of a stage = planes x XG
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_sx;
    const uint32_t XB = a.x_bytes;
    const u32x4 *xb = a.xp + (int64_t)b * a.x_bstride;
    const int64_t pstride = (int64_t)CG * T;  // cells per plane
    // packed weights of one step (one tap of one 16-channel chunk): the tile height they were packed for
    // (16x16x32: a step is k = 32, i.e. twice the bytes per 32-row block: [16-row sub-block][plane] x 1 KiB)
    constexpr int BLKBYTES = (S16 ? 2 : 1) * NPW * 1024;
    const int STEPBYTES = (MB << a.wshift) * BLKBYTES;
    // this wave's A rows of step 0 (uniform address: lives in SGPRs)
    const char *wbase = reinterpret_cast<const char *>(a.wp) + (int64_t)(mt >> a.wshift) * a.nchunks * K * STEPBYTES +
                        (mt & ((1 << a.wshift) - 1)) * (MB * BLKBYTES) + wm * (MW * BLKBYTES);
    const int nit = a.x_bytes >> 12;           // DMA rounds of 256 cells per x tile (the stage is padded to 4 KiB)
    float pk = 0.f;                            // f16 mode: largest |value| this thread split into fp16 planes

    // Fast path of the x-tile DMA (fp16 mode, plane input): which cell a lane fetches in round `it`, and whether that
    // cell lies inside the tensor, does not depend on the chunk - only the channel-group base does, and that is a
    // scalar.  So the per-lane byte offsets and the validity masks are computed once per tile (the general path spends
    // ~25 VALU operations, eight of them quarter-rate multiplies, in front of every DMA, next to a saturated matrix
    // pipe); lanes whose cell is padding are masked off and their LDS cells are zeroed once, below.
    constexpr int MAXIT = S16 ? 12 : 6;
    // (16x16x32: launch_conv_sx admits only what the fast path covers)
    const bool fastx = S16 || (F16 && !RAWIN && nit <= MAXIT && !(a.flags & DBG_NO_DMA) &&
                               (2 * pstride + 2 * (int64_t)T) * 16 < (1ll << 32));  // 32-bit byte offsets inside an utterance
    uint32_t xoffs[MAXIT];
    bool xok[MAXIT];
    int nx_issued = 0;  // DMAs this wave really issues per x tile (a round whose 64 cells are all padding is skipped)
    if constexpr (F16 && !RAWIN) {
        if (fastx) {
#pragma unroll
            for (int it = 0; it < MAXIT; it++) {
                const int i = it * 256 + wave * 64 + lane;
                const int row = (int)__umulhi((unsigned)i, a.magic);  // (magic = ceil(2^32 / RS))
                const int col = i - row * RS;
                const int t = t0 - a.padL + col;
                xok[it] = it < nit && row < XG * NPL && col < LW && t >= 0 && t < TV;
                xoffs[it] = (uint32_t)(((int64_t)(row / XG) * pstride + (int64_t)(row % XG) * T + t) * 16);
                nx_issued += __builtin_amdgcn_ballot_w64(xok[it]) != 0 ? 1 : 0;
            }
            if (t0 - a.padL < 0 || t0 - a.padL + LW > TV) {  // (uniform) only edge tiles have padding columns
                const u32x4 z = {0u, 0u, 0u, 0u};
                for (uint32_t o = (uint32_t)tid * 16u; o < a.lds_bytes; o += 4096u) ds_write128(lds0 + o, z);
                __syncthreads();  // the zeros are in place before the first DMA can land on a neighbouring cell
            }
        }
    }
    // x tile of one chunk -> LDS: rows (plane, channel-group half) x LW cells; every wave issues `nit` DMAs
    auto issue_x = [&](int chunk, uint32_t xoff) __attribute__((always_inline)) {
        if constexpr (F16 && !RAWIN) {
            if (fastx) {
                const char *cb = reinterpret_cast<const char *>(xb) + (int64_t)(XG * chunk) * T * 16;
#pragma unroll
                for (int it = 0; it < MAXIT; it++) {
                    if (it < nit) {  // (uniform) every wave issues the same count, masked-off rounds included
                        const int base = it * 256 + wave * 64;
                        if (xok[it]) lds_dma<16, kSxXAux>(cb + xoffs[it], reinterpret_cast<float *>(lds_sx + xoff + base * 16));
                    }
                }
                return;
            }
        }
        if constexpr (S16) return;
        for (int it = 0; it < nit; it++) {
            const int base = it * 256 + wave * 64;
            const int i = base + lane;
            const int row = (int)__umulhi((unsigned)i, a.magic);
            const int col = i - row * LW;
            const int t = t0 - a.padL + col;
            const bool ok = row < 2 * NPL && t >= 0 && t < TV;  // (rows of planes this mode does not read stay unloaded)
            const u32x4 *src = ok ? xb + ((row >> 1) * pstride + (int64_t)(2 * chunk + (row & 1)) * T + t)
                                  : reinterpret_cast<const u32x4 *>(a.zeros) + lane;
            lds_dma<16, kSxXAux>(src, reinterpret_cast<float *>(lds_sx + xoff + base * 16));
        }
    };
    // ---- RAWIN: x tile through registers.  A thread owns cells i = it*256 + tid of the [2 channel groups][LW]
    // tile (8 fp32 = two 16-byte loads each); rows past the tile and out-of-range columns read the zero page.
    constexpr int NXC = 3;  // cells per thread: 2 * LW <= 768 (launch_conv_sx)
    u32x4 xst[NXC][2];
    const float *xrb = RAWIN ? a.xr + (int64_t)b * a.Cin * T : nullptr;
    const int nxc = (2 * LW + 255) >> 8;
    auto xcell = [&](int it, int &kh, int &col) {
        const int i = it * 256 + tid;
        kh = (i >= LW ? 1 : 0) + (i >= 2 * LW ? 1 : 0);
        col = i - kh * LW;
    };
    // which cell a thread stages, and whether it is padding, does not depend on the chunk: byte offsets from the
    // chunk's (scalar) base and validity are computed once per tile; padding lanes read the chunk base itself (always
    // inside the tensor) and are zeroed in xstore
    uint32_t xroff[NXC];
    bool xrok[NXC];
    if constexpr (RAWIN) {
#pragma unroll
        for (int it = 0; it < NXC; it++) {
            int kh, col;
            xcell(it, kh, col);
            const int t = t0 - a.padL + col;
            xrok[it] = it < nxc && kh < 2 && t >= 0 && t < TV;
            xroff[it] = xrok[it] ? (uint32_t)(((int64_t)kh * T + t) * 32) : 0u;
        }
    }
    auto xload = [&](int chunk) {  // (launch_conv_sx keeps T small enough for 32-bit offsets inside a chunk)
        const float *cbase = xrb + (int64_t)(2 * chunk) * T * 8;
        static_for<NXC>([&](auto I) {
            constexpr int it = decltype(I)::value;
            if (it < nxc) {
                xst[it][0] = global_read128_x<0>(xroff[it], cbase);
                xst[it][1] = global_read128_x<16>(xroff[it], cbase);
            }
        });
    };
    auto xstore = [&](uint32_t xoff) {  // registers -> leaky-ReLU -> three planes -> LDS stage at byte offset xoff
        const float isl = a.islope;
        static_for<NXC>([&](auto I) {
            constexpr int it = decltype(I)::value;
            if (it < nxc) {
                int kh, col;
                xcell(it, kh, col);
                if (kh < 2) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float x = xrok[it] ? __uint_as_float(xst[it][e >> 2][e & 3]) : 0.f;  // padding cell
                        v[e] = fmaxf(x, x * isl);  // leaky_relu for 0 < islope <= 1 (1 = none)
                    }
                    u32x4 w0, w1, w2;
                    unsigned p0[4], p1[4], p2[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if constexpr (F16) split2h_pair_pk(v[2 * e], v[2 * e + 1], p0[e], p1[e], pk);
                        else split3_pair(v[2 * e], v[2 * e + 1], p0[e], p1[e], p2[e]);
                    }
                    w0 = u32x4{p0[0], p0[1], p0[2], p0[3]};
                    w1 = u32x4{p1[0], p1[1], p1[2], p1[3]};
                    const uint32_t ad = lds0 + xoff + (uint32_t)(kh * LW + col) * 16u;
                    const uint32_t pb = (uint32_t)(2 * LW) * 16u;
                    ds_write128(ad, w0);
                    if constexpr (NPL > 1) ds_write128(ad + pb, w1);
                    if constexpr (!F16 && NPL > 2) {
                        w2 = u32x4{p2[0], p2[1], p2[2], p2[3]};
                        ds_write128(ad + 2 * pb, w2);
                    }
                }
            }
        });
    };
    const int nxv = RAWIN ? 2 * nxc : (fastx ? nx_issued : nit);  // vector-memory operations per wave for one x tile

    // wait until at most n of this wave's vector-memory operations are still in flight (they retire in order)
    auto wait_vm = [&](int n) {
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;  // (launch_conv_sx keeps nit <= 12)
        }
    };

    f32x16 acc[MW][NW];
    if constexpr (!S16) {
#pragma unroll
        for (int m = 0; m < MW; m++)
#pragma unroll
            for (int n = 0; n < NW; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;
    }
    constexpr bool PRE = RAWIN && EPI >= 0 && (EPI & SX_RES_EARLY) != 0;
    constexpr int NRND_ = MW * (NW / 2);
    f32x4 pre[PRE ? NRND_ : 1][2][4];

    if constexpr (S16) {
    // ================================================================== main loop, v_mfma_f32_16x16x32_f16
    // A wave's 64 x 128 (MW x NW blocks of 32 x 32) tile as 2 MW x 2 NW blocks of 16 x 16.  One step = one tap of a
    // 32-channel chunk (k = 32): lane (c = lane & 15, g = lane >> 4) holds weights W[row c of its 16-row block][channels 8 g ..
    // 8 g + 7] and activations x[channels 8 g ..][column c of its 16-column block].  96 MFMAs of 16 cycles per step and wave
    // (MW = 2), issued as 2 half-steps (one 32-row block each) x NW quarters (one 32-column block each) x 12.
    //   A (weights): L2 -> registers, two half-step buffers of 16 registers, requested one half-step (48 MFMAs = 768
    //                cycles, the lead the 32x32x16 loop has) ahead; (three buffers, a whole step ahead, do not fit: 128
    //                accumulator + 48 + 32 operand registers and the compiler spilled 80)
    //   B (x tile):  LDS -> registers per quarter (4 ds_read_b128 = 16 registers), two quarter buffers, a quarter ahead;
    //                both row halves of a step read the same quarters (the x tile is re-read from LDS once per 32 rows).
    // The 16-row sub-blocks' rows were permuted at pack time (model.cpp pack_conv_sx, s16) so that after ONE
    // v_permlane16_swap per accumulator pair every lane holds exactly what the 32 x 32 accumulator layout of the epilogue
    // expects: row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of column lane & 31 in register r.
    constexpr int HPS = MW;                 // half-steps per step
    constexpr int NQ = NW;                  // quarters per half-step
    struct AHalf {
        u32x4 f[2][NPL];                    // [16-row sub-block][plane]
    };
    f32x4 c16[MW][2][NW][2];                // [32-row block][sub-block a][32-column block][sub-block b]
#pragma unroll
    for (int m = 0; m < MW; m++)
#pragma unroll
        for (int aa = 0; aa < 2; aa++)
#pragma unroll
            for (int n = 0; n < NW; n++)
#pragma unroll
                for (int bb = 0; bb < 2; bb++) c16[m][aa][n][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int S = a.nchunks * K, H = S * HPS;
    // this wave's all-zero tap of a transposed conv (SxArgs::zt_p), or -1
    int ztap = -1;
    if constexpr (MW == 1 && WN == 1) {
        if (a.zt_p >= 0 && a.ups > 1) {
            const int blk = (mt * BM + wm * 32) >> 5;  // 32-row block = (channel block, phase)
            ztap = __builtin_amdgcn_readfirstlane(((blk % a.ups) + a.zt_p) / a.ups == 0 ? 2 : 0);
        }
    }
    bool zskip = false;  // (per half-step, wave-uniform)
    const uint32_t voff0 = (uint32_t)lane * 16u;
    auto load_ah = [&](AHalf &f, int hs) __attribute__((always_inline)) {  // half-step hs = (step, 32-row block m)
        const int st = hs / HPS, m = hs - st * HPS;
        // (readfirstlane on both halves: the address must be provably wave-uniform for the scalar-base load)
        const uint64_t pa = reinterpret_cast<uint64_t>(wbase) + (uint64_t)((int64_t)st * STEPBYTES + m * BLKBYTES);
        const char *sb = reinterpret_cast<const char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) << 32) |
                                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)pa));
        if constexpr (H1) {
            f.f[0][0] = global_read128<0>(voff0, sb);
            f.f[1][0] = global_read128<1024>(voff0, sb);
        } else {
            f.f[0][0] = global_read128<0>(voff0, sb);
            f.f[0][1] = global_read128<1024>(voff0, sb);
            f.f[1][0] = global_read128<2048>(voff0, sb);
            f.f[1][1] = global_read128<3072>(voff0, sb);
        }
    };
    constexpr int NAL = 2 * NPL;            // weight loads per half-step = B reads per quarter
    const uint32_t b_lane = lds0 + (uint32_t)((lane >> 4) * RS + wn * (NW * 32) + (lane & 15)) * 16u;
    const uint32_t plane_b = (uint32_t)(4 * RS) * 16u;
    u32x4 bq[2][2][NPL];                    // [buffer][sub-block b][plane]
    auto load_bq = [&](auto BUF, auto Q, uint32_t bb0) __attribute__((always_inline)) {
        constexpr int bf = decltype(BUF)::value, q = decltype(Q)::value;
        if constexpr ((SX16_ABL & 2) != 0) return;
        bq[bf][0][0] = ds_read128<q * 512>(bb0);
        if constexpr (!H1) bq[bf][0][1] = ds_read128<q * 512>(bb0 + plane_b);
        bq[bf][1][0] = ds_read128<q * 512 + 256>(bb0);
        if constexpr (!H1) bq[bf][1][1] = ds_read128<q * 512 + 256>(bb0 + plane_b);
    };
    auto wait_bq = [&]() __attribute__((always_inline)) {  // the older of two quarters in flight has landed
        if constexpr (H1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    };
    auto mma_q = [&](const AHalf &f, auto M, auto BUF, auto Q) __attribute__((always_inline)) {
        constexpr int m = decltype(M)::value, bf = decltype(BUF)::value, q = decltype(Q)::value;
        // products in the order of the 32x32x16 loop: g1*h0, g0'*h1', g0*h0; consecutive MFMAs hit different accumulators
        if constexpr ((SX16_ABL & 1) != 0) return;
        if constexpr (MW == 1 && WN == 1) {
            if (zskip) return;  // (uniform) this block's weights of this tap are all zeros
        }
        if constexpr (SX16_PRIO) __builtin_amdgcn_s_setprio(1);
        if constexpr (H1) {
#pragma unroll
            for (int aa = 0; aa < 2; aa++)
#pragma unroll
                for (int bb = 0; bb < 2; bb++)
                    c16[m][aa][q][bb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f.f[aa][0]),
                                                                               __builtin_bit_cast(f16x8, bq[bf][bb][0]),
                                                                               c16[m][aa][q][bb], 0, 0, 0);
        } else
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int aa = 0; aa < 2; aa++)
#pragma unroll
                for (int bb = 0; bb < 2; bb++) {
                    const f16x8 ga = c == 0 ? __builtin_bit_cast(f16x8, f.f[aa][NPL - 1])
                                            : (c == 1 ? __builtin_bit_cast(f16x8, f.f[aa][0]) * (_Float16)0.00048828125f
                                                      : __builtin_bit_cast(f16x8, f.f[aa][0]));
                    c16[m][aa][q][bb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga, __builtin_bit_cast(f16x8, bq[bf][bb][c == 1 ? NPL - 1 : 0]),
                                                                               c16[m][aa][q][bb], 0, 0, 0);
                }
        if constexpr (SX16_PRIO) __builtin_amdgcn_s_setprio(0);
    };
    auto wait_vm16 = [&](int n) __attribute__((always_inline)) {  // at most n of this wave's vector-memory operations still in flight
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;  // (nx_issued <= MAXIT = 12)
        }
    };
    const std::integral_constant<int, 0> I0{};
    const std::integral_constant<int, 1> I1{};
    // prologue: the first x tile, then the first two half-steps' weights (vector-memory operations retire in order:
    // whoever waits for A(0) has waited for x(0))
    // NA weight sets, NA - 1 half-steps ahead.  Two for the wide tiles (128 accumulators leave no room; a half-step is
    // 768 cycles of MFMAs there).  Where a wave holds 32 x 64 outputs (a half-step is 24 MFMAs = 384 cycles; the
    // instantiations short grids run on) three fit - and measured no different from two (SX16_NA_SMALL).
    constexpr int NA = NW == 2 ? SX16_NA_SMALL : 2;
    static_assert(NA >= 2 && NA <= 4, "");
    issue_x(0, 0);
    AHalf ab[NA];
    static_for<NA - 1>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if (i < H) load_ah(ab[i], i);
    });
    int chunk = 0, tap = 0;
    // vector-memory bookkeeping (operations retire in order).  iss[j] / xct[j]: operations / x DMAs this wave issued in the
    // half-step j + 1 back; a_since_x: weight loads issued after the DMAs of the x tile the next chunk start will publish
    int iss[NA - 1], xct[NA - 1];
#pragma unroll
    for (int j = 0; j < NA - 1; j++) {
        iss[j] = NAL;  // (the prologue's weight requests, youngest first; x(0) went out before all of them)
        xct[j] = 0;
    }
    int a_since_x = NAL * (NA - 1);
    auto half_step = [&](AHalf &fc, AHalf &fload, auto M, int hs) __attribute__((always_inline)) {
        constexpr int m = decltype(M)::value;
        // in flight behind A(hs) (requested by the previous half-step): the x DMAs if that half-step issued them
        // ... unless this half-step opens a chunk whose x tile was issued by the previous one (K = 1 and one half-step per
        // step): the barrier below publishes that tile, so this wave's share of it must have landed
        const bool chunk_start = m == 0 && tap == 0;
        zskip = tap == ztap;
        // A(hs) was the FIRST request of the half-step NA - 1 back: in flight behind it may stay that half-step's x DMAs and
        // everything the half-steps since have issued.  A half-step that opens a chunk also needs that chunk's x tile (the
        // barrier below publishes it): behind its DMAs only the weight requests made since may stay.
        int allow = xct[NA - 2];
#pragma unroll
        for (int j = 0; j < NA - 2; j++) allow += iss[j];
        if (chunk_start && a_since_x < allow) allow = a_since_x;
        // (the common case first: the switch below is a tree of scalar branches, ~100 cycles at one wave per SIMD)
        // (its own asm text: identical statements would be merged back into the switch's case 0)
        if (__builtin_expect(allow == 0, 1)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(15)" ::: "memory");
        else wait_vm16(allow);
        const bool more_x = chunk + 1 < a.nchunks;
        if (chunk_start && !(SX16_ABL & 4)) {
            __builtin_amdgcn_s_barrier();  // x(chunk) is complete in LDS; everyone is done with the other stage
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = NA - 2; j > 0; j--) {
            iss[j] = iss[j - 1];
            xct[j] = xct[j - 1];
        }
        iss[0] = xct[0] = 0;
        if (hs + NA - 1 < H && !(SX_NOA && hs > 0)) {
            load_ah(fload, hs + NA - 1);
            iss[0] = NAL;
            a_since_x += NAL;
        }
        const bool do_x = chunk_start && more_x;
        if (do_x) {
            if constexpr (!SX16_SPREAD) issue_x(chunk + 1, ((chunk + 1) & 1) * XB);
            iss[0] += nx_issued;
            xct[0] = nx_issued;
            a_since_x = 0;
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t bb0 = b_lane + (uint32_t)(chunk & 1) * XB + (uint32_t)(tap * a.dil) * 16u;
        // does the half-step after this one read the same stage?  (it does unless it opens the next chunk: then its first
        // quarter can only be requested behind that chunk's barrier)
        const bool last_of_chunk = m == HPS - 1 && tap == K - 1;
        const uint32_t bnext = m == HPS - 1 ? bb0 + (uint32_t)a.dil * 16u : bb0;  // next half-step: next tap, or the other rows
        if (chunk_start) load_bq(I0, I0, bb0);  // (every other half-step found its first quarter requested by its predecessor)
        static_for<NQ>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            constexpr int bf = q & 1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q + 1 < NQ) {
                load_bq(std::integral_constant<int, (q + 1) & 1>{}, std::integral_constant<int, q + 1>{}, bb0);
                wait_bq();  // quarter q has landed, q + 1 is in flight
            } else {
                if (!last_of_chunk && hs + 1 < H) {
                    load_bq(std::integral_constant<int, NQ & 1>{}, I0, bnext);
                    wait_bq();
                } else
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_q(fc, M, std::integral_constant<int, bf>{}, Q);
            if constexpr (SX16_SPREAD) {
                // the next chunk's x tile, a few DMA rounds behind each quarter's MFMAs instead of one burst in front of them
                if (do_x) {
                    __builtin_amdgcn_sched_barrier(0);
                    const char *cb = reinterpret_cast<const char *>(xb) + (int64_t)(XG * (chunk + 1)) * T * 16;
                    const uint32_t xoff = ((chunk + 1) & 1) * XB;
#pragma unroll
                    for (int it = q; it < MAXIT; it += NQ)
                        if (it < nit && xok[it])
                            lds_dma<16, kSxXAux>(cb + xoffs[it], reinterpret_cast<float *>(lds_sx + xoff + (it * 256 + wave * 64) * 16));
                }
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        if (m == HPS - 1 && ++tap == K) {
            tap = 0;
            chunk++;
        }
    };
    static_assert(NQ % 2 == 0, "the first quarter of a half-step always lands in quarter buffer 0");
    // unrolled by the period of (weight set, row half) - both static inside the body
    constexpr int U = NA * HPS / (NA % HPS == 0 ? HPS : 1) / (HPS % NA == 0 && NA != HPS ? NA : 1);
    static_assert(U % NA == 0 && U % HPS == 0, "one period of weight sets and row halves");
    int hs = 0;
    for (; hs + U <= H; hs += U)
        static_for<U>([&](auto I) {
            constexpr int i = decltype(I)::value;
            half_step(ab[i % NA], ab[(i + NA - 1) % NA], std::integral_constant<int, i % HPS>{}, hs + i);
        });
    // hipcc sinks kernel-argument loads (s_load) that only the epilogue uses to THIS point, the block between the unrolled
    // loop and its tail.  A scalar load in flight counts in lgkmcnt and returns out of order, so the tail steps' counted
    // `s_waitcnt lgkmcnt(n)` could pass with a ds_read still outstanding: drain the counter once here (per tile, not per step).
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // (no exit from the middle of the unrolled body: see the 32x32x16 loop)
    static_for<U - 1>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if (hs + i < H) half_step(ab[i % NA], ab[(i + NA - 1) % NA], std::integral_constant<int, i % HPS>{}, hs + i);
    });
    // 16 x 16 -> 32 x 32 accumulator layout: lanes with lane & 16 hold the right rows of the WRONG 16-column half for one
    // accumulator of each (sub-block b = 0, b = 1) pair; one half-row swap per register pair puts every value in place
#pragma unroll
    for (int m = 0; m < MW; m++)
#pragma unroll
        for (int n = 0; n < NW; n++)
#pragma unroll
            for (int aa = 0; aa < 2; aa++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(c16[m][aa][n][0][rr]),
                                                                     __float_as_uint(c16[m][aa][n][1][rr]), false, false);
                    acc[m][n][8 * aa + rr] = __uint_as_float(sw[0]);
                    acc[m][n][8 * aa + 4 + rr] = __uint_as_float(sw[1]);
                }
    } else {
    // Register sets: A fragments ping-pong between two sets (global loads are slow: a whole step ahead); the B
    // fragments have ONE set whose two halves (block columns [0, NH) and [NH, NW)) are refilled as soon as the
    // MFMAs that read them have been issued, i.e. half a step ahead.
    constexpr int NH = NW / 2;
    static_assert(NW % 2 == 0, "the B set is refilled in two halves");
    struct ASet {
        u32x4 fa[MW][3];
    };
    u32x4 fb[NW][3];
    const uint32_t b_lane = lds0 + (uint32_t)(hi * LW + wn * (NW * 32) + l31) * 16u;
    const uint32_t plane_b = (uint32_t)(2 * LW) * 16u;
    const uint32_t voff0 = (uint32_t)lane * 16u, voff1 = voff0 + NPW * 1024u;  // second block row (13-bit imm offsets)

    // A fragments: straight from the packed weights (L2-resident) into registers, one 1 KiB wave-load per
    // (block row, plane); asynchronous like the ds_reads: the consumer waits on vmcnt itself.
    auto load_a = [&](ASet &f, int step) {
        const char *sb = wbase + (int64_t)step * STEPBYTES;
        static_for<MW>([&](auto M) {
            constexpr int m = decltype(M)::value;
            const uint32_t vo = m == 0 ? voff0 : voff1;
            f.fa[m][0] = global_read128<0>(vo, sb);
            if constexpr (NPLA > 1) f.fa[m][1] = global_read128<1024>(vo, sb);
            if constexpr (NPLA > 2) f.fa[m][2] = global_read128<2048>(vo, sb);
        });
    };
    auto load_b_half = [&](auto H, int chunk, int tap) {
        constexpr int h = decltype(H)::value;
        const uint32_t bb0 = b_lane + (uint32_t)(chunk & 1) * XB + (uint32_t)(tap * a.dil) * 16u;
        const uint32_t bb1 = bb0 + plane_b, bb2 = bb1 + plane_b;
        static_for<NH>([&](auto N) {
            constexpr int n = h * NH + decltype(N)::value;
            fb[n][0] = ds_read128<n * 512>(bb0);
            if constexpr (NPL > 1) fb[n][1] = ds_read128<n * 512>(bb1);
            if constexpr (NPL > 2) fb[n][2] = ds_read128<n * 512>(bb2);
        });
    };
    auto mma_half = [&](const ASet &f, auto H) {
        constexpr int h = decltype(H)::value;
        // plane pairs of combined order <= 2, smallest terms first; consecutive MFMAs hit different accumulators
        // (NP < 6 keeps the LAST NP pairs of the list: the three / one most significant products)
        // (f16: g1*h0, g0'*h1', g0*h0)
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int c = 6 - NPROD; c < 6; c++)
#pragma unroll
            for (int m = 0; m < MW; m++)
#pragma unroll
                for (int n = h * NH; n < (h + 1) * NH; n++) {
                    if constexpr (F16) {
                        // g0' = g0 * 2^-11 (exact, or the correctly rounded subnormal): four packed multiplies per
                        // fragment instead of a third weight plane
                        const f16x8 ga = c == 4 ? __builtin_bit_cast(f16x8, f.fa[m][0]) * (_Float16)0.00048828125f
                                                : __builtin_bit_cast(f16x8, f.fa[m][PA[c]]);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga, __builtin_bit_cast(f16x8, fb[n][PB[c]]),
                                                                           acc[m][n], 0, 0, 0);
                    }
                    else
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.fa[m][PA[c]]),
                                                                            __builtin_bit_cast(bf16x8, fb[n][PB[c]]),
                                                                            acc[m][n], 0, 0, 0);
                }
    };
    // LDS reads return in order: "at most NH*3 outstanding" = the older half has landed
    auto wait_lds_older_half = [&]() {  // a half = NH * NPL reads
        if constexpr (NH * NPL == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else if constexpr (NH * NPL == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if constexpr (NH * NPL == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else if constexpr (NH * NPL == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if constexpr (NH * NPL == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    };
    static_assert(NH * NPL <= 4 || NH * NPL == 6 || NH * NPL == 8, "lgkmcnt immediates above");
    const std::integral_constant<int, 0> H0{};
    const std::integral_constant<int, 1> H1{};

    // ---- main loop over steps s = (chunk, tap).  Only the x tile goes through LDS: the tile of chunk c+1 is
    // DMA'd right after the barrier that opens chunk c (issued AFTER that step's A loads, so the A wait of tap 1
    // can leave exactly those `nit` DMAs in flight: vector-memory operations retire in order); it has two steps
    // to land before the tap-2 wait needs it done.  One barrier per chunk: it publishes x(c) and frees the
    // buffer x(c+1) goes to.  Within a chunk the B halves of tap+1 are requested while tap's MFMAs run.
    const int nchunks = a.nchunks, S = nchunks * K;
    const bool dbg_nodma = a.flags & DBG_NO_DMA;
    ASet f0, f1;
    if constexpr (RAWIN) {
        // the first weights are requested behind the first x tile and travel while that tile is converted (these
        // tiles have short K loops: a second exposed round trip in the prologue is a tenth of a 32-channel k = 3 tile)
        xload(0);
        load_a(f0, 0);
        wait_vm(MW * NPLA);  // in-order return: the x loads have landed, A(0) may still be in flight
        xstore(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the barrier of step 0 publishes it
    } else {
        issue_x(0, 0);
        load_a(f0, 0);
    }
    // SX_RES_EARLY: all residual operands of this wave, requested now (same addressing as the epilogue's issue_adds)
    if constexpr (PRE) {
        const int Tout_ = T * a.ups;  // (ups == 1 here: a residual conv)
        const float *resb_ = a.res + (int64_t)b * a.raw_bstride;
        static_for<NRND_>([&](auto R) {
            constexpr int rr = decltype(R)::value;
            constexpr int m = rr / (NW / 2), n0 = (rr % (NW / 2)) * 2;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int row0 = mt * BM + (wm * MW + m) * 32;
                const int t = t0 + (wn * NW + n0 + j) * 32 + l31;
                const int tl = t < T ? t : T - 1;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    pre[rr][j][q] = *reinterpret_cast<const f32x4 *>(resb_ + ((int64_t)((row0 >> 3) + q) * Tout_ + tl) * 8 + 4 * hi);
            }
        });
    }
    int chunk = 0, tap = 0;
    // PROF: where a step's cycles go (s_memtime stamps; tools/conv_bench.py --sx --prof)
    unsigned long long pt = 0;
    unsigned pc[5] = {0, 0, 0, 0, 0};
    auto stamp = [&](int i) {
        if constexpr (PROF) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            pc[i] += (unsigned)(now - pt);
            pt = now;
        }
    };
    unsigned long long prt0 = 0, pt0 = 0;
    if constexpr (PROF) {
        pt = pt0 = __builtin_amdgcn_s_memtime();
        prt0 = __builtin_amdgcn_s_memrealtime();
    }
    auto step = [&](ASet &fc, ASet &fn, int s) {
        stamp(4);  // MFMA issue of the previous step
        const bool more_x = chunk + 1 < nchunks && !dbg_nodma;
        if (tap == 0) {
            if constexpr (RAWIN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's x(chunk) writes
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // A(s) and this wave's share of x(chunk) have landed
            stamp(1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);
            load_b_half(H0, chunk, 0);
            load_b_half(H1, chunk, 0);
        } else {
            // A(s) landed; on tap 1 the x DMAs issued behind it may stay in flight
            if (tap == 1 && more_x) wait_vm(nxv);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < S && !(dbg_nodma && s > 0) && !(SX_NOA && s > 0)) load_a(fn, s + 1);
        if (tap == 0 && more_x) {
            if constexpr (RAWIN) xload(chunk + 1);
            else issue_x(chunk + 1, ((chunk + 1) & 1) * XB);
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp(3);
        const bool more_taps = tap + 1 < K;
        wait_lds_older_half();  // half 0 of B(s) has landed (half 1 may still be in flight)
        __builtin_amdgcn_sched_barrier(0);
        stamp(0);
        mma_half(fc, H0);
        __builtin_amdgcn_sched_barrier(0);
        if (more_taps) {
            load_b_half(H0, chunk, tap + 1);
            wait_lds_older_half();  // half 1 of B(s)
        } else
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_half(fc, H1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RAWIN) {
            // tap 2: the vmcnt(0) above has retired the next chunk's x loads; convert and write them while the
            // MFMAs run.  The writes sit between the two B-half requests in the LDS queue, so the next step's
            // "older half" wait covers them (LDS operations retire in order).
            if (tap == 2 && more_x) xstore(((chunk + 1) & 1) * XB);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more_taps) load_b_half(H1, chunk, tap + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (++tap == K) {
            tap = 0;
            chunk++;
        }
    };
    // (no exit from the middle of the unrolled pair: a mid-loop break makes hipcc copy all accumulators per step)
    for (int s = 0; s + 1 < S; s += 2) {
        step(f0, f1, s);
        step(f1, f0, s + 1);
    }
    // hipcc sinks kernel-argument loads (s_load) that only the epilogue uses to THIS point, the block between the unrolled
    // loop and its tail.  A scalar load in flight counts in lgkmcnt and returns out of order, so the tail steps' counted
    // `s_waitcnt lgkmcnt(n)` could pass with a ds_read still outstanding: drain the counter once here (per tile, not per step).
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (S & 1) step(f0, f1, S - 1);
    if constexpr (PROF) {
        stamp(4);
        if (a.prof && tid == 0) {
            for (int i = 0; i < 5; i++) atomicAdd(a.prof + i, (unsigned long long)pc[i]);
            atomicAdd(a.prof + 5, (unsigned long long)S);
            atomicAdd(a.prof + 6, __builtin_amdgcn_s_memtime() - pt0);          // shader cycles ...
            atomicAdd(a.prof + 7, __builtin_amdgcn_s_memrealtime() - prt0);     // ... per 100 MHz ticks = clock
        }
    }
    }  // (32x32x16 main loop)

    // ---- epilogue.  C/D layout of a 32x32 block: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5):
    // register quad q holds 4 consecutive channels 8q + 4*hi .. +3 of one time step = half a cell.
    // `flags` is a compile-time constant in the specialised instantiations: every test below then folds away
    const int flags = EPI >= 0 ? EPI : a.flags;
    if (a.flags & DBG_NO_EPI) {
        float sdbg = 0.f;
#pragma unroll
        for (int m = 0; m < MW; m++)
#pragma unroll
            for (int n = 0; n < NW; n++) sdbg += acc[m][n][0] + acc[m][n][7] + acc[m][n][15];
        if (sdbg == 12345.678f && a.out_raw) a.out_raw[tid] = sdbg;
        return;
    }
    if constexpr (EPI >= 0 && (EPI & SX_GATE) != 0) {
        static_assert(EPI < 0 || !(EPI & SX_GATE) || (WM == 2 && MW == 1), "gate: a tanh wave and a sigmoid wave per column group");
        const int H = a.Cr >> 1;                      // gate channels
        const int ch0 = mt * 32;                      // first tanh channel of this tile (sigmoid partner: H + ch0)
        const int orow0 = wm == 0 ? ch0 : H + ch0;    // ORIGINAL first row of this wave's block (bias_b index)
        const int vrow0 = mt * BM + wm * 32;          // virtual (packed) first row (bias index)
        const float wsc = a.wscale;
        const float *biasp = a.bias ? a.bias : a.zeros;
        const int b_on = a.bias ? 1 : 0;
        const float *bbp = a.bias_b ? a.bias_b + (int64_t)b * a.bias_b_stride : a.zeros;
        const int bb_on = a.bias_b ? 1 : 0;
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            bq[q] = *reinterpret_cast<const f32x4 *>(biasp + (vrow0 + 8 * q + 4 * hi) * b_on) +
                    *reinterpret_cast<const f32x4 *>(bbp + (orow0 + 8 * q + 4 * hi) * bb_on);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave has left the main loop: the x stages become the exchange buffer
        __builtin_amdgcn_sched_barrier(0);
        // Exchange slot of (column group wn, block column n, quad q): 64 lanes x 16 bytes.  The work is split by block
        // column so that both waves of a pair apply transcendentals to all their values and store half of the result:
        // columns [0, NW/2): the sigmoid wave sends sigmoid(b), the tanh wave multiplies and stores;
        // columns [NW/2, NW): the tanh wave sends tanh(a), the sigmoid wave multiplies and stores.
        const uint32_t exch = lds0 + (uint32_t)wn * (NW * 4 * 1024) + (uint32_t)lane * 16u;
        f32x4 mine[NW][4];
#pragma unroll
        for (int n = 0; n < NW; n++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float v = F16 ? __builtin_fmaf(acc[0][n][4 * q + e], wsc, bq[q][e]) : acc[0][n][4 * q + e] + bq[q][e];
                    mine[n][q][e] = wm == 0 ? tanh_nb(v) : sigmoid_nb(v);
                }
                const bool send = (wm == 1) == (n < NW / 2);  // (uniform per wave)
                if (send) ds_write128(exch + (uint32_t)(n * 4 + q) * 1024u, __builtin_bit_cast(u32x4, mine[n][q]));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {
            float *actb = a.out_raw + (int64_t)b * a.raw_bstride;
            const int n_lo = wm == 0 ? 0 : NW / 2;
            u32x4 got[NW / 2][4];
#pragma unroll
            for (int k = 0; k < NW / 2; k++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(got[k][q]) : "v"(exch + (uint32_t)((n_lo + k) * 4 + q) * 1024u) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < NW / 2; k++) {
                // (compile-time register index for both halves: select after the multiply)
                const int t = t0 + (wn * NW + n_lo + k) * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f32x4 other = __builtin_bit_cast(f32x4, got[k][q]);
                    const f32x4 own = wm == 0 ? mine[k][q] : mine[NW / 2 + k][q];
                    if (t < TV) {
                        if (a.out_pl) {
                            // acts as the fp16 operand planes of the res_skip conv on this engine (|acts| < 1: no range issue)
                            unsigned wa[2], wb[2];
                            split2h_pair(own[0] * other[0], own[1] * other[1], wa[0], wa[1]);
                            split2h_pair(own[2] * other[2], own[3] * other[3], wb[0], wb[1]);
                            uint16_t *pl = a.out_pl + (int64_t)b * a.pl_bstride;
                            const int64_t cell = ((int64_t)((ch0 >> 3) + q) * T + t) * 8 + 4 * hi;
                            *reinterpret_cast<u32x2 *>(pl + cell) = u32x2{wa[0], wb[0]};
                            *reinterpret_cast<u32x2 *>(pl + (int64_t)(H >> 3) * T * 8 + cell) = u32x2{wa[1], wb[1]};
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++) actb[(int64_t)(ch0 + 8 * q + 4 * hi + e) * T + t] = own[e] * other[e];
                        }
                    }
                }
            }
        }
        return;
    }
    if constexpr (EPI < 0 && !H1) {
        if (a.flags & SX_WN_RMW) {
            // planar epilogue: o = old + act(acc * wscale + bias) * mask, stored as [B][rows][T] fp32 and / or as operand planes.
            //   old: the output itself (EPI_ACC), the planar tensor `res` (EPI_RES, x rows only), or nothing
            //   act: ReLU with EPI_RELU; mask: t < len[b] with EPI_MASK
            const bool p_acc = (a.flags & EPI_ACC) != 0, p_res = (a.flags & EPI_RES) != 0, p_relu = (a.flags & EPI_RELU) != 0;
            const int Lb = ((a.flags & EPI_MASK) && a.len) ? a.len[b] : T;
            const float wsc = a.wscale;
            const float *biasp = a.bias ? a.bias : a.zeros;
            const int b_on = a.bias ? 1 : 0;
            const int xrows = a.row_split, srows = a.Cout - a.row_split;
            uint16_t *plb = a.out_pl ? a.out_pl + (int64_t)b * a.pl_bstride : nullptr;
            const int64_t plane_elems = (int64_t)(a.pl_rows >> 3) * T * 8;
#pragma unroll
            for (int m = 0; m < MW; m++) {
                const int row0 = mt * BM + (wm * MW + m) * 32;
                const bool to_x = row0 < xrows;   // (a 32-row block lies on one side: both counts are multiples of 32)
                const int64_t xbs = a.planar_bstride ? a.planar_bstride : (int64_t)xrows * T;
                float *ob = to_x ? (a.out_raw ? a.out_raw + (int64_t)b * xbs : nullptr) : a.out_raw2 + (int64_t)b * srows * T;
                // (SX_PLANAR_STORE2: the rows of the second tensor are stored, not accumulated - the first WN layer's skip)
                const float *oldp = p_acc && !(!to_x && (a.flags & SX_PLANAR_STORE2)) ? ob : (p_res && to_x ? a.res + (int64_t)b * xbs : nullptr);
                const int r0 = to_x ? row0 : row0 - xrows;
                const bool planes = plb && (a.pl_of2 ? !to_x : to_x) && r0 < a.pl_rows;
                const bool coupling = (a.flags & SX_PLANAR_COUPLING) != 0;
                f32x4 bq[4];
#pragma unroll
                for (int q = 0; q < 4; q++) bq[q] = *reinterpret_cast<const f32x4 *>(biasp + (row0 + 8 * q + 4 * hi) * b_on);
#pragma unroll
                for (int n = 0; n < NW; n++) {
                    const int t = t0 + (wn * NW + n) * 32 + l31;
                    const int tl = t < T ? t : T - 1;
                    const float mk = t < Lb ? 1.f : 0.f;
                    float old[16];
                    if (oldp) {
#pragma unroll
                        for (int r = 0; r < 16; r++) old[r] = oldp[(int64_t)(r0 + (r & 3) + 8 * (r >> 2) + 4 * hi) * T + tl];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r++) old[r] = 0.f;
                    }
                    if (t >= TV) continue;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            // register 4 q + e = row e + 8 q + 4 hi of the block
                            float v = __builtin_fmaf(acc[m][n][4 * q + e], wsc, bq[q][e]);
                            if (p_relu) v = __builtin_fmaxf(v, 0.f);
                            o[e] = coupling ? (old[4 * q + e] - v * mk) * mk : old[4 * q + e] + v * mk;
                        }
                        if (ob) {
#pragma unroll
                            for (int e = 0; e < 4; e++) ob[(int64_t)(r0 + e + 8 * q + 4 * hi) * T + t] = o[e];
                        }
                        if (planes) {
                            unsigned wa[2], wb[2];
                            split2h_pair_pk(o[0], o[1], wa[0], wa[1], pk);
                            split2h_pair_pk(o[2], o[3], wb[0], wb[1], pk);
                            const int64_t cell = ((int64_t)((r0 >> 3) + q) * T + t) * 8 + 4 * hi;
                            *reinterpret_cast<u32x2 *>(plb + cell) = u32x2{wa[0], wb[0]};
                            *reinterpret_cast<u32x2 *>(plb + plane_elems + cell) = u32x2{wa[1], wb[1]};
                        }
                    }
                }
            }
            if constexpr (F16) {
                if (a.peak) {
                    if constexpr (S16) sx_publish_peak_at(a.peak, (int)blockIdx.x, pk, reinterpret_cast<float *>(lds_sx));
                    else sx_publish_peak(a.peak, (int)blockIdx.x, pk);
                }
            }
            return;
        }
    }
    const int u = a.ups, Cr = a.Cr, Tout = T * u, CGo = Cr >> 3;
    float *rawb = a.out_raw + (int64_t)b * a.raw_bstride;     // (only dereferenced when the flags say so)
    uint16_t *plb = a.out_pl + (int64_t)b * a.pl_bstride;
    const float *resb = a.res + (int64_t)b * a.raw_bstride;
    const uint16_t *resplb = a.res_pl + (int64_t)b * a.pl_bstride;  // (SX_RES_PL: the residual lives in a plane tensor)
    const float unsl = a.res_unslope;
    // (compile-time in the specialised instantiations, a uniform run-time test in the generic one)
    const bool respl = H1 || (EPI >= 0 ? (EPI & SX_RES_PL) != 0 : a.res_pl != nullptr);
    // operands still to be loaded here: the residual (unless it was requested in the prologue) and / or the accumulate
    const bool res_epi = (flags & EPI_RES) && !PRE;
    const float *addp = res_epi ? resb : rawb;
    const bool has_add = res_epi || (flags & EPI_ACC);
    const bool two_adds = res_epi && (flags & EPI_ACC);
    const float oslope = a.oslope, oslope2 = a.oslope2, rdiv = a.div;
    const float wsc = a.wscale;
    const int64_t plane_elems = (int64_t)CGo * Tout * 8;
    const float *biasp = a.bias ? a.bias : a.zeros;  // a missing bias reads the zero page (no branch per quad)
    const int b_on = a.bias ? 1 : 0;
    // Rounds: (block row m, pair of block columns n0, n0 + 1).  The residual / accumulate operands of a round are 8
    // 16-byte loads per lane; with a compile-time epilogue that has ONE such operand they are requested a round ahead
    // (the A / B fragment registers are dead by now), so only the first round exposes a memory latency.  Loads use a
    // clamped column (always inside the tensor): no branch per load; stores are predicated on the real column.
    constexpr int NRND = MW * (NW / 2);
    constexpr bool PIPE = EPI >= 0 && ((EPI & EPI_RES) != 0 && !PRE) != ((EPI & EPI_ACC) != 0);
    f32x4 adb[PIPE ? 2 : 1][2][4], ad2[2][4];
    auto geom = [&](int m, int n, int &co0, int &r, int &t) {
        const int row0 = mt * BM + (wm * MW + m) * 32;
        // virtual rows of a transposed conv: (block of 32 channels, output phase r, channel) - the 32-row blocks of a tile
        // are the phases of the SAME channels, so a workgroup writes whole 64- / 128-byte runs of an output line (as
        // r-major rows, one phase per tile, PMC showed 1.9x the bytes at the fabric: the phases of a line came from
        // different workgroups and the 8- / 16-byte pieces of a sector left the L2 separately)
        r = u == 1 ? 0 : (row0 >> 5) % u;
        co0 = u == 1 ? row0 : (row0 / (32 * u)) * 32;
        t = t0 + (wn * NW + n) * 32 + l31;
    };
    auto cell_at = [&](int co0, int r, int t, int q) -> int64_t {
        return ((int64_t)((co0 >> 3) + q) * Tout + (t * u + r)) * 8 + 4 * hi;  // element offset inside raw / a plane
    };
    auto issue_adds = [&](auto R, auto P) {
        constexpr int rr = decltype(R)::value, pp = decltype(P)::value;
        constexpr int m = rr / (NW / 2), n0 = (rr % (NW / 2)) * 2;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int co0, r, t;
            geom(m, n0 + j, co0, r, t);
            const int tl = t < T ? t : T - 1;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t c = cell_at(co0, r, tl, q);
                if (respl) {
                    // the residual from its operand plane(s) (8 bytes per lane and plane, leaky-ReLU undone); the running sum
                    // stays fp32 raw
                    if (res_epi) {
                        float o4[4];
                        if constexpr (H1) {
                            unact4h(*reinterpret_cast<const u32x2 *>(resplb + c), unsl, o4);
                        } else {
                            float h0[4], h1[4];
                            unact4h(*reinterpret_cast<const u32x2 *>(resplb + c), 1.f, h0);
                            unact4h(*reinterpret_cast<const u32x2 *>(resplb + plane_elems + c), 1.f, h1);
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const float v = __builtin_fmaf(h1[e], 1.f / 2048.f, h0[e]);
                                o4[e] = v < 0.f ? v * unsl : v;
                            }
                        }
                        adb[pp][j][q] = f32x4{o4[0], o4[1], o4[2], o4[3]};
                    } else if (has_add)
                        adb[pp][j][q] = *reinterpret_cast<const f32x4 *>(rawb + c);
                } else if (has_add)
                    adb[pp][j][q] = *reinterpret_cast<const f32x4 *>(addp + c);
                if (two_adds) ad2[j][q] = *reinterpret_cast<const f32x4 *>(rawb + c);
            }
        }
    };
    if constexpr (PIPE) issue_adds(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    static_for<NRND>([&](auto R) {
        constexpr int rr = decltype(R)::value;
        constexpr int m = rr / (NW / 2), n0 = (rr % (NW / 2)) * 2;
        constexpr int pp = PIPE ? (rr & 1) : 0;
        if constexpr (PIPE) {
            if constexpr (rr + 1 < NRND) issue_adds(std::integral_constant<int, rr + 1>{}, std::integral_constant<int, (rr + 1) & 1>{});
        } else
            issue_adds(R, std::integral_constant<int, 0>{});
        int co0m, rm, tm;
        geom(m, n0, co0m, rm, tm);
        const int row0 = mt * BM + (wm * MW + m) * 32;
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bq[q] = *reinterpret_cast<const f32x4 *>(biasp + (row0 + 8 * q + 4 * hi) * b_on);
            if (flags & SX_HAS_BIASB)
                bq[q] += *reinterpret_cast<const f32x4 *>(a.bias_b + (int64_t)b * a.bias_b_stride + co0m + 8 * q + 4 * hi);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            constexpr int dummy = 0;
            (void)dummy;
            const int n = n0 + j;
            int co0, r, t;
            geom(m, n, co0, r, t);
            if (t >= TV) continue;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t cell = cell_at(co0, r, t, q);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if constexpr (F16) v[e] = __builtin_fmaf(acc[m][n][4 * q + e], wsc, bq[q][e]);  // (power of two: exact)
                    else v[e] = acc[m][n][4 * q + e] + bq[q][e];
                }
                if constexpr (PRE) v += pre[rr][j][q];
                if (has_add) v += adb[pp][j][q];
                if (two_adds) v += ad2[j][q];
                if (flags & EPI_DIV) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = v[e] / rdiv;
                }
                if (flags & SX_HAS_RAW) {
                    f32x4 o = v;
                    if (flags & SX_RAW_ACT) {
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = fmaxf(v[e], v[e] * oslope);  // leaky_relu, 0 < slope < 1
                    }
                    *reinterpret_cast<f32x4 *>(rawb + cell) = o;
                }
                if (flags & SX_HAS_PL) {
                    f32x4 o = v;
                    if (flags & SX_PL_ACT) {
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = fmaxf(v[e], v[e] * oslope2);
                    }
                    unsigned wa[3], wb[3];
                    if constexpr (H1) {
                        wa[0] = cvt1h_pair_pk(o[0], o[1], pk);
                        wb[0] = cvt1h_pair_pk(o[2], o[3], pk);
                    } else if constexpr (F16) {
                        split2h_pair_pk(o[0], o[1], wa[0], wa[1], pk);
                        split2h_pair_pk(o[2], o[3], wb[0], wb[1], pk);
                    } else {
                        split3_pair(o[0], o[1], wa[0], wa[1], wa[2]);
                        split3_pair(o[2], o[3], wb[0], wb[1], wb[2]);
                    }
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++)
                        *reinterpret_cast<u32x2 *>(plb + pl * plane_elems + cell) = u32x2{wa[pl], wb[pl]};
                }
            }
        }
    });
    if constexpr (F16) {
        if (a.peak) {  // (uniform branch)
            if constexpr (S16) sx_publish_peak_at(a.peak, (int)blockIdx.x, pk, reinterpret_cast<float *>(lds_sx));
            else sx_publish_peak(a.peak, (int)blockIdx.x, pk);
        }
    }
}

// sx tile configs: index -> (BM, BN, waves WM x WN, blocks per wave MW x NW):
//   0: 128x256 (2x2 waves of 64x128)   1: 64x256 (2x2 waves of 32x128)   2: 32x256 (1x4 waves of 32x64)
// All are 256 columns wide: the weights of a step then serve 4 (2) block columns per register load.
// 3: 64x128 (2x2 waves of 32x64), run-time choice for short grids of 64-row layers (same packed weights as 1)
inline int sx_tile_m(int cfg) { return cfg == 0 ? 128 : ((cfg == 1 || cfg == 3) ? 64 : 32); }
inline int sx_tile_n(int cfg) { return cfg == 3 ? 128 : 256; }

template <int MW, int NW, int WM, int WN, int EPI = -1, bool PROF = false, bool RAWIN = false, int NP = 6, int SH = 32>
inline hipError_t launch_conv_sx_k(const SxArgs &a, dim3 grid, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv_sx_kernel<MW, NW, WM, WN, EPI, PROF, RAWIN, NP, SH>;
    if (hipError_t e = sx_allow_big_lds(reinterpret_cast<const void *>(kern), attr_done); e != hipSuccess) return e;
    if (g_launch_name_on)
        snprintf(g_launch_name, sizeof g_launch_name, "conv_sx_kernel<%d, %d, %d, %d, %d, %s, %s, %d, %d>", MW, NW, WM, WN, EPI,
                 PROF ? "true" : "false", RAWIN ? "true" : "false", NP, SH);
    kern<<<grid, 256, lds, stream>>>(a);
    return hipGetLastError();
}

// the generator's frequent epilogues get their own instantiation (compile-time flags), the rest run generic
constexpr int kSxEpiPlanes = SX_HAS_PL | SX_PL_ACT;                                  // first conv of a ResBlock1 pair
constexpr int kSxEpiUp = SX_HAS_RAW | SX_HAS_PL | SX_PL_ACT;                         // upsampler
constexpr int kSxEpiInner = EPI_RES | SX_HAS_RAW | SX_HAS_PL | SX_PL_ACT;            // residual conv inside a block
constexpr int kSxEpiFirst = EPI_RES | SX_HAS_RAW;                                    // xs  = block output
constexpr int kSxEpiAccum = EPI_RES | EPI_ACC | SX_HAS_RAW;                          // xs += block output
constexpr int kSxEpiRaw = SX_HAS_RAW;                                                // raw only (raw-format stages)
constexpr int kSxEpiStageOut = EPI_RES | EPI_ACC | EPI_DIV | SX_HAS_PL | SX_PL_ACT;  // x = (xs + block) / n as planes
constexpr int kSxEpiGate = SX_GATE | SX_HAS_RAW;                                     // WN in-layer + gate -> planar acts
constexpr int kSxEpiInnerPl = EPI_RES | SX_HAS_PL | SX_PL_ACT;                       // (f16 single-plane mode) residual conv inside a block: the plane IS the stream

template <int MW, int NW, int WM, int WN, int NP = 6, int SH = 32>
inline hipError_t launch_conv_sx_epi(const SxArgs &a, int epi, dim3 grid, size_t lds, hipStream_t stream) {
    if (epi == -2) return launch_conv_sx_k<MW, NW, WM, WN, -1, false, false, NP, SH>(a, grid, lds, stream);  // SX_WN_RMW
    if constexpr (SH == 32) {
        if (a.flags & (DBG_NO_DMA | DBG_NO_EPI)) return launch_conv_sx_k<MW, NW, WM, WN, -1, false, false, NP>(a, grid, lds, stream);
    }
    switch (epi) {
        case kSxEpiPlanes: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiPlanes, false, false, NP, SH>(a, grid, lds, stream);
        case kSxEpiUp: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiUp, false, false, NP, SH>(a, grid, lds, stream);
        case kSxEpiInner: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiInner, false, false, NP, SH>(a, grid, lds, stream);
        case kSxEpiFirst: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiFirst, false, false, NP, SH>(a, grid, lds, stream);
        case kSxEpiAccum: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiAccum, false, false, NP, SH>(a, grid, lds, stream);
        case kSxEpiRaw: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiRaw, false, false, NP, SH>(a, grid, lds, stream);  // flow WN convs
        case kSxEpiStageOut: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiStageOut, false, false, NP, SH>(a, grid, lds, stream);
        default: break;
    }
    if constexpr (WM == 2 && MW == 1) {  // 64-row tiles: the gated WN in-layer (per-utterance bias or not)
        if ((epi & ~SX_HAS_BIASB) == kSxEpiGate) return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiGate, false, false, NP, SH>(a, grid, lds, stream);
    }
    if (epi & SX_GATE) return hipErrorInvalidValue;  // (no generic form of the gate epilogue)
    return launch_conv_sx_k<MW, NW, WM, WN, -1, false, false, NP, SH>(a, grid, lds, stream);
}

// raw-input kernels (tensors of <= 64 channels): outputs are raw only
template <int MW, int NW, int WM, int WN, int NP = 6>
inline hipError_t launch_conv_sx_rawin(const SxArgs &a, int epi, dim3 grid, size_t lds, hipStream_t stream) {
    if (a.flags & (DBG_NO_DMA | DBG_NO_EPI)) return launch_conv_sx_k<MW, NW, WM, WN, -1, false, true, NP>(a, grid, lds, stream);
    switch (epi) {
        case kSxEpiRaw: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiRaw, false, true, NP>(a, grid, lds, stream);
        case kSxEpiFirst: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiFirst, false, true, NP>(a, grid, lds, stream);
        case kSxEpiAccum: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiAccum, false, true, NP>(a, grid, lds, stream);
        default: break;
    }
    if constexpr (NP == 2 || NW == 2) {  // ResBlock2 (residual == input): the residual is requested in the prologue
        switch (epi) {
            case kSxEpiFirst | SX_RES_EARLY:
                return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiFirst | SX_RES_EARLY, false, true, NP>(a, grid, lds, stream);
            case kSxEpiAccum | SX_RES_EARLY:
                return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiAccum | SX_RES_EARLY, false, true, NP>(a, grid, lds, stream);
            case kSxEpiAccum | EPI_DIV | SX_RES_EARLY:
                return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiAccum | EPI_DIV | SX_RES_EARLY, false, true, NP>(a, grid, lds, stream);
            default: break;
        }
    }
    {
        return launch_conv_sx_k<MW, NW, WM, WN, -1, false, true, NP>(a, grid, lds, stream);
    }
}

// rawin: the input is a.xr (fp32 raw) instead of a.xp (planes); only the 64- and 32-row tiles have the
// registers for it (cfg 1 / 2).  nprod: 6 (bf16x6), 2 (f16x3), 1 (f16 single plane; s16 packings only).
// pack_cfg: the tile config the weights were packed for, if not `cfg` (a taller one: 128-row packing read by the 64-
// or 32-row kernel - same products in the same order, a shorter reduction per workgroup and 2-4x the workgroups).
hipError_t launch_conv_sx(SxArgs a, int cfg, int B, hipStream_t stream, bool rawin = false, int nprod = 6, int pack_cfg = -1);
hipError_t launch_conv_sx_f16_s16(const SxArgs &a, int cfg, int epi, dim3 grid, size_t lds, hipStream_t stream);
hipError_t launch_conv_sx_f16_s32(const SxArgs &a, int cfg, int epi, bool rawin, dim3 grid, size_t lds, hipStream_t stream);
hipError_t launch_conv_sx_bf16(const SxArgs &a, int cfg, int epi, int nprod, bool rawin, dim3 grid, size_t lds, hipStream_t stream);
hipError_t launch_conv_sx_h1_s16(const SxArgs &a, int cfg, int epi, dim3 grid, size_t lds, hipStream_t stream);
hipError_t launch_conv_sx_f16_s16p(const SxArgs &a, int cfg, int epi, dim3 grid, size_t lds, hipStream_t stream);
#ifdef VITSMI_IMPL_SX
hipError_t launch_conv_sx(SxArgs a, int cfg, int B, hipStream_t stream, bool rawin, int nprod, int pack_cfg) {
    const int BM = sx_tile_m(cfg), BN = sx_tile_n(cfg);
    a.wshift = 0;
    if (pack_cfg >= 0 && pack_cfg != cfg) {
        const int PM = sx_tile_m(pack_cfg);
        if (PM < BM || PM % BM || rawin) return hipErrorInvalidValue;
        while ((BM << a.wshift) < PM) a.wshift++;
    }
    a.LW = BN + (a.K - 1) * a.dil;
    a.RS = a.LW;
    // an x stage is padded to whole DMA rounds (256 cells = 4 KiB), so that every wave issues the same count
    // (rows = 2 channel-group halves x the planes the mode reads: 3 bf16 planes, 2 in the fp16 / bf16x3 modes, 1 in bf16)
    int xrows = nprod == 6 ? 6 : 4;
    const bool s16 = a.s16 != 0;
    if (cfg == 3 && !s16) return hipErrorInvalidValue;  // (64 x 128 tiles exist for the 16x16x32 loop only)
    if (s16) {
        // weights packed for the 16x16x32 main loop: chunks of 32 channels (4 channel groups x 2 planes = 8 rows per stage)
        if ((nprod != 2 && nprod != 1) || rawin || a.prof || (a.flags & (DBG_NO_DMA | DBG_NO_EPI)) || a.Cin % 32) return hipErrorInvalidValue;
        xrows = nprod == 1 ? 4 : 8;
        // rows 16 cells apart modulo 16: the ds_read_b128 of a B fragment (lanes 16 apart = the next channel group) is then
        // free of bank conflicts; where that does not fit two workgroups per CU the rows stay packed (mild conflicts)
        // (two workgroups per CU = 80 KiB each, all of it dynamic: the 16x16x32 kernels have no static LDS)
        const int rs16 = (a.LW + 15) / 16 * 16;
        if (((size_t)xrows * rs16 * 16 + 4095) / 4096 * 4096 + (size_t)xrows * rs16 * 16 <= (size_t)80 * 1024) a.RS = rs16;
    } else if (nprod == 1)
        return hipErrorInvalidValue;  // (the single-plane mode exists on the 16x16x32 loop only)
    a.magic = (unsigned)((0x100000000ull + a.RS - 1) / a.RS);
    a.x_bytes = (unsigned)(((size_t)xrows * a.RS * 16 + 4095) / 4096 * 4096);
    if (s16 && (a.x_bytes > 12 * 4096 || ((long long)(2 * (a.Cin / 8) + 2) * a.T) * 16 >= (1ll << 32))) return hipErrorInvalidValue;
    // two x stages; the weights never touch LDS.  (16x16x32: the second stage ends with its last row - the DMA rounds are
    // whole 4 KiB but lanes past the last row are masked off - which keeps the 50-cell halo of a k = 11, dilation 5 conv
    // inside 80 KiB)
    const size_t lds = s16 ? (size_t)a.x_bytes + (size_t)xrows * a.RS * 16 : 2 * (size_t)a.x_bytes;
    a.lds_bytes = (unsigned)lds;
    // (pack_conv_sx pads narrower kernels to 3 taps; model.cpp sx_supported() mirrors the size limits)
    if (lds > (size_t)kSxMaxDynLds || a.x_bytes > 12 * 4096 || (a.K < 3 && !s16)) return hipErrorInvalidValue;
    if (rawin && (cfg == 0 || 2 * a.LW > 768 || !a.xr || (long long)a.T * 64 + 64 >= (1ll << 32))) return hipErrorInvalidValue;
    if (a.oslope == 0.f) a.oslope = 1.f;
    if (a.oslope2 == 0.f) a.oslope2 = 1.f;
    if (a.islope == 0.f) a.islope = 1.f;
    if (a.Cin % 16 || a.Cout % 32 || a.Cr % 32 || a.Cout % BM) return hipErrorInvalidValue;
    a.NT = (a.T + BN - 1) / BN;
    a.MT = a.Cout / BM;
    a.B = B;
    const long long nb = (long long)a.NT * B;  // (time tile, utterance) pairs, dealt to the 8 XCDs in groups of 2^xgs (sx_xcd_tile)
    if (nb == 0) return hipSuccess;
    a.xgs = a.NT >= (2 << sx_xcd_group_shift()) ? sx_xcd_group_shift() : 0;  // (short tensors: nothing to group)
    const long long per_round = 8ll << a.xgs;
    const long long wgs = (nb + per_round - 1) / per_round * per_round * a.MT;
    if (wgs >= (1ll << 31)) return hipErrorInvalidValue;
    dim3 grid((unsigned)wgs, 1, 1);
    // epilogue description (see the SX_* bits): derived from the arguments, then matched against the instantiations
    int epi = a.flags & (EPI_RES | EPI_ACC | EPI_DIV);
    if (a.out_raw && !(a.flags & SX_NO_RAW_STORE)) epi |= SX_HAS_RAW | (a.oslope != 1.f ? SX_RAW_ACT : 0);
    if (a.out_pl && !(a.flags & SX_GATE)) epi |= SX_HAS_PL | (a.oslope2 != 1.f ? SX_PL_ACT : 0);
    if (a.bias_b) epi |= SX_HAS_BIASB;
    if (a.res_pl && nprod == 2 && (a.flags & EPI_RES)) {
        if (a.res || rawin || (a.flags & (SX_WN_RMW | SX_GATE))) return hipErrorInvalidValue;
        epi |= SX_RES_PL;
        if (a.res_unslope == 0.f) a.res_unslope = 1.f;
    }
    if (a.flags & SX_WN_RMW) {
        // (its own epilogue block in the generic instantiation: force that one)
        const bool p_acc = (a.flags & EPI_ACC) != 0;
        if (rawin || nprod != 2 || a.ups != 1 || (a.row_split < a.Cout && !a.out_raw2) ||
            (a.row_split && !a.out_raw && (p_acc || !a.out_pl)) || a.row_split % 32 || a.Cout % 32 || a.row_split > a.Cout ||
            a.pl_rows % 32 || a.pl_rows > (a.pl_of2 ? a.Cout - a.row_split : a.row_split) || (a.pl_rows && !a.out_pl) || a.bias_b ||
            (a.flags & SX_GATE) || ((a.flags & SX_PLANAR_COUPLING) && !p_acc) ||
            ((a.flags & EPI_RES) && (!a.res || p_acc)) || ((a.flags & EPI_MASK) && !a.len))
            return hipErrorInvalidValue;
        epi = -2;
    }
    if (a.flags & SX_GATE) {
        if (rawin || (cfg != 1 && cfg != 3) || (nprod != 2 && nprod != 6) || a.ups != 1 || (!a.out_raw && !a.out_pl) || (a.Cr & 63) ||
            lds < (size_t)2 * (sx_tile_n(cfg) / 64) * 4 * 1024)
            return hipErrorInvalidValue;
        epi |= SX_GATE | SX_HAS_RAW;  // (one instantiation: planar acts to out_raw, or operand planes to out_pl)
    }
    // (SX_RES_EARLY has compile-time instantiations only: where none matches, the residual is read in the epilogue)
    // (bf16x6 on the 64-row tile has no registers left for it: 256 VGPRs and a spill)
    const bool early_ok = rawin && (nprod == 2 || (nprod == 6 && cfg == 2)) && (a.flags & EPI_RES) && !(a.flags & (DBG_NO_DMA | DBG_NO_EPI)) &&
                          a.res == a.xr && a.ups == 1 && a.out_raw && !(a.flags & SX_NO_RAW_STORE) && !a.out_pl && !a.bias_b &&
                          a.oslope == 1.f && (!(a.flags & EPI_DIV) || (a.flags & EPI_ACC));
    if (early_ok && epi != -2) epi |= SX_RES_EARLY;
    if (epi != -2) {
        if ((epi & EPI_RES) && !((nprod == 1 || (epi & SX_RES_PL)) ? (const void *)a.res_pl : (const void *)a.res)) return hipErrorInvalidValue;
        if ((epi & EPI_ACC) && !a.out_raw) return hipErrorInvalidValue;
        a.flags = (a.flags & ~kSxEpiMask) | epi;
    } else
        a.flags = SX_WN_RMW | (a.flags & (EPI_ACC | EPI_RES | EPI_MASK | EPI_RELU | SX_PLANAR_STORE2 | SX_PLANAR_COUPLING));
    if (nprod == 1) {  // one fp16 plane, one product (BASELINE config 4)
        if (a.wscale == 0.f) a.wscale = 1.f;
        if (a.res_unslope == 0.f) a.res_unslope = 1.f;
        if ((a.flags & (SX_GATE | SX_WN_RMW)) || (a.res && !a.res_pl)) return hipErrorInvalidValue;
        return launch_conv_sx_h1_s16(a, cfg, epi, grid, lds, stream);
    }
    if (nprod == 2) {  // two fp16 planes, three products (fp32-grade): the same specialised epilogues
        if (a.wscale == 0.f) a.wscale = 1.f;
        if (s16 && (epi & SX_RES_PL)) return launch_conv_sx_f16_s16p(a, cfg, epi, grid, lds, stream);
        if (s16) return launch_conv_sx_f16_s16(a, cfg, epi, grid, lds, stream);
        return launch_conv_sx_f16_s32(a, cfg, epi, rawin, grid, lds, stream);
    }
    return launch_conv_sx_bf16(a, cfg, epi, nprod, rawin, grid, lds, stream);
}
#endif  // VITSMI_IMPL_SX

// The instantiation families, one translation unit each (tu_sx_*.hip: they compile side by side):
#ifdef VITSMI_IMPL_SX_S16
hipError_t launch_conv_sx_f16_s16(const SxArgs &a, int cfg, int epi, dim3 grid, size_t lds, hipStream_t stream) {
    switch (cfg) {
#if SX16_WIDE
        case 0: return launch_conv_sx_epi<1, 8, 4, 1, 2, 16>(a, epi, grid, lds, stream);
#else
        case 0: return launch_conv_sx_epi<2, 4, 2, 2, 2, 16>(a, epi, grid, lds, stream);
#endif
        case 1: return launch_conv_sx_epi<1, 4, 2, 2, 2, 16>(a, epi, grid, lds, stream);
        case 3: return launch_conv_sx_epi<1, 2, 2, 2, 2, 16>(a, epi, grid, lds, stream);
        default: return launch_conv_sx_epi<1, 2, 1, 4, 2, 16>(a, epi, grid, lds, stream);
    }
}
#endif
#ifdef VITSMI_IMPL_SX_S32
hipError_t launch_conv_sx_f16_s32(const SxArgs &a, int cfg, int epi, bool rawin, dim3 grid, size_t lds, hipStream_t stream) {
    if (rawin)
        return cfg == 1 ? launch_conv_sx_rawin<1, 4, 2, 2, 2>(a, epi, grid, lds, stream)
                        : launch_conv_sx_rawin<1, 2, 1, 4, 2>(a, epi, grid, lds, stream);
    if (a.prof && cfg == 0) return launch_conv_sx_k<2, 4, 2, 2, -1, true, false, 2>(a, grid, lds, stream);
    switch (cfg) {
#if SX_CFG0_WIDE
        case 0: return launch_conv_sx_epi<1, 8, 4, 1, 2>(a, epi, grid, lds, stream);
#else
        case 0: return launch_conv_sx_epi<2, 4, 2, 2, 2>(a, epi, grid, lds, stream);
#endif
        case 1: return launch_conv_sx_epi<1, 4, 2, 2, 2>(a, epi, grid, lds, stream);
        default: return launch_conv_sx_epi<1, 2, 1, 4, 2>(a, epi, grid, lds, stream);
    }
}
#endif
#ifdef VITSMI_IMPL_SX_S16P
// f16x3 with the residual stream kept as operand planes (SX_RES_PL): the epilogues of the plane-stream generator
template <int MW, int NW, int WM, int WN>
inline hipError_t launch_conv_sx_s16p_epi(const SxArgs &a, int epi, dim3 grid, size_t lds, hipStream_t stream) {
    switch (epi) {
        case kSxEpiInnerPl | SX_RES_PL: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiInnerPl | SX_RES_PL, false, false, 2, 16>(a, grid, lds, stream);
        case kSxEpiFirst | SX_RES_PL: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiFirst | SX_RES_PL, false, false, 2, 16>(a, grid, lds, stream);
        case kSxEpiAccum | SX_RES_PL: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiAccum | SX_RES_PL, false, false, 2, 16>(a, grid, lds, stream);
        case kSxEpiStageOut | SX_RES_PL: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiStageOut | SX_RES_PL, false, false, 2, 16>(a, grid, lds, stream);
        default: return launch_conv_sx_k<MW, NW, WM, WN, -1, false, false, 2, 16>(a, grid, lds, stream);
    }
}
hipError_t launch_conv_sx_f16_s16p(const SxArgs &a, int cfg, int epi, dim3 grid, size_t lds, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_conv_sx_s16p_epi<1, 8, 4, 1>(a, epi, grid, lds, stream);
        case 1: return launch_conv_sx_s16p_epi<1, 4, 2, 2>(a, epi, grid, lds, stream);
        case 3: return launch_conv_sx_s16p_epi<1, 2, 2, 2>(a, epi, grid, lds, stream);
        default: return launch_conv_sx_s16p_epi<1, 2, 1, 4>(a, epi, grid, lds, stream);
    }
}
#endif
#ifdef VITSMI_IMPL_SX_H1
// the single-plane mode's own epilogue set: planes (conv_pre, upsamplers, c1 of a ResBlock1 pair), residual conv inside a
// block (plane -> plane), block output into the fp32 multi-receptive-field sum, stage output as a plane
template <int MW, int NW, int WM, int WN>
inline hipError_t launch_conv_sx_h1_epi(const SxArgs &a, int epi, dim3 grid, size_t lds, hipStream_t stream) {
    switch (epi) {
        case kSxEpiPlanes: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiPlanes, false, false, 1, 16>(a, grid, lds, stream);
        case kSxEpiInnerPl: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiInnerPl, false, false, 1, 16>(a, grid, lds, stream);
        case kSxEpiFirst: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiFirst, false, false, 1, 16>(a, grid, lds, stream);
        case kSxEpiAccum: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiAccum, false, false, 1, 16>(a, grid, lds, stream);
        case kSxEpiAccum | EPI_DIV: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiAccum | EPI_DIV, false, false, 1, 16>(a, grid, lds, stream);
        case kSxEpiStageOut: return launch_conv_sx_k<MW, NW, WM, WN, kSxEpiStageOut, false, false, 1, 16>(a, grid, lds, stream);
        default: return launch_conv_sx_k<MW, NW, WM, WN, -1, false, false, 1, 16>(a, grid, lds, stream);
    }
}
hipError_t launch_conv_sx_h1_s16(const SxArgs &a, int cfg, int epi, dim3 grid, size_t lds, hipStream_t stream) {
    // (round 5, measured and not kept: the 128 x 256 tile as 2 x 2 waves of 64 x 128 - a B fragment then feeds two 32-row
    // blocks, half the LDS reads per MFMA, what this one-product mode looked short of - is SLOWER than the four 32 x 256 waves
    // f16x3 runs: headline batch 264.3 vs 277.5 M samples/s, default voice 1039 vs 1059 M; gpurun_out r05c)
    switch (cfg) {
        case 0: return launch_conv_sx_h1_epi<1, 8, 4, 1>(a, epi, grid, lds, stream);
        case 1: return launch_conv_sx_h1_epi<1, 4, 2, 2>(a, epi, grid, lds, stream);
        case 3: return launch_conv_sx_h1_epi<1, 2, 2, 2>(a, epi, grid, lds, stream);
        default: return launch_conv_sx_h1_epi<1, 2, 1, 4>(a, epi, grid, lds, stream);
    }
}
#endif
#ifdef VITSMI_IMPL_SX_BF16
hipError_t launch_conv_sx_bf16(const SxArgs &a, int cfg, int epi, int nprod, bool rawin, dim3 grid, size_t lds, hipStream_t stream) {
    if (nprod != 6) return hipErrorInvalidValue;
    if (rawin)
        return cfg == 1 ? launch_conv_sx_rawin<1, 4, 2, 2>(a, epi, grid, lds, stream)
                        : launch_conv_sx_rawin<1, 2, 1, 4>(a, epi, grid, lds, stream);
    if (a.prof && cfg == 0) return launch_conv_sx_k<2, 4, 2, 2, -1, true>(a, grid, lds, stream);
    switch (cfg) {
        case 0: return launch_conv_sx_epi<2, 4, 2, 2>(a, epi, grid, lds, stream);
        case 1: return launch_conv_sx_epi<1, 4, 2, 2>(a, epi, grid, lds, stream);
        default: return launch_conv_sx_epi<1, 2, 1, 4>(a, epi, grid, lds, stream);
    }
}
#endif

#ifndef VITSMI_TU  // (plain kernels: compiled by vitsmi.hip only, not by the instantiation units tu_*.hip)
// ---- layout conversion kernels ------------------------------------------------------------------------

// planar fp32 x[b][c][t] (row pitch `pitch`, optionally masked by t < len[b]) -> planes [3][C/8][T][8]
// (f16 = 1: two fp16 planes in the same addressing, plane 2 untouched; f16 = 2: the single fp16 plane of the NP = 1 mode)
// peak (f16 only, may be nullptr): 64 range-guard slots as in SxArgs::peak.  This is where tensors ENTER the split
// engine (the generator's z, the flow's WN input), so non-finite values are caught here too: NaN counts as inf.
__global__ __launch_bounds__(256) void sx_split_planes_kernel(const float *x, int64_t x_bstride, int pitch, const int *len,
                                                              uint16_t *out, int C, int T, int f16 = 0, unsigned *peak = nullptr) {
    const int t = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;
    float pk = 0.f;
    if (t < T) {
        const bool live = !len || t < len[b];
        const float *xb = x + (int64_t)b * x_bstride + (int64_t)cg * 8 * pitch + t;
        unsigned short p[3][8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float v = live ? xb[(int64_t)e * pitch] : 0.f;
            if (f16) {
                pk = !(__builtin_fabsf(v) <= kF16Max) ? __builtin_inff() : __builtin_fmaxf(pk, __builtin_fabsf(v));
                const float vc = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
                const _Float16 h0 = (_Float16)vc, h1 = (_Float16)((vc - (float)h0) * 2048.f);
                p[0][e] = __builtin_bit_cast(unsigned short, h0);
                p[1][e] = __builtin_bit_cast(unsigned short, h1);
                p[2][e] = 0;
            } else
                split3(v, p[0][e], p[1][e], p[2][e]);
        }
        const int CG = C >> 3;
        uint16_t *ob = out + (int64_t)b * 3 * CG * T * 8;
#pragma unroll
        for (int pl = 0; pl < (f16 == 2 ? 1 : (f16 ? 2 : 3)); pl++) {
            u32x4 w;
            w.x = (unsigned)p[pl][0] | ((unsigned)p[pl][1] << 16);
            w.y = (unsigned)p[pl][2] | ((unsigned)p[pl][3] << 16);
            w.z = (unsigned)p[pl][4] | ((unsigned)p[pl][5] << 16);
            w.w = (unsigned)p[pl][6] | ((unsigned)p[pl][7] << 16);
            *reinterpret_cast<u32x4 *>(ob + (((int64_t)pl * CG + cg) * T + t) * 8) = w;
        }
    }
    if (f16 && peak) sx_publish_peak(peak, (int)(blockIdx.x + blockIdx.y + blockIdx.z), pk);  // (uniform: whole waves)
}

// planar fp32 x[b][c][t] (row pitch `pitch`, optionally masked by t < len[b]) -> raw fp32 [C/8][T][8]
__global__ __launch_bounds__(256) void sx_block_kernel(const float *x, int64_t x_bstride, int pitch, const int *len,
                                                       float *raw, int C, int T, unsigned *peak = nullptr) {
    const int t = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;
    float pk = 0.f;  // range guard of the f16 arithmetic (see sx_split_planes_kernel): this tensor enters a raw-input conv
    if (t < T) {
        const bool live = !len || t < len[b];
        const float *xb = x + (int64_t)b * x_bstride + (int64_t)cg * 8 * pitch + t;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float v = live ? xb[(int64_t)e * pitch] : 0.f;
            pk = !(__builtin_fabsf(v) <= kF16Max) ? __builtin_inff() : __builtin_fmaxf(pk, __builtin_fabsf(v));
            raw[(int64_t)b * C * T + ((int64_t)cg * T + t) * 8 + e] = v;
        }
    }
    if (peak) sx_publish_peak(peak, (int)(blockIdx.x + blockIdx.y + blockIdx.z), pk);  // (uniform: whole waves)
}

// raw fp32 [C/8][T][8] (or, with planes != nullptr, the sum of the three planes) -> planar [C][T]
__global__ __launch_bounds__(256) void sx_unblock_kernel(const float *raw, const uint16_t *planes, float *out, int C, int T,
                                                         int f16 = 0) {
    const int t = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const int CG = C >> 3;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        float v;
        if (planes) {
            const uint16_t *pb = planes + (int64_t)b * 3 * CG * T * 8;
            const int64_t o = ((int64_t)cg * T + t) * 8 + e, ps = (int64_t)CG * T * 8;
            // same order as the exact reconstruction: small terms first
            if (f16 == 2) v = f16_bits_to_f32(pb[o]);
            else if (f16) v = f16_bits_to_f32(pb[o + ps]) * (1.f / 2048.f) + f16_bits_to_f32(pb[o]);
            else v = (bf16_bits_to_f32(pb[o + 2 * ps]) + bf16_bits_to_f32(pb[o + ps])) + bf16_bits_to_f32(pb[o]);
        } else
            v = raw[(int64_t)b * C * T + ((int64_t)cg * T + t) * 8 + e];
        out[(int64_t)b * C * T + (int64_t)(cg * 8 + e) * T + t] = v;
    }
}

#endif  // VITSMI_TU

}  // namespace vitsmi
