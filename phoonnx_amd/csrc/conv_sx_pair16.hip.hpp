// conv_sx_pair16.hip.hpp — two dependent convs of a ResBlock in ONE launch on the v_mfma_f32_16x16x32_f16 loop, for the
// generator's stages of 32 and 64 channels, in both fp16 arithmetics:
//   NPL = 2  f16x3: tensors in HBM are the two fp16 operand planes (h0, h1' = the low part 2^11 up) of the consumer's
//                   leaky_relu; three products per fp32 product
//   NPL = 1  f16  : tensors in HBM are single fp16 planes holding the consumer's leaky_relu; one product (BASELINE config 4)
// In both, the tensor between two ResBlock steps is stored ONCE, as the operand planes the next conv reads; the residual is
// recovered from them (leaky_relu undone: exact up to the planes' own resolution), and the multi-receptive-field sum is fp32.
//   PAIR   a ResBlock1 step    out = c2(lrelu(c1(lrelu(x)))) + x                 (phoonnx_train/vits/modules.py:301-314)
//   CHAIN  two ResBlock2 steps x1 = c1(lrelu(x)) + x ;  out = c2(lrelu(x1)) + x1  (modules.py:355-364)
// It replaces conv_sx_pair_kernel (conv_sx_pair.hip.hpp, the 32x32x16 form) where the shapes allow.  What changed and why:
//   * k = 32 per MFMA step: a 32-channel conv is ONE step per tap (3 steps for k = 3 instead of 6), a 64-channel one two;
//     these shallow reductions were chains of short dependent steps, each waiting for its weights.
//   * the whole tile's operands live in LDS in the B-fragment layout of the 16x16x32 shape ([plane][8-channel group][column]
//     cells, a lane's ds_read_b128 = 8 channels of one column); the x tile stays resident next to Y, so the residual is READ
//     BACK FROM LDS (NPL = 1: the stored fp16 with the leaky_relu undone; NPL = 2: h0 + h1' 2^-11, then undone) instead of
//     being held in 32-64 registers from the prologue on, or re-read from HBM (PMC: 2.07x the tensor on the 64-channel k = 3
//     step).  Where x + Y do not fit the occupancy target, Y overlays x (OVL) and the residual waits in registers from the
//     hand-over on.
//   * the x tile arrives by LDS-DMA (no registers, no VALU): the stored planes ARE the B operand.
//   * small register footprint (no prologue-to-epilogue residual, 8-16 accumulator registers per 32 columns) and 128- or
//     256-column tiles: 3-6 workgroups per CU instead of 2-3, so that one workgroup's load / hand-over / store phases fall
//     under another's MFMAs.
// Arithmetic (NPL = 2): the products and their order per output element are those of conv_sx_kernel's 16x16x32 loop; the
// residual is x reconstructed from its operand planes, |x' - x| <= 2^-22 |x| (two fp16 planes of leaky_relu(x) carry 22 bits),
// so results differ from the two-launch form in the last bits (not bit-identical; tests: against float64 / the oracle at the
// tolerances of the engine).
#pragma once
#include "conv_sx_engine.hip.hpp"

namespace vitsmi {

enum : int { P16_HAS_RAW = 1 << 9, P16_HAS_PL = 1 << 10 };  // (EPI_ACC / EPI_DIV as everywhere)
#ifndef P16_PROF
#define P16_PROF 0  // diagnostic build: s_memtime stamps at the phase boundaries, summed per launch into SxPair16Args::prof
#endif

struct SxPair16Args {
    const uint16_t *xpl;      // input planes [NPL][C/8][T][8] holding leaky_relu(x, islope) (also the residual)
    int64_t x_bstride;        // elements between batch items of xpl
    float islope, mslope;     // leaky-ReLU slopes: what the input planes were stored with, and between the convs
    float un_islope;          // 1 / islope
    int T;
    const u32x4 *wp1, *wp2;   // packed weights: 16x16x32 layout, NPL planes, tile height C (pack_conv_sx s16 / h1)
    const float *bias1, *bias2;
    float wscale1, wscale2;
    float *out_raw;           // fp32 raw [B][C/8][T][8]: destination (P16_HAS_RAW) and / or EPI_ACC operand
    int64_t raw_bstride;
    uint16_t *out_pl;         // operand planes of leaky_relu(result, oslope) (P16_HAS_PL)
    int64_t pl_bstride;
    float oslope;
    const float *zeros;       // >= 1 KiB of zeros
    int K1, dil1, pad1, K2, dil2, pad2;
    int LW1, RS1, RS2;        // x tile width (cells), row strides of the x tile / of Y (cells, multiples of 16)
    unsigned magic1;          // ceil(2^32 / RS1)
    unsigned y_off;           // byte offset of Y in LDS (0: Y overlays the x tile)
    unsigned b_off;           // byte offset of the two bias vectors in LDS (2 x C floats)
    int BNo, NT, B;
    int flags;                // EPI_ACC | EPI_DIV | P16_HAS_RAW | P16_HAS_PL
    float div;
    unsigned *peak;
    unsigned long long *prof;  // (P16_PROF builds) 8 counters: x landed, converted + barrier, phase 1, hand-over, phase 2, epilogue, stores drained, workgroups
    SxRagged rag;             // per-utterance tensor ends of a padded batch (conv_sx_engine.hip.hpp)
};

// C channels (32 | 64), NPL planes, BN columns per tile (128 | 256), OVL: Y overlays the x tile
template <int C, int NPL, int BN, bool CHAIN, bool OVL, int WPS>
__global__ __launch_bounds__(256, WPS) void conv_sx_pair16_kernel(SxPair16Args a) {
    constexpr bool H1 = NPL == 1;
    constexpr int WM = C / 32, WN = 4 / WM, BNW = BN / WN, NCB = BNW / 16, NQ = NCB / 2, NCH = C / 32, CG = C / 8;
    static_assert((C == 32 || C == 64) && (NPL == 1 || NPL == 2) && (BN == 128 || BN == 256) && NQ >= 1, "shape");
    // 16-column blocks per B unit (a unit = what one buffer slot holds): two, or one where registers are short (f16x3 at 64
    // channels: 32 accumulator + 32 residual + 48 weight registers) or the wave has only two blocks
    constexpr int UCB = (NQ == 1 || (NPL == 2 && C == 64)) ? 1 : 2;
    constexpr int NU = NCB / UCB;              // units per step (even)
    constexpr int RPU = UCB * NPL;             // ds_read_b128 per unit
    constexpr int NAL = 2 * NPL;               // global loads per weight set (one step: two 16-row sub-blocks x planes)
    constexpr int D = H1 ? 3 : 2;              // weight look-ahead in steps
    constexpr int NS = D + 1;                  // weight register sets
    constexpr int BLKBYTES = 2 * NPL * 1024, STEPBYTES = WM * BLKBYTES;
    static_assert(NU % 2 == 0 && (RPU == 1 || RPU == 2 || RPU == 4), "");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_sx[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int tile_nb = (int)blockIdx.x;
    if (tile_nb >= a.NT * a.B) return;
    const int b = tile_nb / a.NT, t0 = (tile_nb - b * a.NT) * a.BNo;  // first kept output column
    const int t1 = t0 - a.pad2;                                        // first column phase 1 computes
    const int T = a.T, LW1 = a.LW1, RS1 = a.RS1, RS2 = a.RS2;
    const int TV = __builtin_amdgcn_readfirstlane(sx_valid_cols(a.rag, b, T));  // this utterance's tensor end (SxRagged); T = row pitch
    if (t0 >= TV) return;                                                        // (uniform exit) no kept column lies inside it
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_sx;
    const uint32_t ylds = lds0 + a.y_off;
    const uint32_t XPB = (uint32_t)(CG * RS1) * 16u, YPB = (uint32_t)(CG * RS2) * 16u;  // bytes per plane
    const char *wbase1 = reinterpret_cast<const char *>(a.wp1) + wm * BLKBYTES;
    const char *wbase2 = reinterpret_cast<const char *>(a.wp2) + wm * BLKBYTES;
    float pk = 0.f;
    // Every kernel argument the later phases use is read NOW and pinned in scalar registers: hipcc sinks a kernarg load
    // (s_load) to the block of its first use - e.g. in front of the tail steps of a phase - and a scalar load in flight counts in
    // lgkmcnt and returns out of order: the counted `s_waitcnt lgkmcnt(n)` of the operand pipeline would then let an MFMA
    // read a register its ds_read has not written yet (seen: every parity case failing after one more late-read argument).
    const int k_flags = a.flags, k_pad2 = a.pad2, k_BNo = a.BNo, k_K2 = a.K2, k_dil2 = a.dil2;
    const unsigned k_b_off = a.b_off;
    float *const k_out_raw = a.out_raw;
    uint16_t *const k_out_pl = a.out_pl;
    const int64_t k_raw_bs = a.raw_bstride, k_pl_bs = a.pl_bstride;
    const float k_ws1 = a.wscale1, k_ws2 = a.wscale2, k_msl = a.mslope, k_osl = a.oslope, k_div = a.div, k_unisl = a.un_islope;
    unsigned *const k_peak = a.peak;
    asm volatile("" ::"s"(k_flags), "s"(k_pad2), "s"(k_BNo), "s"(k_K2), "s"(k_dil2), "s"(k_b_off), "s"(k_out_raw), "s"(k_out_pl));
    asm volatile("" ::"s"(k_raw_bs), "s"(k_pl_bs), "s"(__float_as_uint(k_ws1)), "s"(__float_as_uint(k_ws2)), "s"(__float_as_uint(k_msl)),
                 "s"(__float_as_uint(k_osl)), "s"(__float_as_uint(k_div)), "s"(__float_as_uint(k_unisl)), "s"(k_peak), "s"(wbase2));
    asm volatile("" ::"s"(a.y_off), "s"(a.pad1), "s"(a.K1), "s"(a.dil1), "s"(a.bias1), "s"(a.bias2), "s"(a.zeros));

    struct ASet {
        u32x4 f[2][NPL];
    };
    const uint32_t voff0 = (uint32_t)lane * 16u;
    auto load_a = [&](ASet &f, const char *wb, int step) __attribute__((always_inline)) {
        const uint64_t pa = reinterpret_cast<uint64_t>(wb) + (uint64_t)((int64_t)step * STEPBYTES);
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
    ASet fs[NS];
    auto prefetch_a = [&](const char *wb, int S) __attribute__((always_inline)) {  // the first D sets of a conv
        static_for<D>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (i < S) load_a(fs[i], wb, i);
        });
    };

#if P16_PROF
    auto stamp = [&]() {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    unsigned long long tp[8];
    tp[0] = stamp();
#define P16_STAMP(i) tp[i] = stamp()
#else
#define P16_STAMP(i)
#endif
    // =================================================================== prologue: the whole x tile -> LDS
    // x tile column 0 = time t1 - pad1; cells outside the tensor are zero (the convs' zero padding)
    const int S1 = NCH * a.K1, S2 = NCH * k_K2;
    {
        // LDS-DMA, 16 bytes per lane: the flattened [plane][row][RS1] cell space in rounds of 256 cells; a lane whose cell is
        // padding (column past LW1, row past the tile, time outside the tensor) reads the zero page.  (Plane p, group g of
        // the tensor is row p * C/8 + g of [NPL][C/8][T] cells: the same index as in the tile.)
        const uint16_t *xb = a.xpl + (int64_t)b * a.x_bstride;
        const int ncell = NPL * CG * RS1, nit = (ncell + 255) >> 8;
        for (int it = 0; it < nit; it++) {
            const int base = it * 256 + wave * 64;
            const int i = base + lane;
            const int row = (int)__umulhi((unsigned)i, a.magic1);
            const int col = i - row * RS1;
            const int t = t1 - a.pad1 + col;
            const bool ok = row < NPL * CG && col < LW1 && t >= 0 && t < TV;
            const void *src = ok ? static_cast<const void *>(xb + ((int64_t)row * T + t) * 8)
                                 : static_cast<const void *>(reinterpret_cast<const char *>(a.zeros) + lane * 16);
            // (lanes past the tile's last cell are masked off: nothing is written behind the allocation)
            if (i < ncell) lds_dma<16>(src, reinterpret_cast<float *>(lds_sx + (size_t)base * 16));
        }
        // both bias vectors ride along (one DMA of wave 0): read from LDS in the hand-over / the epilogue - as dependent global
        // loads at those points each cost an exposed L2 round trip per tile (r04f stamps: ~1.5 k cycles, twice)
        if (wave == 0) {
            const int k = lane - (lane >= C / 4 ? C / 4 : 0);
            const float *bp = lane < C / 4 ? a.bias1 : a.bias2;
            const void *src = (bp && lane < C / 2) ? static_cast<const void *>(bp + 4 * k) : static_cast<const void *>(a.zeros + 4 * (lane & 31));
            lds_dma<16>(src, reinterpret_cast<float *>(lds_sx + a.b_off));  // (all 64 lanes: the slot is 1 KiB)
        }
        prefetch_a(wbase1, S1);
        // (vector-memory operations retire in order: when only the weight requests are in flight the tile has landed)
        constexpr int NA0 = D * NAL;
        static_assert(NA0 == 6 || NA0 == 8, "");
        if (S1 >= D) {
            if constexpr (NA0 == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        P16_STAMP(1);
    }
    __builtin_amdgcn_s_barrier();  // the x tile is complete
    __builtin_amdgcn_sched_barrier(0);
    P16_STAMP(2);

    // =================================================================== one conv over an operand resident in LDS
    f32x4 c16[2][NCB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int aa = 0; aa < 2; aa++)
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) c16[aa][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    u32x4 bq[2][UCB][NPL];  // [slot][16-column block of the unit][plane]
    auto load_b = [&](auto SLOT, auto U, uint32_t rows, uint32_t pstride) __attribute__((always_inline)) {
        constexpr int sl = decltype(SLOT)::value, u = decltype(U)::value;
        static_for<UCB>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            // (immediate offsets: unit u, block k of it = (u * UCB + k) * 16 columns of 16 bytes)
            bq[sl][k][0] = ds_read128<(u * UCB + k) * 256>(rows);
            if constexpr (!H1) bq[sl][k][1] = ds_read128<(u * UCB + k) * 256>(rows + pstride);
        });
    };
    auto wait_b = [&](bool more) __attribute__((always_inline)) {  // the older unit's reads have landed
        if (!more) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (RPU == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else if constexpr (RPU == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    };
    auto mma_u = [&](const ASet &f, auto SLOT, auto U) __attribute__((always_inline)) {
        constexpr int sl = decltype(SLOT)::value, u = decltype(U)::value;
        if constexpr (H1) {
#pragma unroll
            for (int aa = 0; aa < 2; aa++)
#pragma unroll
                for (int k = 0; k < UCB; k++)
                    c16[aa][u * UCB + k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f.f[aa][0]),
                                                                                   __builtin_bit_cast(f16x8, bq[sl][k][0]),
                                                                                   c16[aa][u * UCB + k], 0, 0, 0);
        } else {
            // (f16x3: g1*h0, g0'*h1', g0*h0 - the order of conv_sx_kernel)
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int aa = 0; aa < 2; aa++)
#pragma unroll
                    for (int k = 0; k < UCB; k++) {
                        const f16x8 ga = c == 0 ? __builtin_bit_cast(f16x8, f.f[aa][NPL - 1])
                                                : (c == 1 ? __builtin_bit_cast(f16x8, f.f[aa][0]) * (_Float16)0.00048828125f
                                                          : __builtin_bit_cast(f16x8, f.f[aa][0]));
                        c16[aa][u * UCB + k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                            ga, __builtin_bit_cast(f16x8, bq[sl][k][c == 1 ? NPL - 1 : 0]), c16[aa][u * UCB + k], 0, 0, 0);
                    }
        }
    };
    auto wait_a = [&](int younger) __attribute__((always_inline)) {  // A(s) has landed when only `younger` later sets are in flight
        const int n = younger * NAL;
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        }
    };
    const std::integral_constant<int, 0> I0{};
    // rows0: this lane's cell of (plane 0, chunk 0, tap 0, unit 0): row = lane >> 4, column = the wave's first + (lane & 15).
    // The first D weight sets must have been requested (prefetch_a).
    auto run_conv = [&](const char *wb, int K, int dil, uint32_t rows0, uint32_t row_bytes, uint32_t pstride) __attribute__((always_inline)) {
        const int S = NCH * K;
        int chunk = 0, tap = 0;
        load_b(I0, I0, rows0, pstride);
        auto step = [&](ASet &fc, ASet &fload, int s) __attribute__((always_inline)) {
            const int left = S - 1 - s;
            wait_a(left < D - 1 ? left : D - 1);
            __builtin_amdgcn_sched_barrier(0);
            if (s + D < S) load_a(fload, wb, s + D);
            int ntap = tap + 1, nchunk = chunk;
            if (ntap == K) {
                ntap = 0;
                nchunk++;
            }
            const uint32_t cur = rows0 + (uint32_t)chunk * (4u * row_bytes) + (uint32_t)(tap * dil) * 16u;
            const uint32_t next = rows0 + (uint32_t)nchunk * (4u * row_bytes) + (uint32_t)(ntap * dil) * 16u;
            const bool more = s + 1 < S;
            static_for<NU>([&](auto U) {
                constexpr int u = decltype(U)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (u + 1 < NU) {
                    load_b(std::integral_constant<int, (u + 1) & 1>{}, std::integral_constant<int, u + 1>{}, cur, pstride);
                    wait_b(true);
                } else {
                    if (more) load_b(I0, I0, next, pstride);
                    wait_b(more);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma_u(fc, std::integral_constant<int, u & 1>{}, U);
            });
            __builtin_amdgcn_sched_barrier(0);
            tap = ntap;
            chunk = nchunk;
        };
        // (no exit from the middle of the unrolled group: a mid-loop break makes hipcc copy the accumulators)
        int s = 0;
        for (; s + NS <= S; s += NS)
            static_for<NS>([&](auto I) {
                constexpr int i = decltype(I)::value;
                step(fs[i], fs[(i + D) % NS], s + i);
            });
        // hipcc sinks kernel-argument loads (s_load) that only the epilogue uses to THIS point, the block between the unrolled
        // loop and its tail.  A scalar load in flight counts in lgkmcnt and returns out of order, so the tail steps' counted
        // `s_waitcnt lgkmcnt(n)` could pass with a ds_read still outstanding: drain the counter once here (per tile, not per step).
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        static_for<NS - 1>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (s + i < S) step(fs[i], fs[(i + D) % NS], s + i);
        });
    };
    // 16 x 16 accumulators -> the 32 x 32 layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)): the packer
    // permuted the rows of each 16-row sub-block so that one half-row swap per register pair does it (conv_sx_engine.hip.hpp)
    // The swap is done IN PLACE (a second 32 x 32 array next to c16 cost the 64-channel variants their third workgroup per
    // CU): register r of 32-column block n is then ACC(n, r).
    auto gather_acc = [&]() {
#pragma unroll
        for (int n = 0; n < NQ; n++)
#pragma unroll
            for (int aa = 0; aa < 2; aa++)
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(c16[aa][2 * n][rr]),
                                                                     __float_as_uint(c16[aa][2 * n + 1][rr]), false, false);
                    c16[aa][2 * n][rr] = __uint_as_float(sw[0]);
                    c16[aa][2 * n + 1][rr] = __uint_as_float(sw[1]);
                }
    };
#define ACC(n, r) c16[(r) >> 3][2 * (n) + (((r) >> 2) & 1)][(r) & 3]
    // x at tile column j, channels 32 wm + 8 q + 4 hi .. + 3 (q = 0..3: the lane's rows of a 32 x 32 block), from the resident
    // operand planes with the leaky-ReLU undone; the 4 (8) reads travel together
    const float un_isl = k_unisl;
    auto read_x16 = [&](int j, f32x4 (&o)[4]) __attribute__((always_inline)) {
        const uint32_t ad = lds0 + (uint32_t)((4 * wm) * RS1 + j + a.pad1) * 16u + 8u * hi;
        const uint32_t rb = (uint32_t)RS1 * 16u;
        u32x2 w0[4], w1[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            asm volatile("ds_read_b64 %0, %1" : "=v"(w0[q]) : "v"(ad + (uint32_t)q * rb) : "memory");
            if constexpr (!H1) asm volatile("ds_read_b64 %0, %1" : "=v"(w1[q]) : "v"(ad + (uint32_t)q * rb + XPB) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);  // (see the bias reads: nothing else ties the conversions below to the wait)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float v[4];
            if constexpr (H1) {
                unact4h(w0[q], 1.f, v);  // (conversion only; the slope is undone below)
            } else {
                const unsigned a0 = w0[q].x, a1 = w0[q].y, b0 = w1[q].x, b1 = w1[q].y;
                const f16x2 h00 = __builtin_bit_cast(f16x2, a0), h01 = __builtin_bit_cast(f16x2, a1);
                const f16x2 h10 = __builtin_bit_cast(f16x2, b0), h11 = __builtin_bit_cast(f16x2, b1);
                v[0] = __builtin_fmaf((float)h10[0], 1.f / 2048.f, (float)h00[0]);
                v[1] = __builtin_fmaf((float)h10[1], 1.f / 2048.f, (float)h00[1]);
                v[2] = __builtin_fmaf((float)h11[0], 1.f / 2048.f, (float)h01[0]);
                v[3] = __builtin_fmaf((float)h11[1], 1.f / 2048.f, (float)h01[1]);
            }
#pragma unroll
            for (int e = 0; e < 4; e++) o[q][e] = v[e] < 0.f ? v[e] * un_isl : v[e];
        }
    };

    // =================================================================== phase 1: c1 over columns [t1, t1 + BN)
    zero_acc();
    run_conv(wbase1, a.K1, a.dil1, lds0 + (uint32_t)((lane >> 4) * RS1 + wn * BNW + (lane & 15)) * 16u, (uint32_t)RS1 * 16u, XPB);
    P16_STAMP(3);
    prefetch_a(wbase2, S2);  // c2's first weights travel during the hand-over
    gather_acc();

    // =================================================================== hand-over: c1's output -> Y (operand planes in LDS)
    // pre: the residual in the accumulator layout - CHAIN: x1 = c1(..) + x (phase 2's residual); PAIR with OVL: x, saved
    // before Y overwrites the tile.  (PAIR without OVL reads x from LDS in the epilogue: no registers.)
    constexpr bool KEEP = CHAIN || OVL;
    f32x4 pre[KEEP ? NQ : 1][4];
    if constexpr (KEEP) {
#pragma unroll
        for (int n = 0; n < NQ; n++) read_x16(wn * BNW + n * 32 + l31, pre[n]);
    }
    if constexpr (OVL) {
        __builtin_amdgcn_s_barrier();  // every wave has finished reading the x tile Y is about to overwrite
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        const float wsc = k_ws1, msl = k_msl;
        f32x4 bq4[4];
        {
            u32x4 t4[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                asm volatile("ds_read_b128 %0, %1" : "=v"(t4[q]) : "v"(lds0 + k_b_off + (uint32_t)(wm * 32 + 8 * q + 4 * hi) * 4u) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (the consumers of t4 have no data dependence on the wait statement: without this fence the scheduler may hoist
            // them above it and read registers the ds_reads have not written yet)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; q++) bq4[q] = __builtin_bit_cast(f32x4, t4[q]);
        }
#pragma unroll
        for (int n = 0; n < NQ; n++) {
            const int j = wn * BNW + n * 32 + l31;
            const int t = t1 + j;
            const bool live = t >= 0 && t < TV;  // outside the tensor c2 sees zero padding, not c1 evaluated there
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float v = __builtin_fmaf(ACC(n, 4 * q + e), wsc, bq4[q][e]);
                    if constexpr (CHAIN) {
                        v += pre[n][q][e];
                        pre[n][q][e] = v;
                    }
                    o[e] = live ? fmaxf(v, v * msl) : 0.f;
                }
                const uint32_t cell = ylds + (uint32_t)((4 * wm + q) * RS2 + j + k_pad2) * 16u + 8u * hi;
                if constexpr (H1) {
                    const unsigned wa = cvt1h_pair_pk(o[0], o[1], pk), wb = cvt1h_pair_pk(o[2], o[3], pk);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(cell), "v"(u32x2{wa, wb}) : "memory");
                } else {
                    unsigned wa[2], wb[2];
                    split2h_pair_pk(o[0], o[1], wa[0], wa[1], pk);
                    split2h_pair_pk(o[2], o[3], wb[0], wb[1], pk);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(cell), "v"(u32x2{wa[0], wb[0]}) : "memory");
                    asm volatile("ds_write_b64 %0, %1" ::"v"(cell + YPB), "v"(u32x2{wa[1], wb[1]}) : "memory");
                }
            }
        }
    }
    zero_acc();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // Y is complete
    __builtin_amdgcn_sched_barrier(0);
    P16_STAMP(4);

    // =================================================================== phase 2: c2 over Y (stored pad2 columns to the right)
    run_conv(wbase2, k_K2, k_dil2, ylds + (uint32_t)((lane >> 4) * RS2 + wn * BNW + (lane & 15)) * 16u, (uint32_t)RS2 * 16u, YPB);
    P16_STAMP(5);
    gather_acc();

    // =================================================================== epilogue: bias2 + residual [+ xs] [/ n] -> raw / plane
    {
        const int flags = k_flags;
        float *rawb = k_out_raw + (int64_t)b * k_raw_bs;
        uint16_t *plb = k_out_pl + (int64_t)b * k_pl_bs;
        const float wsc = k_ws2, rdiv = k_div, osl = k_osl;
        f32x4 bq4[4];
        {
            u32x4 t4[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                asm volatile("ds_read_b128 %0, %1" : "=v"(t4[q]) : "v"(lds0 + k_b_off + (uint32_t)(C + wm * 32 + 8 * q + 4 * hi) * 4u) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (the consumers of t4 have no data dependence on the wait statement: without this fence the scheduler may hoist
            // them above it and read registers the ds_reads have not written yet)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; q++) bq4[q] = __builtin_bit_cast(f32x4, t4[q]);
        }
#pragma unroll
        for (int n = 0; n < NQ; n++) {
            const int j = wn * BNW + n * 32 + l31;
            const int t = t1 + j;
            const bool kept = j >= k_pad2 && j < k_pad2 + k_BNo && t < TV;  // overlap columns belong to the neighbours
            const int tl = t < 0 ? 0 : (t < T ? t : T - 1);
            f32x4 adl[4], xres[4];
            if constexpr (!KEEP) read_x16(j, xres);
            if (flags & EPI_ACC) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    adl[q] = *reinterpret_cast<const f32x4 *>(rawb + ((int64_t)(4 * wm + q) * T + tl) * 8 + 4 * hi);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = __builtin_fmaf(ACC(n, 4 * q + e), wsc, bq4[q][e]);
                if constexpr (KEEP) v += pre[n][q];
                else v += xres[q];
                if (flags & EPI_ACC) v += adl[q];
                if (flags & EPI_DIV) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = v[e] / rdiv;
                }
                if (!kept) continue;
                const int64_t cell = ((int64_t)(4 * wm + q) * T + t) * 8 + 4 * hi;
                if (flags & P16_HAS_RAW) *reinterpret_cast<f32x4 *>(rawb + cell) = v;
                if (flags & P16_HAS_PL) {
                    if constexpr (H1) {
                        const unsigned wa = cvt1h_pair_pk(fmaxf(v[0], v[0] * osl), fmaxf(v[1], v[1] * osl), pk);
                        const unsigned wb = cvt1h_pair_pk(fmaxf(v[2], v[2] * osl), fmaxf(v[3], v[3] * osl), pk);
                        *reinterpret_cast<u32x2 *>(plb + cell) = u32x2{wa, wb};
                    } else {
                        unsigned wa[2], wb[2];
                        split2h_pair_pk(fmaxf(v[0], v[0] * osl), fmaxf(v[1], v[1] * osl), wa[0], wa[1], pk);
                        split2h_pair_pk(fmaxf(v[2], v[2] * osl), fmaxf(v[3], v[3] * osl), wb[0], wb[1], pk);
                        *reinterpret_cast<u32x2 *>(plb + cell) = u32x2{wa[0], wb[0]};
                        *reinterpret_cast<u32x2 *>(plb + (int64_t)CG * T * 8 + cell) = u32x2{wa[1], wb[1]};
                    }
                }
            }
        }
    }
#undef ACC
#if P16_PROF
    P16_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P16_STAMP(7);
    if (a.prof && tid == 0) {  // (a row per workgroup, plain stores: atomics on eight shared words distorted what they measured)
        for (int i = 0; i < 7; i++) a.prof[(size_t)blockIdx.x * 8 + i] = tp[i + 1] - tp[i];
        a.prof[(size_t)blockIdx.x * 8 + 7] = tp[0];
    }
#endif
    if (k_peak) sx_publish_peak(k_peak, (int)blockIdx.x, pk);  // (uniform branch; every thread arrives)
}

// Geometry of a fused pair at tile width BN; false when it does not fit
struct SxPair16Geom {
    int LW1, RS1, RS2, BNo;
    unsigned y_off, b_off;
    size_t lds;
};
inline bool sx_pair16_geom(int C, int npl, int BN, bool ovl, int K1, int dil1, int K2, int dil2, SxPair16Geom *g) {
    if (K1 < 1 || K2 < 1 || dil1 < 1 || dil2 < 1 || !(K1 & 1) || !(K2 & 1)) return false;
    const int halo1 = (K1 - 1) * dil1, halo2 = (K2 - 1) * dil2, pad2 = halo2 / 2;
    if (halo1 > 96) return false;
    const int LW1 = BN + halo1;
    const int RS1 = (LW1 + 15) / 16 * 16;
    // Y: BN + pad2 written columns; the discarded outputs right of the kept ones read up to pad2 cells past a row's end - the
    // next row's cells, or (last row) the slack behind it
    const int RS2 = (BN + pad2 + 15) / 16 * 16;
    const size_t xb = (size_t)npl * (C / 8) * RS1 * 16, yb = (size_t)npl * (C / 8) * RS2 * 16 + (size_t)(pad2 + 16) * 16;
    if (g) {
        g->LW1 = LW1;
        g->RS1 = RS1;
        g->RS2 = RS2;
        g->BNo = BN - halo2;
        g->y_off = ovl ? 0u : (unsigned)xb;
        g->b_off = (unsigned)(ovl ? (xb > yb ? xb : yb) : xb + yb);
        g->lds = (size_t)g->b_off + 1024;  // (the two bias vectors: one 1 KiB DMA slot)
    }
    return BN - halo2 >= BN / 2 + BN / 8;  // (more than 37 % of a tile recomputed: not worth it)
}

hipError_t launch_conv_sx_pair16(SxPair16Args a, int C, int npl, int B, hipStream_t stream, bool chain);
// which tile a (C, npl, halo) combination runs on: 0 = unsupported, else BN (128 | 256) and whether Y overlays x
int sx_pair16_plan(int C, int npl, int K1, int dil1, int K2, int dil2, bool *ovl);

#ifdef VITSMI_IMPL_PAIR16
template <int C, int NPL, int BN, bool CHAIN, bool OVL, int WPS>
inline hipError_t launch_conv_sx_pair16_k(const SxPair16Args &a, dim3 grid, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv_sx_pair16_kernel<C, NPL, BN, CHAIN, OVL, WPS>;
    if (hipError_t e = sx_allow_big_lds(reinterpret_cast<const void *>(kern), attr_done); e != hipSuccess) return e;
    if (g_launch_name_on)
        snprintf(g_launch_name, sizeof g_launch_name, "conv_sx_pair16_kernel<%d, %d, %d, %s, %s, %d>", C, NPL, BN, CHAIN ? "true" : "false",
                 OVL ? "true" : "false", WPS);
    kern<<<grid, 256, lds, stream>>>(a);
    return hipGetLastError();
}

// Tile choice per (channels, arithmetic): the widest tile that still leaves >= 3 workgroups per CU (LDS) - measured choices
// are recorded in DESIGN.md 5.1e; VITSMI_PAIR16_BN forces 128 / 256 for A/B runs.
int sx_pair16_plan(int C, int npl, int K1, int dil1, int K2, int dil2, bool *ovl) {
    static const int force = [] {
        const char *e = std::getenv("VITSMI_PAIR16_BN");
        return e ? std::atoi(e) : 0;
    }();
    if (!((C == 32 || C == 64) && (npl == 1 || npl == 2))) return 0;
    // f16x3 at 64 channels: x + Y side by side would leave one workgroup per CU; Y overlays the x tile and the residual waits
    // in registers.  (At 32 channels the overlay would pay only with a third workgroup per CU, i.e. <= 168 registers: the
    // 256-column variant then spills 34 - refused, see the build's spill rule.)
    const bool o = npl == 2 && C == 64;
    if (ovl) *ovl = o;
    // (measured r04e: the wider tile wins in every shape: less halo, fewer fixed costs per column.  At 64 channels only the
    // 128-column tile exists: the 256-column one needs more than 256 registers, and a kernel whose operands arrive through
    // asynchronous inline-asm loads must NEVER spill - the compiler would save a register the load has not yet written)
    int pref[2] = {256, 128};
    if (C == 64) pref[0] = pref[1] = 128;
    if ((force == 128 || force == 256) && C == 32) {
        pref[0] = force;
        pref[1] = force == 128 ? 256 : 128;
    }
    for (int i = 0; i < 2; i++) {
        SxPair16Geom g;
        if (sx_pair16_geom(C, npl, pref[i], o, K1, dil1, K2, dil2, &g) && g.lds <= (size_t)80 * 1024 - 256) return pref[i];
    }
    return 0;
}

hipError_t launch_conv_sx_pair16(SxPair16Args a, int C, int npl, int B, hipStream_t stream, bool chain) {
    bool ovl = false;
    const int BN = sx_pair16_plan(C, npl, a.K1, a.dil1, a.K2, a.dil2, &ovl);
    SxPair16Geom g;
    if (!BN || !sx_pair16_geom(C, npl, BN, ovl, a.K1, a.dil1, a.K2, a.dil2, &g)) return hipErrorInvalidValue;
    if (a.pad1 * 2 != (a.K1 - 1) * a.dil1 || a.pad2 * 2 != (a.K2 - 1) * a.dil2) return hipErrorInvalidValue;  // "same" padding
    a.LW1 = g.LW1;
    a.RS1 = g.RS1;
    a.RS2 = g.RS2;
    a.BNo = g.BNo;
    a.y_off = g.y_off;
    a.b_off = g.b_off;
    a.magic1 = (unsigned)((0x100000000ull + a.RS1 - 1) / a.RS1);
    a.NT = (a.T + a.BNo - 1) / a.BNo;
    a.B = B;
    if (a.islope == 0.f) a.islope = 1.f;
    if (a.mslope == 0.f) a.mslope = 1.f;
    if (a.oslope == 0.f) a.oslope = 1.f;
    if (a.wscale1 == 0.f) a.wscale1 = 1.f;
    if (a.wscale2 == 0.f) a.wscale2 = 1.f;
    a.un_islope = 1.f / a.islope;
    if ((long long)a.T * 64 + 64 >= (1ll << 32)) return hipErrorInvalidValue;
    if (!a.xpl) return hipErrorInvalidValue;
    if ((a.flags & (EPI_ACC | P16_HAS_RAW)) && !a.out_raw) return hipErrorInvalidValue;
    if ((a.flags & P16_HAS_PL) && !a.out_pl) return hipErrorInvalidValue;
    if ((a.flags & EPI_DIV) && !(a.flags & EPI_ACC)) return hipErrorInvalidValue;
    const long long nb = (long long)a.NT * B;
    if (nb == 0) return hipSuccess;
    if (nb >= (1ll << 31)) return hipErrorInvalidValue;
    dim3 grid((unsigned)nb, 1, 1);
#define P16_CASE(CC, NP, BNN, OV, W)                                                                              \
    if (C == CC && npl == NP && BN == BNN && ovl == OV)                                                           \
        return chain ? launch_conv_sx_pair16_k<CC, NP, BNN, true, OV, W>(a, grid, g.lds, stream)                  \
                     : launch_conv_sx_pair16_k<CC, NP, BNN, false, OV, W>(a, grid, g.lds, stream);
    P16_CASE(32, 1, 256, false, 3)
    P16_CASE(32, 1, 128, false, 4)
    P16_CASE(64, 1, 128, false, 3)
    P16_CASE(32, 2, 128, false, 3)
    P16_CASE(32, 2, 256, false, 2)
    P16_CASE(64, 2, 128, true, 2)
#undef P16_CASE
    return hipErrorInvalidValue;
}
#endif  // VITSMI_IMPL_PAIR16

}  // namespace vitsmi
