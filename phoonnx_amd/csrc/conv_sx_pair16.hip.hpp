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

enum : int { P16_HAS_RAW = 1 << 9, P16_HAS_PL = 1 << 10,  // (EPI_ACC / EPI_DIV as everywhere)
             P16_NO_OVERLAP = 1 << 11 };  // (A/B, VITSMI_P16_DEBUG=nooverlap) persistent form: the next x tile requested behind phase 2, not under it
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
// PERSIST (round 5): a workgroup renders tiles blockIdx.x, blockIdx.x + gridDim.x, .. (the grid is WPS workgroups per CU) and
// brings tile i + 1's x tile into LDS WHILE tile i's second conv runs: the DMA rounds of the next tile are dealt to phase 2's
// steps, right behind each step's weight request, with exact vector-memory accounting in the counted waits (vector-memory
// operations retire in order: behind the weights of step s sit min(left, D - 1) younger weight sets and the DMA rounds issued
// since - per wave: a wave issues a round only if its 64 cells lie inside the tile).  The x buffer is free by then: Y sits
// beside it (no overlay) and the residual waits in registers from the hand-over on.  What a tile of the one-shot form spends
// waiting for its x tile (4.5-5 k cycles of 16-32 k) passes under the MFMAs of its predecessor.
template <int C, int NPL, int BN, bool CHAIN, bool OVL, int WPS, bool PERSIST = false>
__global__ __launch_bounds__(256, WPS) void conv_sx_pair16_kernel(SxPair16Args a) {
    static_assert(!(PERSIST && OVL), "persistent form: Y beside x (x is refilled under phase 2)");
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
    const int tid = threadIdx.x;
    int lane = tid & 63;  // (PERSIST: re-derived at the top of every tile behind an opaque asm, see the tile loop)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int l31 = lane & 31, hi = lane >> 5;
    const int T = a.T, LW1 = a.LW1, RS1 = a.RS1, RS2 = a.RS2;
    const int NTOT = a.NT * a.B, G = (int)gridDim.x;
    // (utterance, first kept output column, the utterance's tensor end (SxRagged; T = row pitch)) of tile tn
    auto tile_geom = [&](int tn, int &gb, int &gt0, int &gTV) __attribute__((always_inline)) {
        gb = tn / a.NT;
        gt0 = (tn - gb * a.NT) * a.BNo;
        gTV = __builtin_amdgcn_readfirstlane(sx_valid_cols(a.rag, gb, T));
    };
    // the first tile tn, tn + G, .. with a kept column inside its utterance (NTOT: none)
    auto next_valid = [&](int tn, int &gb, int &gt0, int &gTV) __attribute__((always_inline)) {
        while (tn < NTOT) {
            tile_geom(tn, gb, gt0, gTV);
            if (gt0 < gTV) break;
            tn += G;
        }
        return tn;
    };
    int tile_nb = (int)blockIdx.x, b = 0, t0 = 0, TV = 0;  // the current tile
    if constexpr (PERSIST) {
        tile_nb = next_valid(tile_nb, b, t0, TV);
        if (tile_nb >= NTOT) return;                       // (uniform exit)
    } else {
        if (tile_nb >= NTOT) return;
        tile_geom(tile_nb, b, t0, TV);
        if (t0 >= TV) return;                              // (uniform exit) no kept column lies inside the utterance
    }
    int t1 = t0 - a.pad2;                                  // first column phase 1 computes
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
    const uint16_t *const k_xpl = a.xpl;
    const int64_t k_x_bs = a.x_bstride;
    const unsigned k_magic1 = a.magic1;
    const int k_pad1 = a.pad1;
    const float *const k_zeros = a.zeros;
    asm volatile("" ::"s"(k_xpl), "s"(k_x_bs), "s"(k_magic1), "s"(k_pad1), "s"(k_zeros));

    struct ASet {
        u32x4 f[2][NPL];
    };
    auto load_a = [&](ASet &f, const char *wb, int step) __attribute__((always_inline)) {
        const uint32_t voff0 = (uint32_t)lane * 16u;  // (from the tile's own lane id: see the tile loop)
        const uint64_t pa = reinterpret_cast<uint64_t>(wb) + (uint64_t)((int64_t)step * STEPBYTES);
        const char *sb = reinterpret_cast<const char *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(pa >> 32)) << 32) |
                                                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)pa));
        // (one statement per set, behind the wait states its scalar base needs: this kernel's tile loop keeps scalars in VGPR
        // lanes, and a v_readlane directly in front of an inline-asm load is a hazard the compiler does not see)
        if constexpr (H1) global_read128_x2(voff0, sb, f.f[0][0], f.f[1][0]);
        else global_read128_x4(voff0, sb, f.f[0][0], f.f[0][1], f.f[1][0], f.f[1][1]);
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
    // one DMA round = 256 cells (16 bytes per lane) of the flattened [plane][row][RS1] cell space of the x tile of utterance
    // gb whose phase-1 columns start at gt1; a wave issues it iff its 64 cells lie inside the tile (ncell % 64 == 0)
    const int ncell = NPL * CG * RS1, nit = (ncell + 255) >> 8;
    const int my_rounds = (ncell - wave * 64 + 255) >> 8;  // rounds THIS wave takes part in (wave-uniform)
    auto dma_round = [&](int it, int gb, int gt1, int gTV) __attribute__((always_inline)) {
        const uint16_t *xb = k_xpl + (int64_t)gb * k_x_bs;
        const int base = it * 256 + wave * 64;
        const int i = base + lane;
        const int row = (int)__umulhi((unsigned)i, k_magic1);
        const int col = i - row * RS1;
        const int t = gt1 - k_pad1 + col;
        const bool ok = row < NPL * CG && col < LW1 && t >= 0 && t < gTV;
        const void *src = ok ? static_cast<const void *>(xb + ((int64_t)row * T + t) * 8)
                             : static_cast<const void *>(reinterpret_cast<const char *>(k_zeros) + lane * 16);
        // (a wave whose cells lie past the tile's last one skips the round: nothing is written behind the allocation)
        if (base < ncell) lds_dma<16>(src, reinterpret_cast<float *>(lds_sx + (size_t)base * 16));
    };
    {
        // LDS-DMA, 16 bytes per lane: the flattened [plane][row][RS1] cell space in rounds of 256 cells; a lane whose cell is
        // padding (column past LW1, row past the tile, time outside the tensor) reads the zero page.  (Plane p, group g of
        // the tensor is row p * C/8 + g of [NPL][C/8][T] cells: the same index as in the tile.)
        for (int it = 0; it < nit; it++) dma_round(it, b, t1, TV);
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
    // (PERSIST, phase 2) at most n vector-memory operations in flight, n = the weight loads AND the next tile's DMA rounds
    // this wave has issued behind the set it waits for; n <= (D - 1) NAL + D * rounds per step (<= 16 for every shape that
    // fits: a larger count waits for more than it must, which is safe)
    auto wait_vm_n = [&](int n) __attribute__((always_inline)) {
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
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
            case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
            case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        }
    };
    // the NEXT tile of this workgroup (PERSIST): set at the top of a tile, its x tile is requested during phase 2
    int nx_tile = NTOT, nx_b = 0, nx_t0 = 0, nx_TV = 0;
    int dma_R = 0;  // DMA rounds per step of phase 2 (0: no next tile)
    const std::integral_constant<int, 0> I0{};
    // rows0: this lane's cell of (plane 0, chunk 0, tap 0, unit 0): row = lane >> 4, column = the wave's first + (lane & 15).
    // The first D weight sets must have been requested (prefetch_a).
    auto run_conv = [&](auto WITH_DMA, const char *wb, int K, int dil, uint32_t rows0, uint32_t row_bytes, uint32_t pstride) __attribute__((always_inline)) {
        constexpr bool dma = decltype(WITH_DMA)::value;
        const int S = NCH * K;
        int chunk = 0, tap = 0;
        load_b(I0, I0, rows0, pstride);
        // DMA rounds this wave has issued before step s of this conv
        auto dsum = [&](int s) __attribute__((always_inline)) { const int v = s * dma_R; return v < my_rounds ? v : my_rounds; };
        auto step = [&](ASet &fc, ASet &fload, int s) __attribute__((always_inline)) {
            const int left = S - 1 - s;
            const int younger = left < D - 1 ? left : D - 1;
            if constexpr (dma) {
                // behind A(s) (requested first thing in step s - D, or by the prefetch): the younger weight sets and every
                // DMA round of the steps since
                const int s0 = s > D ? s - D : 0;
                wait_vm_n(younger * NAL + (dma_R ? dsum(s) - dsum(s0) : 0));
            } else
                wait_a(younger);
            __builtin_amdgcn_sched_barrier(0);
            if (s + D < S) load_a(fload, wb, s + D);
            if constexpr (dma) {
                if (dma_R) {
                    for (int r = s * dma_R; r < (s + 1) * dma_R; r++)
                        if (r < nit) dma_round(r, nx_b, nx_t0 - k_pad2, nx_TV);
                }
            }
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

#if P16_PROF
    unsigned long long tsum[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned ntiles_done = 0;
#endif
    for (;;) {  // (PERSIST: one iteration per tile of this workgroup; else exactly one)
    if constexpr (PERSIST) {
        // Everything a tile derives from the lane id (LDS addresses of the fragments, of Y, of the residual, the epilogue's
        // cell offsets) is loop-invariant: hipcc hoists it all out of the tile loop and keeps it live across the whole body -
        // tens of registers, spills in every instantiation.  Re-deriving the lane id behind an opaque statement makes each tile
        // compute what it needs where it needs it, as the one-shot form does.
        lane = tid & 63;
        asm volatile("" : "+v"(lane));
        l31 = lane & 31;
        hi = lane >> 5;
        // the tile after this one: found NOW - the scalar loads of the search (SxRagged::len) must not be in flight inside the
        // phases, whose counted lgkmcnt waits assume LDS reads only - and requested during phase 2
        nx_tile = next_valid(tile_nb + G, nx_b, nx_t0, nx_TV);
        dma_R = nx_tile < NTOT && !(k_flags & P16_NO_OVERLAP) ? (nit + S2 - 1) / S2 : 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    // =================================================================== phase 1: c1 over columns [t1, t1 + BN)
    zero_acc();
    run_conv(std::false_type{}, wbase1, a.K1, a.dil1, lds0 + (uint32_t)((lane >> 4) * RS1 + wn * BNW + (lane & 15)) * 16u, (uint32_t)RS1 * 16u, XPB);
    P16_STAMP(3);
    prefetch_a(wbase2, S2);  // c2's first weights travel during the hand-over
    gather_acc();

    // =================================================================== hand-over: c1's output -> Y (operand planes in LDS)
    // pre: the residual in the accumulator layout - CHAIN: x1 = c1(..) + x (phase 2's residual); PAIR with OVL: x, saved
    // before Y overwrites the tile.  (PAIR without OVL reads x from LDS in the epilogue: no registers.)
    constexpr bool KEEP = CHAIN || OVL || PERSIST;  // (PERSIST: the x buffer is refilled under phase 2)
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
    // (PERSIST: every wave is behind the barrier above, i.e. done with the x tile - phase 1 and the residual read - so the
    // next tile's DMA rounds may land in it while c2 runs over Y)
    run_conv(std::integral_constant<bool, PERSIST>{}, wbase2, k_K2, k_dil2,
             ylds + (uint32_t)((lane >> 4) * RS2 + wn * BNW + (lane & 15)) * 16u, (uint32_t)RS2 * 16u, YPB);
    P16_STAMP(5);
    if constexpr (PERSIST) {
        if (nx_tile < NTOT) {
            // (with the overlap on, S2 * dma_R >= nit: nothing is left; off (debug): the whole tile is requested here)
            for (int r = S2 * dma_R; r < nit; r++) dma_round(r, nx_b, nx_t0 - k_pad2, nx_TV);
            prefetch_a(wbase1, S1);  // the next tile's first weights travel under this tile's epilogue
        }
    }
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
#if P16_PROF
    P16_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P16_STAMP(7);
    for (int i = 0; i < 7; i++) tsum[i] += tp[i + 1] - tp[i];
    ntiles_done++;
#endif
    if constexpr (!PERSIST) break;
    else {
        if (nx_tile >= NTOT) break;
        tile_nb = nx_tile;
        b = nx_b;
        t0 = nx_t0;
        TV = nx_TV;
        t1 = t0 - k_pad2;
        // the x tile requested under phase 2, the weights requested before the epilogue and the epilogue's own stores: all behind
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        P16_STAMP(0);
        P16_STAMP(1);
        __builtin_amdgcn_s_barrier();  // the x tile is complete; every wave has left phase 2 (Y may be rewritten)
        __builtin_amdgcn_sched_barrier(0);
        P16_STAMP(2);
    }
    }  // tiles
#undef ACC
#if P16_PROF
    if (a.prof && tid == 0) {  // (a row per workgroup, plain stores: atomics on eight shared words distorted what they measured)
        // phase sums over this workgroup's tiles (one tile in the one-shot form); [7] = tiles
        for (int i = 0; i < 7; i++) a.prof[(size_t)blockIdx.x * 8 + i] = tsum[i];
        a.prof[(size_t)blockIdx.x * 8 + 7] = ntiles_done;
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
template <int C, int NPL, int BN, bool CHAIN, bool OVL, int WPS, bool PERSIST = false>
inline hipError_t launch_conv_sx_pair16_k(const SxPair16Args &a, dim3 grid, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv_sx_pair16_kernel<C, NPL, BN, CHAIN, OVL, WPS, PERSIST>;
    if (hipError_t e = sx_allow_big_lds(reinterpret_cast<const void *>(kern), attr_done); e != hipSuccess) return e;
    if (g_launch_name_on)
        snprintf(g_launch_name, sizeof g_launch_name, "conv_sx_pair16_kernel<%d, %d, %d, %s, %s, %d, %s>", C, NPL, BN, CHAIN ? "true" : "false",
                 OVL ? "true" : "false", WPS, PERSIST ? "true" : "false");
    if constexpr (PERSIST) {
        // WPS workgroups per CU, each walking the tiles blockIdx.x, blockIdx.x + gridDim.x, ..
        static const int cus = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            return n;
        }();
        const unsigned g = (unsigned)(WPS * cus);
        if (grid.x > g) grid.x = g;
    }
    kern<<<grid, 256, lds, stream>>>(a);
    return hipGetLastError();
}

// Tile choice per (channels, arithmetic): the widest tile that still leaves >= 3 workgroups per CU (LDS) - measured choices
// are recorded in DESIGN.md 5.1e; VITSMI_PAIR16_BN forces 128 / 256 for A/B runs.
// The persistent form (Y beside x at the same tile width must fit the same 80 KiB).  OPT-IN, VITSMI_PAIR16_PERSIST=1: built,
// parity-green on multi-tile shapes, measured SLOWER (r05l / r05m, B = 32, ms per launch, persistent vs one-shot: f16x3 32 ch
// k 3 0.670 vs 0.608, k 11 1.297 vs 1.206, 64 ch k 3 0.945 vs 0.875; f16 32 ch k 3 0.455 vs 0.344).  Per-tile stamps
// (profiles/r05_runs/pair16_stamps_r05m.log, 32 ch f16x3 k 3): the wait for the x tile falls from 4 935 to 189 cycles - the HBM
// round trip IS hidden - but phase 2 grows from 2 669 to 6 587 (k 11: 8 505 -> 16 826): vector-memory operations retire in
// order per wave, so every weight set requested behind a DMA round waits out that round's HBM latency (2-5 k cycles against
// a step of ~800), and the rounds' own issue costs the MFMA loop ~150 cycles each; phase 1, the hand-over and the epilogue
// run ~35 % longer (the tile loop's scalar state lives in VGPR lanes).  The measurement confirms what DESIGN 5.1f inferred:
// the next tile's loads need a wave of their OWN (its own vmcnt), which costs the fifth wave's registers.
bool sx_pair16_persist_ok(int C, int npl, int BN, int K1, int dil1, int K2, int dil2) {
    static const bool on = [] {
        const char *e = std::getenv("VITSMI_PAIR16_PERSIST");
        return e && e[0] == '1';
    }();
    SxPair16Geom g;
    return on && sx_pair16_geom(C, npl, BN, false, K1, dil1, K2, dil2, &g) && g.lds <= (size_t)80 * 1024 - 256;
}
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
    // the persistent form where x and Y fit side by side (it cannot overlay: the x buffer is refilled under phase 2)
    const bool persist = BN && sx_pair16_persist_ok(C, npl, BN, a.K1, a.dil1, a.K2, a.dil2);
    if (persist) ovl = false;
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
    {
        static const bool no_overlap = [] {
            const char *e = std::getenv("VITSMI_P16_DEBUG");
            return e && std::string(e) == "nooverlap";
        }();
        if (no_overlap) a.flags |= P16_NO_OVERLAP;
    }
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
#define P16_PCASE(CC, NP, BNN, W)                                                                                \
    if (persist && C == CC && npl == NP && BN == BNN)                                                             \
        return chain ? launch_conv_sx_pair16_k<CC, NP, BNN, true, false, W, true>(a, grid, g.lds, stream)         \
                     : launch_conv_sx_pair16_k<CC, NP, BNN, false, false, W, true>(a, grid, g.lds, stream);
    // (two workgroups per CU throughout: with the residual in registers and the tile loop's state the single-plane variants
    // do not fit the 168 registers of three - and a kernel of this family must not spill)
    P16_PCASE(32, 1, 256, 2)
    P16_PCASE(32, 1, 128, 2)
    P16_PCASE(64, 1, 128, 2)
    P16_PCASE(32, 2, 128, 2)
    P16_PCASE(32, 2, 256, 2)
    P16_PCASE(64, 2, 128, 2)
#undef P16_PCASE
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
