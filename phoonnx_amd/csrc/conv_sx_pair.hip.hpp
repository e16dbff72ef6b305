// conv_sx_pair.hip.hpp — two dependent convs of a ResBlock in ONE launch, on the raw-format stages of the generator
// (<= 64 channels):
//   PAIR   a ResBlock1 step    out = c2(lrelu(c1(lrelu(x)))) + x                 (phoonnx_train/vits/modules.py:301-314)
//   CHAIN  two ResBlock2 steps x1 = c1(lrelu(x)) + x ;  out = c2(lrelu(x1)) + x1  (modules.py:355-364)
//
// Why: these layers are HBM-bound (8-20 FLOP/B, 4-5 TB/s measured).  As two launches of conv_sx_kernel a step moves
// 20 bytes per element: c1 reads x and writes the intermediate, c2 reads the intermediate AND x (the residual) and
// writes the result.  Here the intermediate never leaves the CU - it is produced as the fp16 operand planes c2 reads,
// directly into LDS - and x is read once (the residual is requested next to the x tile, cf. SX_RES_EARLY):
// ~4 * (1 + halo/256) + 4 bytes per element, 2.1-2.3x less.  The matrix work is the same (+ the 2-10 overlap columns
// of a 256-column tile), so the matrix pipe is about twice as busy as in the two-launch form.
//
// Arithmetic: exactly that of two conv_sx_kernel launches in the f16x3 mode (same operand planes, same products, same
// accumulation order); only the tile boundaries move, which changes no sum.  tests/test_gpu_parity.py compares the
// fused launch with the two-launch form bit for bit and with the oracle.
//
// Structure (one workgroup = 4 waves, all C output channels, 256 columns):
//   prologue the whole x tile (all channels, 256 + halo columns) global -> registers -> lrelu -> fp16 split -> LDS, one
//            HBM round trip;
//   phase 1  c1 over columns [t0 - P2, t0 - P2 + 256) from that tile: weights global -> registers one step ahead, B
//            fragments from LDS, no DMA and no barriers;
//   hand-over  accumulators * 2^-k1 + bias1 -> lrelu -> two fp16 planes -> LDS array Y[chunk][plane][half][col]
//            (zero where the column lies outside the tensor: c2's zero padding); Y overlays the x stages, hence one
//            barrier before and one after;
//   phase 2  c2 over Y: weights global -> registers, B fragments from Y, no DMA and no barriers.  Y is stored P2 columns
//            to the right, so that tap k of output column j reads Y[j + k * dil2]: phase 2 then produces column j in the
//            lane that produced column j in phase 1 (its c1 accumulators are the CHAIN residual x1, no exchange);
//   epilogue bias2, residual (x from registers; CHAIN: x1), multi-receptive-field accumulate / divide, fp32 raw store
//            of the 256 - 2 P2 interior columns (P2 = c2's one-sided reach; columns nearer the tile edge saw garbage).
#pragma once
#include <cstdlib>
#include "conv_sx_engine.hip.hpp"

namespace vitsmi {

struct SxPairArgs {
    const float *xr;          // fp32 raw input [B][C/8][T][8]; also the residual
    float islope, mslope;     // leaky-ReLU slopes: on x, and between the convs
    int T;
    const u32x4 *wp1, *wp2;   // packed weights (pack_conv_sx, two fp16 planes, this tile config)
    const float *bias1, *bias2;
    float wscale1, wscale2;
    float *out_raw;           // fp32 raw output [B][C/8][T][8]
    const float *zeros;
    int C, nchunks;           // channels (= Cin = Cout of both convs), C / 16
    int K1, dil1, pad1;       // c1
    int K2, dil2, pad2;       // c2
    int LW1;                  // x tile width in cells = 256 + (K1 - 1) * dil1
    unsigned x_bytes;         // one chunk's x stage (2 planes x 2 halves x LW1 cells)
    int LW2;                  // Y row width in cells (256 + pad2, rounded up: see launch_conv_sx_pair)
    unsigned y_chunk_bytes;   // bytes of one chunk of Y = 4 * LW2 * 16
    int BNo, NT, B;           // kept output columns per tile, tiles along time, utterances
    int flags;                // EPI_ACC | EPI_DIV (the residual is always added)
    float div;
    unsigned *peak;           // f16 range guard slots (SxArgs::peak), may be nullptr
    // NCH > 1 (multi-receptive-field fusion, see conv_sx_pair_kernel): the chains after the first.  Chain 0 is described
    // by the fields above; K1 / dil1 / pad1 and pad2 above are then the LARGEST reaches (the tile geometry), and xoff /
    // yoff are what a chain with a shorter reach adds to its operand columns (pad1 - its own pad1, pad2 - its own pad2).
    struct Chain {
        const u32x4 *wp1, *wp2;
        const float *bias1, *bias2;
        float wscale1, wscale2;
        int K1, dil1, K2, dil2;
        int xoff, yoff;
    } ch[3];
    int nchain;
    unsigned a_ring;          // (32-channel variant) byte offset of the shared weight ring in LDS: 3 groups x 2 steps x 2 KiB
    unsigned bias_off;        // (one chain) byte offset of the two bias vectors in LDS (one 1 KiB DMA slot behind the tile)
    unsigned long long *prof;  // (SX_PAIR_PROF builds) 8 counters of this launch: six phase sums in shader cycles, -, workgroups
    SxRagged rag;             // per-utterance tensor ends of a padded batch (conv_sx_engine.hip.hpp)
    int xgs = 0;              // XCD grouping of the time tiles (SxArgs::xgs)
};

// (the 32-channel variant needs ~165 registers and <= 40 KiB of LDS: three workgroups per CU hide more of each
// other's load / hand-over / store phases than two; the 64-channel one holds 64 accumulators + 64 residual registers)
#ifndef SX_PAIR_PROF
#define SX_PAIR_PROF 0  // diagnostic build: s_memtime stamps at the phase boundaries, summed per launch into SxPairArgs::prof
#endif
#ifndef SX_PAIR_SHARED_A
#define SX_PAIR_SHARED_A 0  // 32-channel variant: weights through a workgroup-shared LDS ring (run_conv_sha): built, bit-identical,
                           // measured neutral (k3 / k5 / k7 chains 540 / 763 / 1129 -> 539 / 753 / 1198 us): off
#endif
#ifndef SX_PAIR_ALIAS
#define SX_PAIR_ALIAS 0
#endif
#ifndef SX_PAIR_DEPTH32
#define SX_PAIR_DEPTH32 2
#endif
#ifndef SX_PAIR_DEPTH64
#define SX_PAIR_DEPTH64 2
#endif
#ifndef SX_PAIR_EARLY_ACC
#define SX_PAIR_EARLY_ACC 0
#endif
#ifndef SX_PAIR_WGS32
#define SX_PAIR_WGS32 3  // workgroups per CU the 32-channel variant is compiled for (4 = 128 registers: 116-172 bytes of scratch
                         // per lane, the three chains of the default voice 551 / 767 / 1125 -> 688 / 1064 / 1336 us)
#endif
#ifndef SX_PAIR_EARLY32
#define SX_PAIR_EARLY32 1
#endif
#ifndef SX_PAIR_ST_SC1
#define SX_PAIR_ST_SC1 0
#endif
// NCH > 1: the multi-receptive-field sum of a ResBlock2 stage, xs = (rb_0(x) + rb_1(x) + ..) / n (models.py:356-363), as ONE
// launch: the x tile (widest halo of the chains) is loaded and split ONCE and stays in LDS (Y no longer overlays it), the
// chains run one after the other over the same 256 columns - a chain with a shorter reach reads its operands xoff / yoff
// cells further right - and their results meet in registers, in the order the separate launches added them (bit-identical).
// As three launches the stage's tensor is read 3 + 2 times (x three times, the running sum twice) and written three
// times; here once each.
template <int MW, int NW, int WM, int WN, int EPI, bool CHAIN, int NCH = 1>
__global__ __launch_bounds__(256, (NW == 2 && !SX_PAIR_EARLY_ACC && NCH == 1) ? SX_PAIR_WGS32 : 2) void conv_sx_pair_kernel(SxPairArgs a) {
    constexpr int BN = NW * WN * 32, NH = NW / 2;
    static_assert(NCH == 1 || (CHAIN && NW == 2 && (EPI & EPI_ACC) == 0), "fused chains: 32 channels, no external running sum");
    // 32-channel variant: the residual is requested in the prologue, right behind the x tile, and waits in registers
    // (its lines are in flight at that moment; after phase 1 they have left the L2: PMC showed the re-read going to
    // the fabric).  The running sum (EPI_ACC) is NOT: with it the variant needs 188 registers, loses the third
    // workgroup per CU and runs 12-16 % slower (profiles r02_v5 -> r02_v6).
    constexpr bool EARLY = NW == 2 && SX_PAIR_EARLY32;
    constexpr int DEPTH = NW == 2 ? SX_PAIR_DEPTH32 : SX_PAIR_DEPTH64;  // weight look-ahead in steps (run_conv)
    static_assert(DEPTH >= 2 && DEPTH <= 6, "wait_a covers up to five younger sets");
    constexpr bool ACC = (EPI & EPI_ACC) != 0 && SX_PAIR_EARLY_ACC;
    static_assert(WM * WN == 4 && MW == 1 && BN == 256, "one block row per wave, 256 columns");
    constexpr int NPW = 2, STEPBYTES = WM * MW * NPW * 1024;
    constexpr int MAXCH = WM * 2;  // 16-channel chunks: C = 32 * WM
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_sx[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wg_xcd = blockIdx.x & 7, wg_seq = blockIdx.x >> 3;
    const int tile_nb = __builtin_amdgcn_readfirstlane(sx_xcd_tile(wg_seq, wg_xcd, a.xgs));
    if (tile_nb >= a.NT * a.B) return;
    const int b = tile_nb / a.NT, t0 = (tile_nb - b * a.NT) * a.BNo;  // first kept output column
    const int t1 = t0 - a.pad2;                                        // first column phase 1 computes
    const int T = a.T, LW = a.LW1;
    const int TV = __builtin_amdgcn_readfirstlane(sx_valid_cols(a.rag, b, T));  // this utterance's tensor end (SxRagged); T = row pitch
    if (t0 >= TV) return;                                                        // (uniform exit) no kept column lies inside it
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_sx;
    const uint32_t XB = a.x_bytes;
    const char *wbase1 = reinterpret_cast<const char *>(a.wp1) + wm * (MW * NPW * 1024);
    const char *wbase2 = reinterpret_cast<const char *>(a.wp2) + wm * (MW * NPW * 1024);
    const float *xrb = a.xr + (int64_t)b * a.C * T;
    float pk = 0.f;

    struct ASet {
        u32x4 fa[2];
    };
    const uint32_t voff0 = (uint32_t)lane * 16u;
    auto load_a = [&](ASet &f, const char *wb, int step) {
        const char *sb = wb + (int64_t)step * STEPBYTES;
        f.fa[0] = global_read128<0>(voff0, sb);
        f.fa[1] = global_read128<1024>(voff0, sb);
    };

    // Residual operands of this wave (x itself at the columns this lane produces: t1 + col) and, with EPI_ACC, the
    // running sum of the resblocks at the same places, in the accumulator layout.
    f32x4 pre[NW / 2][2][4], ad[EARLY && ACC ? NW / 2 : 1][2][4];
    auto load_res = [&](const float *base, f32x4 (*dst)[2][4]) {
        static_for<NW / 2>([&](auto R) {
            constexpr int rr = decltype(R)::value;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int row0 = wm * 32;
                const int t = t1 + (wn * NW + rr * 2 + j) * 32 + l31;
                const int tl = t < 0 ? 0 : (t < T ? t : T - 1);  // (clamped columns are never kept)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float *p = base + ((int64_t)((row0 >> 3) + q) * T + tl) * 8 + 4 * hi;
                    if constexpr (EARLY) dst[rr][j][q] = __builtin_bit_cast(f32x4, global_read128_v<0>(p));
                    else dst[rr][j][q] = *reinterpret_cast<const f32x4 *>(p);
                }
            }
        });
    };
    auto load_pre = [&]() { load_res(xrb, pre); };

    // ---- prologue: the WHOLE x tile (every 16-channel chunk) is requested at once, converted (leaky-ReLU, fp16 split)
    // and written to its LDS stage: one HBM round trip per tile.  (A first version fetched chunk c + 1 during chunk c,
    // like conv_sx_kernel: with K = 3 a chunk is three steps of matrix work, an HBM round trip is ten, and - vector
    // loads return in order - every weight load behind an x load waits for it: ~2.5 us of stall per chunk, 25 of the
    // 31 us a 64-channel k = 3 tile took.)  A thread owns cells i = it*256 + tid of a chunk's [2 halves][LW] cells.
#if SX_PAIR_PROF
    auto stamp = [&]() {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    unsigned long long tp[8];
    tp[0] = stamp();
#endif
    if constexpr (NW == 2 && WM == 1 && NCH == 1 && SX_PAIR_SHARED_A != 0)
        // (shared weight ring, see run_conv_sha: c1's first groups are requested ahead of the x tile)
        for (int g = 0; g < 3; g++)
            if (g < ((a.nchunks * a.K1) >> 1)) {
                const int sub = wave >> 1, pl = wave & 1;
                lds_dma<16>(wbase1 + (int64_t)(2 * g + sub) * STEPBYTES + pl * 1024 + lane * 16,
                            reinterpret_cast<float *>(lds_sx + a.a_ring + (unsigned)((g * 2 + sub) * 2048 + pl * 1024)));
            }
    // (round 4) both bias vectors -> LDS by one DMA of wave 0, ahead of the x tile: as dependent global loads at the hand-over
    // and in the epilogue each exposed an L2 round trip per tile (conv_sx_pair16's stamps: ~1.5 k cycles, twice, of ~29 k)
    constexpr bool BIAS_LDS = NCH == 1;
    if constexpr (BIAS_LDS) {
        if (wave == 0) {
            constexpr int CC = WM * 32;
            const int k = lane - (lane >= CC / 4 ? CC / 4 : 0);
            const float *bp = lane < CC / 4 ? a.bias1 : a.bias2;
            const void *src = (bp && lane < CC / 2) ? static_cast<const void *>(bp + 4 * k) : static_cast<const void *>(a.zeros + 4 * (lane & 31));
            lds_dma<16>(src, reinterpret_cast<float *>(lds_sx + a.bias_off));
        }
    }
    auto bias_from_lds = [&](int conv, f32x4 (&bq)[4]) __attribute__((always_inline)) {
        u32x4 t4[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            asm volatile("ds_read_b128 %0, %1" : "=v"(t4[q]) : "v"(lds0 + a.bias_off + (uint32_t)(conv * (WM * 32) + wm * 32 + 8 * q + 4 * hi) * 4u) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);  // (nothing else ties the consumers to the wait)
#pragma unroll
        for (int q = 0; q < 4; q++) bq[q] = __builtin_bit_cast(f32x4, t4[q]);
    };
    constexpr int NXC = 3;
    {
        u32x4 xst[MAXCH][NXC][2];
        const int nxc = (2 * LW + 255) >> 8;
        uint32_t xroff[NXC];
        bool xrok[NXC];
        int xkh[NXC], xcol[NXC];
#pragma unroll
        for (int it = 0; it < NXC; it++) {
            const int i = it * 256 + tid;
            const int kh = (i >= LW ? 1 : 0) + (i >= 2 * LW ? 1 : 0);
            const int col = i - kh * LW;
            const int t = t1 - a.pad1 + col;
            xkh[it] = kh;
            xcol[it] = col;
            xrok[it] = it < nxc && kh < 2 && t >= 0 && t < TV;
            xroff[it] = xrok[it] ? (uint32_t)(((int64_t)kh * T + t) * 32) : 0u;
        }
        static_for<MAXCH>([&](auto CH) {
            constexpr int ch = decltype(CH)::value;
            const float *cbase = xrb + (int64_t)(2 * ch) * T * 8;
            static_for<NXC>([&](auto I) {
                constexpr int it = decltype(I)::value;
                if (it < nxc) {
                    xst[ch][it][0] = global_read128_x<0>(xroff[it], cbase);
                    xst[ch][it][1] = global_read128_x<16>(xroff[it], cbase);
                }
            });
        });
        if constexpr (EARLY) {  // (vector loads return in order: the x tile has landed when only these are in flight)
            load_pre();
            if constexpr (ACC) {
                load_res(a.out_raw + (int64_t)b * a.C * T, ad);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            } else
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if SX_PAIR_PROF
        tp[1] = stamp();  // the x tile has arrived
#endif
        const float isl = a.islope;
        static_for<MAXCH>([&](auto CH) {
            constexpr int ch = decltype(CH)::value;
            static_for<NXC>([&](auto I) {
                constexpr int it = decltype(I)::value;
                if (it < nxc && xkh[it] < 2) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float x = xrok[it] ? __uint_as_float(xst[ch][it][e >> 2][e & 3]) : 0.f;
                        v[e] = fmaxf(x, x * isl);
                    }
                    unsigned p0[4], p1[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) split2h_pair_pk(v[2 * e], v[2 * e + 1], p0[e], p1[e], pk);
                    const uint32_t ad = lds0 + (uint32_t)ch * XB + (uint32_t)(xkh[it] * LW + xcol[it]) * 16u;
                    ds_write128(ad, u32x4{p0[0], p0[1], p0[2], p0[3]});
                    ds_write128(ad + (uint32_t)(2 * LW) * 16u, u32x4{p1[0], p1[1], p1[2], p1[3]});
                }
            });
        });
    }

    f32x16 acc[NW];
    auto zero_acc = [&]() {
#pragma unroll
        for (int n = 0; n < NW; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    };
    zero_acc();
    u32x4 fb[NW][2];
    // B fragments of one half (block columns [h*NH, (h+1)*NH)) from the rows at byte address `rows` (plane 0,
    // half-group `hi` selected by the lane), plane stride `pstride`
    auto load_b_half = [&](auto H, uint32_t rows, uint32_t pstride) {
        constexpr int h = decltype(H)::value;
        static_for<NH>([&](auto N) {
            constexpr int n = h * NH + decltype(N)::value;
            fb[n][0] = ds_read128<n * 512>(rows);
            fb[n][1] = ds_read128<n * 512>(rows + pstride);
        });
    };
    auto mma_half = [&](const ASet &f, auto H) {
        constexpr int h = decltype(H)::value;
        // (f16x3: g1*h0, g0'*h1', g0*h0 - the order of conv_sx_kernel)
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int n = h * NH; n < (h + 1) * NH; n++) {
                const f16x8 ga = c == 0 ? __builtin_bit_cast(f16x8, f.fa[1])
                                        : (c == 1 ? __builtin_bit_cast(f16x8, f.fa[0]) * (_Float16)0.00048828125f
                                                  : __builtin_bit_cast(f16x8, f.fa[0]));
                const f16x8 gb = __builtin_bit_cast(f16x8, fb[n][c == 1 ? 1 : 0]);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga, gb, acc[n], 0, 0, 0);
            }
    };
    auto wait_lds_older_half = [&]() {  // a half = NH * 2 reads
        if constexpr (NH * 2 == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    };
    const std::integral_constant<int, 0> H0{};
    const std::integral_constant<int, 1> H1{};

    // One conv over an operand that is completely resident in LDS (chunk c at `rows0 + c * chunk_bytes`, tap k `k * dil`
    // cells to the right): weights DEPTH steps ahead in DEPTH + 1 register sets (a step of a one-block-row wave is 6-12
    // MFMAs, 190-380 cycles; an L2 round trip under load is several of those), B fragments half a step ahead, no
    // barriers.  The first weights (step 0) must have been requested into fs[0].
    auto wait_a = [&](int younger) {  // A(s) has landed when only the `younger` later sets are in flight (loads return in order)
        switch (younger) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        }
    };
    auto run_conv = [&](ASet(&fs)[DEPTH + 1], const char *wb, int K, int dil, uint32_t rows0, uint32_t chunk_bytes,
                        uint32_t pstride) {
        constexpr int NS = DEPTH + 1;
        const int S = a.nchunks * K;
        int chunk = 0, tap = 0;
        static_for<DEPTH - 1>([&](auto I) {
            constexpr int i = decltype(I)::value + 1;
            if (i < S) load_a(fs[i], wb, i);
        });
        load_b_half(H0, rows0, pstride);
        load_b_half(H1, rows0, pstride);
        auto step = [&](ASet &fc, ASet &fload, int s) {
            const int left = S - 1 - s;
            wait_a(left < DEPTH - 1 ? left : DEPTH - 1);
            __builtin_amdgcn_sched_barrier(0);
            if (s + DEPTH < S && !SX_NOA) load_a(fload, wb, s + DEPTH);
            int ntap = tap + 1, nchunk = chunk;
            if (ntap == K) {
                ntap = 0;
                nchunk++;
            }
            const bool more = s + 1 < S;
            const uint32_t next = rows0 + (uint32_t)nchunk * chunk_bytes + (uint32_t)(ntap * dil) * 16u;
            __builtin_amdgcn_sched_barrier(0);
            wait_lds_older_half();
            __builtin_amdgcn_sched_barrier(0);
            mma_half(fc, H0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                load_b_half(H0, next, pstride);
                wait_lds_older_half();
            } else
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma_half(fc, H1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) load_b_half(H1, next, pstride);
            __builtin_amdgcn_sched_barrier(0);
            tap = ntap;
            chunk = nchunk;
        };
        // (no exit from the middle of the unrolled group: a mid-loop break makes hipcc copy the accumulators)
        int s = 0;
        for (; s + NS <= S; s += NS)
            static_for<NS>([&](auto I) {
                constexpr int i = decltype(I)::value;
                step(fs[i], fs[(i + DEPTH) % NS], s + i);
            });
        // hipcc sinks kernel-argument loads (s_load) that only the epilogue uses to THIS point, the block between the unrolled
        // loop and its tail.  A scalar load in flight counts in lgkmcnt and returns out of order, so the tail steps' counted
        // `s_waitcnt lgkmcnt(n)` could pass with a ds_read still outstanding: drain the counter once here (per tile, not per step).
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        static_for<NS - 1>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (s + i < S) step(fs[i], fs[(i + DEPTH) % NS], s + i);
        });
    };

    // ---- (experiment, SX_PAIR_SHARED_A) 32-channel variant: the four waves of a workgroup use the SAME weight rows (one
    // block row); as four private register streams the weights cost a quarter of these launches' time by ablation (no
    // re-fetch = -25..32 %), and deeper private look-ahead makes it worse.  Here a step's weights cross the L2 -> CU path
    // once: a ring of RG groups of two steps in LDS, each wave fetching a quarter of a group with one 1 KiB LDS-DMA (step
    // 2 g + (wave >> 1), plane wave & 1), two groups ahead; one barrier per group publishes it (and says that everyone is
    // done with the group before; none when the conv fits the ring).  A fragments then come from LDS like B's.  Measured:
    // no gain, also where the conv is resident before the phase starts (k = 3) - the weight fetch is not what bounds these
    // phases (the ablation's gain is a clock effect of stale operands).  Kept for the record, off by default.
    constexpr bool SHA = NW == 2 && WM == 1 && NCH == 1 && SX_PAIR_SHARED_A != 0;
    constexpr int RG = 3;
    auto sha_dma = [&](const char *wb, int g) {  // this wave's quarter of group g
        const int sub = wave >> 1, pl = wave & 1;
        const char *src = wb + (int64_t)(2 * g + sub) * STEPBYTES + pl * 1024 + lane * 16;
        lds_dma<16>(src, reinterpret_cast<float *>(lds_sx + a.a_ring + (unsigned)(((g % RG) * 2 + sub) * 2048 + pl * 1024)));
    };
    auto sha_prefetch = [&](const char *wb, int K) {  // the first RG groups of a conv (the ring must be free)
        const int NG = (a.nchunks * K) >> 1;
#pragma unroll
        for (int g = 0; g < RG; g++)
            if (g < NG) sha_dma(wb, g);
    };
    auto wait_vm_le = [&](int n) {
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        }
    };
    auto run_conv_sha = [&](ASet &fa0, ASet &fa1, const char *wb, int K, int dil, uint32_t rows0, uint32_t chunk_bytes,
                            uint32_t pstride) {
        const int S = a.nchunks * K, NG = S >> 1;  // (two 16-channel chunks: S is even)
        const uint32_t alds = lds0 + a.a_ring + (uint32_t)lane * 16u;
        auto read_a = [&](ASet &f, int st) {
            const uint32_t ad = alds + (uint32_t)((((st >> 1) % RG) * 2 + (st & 1)) * 2048);
            f.fa[0] = ds_read128<0>(ad);
            f.fa[1] = ds_read128<1024>(ad);
        };
        int chunk = 0, tap = 0;
        // group 0: this wave's quarter has landed when only its younger prefetches are in flight; the barrier publishes it
        wait_vm_le((NG < RG ? NG : RG) - 1);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        read_a(fa0, 0);
        load_b_half(H0, rows0, pstride);
        load_b_half(H1, rows0, pstride);
        auto step = [&](ASet &fc, ASet &fload, int st, bool boundary) {
            int ntap = tap + 1, nchunk = chunk;
            if (ntap == K) {
                ntap = 0;
                nchunk++;
            }
            const bool more = st + 1 < S;
            const uint32_t next = rows0 + (uint32_t)nchunk * chunk_bytes + (uint32_t)(ntap * dil) * 16u;
            __builtin_amdgcn_sched_barrier(0);
            wait_lds_older_half();  // in flight, oldest first: A(st), B half 0, B half 1 -> the first two have landed
            __builtin_amdgcn_sched_barrier(0);
            mma_half(fc, H0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                if (boundary && NG > RG) {  // (a conv of at most RG groups is resident from the start: no hand-shake)
                    // the next step opens group g + 1: my quarter of it is back when only the groups behind it are in
                    // flight; after the barrier everyone has read group g for the last time, its slot takes group g + RG
                    const int g = st >> 1;
                    const int last = g + RG - 1 < NG - 1 ? g + RG - 1 : NG - 1;
                    wait_vm_le(last - (g + 1));
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + RG < NG) sha_dma(wb, g + RG);
                }
                read_a(fload, st + 1);
                load_b_half(H0, next, pstride);
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");  // B half 1 of this step has landed
            } else
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma_half(fc, H1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) load_b_half(H1, next, pstride);
            __builtin_amdgcn_sched_barrier(0);
            tap = ntap;
            chunk = nchunk;
        };
        for (int g = 0; g < NG; g++) {
            step(fa0, fa1, 2 * g, false);
            step(fa1, fa0, 2 * g + 1, true);
        }
    };

    // per-chain parameters: the argument's own fields, or (fused chains) entry ci of a.ch
    struct ChainPar {
        const char *wb1, *wb2;
        const float *bias1, *bias2;
        float ws1, ws2;
        int K1, dil1, K2, dil2;
        uint32_t xoff, yoff;
    };
    auto chain_par = [&](int ci) {
        ChainPar c;
        if constexpr (NCH == 1) {
            c = ChainPar{wbase1, wbase2, a.bias1, a.bias2, a.wscale1, a.wscale2, a.K1, a.dil1, a.K2, a.dil2, 0u, 0u};
        } else {
            const auto &h = a.ch[ci];
            c = ChainPar{reinterpret_cast<const char *>(h.wp1) + wm * (MW * NPW * 1024),
                         reinterpret_cast<const char *>(h.wp2) + wm * (MW * NPW * 1024),
                         h.bias1, h.bias2, h.wscale1, h.wscale2, h.K1, h.dil1, h.K2, h.dil2, (uint32_t)h.xoff, (uint32_t)h.yoff};
        }
        return c;
    };
    static_assert(NCH == 1 || EARLY, "fused chains keep x in registers from the prologue on");
    // Y: over the x stages (one chain: x is dead after phase 1), or behind them (fused chains: x is every chain's input)
    const uint32_t ylds = NCH > 1 ? lds0 + (uint32_t)a.nchunks * XB : lds0;
    // (xkeep = x in the accumulator layout, saved at the first hand-over: `pre` is still in flight here - it was requested
    // by the prologue's asm loads and is only known to have landed once phase 1 has waited for its last weights)
    f32x4 xkeep[NCH > 1 ? NW / 2 : 1][2][4], tot[NCH > 1 ? NW / 2 : 1][2][4];
#if SX_PAIR_ALIAS
    // one set of weight registers for both phases (phase 2's first weights are requested when phase 1 has consumed its
    // last): the look-ahead can be twice as deep for the same register count
    ASet f1s[DEPTH + 1];
    ASet(&f2s)[DEPTH + 1] = f1s;
#else
    ASet f1s[DEPTH + 1], f2s[DEPTH + 1];
#endif
#pragma unroll 1
    for (int ci = 0; ci < NCH; ci++) {
    const ChainPar cp = chain_par(ci);
    if constexpr (NCH > 1) {
        if (ci > 0) {
#pragma unroll
            for (int rr = 0; rr < NW / 2; rr++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int q = 0; q < 4; q++) pre[rr][j][q] = xkeep[rr][j][q];
            zero_acc();
        }
    }
    // =================================================================== phase 1: c1 over columns [t1, t1 + 256)
    if constexpr (!SHA) load_a(f1s[0], cp.wb1, 0);
    if (ci == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if SX_PAIR_PROF
        tp[2] = stamp();  // converted and written to LDS
#endif
        __builtin_amdgcn_s_barrier();  // the x tile is complete
#if SX_PAIR_PROF
        tp[3] = stamp();
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SHA)
        run_conv_sha(f1s[0], f1s[1], cp.wb1, cp.K1, cp.dil1, lds0 + (uint32_t)(hi * LW + wn * (NW * 32) + l31 + (int)cp.xoff) * 16u,
                     XB, (uint32_t)(2 * LW) * 16u);
    else
        run_conv(f1s, cp.wb1, cp.K1, cp.dil1, lds0 + (uint32_t)(hi * LW + wn * (NW * 32) + l31 + (int)cp.xoff) * 16u, XB,
                 (uint32_t)(2 * LW) * 16u);

    // ---- the residual (64-channel variant): the tile's lines were fetched a phase or two ago; CHAIN needs them now
    // (x1 = c1(..) + x at the hand-over), PAIR only in the epilogue and requests them there, so that they do not occupy
    // registers during phase 2.
#if SX_PAIR_PROF
    tp[4] = stamp();  // phase 1 done
#endif
    if constexpr (CHAIN && !EARLY) load_pre();

    // =================================================================== hand-over: c1's output -> Y (fp16 planes in LDS)
    if constexpr (!SHA) load_a(f2s[0], cp.wb2, 0);  // first weights of c2 travel meanwhile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // every wave has finished reading what Y is about to overwrite: the x stages (one chain), the previous chain's Y
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SHA) sha_prefetch(cp.wb2, cp.K2);  // (... and the weight ring: c2's first groups travel during the hand-over)
    {
        const float wsc = cp.ws1, msl = a.mslope;
        const float *biasp = cp.bias1 ? cp.bias1 : a.zeros;
        const int b_on = cp.bias1 ? 1 : 0;
        const uint32_t YC = a.y_chunk_bytes, LW2 = (uint32_t)a.LW2;
        const int row0 = wm * 32;
        f32x4 bq[4];
        if constexpr (BIAS_LDS) bias_from_lds(0, bq);
        else {
#pragma unroll
            for (int q = 0; q < 4; q++) bq[q] = *reinterpret_cast<const f32x4 *>(biasp + (row0 + 8 * q + 4 * hi) * b_on);
        }
#pragma unroll
        for (int n = 0; n < NW; n++) {
            const int col = (wn * NW + n) * 32 + l31;
            const int t = t1 + col;
            const bool live = t >= 0 && t < TV;  // outside the tensor c2 sees zero padding, not c1 evaluated there
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float v = __builtin_fmaf(acc[n][4 * q + e], wsc, bq[q][e]);
                    if constexpr (CHAIN) {  // x1 = c1(lrelu(x)) + x: c2's input and, kept in `pre`, its residual
                        if constexpr (NCH > 1) {
                            if (ci == 0) xkeep[n / 2][n % 2][q][e] = pre[n / 2][n % 2][q][e];
                        }
                        v += pre[n / 2][n % 2][q][e];
                        pre[n / 2][n % 2][q][e] = v;
                    }
                    o[e] = live ? fmaxf(v, v * msl) : 0.f;
                }
                unsigned wa[2], wb[2];
                split2h_pair_pk(o[0], o[1], wa[0], wa[1], pk);
                split2h_pair_pk(o[2], o[3], wb[0], wb[1], pk);
                // channel group g = 4 * wm + q -> chunk g / 2, half g % 2; 4 channels = 8 bytes of the 16-byte cell
                const int g = 4 * wm + q;
                const uint32_t cell = ylds + (uint32_t)(g >> 1) * YC + ((uint32_t)(g & 1) * LW2 + (uint32_t)(col + a.pad2)) * 16u + 8u * hi;
                asm volatile("ds_write_b64 %0, %1" ::"v"(cell), "v"(u32x2{wa[0], wb[0]}) : "memory");
                asm volatile("ds_write_b64 %0, %1" ::"v"(cell + 2u * LW2 * 16u), "v"(u32x2{wa[1], wb[1]}) : "memory");
            }
        }
    }
    zero_acc();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // Y is complete
    __builtin_amdgcn_sched_barrier(0);
#if SX_PAIR_PROF
    tp[5] = stamp();  // hand-over done
#endif

    // =================================================================== phase 2: c2 over Y
    if constexpr (SHA)
        run_conv_sha(f2s[0], f2s[1], cp.wb2, cp.K2, cp.dil2, ylds + (uint32_t)(hi * a.LW2 + wn * (NW * 32) + l31 + (int)cp.yoff) * 16u,
                     a.y_chunk_bytes, (uint32_t)(2 * a.LW2) * 16u);
    else
        run_conv(f2s, cp.wb2, cp.K2, cp.dil2, ylds + (uint32_t)(hi * a.LW2 + wn * (NW * 32) + l31 + (int)cp.yoff) * 16u,
                 a.y_chunk_bytes, (uint32_t)(2 * a.LW2) * 16u);
#if SX_PAIR_PROF
    tp[6] = stamp();  // phase 2 done
#endif
    if constexpr (!CHAIN && !EARLY) load_pre();

    // =================================================================== epilogue: bias2 + x [+ xs] [/ n] -> raw
    {
        constexpr int flags = EPI;
        float *rawb = a.out_raw + (int64_t)b * a.C * T;
        const float wsc = cp.ws2, rdiv = a.div;
        const float *biasp = cp.bias2 ? cp.bias2 : a.zeros;
        const int b_on = cp.bias2 ? 1 : 0;
        const int row0 = wm * 32;
        f32x4 bq[4];
        if constexpr (BIAS_LDS) bias_from_lds(1, bq);
        else {
#pragma unroll
            for (int q = 0; q < 4; q++) bq[q] = *reinterpret_cast<const f32x4 *>(biasp + (row0 + 8 * q + 4 * hi) * b_on);
        }
        static_for<NW / 2>([&](auto R) {
            constexpr int rr = decltype(R)::value;
            f32x4 adl[2][4];
            if constexpr ((flags & EPI_ACC) != 0 && !(EARLY && ACC)) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int t = t1 + (wn * NW + rr * 2 + j) * 32 + l31;
                    const int tl = t < 0 ? 0 : (t < T ? t : T - 1);
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        adl[j][q] = *reinterpret_cast<const f32x4 *>(rawb + ((int64_t)((row0 >> 3) + q) * T + tl) * 8 + 4 * hi);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int n = rr * 2 + j;
                const int col = (wn * NW + n) * 32 + l31;
                const int t = t1 + col;
                const bool kept = !(col < a.pad2 || col >= a.pad2 + a.BNo || t >= TV);  // overlap columns belong to the neighbours
                if constexpr (NCH == 1) {
                    if (!kept) continue;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = __builtin_fmaf(acc[n][4 * q + e], wsc, bq[q][e]);
                    v += pre[rr][j][q];
                    if constexpr (NCH > 1) {
                        // (the order of the separate launches: this chain's result + the running sum)
                        if (ci > 0) v += tot[rr][j][q];
                        tot[rr][j][q] = v;
                        if (ci < NCH - 1 || !kept) continue;
                    }
                    if constexpr ((flags & EPI_ACC) != 0) {
                        if constexpr (EARLY && ACC) v += ad[rr][j][q];
                        else v += adl[j][q];
                    }
                    if constexpr ((flags & EPI_DIV) != 0) {
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = v[e] / rdiv;
                    }
#if SX_PAIR_ST_SC1  // (experiment) write-through stores that do not stay in the XCD's L2
                    {
                        float *sp = rawb + ((int64_t)((row0 >> 3) + q) * T + t) * 8 + 4 * hi;
                        const f32x4 sv = v;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(sp), "v"(sv) : "memory");
                    }
#else
                    *reinterpret_cast<f32x4 *>(rawb + ((int64_t)((row0 >> 3) + q) * T + t) * 8 + 4 * hi) = v;
#endif
                }
            }
        });
    }
    }  // chains
#if SX_PAIR_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tp[7] = stamp();  // stores have left
    if (a.prof && tid == 0) {
        for (int i = 0; i < 7; i++) atomicAdd(a.prof + i, tp[i + 1] - tp[i]);
        atomicAdd(a.prof + 7, 1ull);
    }
#endif
    if (a.peak) sx_publish_peak(a.peak, (int)blockIdx.x, pk);  // (uniform branch; every thread arrives)
}

template <int MW, int NW, int WM, int WN, int EPI, bool CHAIN, int NCH = 1>
inline hipError_t launch_conv_sx_pair_k(const SxPairArgs &a, dim3 grid, size_t lds, hipStream_t stream) {
    static std::atomic<uint64_t> attr_done{0};
    auto kern = conv_sx_pair_kernel<MW, NW, WM, WN, EPI, CHAIN, NCH>;
    if (hipError_t e = sx_allow_big_lds(reinterpret_cast<const void *>(kern), attr_done); e != hipSuccess) return e;
    if (g_launch_name_on)
        snprintf(g_launch_name, sizeof g_launch_name, "conv_sx_pair_kernel<%d, %d, %d, %d, %d, %s, %d>", MW, NW, WM, WN, EPI,
                 CHAIN ? "true" : "false", NCH);
    kern<<<grid, 256, lds, stream>>>(a);
    return hipGetLastError();
}

// Can these two convs (both C -> C, same tile config cfg in {1: 64 rows, 2: 32 rows}, f16 planes) run fused?
inline bool sx_pair_supported(int C, int cfg, int K1, int dil1, int K2, int dil2) {
    if (K1 < 3 || K2 < 3 || dil1 < 1 || dil2 < 1) return false;
    if (!((C == 64 && cfg == 1) || (C == 32 && cfg == 2))) return false;  // one row tile holds every channel
    const int LW1 = 256 + (K1 - 1) * dil1;
    if (2 * LW1 > 768) return false;                                      // x staging: three cells per thread
    const int halo2 = (K2 - 1) * dil2;
    static const int keep_min = [] {
        const char *e = std::getenv("VITSMI_PAIR_MIN_KEEP");  // A/B timing only
        return e ? std::atoi(e) : 160;
    }();
    if (halo2 % 2 || 256 - halo2 < keep_min) return false;                // (> 37 % of a tile recomputed: not worth it)
    // 64 channels are matrix-heavier: a k = 7, dilation (3, 12) chain keeps 184 of 256 columns and measured 909 us fused
    // against 472 + 399 as two launches (32 channels, HBM / latency-bound, still gain at that ratio)
    static const int keep64 = [] {
        const char *e = std::getenv("VITSMI_PAIR_MIN_KEEP64");  // A/B timing only
        return e ? std::atoi(e) : 200;
    }();
    if (C == 64 && 256 - halo2 < keep64) return false;
    const size_t lds_y = (size_t)(C / 16) * 4 * (size_t)((256 + halo2 / 2 + 7) / 8 * 8) * 16 + (size_t)(halo2 / 2) * 16;
    const size_t lds_x = (size_t)(C / 16) * 4 * LW1 * 16;                 // the whole x tile is resident
    return (lds_y > lds_x ? lds_y : lds_x) <= 80 * 1024 - 256;            // two workgroups per CU
}

// flags: EPI_ACC (out += ...), EPI_DIV (then / div).  chain = false: out = c2(lrelu(c1(lrelu(x)))) + x;
// chain = true: x1 = c1(lrelu(x)) + x, out = c2(lrelu(x1)) + x1.
hipError_t launch_conv_sx_pair(SxPairArgs a, int cfg, int B, hipStream_t stream, bool chain = false);
hipError_t launch_conv_sx_mrf(SxPairArgs a, int B, hipStream_t stream);
#ifdef VITSMI_IMPL_PAIR  // (tu_pair.hip)
hipError_t launch_conv_sx_pair(SxPairArgs a, int cfg, int B, hipStream_t stream, bool chain) {
    a.LW1 = 256 + (a.K1 - 1) * a.dil1;
    a.x_bytes = (unsigned)((size_t)4 * a.LW1 * 16);  // one 16-channel chunk: 2 planes x 2 halves x LW1 cells
    if (a.dil2 < 1) a.dil2 = 1;
    const int halo2 = (a.K2 - 1) * a.dil2;
    if (a.pad2 * 2 != halo2) return hipErrorInvalidValue;  // "same" padding
    // Y is stored pad2 columns to the right.  A row holds the 256 + pad2 columns phase 1 writes - all that the KEPT
    // outputs (columns [pad2, 256 - pad2)) read; the outputs right of them read up to pad2 cells past the row's end, i.e.
    // the next row's (or, for the last row, the slack's) bytes, and are discarded.  (With 256 + 2 pad2 cells per row a
    // 64-channel k = 7, dilation 12 step did not fit two workgroups per CU.)
    a.LW2 = (256 + a.pad2 + 7) / 8 * 8;
    a.y_chunk_bytes = (unsigned)(4 * a.LW2 * 16);
    a.BNo = 256 - halo2;
    a.NT = (a.T + a.BNo - 1) / a.BNo;
    a.B = B;
    a.nchunks = a.C / 16;
    if (a.islope == 0.f) a.islope = 1.f;
    if (a.mslope == 0.f) a.mslope = 1.f;
    if (a.wscale1 == 0.f) a.wscale1 = 1.f;
    if (a.wscale2 == 0.f) a.wscale2 = 1.f;
    if ((long long)a.T * 64 + 64 >= (1ll << 32)) return hipErrorInvalidValue;
    // Y overlays the x stages (every chunk has its own)
    const size_t lds_x = (size_t)a.nchunks * a.x_bytes, lds_y = (size_t)a.nchunks * a.y_chunk_bytes + (size_t)a.pad2 * 16;
    size_t lds = lds_x > lds_y ? lds_x : lds_y;
    if (lds > 80 * 1024 - 256 || !sx_pair_supported(a.C, cfg, a.K1, a.dil1, a.K2, a.dil2)) return hipErrorInvalidValue;
    if (cfg != 1 && SX_PAIR_SHARED_A) {  // 32 channels: the shared weight ring behind the tile (3 groups x 2 steps x 2 KiB)
        a.a_ring = (unsigned)((lds + 1023) / 1024 * 1024);
        lds = a.a_ring + 3 * 4096;
    }
    a.bias_off = (unsigned)((lds + 15) / 16 * 16);  // the two bias vectors: one 1 KiB DMA slot
    lds = a.bias_off + 1024;
    const long long nb = (long long)a.NT * B;
    if (nb == 0) return hipSuccess;
    a.xgs = a.NT >= (2 << sx_xcd_group_shift()) ? sx_xcd_group_shift() : 0;
    const long long per_round = 8ll << a.xgs;
    const long long wgs = (nb + per_round - 1) / per_round * per_round;
    if (wgs >= (1ll << 31)) return hipErrorInvalidValue;
    dim3 grid((unsigned)wgs, 1, 1);
    const int epi = a.flags & (EPI_ACC | EPI_DIV);
    if ((epi & EPI_DIV) && !(epi & EPI_ACC)) return hipErrorInvalidValue;
#define SX_PAIR_CASES(MW, NW, WM, WN, CH)                                                                        \
    switch (epi) {                                                                                               \
        case 0: return launch_conv_sx_pair_k<MW, NW, WM, WN, 0, CH>(a, grid, lds, stream);                       \
        case EPI_ACC: return launch_conv_sx_pair_k<MW, NW, WM, WN, EPI_ACC, CH>(a, grid, lds, stream);           \
        default: return launch_conv_sx_pair_k<MW, NW, WM, WN, EPI_ACC | EPI_DIV, CH>(a, grid, lds, stream);      \
    }
    if (cfg == 1) {
        if (chain) {
            SX_PAIR_CASES(1, 4, 2, 2, true)
        }
        SX_PAIR_CASES(1, 4, 2, 2, false)
    }
    if (chain) {
        SX_PAIR_CASES(1, 2, 1, 4, true)
    }
    SX_PAIR_CASES(1, 2, 1, 4, false)
#undef SX_PAIR_CASES
}
#endif  // VITSMI_IMPL_PAIR

// Fused multi-receptive-field stage (conv_sx_pair_kernel<.., NCH>): n = 2 or 3 two-step ResBlock2 chains of a 32-channel
// stage, out = (sum of the chains) / div.  K1 / dil1 / K2 / dil2 per chain ("same" padding on both convs).
struct SxMrfGeom {
    int pad1, pad2, LW1, LW2;
    size_t lds;
};
inline bool sx_mrf_geom(int C, int n, const int *K1, const int *dil1, const int *K2, const int *dil2, SxMrfGeom *g) {
    if (C != 32 || n < 2 || n > 3) return false;
    int p1 = 0, p2 = 0;
    for (int i = 0; i < n; i++) {
        if (K1[i] < 3 || K2[i] < 3 || !(K1[i] & 1) || !(K2[i] & 1) || dil1[i] < 1 || dil2[i] < 1) return false;
        const int a1 = (K1[i] - 1) / 2 * dil1[i], a2 = (K2[i] - 1) / 2 * dil2[i];
        p1 = a1 > p1 ? a1 : p1;
        p2 = a2 > p2 ? a2 : p2;
    }
    const int LW1 = 256 + 2 * p1;
    if (2 * LW1 > 768) return false;       // x staging: three cells per thread
    if (256 - 2 * p2 < 160) return false;  // (as sx_pair_supported: more than 37 % of a tile recomputed)
    const int LW2 = (256 + p2 + 7) / 8 * 8;
    const size_t lds = (size_t)(C / 16) * 4 * LW1 * 16 + (size_t)(C / 16) * 4 * LW2 * 16 + (size_t)p2 * 16;
    if (lds > 80 * 1024 - 256) return false;  // two workgroups per CU
    if (g) *g = SxMrfGeom{p1, p2, LW1, LW2, lds};
    return true;
}

// a: xr, islope, mslope, T, out_raw, zeros, C, div, peak and ch[0 .. nchain) (wp / bias / wscale / K / dil; xoff, yoff are
// filled in here)
#ifdef VITSMI_IMPL_PAIR
hipError_t launch_conv_sx_mrf(SxPairArgs a, int B, hipStream_t stream) {
    int K1[3], d1[3], K2[3], d2[3];
    if (a.nchain < 2 || a.nchain > 3) return hipErrorInvalidValue;
    for (int i = 0; i < a.nchain; i++) {
        K1[i] = a.ch[i].K1;
        d1[i] = a.ch[i].dil1;
        K2[i] = a.ch[i].K2;
        d2[i] = a.ch[i].dil2;
    }
    SxMrfGeom g;
    if (!sx_mrf_geom(a.C, a.nchain, K1, d1, K2, d2, &g)) return hipErrorInvalidValue;
    for (int i = 0; i < a.nchain; i++) {
        a.ch[i].xoff = g.pad1 - (K1[i] - 1) / 2 * d1[i];
        a.ch[i].yoff = g.pad2 - (K2[i] - 1) / 2 * d2[i];
        if (a.ch[i].wscale1 == 0.f) a.ch[i].wscale1 = 1.f;
        if (a.ch[i].wscale2 == 0.f) a.ch[i].wscale2 = 1.f;
    }
    a.pad1 = g.pad1;
    a.pad2 = g.pad2;
    a.LW1 = g.LW1;
    a.x_bytes = (unsigned)((size_t)4 * a.LW1 * 16);
    a.LW2 = g.LW2;
    a.y_chunk_bytes = (unsigned)(4 * a.LW2 * 16);
    a.BNo = 256 - 2 * g.pad2;
    a.NT = (a.T + a.BNo - 1) / a.BNo;
    a.B = B;
    a.nchunks = a.C / 16;
    if (a.islope == 0.f) a.islope = 1.f;
    if (a.mslope == 0.f) a.mslope = 1.f;
    if ((long long)a.T * 64 + 64 >= (1ll << 32)) return hipErrorInvalidValue;
    const long long nb = (long long)a.NT * B;
    if (nb == 0) return hipSuccess;
    a.xgs = a.NT >= (2 << sx_xcd_group_shift()) ? sx_xcd_group_shift() : 0;
    const long long per_round = 8ll << a.xgs;
    const long long wgs = (nb + per_round - 1) / per_round * per_round;
    if (wgs >= (1ll << 31)) return hipErrorInvalidValue;
    dim3 grid((unsigned)wgs, 1, 1);
    a.flags = EPI_DIV;
    if (a.nchain == 2) return launch_conv_sx_pair_k<1, 2, 1, 4, EPI_DIV, true, 2>(a, grid, g.lds, stream);
    return launch_conv_sx_pair_k<1, 2, 1, 4, EPI_DIV, true, 3>(a, grid, g.lds, stream);
}
#endif  // VITSMI_IMPL_PAIR

}  // namespace vitsmi
