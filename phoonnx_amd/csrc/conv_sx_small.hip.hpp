// conv_sx_small.hip.hpp — the split-operand conv (f16x3 arithmetic, 16x16x32 packing) for SHORT launches: one utterance, a
// streaming chunk, a handful of frames.  conv_sx_kernel (conv_sx_engine.hip.hpp) is built for throughput: 64..128-row tiles
// that walk the whole reduction (Cin / 32 chunks x K taps) one step at a time, each step behind one global-load latency of
// its weights and one barrier per chunk.  At batch 1 the grid is a few dozen workgroups, nothing hides those latencies, and a
// token- or frame-domain conv takes 12..48 us (launch table) for a few MFLOP.  This kernel shortens the chain:
//   * the four waves of a workgroup SPLIT THE REDUCTION of one 32 x 32 output tile (wave w takes steps w, w + 4, ..) and add
//     their partial tiles through LDS in a fixed order (deterministic);
//   * both operands stream from global memory (L2) straight into registers - the weights as packed for the 16x16x32 loop
//     (model.cpp pack_conv_sx: 1 KiB per (32-row block, sub-block, plane) and step), the activations as the 16-byte cells
//     of the plane tensor, which ARE the B operand of v_mfma_f32_16x16x32_f16 (a lane: 8 channels of one time step); taps
//     outside [0, T) read a zero page - through a ring of D steps in flight with counted vmcnt waits; no LDS staging, no
//     barrier in the loop;
//   * three MFMAs per fp32 product as in the engine, here as two accumulators (g0 h0 + g1 h0, and g0 h1' which carries 2^-11).
// Epilogues: the planar one of the token / frame domain (SX_WN_RMW: o = old + act(acc + bias) * mask into one or two planar
// fp32 tensors and / or operand planes; the coupling variant), the WN gate (SX_GATE: tanh(a) * sigmoid(b) of a packed row
// pair, per-utterance conditioning bias) and the generator's (raw cells and / or planes, residual, running sum; ups == 1).  Same argument block (SxArgs) and semantics as conv_sx_kernel; conv_sx() in
// vitsmi.hip picks this kernel when the launch would be at most a few workgroups per CU.
#pragma once
#include "conv_sx_engine.hip.hpp"

namespace vitsmi {

template <bool GATE>
__global__ __launch_bounds__(256) void conv_sx_small_kernel(SxArgs a, int MBP) {
    constexpr int NBLK = GATE ? 2 : 1;   // 32-row blocks of the tile (gate: the tanh block and its sigmoid partner)
    constexpr int NCT = GATE ? 1 : 2;    // 16-column tiles
    constexpr int LA = NBLK * 4, LB = NCT * 2, LPS = LA + LB;  // 16-byte loads per lane and step: weights, activations
    constexpr int D = GATE ? 3 : 4;      // steps in flight
    __shared__ f32x4 red[4][4][64];      // [wave][MFMA tile][lane]: the partial tiles
    __shared__ float s_pk[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int T = a.T, K = a.K, dil = a.dil, S = a.nchunks * K;
    int mt, nt, b;
    {
        int id = blockIdx.x;
        mt = id % a.MT;
        id /= a.MT;
        nt = id % a.NT;
        b = id / a.NT;
    }
    const int t0 = nt * (16 * NCT);
    // TV: where this utterance's tensors end (SxRagged: the masked frame domain of a padded batch; = T otherwise).  T = row pitch.
    const int TV = __builtin_amdgcn_readfirstlane(sx_valid_cols(a.rag, b, T));
    if (t0 >= TV) return;  // (uniform exit) the whole tile lies behind the utterance's end
    // weights: step s of 32-row block mbi starts at ((mbi / MBP) * S * MBP + s * MBP + mbi % MBP) * 4 KiB
    const char *wb[NBLK];
#pragma unroll
    for (int k = 0; k < NBLK; k++) {
        const int mbi = mt * NBLK + k;
        wb[k] = reinterpret_cast<const char *>(a.wp) + ((int64_t)(mbi / MBP) * S * MBP + (mbi % MBP)) * 4096 + lane * 16;
    }
    const int64_t wstep = (int64_t)MBP * 4096;
    const u32x4 *xb = a.xp + (int64_t)b * a.x_bstride + (int64_t)g * T;  // this lane's channel group of chunk 0
    const int64_t plane_cells = (int64_t)(a.Cin >> 3) * T;
    const u32x4 *zero16 = reinterpret_cast<const u32x4 *>(a.zeros);
    int tcol[NCT];
#pragma unroll
    for (int n = 0; n < NCT; n++) tcol[n] = t0 + n * 16 + c - a.padL;

    u32x4 ra[D][LA], rb[D][LB];
    auto issue = [&](auto J, int s) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        const int chunk = s / K, tap = s - chunk * K;  // (uniform)
        static_for<NBLK>([&](auto Kb) {
            constexpr int k = decltype(Kb)::value;
            const char *p = wb[k] + (int64_t)s * wstep;
            ra[j][k * 4 + 0] = global_read128_v<0>(p);
            ra[j][k * 4 + 1] = global_read128_v<1024>(p);
            ra[j][k * 4 + 2] = global_read128_v<2048>(p);
            ra[j][k * 4 + 3] = global_read128_v<3072>(p);
        });
        static_for<NCT>([&](auto Nn) {
            constexpr int n = decltype(Nn)::value;
            const int tt = tcol[n] + tap * dil;
            const u32x4 *p0 = (unsigned)tt < (unsigned)TV ? xb + (int64_t)chunk * 4 * T + tt : zero16;
            const u32x4 *p1 = (unsigned)tt < (unsigned)TV ? xb + plane_cells + (int64_t)chunk * 4 * T + tt : zero16;
            rb[j][n * 2 + 0] = global_read128_v<0>(p0);
            rb[j][n * 2 + 1] = global_read128_v<0>(p1);
        });
    };

    // accumulators of the 4 MFMA tiles: non-gate (sub-block, column tile), gate (block, sub-block)
    f32x4 accA[4], accB[4];
#pragma unroll
    for (int k = 0; k < 4; k++) accA[k] = accB[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nw = S > wave ? (S - wave + 3) / 4 : 0;  // this wave's steps: wave, wave + 4, ..
    // (every load that is issued is consumed: a load still in flight when its destination registers look dead to the compiler
    // would land in whatever was allocated there next)
    static_for<D>([&](auto J) {
        constexpr int j = decltype(J)::value;
        if (j < nw) issue(J, wave + 4 * j);
    });
    for (int i0 = 0; i0 < nw; i0 += D) {
        static_for<D>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int i = i0 + j;
            if (i < nw) {  // (uniform)
                // steps i .. i + D - 1 are in flight while that many remain; at the tail everything left is awaited at once
                if (i + D <= nw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // tile k: weights of (block, sub-block) = non-gate (0, k >> 1), gate (k >> 1, k & 1); columns: non-gate k & 1, gate 0
                    const int fa = GATE ? k * 2 : (k >> 1) * 2, fb = GATE ? 0 : (k & 1) * 2;
                    const f16x8 g0 = __builtin_bit_cast(f16x8, ra[j][fa]), g1 = __builtin_bit_cast(f16x8, ra[j][fa + 1]);
                    const f16x8 h0 = __builtin_bit_cast(f16x8, rb[j][fb]), h1 = __builtin_bit_cast(f16x8, rb[j][fb + 1]);
                    accA[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g0, h0, accA[k], 0, 0, 0);
                    accB[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g0, h1, accB[k], 0, 0, 0);
                    accA[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g1, h0, accA[k], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i + D < nw) issue(J, wave + 4 * (i + D));
            }
        });
    }
#pragma unroll
    for (int k = 0; k < 4; k++) red[wave][k][lane] = accA[k] + accB[k] * (1.f / 2048.f);
    __syncthreads();

    const float wsc = a.wscale;
    const float *biasp = a.bias ? a.bias : a.zeros;
    const int b_on = a.bias ? 1 : 0;
    float pk = 0.f;
    if constexpr (GATE) {
        if (wave < 2) {
            const int sub = wave;
            f32x4 va = red[0][sub][lane], vs = red[0][2 + sub][lane];
#pragma unroll
            for (int w = 1; w < 4; w++) {
                va += red[w][sub][lane];
                vs += red[w][2 + sub][lane];
            }
            const int H = a.Cr >> 1, ch0 = mt * 32;
            const int rbk = 8 * (2 * sub + (g & 1)) + 4 * (g >> 1);  // first of this lane's 4 consecutive rows inside a block
            const float *bbp = a.bias_b ? a.bias_b + (int64_t)b * a.bias_b_stride : a.zeros;
            const int bb_on = a.bias_b ? 1 : 0;
            const f32x4 ba = *reinterpret_cast<const f32x4 *>(biasp + (mt * 64 + rbk) * b_on) +
                             *reinterpret_cast<const f32x4 *>(bbp + (ch0 + rbk) * bb_on);
            const f32x4 bs = *reinterpret_cast<const f32x4 *>(biasp + (mt * 64 + 32 + rbk) * b_on) +
                             *reinterpret_cast<const f32x4 *>(bbp + (H + ch0 + rbk) * bb_on);
            const int t = t0 + c;
            if (t < TV) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    o[e] = tanh_nb(__builtin_fmaf(va[e], wsc, ba[e])) * sigmoid_nb(__builtin_fmaf(vs[e], wsc, bs[e]));
                if (a.out_pl) {  // (|acts| < 1: no range issue)
                    unsigned wa[2], wb2[2];
                    split2h_pair(o[0], o[1], wa[0], wa[1]);
                    split2h_pair(o[2], o[3], wb2[0], wb2[1]);
                    uint16_t *pl = a.out_pl + (int64_t)b * a.pl_bstride;
                    const int64_t cell = ((int64_t)((ch0 + rbk) >> 3) * T + t) * 8 + (rbk & 4);
                    *reinterpret_cast<u32x2 *>(pl + cell) = u32x2{wa[0], wb2[0]};
                    *reinterpret_cast<u32x2 *>(pl + (int64_t)(H >> 3) * T * 8 + cell) = u32x2{wa[1], wb2[1]};
                } else {
                    float *actb = a.out_raw + (int64_t)b * a.raw_bstride;
#pragma unroll
                    for (int e = 0; e < 4; e++) actb[(int64_t)(ch0 + rbk + e) * T + t] = o[e];
                }
            }
        }
        return;
    } else {
        // wave w finishes tile w = (sub-block w >> 1, column tile w & 1): lane (c, g) holds 4 consecutive rows of column c
        const int sub = wave >> 1, n = wave & 1;
        f32x4 val = red[0][wave][lane];
#pragma unroll
        for (int w = 1; w < 4; w++) val += red[w][wave][lane];
        if (!(a.flags & SX_WN_RMW)) {
            // the generator's epilogue (conv_sx_kernel's, ups == 1): bias [+ per-utterance bias] [+ residual] [+ running sum] [/ n],
            // then fp32 raw cells [Cr/8][T][8] with leaky_relu(oslope) and / or operand planes with leaky_relu(oslope2)
            const int row0 = mt * 32 + 8 * (2 * sub + (g & 1)) + 4 * (g >> 1);
            const int t = t0 + n * 16 + c;
            if (t < TV) {
                const int64_t cell = ((int64_t)(row0 >> 3) * T + t) * 8 + (row0 & 4);
                float *rawb = a.out_raw ? a.out_raw + (int64_t)b * a.raw_bstride : nullptr;
                f32x4 v;
                const f32x4 bq = *reinterpret_cast<const f32x4 *>(biasp + row0 * b_on);
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = __builtin_fmaf(val[e], wsc, bq[e]);
                if (a.bias_b) v += *reinterpret_cast<const f32x4 *>(a.bias_b + (int64_t)b * a.bias_b_stride + row0);
                if (a.flags & EPI_RES) v += *reinterpret_cast<const f32x4 *>(a.res + (int64_t)b * a.raw_bstride + cell);
                if (a.flags & EPI_ACC) v += *reinterpret_cast<const f32x4 *>(rawb + cell);
                if (a.flags & EPI_DIV) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = v[e] / a.div;
                }
                if (rawb && !(a.flags & SX_NO_RAW_STORE)) {
                    f32x4 o = v;
                    if (a.oslope != 1.f) {
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = fmaxf(v[e], v[e] * a.oslope);
                    }
                    *reinterpret_cast<f32x4 *>(rawb + cell) = o;
                }
                if (a.out_pl) {
                    f32x4 o = v;
                    if (a.oslope2 != 1.f) {
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = fmaxf(v[e], v[e] * a.oslope2);
                    }
                    unsigned wa[2], wb2[2];
                    split2h_pair_pk(o[0], o[1], wa[0], wa[1], pk);
                    split2h_pair_pk(o[2], o[3], wb2[0], wb2[1], pk);
                    uint16_t *plb = a.out_pl + (int64_t)b * a.pl_bstride;
                    *reinterpret_cast<u32x2 *>(plb + cell) = u32x2{wa[0], wb2[0]};
                    *reinterpret_cast<u32x2 *>(plb + (int64_t)a.Cr * T + cell) = u32x2{wa[1], wb2[1]};
                }
            }
            if (a.peak) sx_publish_peak_at(a.peak, (int)blockIdx.x, pk, s_pk);  // (all threads arrive)
            return;
        }
        const bool p_acc = (a.flags & EPI_ACC) != 0, p_res = (a.flags & EPI_RES) != 0, p_relu = (a.flags & EPI_RELU) != 0;
        const bool coupling = (a.flags & SX_PLANAR_COUPLING) != 0;
        const int Lb = ((a.flags & EPI_MASK) && a.len) ? a.len[b] : T;
        const int row0 = mt * 32 + 8 * (2 * sub + (g & 1)) + 4 * (g >> 1);
        const int xrows = a.row_split, srows = a.Cout - a.row_split;
        const bool to_x = row0 < xrows;
        const int64_t xbs = a.planar_bstride ? a.planar_bstride : (int64_t)xrows * T;
        float *ob = to_x ? (a.out_raw ? a.out_raw + (int64_t)b * xbs : nullptr) : a.out_raw2 + (int64_t)b * srows * T;
        const float *oldp = p_acc && !(!to_x && (a.flags & SX_PLANAR_STORE2)) ? ob : (p_res && to_x ? a.res + (int64_t)b * xbs : nullptr);
        const int r0 = to_x ? row0 : row0 - xrows;
        uint16_t *plb = a.out_pl ? a.out_pl + (int64_t)b * a.pl_bstride : nullptr;
        const bool planes = plb && (a.pl_of2 ? !to_x : to_x) && r0 < a.pl_rows;
        const f32x4 bq = *reinterpret_cast<const f32x4 *>(biasp + row0 * b_on);
        const int t = t0 + n * 16 + c;
        if (t < TV) {
            const float mk = t < Lb ? 1.f : 0.f;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float old = oldp ? oldp[(int64_t)(r0 + e) * T + t] : 0.f;
                float v = __builtin_fmaf(val[e], wsc, bq[e]);
                if (p_relu) v = __builtin_fmaxf(v, 0.f);
                o[e] = coupling ? (old - v * mk) * mk : old + v * mk;
            }
            if (ob) {
#pragma unroll
                for (int e = 0; e < 4; e++) ob[(int64_t)(r0 + e) * T + t] = o[e];
            }
            if (planes) {
                unsigned wa[2], wb2[2];
                split2h_pair_pk(o[0], o[1], wa[0], wa[1], pk);
                split2h_pair_pk(o[2], o[3], wb2[0], wb2[1], pk);
                const int64_t cell = ((int64_t)(r0 >> 3) * T + t) * 8 + (r0 & 4);
                *reinterpret_cast<u32x2 *>(plb + cell) = u32x2{wa[0], wb2[0]};
                *reinterpret_cast<u32x2 *>(plb + (int64_t)(a.pl_rows >> 3) * T * 8 + cell) = u32x2{wa[1], wb2[1]};
            }
        }
        if (a.peak) sx_publish_peak_at(a.peak, (int)blockIdx.x, pk, s_pk);  // (all threads arrive)
    }
}

// Whether conv_sx() may hand this launch to the small kernel (the caller has decided that the launch is short), and the launch.
// `pack_cfg`: the tile height the weights were packed for (ConvDesc::cfg).
inline bool conv_sx_small_ok(const SxArgs &a, bool rawin, int nprod) {
    if (rawin || nprod != 2 || !a.s16 || a.ups != 1 || a.Cin % 32 || a.Cout % 32 || a.res_pl || a.prof) return false;
    if (a.flags & SX_GATE) return !(a.flags & SX_WN_RMW) && a.Cr % 64 == 0 && (a.out_raw || a.out_pl);
    if (!(a.flags & SX_WN_RMW))  // the generator's epilogue (plane-input convs of the > 64-channel stages, conv_pre)
        return a.Cr == a.Cout && (a.out_raw || a.out_pl) && !((a.flags & EPI_RES) && !a.res) && !((a.flags & EPI_ACC) && !a.out_raw) &&
               !((a.flags & EPI_DIV) && a.div == 0.f) && !(a.flags & (DBG_NO_DMA | DBG_NO_EPI));
    const bool p_acc = (a.flags & EPI_ACC) != 0;
    return !((a.row_split < a.Cout && !a.out_raw2) || (a.row_split && !a.out_raw && (p_acc || !a.out_pl)) || a.row_split % 32 ||
             a.row_split > a.Cout || a.pl_rows % 32 || a.pl_rows > (a.pl_of2 ? a.Cout - a.row_split : a.row_split) ||
             (a.pl_rows && !a.out_pl) || a.bias_b || ((a.flags & SX_PLANAR_COUPLING) && !p_acc) ||
             ((a.flags & EPI_RES) && (!a.res || p_acc)) || ((a.flags & EPI_MASK) && !a.len));
}
inline long long conv_sx_small_wgs(const SxArgs &a, int B) {
    const bool gate = (a.flags & SX_GATE) != 0;
    return (long long)(a.Cout / (gate ? 64 : 32)) * ((a.T + (gate ? 15 : 31)) / (gate ? 16 : 32)) * B;
}
inline hipError_t launch_conv_sx_small(SxArgs a, int B, int pack_cfg, hipStream_t stream) {
    const bool gate = (a.flags & SX_GATE) != 0;
    if (a.wscale == 0.f) a.wscale = 1.f;
    if (a.oslope == 0.f) a.oslope = 1.f;
    if (a.oslope2 == 0.f) a.oslope2 = 1.f;
    a.MT = a.Cout / (gate ? 64 : 32);
    a.NT = (a.T + (gate ? 15 : 31)) / (gate ? 16 : 32);
    a.B = B;
    const long long wgs = (long long)a.MT * a.NT * B;
    if (wgs == 0) return hipSuccess;
    if (wgs >= (1ll << 31)) return hipErrorInvalidValue;
    const int MBP = sx_tile_m(pack_cfg) / 32;
    if (g_launch_name_on) snprintf(g_launch_name, sizeof g_launch_name, "conv_sx_small_kernel<%s>", gate ? "true" : "false");
    if (gate) conv_sx_small_kernel<true><<<dim3((unsigned)wgs), 256, 0, stream>>>(a, MBP);
    else conv_sx_small_kernel<false><<<dim3((unsigned)wgs), 256, 0, stream>>>(a, MBP);
    return hipGetLastError();
}

}  // namespace vitsmi
