// attention16.hip.hpp — a3: the text encoder's relative-position self-attention (phoonnx_train/vits/attentions.py:215-272,
// helpers :274-348) on the 16-bit matrix pipe with fp32-grade products: QK^T and P V as f16x3 products on
// v_mfma_f32_16x16x32_f16 (operands as two fp16 planes, three MFMAs per fp32 product; softmax, relative-key bias and
// relative-value term in fp32).  It replaces attention_relpos_kernel (kernels.hip.hpp: v_mfma_f32_32x32x2_f32, 64 cycles per
// 2048 MACs - 96 such MFMAs per 32 x 32 block of scores - and 14 % matrix-pipe busy) where the head width is a multiple of 32
// and the q | k | v conv has written its result as operand planes (cells [channels / 8][T][8], conv_sx_engine.hip.hpp).
//
//   S^T[j, i] = sum_d K[d, j] Q[d, i]          A = K (a lane: 8 channels of one key = half a staged row), B = Q cells in
//                                              registers; two accumulators: the h0 h0 product, and the two cross products
//                                              that carry 2^-11
//   logits    = S^T / sqrt(dk) + q_i . E_k[j - i + w] (|j - i| <= w);  keys >= len: -1e4            (attentions.py:232-247)
//   O^T[d, i] = sum_j V[d, j] P^T[j, i]        B = P from the lane's OWN score accumulators (a lane of the 16 x 16 C layout
//                                              holds keys 4 g + r of query c; the k-slots of the next MFMA are ordered to
//                                              match), A = V with keys along k: read from the same [key][16 channels] image
//                                              as K with the transposing LDS read (ds_read_b64_tr_b16)
//   + sum_m w[m] E_v[m] (w[m] = sum of p over the keys at relative position m), all divided by the row sum  (:261-268)
// Queries sit on lanes (lane & 15) in both products: softmax statistics are per-lane scalars, reduced over the four lane
// groups by two shuffles per block.  One workgroup = 4 waves x QT x 16 queries; a 32-key block of K and V (both planes) is
// staged by LDS-DMA straight from the plane tensor - one instruction = one (plane, 16-channel) row of 32 keys x 32 bytes: the
// image [plane][d-tile][key][16 channels] serves the 16-byte row reads of K and the transposed reads of V without bank
// conflicts - NS stages deep (NS - 1 blocks in flight), one barrier per block.  The workgroups of one (utterance, head) run
// on one XCD (ids 8 apart): K and V come from HBM once and from that XCD's L2 for the other query tiles.
#pragma once
#include "conv_sx_engine.hip.hpp"

namespace vitsmi {

struct Att16Args {
    const uint16_t *qkv_pl;  // operand planes of q | k | v [B][3 slots][3 Hc / 8][T][8]
    float *out;              // planar fp32 [B][Hc][T] (may be nullptr when out_pl is given)
    uint16_t *out_pl;        // (optional) operand planes of the output for conv_o [B][3 slots][Hc / 8][T][8]
    const float *relk, *relv;  // [2 w + 1][dk]
    const int *len;
    int Hc, T, dk, win, nh, B;
    unsigned *peak;          // f16 range-guard slots for out_pl (may be nullptr)
    unsigned long long *prof;  // ATT16_PROF builds: 4 s_memtime stamps per workgroup (start, loop start, loop end, end)
};

#ifndef ATT16_PROF
#define ATT16_PROF 0
#endif
#if ATT16_PROF
#define ATT16_STAMP(k)                                                                   \
    do {                                                                                 \
        if (a.prof && threadIdx.x == 0) a.prof[(int64_t)blockIdx.x * 4 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define ATT16_STAMP(k) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void att16_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void att16_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ u32x2 ds_read64_tr16(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}

// max / sum over the four lanes that share lane & 15 (the four 16-lane rows of a wave): two row swaps instead of two
// ds_bpermute round trips through the LDS
__device__ __forceinline__ float att16_rows_max(float x) {
    auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
    s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
}
__device__ __forceinline__ float att16_rows_sum(float x) {
    auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
constexpr int kAtt16RelPitch = 12;  // floats per query in the relative-position tables (2 w + 1 <= 9 used)

template <int DKS, int NS, int QT>  // dk / 32, LDS stages, query tiles of 16 per wave
__global__ __launch_bounds__(256) void attention_relpos16_kernel(Att16Args a) {
    constexpr int DK = DKS * 32, NDT = DK / 16;
    constexpr int ROW = 1024;          // one staged row: (plane, d-tile) x 32 keys x 16 channels x 2 bytes
    constexpr int KB = 2 * NDT * ROW;  // the K half of a stage; V follows in the same layout
    constexpr int STAGE = 2 * KB;
    constexpr int RPW = NDT;           // DMA instructions per wave per block (4 NDT rows over 4 waves)
    constexpr int QW = 64 * QT;        // queries per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *relv_s = reinterpret_cast<float *>(smem + NS * STAGE);  // [9][DK]
    float *s_pk = relv_s + 9 * DK;
    // per query: the relative-key logits q_i . E_k[m] / sqrt(dk); entry m becomes the FINAL logit of key i + m - w when the
    // sweep meets that key (each entry is met once), from which the epilogue takes the relative-value weights
    float *rq_s = s_pk + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int T = a.T, Hc = a.Hc, win = a.win, nrel = 2 * win + 1;
    // workgroup id -> (utterance, head, query tile): ids 8 apart share an XCD, and with it the L2 that holds this head's K, V
    const int ntile = (T + QW - 1) / QW;
    int h, b, tile;
    {
        const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
        const int pl = k / ntile, pair = pl * 8 + xcd;
        tile = k - pl * ntile;
        if (pair >= a.nh * a.B) return;
        b = pair / a.nh;
        h = pair - b * a.nh;
    }
    ATT16_STAMP(0);
    const int L = a.len[b] < T ? a.len[b] : T;
    const int i0 = tile * QW + wave * (16 * QT);
    const bool active = i0 < L;        // (wave-uniform) a fully padded query tile only helps staging
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const float rsq = 1.f / sqrtf((float)a.dk);
    const uint16_t *plb = a.qkv_pl + (int64_t)b * 9 * Hc * T;  // batch stride: three slots of 3 Hc x T elements
    const int64_t plane_el = (int64_t)3 * Hc * T;              // elements per plane

    // ---- staging of one 32-key block: 4 NDT rows (K plane 0, K plane 1, V plane 0, V plane 1; NDT d-tiles each), one LDS-DMA
    // instruction per row: lane l carries the 16-byte cell (channel group 2 dt + (l & 1), key l >> 1)
    const uint16_t *kvb = plb + (int64_t)((Hc + h * DK) / 8) * T * 8;  // K's first cell row; V's is Hc / 8 rows further
    auto stage_issue = [&](int kb, int st) __attribute__((always_inline)) {
        const int j = kb * 32 + (lane >> 1);
        const int lane_off = ((lane & 1) * T + (j < T ? j : T - 1)) * 8;
#pragma unroll
        for (int k = 0; k < RPW; k++) {
            const int R = wave + 4 * k;                       // (uniform)
            const int isv = R >= 2 * NDT ? 1 : 0, rr = R - isv * 2 * NDT;
            const int pl = rr >= NDT ? 1 : 0, dt = rr - pl * NDT;
            const uint16_t *src = kvb + pl * plane_el + (int64_t)(isv * (Hc / 8) + 2 * dt) * T * 8 + lane_off;
            lds_dma<16>(src, reinterpret_cast<float *>(smem + st * STAGE + R * ROW));
        }
    };

    // the first blocks are requested before anything else: their latency passes under the prologue (Q, the relative-key logits)
    const int nkb = tile * QW < L ? (L + 31) / 32 : 0;  // (a fully padded workgroup only writes zeros)
#pragma unroll
    for (int p = 0; p < NS - 1; p++)
        if (p < nkb) stage_issue(p, p);

    // ---- Q fragments (B operand of K Q^T): lane (c, g) holds channels 32 s + 8 g .. + 7 of query i, both planes
    u32x4 q0[QT][DKS], q1[QT][DKS];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int i = i0 + 16 * qt + c, ic = i < T ? i : T - 1;
#pragma unroll
        for (int s = 0; s < DKS; s++) {
            const uint16_t *p = plb + ((int64_t)((h * DK) / 8 + 4 * s + g) * T + ic) * 8;
            q0[qt][s] = *reinterpret_cast<const u32x4 *>(p);
            q1[qt][s] = *reinterpret_cast<const u32x4 *>(p + plane_el);
        }
    }
    // ---- relative-key logits rq[m] = q_i . E_k[m] / sqrt(dk), m = 0 .. 2 w: E_k as a 16-row A operand (rows >= 2 w + 1 zero)
    float *rq_w = rq_s + wave * (16 * QT) * kAtt16RelPitch;
    {
        f32x4 ra[QT], rb[QT];
#pragma unroll
        for (int qt = 0; qt < QT; qt++) ra[qt] = rb[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < DKS; s++) {
            f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
            if (c < nrel) {
                const float *rp = a.relk + c * a.dk + 32 * s + 8 * g;
                x0 = *reinterpret_cast<const f32x4 *>(rp);
                x1 = *reinterpret_cast<const f32x4 *>(rp + 4);
            }
            unsigned e0[4], e1[4];
            split2h_pair(x0[0], x0[1], e0[0], e1[0]);
            split2h_pair(x0[2], x0[3], e0[1], e1[1]);
            split2h_pair(x1[0], x1[1], e0[2], e1[2]);
            split2h_pair(x1[2], x1[3], e0[3], e1[3]);
            const f16x8 ek0 = __builtin_bit_cast(f16x8, u32x4{e0[0], e0[1], e0[2], e0[3]});
            const f16x8 ek1 = __builtin_bit_cast(f16x8, u32x4{e1[0], e1[1], e1[2], e1[3]});
#pragma unroll
            for (int qt = 0; qt < QT; qt++) {
                ra[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ek0, __builtin_bit_cast(f16x8, q0[qt][s]), ra[qt], 0, 0, 0);
                rb[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ek1, __builtin_bit_cast(f16x8, q0[qt][s]), rb[qt], 0, 0, 0);
                rb[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ek0, __builtin_bit_cast(f16x8, q1[qt][s]), rb[qt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            // this lane: rows m = 4 g + r of query c; the tables are per wave, read back by the same wave only
            if (g < 3) {
                f32x4 rv;
#pragma unroll
                for (int r = 0; r < 4; r++) rv[r] = (ra[qt][r] + rb[qt][r] * (1.f / 2048.f)) * rsq;
                *reinterpret_cast<f32x4 *>(&rq_w[(16 * qt + c) * kAtt16RelPitch + 4 * g]) = rv;
            }
        }
    }
    // relative-value table -> LDS (all nine rows, zeros beyond 2 w + 1: the epilogue needs no predicate)
    for (int e = tid; e < 9 * DK; e += 256) {
        const int m = e / DK, d = e - m * DK;
        relv_s[e] = m < nrel ? a.relv[m * a.dk + d] : 0.f;
    }

    float mrun[QT], lpart[QT];  // (running maximum in the exp2 domain: logit * log2(e))
    f32x4 oacc[QT][NDT];
    const float kexp = rsq * 1.44269504f;
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        mrun[qt] = -INFINITY;
        lpart[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; dt++) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // lane addresses inside a stage: K row reads (16 bytes: channels 8 g .. of key c: d-tile g >> 1, half g & 1) and V
    // transposed reads (lane 4 q + p of group g names key 4 g + q, channels 4 p .. 4 p + 3)
    const uint32_t k_lane = (uint32_t)((g >> 1) * ROW + c * 32 + (g & 1) * 16);
    const uint32_t v_lane = (uint32_t)(KB + (4 * g + ((lane >> 2) & 3)) * 32 + (lane & 3) * 8);
    int st = 0;
    // (hazard 2 of DESIGN 5.1g: a kernel-argument s_load still in flight would let the loop's counted lgkmcnt waits pass early)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ATT16_STAMP(1);
    for (int kb = 0; kb < nkb; kb++) {
        const int j0 = kb * 32;
        {
            const int ahead = nkb - 1 - kb < NS - 2 ? nkb - 1 - kb : NS - 2;  // later blocks already in flight
            if (NS >= 4 && ahead >= 2) att16_wait_vm<2 * RPW>();
            else if (NS >= 3 && ahead == 1) att16_wait_vm<RPW>();
            else att16_wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();  // block kb is complete in LDS; everyone has finished reading block kb - 1's stage
        __builtin_amdgcn_sched_barrier(0);
        if (kb + NS - 1 < nkb) stage_issue(kb + NS - 1, st == 0 ? NS - 1 : st - 1);
        const uint32_t kst = lds0 + (uint32_t)(st * STAGE) + k_lane, vst = lds0 + (uint32_t)(st * STAGE) + v_lane;
        st = st + 1 == NS ? 0 : st + 1;
        if (!active) continue;
        // ---- S^T = K Q^T: two key tiles of 16; the reads of k-step s + 1 fly under the MFMAs of s
        f32x4 sa[QT][2], sb[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; qt++)
#pragma unroll
            for (int kt = 0; kt < 2; kt++) sa[qt][kt] = sb[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 kf[2][2][2];  // [set][kt][plane]
        kf[0][0][0] = ds_read128<0>(kst);
        kf[0][0][1] = ds_read128<NDT * ROW>(kst);
        kf[0][1][0] = ds_read128<512>(kst);
        kf[0][1][1] = ds_read128<NDT * ROW + 512>(kst);
        static_for<DKS>([&](auto S) {
            constexpr int s = decltype(S)::value, cur = s & 1, nxt = cur ^ 1;
            if constexpr (s + 1 < DKS) {
                kf[nxt][0][0] = ds_read128<(s + 1) * 2 * ROW>(kst);
                kf[nxt][0][1] = ds_read128<(s + 1) * 2 * ROW + NDT * ROW>(kst);
                kf[nxt][1][0] = ds_read128<(s + 1) * 2 * ROW + 512>(kst);
                kf[nxt][1][1] = ds_read128<(s + 1) * 2 * ROW + NDT * ROW + 512>(kst);
                att16_wait_lgkm<4>();
            } else
                att16_wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < 2; kt++)
#pragma unroll
                for (int qt = 0; qt < QT; qt++) {
                    const f16x8 k0 = __builtin_bit_cast(f16x8, kf[cur][kt][0]), k1 = __builtin_bit_cast(f16x8, kf[cur][kt][1]);
                    sa[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, __builtin_bit_cast(f16x8, q0[qt][s]), sa[qt][kt], 0, 0, 0);
                    sb[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, __builtin_bit_cast(f16x8, q0[qt][s]), sb[qt][kt], 0, 0, 0);
                    sb[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, __builtin_bit_cast(f16x8, q1[qt][s]), sb[qt][kt], 0, 0, 0);
                }
        });
        // the first V fragments fly under the softmax
        u32x2 vf[2][2][2];  // [set][plane][kt]
        vf[0][0][0] = ds_read64_tr16<0>(vst);
        vf[0][0][1] = ds_read64_tr16<512>(vst);
        vf[0][1][0] = ds_read64_tr16<NDT * ROW>(vst);
        vf[0][1][1] = ds_read64_tr16<NDT * ROW + 512>(vst);
        // ---- logits, online softmax (this lane: keys j0 + 16 kt + 4 g + r of query i), P as the B operand of V P^T: k-slot
        // e of lane (c, g) = key j0 + 16 (e >> 2) + 4 g + (e & 3): the lane's own values
        f16x8 p0[QT], p1[QT], p0s[QT];
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            const int it0 = i0 + 16 * qt;
            const bool near = (j0 + 31 >= it0 - win) && (j0 <= it0 + 15 + win);
            float sv[8];  // logits * log2(e)
#pragma unroll
            for (int kt = 0; kt < 2; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) sv[4 * kt + r] = (sa[qt][kt][r] + sb[qt][kt][r] * (1.f / 2048.f)) * kexp;
            if (near) {  // (uniform) the band |j - i| <= w crosses this block: relative-key logits in, final logits out
                const int mb = j0 - it0 + win + 4 * g - c;  // m of this lane's first key
                float *rqq = rq_w + (16 * qt + c) * kAtt16RelPitch;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const unsigned m = (unsigned)(mb + 16 * (e >> 2) + (e & 3));
                    if (m < (unsigned)nrel) {
                        sv[e] += rqq[m] * 1.44269504f;
                        rqq[m] = sv[e];
                    }
                }
            }
            if (j0 + 32 > L) {  // (uniform) the last block: keys >= len
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (j0 + 16 * (e >> 2) + 4 * g + (e & 3) >= L) sv[e] = -1e4f * 1.44269504f;
            }
            float bm = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
            bm = att16_rows_max(bm);
            const float mnew = fmaxf(mrun[qt], bm);
            const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
            mrun[qt] = mnew;
            float psum = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                sv[e] = __builtin_amdgcn_exp2f(sv[e] - mnew);
                psum += sv[e];
            }
            lpart[qt] = lpart[qt] * alpha + psum;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {  // (uniform) the running maximum moved for some query
#pragma unroll
                for (int dt = 0; dt < NDT; dt++) oacc[qt][dt] *= alpha;
            }
            unsigned ph0[4], ph1[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                ph0[e] = cvt_pk_f16(sv[2 * e], sv[2 * e + 1]);
                const f16x2 hh = __builtin_bit_cast(f16x2, ph0[e]);
                ph1[e] = cvt_pk_f16(sv[2 * e] - (float)hh[0], sv[2 * e + 1] - (float)hh[1]);  // (unscaled low part: p <= 1)
            }
            p0[qt] = __builtin_bit_cast(f16x8, u32x4{ph0[0], ph0[1], ph0[2], ph0[3]});
            p1[qt] = __builtin_bit_cast(f16x8, u32x4{ph1[0], ph1[1], ph1[2], ph1[3]});
            p0s[qt] = p0[qt] * (_Float16)0.00048828125f;  // meets V's low plane, which is stored 2^11 up
        }
        // ---- O^T += V P^T, d-tile by d-tile; the reads of d-tile dt + 1 fly under the MFMAs of dt
        static_for<NDT>([&](auto D) {
            constexpr int dt = decltype(D)::value, cur = dt & 1, nxt = cur ^ 1;
            if constexpr (dt + 1 < NDT) {
                vf[nxt][0][0] = ds_read64_tr16<(dt + 1) * ROW>(vst);
                vf[nxt][0][1] = ds_read64_tr16<(dt + 1) * ROW + 512>(vst);
                vf[nxt][1][0] = ds_read64_tr16<(dt + 1) * ROW + NDT * ROW>(vst);
                vf[nxt][1][1] = ds_read64_tr16<(dt + 1) * ROW + NDT * ROW + 512>(vst);
                att16_wait_lgkm<4>();
            } else
                att16_wait_lgkm<0>();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned a0 = vf[cur][0][0].x, a1 = vf[cur][0][0].y, a2 = vf[cur][0][1].x, a3 = vf[cur][0][1].y;
            const unsigned b0 = vf[cur][1][0].x, b1 = vf[cur][1][0].y, b2 = vf[cur][1][1].x, b3 = vf[cur][1][1].y;
            const f16x8 v0 = __builtin_bit_cast(f16x8, u32x4{a0, a1, a2, a3});
            const f16x8 v1 = __builtin_bit_cast(f16x8, u32x4{b0, b1, b2, b3});
#pragma unroll
            for (int qt = 0; qt < QT; qt++) {
                oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, p0s[qt], oacc[qt][dt], 0, 0, 0);
                oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, p1[qt], oacc[qt][dt], 0, 0, 0);
                oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, p0[qt], oacc[qt][dt], 0, 0, 0);
            }
        });
    }
    ATT16_STAMP(2);
    __syncthreads();  // (relv_s is complete)

    // ---- epilogue: O^T lane (c, g), register r = channel 16 dt + 4 g + r of query i
    float *ob = a.out + ((int64_t)b * Hc + (int64_t)h * DK) * T;
    uint16_t *pb = a.out_pl ? a.out_pl + (int64_t)b * 3 * Hc * T : nullptr;
    const int64_t oplane = (int64_t)Hc * T;
    float pk = 0.f;
    float lsums[QT];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) lsums[qt] = att16_rows_sum(lpart[qt]);  // (whole wave: row swaps)
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int i = i0 + 16 * qt + c;
        if (i >= T) continue;
        if (i >= L) {
#pragma unroll
            for (int dt = 0; dt < NDT; dt++) {
                const int d0 = dt * 16 + 4 * g;
                if (a.out)
#pragma unroll
                    for (int r = 0; r < 4; r++) ob[(int64_t)(d0 + r) * T + i] = 0.f;
                if (pb) {
                    const int64_t cell = ((int64_t)((h * DK + d0) >> 3) * T + i) * 8 + (d0 & 4);
                    *reinterpret_cast<u32x2 *>(pb + cell) = u32x2{0u, 0u};
                    *reinterpret_cast<u32x2 *>(pb + oplane + cell) = u32x2{0u, 0u};
                }
            }
            continue;
        }
        const float rl = 1.f / lsums[qt];
        // relative-value weights: p of the keys i - w .. i + w that exist (keys >= len have p = 0 exactly: their logit is -1e4)
        float wrel[9];
        {
            const float *bq = rq_w + (16 * qt + c) * kAtt16RelPitch;
#pragma unroll
            for (int m = 0; m < 9; m++) {
                const int j = i + m - win;
                wrel[m] = (m < nrel && j >= 0 && j < L) ? __builtin_amdgcn_exp2f(bq[m] - mrun[qt]) : 0.f;
            }
        }
#pragma unroll
        for (int dt = 0; dt < NDT; dt++) {
            const int d0 = dt * 16 + 4 * g;
            f32x4 val = oacc[qt][dt];
#pragma unroll
            for (int m = 0; m < 9; m++) val += wrel[m] * *reinterpret_cast<const f32x4 *>(&relv_s[m * DK + d0]);  // (w[m >= nrel] = 0)
            val *= rl;
            if (a.out)
#pragma unroll
                for (int r = 0; r < 4; r++) ob[(int64_t)(d0 + r) * T + i] = val[r];
            if (pb) {
                unsigned wa[2], wb[2];
                split2h_pair_pk(val[0], val[1], wa[0], wa[1], pk);
                split2h_pair_pk(val[2], val[3], wb[0], wb[1], pk);
                const int64_t cell = ((int64_t)((h * DK + d0) >> 3) * T + i) * 8 + (d0 & 4);
                *reinterpret_cast<u32x2 *>(pb + cell) = u32x2{wa[0], wb[0]};
                *reinterpret_cast<u32x2 *>(pb + oplane + cell) = u32x2{wa[1], wb[1]};
            }
        }
    }
    if (a.peak) sx_publish_peak_at(a.peak, (int)blockIdx.x, pk, s_pk);  // (all threads arrive)
    ATT16_STAMP(3);
}

}  // namespace vitsmi
