// kernels.hip.hpp — the non-conv kernels of the VITS pipeline (gfx950, wave64).
// Activations are [B, C, T] fp32 with time contiguous, so one lane per time step gives
// coalesced rows; channel reductions (LayerNorm) loop over C per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "conv_engine.hip.hpp"

namespace vitsmi {

// erf without branches (both ranges evaluated, one select; < 1 ulp against double erf over the whole range): the library
// erff branches on |x|, and a branch per element splits an unrolled loop into basic blocks - the loads of each
// element's parameters then wait for their own round trip instead of travelling together
__device__ __forceinline__ float erf_nb(float a) {
    const float t = __builtin_fabsf(a), s = a * a;
    float r = __builtin_fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = __builtin_fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = __builtin_fmaf(r, s, u);
    r = __builtin_fmaf(r, t, -1.06777877e-1f);
    r = __builtin_fmaf(r, t, -6.34846687e-1f);
    r = __builtin_fmaf(r, t, -1.28717512e-1f);
    r = __builtin_fmaf(r, t, -t);
    r = 1.0f - __expf(r);  // (r <= 0: exp2 of a scaled argument, no range cases)
    r = __builtin_copysignf(r, a);
    float q = -5.96761703e-4f;
    q = __builtin_fmaf(q, s, 4.99119423e-3f);
    q = __builtin_fmaf(q, s, -2.67681349e-2f);
    q = __builtin_fmaf(q, s, 1.12819925e-1f);
    q = __builtin_fmaf(q, s, -3.76125336e-1f);
    q = __builtin_fmaf(q, s, 1.28379166e-1f);
    q = __builtin_fmaf(q, a, a);
    return t > 0.927734375f ? r : q;
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_nb(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float softplus_f(float v) { return v > 20.0f ? v : log1pf(expf(v)); }

// ---- lengths: int64 -> int32 (clamped to [0,T]) ---------------------------------------------
__global__ void lens_to_i32(const int64_t *in, int *out, int B, int T) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        int64_t v = in[b];
        out[b] = v < 0 ? 0 : (v > T ? T : (int)v);
    }
}

// ---- a1: embedding gather * sqrt(H), transposed to [B,H,T], masked (models.py:199-205) -------
// grid (T / 64, H / 16, B): a thread writes 16 channels of one token (a loop over all H channels per thread is one
// long chain of dependent round trips on very few waves)
__global__ void embed_kernel(const int64_t *ids, const int *len, const float *emb, float *x, int H, int T,
                             int n_vocab, float scale) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.z;
    if (t >= T) return;
    int64_t id = ids[(int64_t)b * T + t];
    bool ok = t < len[b] && id >= 0 && id < n_vocab;
    const float *row = emb + id * H;
    float *o = x + (int64_t)b * H * T + t;
    const int c0 = blockIdx.y * 16, c1 = c0 + 16 < H ? c0 + 16 : H;
    for (int c = c0; c < c1; c++) o[(int64_t)c * T] = ok ? row[c] * scale : 0.f;
}

// ---- LayerNorm over channels (modules.py:14-26), options: GELU, accumulate, mask ------------
enum : int { LN_GELU = 1, LN_ACCUM = 2, LN_MASK = 4, LN_RELU_IN = 8 };
__global__ void layernorm_c_kernel(const float *in, float *out, const float *gamma, const float *beta,
                                   const int *len, int C, int T, int flags) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= T) return;
    const float *p = in + (int64_t)b * C * T + t;
    float *o = out + (int64_t)b * C * T + t;
    const bool relu_in = flags & LN_RELU_IN;
    float mean = 0.f;
    for (int c = 0; c < C; c++) {
        float v = p[(int64_t)c * T];
        if (relu_in) v = v > 0.f ? v : 0.f;
        mean += v;
    }
    mean /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; c++) {
        float v = p[(int64_t)c * T];
        if (relu_in) v = v > 0.f ? v : 0.f;
        float d = v - mean;
        var += d * d;
    }
    var /= (float)C;
    float rs = 1.0f / sqrtf(var + 1e-5f);
    float mk = (!(flags & LN_MASK) || t < len[b]) ? 1.f : 0.f;
    for (int c = 0; c < C; c++) {
        float v = p[(int64_t)c * T];
        if (relu_in) v = v > 0.f ? v : 0.f;
        float y = (v - mean) * rs * gamma[c] + beta[c];
        if (flags & LN_GELU) y = gelu_erf(y);
        if (flags & LN_ACCUM) y += o[(int64_t)c * T];
        o[(int64_t)c * T] = y * mk;
    }
}

// ---- a6: depthwise conv (groups=C) on x*mask -> LayerNorm -> GELU (modules.py:121-123) --------
__global__ void dds_dw_ln_gelu_kernel(const float *x, float *y, const float *w, const float *bias,
                                      const float *gamma, const float *beta, const int *len, int C, int T, int K,
                                      int dil) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= T) return;
    const int L = len[b];
    const float *p = x + (int64_t)b * C * T;
    float *o = y + (int64_t)b * C * T + t;
    const int pad = (K * dil - dil) / 2;
    auto conv = [&](int c) {
        float s = bias[c];
        for (int k = 0; k < K; k++) {
            int tt = t + k * dil - pad;
            float v = (tt >= 0 && tt < T && tt < L) ? p[(int64_t)c * T + tt] : 0.f;
            s += w[c * K + k] * v;
        }
        return s;
    };
    float mean = 0.f;
    for (int c = 0; c < C; c++) mean += conv(c);
    mean /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; c++) {
        float d = conv(c) - mean;
        var += d * d;
    }
    var /= (float)C;
    float rs = 1.0f / sqrtf(var + 1e-5f);
    for (int c = 0; c < C; c++) o[(int64_t)c * T] = gelu_erf((conv(c) - mean) * rs * gamma[c] + beta[c]);
}

// ---- tiled LayerNorm family (C <= 256): one workgroup = 32 time steps x 8 channel groups.
// Thread (tl = tid&31, cg = tid>>5) owns channels cg, cg+8, ... of time step t0+tl, keeps them in
// registers (one global read, one write), and the 8 partial sums per time step meet in LDS.  A wave reads
// two 128-byte rows per instruction, and B*T/32 workgroups fill the chip where the one-lane-per-(b,t)
// kernels above ran on 128 wavefronts.  DW = 1 computes the DDSConv depthwise conv of x*mask on the fly
// (modules.py:121-122) instead of reading `in` directly.
// PL: the result is written once more as the two fp16 operand planes of the split-operand conv engine ([C/8][T][8] cells
// per plane, batch stride 3 planes; conv_sx_engine.hip.hpp), transposed through LDS so that every cell leaves as one
// 16-byte store; `peak` = that engine's range-guard slots (C % 8 == 0).
// TS: time steps per workgroup (32, or 16: twice the workgroups with half the loads per thread - the plain LayerNorms of the
// encoder at batch 32 x 256 tokens are 256 workgroups of pure latency at TS = 32: 11.4 -> see DESIGN 5.3)
template <int DW, bool PL = false, int TS = 32>
__global__ __launch_bounds__(256) void ln_tile_kernel(const float *in, float *out, const float *gamma,
                                                      const float *beta, const int *len, int C, int T, int flags,
                                                      const float *dw_w, const float *dw_b, int K, int dil,
                                                      uint16_t *planes = nullptr, unsigned *peak = nullptr) {
    static_assert(TS == 32 || TS == 16, "time steps per workgroup");
    constexpr int NG = 256 / TS;  // channel groups: thread (tl, cg) owns channels cg, cg + NG, ..
    __shared__ float red[NG][TS];
    __shared__ __attribute__((aligned(16))) uint16_t cells[PL ? 2 : 1][PL ? 32 * TS * 8 : 8];  // [plane][channel group of 8][t][8]
    const int tid = threadIdx.x, tl = tid & (TS - 1), cg = tid / TS;
    const int t = blockIdx.x * TS + tl, b = blockIdx.y;
    const int L = len ? len[b] : T;
    const bool tv = t < T;
    const float *p = in + (int64_t)b * C * T;
    float *o = out + (int64_t)b * C * T;
    constexpr int CPT = 256 / NG;  // channels per thread, C <= 256
    float v[CPT];
    const int pad = (K * dil - dil) / 2;
    float s = 0.f;
    // Every per-channel operand is requested up front with a clamped (always valid) address and selected afterwards: a
    // predicate per element makes each load its own basic block, i.e. one exposed round trip per channel (the plain
    // LayerNorm spent 20 us on 32 of them).  gv / bv: gamma, beta; av: the LN_ACCUM operand.
    float gv[CPT], bv[CPT], av[CPT];
    const int tcl = tv ? t : T - 1;
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int c = cg + NG * i, cc = c < C ? c : cg;
        gv[i] = gamma[cc];
        bv[i] = beta[cc];
        if (DW == 0) v[i] = p[(int64_t)cc * T + tcl];
    }
    if (flags & LN_ACCUM) {  // (uniform; `in` and `out` may be the same tensor: read before anything is stored)
#pragma unroll
        for (int i = 0; i < CPT; i++) av[i] = o[(int64_t)(cg + NG * i < C ? cg + NG * i : cg) * T + tcl];
    } else {
#pragma unroll
        for (int i = 0; i < CPT; i++) av[i] = 0.f;
    }
    const float relu_floor = (flags & LN_RELU_IN) ? 0.f : -__builtin_inff();
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int c = cg + NG * i;
        float x = 0.f;
        if (DW == 0) {
            x = (c < C && tv) ? fmaxf(v[i], relu_floor) : 0.f;
        } else if (c < C && tv) {
            if (DW > 1) {
                // tap count known at compile time (DW = K; 3 in every DDSConv): the loads of all taps and channels are
                // in flight together instead of one round trip per (channel, tap)
                x = dw_b[c];
#pragma unroll
                for (int k = 0; k < DW; k++) {
                    const int tt = t + k * dil - pad;
                    const float xv = (tt >= 0 && tt < T && tt < L) ? p[(int64_t)c * T + tt] : 0.f;
                    x += dw_w[c * DW + k] * xv;
                }
            } else if (DW) {
                x = dw_b[c];
                for (int k = 0; k < K; k++) {
                    const int tt = t + k * dil - pad;
                    const float xv = (tt >= 0 && tt < T && tt < L) ? p[(int64_t)c * T + tt] : 0.f;
                    x += dw_w[c * K + k] * xv;
                }
            }
        }
        v[i] = x;
        s += x;
    }
    red[cg][tl] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int g = 0; g < NG; g++) mean += red[g][tl];
    mean /= (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int c = cg + NG * i;
        const float d = (c < C) ? v[i] - mean : 0.f;
        q += d * d;
    }
    red[cg][tl] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int g = 0; g < NG; g++) var += red[g][tl];
    var /= (float)C;
    const float rs = 1.0f / sqrtf(var + 1e-5f);
    const float mk = (!(flags & LN_MASK) || t < L) ? 1.f : 0.f;
    if (!PL && !tv) return;
    float pk = 0.f;
    float yv[CPT];
    if (flags & LN_GELU) {  // (uniform, outside the element loop)
#pragma unroll
        for (int i = 0; i < CPT; i++) yv[i] = (gelu_erf((v[i] - mean) * rs * gv[i] + bv[i]) + av[i]) * mk;
    } else {
#pragma unroll
        for (int i = 0; i < CPT; i++) yv[i] = ((v[i] - mean) * rs * gv[i] + bv[i] + av[i]) * mk;
    }
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int c = cg + NG * i;
        if (c < C && tv) {
            const float y = yv[i];
            o[(int64_t)c * T + t] = y;
            if constexpr (PL) {
                pk = !(__builtin_fabsf(y) <= kF16Max) ? __builtin_inff() : __builtin_fmaxf(pk, __builtin_fabsf(y));
                const float yc = __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);
                const _Float16 h0 = (_Float16)yc, h1 = (_Float16)((yc - (float)h0) * 2048.f);
                cells[0][((c >> 3) * TS + tl) * 8 + (c & 7)] = __builtin_bit_cast(unsigned short, h0);
                cells[1][((c >> 3) * TS + tl) * 8 + (c & 7)] = __builtin_bit_cast(unsigned short, h1);
            }
        }
    }
    if constexpr (PL) {
        __syncthreads();
        const int CG = C >> 3;
        uint16_t *pb = planes + (int64_t)b * 3 * CG * T * 8;
        const int tb = blockIdx.x * TS;
        for (int cell = tid; cell < CG * TS; cell += 256) {
            const int g = cell / TS, tt = tb + (cell & (TS - 1));
            if (tt < T) {
#pragma unroll
                for (int pl = 0; pl < 2; pl++)
                    *reinterpret_cast<u32x4 *>(pb + (((int64_t)pl * CG + g) * T + tt) * 8) =
                        *reinterpret_cast<const u32x4 *>(&cells[pl][cell * 8]);
            }
        }
        if (peak) sx_publish_peak(peak, (int)(blockIdx.x + blockIdx.y), pk);
    }
}

// ---- a6 fused: one whole DDSConv layer per launch (modules.py:117-129):
//     y = GELU(LN1(dwconv_k3,dil(x * mask)));  y = GELU(LN2(conv1x1(y)));  out = (x + y) [* mask]
// As three launches (depthwise + LN, 1x1 conv on the conv engine, LN + accumulate) a layer costs ~85 us at batch 32 x 256
// tokens, nearly all of it launch / drain latency on 256 small workgroups; here one workgroup takes 32 time steps
// through all three stages: the depthwise + LN1 result goes to LDS as the B operand of the 1x1 conv, which runs on
// v_mfma_f32_32x32x2_f32 with the packed conv-engine weights read straight from L2 (each lane's float4 = four k-steps
// of its row), and LN2 reduces the accumulators over channels through LDS.  C % 32 == 0, C <= 256, K = 3.
// `in` and `out` must be different buffers (neighbouring workgroups read `in` in their halo).
struct DdsLayerArgs {
    const float *in;
    float *out;
    const int *len;
    const float *dw_w, *dw_b, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const float *pw;       // packed 1x1 weights (pack_conv layout), pw_bias [C]
    const float *pw_bias;
    int C, T, dil, mask_out;
    int CK, nchunks, MB;   // packing geometry of pw: chunk depth, chunks, 32-row blocks per m-tile
};

// NBLK = C / 32 at compile time: every channel loop is straight-line code (the run-time form branched around each
// depthwise tap, each LDS read and each group of four MFMAs: every MFMA waited for its own predicated ds_read_b32, and
// the 1x1 conv ran at a third of the matrix pipe's issue rate; 55 -> see DESIGN §5.3).
template <int NBLK>
__global__ __launch_bounds__(256) void dds_layer_kernel(DdsLayerArgs a) {
    typedef float f32x16_ __attribute__((ext_vector_type(16)));
    constexpr int C = NBLK * 32;
    __shared__ float y1[C * 32];   // [ci][tl]: B operand of the 1x1 conv
    __shared__ float red[8][32];
    __shared__ float stat[2][32];
    const int tid = threadIdx.x, tl = tid & 31, cg = tid >> 5;
    const int lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
    const int t0 = blockIdx.x * 32, t = t0 + tl, b = blockIdx.y;
    const int T = a.T;
    const int L = a.len ? a.len[b] : T;
    const bool tv = t < T;
    const float *p = a.in + (int64_t)b * C * T;
    float *o = a.out + (int64_t)b * C * T;
    // packed 1x1 weights: float4 group gi of block mb = k-steps 4 gi .. 4 gi + 3 (two input channels each) of its 32 rows
    const int spc = a.CK >> 3;           // float4 groups per (block, chunk)
    constexpr int NG = C >> 3;           // groups per block
    constexpr int MAXB = (NBLK + 3) / 4; // blocks per wave
    // (all of this wave's blocks are requested here, ahead of stage 1: their round trip is hidden behind it)
    float4 wa[MAXB][NG];
#pragma unroll
    for (int j = 0; j < MAXB; j++) {
        const int mb = wave + 4 * j;
        if (mb < NBLK) {
#pragma unroll
            for (int gi = 0; gi < NG; gi++) {
                const int ch = gi / spc, g = gi - ch * spc;
                wa[j][gi] = reinterpret_cast<const float4 *>(a.pw)[(((int64_t)((mb / a.MB) * a.nchunks + ch) * a.MB + (mb % a.MB)) * spc + g) * 64 + lane];
            }
        }
    }
    // per-channel parameters of all three stages: one coalesced batch into LDS (read per element from global memory, inside
    // loops the GELU used to split into basic blocks, every element waited for its own round trip: ~25 of the layer's 50 us)
    __shared__ float prm[9][C];  // dw_b, dw_w tap 0..2, ln1_g, ln1_b, pw_bias, ln2_g, ln2_b
    for (int e = tid; e < C; e += 256) {
        prm[0][e] = a.dw_b[e];
        prm[1][e] = a.dw_w[e * 3];
        prm[2][e] = a.dw_w[e * 3 + 1];
        prm[3][e] = a.dw_w[e * 3 + 2];
        prm[4][e] = a.ln1_g[e];
        prm[5][e] = a.ln1_b[e];
        prm[6][e] = a.pw_bias[e];
        prm[7][e] = a.ln2_g[e];
        prm[8][e] = a.ln2_b[e];
    }
    // ---- stage 1: depthwise conv (k = 3) of x * mask, LayerNorm over channels, GELU -> y1 (thread: channels cg + 8 i)
    constexpr int CPT = C / 8;
    {
        float v[CPT];
        const int pad = a.dil;  // (3 * dil - dil) / 2
        float s = 0.f;
        // the three taps' validity does not depend on the channel
        bool ok[3];
        int tt[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            tt[k] = t + k * a.dil - pad;
            ok[k] = tv && tt[k] >= 0 && tt[k] < T && tt[k] < L;
            tt[k] = ok[k] ? tt[k] : 0;
        }
        // all 3 * CPT loads travel together (always inside the row; masked taps are zeroed below) ...
        float xr[CPT][3];
#pragma unroll
        for (int i = 0; i < CPT; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) xr[i][k] = p[(int64_t)(cg + 8 * i) * T + tt[k]];
        __builtin_amdgcn_sched_barrier(0);  // ... (the scheduler otherwise sinks each load to its use, three in flight at a time)
        __syncthreads();                    // prm is in LDS
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = cg + 8 * i;
            float x = prm[0][c];
#pragma unroll
            for (int k = 0; k < 3; k++) x += prm[1 + k][c] * (ok[k] ? xr[i][k] : 0.f);
            v[i] = tv ? x : 0.f;
            s += v[i];
        }
        red[cg][tl] = s;
        __syncthreads();
        float mean = 0.f;
#pragma unroll
        for (int g = 0; g < 8; g++) mean += red[g][tl];
        mean /= (float)C;
        __syncthreads();
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const float d = v[i] - mean;
            q += d * d;
        }
        red[cg][tl] = q;
        __syncthreads();
        float var = 0.f;
#pragma unroll
        for (int g = 0; g < 8; g++) var += red[g][tl];
        var /= (float)C;
        const float rs = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = cg + 8 * i;
            y1[c * 32 + tl] = tv ? gelu_erf((v[i] - mean) * rs * prm[4][c] + prm[5][c]) : 0.f;
        }
    }
    __syncthreads();
    // ---- stage 2: 1x1 conv on the matrix cores.  Block rows (32 output channels each) are dealt to the waves round-robin.
    // Group gi covers input channels 8 gi .. 8 gi + 7 (chunks are consecutive): this lane's B values of its four k-steps
    // sit at compile-time offsets from one base, so the reads of a whole block are plain loads the compiler batches ahead
    // of the MFMA chain.
    // (stage 3's column and its residual operands, requested ahead of the MFMA chain)
    const int col = lane & 31;
    const int tc = t0 + col;
    const bool tcv = tc < T;
    float resv[MAXB][16];
#pragma unroll
    for (int j = 0; j < MAXB; j++) {
        const int mb = wave + 4 * j;
        if (mb < NBLK) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int c = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                resv[j][r] = p[(int64_t)c * T + (tcv ? tc : 0)];
            }
        }
    }
    f32x16_ acc[MAXB];
#pragma unroll
    for (int j = 0; j < MAXB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    const float *yb = y1 + hi * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < MAXB; j++) {
        const int mb = wave + 4 * j;
        if (mb < NBLK) {  // (uniform per wave)
#pragma unroll
            for (int gi = 0; gi < NG; gi++) {
                const float b0 = yb[(8 * gi + 0) * 32], b1 = yb[(8 * gi + 2) * 32], b2 = yb[(8 * gi + 4) * 32], b3 = yb[(8 * gi + 6) * 32];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j][gi].x, b0, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j][gi].y, b1, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j][gi].z, b2, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j][gi].w, b3, acc[j], 0, 0, 0);
            }
        }
    }
    // ---- stage 3: + bias, LayerNorm over channels (column = lane & 31, rows spread over registers, `hi`, blocks, waves),
    // GELU, residual, mask.  C/D layout: row = (r & 3) + 8 * (r >> 2) + 4 * hi, col = lane & 31.
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXB; j++) {
        const int mb = wave + 4 * j;
        if (mb < NBLK) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                acc[j][r] += prm[6][mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];
                s += acc[j][r];
            }
        }
    }
    s += __shfl_xor(s, 32, 64);
    __syncthreads();  // (red is free again)
    if (hi == 0) red[wave][col] = s;
    __syncthreads();
    if (tid < 32) stat[0][tid] = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) / (float)C;
    __syncthreads();
    const float mean = stat[0][col];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXB; j++) {
        const int mb = wave + 4 * j;
        if (mb < NBLK) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float d = acc[j][r] - mean;
                q += d * d;
            }
        }
    }
    q += __shfl_xor(q, 32, 64);
    if (hi == 0) red[4 + wave][col] = q;
    __syncthreads();
    if (tid < 32) stat[1][tid] = 1.0f / sqrtf((red[4][tid] + red[5][tid] + red[6][tid] + red[7][tid]) / (float)C + 1e-5f);
    __syncthreads();
    const float rs = stat[1][col];
    if (!tcv) return;
    const float mk = (!a.mask_out || tc < L) ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < MAXB; j++) {
        const int mb = wave + 4 * j;
        if (mb < NBLK) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int c = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float y = gelu_erf((acc[j][r] - mean) * rs * prm[7][c] + prm[8][c]);
                o[(int64_t)c * T + tc] = (resv[j][r] + y) * mk;
            }
        }
    }
}

// ---- a6, 16 columns per workgroup (round 4).  dds_layer_kernel above takes 32 time steps per workgroup because its matrix
// instruction (32x32x2) has 32 columns: a 256-token utterance is 8 workgroups, each with a chain of 192 dependent 64-cycle MFMAs
// on its busiest waves (6 row blocks over 4 waves: 2, 2, 1, 1) behind a stage 1 of 72 loads and 24 GELUs per thread - 35 k cycles
// per layer, 12 layers per utterance.  Here: v_mfma_f32_16x16x4_f32 (the same exact fp32 FMA chains, 16 columns), twice the
// workgroups, C / 16 row tiles dealt evenly (192 channels: 3 per wave, 144 dependent 32-cycle MFMAs), half the stage-1 work per
// thread.  Weights: DDSDesc::L::pw16 (model.cpp: lane (row & 15, k & 3) reads float4 = four consecutive k-steps of its row).
struct DdsLayer16Args {
    const float *in;
    float *out;
    const int *len;
    const float *dw_w, *dw_b, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const float *pw16, *pw_bias;
    int T, dil, mask_out;
    // optional tail (the LAST layer of a stack): the 1 x 1 conv that follows the stack - C -> tail_rows, weights in the pw16
    // layout with the rows padded to 16 - applied to the layer's result while it is in LDS: tail_out[b][row][t] = (W out + bias)
    // [* (t < len)]; `out` itself is then not stored.  (models.py:69, modules.py:500: proj(h) * x_mask)
    const float *tail_w16, *tail_b;
    float *tail_out;
    int tail_rows, tail_mask;
    // optional head (the FIRST layer of a ConvFlow's stack): the layer's input is h = head_w[c] * z[b][t] + head_b[c] + in[b][c][t]
    // (ConvFlow.pre, a 1 -> C conv of one channel of z, plus the conditioning tensor `in`: modules.py:498-499) formed while
    // loading instead of read from a tensor a kernel of its own wrote.  head_z = that channel's row of utterance 0, rows of the
    // utterances head_zstride apart.
    const float *head_z, *head_w, *head_b;
    int64_t head_zstride;
};

template <int NBLK>  // C / 32
__global__ __launch_bounds__(256) void dds_layer16_kernel(DdsLayer16Args a) {
    constexpr int C = NBLK * 32, NRT = C / 16, NS = C / 4, NS4 = NS / 4;  // row tiles, k-steps, float4 groups of steps per row
    constexpr int MAXR = (NRT + 3) / 4;                                     // row tiles per wave
    constexpr int CPT = C / 16;                                            // stage 1: channels per thread
    __shared__ float y1[C * 16];   // [ci][tl]: B operand of the 1x1 conv
    __shared__ float red[16][16];
    __shared__ float stat[2][16];
    __shared__ float prm[11][C];   // dw_b, dw_w tap 0..2, ln1_g, ln1_b, pw_bias, ln2_g, ln2_b, head_w, head_b
    const int tid = threadIdx.x, tl = tid & 15, cg = tid >> 4;
    const int lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    const bool head = a.head_z != nullptr;  // (uniform)
    const int t0 = blockIdx.x * 16, t = t0 + tl, b = blockIdx.y;
    const int T = a.T;
    const int L = a.len ? a.len[b] : T;
    const bool tv = t < T;
    const float *p = a.in + (int64_t)b * C * T;
    float *o = a.out + (int64_t)b * C * T;
    for (int e = tid; e < C; e += 256) {
        prm[0][e] = a.dw_b[e];
        prm[1][e] = a.dw_w[e * 3];
        prm[2][e] = a.dw_w[e * 3 + 1];
        prm[3][e] = a.dw_w[e * 3 + 2];
        prm[4][e] = a.ln1_g[e];
        prm[5][e] = a.ln1_b[e];
        prm[6][e] = a.pw_bias[e];
        prm[7][e] = a.ln2_g[e];
        prm[8][e] = a.ln2_b[e];
        prm[9][e] = head ? a.head_w[e] : 0.f;
        prm[10][e] = head ? a.head_b[e] : 0.f;
    }
    // This wave's first row tile of weights and the residual operands of all its tiles depend on nothing: they are requested
    // right behind stage 1's own operands (vector loads return in order: stage 1 does not wait for them) instead of behind
    // stage 1, where their round trip to L2 was exposed - at batch 1 a layer is 16 workgroups with nothing else to run.
    // (All three tiles' weights up front with their MFMA chains interleaved measured SLOWER: 1.39 vs 1.36 ms per call.)
    const float4 *wp = reinterpret_cast<const float4 *>(a.pw16);
    float4 wa[2][NS4];
    auto load_w = [&](int set, int rt) {
#pragma unroll
        for (int i = 0; i < NS4; i++) wa[set][i] = wp[((int64_t)(rt * 4 + kq) * 16 + col) * NS4 + i];
    };
    const int tc = t0 + col;
    const bool tcv = tc < T;
    float resv[MAXR][4];
    float zres = 0.f;
    auto request_stage2_operands = [&]() __attribute__((always_inline)) {
        load_w(0, wave < NRT ? wave : NRT - 1);
#pragma unroll
        for (int j = 0; j < MAXR; j++) {
            const int rt = wave + 4 * j < NRT ? wave + 4 * j : NRT - 1;  // (a wave without a j-th tile re-reads the last one, unused)
#pragma unroll
            for (int r = 0; r < 4; r++) resv[j][r] = p[(int64_t)(rt * 16 + 4 * kq + r) * T + (tcv ? tc : 0)];
        }
        if (head) zres = a.head_z[(int64_t)b * a.head_zstride + (tcv ? tc : 0)];
    };
    // ---- stage 1: depthwise conv (k = 3) of x * mask, LayerNorm over channels, GELU -> y1 (thread: channels cg + 16 i)
    {
        float v[CPT];
        const int pad = a.dil;
        float s = 0.f;
        bool ok[3];
        int tt[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            tt[k] = t + k * a.dil - pad;
            ok[k] = tv && tt[k] >= 0 && tt[k] < T && tt[k] < L;
            tt[k] = ok[k] ? tt[k] : 0;
        }
        float xr[CPT][3], zt[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CPT; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) xr[i][k] = p[(int64_t)(cg + 16 * i) * T + tt[k]];
        if (head) {
#pragma unroll
            for (int k = 0; k < 3; k++) zt[k] = a.head_z[(int64_t)b * a.head_zstride + tt[k]];
        }
        __builtin_amdgcn_sched_barrier(0);
        request_stage2_operands();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // prm is in LDS
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = cg + 16 * i;
            float x = prm[0][c];
#pragma unroll
            for (int k = 0; k < 3; k++) x += prm[1 + k][c] * (ok[k] ? xr[i][k] + (prm[9][c] * zt[k] + prm[10][c]) : 0.f);
            v[i] = tv ? x : 0.f;
            s += v[i];
        }
        red[cg][tl] = s;
        __syncthreads();
        float mean = 0.f;
#pragma unroll
        for (int g = 0; g < 16; g++) mean += red[g][tl];
        mean /= (float)C;
        __syncthreads();
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const float d = v[i] - mean;
            q += d * d;
        }
        red[cg][tl] = q;
        __syncthreads();
        float var = 0.f;
#pragma unroll
        for (int g = 0; g < 16; g++) var += red[g][tl];
        var /= (float)C;
        const float rs = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = cg + 16 * i;
            y1[c * 16 + tl] = tv ? gelu_erf((v[i] - mean) * rs * prm[4][c] + prm[5][c]) : 0.f;
        }
    }
    __syncthreads();
    if (head) {  // (prm is visible since stage 1's first barrier)
#pragma unroll
        for (int j = 0; j < MAXR; j++) {
            const int rt = wave + 4 * j;
            if (rt < NRT) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int c = rt * 16 + 4 * kq + r;
                    resv[j][r] += prm[9][c] * zres + prm[10][c];
                }
            }
        }
    }
    // ---- stage 2: 1x1 conv.  A lane (row & 15 = col id of A, k & 3 = kq) x B lane (column col, k & 3 = kq); C/D: column col,
    // rows 4 kq + r of the tile.  The next tile's weights are requested before this tile's chain starts.
    f32x4 acc[MAXR];
    const float *yb = y1 + kq * 16 + col;
#pragma unroll
    for (int j = 0; j < MAXR; j++) {
        const int rt = wave + 4 * j;
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rt < NRT) {  // (uniform per wave)
            if (j + 1 < MAXR && rt + 4 < NRT) load_w((j + 1) & 1, rt + 4);
#pragma unroll
            for (int i = 0; i < NS4; i++) {
                const float4 w = wa[j & 1][i];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, yb[(16 * i + 0) * 16], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, yb[(16 * i + 4) * 16], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, yb[(16 * i + 8) * 16], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, yb[(16 * i + 12) * 16], acc[j], 0, 0, 0);
            }
        }
    }
    // ---- stage 3: + bias, LayerNorm over channels (this lane: column col, rows 4 kq + r of its tiles), GELU, residual, mask
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXR; j++) {
        const int rt = wave + 4 * j;
        if (rt < NRT) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[j][r] += prm[6][rt * 16 + 4 * kq + r];
                s += acc[j][r];
            }
        }
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    __syncthreads();  // (red is free again)
    if (kq == 0) red[wave][col] = s;
    __syncthreads();
    if (tid < 16) stat[0][tid] = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) / (float)C;
    __syncthreads();
    const float mean = stat[0][col];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXR; j++) {
        const int rt = wave + 4 * j;
        if (rt < NRT) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float d = acc[j][r] - mean;
                q += d * d;
            }
        }
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (kq == 0) red[4 + wave][col] = q;
    __syncthreads();
    if (tid < 16) stat[1][tid] = 1.0f / sqrtf((red[4][tid] + red[5][tid] + red[6][tid] + red[7][tid]) / (float)C + 1e-5f);
    __syncthreads();
    const float rs = stat[1][col];
    const float mk = (!a.mask_out || tc < L) ? 1.f : 0.f;
    const bool tail = a.tail_w16 != nullptr;  // (uniform)
    if (!tail && !tcv) return;
#pragma unroll
    for (int j = 0; j < MAXR; j++) {
        const int rt = wave + 4 * j;
        if (rt < NRT) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int c = rt * 16 + 4 * kq + r;
                const float y = gelu_erf((acc[j][r] - mean) * rs * prm[7][c] + prm[8][c]);
                const float v = (resv[j][r] + y) * mk;
                if (tail) y1[c * 16 + col] = tcv ? v : 0.f;  // (y1 is free: every wave passed a barrier after its MFMA chain)
                else o[(int64_t)c * T + tc] = v;
            }
        }
    }
    if (!tail) return;
    __syncthreads();
    // ---- tail: tail_rows x C times the [C][16] result, row tiles of 16 dealt to the waves
    const float4 *tw = reinterpret_cast<const float4 *>(a.tail_w16);
    const int nrt2 = (a.tail_rows + 15) >> 4;
    float *to = a.tail_out + (int64_t)b * a.tail_rows * T;
    const float tmk = (!a.tail_mask || tc < L) ? 1.f : 0.f;
    for (int rt = wave; rt < nrt2; rt += 4) {
        float4 w[NS4];
#pragma unroll
        for (int i = 0; i < NS4; i++) w[i] = tw[((int64_t)(rt * 4 + kq) * 16 + col) * NS4 + i];
        f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NS4; i++) {
            t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].x, yb[(16 * i + 0) * 16], t4, 0, 0, 0);
            t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].y, yb[(16 * i + 4) * 16], t4, 0, 0, 0);
            t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].z, yb[(16 * i + 8) * 16], t4, 0, 0, 0);
            t4 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i].w, yb[(16 * i + 12) * 16], t4, 0, 0, 0);
        }
        if (tcv) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = rt * 16 + 4 * kq + r;
                if (row < a.tail_rows) to[(int64_t)row * T + tc] = (t4[r] + (a.tail_b ? a.tail_b[row] : 0.f)) * tmk;
            }
        }
    }
}

// ---- a7: ConvFlow pre (1 -> C) + conditioning add: h = w*z[ch] + b + cond (modules.py:498-499,119)
// grid (T / 256, C, B): one element per thread (a per-thread loop over the channels is a chain of C dependent
// load -> store round trips on a handful of waves)
__global__ void cf_pre_kernel(const float *z, int ch, const float *w, const float *bias, const float *cond,
                              float *h, int C, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const float z0 = z[((int64_t)b * 2 + ch) * T + t];
    const int64_t o = ((int64_t)b * C + c) * T + t;
    h[o] = w[c] * z0 + bias[c] + cond[o];
}

// ---- a7: inverse rational-quadratic spline with linear tails (transforms.py:50-98,101-191) ----
// pr: [B, 3*nb-1, T] (already masked); z: [B,2,T]; logical x0 = z[ch0], x1 = z[ch1].
template <int NBMAX>
__global__ void rqs_inverse_kernel(const float *pr, float *z, const int *len, int ch0, int ch1, int nb, int T,
                                   float inv_sqrt_c_div) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= T) return;
    const int P = 3 * nb - 1;
    const float *q = pr + (int64_t)b * P * T + t;
    float mk = t < len[b] ? 1.f : 0.f;
    float *px0 = z + ((int64_t)b * 2 + ch0) * T + t, *px1 = z + ((int64_t)b * 2 + ch1) * T + t;
    // Every operand is requested up front, unconditionally (indices clamped into the spline's rows): x, then the widths
    // where the branch opens, then the heights, then the derivatives were four dependent memory round trips on a grid that
    // is a handful of workgroups at batch 1.  The arithmetic below is unchanged.
    float qw[NBMAX], qh[NBMAX], qd[NBMAX];
#pragma unroll
    for (int i = 0; i < NBMAX; i++) {
        const int ic = i < nb ? i : nb - 1, id = i < nb - 1 ? i : (nb > 1 ? nb - 2 : 0);
        qw[i] = q[(int64_t)ic * T];
        qh[i] = q[(int64_t)(nb + ic) * T];
        qd[i] = q[(int64_t)(2 * nb + id) * T];
    }
    const float x0 = *px0;
    float x = *px1;
    const float tb = 5.0f, minw = 1e-3f, minh = 1e-3f, mind = 1e-3f;
    float y = x;
    if (x >= -tb && x <= tb) {
        float w[NBMAX], h[NBMAX], cw[NBMAX + 1], ch[NBMAX + 1], d[NBMAX + 1];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NBMAX; i++)
            if (i < nb) {
                w[i] = qw[i] / inv_sqrt_c_div;
                mx = fmaxf(mx, w[i]);
            }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NBMAX; i++)
            if (i < nb) {
                w[i] = expf(w[i] - mx);
                s += w[i];
            }
        cw[0] = 0.f;
#pragma unroll
        for (int i = 0; i < NBMAX; i++)
            if (i < nb) {
                w[i] = minw + (1.0f - minw * nb) * (w[i] / s);
                cw[i + 1] = cw[i] + w[i];
            }
#pragma unroll
        for (int i = 0; i <= NBMAX; i++)
            if (i <= nb) cw[i] = (tb - (-tb)) * cw[i] + (-tb);
        cw[0] = -tb;
#pragma unroll
        for (int i = 0; i <= NBMAX; i++)
            if (i == nb) cw[i] = tb;
#pragma unroll
        for (int i = 0; i < NBMAX; i++)
            if (i < nb) w[i] = cw[i + 1] - cw[i];

        mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NBMAX; i++)
            if (i < nb) {
                h[i] = qh[i] / inv_sqrt_c_div;
                mx = fmaxf(mx, h[i]);
            }
        s = 0.f;
#pragma unroll
        for (int i = 0; i < NBMAX; i++)
            if (i < nb) {
                h[i] = expf(h[i] - mx);
                s += h[i];
            }
        ch[0] = 0.f;
#pragma unroll
        for (int i = 0; i < NBMAX; i++)
            if (i < nb) {
                h[i] = minh + (1.0f - minh * nb) * (h[i] / s);
                ch[i + 1] = ch[i] + h[i];
            }
#pragma unroll
        for (int i = 0; i <= NBMAX; i++)
            if (i <= nb) ch[i] = (tb - (-tb)) * ch[i] + (-tb);
        ch[0] = -tb;
#pragma unroll
        for (int i = 0; i <= NBMAX; i++)
            if (i == nb) ch[i] = tb;
#pragma unroll
        for (int i = 0; i < NBMAX; i++)
            if (i < nb) h[i] = ch[i + 1] - ch[i];

        const float cst = 0.5397424172369522f;  // log(exp(1 - 1e-3) - 1), transforms.py:70
        const float dedge = mind + softplus_f(cst);
#pragma unroll
        for (int i = 0; i <= NBMAX; i++)
            if (i <= nb) d[i] = (i == 0 || i == nb) ? dedge : mind + softplus_f(qd[i > 0 ? (i - 1 < NBMAX ? i - 1 : NBMAX - 1) : 0]);

        // searchsorted on cumheights (last knot + 1e-6), then gather by select (registers only)
        int bin = -1;
#pragma unroll
        for (int i = 0; i <= NBMAX; i++)
            if (i <= nb) {
                float loc = ch[i] + (i == nb ? 1e-6f : 0.f);
                if (x >= loc) bin++;
            }
        bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
        float icw = 0.f, ibw = 1.f, ich = 0.f, ih = 1.f, dd = 1.f, dp1 = 1.f;
#pragma unroll
        for (int i = 0; i < NBMAX; i++)
            if (i == bin) {
                icw = cw[i];
                ibw = w[i];
                ich = ch[i];
                ih = h[i];
                dd = d[i];
                dp1 = d[i + 1];
            }
        float delta = ih / ibw;
        float a = (x - ich) * (dd + dp1 - 2.0f * delta) + ih * (delta - dd);
        float bq = ih * dd - (x - ich) * (dd + dp1 - 2.0f * delta);
        float c = -delta * (x - ich);
        float disc = bq * bq - 4.0f * a * c;
        float root = (2.0f * c) / (-bq - sqrtf(disc));
        y = root * ibw + icw;
    }
    *px0 = x0 * mk;  // cat([x0, x1]) * x_mask (modules.py:521)
    *px1 = y * mk;
}

// ---- SDP tail: ElementwiseAffine reverse on channel ch -> logw (modules.py:407-409) -------------
__global__ void ea_logw_kernel(const float *z, int ch, float m0, float logs0, const int *len, float *logw, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (t >= T) return;
    float mk = t < len[b] ? 1.f : 0.f;
    logw[(int64_t)b * T + t] = (z[((int64_t)b * 2 + ch) * T + t] - m0) * expf(-logs0) * mk;
}

__global__ void scale_kernel(const float *in, float *out, float s, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * s;
}

// x[b,c,t] = (x[b,c,t] + bias_b[b,c]) (DurationPredictor cond, models.py:153-155)
__global__ void add_bias_b_kernel(const float *in, float *out, const float *bias_b, int stride, int C, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    int64_t o = ((int64_t)b * C + c) * T + t;
    out[o] = in[o] + bias_b[(int64_t)b * stride + c];
}

// ---- a9: durations (models.py:702-704): one block per utterance ----------------------------------
// w = exp(logw)*mask*length_scale; w_ceil = ceil(w); cum = inclusive scan (ints); y_len = max(sum,1)
__global__ void duration_kernel(const float *logw, const int *len, float length_scale, float *w_ceil, int *cum,
                                int *y_len, int T) {
    __shared__ int sh[256];
    __shared__ int carry_s;
    int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    const int L = len[b];
    for (int base = 0; base < T; base += 256) {
        int t = base + tid;
        int v = 0;
        if (t < T) {
            float mk = t < L ? 1.f : 0.f;
            float w = expf(logw[(int64_t)b * T + t]) * mk * length_scale;
            float c = ceilf(w);
            w_ceil[(int64_t)b * T + t] = c;
            v = (int)c;
        }
        sh[tid] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int add = tid >= off ? sh[tid - off] : 0;
            __syncthreads();
            sh[tid] += add;
            __syncthreads();
        }
        int carry = carry_s;
        if (t < T) cum[(int64_t)b * T + t] = carry + sh[tid];
        __syncthreads();
        if (tid == 255) carry_s = carry + sh[255];
        __syncthreads();
    }
    if (tid == 0) y_len[b] = carry_s < 1 ? 1 : carry_s;
}

// ---- a9: expand prior by durations + sample (commons.py:116-129, models.py:711-718) --------------
// z_p[b,c,f] = m_p[b,c,i(f)] + noise[b,c,f] * exp(logs_p[b,c,i(f)]) * noise_scale ; frames with no
// token (f >= y_len[b]) see m=0, logs=0 exactly as the reference's masked path matmul gives.
// m_p / logs_p are strided views (batch stride `bstride`, channel stride T).
__global__ void expand_prior_strided_kernel(const float *m_p, const float *logs_p, int64_t bstride, const int *cum,
                                            const int *len, const int *y_len, const float *noise,
                                            int64_t noise_stride, float noise_scale, float *z_p, int C, int T, int F,
                                            int Fnoise) {
    // grid (F / 64, C / 16, B): 16 channels of one frame per thread (the token search is repeated per channel group:
    // eight steps, against 16 x 3 loads)
    int f = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.z;
    if (f >= F) return;
    const int *cb = cum + (int64_t)b * T;
    int tok = -1;
    if (f < y_len[b]) {
        int lo = 0, hi = T;  // smallest i with cum[i] > f
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (cb[mid] > f) hi = mid;
            else lo = mid + 1;
        }
        if (lo < T && lo < len[b]) tok = lo;
    }
    const float *mb = m_p + (int64_t)b * bstride, *lb = logs_p + (int64_t)b * bstride;
    const int c0 = blockIdx.y * 16, c1 = c0 + 16 < C ? c0 + 16 : C;
    for (int c = c0; c < c1; c++) {
        float mp = 0.f, lp = 0.f;
        if (tok >= 0) {
            mp = mb[(int64_t)c * T + tok];
            lp = lb[(int64_t)c * T + tok];
        }
        float e = (noise && f < Fnoise) ? noise[((int64_t)b * C + c) * noise_stride + f] : 0.f;
        z_p[((int64_t)b * C + c) * F + f] = mp + e * expf(lp) * noise_scale;
    }
}

__global__ void mask_kernel(float *x, const int *len, int C, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    if (t >= len[b]) x[((int64_t)b * C + c) * T + t] = 0.f;
}

__global__ void ylen_to_i64(const int *in, int64_t *out, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = in[b];
}

// ---- f2: synthesize()'s post-processing on the device (voice.py:271-282 + AudioChunk, :88-91) ----------
// out[b][c][t] = t < len[b] ? in[b][c][t] : 0 over [B][C][T] tensors (run_frames: z = z_p * y_mask for the ragged flow)
__global__ __launch_bounds__(256) void masked_copy_kernel(const float *in, float *out, const int *len, int C, int T) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const int64_t i = ((int64_t)b * C + c) * T + t;
    out[i] = t < len[b] ? in[i] : 0.f;
}

// per utterance: peak = max|x| over its valid samples; x = peak < 1e-8 ? 0 : x / peak; x *= volume;
// clip to [-1,1]; int16 = trunc(clip(x * 32767, -32767, 32767)).  Same float32 operations in the same
// order as the NumPy code, so the PCM is bit-identical.
__global__ void peak_abs_kernel(const float *x, const int *y_len, int hop, int S, unsigned *peak_bits) {
    const int b = blockIdx.y;
    const int n = y_len[b] * hop < S ? y_len[b] * hop : S;
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(x[(int64_t)b * S + i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(&peak_bits[b], __float_as_uint(m));  // non-negative floats order as uints
}

__global__ void pcm16_kernel(const float *x, const int *y_len, int hop, int S, const unsigned *peak_bits,
                             int normalize, float volume, int16_t *out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const int n = y_len[b] * hop < S ? y_len[b] * hop : S;
    float v = 0.f;
    if (i < n) {
        v = x[(int64_t)b * S + i];
        if (normalize) {
            const float peak = __uint_as_float(peak_bits[b]);
            v = peak < 1e-8f ? 0.f : v / peak;
        }
        if (volume != 1.0f) v = v * volume;
        v = fminf(fmaxf(v, -1.0f), 1.0f);
        v = fminf(fmaxf(v * 32767.0f, -32767.0f), 32767.0f);
    }
    out[(int64_t)b * S + i] = (int16_t)v;
}

// copy a strided [B][C][T] view (batch stride bstride) into a contiguous buffer
__global__ void gather_view_kernel(const float *in, int64_t bstride, int cstride, float *out, int C, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t < T) out[((int64_t)b * C + c) * T + t] = in[(int64_t)b * bstride + (int64_t)c * cstride + t];
}

// ---- speaker conditioning: out[b,r] = bias[r] + W[r,:] . emb_g[sid[b],:] (1x1 conv on g) -------
__global__ void cond_matvec_kernel(const float *emb_g, const int64_t *sid, int n_speakers, const float *W,
                                   const float *bias, float *out, int rows, int gin) {
    int r = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (r >= rows) return;
    int64_t s = sid[b];
    s = s < 0 ? 0 : (s >= n_speakers ? n_speakers - 1 : s);
    const float *g = emb_g + s * gin;
    const float *w = W + (int64_t)r * gin;
    float acc = bias ? bias[r] : 0.f;
    for (int k = 0; k < gin; k++) acc += w[k] * g[k];
    out[(int64_t)b * rows + r] = acc;
}

// ---- a11: WN gate tanh(a[:H]) * sigmoid(a[H:]) (commons.py:99-106) --------------------------------
__global__ void wn_gate_kernel(const float *a, float *acts, int H, int T) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    float ta = a[((int64_t)b * 2 * H + c) * T + t], sa = a[((int64_t)b * 2 * H + H + c) * T + t];
    acts[((int64_t)b * H + c) * T + t] = tanhf(ta) * sigmoid_f(sa);
}

// The same gate when `a` comes from the split-exact conv engine: raw layout [2H/8][T][8] per utterance
// (conv_sx_engine.hip.hpp); acts stays planar [H][T] with row pitch T.  One thread = 8 channels of one frame.
__global__ __launch_bounds__(256) void wn_gate_blocked_kernel(const float *a, float *acts, int H, int T) {
    const int t = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const float *ab = a + (int64_t)b * 2 * H * T;
    const float4 *pt = reinterpret_cast<const float4 *>(ab + ((int64_t)cg * T + t) * 8);
    const float4 *ps = reinterpret_cast<const float4 *>(ab + ((int64_t)(H / 8 + cg) * T + t) * 8);
    const float4 t0 = pt[0], t1 = pt[1], s0 = ps[0], s1 = ps[1];
    const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int e = 0; e < 8; e++) acts[((int64_t)b * H + cg * 8 + e) * T + t] = tanhf(tv[e]) * sigmoid_f(sv[e]);
}

// ---- a11: WN residual/skip update (modules.py:203-209) --------------------------------------------
// not last: x = (x + rs[:, :H]) * mask; skip (+)= rs[:, H:]      last: skip = (skip + rs) * mask
__global__ void wn_update_kernel(float *x, float *skip, const float *rs, const int *len, int H, int T, int first,
                                 int last) {
    int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    float mk = t < len[b] ? 1.f : 0.f;
    int64_t o = ((int64_t)b * H + c) * T + t;
    if (!last) {
        int64_t r0 = ((int64_t)b * 2 * H + c) * T + t, r1 = ((int64_t)b * 2 * H + H + c) * T + t;
        x[o] = (x[o] + rs[r0]) * mk;
        skip[o] = first ? rs[r1] : skip[o] + rs[r1];
    } else {
        float s = first ? rs[o] : skip[o] + rs[o];
        skip[o] = s * mk;
    }
}

// ---- a12 tail: leaky_relu(0.01) -> conv_post (C -> 1, k taps, no bias) -> tanh ---------------------
// The one genuinely HBM-bound kernel: C*4 bytes read + 4 written per sample.
// vlen / hop (optional): samples at and behind vlen[b] * hop - the padding of a shorter utterance in a batch - are written as
// zeros (and, with the ragged generator, were never rendered: nothing behind them is read for a sample that is kept).
__global__ __launch_bounds__(256) void post_conv_tanh_kernel(const float *x, const float *w, float *out, int C,
                                                             int K, int T, float slope, const int *vlen = nullptr, int hop = 1) {
    extern __shared__ float sm[];  // [C][256 + K - 1] staged tile, then weights [C*K]
    const int LW = 256 + K - 1;
    float *ws = sm + (size_t)C * LW;
    int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    const long long nv_ = vlen ? (long long)vlen[b] * hop : (long long)T;
    const int NV = nv_ < T ? (int)nv_ : T;
    if (t0 >= NV) {  // (uniform) the whole tile is padding
        if (t0 + tid < T) out[(int64_t)b * T + t0 + tid] = 0.f;
        return;
    }
    const float *xb = x + (int64_t)b * C * T;
    for (int i = tid; i < C * K; i += 256) ws[i] = w[i];
    const int pad = (K - 1) / 2;
    for (int c = 0; c < C; c++)
        for (int i = tid; i < LW; i += 256) {
            int t = t0 - pad + i;
            float v = (t >= 0 && t < T) ? xb[(int64_t)c * T + t] : 0.f;
            sm[c * LW + i] = v > 0.f ? v : v * slope;
        }
    __syncthreads();
    int t = t0 + tid;
    if (t >= T) return;
    float acc = 0.f;
    for (int c = 0; c < C; c++)
        for (int k = 0; k < K; k++) acc += ws[c * K + k] * sm[c * LW + tid + k];
    out[(int64_t)b * T + t] = t < NV ? tanhf(acc) : 0.f;
}

// Same tail for the split-exact generator: x is the fp32 raw layout [C/8][T][8] (conv_sx_engine.hip.hpp).
// Same (channel, tap) summation order as above.
// KT: the tap count as a compile-time constant (7 in every VITS voice; 0 = generic): with runtime loop bounds the
// 32 x 7 LDS reads of a thread are issued and awaited one by one.
template <int KT>
__global__ __launch_bounds__(256) void post_conv_tanh_blocked_kernel(const float *x, const float *__restrict__ w, float *out,
                                                                     int C, int K, int T, float slope, const int *vlen = nullptr,
                                                                     int hop = 1) {
    extern __shared__ float sm[];  // [C][256 + K - 1] staged tile (the weights are uniform: scalar loads, no LDS)
    const int LW = 256 + K - 1;
    int b = blockIdx.y, t0 = blockIdx.x * 256, tid = threadIdx.x;
    // (vlen / hop: see post_conv_tanh_kernel - zeros at and behind the utterance's end, whose tiles are not even read)
    const long long nv_ = vlen ? (long long)vlen[b] * hop : (long long)T;
    const int NV = nv_ < T ? (int)nv_ : T;
    if (t0 >= NV) {  // (uniform) the whole tile is padding
        if (t0 + tid < T) out[(int64_t)b * T + t0 + tid] = 0.f;
        return;
    }
    const float *xb = x + (int64_t)b * C * T;
    const int pad = (K - 1) / 2;
    // one thread moves a whole cell (8 channels of one time step, 32 contiguous bytes) per round: a wave reads
    // 2 KiB of consecutive addresses
    // (four rounds of loads are requested before the first is consumed: a round trip to HBM per round otherwise,
    // and a workgroup has only ~5 rounds of work)
    const int ncell = (C / 8) * LW;
    for (int i0 = tid; i0 < ncell; i0 += 4 * 256) {
        float4 v0[4], v1[4];
        int colr[4], cgr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = i0 + r * 256;
            cgr[r] = i / LW;
            colr[r] = i - cgr[r] * LW;
            const int t = t0 - pad + colr[r];
            v0[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            v1[r] = v0[r];
            if (i < ncell && t >= 0 && t < T) {
                const float4 *p = reinterpret_cast<const float4 *>(xb + ((int64_t)cgr[r] * T + t) * 8);
                v0[r] = p[0];
                v1[r] = p[1];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (i0 + r * 256 < ncell) {
                const float v[8] = {v0[r].x, v0[r].y, v0[r].z, v0[r].w, v1[r].x, v1[r].y, v1[r].z, v1[r].w};
#pragma unroll
                for (int e = 0; e < 8; e++) sm[(cgr[r] * 8 + e) * LW + colr[r]] = v[e] > 0.f ? v[e] : v[e] * slope;
            }
        }
    }
    __syncthreads();
    int t = t0 + tid;
    if (t >= T) return;
    float acc = 0.f;
    if constexpr (KT > 0) {
        for (int c0 = 0; c0 < C; c0 += 4) {  // (C % 8 == 0 in this layout) same (channel, tap) summation order
            float xv[4][KT];
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
#pragma unroll
                for (int k = 0; k < KT; k++) xv[cc][k] = sm[(c0 + cc) * LW + tid + k];
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
#pragma unroll
                for (int k = 0; k < KT; k++) acc += w[(c0 + cc) * KT + k] * xv[cc][k];
        }
    } else {
        for (int c = 0; c < C; c++)
            for (int k = 0; k < K; k++) acc += w[c * K + k] * sm[c * LW + tid + k];
    }
    out[(int64_t)b * T + t] = t < NV ? tanhf(acc) : 0.f;
}

// ---- noise: Philox4x32-10 counter RNG + Box-Muller (production path; parity uses injected noise) ---
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void fill_normal_kernel(float *out, int64_t n, uint64_t seed, uint64_t stream_id) {
    int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= n) return;
    uint32_t r[4];
    philox4x32_10((uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32),
                  (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float k = 2.3283064365386963e-10f;  // 2^-32
    float u0 = ((float)r[0] + 0.5f) * k, u1 = ((float)r[1] + 0.5f) * k;
    float u2 = ((float)r[2] + 0.5f) * k, u3 = ((float)r[3] + 0.5f) * k;
    u0 = fminf(fmaxf(u0, 1e-12f), 1.0f);
    u2 = fminf(fmaxf(u2, 1e-12f), 1.0f);
    float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    float v[4] = {ra * cosf(6.283185307179586f * u1), ra * sinf(6.283185307179586f * u1),
                  rb * cosf(6.283185307179586f * u3), rb * sinf(6.283185307179586f * u3)};
    for (int j = 0; j < 4; j++)
        if (i4 * 4 + j < n) out[i4 * 4 + j] = v[j];
}

// ---- a3: relative-position self-attention core on the f32 matrix cores ---------------------------
// One wavefront per (32-query block, head, utterance); flash-style online softmax over 32-key blocks.
//   S^T[j,i] = sum_d k[d,j] * (q[d,i]/sqrt(dk))        A = K rows (coalesced global), B = Q fragment
//   + q_i . E_k[j-i+w] for |j-i| <= w ; masked keys -> -1e4 (attentions.py:232-247)
//   O^T[d,i] = sum_j v[d,j] * P^T[j,i]                  B = the S^T accumulator registers, in place:
//       register r of lane (c, h) holds row rho(r) + 4h of column c, so the k-pair of MFMA #r is
//       (rho(r), rho(r)+4) and the A operand is V[d, j0 + rho(r) + 4h] (staged through LDS).
//   + sum_{|j-i|<=w} p[i,j] * E_v[j-i+w] (attentions.py:261-268); everything divided by the row sum.
// Query columns live on lanes (lane&31) in both products, so the softmax statistics are per-lane
// scalars and the only cross-lane traffic is one lane^32 exchange per block.
// KH = 2: the key blocks of an utterance are dealt to two halves of 4 waves each (first / second half of the blocks, own
// K/V staging buffers), whose online-softmax states (m, l, O, relative-value weights) meet in LDS at the end: the
// dependent chain of a wave - 2 x DKB x 16 64-cycle MFMAs plus a softmax per key block, one wave per SIMD - halves, and a
// batch of 32 fills 1024 instead of 512 of the chip's wave slots.  The split depends on the utterance's length only.
template <int DKB, int KH = 2>  // ceil(dk/32), key halves
__global__ __launch_bounds__(256 * KH) void attention_relpos_kernel(const float *qkv, float *out, const float *relk,
                                                               const float *relv, const int *len, int Hc, int T,
                                                               int dk, int win, uint16_t *planes = nullptr,
                                                               unsigned *peak = nullptr) {
    // 4 wavefronts = 4 consecutive 32-query blocks of one (utterance, head); every 32-key block of K and V
    // is staged ONCE in LDS (coalesced 128-byte rows) and shared by the four waves.
    // K block transposed: kt[key j][d], d contiguous (row pitch KP), so that a lane's A values of four consecutive
    // k-steps (d = 2 st + hi) are one 32-byte read; V block vs[d][key j] (row pitch VP): the four keys a lane needs
    // per accumulator quad are one 16-byte read.  Pitches = 4 mod 32 floats: 16-byte reads of 8 lanes tile the banks.
    constexpr int KP = DKB * 32 + 4, VP = 36;
    constexpr int HALF_FLOATS = 32 * KP + DKB * 32 * VP;      // one half's K and V staging
    constexpr int MERGE_ITEMS = 2 + DKB * 16 + 9;             // m, l, O accumulators, relative-value weights
    constexpr int SMEM = KH * HALF_FLOATS > 256 * MERGE_ITEMS || KH == 1 ? KH * HALF_FLOATS : 256 * MERGE_ITEMS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    __shared__ float s_pk8[4];
    const int half = KH == 1 ? 0 : (int)(threadIdx.x >> 8);
    float *kt = smem + half * HALF_FLOATS, *vs = kt + 32 * KP;
    const int tid = threadIdx.x & 255, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int i0 = (blockIdx.x * 4 + wave) * 32, h = blockIdx.y, b = blockIdx.z;
    const int L = len[b] < T ? len[b] : T;
    const int i = i0 + l31;
    const float *q = qkv + ((int64_t)b * 3 * Hc + (int64_t)h * dk) * T;
    const float *k = q + (int64_t)Hc * T;
    const float *v = k + (int64_t)Hc * T;
    float *o = out + ((int64_t)b * Hc + (int64_t)h * dk) * T;
    constexpr int STEPS = DKB * 16;
    const bool active = i0 < L;  // a fully padded query block only helps staging and writes zeros
    const float rsq = 1.f / sqrtf((float)dk);
    // dk a whole number of 32-row blocks (every VITS voice: dk = 96): loads need no per-element predicate - clamped
    // token indices keep them inside the tensor, padded keys are masked by their -1e4 logits, padded queries by the
    // predicated store - and their addresses step by a constant (one 64-bit add per load instead of ~15 operations
    // and a branch: the kernel is issue-bound, one wave per SIMD)
    const bool full = dk == DKB * 32;
    float qf[STEPS];
    if (full) {
        const float *qp = q + (int64_t)hi * T + (i < T ? i : T - 1);
        const int64_t two_t = 2 * (int64_t)T;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            qf[s] = *qp * rsq;
            qp += two_t;
        }
    } else {
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            int d = 2 * s + hi;
            qf[s] = (active && d < dk && i < T) ? q[(int64_t)d * T + i] * rsq : 0.f;
        }
    }
    // relative-key logits of this lane's query: rq[m] = q_i . E_k[m]   (E_k staged once per workgroup: the nine
    // dk-vectors would otherwise be 9 x STEPS dependent global loads per lane in front of the first key block)
    float rq[9];
    const int nrel = 2 * win + 1;
    float *relk_s = vs;  // (vs is not in use yet; DKB * 32 * VP >= 9 * DKB * 32 floats)
    for (int e = tid; e < nrel * DKB * 32; e += 256) {
        const int m = e / (DKB * 32), d = e - m * (DKB * 32);
        relk_s[e] = d < dk ? relk[m * dk + d] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 9; m++) {
        float s0 = 0.f, s1 = 0.f;
        if (m < nrel) {
#pragma unroll
            for (int st = 0; st < STEPS; st += 2) {
                s0 += qf[st] * relk_s[m * (DKB * 32) + 2 * st + hi];
                s1 += qf[st + 1] * relk_s[m * (DKB * 32) + 2 * st + 2 + hi];
            }
        }
        float s = s0 + s1;
        s += __shfl_xor(s, 32);
        rq[m] = s;
    }
    float mrun = -INFINITY, lrun = 0.f;
    float wrel[9];
#pragma unroll
    for (int m = 0; m < 9; m++) wrel[m] = 0.f;
    f32x16 oacc[DKB];
#pragma unroll
    for (int db = 0; db < DKB; db++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[db][r] = 0.f;

    const int nkb = (L + 31) / 32;
    const int nkb_h = (nkb + KH - 1) / KH, kb0 = half * nkb_h;  // this half: blocks kb0 .. kb0 + nkb_h - 1 (those < nkb)
    const int srow = tid >> 5, scol = tid & 31;  // staging: 8 rows x 32 columns per pass
    // K/V block kb+1 is fetched into registers while block kb is being consumed from LDS (the global latency of a
    // block would otherwise sit in front of every one of its 2 x DKB x 16 MFMAs)
    constexpr int NST = DKB * 4;  // rows of a block staged per thread
    float kreg[NST], vreg[NST];
    auto fetch = [&](int kb) {
        const int j = kb * 32 + scol;
        if (full) {
            const int64_t off = (int64_t)srow * T + (j < T ? j : T - 1), eight_t = 8 * (int64_t)T;
            const float *kp = k + off, *vp = v + off;
#pragma unroll
            for (int r = 0; r < NST; r++) {
                kreg[r] = *kp;
                vreg[r] = *vp;
                kp += eight_t;
                vp += eight_t;
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < NST; r++) {
            const int d = srow + 8 * r;
            const bool ok = d < dk && j < T;
            kreg[r] = ok ? k[(int64_t)d * T + j] : 0.f;
            vreg[r] = ok ? v[(int64_t)d * T + j] : 0.f;
        }
    };
    if (kb0 < nkb) fetch(kb0);
    for (int it = 0; it < nkb_h; it++) {
        const int kb = kb0 + it;
        const bool kvalid = kb < nkb;  // (uniform per half; both halves pass the same barriers)
        const int j0 = kb * 32;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NST; r++) {
            const int d = srow + 8 * r;
            kt[scol * KP + d] = kreg[r];
            vs[d * VP + scol] = vreg[r];
        }
        if (it + 1 < nkb_h && kb + 1 < nkb) fetch(kb + 1);
        __syncthreads();
        if (!active || !kvalid) continue;
        // two accumulators: the STEPS MFMAs are one dependent chain otherwise (16 passes each)
        f32x16 s, s2;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = s2[r] = 0.f;
#pragma unroll
        for (int g = 0; g < STEPS / 4; g++) {  // eight d-values = four k-steps per 32-byte read
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(&kt[l31 * KP + 8 * g]);
            const f32x4 up = *reinterpret_cast<const f32x4 *>(&kt[l31 * KP + 8 * g + 4]);
            const float a0 = hi ? lo[1] : lo[0], a1 = hi ? lo[3] : lo[2], a2 = hi ? up[1] : up[0], a3 = hi ? up[3] : up[2];
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, qf[4 * g], s, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, qf[4 * g + 1], s2, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, qf[4 * g + 2], s, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, qf[4 * g + 3], s2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] += s2[r];
        const bool near = (j0 + 31 >= i0 - win) && (j0 <= i0 + 31 + win);
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            float sv = s[r];
            if (near) {
                int m = j - i + win;
#pragma unroll
                for (int mm = 0; mm < 9; mm++)
                    if (mm == m && mm < nrel) sv += rq[mm];
            }
            if (j >= L) sv = -1e4f;
            s[r] = sv;
            bm = fmaxf(bm, sv);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float mnew = fmaxf(mrun, bm);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float p = __expf(s[r] - mnew);
            s[r] = p;
            psum += p;
        }
        psum += __shfl_xor(psum, 32);
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int db = 0; db < DKB; db++)
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[db][r] *= alpha;
#pragma unroll
        for (int m = 0; m < 9; m++) wrel[m] *= alpha;
        if (near) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                int m = j - i + win;
#pragma unroll
                for (int mm = 0; mm < 9; mm++)
                    if (mm == m && mm < nrel) wrel[mm] += s[r];
            }
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {  // keys 8 q4 + 4 hi .. + 3: one 16-byte read per V row block
            f32x4 av[DKB];
#pragma unroll
            for (int db = 0; db < DKB; db++)
                av[db] = *reinterpret_cast<const f32x4 *>(&vs[(db * 32 + l31) * VP + 8 * q4 + 4 * hi]);
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int db = 0; db < DKB; db++)
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[db][e], s[4 * q4 + e], oacc[db], 0, 0, 0);
        }
    }
    if constexpr (KH == 2) {
        // the second half's state -> LDS ([item][lane of the half]: conflict free) -> folded into the first half's
        __syncthreads();  // (both halves are done with their staging buffers)
        if (half == 1) {
            smem[0 * 256 + tid] = mrun;
            smem[1 * 256 + tid] = lrun;
#pragma unroll
            for (int db = 0; db < DKB; db++)
#pragma unroll
                for (int r = 0; r < 16; r++) smem[(2 + db * 16 + r) * 256 + tid] = oacc[db][r];
#pragma unroll
            for (int m = 0; m < 9; m++) smem[(2 + DKB * 16 + m) * 256 + tid] = wrel[m];
        }
        __syncthreads();
        if (half == 0 && active) {
            const float m1 = smem[0 * 256 + tid], l1 = smem[1 * 256 + tid];
            const float mnew = fmaxf(mrun, m1);  // (mrun is finite: an active wave's first half holds key block 0)
            const float a0 = __expf(mrun - mnew), a1 = __expf(m1 - mnew);
            lrun = lrun * a0 + l1 * a1;
            mrun = mnew;
#pragma unroll
            for (int db = 0; db < DKB; db++)
#pragma unroll
                for (int r = 0; r < 16; r++) oacc[db][r] = oacc[db][r] * a0 + smem[(2 + db * 16 + r) * 256 + tid] * a1;
#pragma unroll
            for (int m = 0; m < 9; m++) wrel[m] = wrel[m] * a0 + smem[(2 + DKB * 16 + m) * 256 + tid] * a1;
        }
    }
    // relative-value embeddings E_v staged once (as E_k was): nine values per output element otherwise come as
    // dependent global loads in the epilogue
    __syncthreads();
    float *relv_s = smem;  // (32 * KP >= 9 * DKB * 32 floats)
    // (all nine rows, zeros beyond 2 win + 1: the epilogue's nine terms per element then need no predicate - as 432
    // branch + LDS read + wait triples they cost ~18 us of this kernel's ~95)
    for (int e = threadIdx.x; e < 9 * DKB * 32; e += 256 * KH) {
        const int m = e / (DKB * 32), d = e - m * (DKB * 32);
        relv_s[e] = (m < nrel && d < dk) ? relv[m * dk + d] : 0.f;
    }
    __syncthreads();
    // planes (optional; dk % 8 == 0): the output once more as the fp16 operand planes of conv_o on the split-operand engine
    // ([Hc/8][T][8] cells per plane): a lane holds four consecutive channels of each group of eight = half a cell
    uint16_t *pb = planes ? planes + (int64_t)b * 3 * Hc * T : nullptr;
    const int64_t plane_elems = (int64_t)Hc * T;
    float pk = 0.f;
    if (half != 0) {
        // (the first half writes the result)
    } else if (!active) {
        for (int d = hi; d < dk; d += 2)
            if (i < T) o[(int64_t)d * T + i] = 0.f;
        if (pb && i < T)
            for (int d = 4 * hi; d < dk; d += 8) {
                const int64_t cell = ((int64_t)((h * dk + d) >> 3) * T + i) * 8 + 4 * hi;
                *reinterpret_cast<u32x2 *>(pb + cell) = u32x2{0u, 0u};
                *reinterpret_cast<u32x2 *>(pb + plane_elems + cell) = u32x2{0u, 0u};
            }
    } else {
#pragma unroll
        for (int m = 0; m < 9; m++) wrel[m] += __shfl_xor(wrel[m], 32);
        const bool qvalid = i < L;
        const float rl = 1.f / lrun;
        if (i < T) {
#pragma unroll
            for (int db = 0; db < DKB; db++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float ov[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * q + e;
                        const int d = db * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        ov[e] = 0.f;
                        if (full || d < dk) {
                            float val = oacc[db][r];
#pragma unroll
                            for (int m = 0; m < 9; m++) val += wrel[m] * relv_s[m * (DKB * 32) + d];  // (wrel[m >= nrel] = 0)
                            ov[e] = qvalid ? val * rl : 0.f;
                            o[(int64_t)d * T + i] = ov[e];
                        }
                    }
                    const int d0 = db * 32 + 8 * q + 4 * hi;
                    if (pb && (full || d0 < dk)) {
                        unsigned wa[2], wb[2];
                        split2h_pair_pk(ov[0], ov[1], wa[0], wa[1], pk);
                        split2h_pair_pk(ov[2], ov[3], wb[0], wb[1], pk);
                        const int64_t cell = ((int64_t)((h * dk + d0) >> 3) * T + i) * 8 + 4 * hi;
                        *reinterpret_cast<u32x2 *>(pb + cell) = u32x2{wa[0], wb[0]};
                        *reinterpret_cast<u32x2 *>(pb + plane_elems + cell) = u32x2{wa[1], wb[1]};
                    }
                }
        }
    }
    if (pb && peak) {  // (uniform: every thread is here; the first half's four waves carry the peak)
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) pk = __builtin_fmaxf(pk, __shfl_xor(pk, o2, 64));
        if (half == 0 && lane == 0) s_pk8[wave] = pk;
        __syncthreads();
        if (threadIdx.x == 0) {
            pk = __builtin_fmaxf(__builtin_fmaxf(s_pk8[0], s_pk8[1]), __builtin_fmaxf(s_pk8[2], s_pk8[3]));
            unsigned *slot = peak + ((blockIdx.x + blockIdx.y + blockIdx.z) & (kSxPeakSlots - 1)) * kSxPeakStride;
            const unsigned bits = __float_as_uint(pk);
            if (bits > __builtin_nontemporal_load(slot)) atomicMax(slot, bits);
        }
    }
}

}  // namespace vitsmi
