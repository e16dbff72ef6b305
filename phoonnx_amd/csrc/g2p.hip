// g2p.hip — ByT5 G2P engine: host pipeline + the C ABI of include/g2pmi.h.
//
// T5ForConditionalGeneration as phoonnx/phonemizers/mul.py runs it (batch 1): encoder over the input bytes, decoder over
// the generated prefix, lm_head.  Activations are [channels][time] fp32 (the layout of the f32 conv engine, whose 1x1
// convolution IS the Linear layer: every projection runs on v_mfma_f32_32x32x2_f32 through conv_engine_kernel); RMS norm,
// attention with the bucketed relative-position bias, the gated activation, embedding and argmax are small kernels below.
// g2p_generate keeps the decoder's self-attention keys / values and the cross-attention keys / values of the encoder
// output in a cache, so a generated token costs one decoder step instead of a whole-graph run.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/g2pmi.h"
#include "../../include/vitsmi.h"
#include "g2p_model.hpp"

using namespace vitsmi;

namespace {
thread_local std::string g_g2p_open_error;

// x[c][t] = table[ids[t]][c]
__global__ void g2p_embed_kernel(const int64_t *ids, int id_stride, const float *table, float *x, int C, int T, int pitch,
                                 int vocab) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (c >= C) return;
    int64_t id = ids[(int64_t)t * id_stride];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // (range-checked on the host for host inputs; generated ids are in range)
    x[(int64_t)c * pitch + t] = table[id * C + c];
}

// T5LayerNorm (modeling_t5.py): y = x * rsqrt(mean(x^2) + eps) * g, over channels, one workgroup per time step
__global__ __launch_bounds__(256) void g2p_rmsnorm_kernel(const float *x, const float *g, float *y, int C, int T, int xpitch,
                                                          int ypitch, float eps) {
    __shared__ float red[256];
    const int t = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int c = tid; c < C; c += 256) {
        const float v = x[(int64_t)c * xpitch + t];
        s += v * v;
    }
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const float rs = 1.0f / sqrtf(red[0] / (float)C + eps);
    for (int c = tid; c < C; c += 256) y[(int64_t)c * ypitch + t] = x[(int64_t)c * xpitch + t] * rs * g[c];
}

// Attention of one (query column, head): scores_j = q . k_j + bias[bucket(j - (i + q_off))][head] (no 1/sqrt(d) in T5),
// causal: keys j <= i + q_off only; softmax; out = sum_j p_j v_j.
// Columns are grouped in sequences of `seg` queries: column c is query i = c % seg of sequence b = c / seg, whose keys /
// values start kv_bs floats into k / v (channel stride kp, time contiguous) and number lens[b] (or Tk if lens is null).
// q / out element (channel ch, column c) sits at ch * q_cs + c * q_ts: [C][T] activations have (T, 1), the decoder step's
// [NB][C] vectors (1, C).  bucket_lut: bucket of (j - i) at index (j - i) + lut_zero, or nullptr (cross attention).
// Keys / values are channel-major (time contiguous): lanes run along time, and every loop over channels issues its
// loads eight at a time (one dependent load per channel made a decoder step's eight attention calls 30 us each).
__device__ __forceinline__ void g2p_attention_body_generic(const float *__restrict__ q, int q_cs, int q_ts,
                                                   const float *__restrict__ k, const float *__restrict__ v, int kp,
                                                   int64_t kv_bs, const float *__restrict__ bias,
                                                   const int *__restrict__ bucket_lut, int lut_zero,
                                                   float *__restrict__ out, int heads, int dk, int seg,
                                                   const int *__restrict__ lens, int Tk, int q_off, int causal, int c, int h,
                                                   float *sc, float *red) {
    // sc: [Tk] scores, then [dk] the query; red: 256 floats (both workgroup-shared)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = c / seg, i = c - b * seg;
    const int ipos = i + q_off;
    const int nk = lens ? lens[b] : Tk;
    const int lim = causal ? (ipos + 1 < nk ? ipos + 1 : nk) : nk;
    float *qs = sc + Tk;
    for (int d = tid; d < dk; d += 256) qs[d] = q[((int64_t)h * dk + d) * q_cs + (int64_t)c * q_ts];
    __syncthreads();
    const float *kh = k + (int64_t)b * kv_bs + (int64_t)h * dk * kp, *vh = v + (int64_t)b * kv_bs + (int64_t)h * dk * kp;
    float mx = -__builtin_inff();
    for (int j = tid; j < lim; j += 256) {
        float s = 0.f;
        int d = 0;
        for (; d + 8 <= dk; d += 8) {
            float kv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) kv[e] = kh[(int64_t)(d + e) * kp + j];
#pragma unroll
            for (int e = 0; e < 8; e++) s += qs[d + e] * kv[e];
        }
        for (; d < dk; d++) s += qs[d] * kh[(int64_t)d * kp + j];
        if (bucket_lut) s += bias[bucket_lut[j - ipos + lut_zero] * heads + h];
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < lim; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    red[tid] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const float inv = 1.0f / red[0];
    // out[d] = sum_j p_j v[d][j]: a wave per group of eight channels, lanes along the keys, wave reduction
    for (int d0 = wave * 8; d0 < dk; d0 += 32) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = 0.f;
        for (int j = lane; j < lim; j += 64) {
            const float pj = sc[j];
            float vv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) vv[e] = d0 + e < dk ? vh[(int64_t)(d0 + e) * kp + j] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] += pj * vv[e];
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[e] += __shfl_xor(acc[e], o, 64);
            if (lane == 0 && d0 + e < dk) out[((int64_t)h * dk + d0 + e) * q_cs + (int64_t)c * q_ts] = acc[e] * inv;
        }
    }
}


// The same attention for a head width known at compile time, restructured for memory-level parallelism.  At a decoder step
// the generic body above is a chain of dependent round trips to L2 / HBM - eight groups of eight key loads, then six of
// value loads, ~14 latencies for a few hundred KB - and the eight attention launches of a token were a third of its time
// (13-16 us each, tools/g2p_decode_prof.py).  Here a thread requests ALL DK channels of its key at once, and the value
// fragments of the first 256 keys are requested BEFORE the softmax reductions (they do not depend on the scores), so a
// step's attention pays about one memory latency.  Sums run in the generic body's order (channels ascending per score,
// keys ascending per output channel): same bits.
template <int DK>
__device__ __forceinline__ void g2p_attention_body_t(const float *__restrict__ q, int q_cs, int q_ts,
                                                     const float *__restrict__ k, const float *__restrict__ v, int kp,
                                                     int64_t kv_bs, const float *__restrict__ bias,
                                                     const int *__restrict__ bucket_lut, int lut_zero,
                                                     float *__restrict__ out, int heads, int seg,
                                                     const int *__restrict__ lens, int Tk, int q_off, int causal, int c, int h,
                                                     float *sc, float *red) {
    static_assert(DK % 32 == 0, "a wave per group of eight channels, four waves");
    constexpr int VR = 4, NR = DK / 32;  // value prefetch: key rounds of 64 (256 keys), channel rounds of 32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = c / seg, i = c - b * seg;
    const int ipos = i + q_off;
    const int nk = lens ? lens[b] : Tk;
    const int lim = causal ? (ipos + 1 < nk ? ipos + 1 : nk) : nk;
    float *qs = sc + Tk;
    const float *kh = k + (int64_t)b * kv_bs + (int64_t)h * DK * kp, *vh = v + (int64_t)b * kv_bs + (int64_t)h * DK * kp;
    const float qv = q[((int64_t)h * DK + (tid < DK ? tid : 0)) * q_cs + (int64_t)c * q_ts];
    float kv[DK];
    // (loads are unconditional - a lane without a key re-reads the last one and ignores it: a conditional load is a branch
    // with a full wait at its join, which serialises exactly what this body wants in flight together)
    const int jlast = lim > 0 ? lim - 1 : 0;
    auto load_k = [&](int j) {
        const int jc = j < lim ? j : jlast;
#pragma unroll
        for (int d = 0; d < DK; d++) kv[d] = kh[(int64_t)d * kp + jc];
    };
    load_k(tid);
    float vpre[NR][VR][8];
#pragma unroll
    for (int rr = 0; rr < NR; rr++)
#pragma unroll
        for (int r = 0; r < VR; r++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int j = lane + 64 * r;
                vpre[rr][r][e] = vh[(int64_t)(wave * 8 + 32 * rr + e) * kp + (j < lim ? j : jlast)];
            }
    if (tid < DK) qs[tid] = qv;
    __syncthreads();
    float mx = -__builtin_inff();
    for (int jb = 0; jb < lim; jb += 256) {
        const int j = jb + tid;
        if (jb) load_k(j);
        if (j < lim) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DK; d++) s += qs[d] * kv[d];
            if (bucket_lut) s += bias[bucket_lut[j - ipos + lut_zero] * heads + h];
            sc[j] = s;
            mx = fmaxf(mx, s);
        }
    }
    // maximum and sum over the workgroup: butterflies inside a wave, four values through LDS (the generic body's eight-level
    // tree costs sixteen barriers per launch; the order of the sum differs from it in the last bits)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < lim; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();  // (also: every sc[j] is written before the value phase reads it)
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
#pragma unroll
    for (int rr = 0; rr < NR; rr++) {
        const int d0 = wave * 8 + 32 * rr;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = 0.f;
#pragma unroll
        for (int r = 0; r < VR; r++) {
            const int j = lane + 64 * r;
            if (j < lim) {
                const float pj = sc[j];
#pragma unroll
                for (int e = 0; e < 8; e++) acc[e] += pj * vpre[rr][r][e];
            }
        }
        for (int j = lane + 64 * VR; j < lim; j += 64) {  // keys beyond the prefetch
            const float pj = sc[j];
            float vv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) vv[e] = vh[(int64_t)(d0 + e) * kp + j];
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] += pj * vv[e];
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[e] += __shfl_xor(acc[e], o, 64);
            if (lane == 0) out[((int64_t)h * DK + d0 + e) * q_cs + (int64_t)c * q_ts] = acc[e] * inv;
        }
    }
}

__device__ __forceinline__ void g2p_attention_body(const float *__restrict__ q, int q_cs, int q_ts,
                                                   const float *__restrict__ k, const float *__restrict__ v, int kp,
                                                   int64_t kv_bs, const float *__restrict__ bias,
                                                   const int *__restrict__ bucket_lut, int lut_zero,
                                                   float *__restrict__ out, int heads, int dk, int seg,
                                                   const int *__restrict__ lens, int Tk, int q_off, int causal, int c, int h,
                                                   float *sc, float *red) {
    if (dk == 64)  // (uniform) every ByT5 / mT5 size: d_kv = 64
        g2p_attention_body_t<64>(q, q_cs, q_ts, k, v, kp, kv_bs, bias, bucket_lut, lut_zero, out, heads, seg, lens, Tk, q_off, causal,
                                 c, h, sc, red);
    else
        g2p_attention_body_generic(q, q_cs, q_ts, k, v, kp, kv_bs, bias, bucket_lut, lut_zero, out, heads, dk, seg, lens, Tk, q_off,
                                   causal, c, h, sc, red);
}

__global__ __launch_bounds__(256) void g2p_attention_kernel(const float *__restrict__ q, int q_cs, int q_ts,
                                                            const float *__restrict__ k, const float *__restrict__ v, int kp,
                                                            int64_t kv_bs, const float *__restrict__ bias,
                                                            const int *__restrict__ bucket_lut, int lut_zero,
                                                            float *__restrict__ out, int heads, int dk, int seg,
                                                            const int *__restrict__ lens, int Tk, int q_off, int causal) {
    extern __shared__ float sc[];
    __shared__ float red[256];
    g2p_attention_body(q, q_cs, q_ts, k, v, kp, kv_bs, bias, bucket_lut, lut_zero, out, heads, dk, seg, lens, Tk, q_off, causal,
                       (int)blockIdx.x, (int)blockIdx.y, sc, red);
}

// T5DenseGatedActDense: h = act(a) * b;  T5DenseActDense: h = act(a)     act: 0 gelu_new, 1 relu, 2 gelu (erf)
__global__ void g2p_act_kernel(const float *a, const float *b, float *out, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i];
    float y;
    if (act == 1) y = fmaxf(x, 0.f);
    else if (act == 2) y = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    else y = 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
    out[i] = b ? y * b[i] : y;
}

// Linear layers over a short sequence (the encoder over the input bytes, teacher-forced decoder prefixes, the cross
// keys / values): Y_j[out_j][T] = W_j[out_j][in] X[in][T] (+ res), up to three W per launch.  T is tens to a few
// hundred columns, so a launch is bound by streaming W once and - per wave - by the depth of the reduction, not by
// matrix throughput: a workgroup owns 16 output rows x up to 128 columns and splits the reduction over its 8 waves
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains), partial tiles are summed through LDS in a fixed order.
// A operand: lane (m = l % 16, kq = l / 16) holds W[row0 + m][k0 + 4 kq .. + 3] (one 16-byte load per 16 k); the j-th
// MFMA of a step contracts k = k0 + 4 kq + j.  B operand: x[k0 + 4 kq + j][t0 + 16 n + l % 16].
struct G2PLinJob {
    const float *W;
    float *y;
    const float *res;
    int out, tiles;      // tiles = ceil(out / 16)
    int64_t y_rs, y_cs;  // output (and residual) element (row, column) at row * y_rs + column * y_cs
};
struct G2PLinArgs {
    G2PLinJob job[3];
    const float *x;
    int njobs, in, T, xp;
    int cb;              // 16-column blocks per workgroup (1 .. 8): narrower tiles = more workgroups on a short grid
};
typedef float g2p_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void g2p_linear_generic_kernel(G2PLinArgs a) {
    extern __shared__ float part[];  // [8 waves][8 blocks][4][64]
    int blk = blockIdx.x, jb = 0;
    while (jb + 1 < a.njobs && blk >= a.job[jb].tiles) blk -= a.job[jb++].tiles;
    const G2PLinJob &J = a.job[jb];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int row0 = blk * 16, t0 = blockIdx.y * (16 * a.cb);
    const int ncol = a.T - t0 < 16 * a.cb ? a.T - t0 : 16 * a.cb, nb = (ncol + 15) >> 4;
    const int steps = (a.in + 15) >> 4;                        // 16 k per step
    const int s0 = steps * wave / 8, s1 = steps * (wave + 1) / 8;
    const bool row_ok = row0 + m < J.out;
    const float *wrow = J.W + (int64_t)(row0 + m < J.out ? row0 + m : 0) * a.in;
    const bool vec_ok = (a.in & 3) == 0;
    g2p_f32x4 acc[8];
#pragma unroll
    for (int n = 0; n < 8; n++) acc[n] = g2p_f32x4{0.f, 0.f, 0.f, 0.f};
    // The weights are the stream that comes from HBM (each byte once per launch): a wave requests its whole share - up to
    // 16 steps, 64 registers - before anything else, so that it pays that latency once.  The activations (L2) follow in
    // groups of four steps.
    auto load_w = [&](float (&av)[4], int s) {
        const int k = s * 16 + 4 * kq;
        if (vec_ok && k + 4 <= a.in && row_ok) {
            const float4 w = *reinterpret_cast<const float4 *>(wrow + k);
            av[0] = w.x, av[1] = w.y, av[2] = w.z, av[3] = w.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) av[j] = row_ok && k + j < a.in ? wrow[k + j] : 0.f;
        }
    };
    auto load_x = [&](float (&bv)[4][8], int s) {
        const int k = s * 16 + 4 * kq;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float *xr = a.x + (int64_t)(k + j < a.in ? k + j : 0) * a.xp + t0 + m;
            const bool k_ok = k + j < a.in;
#pragma unroll
            for (int n = 0; n < 8; n++) bv[j][n] = n < nb && k_ok && 16 * n + m < ncol ? xr[16 * n] : 0.f;
        }
    };
    auto mma = [&](const float (&av)[4], const float (&bv)[4][8]) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int n = 0; n < 8; n++)
                if (n < nb) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j][n], acc[n], 0, 0, 0);
    };
    for (int g0 = s0; g0 < s1; g0 += 16) {
        float av[16][4];
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (g0 + u < s1) load_w(av[u], g0 + u);
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            if (g0 + 4 * q4 >= s1) break;
            float bv[4][4][8];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (g0 + 4 * q4 + u < s1) load_x(bv[u], g0 + 4 * q4 + u);
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (g0 + 4 * q4 + u < s1) mma(av[4 * q4 + u], bv[u]);
        }
    }
#pragma unroll
    for (int n = 0; n < 8; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) part[((wave * 8 + n) * 4 + r) * 64 + lane] = acc[n][r];
    __syncthreads();
    // element e = (n * 4 + r) * 64 + l: row = row0 + 4 (l / 16) + r, column = t0 + 16 n + l % 16
    for (int e = tid; e < nb * 256; e += 512) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; w++) v += part[w * 2048 + e];
        const int l = e & 63, r = (e >> 6) & 3, n = e >> 8;
        const int row = row0 + 4 * (l >> 4) + r, col = 16 * n + (l & 15);
        if (row < J.out && col < ncol) {
            const int64_t o = (int64_t)row * J.y_rs + (int64_t)(t0 + col) * J.y_cs;
            J.y[o] = J.res ? v + J.res[o] : v;
        }
    }
}

// The same kernel for reductions that are whole 16-k steps (every T5 size: d_model, d_ff and heads x d_kv are multiples of
// 16), CB = 16-column blocks per workgroup as a compile-time constant - and NO conditional load.  In the generic kernel above
// every load sits behind a lane-dependent condition (`row_ok && k + 4 <= in ? .. : 0`, `16 n + m < ncol ? x[..] : 0`), which
// compiles to a branch per load with `s_waitcnt vmcnt(0)` at its join: the "whole share of the weights requested at once"
// went out one load at a time, and so did the activations (the ISA: `L b b L b b L .. [vmcnt(0)] MFMA`, 31.6 us per launch at
// 80 columns where the weights stream in 2-8 us).  Here rows, steps and columns past the end are CLAMPED to the last valid one
// (re-read, harmless: such rows / columns are never stored, such steps never multiplied) and every load of a group is in
// flight before the group's first MFMA.  Same products in the same order: bit-identical to the generic kernel.
template <int CB>
__global__ __launch_bounds__(512) void g2p_linear_kernel(G2PLinArgs a) {
    extern __shared__ float part[];  // [8 waves][8 blocks][4][64]
    int blk = blockIdx.x, jb = 0;
    while (jb + 1 < a.njobs && blk >= a.job[jb].tiles) blk -= a.job[jb++].tiles;
    const G2PLinJob &J = a.job[jb];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int row0 = blk * 16, t0 = blockIdx.y * (16 * CB);
    const int ncol = a.T - t0 < 16 * CB ? a.T - t0 : 16 * CB, nb = (ncol + 15) >> 4;
    const int steps = a.in >> 4;                               // 16 k per step, no tail
    const int s0 = steps * wave / 8, s1 = steps * (wave + 1) / 8;
    const float *wrow = J.W + (int64_t)(row0 + m < J.out ? row0 + m : J.out - 1) * a.in + 4 * kq;
    int coff[CB];                                              // this lane's column of block n, clamped into the tensor
#pragma unroll
    for (int n = 0; n < CB; n++) coff[n] = t0 + (16 * n + m < ncol ? 16 * n + m : ncol - 1);
    g2p_f32x4 acc[CB];
#pragma unroll
    for (int n = 0; n < CB; n++) acc[n] = g2p_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int g0 = s0; g0 < s1; g0 += 16) {
        float4 av[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int sc = g0 + u < s1 ? g0 + u : s1 - 1;
            av[u] = *reinterpret_cast<const float4 *>(wrow + sc * 16);
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            if (g0 + 4 * q4 >= s1) break;                      // (wave-uniform)
            float bv[4][4][CB];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int sc = g0 + 4 * q4 + u < s1 ? g0 + 4 * q4 + u : s1 - 1;
                const float *xr = a.x + (int64_t)(sc * 16 + 4 * kq) * a.xp;
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int n = 0; n < CB; n++) bv[u][j][n] = xr[(int64_t)j * a.xp + coff[n]];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (g0 + 4 * q4 + u < s1) {                    // (wave-uniform; no load inside)
                    const float4 w = av[4 * q4 + u];
                    const float wj[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int n = 0; n < CB; n++) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wj[j], bv[u][j][n], acc[n], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int n = 0; n < CB; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) part[((wave * 8 + n) * 4 + r) * 64 + lane] = acc[n][r];
    __syncthreads();
    // element e = (n * 4 + r) * 64 + l: row = row0 + 4 (l / 16) + r, column = t0 + 16 n + l % 16
    for (int e = tid; e < nb * 256; e += 512) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; w++) v += part[w * 2048 + e];
        const int l = e & 63, r = (e >> 6) & 3, n = e >> 8;
        const int row = row0 + 4 * (l >> 4) + r, col = 16 * n + (l & 15);
        if (row < J.out && col < ncol) {
            const int64_t o = (int64_t)row * J.y_rs + (int64_t)(t0 + col) * J.y_cs;
            J.y[o] = J.res ? v + J.res[o] : v;
        }
    }
}

// The decoder step's matrix-vector products for NB sequences decoded side by side, up to three W per launch (q | k | v of
// one attention share their input):
//   y_j[b * yb_j + row * ys_j] = post * rs_b * sum_i W_j[row][i] g[i] x[b][i]  (+ res_j[same place])
//   rs_b = rsqrt(mean(x[b]^2) + eps) if g: the T5 RMS norm in front of the projection is folded in (every wave streams
// x anyway, so the sum of squares is free); with `W2` the two input projections of T5DenseGatedActDense are one job:
// y = act(W x') * (W2 x').  One wave per output row, four rows per workgroup; the weights stream ONCE for all NB
// sequences (float4 per lane), the NB input vectors [NB][in] come from L2.
struct G2PStepJob {
    const float *W, *W2;
    float *y;
    const float *res;
    int ys, yb, out, blocks;  // row stride, sequence stride; blocks = ceil(out / 4)
    int yt;                   // (persistent step) y is a cache row: the step's position t is added to it
};
struct G2PStepArgs {
    G2PStepJob job[3];
    const float *x, *g;
    int njobs, in, act;
    float eps, post;
};

__device__ __forceinline__ float g2p_activation(float x, int act) {
    if (act == 1) return fmaxf(x, 0.f);
    if (act == 2) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

#ifndef G2P_UN
#define G2P_UN 6
#endif
#ifndef G2P_UN_GATED
#define G2P_UN_GATED 3
#endif
template <int NB>
__device__ __forceinline__ void g2p_step_body(const G2PStepArgs &a, int blk, int t) {
    int j = 0;
    while (j + 1 < a.njobs && blk >= a.job[j].blocks) blk -= a.job[j++].blocks;
    const G2PStepJob &J = a.job[j];
    const int lane = threadIdx.x & 63, row = blk * 4 + (threadIdx.x >> 6);
    if (row >= J.out) return;
    const float4 *w4 = reinterpret_cast<const float4 *>(J.W + (int64_t)row * a.in);
    const float4 *v4 = J.W2 ? reinterpret_cast<const float4 *>(J.W2 + (int64_t)row * a.in) : nullptr;
    const float4 *x4 = reinterpret_cast<const float4 *>(a.x);
    const float4 *g4 = reinterpret_cast<const float4 *>(a.g);
    float s[NB], s2[NB], ss[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) s[b] = s2[b] = ss[b] = 0.f;
    if (a.in & 3) {  // rows not 16-byte aligned: one float per lane
        const float *w = J.W + (int64_t)row * a.in, *v = J.W2 ? J.W2 + (int64_t)row * a.in : nullptr;
        for (int i = lane; i < a.in; i += 64) {
            const float wi = w[i], vi = v ? v[i] : 0.f, gi = a.g ? a.g[i] : 1.f;
#pragma unroll
            for (int b = 0; b < NB; b++) {
                float x = a.x[(int64_t)b * a.in + i];
                ss[b] += x * x;
                x *= gi;
                s[b] += wi * x;
                s2[b] += vi * x;
            }
        }
    } else {
        // A row is a chain of 1 KiB wave loads (6 for d_model 1472, 14 for d_ff 3584): the loads of up to UN of them are
        // requested before the first FMA, so a row pays one or two memory latencies instead of one per KiB (the 3584-wide
        // output projection ran at 1.8 TB/s).  The loads are unconditional (a lane past the row's end re-reads the row's
        // last 16 bytes and its term is not added: a conditional load is a branch with a full wait at its join), and the
        // launch-uniform options - second matrix, norm weights - select one of four branch-free variants.  Same sums in the
        // same order (i ascending per lane).
        const int n4 = a.in >> 2;
        auto rows = [&](auto GATED, auto NORM) __attribute__((always_inline)) {
            constexpr bool gated = decltype(GATED)::value, norm = decltype(NORM)::value;
            // (two matrices double the loads and registers per iteration: fewer iterations in flight keep more waves resident)
            constexpr int UN1 = gated ? G2P_UN_GATED : G2P_UN;
            constexpr int UN = NB == 1 ? UN1 : (NB == 2 ? (UN1 + 1) / 2 : (UN1 + 3) / 4);
            for (int i0 = lane; i0 < n4; i0 += 64 * UN) {
                float4 w[UN], v[gated ? UN : 1], g[norm ? UN : 1], x[UN][NB];
#pragma unroll
                for (int u = 0; u < UN; u++) {
                    const int i = i0 + 64 * u, ic = i < n4 ? i : n4 - 1;
                    w[u] = w4[ic];
                    if constexpr (gated) v[u] = v4[ic];
                    if constexpr (norm) g[u] = g4[ic];
#pragma unroll
                    for (int b = 0; b < NB; b++) x[u][b] = x4[(int64_t)b * n4 + ic];
                }
#pragma unroll
                for (int u = 0; u < UN; u++) {
                    const bool ok = i0 + 64 * u < n4;
#pragma unroll
                    for (int b = 0; b < NB; b++) {
                        float4 xb = x[u][b];
                        const float q2 = xb.x * xb.x + xb.y * xb.y + xb.z * xb.z + xb.w * xb.w;
                        if constexpr (norm) xb.x *= g[u].x, xb.y *= g[u].y, xb.z *= g[u].z, xb.w *= g[u].w;
                        const float t1 = w[u].x * xb.x + w[u].y * xb.y + w[u].z * xb.z + w[u].w * xb.w;
                        ss[b] = ok ? ss[b] + q2 : ss[b];
                        s[b] = ok ? s[b] + t1 : s[b];
                        if constexpr (gated) {
                            const float t2 = v[u].x * xb.x + v[u].y * xb.y + v[u].z * xb.z + v[u].w * xb.w;
                            s2[b] = ok ? s2[b] + t2 : s2[b];
                        }
                    }
                }
            }
        };
        const std::true_type yes{};
        const std::false_type no{};
        if (v4) {
            if (g4) rows(yes, yes);
            else rows(yes, no);
        } else {
            if (g4) rows(no, yes);
            else rows(no, no);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s[b] += __shfl_xor(s[b], o, 64);
            s2[b] += __shfl_xor(s2[b], o, 64);
            ss[b] += __shfl_xor(ss[b], o, 64);
        }
    if (lane) return;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const float rs = (g4 ? 1.0f / sqrtf(ss[b] / (float)a.in + a.eps) : 1.0f) * a.post;
        float y = s[b] * rs;
        if (v4) y = g2p_activation(y, a.act) * (s2[b] * rs);
        else if (a.act >= 0) y = g2p_activation(y, a.act);
        const int64_t o = (int64_t)b * J.yb + (int64_t)row * J.ys + (int64_t)J.yt * t;
        J.y[o] = J.res ? y + J.res[o] : y;
    }
}
template <int NB>
__global__ __launch_bounds__(256) void g2p_step_kernel(G2PStepArgs a) {
    g2p_step_body<NB>(a, (int)blockIdx.x, 0);
}

// x[b][c] = table[ids[b * id_stride]][c]: the embeddings of the tokens the NB sequences decode next
__global__ void g2p_embed_rows_kernel(const int64_t *ids, int id_stride, const float *table, float *x, int C, int vocab) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    int64_t id = ids[(int64_t)b * id_stride];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    x[(int64_t)b * C + c] = table[id * C + c];
}

__global__ void g2p_scale_kernel(float *x, int64_t n, float s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= s;
}

// argmax over the vocabulary (first maximum, as np.argmax), one workgroup per sequence b = blockIdx.x:
// logits element v of sequence b at b * lb + v * pitch + t  ->  ids[b * ib + slot] (int64) and, if given, the same
// place of the pinned host copy
__device__ __forceinline__ void g2p_argmax_body(const float *logits, int V, int pitch, int t, int64_t lb, int64_t *ids, int64_t ib,
                                                int slot, int64_t *host_copy, int b, float *bv, int *bi) {
    const int tid = threadIdx.x;
    logits += (int64_t)b * lb;
    float best = -__builtin_inff();
    int idx = 0x7fffffff;
    for (int c = tid; c < V; c += 256) {
        const float v = logits[(int64_t)c * pitch + t];
        if (v > best || (v == best && c < idx)) {
            best = v;
            idx = c;
        }
    }
    bv[tid] = best;
    bi[tid] = idx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o && (bv[tid + o] > bv[tid] || (bv[tid + o] == bv[tid] && bi[tid + o] < bi[tid]))) {
            bv[tid] = bv[tid + o];
            bi[tid] = bi[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        ids[(int64_t)b * ib + slot] = bi[0];
        if (host_copy) host_copy[(int64_t)b * ib + slot] = bi[0];  // (pinned, mapped: visible once the step's event has fired)
    }
}
__global__ __launch_bounds__(256) void g2p_argmax_kernel(const float *logits, int V, int pitch, int t, int64_t lb, int64_t *ids,
                                                         int64_t ib, int slot, int64_t *host_copy) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    g2p_argmax_body(logits, V, pitch, t, lb, ids, ib, slot, host_copy, (int)blockIdx.x, bv, bi);
}

// ---- the decoder step as ONE launch ------------------------------------------------------------------------------------
// A step of the narrow path is 3 + 8 * layers launches of 5-30 us of work each, so the step's time is mostly the gaps
// between them.  The persistent form runs the same bodies, in the same order, from a phase table in device memory:
// every workgroup walks the table, takes the virtual blocks `vb = blockIdx.x, + gridDim.x, ...` of each phase, and meets
// the others at a grid barrier between phases.  The arithmetic (and so every generated token) is the launch-per-phase
// path's, bit for bit: the bodies are shared and a virtual block computes exactly what the real block did.
//
// Grid barrier: one counter that only grows.  Launch t, phase p waits for (t * nph + p + 1) * gridDim.x arrivals (every
// launch makes exactly nph * gridDim.x, so no reset between launches).  Thread 0 of a workgroup: release fence (L2
// write-back: the XCDs' L2s are not coherent with each other), atomic add, spin with s_sleep on a device-scope load,
// acquire fence.  The grid is clamped to what the occupancy calculator says is co-resident; a barrier that has not
// completed after ~2 s of the constant 100 MHz clock raises `bar[1]` and every workgroup leaves - a wrong launch can
// fail a call but never hang the device.
enum { G2P_PH_EMBED = 0, G2P_PH_STEP = 1, G2P_PH_ATT = 2, G2P_PH_ARGMAX = 3 };
struct G2PPhase {
    int kind, nblocks;
    G2PStepArgs step;  // STEP (job.yt set where the output is a cache column of the step's position)
    // ATT: the arguments of g2p_attention_kernel; self_att: Tk = t + 1 and q_off = t
    const float *q, *k, *v, *bias;
    const int *lut, *lens;
    float *out;
    int q_cs, q_ts, kp, lut_zero, seg, Tk, causal, self_att, heads, dk, cols;
    int64_t kv_bs;
    // EMBED: x[b][c] = table[ids[b * ib + t]][c];  ARGMAX: logits [b][V] -> ids[b * ib + t + 1]
    int64_t *ids, *host_copy;
    const float *table;
    float *x;
    int C, vocab;
    int64_t ib;
};

// `ep`: the barrier's index since the counters were cleared (launch t, phase p: t * nph + p).  Two levels: a workgroup
// arrives on its group's counter (groups of `gs` workgroups, a cache line each), the group's last arrival on the top
// counter - same-address device-scope atomics serialise at ~0.1 us each, and a flat counter cost 26 us per barrier at
// 256 workgroups.
__device__ __forceinline__ bool g2p_grid_barrier(unsigned *bar, unsigned ep, int gs, bool wait) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned grp = blockIdx.x / gs, ngrp = (gridDim.x + gs - 1) / gs;
        const unsigned gsize = grp + 1 < ngrp ? gs : gridDim.x - grp * gs;
        __threadfence();  // release: this workgroup's stores are in L2 (the __syncthreads above); write them back
        const unsigned old = __hip_atomic_fetch_add(bar + 32 * (grp + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == (ep + 1) * gsize) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        if (wait) {
            const unsigned target = (ep + 1) * ngrp;
            const uint64_t w0 = wall_clock64();
            while ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                    wall_clock64() - w0 > 200000000ull) {  // 2 s at 100 MHz
                    __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __threadfence();  // acquire: drop this CU's L1 and the L2's lines of other XCDs' data
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

template <int NB>
__global__ __launch_bounds__(256) void g2p_decode_step_kernel(const G2PPhase *__restrict__ ph, int nph, unsigned *bar, int t,
                                                              int64_t *gave_up, int gs) {
    extern __shared__ float sc[];
    __shared__ float red[256];
    __shared__ int redi[256];
    if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {  // (an earlier step gave up)
        if (blockIdx.x == 0 && threadIdx.x == 0) *gave_up = 1;
        return;
    }
    for (int p = 0; p < nph; p++) {
        const G2PPhase &P = ph[p];
        const int kind = P.kind, nblocks = P.nblocks;
        for (int vb = blockIdx.x; vb < nblocks; vb += gridDim.x) {
            if (kind == G2P_PH_STEP) {
                g2p_step_body<NB>(P.step, vb, t);
            } else if (kind == G2P_PH_ATT) {
                __syncthreads();  // (sc / red of the previous virtual block)
                g2p_attention_body(P.q, P.q_cs, P.q_ts, P.k, P.v, P.kp, P.kv_bs, P.bias, P.lut, P.lut_zero, P.out, P.heads, P.dk,
                                   P.seg, P.lens, P.self_att ? t + 1 : P.Tk, P.self_att ? t : 0, P.causal, vb % P.cols, vb / P.cols,
                                   sc, red);
            } else if (kind == G2P_PH_EMBED) {
                const int per = (P.C + 255) / 256, b = vb / per, c = (vb - b * per) * 256 + (int)threadIdx.x;
                if (c < P.C) {
                    int64_t id = P.ids[(int64_t)b * P.ib + t];
                    id = id < 0 ? 0 : (id >= P.vocab ? P.vocab - 1 : id);
                    P.x[(int64_t)b * P.C + c] = P.table[id * P.C + c];
                }
            } else {
                __syncthreads();
                g2p_argmax_body(P.step.x, P.vocab, 1, 0, P.vocab, P.ids, P.ib, t + 1, P.host_copy, vb, red, redi);
            }
        }
        if (!g2p_grid_barrier(bar, (unsigned)t * (unsigned)nph + (unsigned)p, gs, p + 1 < nph)) {
            if (threadIdx.x == 0) *gave_up = 1;  // (pinned: the host reads it after the step's event)
            return;
        }
    }
}

// logits [V][T] (channels-first) -> [T][V] (what the graph returns)
__global__ void g2p_transpose_kernel(const float *in, float *out, int V, int T) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (c < V) out[(int64_t)t * V + c] = in[(int64_t)c * T + t];
}

}  // namespace

struct g2p_handle {
    G2PModel model;
    int device = -1;
    hipStream_t stream = nullptr;
    float *arena_dev = nullptr;
    int *d_bucket_enc = nullptr, *d_bucket_dec = nullptr;
    char *ws = nullptr;
    size_t ws_cap = 0;
    int64_t *tok_host = nullptr, *tok_dev = nullptr;  // generated ids, pinned + mapped: the step's argmax writes them
    hipEvent_t step_done[2] = {nullptr, nullptr};
    std::mutex mu;
    std::string err;
};

namespace {

int gfail(g2p_handle *h, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_g2p_open_error = buf;
    return code;
}

struct Run {
    g2p_handle *h;
    hipStream_t st;
    const float *A;
    hipError_t err = hipSuccess;
    char *ws;
    size_t used = 0;
    const float *P(int64_t off) const { return A + off; }
    void note(hipError_t e) {
        if (err == hipSuccess && e != hipSuccess) err = e;
    }
    template <class Tp>
    Tp *take(size_t n) {
        size_t off = (used + 255) & ~size_t(255);
        used = off + n * sizeof(Tp);
        return reinterpret_cast<Tp *>(ws + off);
    }
};

// y_j[out_j][T] (pitch yp) = W_j x (x [in][T], pitch xp) [+ res_j], up to three W over the same x per launch
struct LinJob {
    const T5Linear *L;
    float *y;
    const float *res;
    int64_t y_rs = 0, y_cs = 1;  // (0: the call's default row stride)
};
void linear(Run &r, std::initializer_list<LinJob> jobs, const float *x, int xp, int T, int64_t y_rs) {
    // (the attribute belongs to the (function, device) pair: handles on different GPUs each set it, and no flag is
    // shared between their threads; the call is idempotent and costs ~1 us)
    G2PLinArgs a{};
    int tiles = 0;
    for (const LinJob &j : jobs) {
        G2PLinJob &d = a.job[a.njobs++];
        d.W = r.P(j.L->rowmajor);
        d.y = j.y;
        d.res = j.res;
        d.out = j.L->out;
        d.tiles = (j.L->out + 15) / 16;
        d.y_rs = j.y_rs ? j.y_rs : y_rs;
        d.y_cs = j.y_cs;
        tiles += d.tiles;
        a.in = j.L->in;
    }
    a.x = x;
    a.T = T;
    a.xp = xp;
    // column tile: 128 wide.  (Narrower tiles on short grids - 32 columns at T = 80 - were measured slower: a
    // workgroup's time is the latency of its weight rows, not its matrix work.)
    a.cb = 8;
    // The branch-free kernel serves the short column counts - the decoder step of 8 .. 32 sequences side by side: 0.67 -> 0.44 ms
    // per step at 8 and 16 sequences - and loses to the generic one on wide tiles (encoder over 80 bytes x 4 inputs 4.4 -> 5.8 ms,
    // x 64 inputs 48 -> 59 ms: at 5-8 column blocks its 190-230 registers and ~150 loads in flight per lane cost more than the
    // generic kernel's serialised ones; profiles/r06_runs/g2p_linear_ab.txt), so those keep the generic kernel.
    bool whole = a.in % 16 == 0 && a.in >= 128 && T <= 32;  // (whole steps; every wave has at least one)
    for (const LinJob &j : jobs) whole = whole && j.L->in == a.in;
    static const bool generic_only = std::getenv("VITSMI_G2P_LINEAR_GENERIC") != nullptr;  // (A/B)
    if (!whole || generic_only) {
        r.note(hipFuncSetAttribute(reinterpret_cast<const void *>(g2p_linear_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   64 * 1024));
        g2p_linear_generic_kernel<<<dim3(tiles, (T + 16 * a.cb - 1) / (16 * a.cb)), 512, 64 * 1024, r.st>>>(a);
        return;
    }
    const int cb = (T + 15) / 16;  // 1 or 2
    a.cb = cb;
    const dim3 grid(tiles, 1);
    if (cb == 1) {
        r.note(hipFuncSetAttribute(reinterpret_cast<const void *>(g2p_linear_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        g2p_linear_kernel<1><<<grid, 512, 64 * 1024, r.st>>>(a);
    } else {
        r.note(hipFuncSetAttribute(reinterpret_cast<const void *>(g2p_linear_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        g2p_linear_kernel<2><<<grid, 512, 64 * 1024, r.st>>>(a);
    }
}
void linear(Run &r, const T5Linear &L, const float *x, int xp, int T, float *y, int yp, const float *res = nullptr) {
    linear(r, {{&L, y, res}}, x, xp, T, yp);
}

// One launch of the decoder step for nb sequences (see g2p_step_kernel): `g` = RMS-norm weight folded in front (or -1).
// y / res element (sequence b, row) at b * yb + row * ys.
struct StepJob {
    const T5Linear *L, *gate;
    float *y;
    int ys, yb;
    const float *res;
};
void step(Run &r, int nb, std::initializer_list<StepJob> jobs, const float *x, int64_t g, int act = -1, float post = 1.f) {
    G2PStepArgs a{};
    int blocks = 0;
    for (const StepJob &j : jobs) {
        G2PStepJob &d = a.job[a.njobs++];
        d.W = r.P(j.L->rowmajor);
        d.W2 = j.gate ? r.P(j.gate->rowmajor) : nullptr;
        d.y = j.y;
        d.ys = j.ys;
        d.yb = j.yb;
        d.res = j.res;
        d.out = j.L->out;
        d.blocks = (j.L->out + 3) / 4;
        blocks += d.blocks;
        a.in = j.L->in;
    }
    a.x = x;
    a.g = g >= 0 ? r.P(g) : nullptr;
    a.act = act;
    a.eps = r.h->model.eps;
    a.post = post;
    switch (nb) {
        case 1: g2p_step_kernel<1><<<blocks, 256, 0, r.st>>>(a); break;
        case 2: g2p_step_kernel<2><<<blocks, 256, 0, r.st>>>(a); break;
        case 4: g2p_step_kernel<4><<<blocks, 256, 0, r.st>>>(a); break;
        default: g2p_step_kernel<4><<<blocks, 256, 0, r.st>>>(a); break;
    }
}

void rmsnorm(Run &r, const float *x, int xp, int64_t g, float *y, int yp, int T) {
    const G2PModel &m = r.h->model;
    g2p_rmsnorm_kernel<<<T, 256, 0, r.st>>>(x, r.P(g), y, m.d_model, T, xp, yp, m.eps);
}

// attention over `cols` query columns in sequences of `seg` (see g2p_attention_kernel for the strides)
void attention(Run &r, const float *q, int q_cs, int q_ts, const float *k, const float *v, int kp, int64_t kv_bs, int64_t bias,
               const int *lut, float *out, int cols, int seg, const int *lens, int Tk, int q_off, bool causal) {
    const G2PModel &m = r.h->model;
    g2p_attention_kernel<<<dim3(cols, m.heads), 256, (size_t)(Tk + m.d_kv) * sizeof(float), r.st>>>(
        q, q_cs, q_ts, k, v, kp, kv_bs, bias >= 0 ? r.P(bias) : nullptr, lut, G2PModel::kMaxPos - 1, out, m.heads, m.d_kv, seg,
        lens, Tk, q_off, causal ? 1 : 0);
}

void ffn(Run &r, const T5FfnDesc &f, const float *hn, float *x, int T, float *a, float *b) {
    const G2PModel &m = r.h->model;
    if (f.gated) linear(r, {{&f.wi0, a, nullptr}, {&f.wi1, b, nullptr}}, hn, T, T, T);
    else linear(r, f.wi0, hn, T, T, a, T);
    const int64_t n = (int64_t)m.d_ff * T;
    g2p_act_kernel<<<(unsigned)((n + 255) / 256), 256, 0, r.st>>>(a, f.gated ? b : nullptr, a, n, m.act);
    linear(r, f.wo, a, T, T, x, T, x);  // x += wo(h)
}

// OPT-IN (VITSMI_G2P_PERSIST=1).  Measured on the ByT5-small shape, one sequence, ms per token: a launch per phase 0.307;
// persistent, 256 workgroups: flat barrier 0.92, two-level 0.70, two-level without the fences (wrong tokens - the XCDs'
// L2s need them) 0.45; 512 workgroups 0.91 / 0.39.  A barrier costs more than the command processor's gap between two
// dependent launches (~6 us, which already includes the same L2 write-back and invalidate), so the default stays a
// launch per phase; the persistent form is kept for the measurement and is token-identical.
bool g2p_persist_on() {
    const char *e = std::getenv("VITSMI_G2P_PERSIST");
    return e && e[0] == '1';
}

int ws_reserve(g2p_handle *h, size_t bytes) {
    if (bytes <= h->ws_cap) return 0;
    if (h->ws) {
        hipStreamSynchronize(h->stream);
        hipFree(h->ws);
        h->ws = nullptr;
        h->ws_cap = 0;
    }
    const size_t want = bytes + bytes / 4 + (1 << 20);
    if (hipMalloc((void **)&h->ws, want) != hipSuccess) return gfail(h, VITS_E_NOMEM, "hipMalloc(%zu) failed", want);
    h->ws_cap = want;
    return 0;
}

// encoder over nseq sequences of S ids each (device, zero-padded; true lengths in d_lens, or nullptr = all S)
// -> enc_out [d_model][nseq * S]; a padded column attends like the others and is never attended to
void run_encoder(Run &r, const int64_t *d_ids, int S, int nseq, const int *d_lens, float *x, float *hn, float *q, float *k,
                 float *v, float *att, float *fa, float *fb) {
    g2p_handle *h = r.h;
    const G2PModel &m = h->model;
    const int T = S * nseq;
    g2p_embed_kernel<<<dim3((m.d_model + 255) / 256, T), 256, 0, r.st>>>(d_ids, 1, r.P(m.shared), x, m.d_model, T, T, m.vocab);
    for (const auto &b : m.enc) {
        rmsnorm(r, x, T, b.ln_self, hn, T, T);
        linear(r, {{&b.self.q, q, nullptr}, {&b.self.k, k, nullptr}, {&b.self.v, v, nullptr}}, hn, T, T, T);
        attention(r, q, T, 1, k, v, T, S, m.enc_bias, h->d_bucket_enc, att, T, S, d_lens, S, 0, false);
        linear(r, b.self.o, att, T, T, x, T, x);
        rmsnorm(r, x, T, b.ln_ffn, hn, T, T);
        ffn(r, b.ffn, hn, x, T, fa, fb);
    }
    rmsnorm(r, x, T, m.enc_final_ln, x, T, T);
}

int check_dev(g2p_handle *h) {
    if (!h) return VITS_E_ARG;
    if (h->device < 0) return gfail(h, VITS_E_DEVICE, "handle was opened host-only");
    if (hipSetDevice(h->device) != hipSuccess) return gfail(h, VITS_E_DEVICE, "hipSetDevice(%d) failed", h->device);
    return 0;
}

int check_ids(g2p_handle *h, const int64_t *ids, int n, const char *what) {
    for (int i = 0; i < n; i++)
        if (ids[i] < 0 || ids[i] >= h->model.vocab)
            return gfail(h, VITS_E_ARG, "%s[%d]=%lld is out of range [0,%d)", what, i, (long long)ids[i], h->model.vocab);
    return 0;
}

}  // namespace

extern "C" {

int g2p_open(const char *path, int device, g2p_handle **out) {
    if (!out || !path) return gfail(nullptr, VITS_E_ARG, "null argument");
    *out = nullptr;
    OnnxModel om;
    std::string e = om.load(path);
    if (!e.empty()) return gfail(nullptr, e.rfind("cannot open", 0) == 0 ? VITS_E_IO : VITS_E_FORMAT, "%s", e.c_str());
    g2p_handle *h = new g2p_handle();
    e = h->model.build(om);
    if (!e.empty()) {
        delete h;
        return gfail(nullptr, VITS_E_FORMAT, "%s: %s", path, e.c_str());
    }
    if (device >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device >= n) {
            delete h;
            return gfail(nullptr, VITS_E_DEVICE, "no usable HIP device %d", device);
        }
        h->device = device;
        const size_t bytes = h->model.arena.size() * 4, lut = h->model.bucket_enc.size() * sizeof(int);
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void **)&h->arena_dev, bytes) != hipSuccess ||
            hipMemcpy(h->arena_dev, h->model.arena.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMalloc((void **)&h->d_bucket_enc, lut) != hipSuccess || hipMalloc((void **)&h->d_bucket_dec, lut) != hipSuccess ||
            hipMemcpy(h->d_bucket_enc, h->model.bucket_enc.data(), lut, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_bucket_dec, h->model.bucket_dec.data(), lut, hipMemcpyHostToDevice) != hipSuccess ||
            hipHostMalloc((void **)&h->tok_host, ((size_t)G2P_MAX_BATCH * G2PModel::kMaxPos + 8) * sizeof(int64_t), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&h->tok_dev, h->tok_host, 0) != hipSuccess ||
            hipEventCreateWithFlags(&h->step_done[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->step_done[1], hipEventDisableTiming) != hipSuccess) {
            g2p_close(h);
            return gfail(nullptr, VITS_E_NOMEM, "cannot place the G2P model on device %d", device);
        }
        std::vector<float>().swap(h->model.arena);  // the host copy is not needed again
    }
    *out = h;
    return VITS_OK;
}

void g2p_close(g2p_handle *h) {
    if (!h) return;
    if (h->device >= 0) {
        hipSetDevice(h->device);
        if (h->stream) hipStreamSynchronize(h->stream);
        if (h->arena_dev) hipFree(h->arena_dev);
        if (h->d_bucket_enc) hipFree(h->d_bucket_enc);
        if (h->d_bucket_dec) hipFree(h->d_bucket_dec);
        if (h->ws) hipFree(h->ws);
        if (h->tok_host) hipHostFree(h->tok_host);
        for (hipEvent_t e : h->step_done)
            if (e) hipEventDestroy(e);
        if (h->stream) hipStreamDestroy(h->stream);
    }
    delete h;
}

const char *g2p_last_error(g2p_handle *h) { return h ? h->err.c_str() : g_g2p_open_error.c_str(); }

int g2p_hparam(g2p_handle *h, const char *key, int64_t *out) {
    if (!h || !key || !out) return VITS_E_ARG;
    const G2PModel &m = h->model;
    const std::string k = key;
    if (k == "vocab") *out = m.vocab;
    else if (k == "d_model") *out = m.d_model;
    else if (k == "heads") *out = m.heads;
    else if (k == "d_kv") *out = m.d_kv;
    else if (k == "d_ff") *out = m.d_ff;
    else if (k == "n_enc") *out = (int64_t)m.enc.size();
    else if (k == "n_dec") *out = (int64_t)m.dec.size();
    else if (k == "num_buckets") *out = m.num_buckets;
    else if (k == "max_distance") *out = m.max_distance;
    else if (k == "gated") *out = m.enc[0].ffn.gated ? 1 : 0;
    else if (k == "act") *out = m.act;
    else if (k == "scale_out") *out = m.scale_out ? 1 : 0;
    else if (k == "max_positions") *out = G2PModel::kMaxPos;
    else return gfail(h, VITS_E_ARG, "unknown hparam %s", key);
    return VITS_OK;
}

int g2p_num_outputs(g2p_handle *h) { return h ? (int)(h->model.output_names.empty() ? 1 : h->model.output_names.size()) : 0; }
const char *g2p_output_name(g2p_handle *h, int i) {
    if (!h || i < 0) return nullptr;
    if (h->model.output_names.empty()) return i == 0 ? "logits" : nullptr;
    return i < (int)h->model.output_names.size() ? h->model.output_names[i].c_str() : nullptr;
}

int g2p_bucket(g2p_handle *h, int decoder, int rel) {
    if (!h || rel <= -G2PModel::kMaxPos || rel >= G2PModel::kMaxPos) return VITS_E_ARG;
    return (decoder ? h->model.bucket_dec : h->model.bucket_enc)[rel + G2PModel::kMaxPos - 1];
}

int g2p_run(g2p_handle *h, const int64_t *input_ids, int S, const int64_t *mask, const int64_t *dec_ids, int T, float *logits) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    const G2PModel &m = h->model;
    if (!input_ids || !dec_ids || !logits || S <= 0 || T <= 0) return gfail(h, VITS_E_ARG, "bad g2p_run arguments");
    if (S >= G2PModel::kMaxPos || T >= G2PModel::kMaxPos) return gfail(h, VITS_E_ARG, "sequence longer than %d", G2PModel::kMaxPos - 1);
    if (int rc = check_ids(h, input_ids, S, "input_ids")) return rc;
    if (int rc = check_ids(h, dec_ids, T, "decoder_input_ids")) return rc;
    if (mask) {
        // mul.py:187 passes all ones.  A right-padded mask (ones, then zeros) is honoured by dropping the padding: no
        // position attends to a masked key and the masked positions' own outputs are not part of the result, so the
        // logits are those of the unpadded input.  Masks with holes are refused.
        int n1 = 0;
        while (n1 < S && mask[n1] == 1) n1++;
        for (int i = n1; i < S; i++)
            if (mask[i] != 0) return gfail(h, VITS_E_ARG, "attention_mask must be ones followed by zeros (right padding)");
        if (n1 == 0) return gfail(h, VITS_E_ARG, "attention_mask masks every input position");
        S = n1;
    }
    const int L = S > T ? S : T;
    const size_t per = (size_t)(m.d_model > m.inner ? m.d_model : m.inner) * L * 4 + 512;
    const size_t need = 8 * per + 2 * ((size_t)m.d_ff * L * 4 + 512) + (size_t)m.vocab * T * 8 + (size_t)(S + T) * 8 + 4096 +
                        2 * ((size_t)m.inner * S * 4 + 512);
    if (int rc = ws_reserve(h, need)) return rc;
    Run r{h, h->stream, h->arena_dev};
    r.ws = h->ws;
    int64_t *d_in = r.take<int64_t>(S), *d_dec = r.take<int64_t>(T);
    const size_t nA = (size_t)(m.d_model > m.inner ? m.d_model : m.inner) * L;
    float *xe = r.take<float>(nA), *xd = r.take<float>(nA), *hn = r.take<float>(nA), *q = r.take<float>(nA);
    float *k = r.take<float>(nA), *v = r.take<float>(nA), *att = r.take<float>(nA);
    float *fa = r.take<float>((size_t)m.d_ff * L), *fb = r.take<float>((size_t)m.d_ff * L);
    float *kc = r.take<float>((size_t)m.inner * S), *vc = r.take<float>((size_t)m.inner * S);
    float *lg = r.take<float>((size_t)m.vocab * T), *lgt = r.take<float>((size_t)m.vocab * T);
    hipStream_t st = h->stream;
    r.note(hipMemcpyAsync(d_in, input_ids, (size_t)S * 8, hipMemcpyHostToDevice, st));
    r.note(hipMemcpyAsync(d_dec, dec_ids, (size_t)T * 8, hipMemcpyHostToDevice, st));
    run_encoder(r, d_in, S, 1, nullptr, xe, hn, q, k, v, att, fa, fb);
    // decoder over the whole prefix (teacher forced, causal)
    g2p_embed_kernel<<<dim3((m.d_model + 255) / 256, T), 256, 0, st>>>(d_dec, 1, r.P(m.shared), xd, m.d_model, T, T, m.vocab);
    for (const auto &b : m.dec) {
        rmsnorm(r, xd, T, b.ln_self, hn, T, T);
        linear(r, {{&b.self.q, q, nullptr}, {&b.self.k, k, nullptr}, {&b.self.v, v, nullptr}}, hn, T, T, T);
        attention(r, q, T, 1, k, v, T, 0, m.dec_bias, h->d_bucket_dec, att, T, T, nullptr, T, 0, true);
        linear(r, b.self.o, att, T, T, xd, T, xd);
        rmsnorm(r, xd, T, b.ln_cross, hn, T, T);
        linear(r, b.cross.q, hn, T, T, q, T);
        linear(r, {{&b.cross.k, kc, nullptr}, {&b.cross.v, vc, nullptr}}, xe, S, S, S);
        attention(r, q, T, 1, kc, vc, S, 0, -1, nullptr, att, T, T, nullptr, S, 0, false);
        linear(r, b.cross.o, att, T, T, xd, T, xd);
        rmsnorm(r, xd, T, b.ln_ffn, hn, T, T);
        ffn(r, b.ffn, hn, xd, T, fa, fb);
    }
    rmsnorm(r, xd, T, m.dec_final_ln, xd, T, T);
    if (m.scale_out) {
        const int64_t n = (int64_t)m.d_model * T;
        g2p_scale_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(xd, n, 1.0f / std::sqrt((float)m.d_model));
    }
    linear(r, m.lm_head, xd, T, T, lg, T);
    g2p_transpose_kernel<<<dim3((m.vocab + 255) / 256, T), 256, 0, st>>>(lg, lgt, m.vocab, T);
    r.note(hipGetLastError());
    r.note(hipMemcpyAsync(logits, lgt, (size_t)m.vocab * T * 4, hipMemcpyDeviceToHost, st));
    r.note(hipStreamSynchronize(st));
    if (r.err != hipSuccess) return gfail(h, VITS_E_DEVICE, "g2p_run failed: %s", hipGetErrorString(r.err));
    return VITS_OK;
}

// forced / step_logits (g2p_test_forced_steps): the decoder inputs are GIVEN - forced[b][t], t < max_length, forced[b][0] the
// start token - instead of fed back from the argmax, and the logits every step computes, [max_length][NB][vocab], are copied
// out: the step path (matrix-vector kernels, one-query attention over the caches) as a function that can be compared with
// g2p_run's logits for the same decoder_input_ids, number by number.  Narrow steps only (B <= 4).
static int generate_impl(g2p_handle *h, const int64_t *input_ids, const int *lens, int B, int max_length, int64_t start_id,
                         int64_t eos_id, int64_t *out_ids, int *n_out, const int64_t *forced, float *step_logits) {
    if (int rc = check_dev(h)) return rc;
    std::lock_guard<std::mutex> lk(h->mu);
    const G2PModel &m = h->model;
    if (!input_ids || !lens || !out_ids || !n_out || B <= 0 || max_length <= 0) return gfail(h, VITS_E_ARG, "bad g2p_generate arguments");
    if (forced && (!step_logits || B > 4)) return gfail(h, VITS_E_ARG, "forced steps: at most 4 sequences, logits required");
    if (B > G2P_MAX_BATCH) return gfail(h, VITS_E_ARG, "at most %d sequences per call", G2P_MAX_BATCH);
    if (max_length >= G2PModel::kMaxPos) return gfail(h, VITS_E_ARG, "sequence longer than %d", G2PModel::kMaxPos - 1);
    if (start_id < 0 || start_id >= m.vocab) return gfail(h, VITS_E_ARG, "start id out of range");
    int S = 0, total = 0;
    for (int b = 0; b < B; b++) {
        if (lens[b] <= 0 || lens[b] >= G2PModel::kMaxPos) return gfail(h, VITS_E_ARG, "input %d: length %d not in [1, %d)", b, lens[b], G2PModel::kMaxPos);
        S = lens[b] > S ? lens[b] : S;
        total += lens[b];
    }
    if (int rc = check_ids(h, input_ids, total, "input_ids")) return rc;
    // NB sequences run side by side (B rounded up to a size the step kernel is instantiated for; the spare ones decode
    // a one-token input and are ignored)
    int NB = 1;
    while (NB < B && NB < 4) NB *= 2;
    if (B > 4) NB = (B + 7) / 8 * 8;
    // Up to 4 sequences: the fused matrix-vector step (norm and gate folded in, 8 launches per layer).  More: the
    // sequences become the columns of [C][NB] activations and the step runs on the short-sequence kernels the encoder
    // uses (12 launches per layer, but one MFMA column per sequence instead of NB dot products per weight).
    const bool wide = NB > 4;
    const int TM = max_length + 1;  // decoder positions: the start token + every generated one
    const int T = S * NB;
    const size_t nA = (size_t)(m.d_model > m.inner ? m.d_model : m.inner) * T;
    const int nd = (int)m.dec.size();
    const size_t need = 8 * (nA * 4 + 512) + 2 * ((size_t)m.d_ff * T * 4 + 512) +
                        (size_t)nd * 2 * ((size_t)NB * m.inner * TM * 4 + 512) + (size_t)nd * 2 * ((size_t)m.inner * T * 4 + 512) +
                        (size_t)(T + (size_t)NB * TM) * 8 + (size_t)NB * (m.vocab + 2 * m.d_model + 2 * m.inner + 2 * m.d_ff + 64) * 4 +
                        (size_t)(3 + 8 * nd) * sizeof(G2PPhase) + (1 << 18) +
                        (forced ? (size_t)NB * TM * 8 + (size_t)max_length * NB * m.vocab * 4 + 1024 : 0);
    if (int rc = ws_reserve(h, need)) return rc;
    Run r{h, h->stream, h->arena_dev};
    r.ws = h->ws;
    hipStream_t st = h->stream;
    int64_t *d_in = r.take<int64_t>(T), *d_gen = r.take<int64_t>((size_t)NB * TM);
    int *d_lens = r.take<int>(NB);
    float *xe = r.take<float>(nA), *hn = r.take<float>(nA), *q = r.take<float>(nA), *k = r.take<float>(nA);
    float *v = r.take<float>(nA), *att = r.take<float>(nA);
    float *fa = r.take<float>((size_t)m.d_ff * T), *fb = r.take<float>((size_t)m.d_ff * T);
    std::vector<float *> ks(nd), vs(nd), kc(nd), vc(nd);
    for (int l = 0; l < nd; l++) {
        ks[l] = r.take<float>((size_t)NB * m.inner * TM);  // self-attention cache [NB][inner][TM]
        vs[l] = r.take<float>((size_t)NB * m.inner * TM);
        kc[l] = r.take<float>((size_t)m.inner * T);        // cross-attention keys / values of the encoder output [inner][NB * S]
        vc[l] = r.take<float>((size_t)m.inner * T);
    }
    float *x1 = r.take<float>((size_t)NB * m.d_model), *q1 = r.take<float>((size_t)NB * m.inner);
    float *a1 = r.take<float>((size_t)NB * m.inner), *f1 = r.take<float>((size_t)NB * m.d_ff), *lg = r.take<float>((size_t)NB * m.vocab);
    float *h1 = r.take<float>((size_t)NB * m.d_model), *f2 = r.take<float>((size_t)NB * m.d_ff);  // (wide step only)
    // forced steps: the argmax lands in a scratch copy of the id table (the given inputs stay), every step's logits are kept
    int64_t *d_arg = forced ? r.take<int64_t>((size_t)NB * TM) : d_gen;
    float *d_steplog = forced ? r.take<float>((size_t)max_length * NB * m.vocab) : nullptr;
    // The narrow step as one persistent launch (see g2p_decode_step_kernel), opt-in: VITSMI_G2P_PERSIST=1.
    const float post = m.scale_out ? 1.0f / std::sqrt((float)m.d_model) : 1.0f;
    const int D = m.d_model, I = m.inner;
    const int64_t cache_bs = (int64_t)I * TM;
    const bool persist = !wide && g2p_persist_on();
    const int nph = 3 + 8 * nd;
    G2PPhase *d_ph = persist ? r.take<G2PPhase>(nph) : nullptr;
    const size_t nbar = 32 * (1 + 4096 / 4);  // top counter + give-up flag, then a line per group
    unsigned *d_bar = persist ? r.take<unsigned>(nbar) : nullptr;
    int persist_gs = 32;
    if (const char *e = std::getenv("VITSMI_G2P_PERSIST_GROUP")) persist_gs = std::atoi(e) >= 4 ? std::atoi(e) : persist_gs;
    int64_t *gave_up_host = h->tok_host + (size_t)G2P_MAX_BATCH * G2PModel::kMaxPos, *gave_up_dev = h->tok_dev + (size_t)G2P_MAX_BATCH * G2PModel::kMaxPos;
    std::vector<G2PPhase> phases;
    int persist_grid = 0;
    const size_t persist_lds = (size_t)((S > TM ? S : TM) + m.d_kv) * sizeof(float);
    if (persist) {
        auto step_phase = [&](std::initializer_list<StepJob> jobs, const float *x, int64_t g, int act, float pst, int yt_from) {
            G2PPhase P{};
            P.kind = G2P_PH_STEP;
            G2PStepArgs &a = P.step;
            int jn = 0;
            for (const StepJob &j : jobs) {
                G2PStepJob &d = a.job[a.njobs++];
                d.W = r.P(j.L->rowmajor);
                d.W2 = j.gate ? r.P(j.gate->rowmajor) : nullptr;
                d.y = j.y;
                d.ys = j.ys;
                d.yb = j.yb;
                d.res = j.res;
                d.out = j.L->out;
                d.blocks = (j.L->out + 3) / 4;
                d.yt = jn >= yt_from ? 1 : 0;  // (k | v of the step's position: column t of the cache)
                P.nblocks += d.blocks;
                a.in = j.L->in;
                jn++;
            }
            a.x = x;
            a.g = g >= 0 ? r.P(g) : nullptr;
            a.act = act;
            a.eps = m.eps;
            a.post = pst;
            phases.push_back(P);
        };
        auto att_phase = [&](const float *k_, const float *v_, int kp, int64_t kv_bs, int64_t bias, const int *lut, const int *ln,
                             int Tk, bool self_att) {
            G2PPhase P{};
            P.kind = G2P_PH_ATT;
            P.q = q1, P.k = k_, P.v = v_, P.bias = bias >= 0 ? r.P(bias) : nullptr, P.lut = lut, P.lens = ln, P.out = a1;
            P.q_cs = 1, P.q_ts = I, P.kp = kp, P.kv_bs = kv_bs, P.lut_zero = G2PModel::kMaxPos - 1, P.seg = 1, P.Tk = Tk;
            P.causal = self_att ? 1 : 0, P.self_att = self_att ? 1 : 0, P.heads = m.heads, P.dk = m.d_kv, P.cols = NB;
            P.nblocks = NB * m.heads;
            phases.push_back(P);
        };
        {
            G2PPhase P{};
            P.kind = G2P_PH_EMBED;
            P.ids = d_gen, P.ib = TM, P.table = r.P(m.shared), P.x = x1, P.C = D, P.vocab = m.vocab;
            P.nblocks = (D + 255) / 256 * NB;
            phases.push_back(P);
        }
        for (int l = 0; l < nd; l++) {
            const auto &b = m.dec[l];
            step_phase({{&b.self.q, nullptr, q1, 1, I, nullptr}, {&b.self.k, nullptr, ks[l], TM, (int)cache_bs, nullptr},
                        {&b.self.v, nullptr, vs[l], TM, (int)cache_bs, nullptr}}, x1, b.ln_self, -1, 1.f, 1);
            att_phase(ks[l], vs[l], TM, cache_bs, m.dec_bias, h->d_bucket_dec, nullptr, 0, true);
            step_phase({{&b.self.o, nullptr, x1, 1, D, x1}}, a1, -1, -1, 1.f, 9);
            step_phase({{&b.cross.q, nullptr, q1, 1, I, nullptr}}, x1, b.ln_cross, -1, 1.f, 9);
            att_phase(kc[l], vc[l], T, S, -1, nullptr, d_lens, S, false);
            step_phase({{&b.cross.o, nullptr, x1, 1, D, x1}}, a1, -1, -1, 1.f, 9);
            step_phase({{&b.ffn.wi0, b.ffn.gated ? &b.ffn.wi1 : nullptr, f1, 1, m.d_ff, nullptr}}, x1, b.ln_ffn, m.act, 1.f, 9);
            step_phase({{&b.ffn.wo, nullptr, x1, 1, D, x1}}, f1, -1, -1, 1.f, 9);
        }
        step_phase({{&m.lm_head, nullptr, lg, 1, m.vocab, nullptr}}, x1, m.dec_final_ln, -1, post, 9);
        {
            G2PPhase P{};
            P.kind = G2P_PH_ARGMAX;
            P.step.x = lg, P.vocab = m.vocab, P.ids = d_arg, P.ib = TM, P.host_copy = h->tok_dev;
            P.nblocks = NB;
            phases.push_back(P);
        }
        // every workgroup has to be resident at once: the grid is what the occupancy calculator allows, at most
        // VITSMI_G2P_PERSIST_WGS per CU (default 1: a barrier's cost grows with the number of workgroups)
        const void *fn = NB == 1 ? (const void *)g2p_decode_step_kernel<1> : NB == 2 ? (const void *)g2p_decode_step_kernel<2>
                                                                                    : (const void *)g2p_decode_step_kernel<4>;
        int per_cu = 0, cus = 0;
        r.note(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, persist_lds));
        r.note(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
        int want = 1;
        if (const char *e = std::getenv("VITSMI_G2P_PERSIST_WGS")) want = std::atoi(e) > 0 ? std::atoi(e) : want;
        per_cu = per_cu < want ? per_cu : want;
        persist_grid = per_cu * cus;
        if (persist_grid <= 0) return gfail(h, VITS_E_DEVICE, "g2p: the persistent decoder step does not fit the device");
        *gave_up_host = 0;
    }
    {   // padded ids [NB][S], lengths, start tokens: one staging buffer each (pageable -> the copies are staged at once)
        std::vector<int64_t> ids((size_t)T, 0), gen((size_t)NB * TM, 0);
        std::vector<int> ln(NB, 1);
        for (int b = 0, o = 0; b < B; o += lens[b], b++) {
            std::memcpy(ids.data() + (size_t)b * S, input_ids + o, (size_t)lens[b] * 8);
            ln[b] = lens[b];
        }
        for (int b = 0; b < NB; b++) gen[(size_t)b * TM] = start_id;
        if (forced)
            for (int b = 0; b < B; b++)
                for (int t = 0; t < max_length; t++) gen[(size_t)b * TM + t] = forced[(size_t)b * max_length + t];
        r.note(hipMemcpyAsync(d_in, ids.data(), (size_t)T * 8, hipMemcpyHostToDevice, st));
        r.note(hipMemcpyAsync(d_gen, gen.data(), (size_t)NB * TM * 8, hipMemcpyHostToDevice, st));
        r.note(hipMemcpyAsync(d_lens, ln.data(), (size_t)NB * 4, hipMemcpyHostToDevice, st));
        if (persist) {
            r.note(hipMemcpyAsync(d_ph, phases.data(), phases.size() * sizeof(G2PPhase), hipMemcpyHostToDevice, st));
            r.note(hipMemsetAsync(d_bar, 0, nbar * sizeof(unsigned), st));
        }
        r.note(hipStreamSynchronize(st));  // (the staging vectors go out of scope)
    }
    run_encoder(r, d_in, S, NB, d_lens, xe, hn, q, k, v, att, fa, fb);
    for (int l = 0; l < nd; l++)
        linear(r, {{&m.dec[l].cross.k, kc[l], nullptr}, {&m.dec[l].cross.v, vc[l], nullptr}}, xe, T, T, T);
    // One decoder step for position t of every sequence (inputs d_gen[b][t]; the argmax lands in d_gen[b][t + 1] and in
    // the pinned copy); keys / values of position t join the caches.  8 launches per layer: norm + q|k|v, attention,
    // o (+x), norm + q, cross attention, o (+x), norm + gated input projections, wo (+x).
    auto enqueue_wide = [&](int t) {  // activations [C][NB]: column b = sequence b at position t
        g2p_embed_kernel<<<dim3((D + 255) / 256, NB), 256, 0, st>>>(d_gen + t, TM, r.P(m.shared), x1, D, NB, NB, m.vocab);
        for (int l = 0; l < nd; l++) {
            const auto &b = m.dec[l];
            rmsnorm(r, x1, NB, b.ln_self, h1, NB, NB);
            linear(r, {{&b.self.q, q1, nullptr}, {&b.self.k, ks[l] + t, nullptr, TM, cache_bs},
                       {&b.self.v, vs[l] + t, nullptr, TM, cache_bs}}, h1, NB, NB, NB);
            attention(r, q1, NB, 1, ks[l], vs[l], TM, cache_bs, m.dec_bias, h->d_bucket_dec, a1, NB, 1, nullptr, t + 1, t, true);
            linear(r, b.self.o, a1, NB, NB, x1, NB, x1);
            rmsnorm(r, x1, NB, b.ln_cross, h1, NB, NB);
            linear(r, b.cross.q, h1, NB, NB, q1, NB);
            attention(r, q1, NB, 1, kc[l], vc[l], T, S, -1, nullptr, a1, NB, 1, d_lens, S, 0, false);
            linear(r, b.cross.o, a1, NB, NB, x1, NB, x1);
            rmsnorm(r, x1, NB, b.ln_ffn, h1, NB, NB);
            ffn(r, b.ffn, h1, x1, NB, f1, f2);
        }
        rmsnorm(r, x1, NB, m.dec_final_ln, h1, NB, NB);
        if (m.scale_out) g2p_scale_kernel<<<(unsigned)(((size_t)D * NB + 255) / 256), 256, 0, st>>>(h1, (int64_t)D * NB, post);
        linear(r, m.lm_head, h1, NB, NB, lg, NB);
        g2p_argmax_kernel<<<NB, 256, 0, st>>>(lg, m.vocab, NB, 0, 1, d_gen, TM, t + 1, h->tok_dev);
        r.note(hipGetLastError());
        r.note(hipEventRecord(h->step_done[t & 1], st));
    };
    auto enqueue_step = [&](int t) {
        if (wide) return enqueue_wide(t);
        if (persist) {
            switch (NB) {
                case 1: g2p_decode_step_kernel<1><<<persist_grid, 256, persist_lds, st>>>(d_ph, nph, d_bar, t, gave_up_dev, persist_gs); break;
                case 2: g2p_decode_step_kernel<2><<<persist_grid, 256, persist_lds, st>>>(d_ph, nph, d_bar, t, gave_up_dev, persist_gs); break;
                default: g2p_decode_step_kernel<4><<<persist_grid, 256, persist_lds, st>>>(d_ph, nph, d_bar, t, gave_up_dev, persist_gs); break;
            }
            r.note(hipGetLastError());
            if (forced)
                r.note(hipMemcpyAsync(d_steplog + (size_t)t * NB * m.vocab, lg, (size_t)NB * m.vocab * 4, hipMemcpyDeviceToDevice, st));
            r.note(hipEventRecord(h->step_done[t & 1], st));
            return;
        }
        g2p_embed_rows_kernel<<<dim3((D + 255) / 256, NB), 256, 0, st>>>(d_gen + t, TM, r.P(m.shared), x1, D, m.vocab);
        for (int l = 0; l < nd; l++) {
            const auto &b = m.dec[l];
            step(r, NB, {{&b.self.q, nullptr, q1, 1, I, nullptr}, {&b.self.k, nullptr, ks[l] + t, TM, (int)cache_bs, nullptr},
                         {&b.self.v, nullptr, vs[l] + t, TM, (int)cache_bs, nullptr}}, x1, b.ln_self);
            attention(r, q1, 1, I, ks[l], vs[l], TM, cache_bs, m.dec_bias, h->d_bucket_dec, a1, NB, 1, nullptr, t + 1, t, true);
            step(r, NB, {{&b.self.o, nullptr, x1, 1, D, x1}}, a1, -1);
            step(r, NB, {{&b.cross.q, nullptr, q1, 1, I, nullptr}}, x1, b.ln_cross);
            attention(r, q1, 1, I, kc[l], vc[l], T, S, -1, nullptr, a1, NB, 1, d_lens, S, 0, false);
            step(r, NB, {{&b.cross.o, nullptr, x1, 1, D, x1}}, a1, -1);
            step(r, NB, {{&b.ffn.wi0, b.ffn.gated ? &b.ffn.wi1 : nullptr, f1, 1, m.d_ff, nullptr}}, x1, b.ln_ffn, m.act);
            step(r, NB, {{&b.ffn.wo, nullptr, x1, 1, D, x1}}, f1, -1);
        }
        step(r, NB, {{&m.lm_head, nullptr, lg, 1, m.vocab, nullptr}}, x1, m.dec_final_ln, -1, post);
        g2p_argmax_kernel<<<NB, 256, 0, st>>>(lg, m.vocab, 1, 0, m.vocab, d_arg, TM, t + 1, h->tok_dev);
        r.note(hipGetLastError());
        if (forced)
            r.note(hipMemcpyAsync(d_steplog + (size_t)t * NB * m.vocab, lg, (size_t)NB * m.vocab * 4, hipMemcpyDeviceToDevice, st));
        r.note(hipEventRecord(h->step_done[t & 1], st));
    };
    // The loop's one data-dependent decision (stop when every sequence has produced EOS) needs the tokens on the host;
    // step t + 1 is queued before the host waits for step t, so the device never idles on that round trip.  The step
    // queued behind the last one is wasted work on private buffers (everything later on this stream is ordered behind
    // it); a sequence that has finished keeps decoding until the others have, its tokens are dropped.
    std::vector<char> done(B, 0);
    int open_seqs = B;
    for (int b = 0; b < B; b++) n_out[b] = 0;
    enqueue_step(0);
    for (int t = 0; t < max_length && open_seqs > 0; t++) {
        if (t + 1 < max_length) enqueue_step(t + 1);
        r.note(hipEventSynchronize(h->step_done[t & 1]));
        if (r.err != hipSuccess) {
            hipStreamSynchronize(st);
            return gfail(h, VITS_E_DEVICE, "g2p_generate failed: %s", hipGetErrorString(r.err));
        }
        if (persist && *gave_up_host) {
            hipStreamSynchronize(st);
            return gfail(h, VITS_E_DEVICE, "g2p_generate: the persistent decoder step's grid barrier timed out at step %d "
                         "(unset VITSMI_G2P_PERSIST: a launch per phase)", t);
        }
        for (int b = 0; b < B; b++) {
            if (done[b]) continue;
            const int64_t tok = h->tok_host[(size_t)b * TM + t + 1];
            out_ids[(size_t)b * max_length + n_out[b]++] = tok;
            if (tok == eos_id && !forced) {
                done[b] = 1;
                open_seqs--;
            }
        }
    }
    if (forced) {  // [max_length][NB][vocab] on the device -> [B][max_length][vocab] for the caller
        std::vector<float> tmp((size_t)max_length * NB * m.vocab);
        r.note(hipMemcpyAsync(tmp.data(), d_steplog, tmp.size() * 4, hipMemcpyDeviceToHost, st));
        r.note(hipStreamSynchronize(st));
        if (r.err != hipSuccess) return gfail(h, VITS_E_DEVICE, "g2p forced steps failed: %s", hipGetErrorString(r.err));
        for (int b = 0; b < B; b++)
            for (int t = 0; t < max_length; t++)
                std::memcpy(step_logits + ((size_t)b * max_length + t) * m.vocab, tmp.data() + ((size_t)t * NB + b) * m.vocab,
                            (size_t)m.vocab * 4);
    }
    return VITS_OK;
}

int g2p_generate_batch(g2p_handle *h, const int64_t *input_ids, const int *lens, int B, int max_length, int64_t start_id,
                       int64_t eos_id, int64_t *out_ids, int *n_out) {
    return generate_impl(h, input_ids, lens, B, max_length, start_id, eos_id, out_ids, n_out, nullptr, nullptr);
}

int g2p_test_forced_steps(g2p_handle *h, const int64_t *input_ids, const int *lens, int B, const int64_t *decoder_input_ids, int T,
                          float *logits) {
    if (!h) return VITS_E_ARG;
    if (!decoder_input_ids || !logits || T <= 0 || B <= 0) return gfail(h, VITS_E_ARG, "bad g2p_test_forced_steps arguments");
    for (size_t i = 0; i < (size_t)B * T; i++)
        if (decoder_input_ids[i] < 0 || decoder_input_ids[i] >= h->model.vocab) return gfail(h, VITS_E_ARG, "decoder id out of range");
    std::vector<int64_t> ids((size_t)B * T);
    std::vector<int> n(B);
    return generate_impl(h, input_ids, lens, B, T, decoder_input_ids[0], -1, ids.data(), n.data(), decoder_input_ids, logits);
}

int g2p_generate(g2p_handle *h, const int64_t *input_ids, int S, int max_length, int64_t start_id, int64_t eos_id,
                 int64_t *out_ids, int *n_out) {
    if (!n_out) return gfail(h, VITS_E_ARG, "bad g2p_generate arguments");
    return g2p_generate_batch(h, input_ids, &S, 1, max_length, start_id, eos_id, out_ids, n_out);
}

}  // extern "C"
