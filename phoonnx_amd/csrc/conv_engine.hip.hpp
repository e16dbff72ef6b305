// conv_engine.hip.hpp — dense Conv1d as an implicit GEMM on the gfx950 f32 matrix cores.
//
//   out[b, co, t] = epi( bias[co] + sum_{ci,tap} W[co,ci,tap] * pro(x[b, ci, t + tap*dil - padL]) )
//
// GEMM view per batch item: M = Cout, N = T, K = Cin*taps.  v_mfma_f32_32x32x2_f32 computes a
// 32(co) x 32(t) tile per instruction with two k-values (an input-channel pair) — exact fp32 FMA
// chains, the same arithmetic the reference's fp32 convs perform (only the summation order differs).
//   A operand (weights)  : pre-packed at load time in lane order (model.cpp pack_conv), streamed from
//                          L2 with one 16-byte load per lane per four k-steps.
//   B operand (activation): a [CK channels] x [BN + halo] tile staged once per channel chunk into LDS
//                          (prologue activation / mask applied while staging, zero padding at the
//                          sequence ends), then read as conflict-free 32-lane rows, shifted per tap.
// One 256-thread workgroup = 4 wavefronts (64 lanes) arranged WM x WN; each wave owns MW x NW
// accumulator tiles (64 accumulator VGPRs).  Three tile shapes cover the model's channel widths:
//   cfg 2: 128(co) x 128(t)   Cout % 128 == 0      cfg 1: 64 x 256      cfg 0: 32 x 512 (Cout <= 32)
// Epilogues fuse bias, per-utterance conditioning bias, sequence mask, ReLU, residual add, multi-
// receptive-field accumulation (/3), the coupling update and the transposed-conv pixel shuffle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vitsmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum : int {
    PRO_LRELU = 1,      // leaky-relu(slope) on the input while staging
    PRO_MASK = 2,       // input * (t < len[b])
    EPI_RELU = 4,
    EPI_MASK = 8,       // value * (t < len[b])   (before the residual is added)
    EPI_RES = 16,       // + res[b,co,t]
    EPI_ACC = 32,       // out = out + value
    EPI_DIV = 64,       // out = value / div      (after EPI_ACC)
    EPI_COUPLING = 128  // out = (out - value*mask) * mask   (modules.py:464, mean_only)
};

struct ConvArgs {
    const float *x;
    int64_t x_bstride;  // floats between batch items of x (channel stride is T)
    int T;              // input (= virtual output) length
    const int *len;     // [B] valid lengths or nullptr
    const float *wp;    // packed weights
    const float *bias;  // [Cout] or nullptr
    const float *bias_b;  // per-batch bias [B][bias_b_stride] or nullptr
    int bias_b_stride;
    float *out;
    int64_t out_bstride;  // channel stride of out is T*ups
    const float *res;
    int64_t res_bstride;
    int Cin, Cout, K, dil, padL, CK, nchunks, steps4, LW, ups;
    int flags;
    float slope, div;
};

template <int MW, int NW, int WM, int WN>
__global__ __launch_bounds__(256) void conv_engine_kernel(ConvArgs a) {
    extern __shared__ float xs[];  // [CK][LW]
    constexpr int BM = WM * MW * 32, BN = WN * NW * 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * BN;
    const int T = a.T;
    const int len_b = a.len ? a.len[b] : T;
    const int in_lim = (a.flags & PRO_MASK) ? (len_b < T ? len_b : T) : T;

    f32x16 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; m++)
#pragma unroll
        for (int n = 0; n < NW; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    const float *xb = a.x + (int64_t)b * a.x_bstride;
    const float4 *wp4 = reinterpret_cast<const float4 *>(a.wp);
    const int mblk0 = blockIdx.y * (BM / 32) + wm * MW;
    const int LW = a.LW, CK = a.CK, K = a.K;
    const int g_per_tap = CK >> 3;
    const bool lrelu = a.flags & PRO_LRELU;
    const float slope = a.slope;

    for (int chunk = 0; chunk < a.nchunks; chunk++) {
        __syncthreads();  // previous chunk fully consumed
        for (int r = 0; r < CK; r++) {
            const int ci = chunk * CK + r;
            const float *row = xb + (int64_t)ci * T;
            const bool rv = ci < a.Cin;
            for (int c = tid; c < LW; c += 256) {
                const int t = t0 - a.padL + c;
                float v = 0.f;
                if (rv && t >= 0 && t < in_lim) {
                    v = row[t];
                    if (lrelu) v = v > 0.f ? v : v * slope;
                }
                xs[r * LW + c] = v;
            }
        }
        __syncthreads();
        for (int tap = 0; tap < K; tap++) {
            const int col = tap * a.dil + wn * (NW * 32) + l31;
            const int s4base = (chunk * K + tap) * g_per_tap;
            for (int g = 0; g < g_per_tap; g++) {
                float4 av[MW];
#pragma unroll
                for (int m = 0; m < MW; m++)
                    av[m] = wp4[((int64_t)(mblk0 + m) * a.steps4 + s4base + g) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float *brow = xs + (g * 8 + j * 2 + hi) * LW + col;
                    float bv[NW];
#pragma unroll
                    for (int n = 0; n < NW; n++) bv[n] = brow[n * 32];
#pragma unroll
                    for (int m = 0; m < MW; m++) {
                        const float aval = j == 0 ? av[m].x : (j == 1 ? av[m].y : (j == 2 ? av[m].z : av[m].w));
#pragma unroll
                        for (int n = 0; n < NW; n++)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bv[n], acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue.  C/D layout of the 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int flags = a.flags;
    const int ups = a.ups;
    const int64_t Tout = (int64_t)T * ups;
    float *ob = a.out + (int64_t)b * a.out_bstride;
    const float *rb = a.res ? a.res + (int64_t)b * a.res_bstride : nullptr;
    const float *bbp = a.bias_b ? a.bias_b + (int64_t)b * a.bias_b_stride : nullptr;
#pragma unroll
    for (int m = 0; m < MW; m++) {
#pragma unroll
        for (int n = 0; n < NW; n++) {
            const int t = t0 + wn * (NW * 32) + n * 32 + l31;
            if (t >= T) continue;
            const float mk = (t < len_b) ? 1.f : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int co = (mblk0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (co >= a.Cout) continue;
                float v = acc[m][n][r];
                if (a.bias) v += a.bias[co];
                if (bbp) v += bbp[co];
                if (flags & EPI_RELU) v = v > 0.f ? v : 0.f;
                int64_t o;
                if (ups == 1)
                    o = (int64_t)co * T + t;
                else
                    o = (int64_t)(co / ups) * Tout + (int64_t)t * ups + (co % ups);
                if (flags & EPI_COUPLING) {
                    ob[o] = (ob[o] - v * mk) * mk;
                    continue;
                }
                if (flags & EPI_MASK) v *= mk;
                if (flags & EPI_RES) v += rb[o];
                if (flags & EPI_ACC) v += ob[o];
                if (flags & EPI_DIV) v = v / a.div;
                ob[o] = v;
            }
        }
    }
}

inline int conv_tile_n(int cfg) { return cfg == 2 ? 128 : (cfg == 1 ? 256 : 512); }
inline int conv_tile_m(int cfg) { return cfg == 2 ? 128 : (cfg == 1 ? 64 : 32); }

// Launch on `stream`; `a.LW` is filled in here.  Returns hipError_t.
inline hipError_t launch_conv(ConvArgs a, int cfg, int B, hipStream_t stream) {
    const int BN = conv_tile_n(cfg), BM = conv_tile_m(cfg);
    a.LW = BN + (a.K - 1) * a.dil;
    dim3 grid((a.T + BN - 1) / BN, (a.Cout + BM - 1) / BM, B);
    size_t lds = (size_t)a.CK * a.LW * sizeof(float);
    if (grid.x == 0 || grid.y == 0 || B == 0) return hipSuccess;
    switch (cfg) {
        case 2: conv_engine_kernel<2, 2, 2, 2><<<grid, 256, lds, stream>>>(a); break;
        case 1: conv_engine_kernel<2, 2, 1, 4><<<grid, 256, lds, stream>>>(a); break;
        default: conv_engine_kernel<1, 4, 1, 4><<<grid, 256, lds, stream>>>(a); break;
    }
    return hipGetLastError();
}

}  // namespace vitsmi
