// conv_sx_engine.hip.hpp — dense Conv1d as an implicit GEMM on the gfx950 bf16 matrix cores with
// fp32-exact operands ("split-exact", sx): every fp32 operand is carried as THREE bf16 planes
//     v = p0 + p1 + p2,   p0 = bf16(v), p1 = bf16(v - p0), p2 = bf16(v - p0 - p1)     (exact: 3 x 8 = 24 bits)
// and a product w*x is evaluated as the six plane products of combined order <= 2
//     w0x0 + w0x1 + w1x0 + w0x2 + w1x1 + w2x0
// each of which v_mfma_f32_32x32x16_bf16 forms exactly (8 x 8 significant bits) and accumulates in fp32.
// The three dropped terms are bounded by 3 * 2^-24 |w||x|, i.e. the result carries the same error
// bound as an fp32 FMA chain (what the reference's fp32 convolutions and the f32 engine in
// conv_engine.hip.hpp compute); tests/test_gpu_parity.py checks both engines against float64.
//
// Why: the f32 matrix pipe peaks at 157 TFLOP/s (measured 155), the bf16 pipe at 2.5 PFLOP/s; six bf16
// MFMAs per fp32-equivalent product leave a 419 TFLOP/s ceiling (tools/mfma_bf16x6_probe.hip sustains
// 340-375 TFLOP/s fp32-equivalent with both operands streaming from LDS).
//
// Layouts (T = time steps of the tensor, C % 16 == 0 on inputs, C % 32 == 0 on outputs):
//   planes  bf16 [3][C/8][T][8]   conv inputs; one 16-byte cell = 8 channels of one time step, which is
//                                 exactly one lane's B operand (8 k-values) of the MFMA
//   raw     fp32 [C/8][T][8]      residual stream (same cell structure, 32-byte cells)
//   weights bf16 [m-tile][chunk of 16 ci][tap][32-row block][plane][lane][8]  (model.cpp pack_conv_sx):
//                                 the A slab of a (chunk, tap group) is one contiguous range
// Both operands reach LDS by 16-byte LDS-DMA; zero padding = out-of-range cells read a zero page.
// Pipeline: the A slab streams per step (= TG taps of one chunk), double-buffered; the x tile of the NEXT
// chunk is fetched in slices spread over the steps of the current chunk; one barrier per step.
// Epilogue: bias, per-utterance bias, residual, multi-receptive-field accumulate and /n, leaky-ReLU,
// pixel shuffle of the transposed conv (virtual rows are r-major: row = r*Cr + co), then an fp32 raw
// store and/or a split into the three planes the next conv reads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "conv_engine.hip.hpp"

namespace vitsmi {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

enum : int { SX_NO_RAW_STORE = 256 };  // out_raw is only the EPI_ACC operand, not a destination

struct SxArgs {
    const u32x4 *xp;      // input planes, cells of 8 bf16
    int64_t x_bstride;    // cells between batch items (= 3 * Cin/8 * T)
    int T;                // input length
    const u32x4 *wp;      // packed weights
    const float *bias;    // [virtual rows] or nullptr
    const float *bias_b;  // per-utterance bias [B][bias_b_stride] over real channels, or nullptr
    int bias_b_stride;
    float *out_raw;       // fp32 raw [Cr/8][T*ups][8] or nullptr; stores leaky_relu(value, oslope)
    int64_t raw_bstride;  // floats between batch items (out_raw and res)
    uint16_t *out_pl;     // planes [3][Cr/8][T*ups][8] or nullptr; stores split(leaky_relu(value, oslope2))
    int64_t pl_bstride;   // bf16 elements between batch items
    const float *res;     // fp32 raw residual or nullptr
    const float *zeros;   // >= 1 KiB of zeros, 16-byte aligned
    int Cin, Cout, Cr;    // Cout = virtual rows (Cr * ups)
    int K, dil, padL, nchunks, ups;
    int TG;               // taps per pipeline step                      (filled by launch_conv_sx)
    int LW;               // x tile width in cells                       ( " )
    unsigned magic;       // ceil(2^32 / LW)                             ( " )
    unsigned x_bytes, a_bytes;  // bytes of one x stage / one A stage    ( " )
    int flags;            // EPI_RES | EPI_ACC | EPI_DIV | DBG_*
    float div, oslope, oslope2;
};

template <int OFF>
__device__ __forceinline__ u32x4 ds_read128(uint32_t addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
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

// One 256-thread workgroup = 4 waves arranged WM x WN, each owning MW x NW 32x32 accumulator blocks.
template <int MW, int NW, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_sx_kernel(SxArgs a) {
    constexpr int BM = MW * WM * 32, BN = NW * WN * 32, MB = BM / 32;
    static_assert(WM * WN == 4, "four waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_sx[];  // [x0][x1][a0][a1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, t0 = blockIdx.x * BN;
    const int T = a.T, LW = a.LW, K = a.K, TG = a.TG, CG = a.Cin >> 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds_sx;
    const uint32_t XB = a.x_bytes, AB = a.a_bytes;
    const u32x4 *xb = a.xp + (int64_t)b * a.x_bstride;
    const int64_t pstride = (int64_t)CG * T;  // cells per plane
    constexpr int TAPCELLS = MB * 3 * 64;      // cells of one tap of the A slab
    const u32x4 *wmt = a.wp + (int64_t)blockIdx.y * a.nchunks * K * TAPCELLS + lane;
    const int npieces = 6 * LW;                // x tile: rows (plane, channel-group half) x LW cells
    const int nit = (npieces + 255) >> 8;      // DMA rounds of 256 pieces
    const int spc = (K + TG - 1) / TG;         // steps per chunk
    const int ips = (nit + spc - 1) / spc;     // x rounds issued per step

    auto issue_a = [&](int chunk, int tap0, int nt, uint32_t abuf) {
        const u32x4 *src = wmt + ((int64_t)chunk * K + tap0) * TAPCELLS;
        const int n = nt * MB * 3;  // 1 KiB pieces
        for (int i = wave; i < n; i += 4)
            lds_dma<16>(src + i * 64, reinterpret_cast<float *>(lds_sx + (abuf - lds0) + i * 1024));
    };
    auto issue_x = [&](int chunk, int it0, int it1, uint32_t xbuf) {
        for (int it = it0; it < it1; it++) {
            const int i = it * 256 + tid;
            const int row = (int)__umulhi((unsigned)i, a.magic);
            const int col = i - row * LW;
            const int t = t0 - a.padL + col;
            const bool ok = row < 6 && t >= 0 && t < T;
            const u32x4 *src = ok ? xb + ((row >> 1) * pstride + (int64_t)(2 * chunk + (row & 1)) * T + t)
                                  : reinterpret_cast<const u32x4 *>(a.zeros) + lane;
            lds_dma<16>(src, reinterpret_cast<float *>(lds_sx + (xbuf - lds0) + (it * 256 + wave * 64) * 16));
        }
    };

    f32x16 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; m++)
#pragma unroll
        for (int n = 0; n < NW; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    struct Frag {
        u32x4 fa[MW][3], fb[NW][3];
    };
    const uint32_t a_lane = (uint32_t)(wm * MW * 3) * 1024u + (uint32_t)lane * 16u;
    const uint32_t b_lane = (uint32_t)(hi * LW + wn * (NW * 32) + l31) * 16u;
    const uint32_t plane_b = (uint32_t)(2 * LW) * 16u;

    auto load = [&](Frag &f, uint32_t abuf, uint32_t xbuf, int tap_in_step, int tap) {
        const uint32_t aa = abuf + a_lane + (uint32_t)tap_in_step * (uint32_t)(TAPCELLS * 16);
        const uint32_t bb0 = xbuf + b_lane + (uint32_t)(tap * a.dil) * 16u;
        const uint32_t bb1 = bb0 + plane_b, bb2 = bb1 + plane_b;
        static_for<MW>([&](auto M) {
            constexpr int m = decltype(M)::value;
            f.fa[m][0] = ds_read128<(m * 3 + 0) * 1024>(aa);
            f.fa[m][1] = ds_read128<(m * 3 + 1) * 1024>(aa);
            f.fa[m][2] = ds_read128<(m * 3 + 2) * 1024>(aa);
        });
        static_for<NW>([&](auto N) {
            constexpr int n = decltype(N)::value;
            f.fb[n][0] = ds_read128<n * 512>(bb0);
            f.fb[n][1] = ds_read128<n * 512>(bb1);
            f.fb[n][2] = ds_read128<n * 512>(bb2);
        });
    };
    auto mma = [&](const Frag &f) {
        // plane pairs of combined order <= 2, smallest terms first; consecutive MFMAs hit different accumulators
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int c = 0; c < 6; c++)
#pragma unroll
            for (int m = 0; m < MW; m++)
#pragma unroll
                for (int n = 0; n < NW; n++)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.fa[m][PA[c]]),
                                                                        __builtin_bit_cast(bf16x8, f.fb[n][PB[c]]),
                                                                        acc[m][n], 0, 0, 0);
    };
    // taps [tap0, tap0+nt) of one chunk; fragments of tap+1 are fetched while tap's MFMAs run
    auto compute = [&](uint32_t abuf, uint32_t xbuf, int tap0, int nt) {
        Frag f0, f1;
        load(f0, abuf, xbuf, 0, tap0);
        for (int i = 0; i < nt; i += 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < nt) load(f1, abuf, xbuf, i + 1, tap0 + i + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(f0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 >= nt) break;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (i + 2 < nt) load(f0, abuf, xbuf, i + 2, tap0 + i + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(f1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const uint32_t xbuf0 = lds0, abuf0 = lds0 + 2 * XB;
    const int nchunks = a.nchunks;
    const bool dbg_nodma = a.flags & DBG_NO_DMA;
    issue_x(0, 0, nit, xbuf0);
    issue_a(0, 0, TG < K ? TG : K, abuf0);
    int s = 0;
    for (int chunk = 0; chunk < nchunks; chunk++) {
        const uint32_t xcur = xbuf0 + (chunk & 1) * XB, xnext = xbuf0 + ((chunk + 1) & 1) * XB;
        for (int g = 0; g < spc; g++, s++) {
            __syncthreads();  // own DMA drained; everyone is done with the buffers refilled below
            const int tap0 = g * TG, nt = (K - tap0) < TG ? (K - tap0) : TG;
            // next step's A slab
            int nchunk = chunk, ntap0 = tap0 + TG;
            if (ntap0 >= K) { nchunk++; ntap0 = 0; }
            if (nchunk < nchunks && !(dbg_nodma && s > 0))
                issue_a(nchunk, ntap0, (K - ntap0) < TG ? (K - ntap0) : TG, abuf0 + ((s + 1) & 1) * AB);
            // a slice of the next chunk's x tile
            if (chunk + 1 < nchunks && !dbg_nodma) {
                const int it0 = g * ips, it1 = (it0 + ips) < nit ? (it0 + ips) : nit;
                issue_x(chunk + 1, it0, it1, xnext);
            }
            compute(abuf0 + (s & 1) * AB, xcur, tap0, nt);
        }
    }

    // ---- epilogue.  C/D layout of a 32x32 block: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5):
    // register quad q holds 4 consecutive channels 8q + 4*hi .. +3 of one time step = half a cell.
    const int flags = a.flags;
    if (flags & DBG_NO_EPI) {
        float sdbg = 0.f;
#pragma unroll
        for (int m = 0; m < MW; m++)
#pragma unroll
            for (int n = 0; n < NW; n++) sdbg += acc[m][n][0] + acc[m][n][7] + acc[m][n][15];
        if (sdbg == 12345.678f && a.out_raw) a.out_raw[tid] = sdbg;
        return;
    }
    const int u = a.ups, Cr = a.Cr, Tout = T * u, CGo = Cr >> 3;
    float *rawb = a.out_raw ? a.out_raw + (int64_t)b * a.raw_bstride : nullptr;
    uint16_t *plb = a.out_pl ? a.out_pl + (int64_t)b * a.pl_bstride : nullptr;
    const float *resb = a.res ? a.res + (int64_t)b * a.raw_bstride : nullptr;
    const float *addp = (flags & EPI_RES) ? resb : rawb;
    const bool has_add = (flags & (EPI_RES | EPI_ACC)) != 0;
    const bool two_adds = (flags & EPI_RES) && (flags & EPI_ACC);
    const float oslope = a.oslope, oslope2 = a.oslope2, rdiv = a.div;
    const int64_t plane_elems = (int64_t)CGo * Tout * 8;
#pragma unroll
    for (int m = 0; m < MW; m++) {
        const int row0 = blockIdx.y * BM + (wm * MW + m) * 32;
        if (row0 >= a.Cout) continue;
        const int r = u == 1 ? 0 : row0 / Cr;
        const int co0 = row0 - r * Cr;
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bq[q] = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + row0 + 8 * q + 4 * hi) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias_b)
                bq[q] += *reinterpret_cast<const f32x4 *>(a.bias_b + (int64_t)b * a.bias_b_stride + co0 + 8 * q + 4 * hi);
        }
        f32x4 ad[NW][4], ad2[NW][4];
        int64_t cell[NW][4];
        bool okn[NW];
#pragma unroll
        for (int n = 0; n < NW; n++) {
            const int t = t0 + (wn * NW + n) * 32 + l31;
            okn[n] = t < T;
            const int to = t * u + r;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                cell[n][q] = ((int64_t)((co0 >> 3) + q) * Tout + to) * 8 + 4 * hi;  // element offset inside raw / a plane
                if (has_add && okn[n]) ad[n][q] = *reinterpret_cast<const f32x4 *>(addp + cell[n][q]);
                if (two_adds && okn[n]) ad2[n][q] = *reinterpret_cast<const f32x4 *>(rawb + cell[n][q]);
            }
        }
#pragma unroll
        for (int n = 0; n < NW; n++) {
            if (!okn[n]) continue;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = acc[m][n][4 * q + e] + bq[q][e];
                if (has_add) v += ad[n][q];
                if (two_adds) v += ad2[n][q];
                if (flags & EPI_DIV) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = v[e] / rdiv;
                }
                if (rawb && !(flags & SX_NO_RAW_STORE)) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = lrelu_f(v[e], oslope);
                    *reinterpret_cast<f32x4 *>(rawb + cell[n][q]) = o;
                }
                if (plb) {
                    unsigned short p[3][4];
#pragma unroll
                    for (int e = 0; e < 4; e++) split3(lrelu_f(v[e], oslope2), p[0][e], p[1][e], p[2][e]);
#pragma unroll
                    for (int pl = 0; pl < 3; pl++) {
                        u32x2 w;
                        w.x = (unsigned)p[pl][0] | ((unsigned)p[pl][1] << 16);
                        w.y = (unsigned)p[pl][2] | ((unsigned)p[pl][3] << 16);
                        *reinterpret_cast<u32x2 *>(plb + pl * plane_elems + cell[n][q]) = w;
                    }
                }
            }
        }
    }
}

// sx tile configs: index -> (BM, BN): 0: 128x128, 1: 64x256, 2: 32x256
inline int sx_tile_m(int cfg) { return cfg == 0 ? 128 : (cfg == 1 ? 64 : 32); }
inline int sx_tile_n(int cfg) { return cfg == 0 ? 128 : 256; }
constexpr size_t kSxLdsBudget = 80 * 1024;  // two workgroups per CU

template <int MW, int NW, int WM, int WN>
inline hipError_t launch_conv_sx_k(const SxArgs &a, dim3 grid, size_t lds, hipStream_t stream) {
    static bool attr_set = false;
    auto kern = conv_sx_kernel<MW, NW, WM, WN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    kern<<<grid, 256, lds, stream>>>(a);
    return hipGetLastError();
}

inline hipError_t launch_conv_sx(SxArgs a, int cfg, int B, hipStream_t stream) {
    const int BM = sx_tile_m(cfg), BN = sx_tile_n(cfg), MB = BM / 32;
    a.LW = BN + (a.K - 1) * a.dil;
    a.magic = (unsigned)((0x100000000ull + a.LW - 1) / a.LW);
    a.x_bytes = (unsigned)(((size_t)6 * a.LW * 16 + 4095) / 4096 * 4096);
    int tg = 4 / MB > 0 ? 4 / MB : 1;  // 12 KiB A stage
    if (tg > a.K) tg = a.K;
    while (tg > 1 && 2 * ((size_t)a.x_bytes + (size_t)tg * MB * 3072) > kSxLdsBudget) tg >>= 1;
    a.TG = tg;
    a.a_bytes = (unsigned)(tg * MB * 3072);
    const size_t lds = 2 * ((size_t)a.x_bytes + a.a_bytes);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (a.oslope == 0.f) a.oslope = 1.f;
    if (a.oslope2 == 0.f) a.oslope2 = 1.f;
    if (a.Cin % 16 || a.Cout % 32 || a.Cr % 32 || a.Cout % BM) return hipErrorInvalidValue;
    dim3 grid((a.T + BN - 1) / BN, a.Cout / BM, B);
    if (grid.x == 0 || B == 0) return hipSuccess;
    switch (cfg) {
        case 0: return launch_conv_sx_k<2, 2, 2, 2>(a, grid, lds, stream);
        case 1: return launch_conv_sx_k<2, 2, 1, 4>(a, grid, lds, stream);
        default: return launch_conv_sx_k<1, 2, 1, 4>(a, grid, lds, stream);
    }
}

// ---- layout conversion kernels ------------------------------------------------------------------------

// planar fp32 x[b][c][t] (row pitch `pitch`, optionally masked by t < len[b]) -> planes [3][C/8][T][8]
__global__ __launch_bounds__(256) void sx_split_planes_kernel(const float *x, int64_t x_bstride, int pitch, const int *len,
                                                              uint16_t *out, int C, int T) {
    const int t = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const bool live = !len || t < len[b];
    const float *xb = x + (int64_t)b * x_bstride + (int64_t)cg * 8 * pitch + t;
    unsigned short p[3][8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float v = live ? xb[(int64_t)e * pitch] : 0.f;
        split3(v, p[0][e], p[1][e], p[2][e]);
    }
    const int CG = C >> 3;
    uint16_t *ob = out + (int64_t)b * 3 * CG * T * 8;
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
        u32x4 w;
        w.x = (unsigned)p[pl][0] | ((unsigned)p[pl][1] << 16);
        w.y = (unsigned)p[pl][2] | ((unsigned)p[pl][3] << 16);
        w.z = (unsigned)p[pl][4] | ((unsigned)p[pl][5] << 16);
        w.w = (unsigned)p[pl][6] | ((unsigned)p[pl][7] << 16);
        *reinterpret_cast<u32x4 *>(ob + (((int64_t)pl * CG + cg) * T + t) * 8) = w;
    }
}

// planar fp32 [C][T] -> raw fp32 [C/8][T][8]
__global__ __launch_bounds__(256) void sx_block_kernel(const float *x, float *raw, int C, int T) {
    const int t = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
#pragma unroll
    for (int e = 0; e < 8; e++)
        raw[(int64_t)b * C * T + ((int64_t)cg * T + t) * 8 + e] = x[(int64_t)b * C * T + (int64_t)(cg * 8 + e) * T + t];
}

// raw fp32 [C/8][T][8] (or, with planes != nullptr, the sum of the three planes) -> planar [C][T]
__global__ __launch_bounds__(256) void sx_unblock_kernel(const float *raw, const uint16_t *planes, float *out, int C, int T) {
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
            v = (bf16_bits_to_f32(pb[o + 2 * ps]) + bf16_bits_to_f32(pb[o + ps])) + bf16_bits_to_f32(pb[o]);
        } else
            v = raw[(int64_t)b * C * T + ((int64_t)cg * T + t) * 8 + e];
        out[(int64_t)b * C * T + (int64_t)(cg * 8 + e) * T + t] = v;
    }
}

}  // namespace vitsmi
