// mfma_shape_probe.hip — which f16 MFMA shape should the split-operand conv engine issue?
// MI355X_MICROARCH.md ("DVFS give-back", item 7): under the power cap the chip holds a higher clock on
// v_mfma_f32_16x16x32 than on v_mfma_f32_32x32x16 at equal cycles per FLOP (1.12-1.15x FLOP/s on bf16, operands from
// registers or re-read from LDS).  conv_sx_kernel is power-limited (1.5-1.7 GHz on its 128-row tiles), so this probe
// repeats that comparison with the engine's own arithmetic and operand pattern: f16x3 (two fp16 planes per operand,
// products g1*h0, g0'*h1', g0*h0 with g0' = g0 * 2^-11 made by packed multiplies), a 64 x 128 output tile per wave, two
// 256-thread workgroups per CU, every operand fragment re-read from LDS each k-step (B) or kept in registers (R).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/bin/mfma_shape_probe
// Reports f16 MFMA TFLOP/s, the fp32-equivalent (/3) and the in-kernel shader clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// SHAPE 32: v_mfma_f32_32x32x16_f16, k-step 16: A 2 blocks x 2 planes, B 4 blocks x 2 planes, 24 MFMAs of 32 cycles
// SHAPE 16: v_mfma_f32_16x16x32_f16, k-step 32: A 4 blocks x 2 planes, B 8 blocks x 2 planes, 96 MFMAs of 16 cycles
// (one iteration of the timed loop = k 32 in both: two steps of the first, one of the second)
template <int SHAPE, bool LDS>
__global__ __launch_bounds__(256, 2) void k(const float *in, float *out, int iters, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[32768];  // 64 KiB of fp16 operands
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768; i += 256) lds[i] = (_Float16)(in[(blockIdx.x * 64 + i) & 0xFFFFF] * 4.f);
    __syncthreads();
    const f16x8 *base = reinterpret_cast<const f16x8 *>(lds) + lane;  // a fragment = 64 lanes x 16 B = 1 KiB
    unsigned long long t0 = 0, r0 = 0;
    if (tid == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][4];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 4; j++)
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        f16x8 a[2][2], b[4][2];
        for (int i = 0; i < 2; i++)
            for (int p = 0; p < 2; p++) a[i][p] = base[(i * 2 + p) * 64];
        for (int j = 0; j < 4; j++)
            for (int p = 0; p < 2; p++) b[j][p] = base[(4 + j * 2 + p) * 64];
        for (int it = 0; it < 2 * iters; it++) {
            const f16x8 *q = base + ((it & 3) * 12 * 64);
            if (LDS) {
                for (int i = 0; i < 2; i++)
                    for (int p = 0; p < 2; p++) a[i][p] = q[(i * 2 + p) * 64];
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (LDS) {
                    for (int j = 2 * h; j < 2 * h + 2; j++)
                        for (int p = 0; p < 2; p++) b[j][p] = q[(4 + j * 2 + p) * 64];
                }
                for (int c = 0; c < 3; c++)
                    for (int i = 0; i < 2; i++)
                        for (int j = 2 * h; j < 2 * h + 2; j++) {
                            const f16x8 ga = c == 0 ? a[i][1] : (c == 1 ? a[i][0] * (_Float16)0.00048828125f : a[i][0]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga, b[j][c == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
                        }
            }
        }
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 4; j++)
                for (int r = 0; r < 16; r++) s += acc[i][j][r];
    } else {
        f32x4 acc[4][8];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 8; j++)
                for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;
        f16x8 a[4][2], b[8][2];
        for (int i = 0; i < 4; i++)
            for (int p = 0; p < 2; p++) a[i][p] = base[(i * 2 + p) * 64];
        for (int j = 0; j < 8; j++)
            for (int p = 0; p < 2; p++) b[j][p] = base[(8 + j * 2 + p) * 64];
        for (int it = 0; it < iters; it++) {
            const f16x8 *q = base + ((it & 1) * 24 * 64);
            if (LDS) {
                for (int i = 0; i < 4; i++)
                    for (int p = 0; p < 2; p++) a[i][p] = q[(i * 2 + p) * 64];
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (LDS) {
                    for (int j = 4 * h; j < 4 * h + 4; j++)
                        for (int p = 0; p < 2; p++) b[j][p] = q[(8 + j * 2 + p) * 64];
                }
                for (int c = 0; c < 3; c++)
                    for (int i = 0; i < 4; i++)
                        for (int j = 4 * h; j < 4 * h + 4; j++) {
                            const f16x8 ga = c == 0 ? a[i][1] : (c == 1 ? a[i][0] * (_Float16)0.00048828125f : a[i][0]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga, b[j][c == 1 ? 1 : 0], acc[i][j], 0, 0, 0);
                        }
            }
        }
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 8; j++)
                for (int r = 0; r < 4; r++) s += acc[i][j][r];
    }
    if (tid == 0 && clk) {
        atomicAdd(clk, __builtin_amdgcn_s_memtime() - t0);
        atomicAdd(clk + 1, __builtin_amdgcn_s_memrealtime() - r0);
    }
    out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE, bool LDS>
void run(const float *in, float *out, unsigned long long *clk, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 6000, wgs = 512;
    float best = 1e30f;
    double ghz = 0;
    for (int rep = 0; rep < 4; rep++) {
        hipMemset(clk, 0, 16);
        hipEventRecord(e0);
        k<SHAPE, LDS><<<wgs, 256>>>(in, out, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        if (ms < best) {
            best = ms;
            ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
        }
    }
    // per iteration and wave: 64 x 128 x 32 MACs x 3 products
    const double flop = (double)wgs * 4 * iters * 64.0 * 128.0 * 32.0 * 2.0 * 3.0;
    const double tf = flop / (best * 1e-3) / 1e12;
    printf("%-40s %8.3f ms  %6.0f TFLOP/s f16 = %5.0f fp32-equivalent (%.3f of 838.9)  clock %.2f GHz\n", name, best, tf, tf / 3,
           tf / 3 / 838.9, ghz);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n = 1 << 20;
    std::vector<float> h(n);
    unsigned s = 1;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) / 8388608.0f) - 1.0f; }
    float *in, *out;
    unsigned long long *clk;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&clk, 16);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; round++) {  // interleaved rounds in one process (same device, same thermal state)
        run<32, false>(in, out, clk, "32x32x16  registers only");
        run<16, false>(in, out, clk, "16x16x32  registers only");
        run<32, true>(in, out, clk, "32x32x16  operands re-read from LDS");
        run<16, true>(in, out, clk, "16x16x32  operands re-read from LDS");
    }
    return 0;
}
