// mfma_bf16x6_probe.hip — feasibility probe for a split-precision ("bf16x6") conv engine:
// fp32 a*b computed as 6 bf16 MFMA products (a0b0,a0b1,a1b0,a0b2,a1b1,a2b0) with fp32 accumulate.
// What rate does v_mfma_f32_32x32x16_bf16 sustain on this chip when its operands stream from LDS at
// the rate such an engine needs (3 A planes x WM blocks + 3 B planes x WN blocks per k=16 slab)?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16x6_probe.hip -o tools/bin/mfma_bf16x6_probe
// Reported TFLOP/s are bf16 MFMA flops; "fp32-equivalent" = /6.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int WM, int WN, bool LDS, int BAR = 0>
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[16384];  // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) lds[i] = __float_as_uint(in[(blockIdx.x * 64 + i) & 0xFFFFF]) & 0x3F803F80u;
    __syncthreads();
    f32x16 acc[WM][WN];
    for (int i = 0; i < WM; i++)
        for (int j = 0; j < WN; j++)
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const bf16x8 *base = reinterpret_cast<const bf16x8 *>(lds) + lane;
    bf16x8 a[WM][3], b[WN][3];
    for (int i = 0; i < WM; i++)
        for (int p = 0; p < 3; p++) a[i][p] = base[(i * 3 + p) * 64];
    for (int j = 0; j < WN; j++)
        for (int p = 0; p < 3; p++) b[j][p] = base[((WM + j) * 3 + p) * 64];
    unsigned sdummy = blockIdx.x, vdummy = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        bf16x8 an[WM][3], bn[WN][3];
        if (BAR >= 1) __builtin_amdgcn_s_barrier();
        if (BAR >= 2) {
#pragma unroll
            for (int q = 0; q < 100; q++) {
                if (BAR == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sdummy) : : "scc");
                else asm volatile("v_add_u32 %0, %0, 1" : "+v"(vdummy));
            }
        }
        if (LDS) {
            const bf16x8 *q = base + ((it & 7) * 512);
            for (int i = 0; i < WM; i++)
                for (int p = 0; p < 3; p++) an[i][p] = q[(i * 3 + p) * 64];
            for (int j = 0; j < WN; j++)
                for (int p = 0; p < 3; p++) bn[j][p] = q[((WM + j) * 3 + p) * 64];
        }
        for (int i = 0; i < WM; i++)
            for (int j = 0; j < WN; j++) {
                // smallest terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            }
        if (LDS) {
            for (int i = 0; i < WM; i++)
                for (int p = 0; p < 3; p++) a[i][p] = an[i][p];
            for (int j = 0; j < WN; j++)
                for (int p = 0; p < 3; p++) b[j][p] = bn[j][p];
        }
    }
    float s = 0.f;
    for (int i = 0; i < WM; i++)
        for (int j = 0; j < WN; j++)
            for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s + (float)sdummy + (float)vdummy;
}

template <int WM, int WN, bool LDS, int BAR = 0>
void run(const float *in, float *out, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    for (int wgs : {256, 512}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            k<WM, WN, LDS, BAR><<<wgs, 256>>>(in, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)wgs * 4 * iters * WM * WN * 6 * 32768.0;
        double tf = flop / (ms * 1e-3) / 1e12;
        printf("%-28s %4d WGs: %.3f ms  %.0f TFLOP/s bf16  = %.0f TFLOP/s fp32-equivalent\n", name, wgs, ms, tf, tf / 6);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n = 1 << 20;
    std::vector<float> h(n);
    unsigned s = 1;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) / 8388608.0f) - 1.0f; }
    float *in, *out;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    run<2, 2, false>(in, out, "2x2 registers only");
    run<2, 2, true>(in, out, "2x2 operands from LDS");
    run<2, 2, true, 1>(in, out, "2x2 LDS + barrier/step");
    run<2, 2, true, 2>(in, out, "2x2 LDS + barrier + 100 salu");
    run<2, 2, true, 3>(in, out, "2x2 LDS + barrier + 100 valu");
    run<4, 2, true>(in, out, "4x2 operands from LDS");
    run<2, 4, true>(in, out, "2x4 operands from LDS");
    run<1, 4, true>(in, out, "1x4 operands from LDS");
    run<1, 2, true>(in, out, "1x2 operands from LDS");
    return 0;
}
