// mfma_peak.hip — what does v_mfma_f32_32x32x2_f32 sustain on THIS chip, under the DVFS regime a
// real kernel sees?  (Tuning aid: a ceiling must come from a known-good loop measured on the same
// hardware, not from the spec sheet.)   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
//   mode 0: pure register MFMA loop, 4 independent accumulators
//   mode 1: + one ds_read2_b32 and 6 VALU ops per 4 MFMAs (the conv engine's k-step), data from LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters, unsigned long long *clk) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = in[(blockIdx.x * 8192 + i) & 0xFFFFF];
    __syncthreads();
    f32x16 acc[4];
    for (int q = 0; q < 4; q++)
        for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
    float a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768];
    const float slope = in[5] * 0.f + 0.1f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t ad = (uint32_t)(uintptr_t)lds + lane * 4;
    for (int it = 0; it < iters; it++) {
        if (MODE == 1) {
            f32x2 r;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(r) : "v"(ad + (uint32_t)((it & 63) * 256)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            b0 = __builtin_amdgcn_fmed3f(b0, b0 * slope, __builtin_inff());
            b1 = __builtin_amdgcn_fmed3f(b1, b1 * slope, __builtin_inff());
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            b0 = r.x;
            b1 = r.y;
        } else {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int q = 0; q < 4; q++)
        for (int r = 0; r < 16; r++) s += acc[q][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}

int main() {
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
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 2; mode++)
        for (int wgs : {256, 512, 768, 1024}) {  // 1..4 workgroups (4 waves each) per CU
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<wgs, 256>>>(in, out, iters, clk);
                else k<1><<<wgs, 256>>>(in, out, iters, clk);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[2];
            hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            double flop = (double)wgs * 4 * iters * 4 * 4096.0;
            printf("mode %d  %4d WGs (%d waves/SIMD): %.3f ms  %.1f TFLOP/s   clock %.2f GHz (memtime/memrealtime)\n", mode, wgs,
                   wgs / 256, ms, flop / (ms * 1e-3) / 1e12, (double)c[0] / (double)c[1] * 0.1);
        }
    return 0;
}
