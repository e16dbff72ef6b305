// Does this stack grow a device buffer IN PLACE?  hipMemAddressReserve + hipMemCreate / hipMemMap chunk by chunk: a kernel writes
// the first chunk, a second chunk is mapped behind it, the first chunk's contents and pointer must survive; times each call.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(float *p, size_t n, float v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v + (float)(i & 1023); }
__global__ void check(const float *p, size_t n, float v, unsigned *bad) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n && p[i] != v + (float)(i & 1023)) atomicAdd(bad, 1u); }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int dev = 0, vmm = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("virtual memory management supported: %d\n", vmm);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity (recommended): %zu bytes\n", gran);
    const size_t VA = 64ull << 30, chunk = 2ull << 30;
    void *base = nullptr;
    double t0 = now();
    CK(hipMemAddressReserve(&base, VA, 0, nullptr, 0));
    printf("reserve 64 GiB of address space: %.3f ms -> %p\n", now() - t0, base);
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t h[8];
    unsigned *bad;
    CK(hipMalloc(&bad, 4));
    CK(hipMemset(bad, 0, 4));
    for (int c = 0; c < 8; c++) {
        t0 = now();
        CK(hipMemCreate(&h[c], chunk, &prop, 0));
        double t1 = now();
        CK(hipMemMap((char *)base + c * chunk, chunk, 0, h[c], 0));
        double t2 = now();
        CK(hipMemSetAccess((char *)base + c * chunk, chunk, &acc, 1));
        double t3 = now();
        const size_t n = chunk / 4;
        fill<<<(unsigned)((n + 255) / 256), 256>>>((float *)((char *)base + c * chunk), n, (float)c);
        CK(hipDeviceSynchronize());
        printf("chunk %d (2 GiB): create %.3f ms, map %.3f ms, set access %.3f ms, first touch %.3f ms\n", c, t1 - t0, t2 - t1, t3 - t2, now() - t3);
    }
    for (int c = 0; c < 8; c++) {
        const size_t n = chunk / 4;
        check<<<(unsigned)((n + 255) / 256), 256>>>((const float *)((char *)base + c * chunk), n, (float)c, bad);
    }
    // one kernel over a range that spans chunk boundaries
    unsigned hb = 1;
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("contents after growth: %s (%u mismatches)\n", hb == 0 ? "intact" : "DAMAGED", hb);
    t0 = now();
    void *m = nullptr;
    CK(hipMalloc(&m, 16ull << 30));
    printf("for comparison: hipMalloc(16 GiB) %.3f ms", now() - t0);
    t0 = now();
    CK(hipFree(m));
    printf(", hipFree %.3f ms\n", now() - t0);
    for (int c = 0; c < 8; c++) {
        CK(hipMemUnmap((char *)base + c * chunk, chunk));
        CK(hipMemRelease(h[c]));
    }
    CK(hipMemAddressFree(base, VA));
    printf("VMM_OK\n");
    return 0;
}
