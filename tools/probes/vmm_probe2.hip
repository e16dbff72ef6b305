// Which reservation alignments / mapping sizes / offsets does this stack accept?  (hipMemSetAccess is where it says no.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static const size_t MiB = 1ull << 20;
static bool try_seq(size_t align, const std::vector<size_t> &sizes, const char *what) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void *base = nullptr;
    if (hipMemAddressReserve(&base, 64ull << 30, align, nullptr, 0) != hipSuccess) { printf("%s: reserve failed\n", what); (void)hipGetLastError(); return false; }
    printf("%s: reserve align %zu MiB -> %p\n", what, align / MiB, base);
    size_t off = 0;
    std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> maps;
    bool all = true;
    for (size_t sz : sizes) {
        hipMemGenericAllocationHandle_t h{};
        hipError_t e = hipMemCreate(&h, sz, &prop, 0);
        const char *stage = "create";
        if (e == hipSuccess) { e = hipMemMap((char *)base + off, sz, 0, h, 0); stage = "map"; }
        if (e == hipSuccess) { e = hipMemSetAccess((char *)base + off, sz, &acc, 1); stage = "set access"; if (e != hipSuccess) (void)hipMemUnmap((char *)base + off, sz); }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            printf("   %6zu MiB at offset %6zu MiB: FAILED in %s (%s)\n", sz / MiB, off / MiB, stage, hipGetErrorString(e));
            (void)hipMemRelease(h);
            all = false;
            break;
        }
        printf("   %6zu MiB at offset %6zu MiB: ok\n", sz / MiB, off / MiB);
        maps.emplace_back(h, sz);
        off += sz;
    }
    if (all) {
        hipError_t e = hipMemset(base, 3, off);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        printf("   memset over all %zu MiB: %s\n", off / MiB, e == hipSuccess ? "ok" : hipGetErrorString(e));
    }
    size_t o = 0;
    for (auto &m : maps) { (void)hipMemUnmap((char *)base + o, m.second); (void)hipMemRelease(m.first); o += m.second; }
    (void)hipMemAddressFree(base, 64ull << 30);
    return all;
}
int main() {
    (void)hipSetDevice(0);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    (void)hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
    (void)hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
    const std::vector<size_t> dbl = {64 * MiB, 64 * MiB, 128 * MiB, 256 * MiB, 512 * MiB, 1024 * MiB, 2048 * MiB, 2048 * MiB, 2048 * MiB};
    const std::vector<size_t> eq256(12, 256 * MiB), eq2g(6, 2048 * MiB), odd = {320 * MiB, 1536 * MiB, 3584 * MiB, 64 * MiB};
    for (size_t al : {size_t(0), 64 * MiB, 2048 * MiB}) {
        try_seq(al, dbl, "doubling");
        try_seq(al, eq256, "equal 256 MiB");
        try_seq(al, eq2g, "equal 2 GiB");
        try_seq(al, odd, "odd sizes");
    }
    return 0;
}
