// dev experiment: which CUs / XCDs does a stream created with hipExtStreamCreateWithCUMask run on?
// build: hipcc --offload-arch=gfx950 -O2 cu_mask_census.hip -o cu_mask_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
__global__ void census(unsigned *out)
{
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11));      // HW_REG_XCC_ID bits 0..3
        unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11));    // HW_REG_HW_ID
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hwid;
    }
    // stay resident a little so that workgroups spread over every CU the stream may use
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) {}
}
static void run(const char *name, const std::vector<uint32_t> &mask)
{
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int nb = 2048;
    unsigned *d, h[2 * nb];
    hipMalloc(&d, sizeof(h));
    census<<<nb, 64, 0, st>>>(d);
    hipStreamSynchronize(st);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per;
    for (int b = 0; b < nb; b++) {
        unsigned hw = h[2 * b + 1];
        unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per[h[2 * b]].insert((se << 8) | (sh << 4) | cu);
    }
    printf("%s:", name);
    int tot = 0;
    for (auto &kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total CUs %d\n", tot);
    hipFree(d); hipStreamDestroy(st);
}
int main()
{
    std::vector<uint32_t> all(8, 0xFFFFFFFFu);
    run("all 256 bits", all);
    std::vector<uint32_t> low32(8, 0); low32[0] = 0xFFFFFFFFu;
    run("bits 0..31", low32);
    std::vector<uint32_t> every8(8, 0x01010101u);
    run("every 8th bit", every8);
    std::vector<uint32_t> firstbyte(8, 0x000000FFu);
    run("low byte of each word", firstbyte);
    std::vector<uint32_t> lowhalf(8, 0x0000FFFFu);
    run("low half of each word", lowhalf);
    return 0;
}
