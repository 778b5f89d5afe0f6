// Which CUs does a CU-masked HIP stream run on?  hipExtStreamCreateWithCUMask(bits 0 .. n-1 set): histogram of the XCC and
// of (SE, SH, CU) ids the workgroups of a long-enough launch report.  build: hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
    unsigned xcc = 0, hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

int main(int argc, char** argv) {
    const int nbits = argc > 1 ? atoi(argv[1]) : 96;
    std::vector<uint32_t> mask(8, 0u);
    for (int i = 0; i < nbits; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("create failed\n"); return 1; }
    const int nb = 4096;
    unsigned* d;
    hipMalloc(&d, nb * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, s, d, 2000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> cus;
    for (int b = 0; b < nb; ++b) cus[h[2 * b] & 0xf].insert((h[2 * b + 1] >> 8) & 0xff);      // cu_id | sh_id | se_id bits
    int total = 0;
    for (auto& kv : cus) { printf("xcc %u: %zu distinct CU ids\n", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("mask bits 0..%d -> %d CUs in use\n", nbits - 1, total);
    return 0;
}
