// Probe: does hipExtStreamCreateWithCUMask confine a stream's waves to the CUs named in the mask on gfx950, and how are the mask bits numbered?
// Every wave records {XCC_ID, SE_ID, CU_ID} from the hardware id registers; the host prints, per mask, which (xcc, se, cu) triples were used.
//   hipcc --offload-arch=gfx950 -O3 -o cumask cumask.hip && ./cumask
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(uint32_t* out, int spin)
{
    // HW_ID (hwreg 4): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se ... ; XCC_ID (hwreg 20) [3:0]
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}

int main()
{
    hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
    printf("%s: %d CUs, mask words %d\n", prop.gcnArchName, cus, words);
    const int blocks = 4096;
    uint32_t* d; if (hipMalloc(&d, blocks * 8) != hipSuccess) return 1;
    std::vector<uint32_t> h(blocks * 2);
    auto run = [&](const char* name, const std::vector<uint32_t>& mask) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
        where_kernel<<<blocks, 64, 0, st>>>(d, 200);       // 2 us per wave: enough to spread over every enabled CU
        if (hipStreamSynchronize(st) != hipSuccess) { printf("%s: launch failed\n", name); return; }
        hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
        std::map<int, std::set<int>> perXcc;
        for (int b = 0; b < blocks; ++b) { const uint32_t hw = h[b * 2], xcc = h[b * 2 + 1] & 15u; perXcc[(int)xcc].insert((int)(((hw >> 13) & 7u) * 100 + ((hw >> 12) & 1u) * 50 + ((hw >> 8) & 15u))); }
        printf("%s:", name);
        int total = 0;
        for (auto& kv : perXcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
        printf("  -> %d distinct CUs\n", total);
        if (total <= 24) { for (auto& kv : perXcc) { printf("   xcc%d se*100+sh*50+cu:", kv.first); for (int c : kv.second) printf(" %d", c); printf("\n"); } }
        hipStreamDestroy(st);
    };
    std::vector<uint32_t> all(words, 0xFFFFFFFFu);
    run("all bits", all);
    { std::vector<uint32_t> m(words, 0); m[0] = 0xFFu; run("bits 0-7", m); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0xFFFFu; run("bits 0-15", m); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0xFFFF0000u; run("bits 16-31", m); }
    { std::vector<uint32_t> m(words, 0); for (int i = 0; i < cus; i += 16) m[i / 32] |= 1u << (i % 32); run("every 16th bit", m); }
    { std::vector<uint32_t> m(words, 0); for (int i = 0; i < cus; ++i) if ((i / 8) % 16 != 0) m[i / 32] |= 1u << (i % 32); run("all but bits 8k..8k+7 of every 128", m); }
    { std::vector<uint32_t> m(words, 0xFFFFFFFFu); m[0] &= ~0xFFFFu; run("all but bits 0-15", m); }
    hipFree(d);
    return 0;
}
