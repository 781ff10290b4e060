// Microbenchmark: cost of per-lane 64-byte record gathers through the vector L1 (TCP) on gfx950.
//   A: each lane loads its own 64-B record with 4 x dwordx4 (what a per-lane BVH pair fetch does)
//   B: quad-cooperative: 4 lanes load the 4 x 16-B pieces of one record in ONE instruction; 4 instructions
//      cover the wave's 64 records (same bytes, 4x fewer distinct lines per instruction)
//   C: each lane loads 16 B only (1 x dwordx4)
//   D: like A but all lanes of a wave read the SAME record (uniform)
// hipcc --offload-arch=gfx950 -O3 -o gather gather.hip && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t hashu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(64) void k(const float4* __restrict__ buf, uint32_t mask, int iters, float* __restrict__ out, int coherent)
{
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    float acc = 0.f;
    uint32_t h = hashu(wave * 64u + lane + 1u);
    for (int it = 0; it < iters; ++it) {
        h = hashu(h + (uint32_t)it);
        uint32_t rec = h & mask;
        if (coherent) rec = (hashu(wave * 977u + it) + (lane >> 2)) & mask;   // neighbouring lanes share / sit next to each other
        if (MODE == 0) {
            const float4* p = buf + (size_t)rec * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 1) {
            // record index of ray r lives in lane r; quad q of instruction j serves ray 16*j + q
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t r = __shfl(rec, 16 * j + (lane >> 2), 64);
                v[j] = buf[(size_t)r * 4 + (lane & 3)];
            }
            acc += v[0].x + v[1].y + v[2].z + v[3].w;
        } else if (MODE == 2) {
            float4 a = buf[(size_t)rec * 4];
            acc += a.x;
        } else {
            uint32_t r = __builtin_amdgcn_readfirstlane(rec);
            const float4* p = buf + (size_t)r * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        }
        // dependent chain like a traversal: next index depends on loaded data
        h += (uint32_t)__float_as_uint(acc) & 1u;
    }
    out[wave * 64 + lane] = acc;
}

template <int MODE>
float run(const float4* buf, uint32_t mask, int iters, float* out, int coherent, int waves)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<waves, 64>>>(buf, mask, 8, out, coherent);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<waves, 64>>>(buf, mask, iters, out, coherent);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int waves = 256 * 20 * 4, iters = 256;
    float* out; hipMalloc(&out, waves * 64 * sizeof(float));
    const char* names[4] = { "A per-lane 4x16B", "B quad-coop 4x16B", "C per-lane 16B", "D uniform 64B" };
    for (size_t mb : { (size_t)1, (size_t)16, (size_t)128, (size_t)1024 }) {
        const size_t recs = mb * 1024 * 1024 / 64;
        float4* buf; hipMalloc(&buf, recs * 64); hipMemset(buf, 0, recs * 64);
        for (int coherent = 0; coherent < 2; ++coherent) {
            float ms[4] = { run<0>(buf, (uint32_t)recs - 1, iters, out, coherent, waves), run<1>(buf, (uint32_t)recs - 1, iters, out, coherent, waves),
                            run<2>(buf, (uint32_t)recs - 1, iters, out, coherent, waves), run<3>(buf, (uint32_t)recs - 1, iters, out, coherent, waves) };
            for (int m = 0; m < 4; ++m) {
                const double wl = (double)waves * iters;       // wave-level record fetches
                printf("table %5zu MB %s  %-18s: %8.3f ms  %7.1f M wave-fetch/s  %6.0f cycles/wave-fetch/CU @2.3GHz\n", mb, coherent ? "coherent" : "random  ",
                       names[m], ms[m], wl / ms[m] / 1e3, ms[m] * 1e-3 * 2.3e9 * 256 / wl);
            }
        }
        hipFree(buf);
    }
    return 0;
}
