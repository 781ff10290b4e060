// Exhaustive check (all 2^32 float bit patterns) of candidate replacements for the IEEE reciprocal `1.0f / x` that hipcc expands to
// v_div_scale x2, v_rcp, 4-5 fma/mul, v_div_fmas, v_div_fixup (11 VALU instructions). The trace kernel computes three of them per
// instance entry and one per triangle test (native_recip / 1.0f / a pinned to IEEE division, oracle/crt_oracle.h), so a shorter
// sequence is only admissible if it returns the SAME BITS for every input -- zeros, denormals, infinities and every NaN included
// (a NaN `t` travels into hit records that the tests compare bit for bit).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o recip_exhaustive recip_exhaustive.hip && ./recip_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float ref_recip(float x) { return 1.0f / x; }

// A: one Newton step on v_rcp_f32 + v_div_fixup for the special inputs
__device__ __forceinline__ float cand_a(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    return __builtin_amdgcn_div_fixupf(y1, x, 1.0f);
}
// B: two Newton steps + fixup
__device__ __forceinline__ float cand_b(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float e1 = __builtin_fmaf(-x, y1, 1.0f);
    const float y2 = __builtin_fmaf(e1, y1, y1);
    return __builtin_amdgcn_div_fixupf(y2, x, 1.0f);
}
// C: Markstein: y1 as in A, then a residual correction of the quotient
__device__ __forceinline__ float cand_c(float x)
{
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float r = __builtin_fmaf(-x, y1, 1.0f);
    const float q = __builtin_fmaf(r, y0, y1);
    return __builtin_amdgcn_div_fixupf(q, x, 1.0f);
}

template <int WHICH>
__global__ void sweep(uint32_t base, unsigned long long* __restrict__ count, uint32_t* __restrict__ expLo, uint32_t* __restrict__ expHi, uint32_t* __restrict__ example, unsigned int* __restrict__ hist)
{
    const uint32_t bits = base + blockIdx.x * 256u + threadIdx.x;
    const float x = __uint_as_float(bits);
    const float r = ref_recip(x);
    const float c = WHICH == 0 ? cand_a(x) : (WHICH == 1 ? cand_b(x) : cand_c(x));
    if (__float_as_uint(r) != __float_as_uint(c)) {
        atomicAdd(count, 1ull);
        const uint32_t e = (bits >> 23) & 0xFFu;
        atomicMin(expLo, e); atomicMax(expHi, e); atomicAdd(&hist[e], 1u);
        *example = bits;
    }
}

int main()
{
    unsigned long long* dCount; uint32_t *dLo, *dHi, *dEx;
    hipMalloc(&dCount, 8); hipMalloc(&dLo, 4); hipMalloc(&dHi, 4); hipMalloc(&dEx, 4);
    unsigned int* dHist; hipMalloc(&dHist, 256 * 4);
    const char* names[3] = { "A: rcp + 1 Newton step + div_fixup (4 instructions)", "B: rcp + 2 Newton steps + div_fixup (6)", "C: rcp + Newton + residual correction + div_fixup (6)" };
    for (int which = 0; which < 3; ++which) {
        unsigned long long zero = 0; uint32_t lo = 255, hi = 0, ex = 0;
        hipMemcpy(dCount, &zero, 8, hipMemcpyHostToDevice); hipMemcpy(dLo, &lo, 4, hipMemcpyHostToDevice); hipMemcpy(dHi, &hi, 4, hipMemcpyHostToDevice); hipMemcpy(dEx, &ex, 4, hipMemcpyHostToDevice); hipMemset(dHist, 0, 256 * 4);
        for (uint32_t chunk = 0; chunk < 256; ++chunk) {          // 256 launches of 2^24 inputs
            const uint32_t base = chunk << 24;
            if (which == 0) sweep<0><<<65536, 256>>>(base, dCount, dLo, dHi, dEx, dHist);
            else if (which == 1) sweep<1><<<65536, 256>>>(base, dCount, dLo, dHi, dEx, dHist);
            else sweep<2><<<65536, 256>>>(base, dCount, dLo, dHi, dEx, dHist);
        }
        hipDeviceSynchronize();
        unsigned long long n; hipMemcpy(&n, dCount, 8, hipMemcpyDeviceToHost); hipMemcpy(&lo, dLo, 4, hipMemcpyDeviceToHost); hipMemcpy(&hi, dHi, 4, hipMemcpyDeviceToHost); hipMemcpy(&ex, dEx, 4, hipMemcpyDeviceToHost);
        printf("%s: %llu of 4294967296 inputs differ from 1.0f / x", names[which], n);
        if (n) printf(" (biased exponents of the differing inputs: %u..%u; one example: 0x%08x)", lo, hi, ex);
        printf("\n");
        unsigned int h[256]; hipMemcpy(h, dHist, sizeof h, hipMemcpyDeviceToHost);
        printf("   differing inputs by biased exponent (exponents with any):");
        for (int e = 0; e < 256; ++e) if (h[e]) printf(" %d:%u", e, h[e]);
        printf("\n");
    }
    return 0;
}
