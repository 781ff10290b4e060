// crt_ldstile.h -- Trace with the hot BVH tiles staged in LDS (CRT_KERNEL=lds).
//
// One 768-thread workgroup per CU stays resident for the whole frame:
//   * all 12 waves first copy pairs[0 .. CRT_HOT_PAIRS) -- the top levels of every mesh's tree, 64 KiB, renumbered
//     to the lowest pair indices at upload (crt_assign_hot_slots) -- into LDS; one barrier, none afterwards;
//   * each wave then consumes entries of the feedback launch lists (crt_order_kernel: heaviest tiles first, the very
//     heaviest as four quadrant entries) through one atomic counter per XCD list, its own XCD's list first, then the
//     others', and traces each 8x8 tile exactly like crt_trace_kernel;
//   * an inner-node visit whose pair index is below CRT_HOT_PAIRS reads its 64-byte record with ds_read_b128 from LDS,
//     every other visit gathers it through the vector L1 as before.
// LDS: 64 KiB hot tiles + 12 waves x 8 KiB traversal stacks = 160 KiB -> one workgroup (12 waves) per CU.
#pragma once
#include "crt_device.h"

#define CRT_LDS_WAVES 12

typedef const float __attribute__((address_space(3))) * crt_lds_cfloat_ptr;   // explicit LDS pointer: ds_read, not flat_load

struct LdsPairLoader {
    crt_lds_cfloat_ptr hot;   // LDS copy of pairs[0 .. CRT_HOT_PAIRS), 16 floats per pair
    __device__ __forceinline__ void operator()(const CrtDevScene& S, uint32_t ref, float4& lmin, float4& lmax, float4& rmin, float4& rmax) const
    {
        if (ref < (uint32_t)CRT_HOT_PAIRS) {
            crt_lds_cfloat_ptr q = hot + ref * 16;
            lmin = make_float4(q[0], q[1], q[2], q[3]); lmax = make_float4(q[4], q[5], q[6], q[7]);
            rmin = make_float4(q[8], q[9], q[10], q[11]); rmax = make_float4(q[12], q[13], q[14], q[15]);
        } else {
            const float4* p = S.pairs + (size_t)ref * 4;
            lmin = p[0]; lmax = p[1]; rmin = p[2]; rmax = p[3];
        }
    }
};

template <bool COUNT>
__global__ __launch_bounds__(64 * CRT_LDS_WAVES) void crt_trace_lds_kernel(CrtDevScene S, CrtFrame F, float4* __restrict__ out,
                                                                         unsigned long long* __restrict__ counters,
                                                                         uint32_t* __restrict__ listNext)
{
    __shared__ float4 s_hot[CRT_HOT_PAIRS * 4];
    __shared__ uint32_t s_stacks[CRT_LDS_WAVES * CRT_STACK_DEPTH * 64];
    for (int i = threadIdx.x; i < CRT_HOT_PAIRS * 4; i += 64 * CRT_LDS_WAVES) s_hot[i] = S.pairs[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const CrtStack stack = { (crt_lds_u32_ptr)s_stacks + wave * (CRT_STACK_DEPTH * 64) + lane, S.stackOverflow };   // slot s of this lane at lds[s * 64]
    LdsPairLoader loadPair; loadPair.hot = (crt_lds_cfloat_ptr)(const float*)&s_hot[0].x;
    LaneCounters lc; zero_counters(lc);
    const int lx = (int)((lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4));
    const int ly = (int)(((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4));
    const int homeQ = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);   // HW_REG_XCC_ID
    int stolen = 0;

    for (;;) {
        // next launch-list entry: own XCD's list first, then the others'
        uint32_t entry = 0; int xcd = 0; bool got = false;
        while (stolen < 8) {
            xcd = (homeQ + stolen) & 7;
            uint32_t idx = 0;
            if (lane == 0) idx = atomicAdd(&listNext[xcd], 1u);
            idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
            if (idx < F.listLen[xcd]) { entry = F.order[xcd * F.listCap + idx]; got = true; break; }
            ++stolen;
        }
        if (!got) break;
        const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
        const int slot = (int)(entry & 0x0FFFFFFFu);
        const int quadrant = (entry & 0x80000000u) ? (int)((entry >> 28) & 3u) : -1;
        const int round = slot / F.tilesX, tx = slot - round * F.tilesX;
        const int k = round * 8 + xcd;
        if (k >= F.ownedTileRows) continue;                 // padding slot of the last round
        const int bandK = k / F.tileRowsPerBand;
        const int tileRow = (F.rank + bandK * F.nRanks) * F.tileRowsPerBand + (k - bandK * F.tileRowsPerBand);
        const int px = tx * CRT_TILE + lx, py = tileRow * CRT_TILE + ly;
        const bool active = (quadrant < 0 || (int)(lane >> 4) == quadrant) && px < F.width && py < F.height;
        if (active) {
            PathState ps;
            ps.o = mk3(F.camPos[0], F.camPos[1], F.camPos[2]);
            ps.d = raygen_dir(F, px, py);
            ps.result = mk3(0.0f, 0.0f, 0.0f);
            ps.energy = 1.0f;
            for (int bounce = 0; bounce < 2; ++bounce) {
                if (COUNT) { lc.rays++; if (bounce == 0) lc.primary++; else lc.secondary++; }
                Closest c = closest_hit<COUNT, false, LdsPairLoader>(S, ps.o, ps.d, stack, lc, loadPair);
                const bool cont = shade_bounce(S, c, ps, bounce, F.lightY, F.lightZ);
                if (COUNT) { if (cont) lc.hits++; else lc.misses++; }
                if (!cont) break;
            }
            out[(size_t)py * (size_t)F.width + (size_t)px] = make_float4(ps.result.x, ps.result.y, ps.result.z, 1.0f);
        }
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - tc0;
        if (lane == 0) atomicAdd(&F.cost[xcd * F.slotsPerXcd + slot], dt > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)dt);
    }
    if (COUNT) flush_counters(lc, counters);
}
