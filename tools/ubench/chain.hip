// Microbenchmark: the ceiling of DEPENDENT 64-byte record gathers on gfx950 -- what binds crt_trace_kernel.
//
// Every lane of every wave chases its own chain through a table of 64-B child-pair records laid out like the trace kernel's
// `pairs` array {L.min,L.ref | L.max,- | R.min,R.ref | R.max,-}: fetch the record (4 x global_load_dwordx4), run the inner
// step's arithmetic on it (two slab tests against the lane's ray = IntersectAABB of kernel_main.cl:108-117, near/far
// ordering, an LDS stack push), and follow the nearer child's reference -- the next address is known only when that
// arithmetic is done, exactly as in Traversal::inner (clraytracer_amd/csrc/crt_device.h). No leaves, no shading, no idle
// lanes, no tails: the chip is full of chains for the whole launch (8 waves per SIMD x 64 lanes = the trace kernel's
// occupancy), so records per cycle per CU here is the rate that chain shape can reach at a given cache-hit mix.
//
// The hit mix is built into the table: a child reference points into a HOT region (128 records, resident in every CU's
// 32 KiB vector L1), a WARM region (1 MiB: misses L1, resident in every XCD's 4 MiB L2) or a COLD region (128 MiB: misses L2,
// resident in the 256 MiB Infinity Cache) with probabilities chosen per run. The trace kernel's measured mix on multi-1M
// (profiles/r02_summary.md: 89 % of vector L1 line accesses hit; a record is 4 accesses of which the last 3 always hit, so
// 44 % of the RECORDS miss L1; 72 % of L1 misses hit L2) is hot 0.56 / warm 0.317 / cold 0.123.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o chain chain.hip && ./chain [json-path]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define HOT_RECS 128u
#define WARM_RECS (1u << 14)      // 1 MiB
#define COLD_RECS (1u << 21)      // 128 MiB
#define TOTAL_RECS (HOT_RECS + WARM_RECS + COLD_RECS)

struct Rng { uint64_t s; uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); } float unit() { return (float)(next() & 0xFFFFFF) / 16777216.0f; } };

static uint32_t pick(Rng& r, float pHot, float pWarm)
{
    const float u = r.unit();
    if (u < pHot) return r.next() % HOT_RECS;
    if (u < pHot + pWarm) return HOT_RECS + r.next() % WARM_RECS;
    return HOT_RECS + WARM_RECS + r.next() % COLD_RECS;
}

// the inner step of the trace kernel, on a record of the same layout
__device__ __forceinline__ float slab(float ox, float oy, float oz, float ix, float iy, float iz, float4 bmin, float4 bmax, float minSoFar)
{
    float tminx = (bmin.x - ox) * ix, tminy = (bmin.y - oy) * iy, tminz = (bmin.z - oz) * iz;
    float tmaxx = (bmax.x - ox) * ix, tmaxy = (bmax.y - oy) * iy, tmaxz = (bmax.z - oz) * iz;
    float tnear = fmaxf(fmaxf(fminf(tminx, tmaxx), fminf(tminy, tmaxy)), fminf(tminz, tmaxz));
    float tfar = fminf(fminf(fmaxf(tminx, tmaxx), fmaxf(tminy, tmaxy)), fmaxf(tminz, tmaxz));
    return (tnear < tfar && tnear > 0.0f && tnear < minSoFar) ? tnear : 1e30f;
}

// stamps[wave] = {s_memrealtime start, end, s_memtime cycles}
// LOADS: dwordx4 loads per record (4 = the whole 64-B record as the trace kernel reads it; 1 and 2 are diagnostics that read only
// the first 16 / 32 bytes and reuse them: is the cost per instruction or per record?). SHARE: lanes that walk the same chain
// (1 = every lane its own; 4 = groups of four neighbouring lanes share a ray and therefore every record: is the cost per lane
// or per distinct record?)
// KIND: 0 = global_load_dwordx4 (the trace kernel's), 1 = buffer_load_dwordx4 with a 32-bit per-lane offset (offen) against an
// SGPR buffer resource, 2 = global_load_dwordx2 (8 of every 16 bytes), 3 = global_load_dword (4 of every 16 bytes): is the fixed
// cost of a vector load its address (64 x 64-bit addresses) or its data (64 x 16 B)?
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__device__ __forceinline__ float4 load16(const float4* __restrict__ recs, __amdgpu_buffer_rsrc_t rsrc, uint32_t idx16)
{
    if (KIND == 1) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(idx16 * 16u), 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    } else if (KIND == 2) {
        const float2 v = *reinterpret_cast<const float2*>(recs + idx16);
        return make_float4(v.x, v.y, v.x + 0.1f, v.y);
    } else if (KIND == 3) {
        const float v = *reinterpret_cast<const float*>(recs + idx16);
        return make_float4(v, v + 0.2f, v + 0.1f, v);
    }
    return recs[idx16];
}
template <int LOADS, int SHARE, int KIND = 0>
__global__ __launch_bounds__(64, 8) void chain_kernel(const float4* __restrict__ recs, int hops, uint32_t activeLanes, uint32_t* __restrict__ out,
                                                      unsigned long long* __restrict__ stamps)
{
    __shared__ uint32_t s_stack[20 * 64];      // the trace kernel's 5 KiB per wave: 32 waves fill the CU's 160 KiB
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    if (lane < activeLanes) {
        // SHARE == 0 (round 5): 1.5 lanes per chain -- lanes {0},{1,2},{3},{4,5},... -- the trace kernel's measured coherence (28 working lanes touch 18.45 lines)
        uint32_t h = (wave * 64u + (SHARE == 0 ? (2u * lane + 1u) / 3u : lane / (SHARE == 0 ? 1 : SHARE))) * 2654435761u + 12345u;
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
        // a ray per lane: origin outside the unit cube the boxes live in, direction into it
        const float ox = -1.5f - (float)(h & 255u) * (1.0f / 256.0f), oy = 0.3f + (float)((h >> 8) & 255u) * (0.4f / 256.0f), oz = 0.3f + (float)((h >> 16) & 255u) * (0.4f / 256.0f);
        const float dx = 1.0f, dy = ((float)((h >> 4) & 255u) - 127.5f) * (0.2f / 256.0f), dz = ((float)((h >> 12) & 255u) - 127.5f) * (0.2f / 256.0f);
        const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
        uint32_t ref = h % TOTAL_RECS;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(recs), 0, (int)0x7FFFFFFF, 0x00027000);
        float best = 1e30f;
        int sp = 0;
        for (int it = 0; it < hops; ++it) {
            float4 lmin = load16<KIND>(recs, rsrc, ref * 4u);
            if (KIND >= 2) lmin.w = __uint_as_float((__float_as_uint(lmin.x) * 2654435761u + (uint32_t)it) % TOTAL_RECS);      // narrow loads do not reach the reference: keep the chain going
            const float4 lmax = LOADS > 1 ? load16<KIND>(recs, rsrc, ref * 4u + 1u) : make_float4(lmin.x + 0.3f, lmin.y + 0.3f, lmin.z + 0.3f, 0.f);
            float4 rmin = LOADS > 2 ? load16<KIND>(recs, rsrc, ref * 4u + 2u) : make_float4(lmin.y, lmin.z, lmin.x, lmax.w);
            const float4 rmax = LOADS > 2 ? load16<KIND>(recs, rsrc, ref * 4u + 3u) : make_float4(lmax.y, lmax.z, lmax.x, 0.f);
            if (KIND >= 2) rmin.w = __uint_as_float((__float_as_uint(rmin.x) * 2246822519u + (uint32_t)it) % TOTAL_RECS);
            float d1 = slab(ox, oy, oz, ix, iy, iz, lmin, lmax, best);
            float d2 = slab(ox, oy, oz, ix, iy, iz, rmin, rmax, best);
            uint32_t nearRef = __float_as_uint(lmin.w), farRef = LOADS > 2 ? __float_as_uint(rmin.w) : (__float_as_uint(lmin.w) * 2654435761u) % TOTAL_RECS;
            if (d1 > d2) { float t = d1; d1 = d2; d2 = t; uint32_t u = nearRef; nearRef = farRef; farRef = u; }
            if (d2 != 1e30f) { s_stack[(sp & 15) * 64 + lane] = farRef; sp++; }         // push the far child
            else if (d1 == 1e30f && sp > 0) { --sp; acc += s_stack[(sp & 15) * 64 + lane] & 1u; }   // a pop's LDS read (the chain itself goes on with nearRef)
            ref = nearRef < TOTAL_RECS ? nearRef : TOTAL_RECS - 1u;      // never leave the table, whatever a diagnostic variant made of the reference
            acc += (uint32_t)(d1 != 1e30f);
        }
        acc += ref;
    }
    out[wave * 64 + lane] = acc;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { stamps[wave * 3 + 0] = r0; stamps[wave * 3 + 1] = r1; stamps[wave * 3 + 2] = c1 - c0; }
}

// ---- round 4: ROW-TRANSPOSED record fetch ---------------------------------------------------------------------
// The four dwordx4 loads of a record each touch the lines of ALL chasing lanes' records (the distinct-lines term of the cost line is
// paid four times). Transposed: load i is issued by all 64 lanes for the records of ROW i (lanes 16i..16i+15) -- lane (r, c) fetches
// quarter r of the record lane (i, c) wants, so one instruction touches 16 records instead of 64 -- and a 4x4 transpose across the
// wave's four rows puts every record back into its owner's registers: gfx950's v_permlane16_swap / v_permlane32_swap exchange
// rows between two registers in place, 16 instructions for the 16 dwords, no LDS, no select masks. (The quad form the VERDICT
// sketched needs the register index to depend on the lane -- 48 cndmask/DPP instructions; rows need none.) Lanes that are not
// chasing take part in the loads with the first lane's reference. MASKED: chasing lanes given by a 64-bit mask (scattered idle
// lanes, as in the trace kernel) instead of `lane < activeLanes`.
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) { const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = r[0]; b = r[1]; }
__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b) { const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = r[0]; b = r[1]; }
__device__ __forceinline__ void swap16(float4& a, float4& b)
{
    uint32_t* x = reinterpret_cast<uint32_t*>(&a); uint32_t* y = reinterpret_cast<uint32_t*>(&b);
#pragma unroll
    for (int k = 0; k < 4; ++k) swap16(x[k], y[k]);
}
__device__ __forceinline__ void swap32(float4& a, float4& b)
{
    uint32_t* x = reinterpret_cast<uint32_t*>(&a); uint32_t* y = reinterpret_cast<uint32_t*>(&b);
#pragma unroll
    for (int k = 0; k < 4; ++k) swap32(x[k], y[k]);
}
// every lane calls this (uniform control flow); `ref` must be a valid record index in every lane
__device__ __forceinline__ void fetch_transposed(const float4* __restrict__ recs, uint32_t ref, uint32_t row, float4& lmin, float4& lmax, float4& rmin, float4& rmax)
{
    uint32_t a = ref, b = ref;
    swap32(a, b);                        // a = rows {0,1,0,1} of ref, b = rows {2,3,2,3}
    uint32_t a2 = a, b2 = b;
    swap16(a, a2);                       // a = row 0's refs in every row, a2 = row 1's
    swap16(b, b2);                       // b = row 2's, b2 = row 3's
    lmin = recs[(size_t)a * 4u + row];
    lmax = recs[(size_t)a2 * 4u + row];
    rmin = recs[(size_t)b * 4u + row];
    rmax = recs[(size_t)b2 * 4u + row];
    swap16(lmin, lmax); swap16(rmin, rmax);
    swap32(lmin, rmin); swap32(lmax, rmax);
}
// QUAD-TRANSPOSED (the form VERDICT r3 #1 names): the four lanes of a quad fetch ONE record per instruction -- instruction i serves
// the quad's lane i, lane q reading quarter q^i of it -- so a quad costs the L1 one tag access per instruction and active lane
// instead of one per distinct line and instruction. The tag-access count is what the `lines` of the r3 cost line measures (16 lanes
// on one chain read 4 distinct lines per instruction but the counter says 16.0: one per quad), which is also why the ROW form above
// gains nothing: a row-transposed instruction still has four different records under every quad. Getting the record back to its
// owner is a 4x4 transpose inside the quad, and DPP cannot index registers by lane: the XOR assignment of quarters turns it into two
// lane-local butterfly stages (v_cndmask on lane bits 0 and 1: 28 selects for the 14 live dwords) after which quarter k of lane i's
// record sits in register group k of lane i^k for EVERY i -- a fixed quad_perm per group, folded into the consuming v_sub as a DPP
// operand (which needs the neighbour lanes enabled: the slab tests run for whole quads).
template <int CTRL>
__device__ __forceinline__ float dppf(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true)); }
template <int CTRL>
__device__ __forceinline__ float4 dpp4(const float4& v) { return make_float4(dppf<CTRL>(v.x), dppf<CTRL>(v.y), dppf<CTRL>(v.z), dppf<CTRL>(v.w)); }
__device__ __forceinline__ float4 sel4(bool c, const float4& a, const float4& b) { return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
// every lane of a quad with a chasing lane calls this; `active` = this lane wants its record
__device__ __forceinline__ void fetch_quad(const float4* __restrict__ recs, uint32_t ref, bool active, uint32_t q, float4& lmin, float4& lmax, float4& rmin, float4& rmax)
{
    const int a = active ? 1 : 0;
    float4 g0, g1, g2, g3;       // a quarter nobody asked for stays undefined: it is never looked at
    // quad_perm:[i,i,i,i] = i * 0x55
    if (__builtin_amdgcn_update_dpp(0, a, 0x00, 0xF, 0xF, true)) g0 = recs[(size_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)ref, 0x00, 0xF, 0xF, true) * 4u + (q ^ 0u)];
    if (__builtin_amdgcn_update_dpp(0, a, 0x55, 0xF, 0xF, true)) g1 = recs[(size_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)ref, 0x55, 0xF, 0xF, true) * 4u + (q ^ 1u)];
    if (__builtin_amdgcn_update_dpp(0, a, 0xAA, 0xF, 0xF, true)) g2 = recs[(size_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)ref, 0xAA, 0xF, 0xF, true) * 4u + (q ^ 2u)];
    if (__builtin_amdgcn_update_dpp(0, a, 0xFF, 0xF, 0xF, true)) g3 = recs[(size_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)ref, 0xFF, 0xF, 0xF, true) * 4u + (q ^ 3u)];
    const bool b0 = q & 1u, b1 = q & 2u;
    const float4 x0 = sel4(b0, g1, g0), x1 = sel4(b0, g0, g1), x2 = sel4(b0, g3, g2), x3 = sel4(b0, g2, g3);      // x[g] = g[g ^ (q & 1)]
    const float4 y0 = sel4(b1, x2, x0), y1 = sel4(b1, x3, x1), y2 = sel4(b1, x0, x2), y3 = sel4(b1, x1, x3);      // y[k] = g[k ^ q]: quarter k of record k ^ q
    lmin = y0; lmax = y1; rmin = y2; rmax = y3;       // quarter k is still in lane i ^ k: slab_quad reads it through DPP operands
}
// dpp(b) - o with the quad permutation as an operand modifier of the subtraction itself (the compiler's DPP combiner leaves a
// v_mov_b32_dpp in front of every consumer here)
#define CRT_SUB_DPP(NAME, PERM) \
__device__ __forceinline__ float NAME(float b, float o) { float r; asm volatile("v_sub_f32_dpp %0, %1, %2 " PERM " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(b), "v"(o)); return r; }
CRT_SUB_DPP(sub_x1, "quad_perm:[1,0,3,2]")
CRT_SUB_DPP(sub_x2, "quad_perm:[2,3,0,1]")
CRT_SUB_DPP(sub_x3, "quad_perm:[3,2,1,0]")
// both slab tests of the inner step on the un-permuted quarters y0..y3 of fetch_quad; *lref / *rref receive the child references
__device__ __forceinline__ void slab_quad(float ox, float oy, float oz, float ix, float iy, float iz, const float4& y0, const float4& y1, const float4& y2, const float4& y3,
                                          float minSoFar, float& d1, float& d2, uint32_t& lref, uint32_t& rref)
{
    {
        const float tminx = (y0.x - ox) * ix, tminy = (y0.y - oy) * iy, tminz = (y0.z - oz) * iz;
        const float tmaxx = sub_x1(y1.x, ox) * ix, tmaxy = sub_x1(y1.y, oy) * iy, tmaxz = sub_x1(y1.z, oz) * iz;
        const float tnear = fmaxf(fmaxf(fminf(tminx, tmaxx), fminf(tminy, tmaxy)), fminf(tminz, tmaxz));
        const float tfar = fminf(fminf(fmaxf(tminx, tmaxx), fmaxf(tminy, tmaxy)), fmaxf(tminz, tmaxz));
        d1 = (tnear < tfar && tnear > 0.0f && tnear < minSoFar) ? tnear : 1e30f;
    }
    {
        const float tminx = sub_x2(y2.x, ox) * ix, tminy = sub_x2(y2.y, oy) * iy, tminz = sub_x2(y2.z, oz) * iz;
        const float tmaxx = sub_x3(y3.x, ox) * ix, tmaxy = sub_x3(y3.y, oy) * iy, tmaxz = sub_x3(y3.z, oz) * iz;
        const float tnear = fmaxf(fmaxf(fminf(tminx, tmaxx), fminf(tminy, tmaxy)), fminf(tminz, tmaxz));
        const float tfar = fminf(fminf(fmaxf(tminx, tmaxx), fmaxf(tminy, tmaxy)), fmaxf(tminz, tmaxz));
        d2 = (tnear < tfar && tnear > 0.0f && tnear < minSoFar) ? tnear : 1e30f;
    }
    lref = __float_as_uint(y0.w);
    rref = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(y2.w), 0x4E, 0xF, 0xF, true);
}
// TRANSPOSED: 0 = per-lane fetch, 1 = rows, 2 = quads
// TRANSPOSED = false: the same kernel with the per-lane fetch, for an apples-to-apples pair under the same lane mask
__device__ __forceinline__ uint32_t differs(const float4& a, const float4& b)
{
    return (uint32_t)(__float_as_uint(a.x) != __float_as_uint(b.x) || __float_as_uint(a.y) != __float_as_uint(b.y) || __float_as_uint(a.z) != __float_as_uint(b.z) ||
                      __float_as_uint(a.w) != __float_as_uint(b.w));
}
template <int TRANSPOSED, bool CHECK>
__global__ __launch_bounds__(64, 8) void chain_masked_kernel(const float4* __restrict__ recs, int hops, unsigned long long laneMask, uint32_t* __restrict__ out,
                                                             unsigned long long* __restrict__ stamps, uint32_t* __restrict__ mismatches)
{
    __shared__ uint32_t s_stack[20 * 64];
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    uint32_t acc = 0;
    const bool chasing = (laneMask >> lane) & 1ull;
    unsigned long long quadMask = laneMask | (laneMask >> 1) | (laneMask >> 2) | (laneMask >> 3);
    quadMask &= 0x1111111111111111ull; quadMask *= 15ull;      // every lane of a quad with a chasing lane
    uint32_t h = (wave * 64u + lane) * 2654435761u + 12345u;
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    const float ox = -1.5f - (float)(h & 255u) * (1.0f / 256.0f), oy = 0.3f + (float)((h >> 8) & 255u) * (0.4f / 256.0f), oz = 0.3f + (float)((h >> 16) & 255u) * (0.4f / 256.0f);
    const float dx = 1.0f, dy = ((float)((h >> 4) & 255u) - 127.5f) * (0.2f / 256.0f), dz = ((float)((h >> 12) & 255u) - 127.5f) * (0.2f / 256.0f);
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    uint32_t ref = h % TOTAL_RECS;
    float best = 1e30f;
    int sp = 0;
    uint32_t bad = 0;
    for (int it = 0; it < hops; ++it) {
        float4 lmin, lmax, rmin, rmax;
        if (TRANSPOSED == 1) {
            // idle lanes lend their address slots: they ask for the record of the wave's first chasing lane
            const uint32_t ref0 = (uint32_t)__builtin_amdgcn_readlane((int)ref, __ffsll((long long)laneMask) - 1);
            fetch_transposed(recs, chasing ? ref : ref0, lane >> 4, lmin, lmax, rmin, rmax);
        }
        float d1 = 1e30f, d2 = 1e30f;
        uint32_t qLref = 0, qRref = 0;
        if (TRANSPOSED == 2) {
            // whole quads: the fetch and the slab tests run for every lane of a quad that has a chasing lane (the DPP operands of the
            // tests read the neighbours' registers, which must be enabled); an idle lane's results are never used
            if ((quadMask >> lane) & 1ull) {
                fetch_quad(recs, ref, chasing, lane & 3u, lmin, lmax, rmin, rmax);
                slab_quad(ox, oy, oz, ix, iy, iz, lmin, lmax, rmin, rmax, best, d1, d2, qLref, qRref);
                if (CHECK) { lmax = dpp4<0xB1>(lmax); rmin = dpp4<0x4E>(rmin); rmax = dpp4<0x1B>(rmax); }
            }
        }
        if (CHECK && TRANSPOSED && chasing) {       // self-check launch: the transposed record equals the directly loaded one
            const float4* p = recs + (size_t)ref * 4u;
            bad += differs(p[0], lmin) + differs(p[1], lmax) + differs(p[2], rmin) + differs(p[3], rmax);
        }
        if (chasing) {
            if (!TRANSPOSED) { const float4* p = recs + (size_t)ref * 4u; lmin = p[0]; lmax = p[1]; rmin = p[2]; rmax = p[3]; }
            if (TRANSPOSED != 2) {
                d1 = slab(ox, oy, oz, ix, iy, iz, lmin, lmax, best);
                d2 = slab(ox, oy, oz, ix, iy, iz, rmin, rmax, best);
            }
            uint32_t nearRef = TRANSPOSED == 2 ? qLref : __float_as_uint(lmin.w), farRef = TRANSPOSED == 2 ? qRref : __float_as_uint(rmin.w);
            if (d1 > d2) { float t = d1; d1 = d2; d2 = t; uint32_t u = nearRef; nearRef = farRef; farRef = u; }
            if (d2 != 1e30f) { s_stack[(sp & 15) * 64 + lane] = farRef; sp++; }
            else if (d1 == 1e30f && sp > 0) { --sp; acc += s_stack[(sp & 15) * 64 + lane] & 1u; }
            ref = nearRef < TOTAL_RECS ? nearRef : TOTAL_RECS - 1u;
            acc += (uint32_t)(d1 != 1e30f);
        }
    }
    acc += ref;
    out[wave * 64 + lane] = acc;
    if (CHECK) { atomicAdd(mismatches, bad); atomicAdd(mismatches + 1, chasing ? (uint32_t)hops * 4u : 0u); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { stamps[wave * 3 + 0] = r0; stamps[wave * 3 + 1] = r1; stamps[wave * 3 + 2] = c1 - c0; }
}

struct Result { double ms, clockGhz, recPerCyclePerCu, cyclesPerHop; };

template <int LOADS, int SHARE, int KIND = 0>
static Result run(const float4* dRecs, int waves, int hops, uint32_t activeLanes, uint32_t* dOut, unsigned long long* dStamps, int numCus)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chain_kernel<LOADS, SHARE, KIND><<<waves, 64>>>(dRecs, 16, activeLanes, dOut, dStamps);      // warm the caches / clocks
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain_kernel<LOADS, SHARE, KIND><<<waves, 64>>>(dRecs, hops, activeLanes, dOut, dStamps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st((size_t)waves * 3);
    hipMemcpy(st.data(), dStamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    // in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS item 6), median over waves;
    // launch extent = last end - first start on the 100 MHz real-time counter
    std::vector<double> clk; unsigned long long first = ~0ull, last = 0; double cycSum = 0;
    for (int w = 0; w < waves; ++w) {
        const unsigned long long a = st[(size_t)w * 3], b = st[(size_t)w * 3 + 1], c = st[(size_t)w * 3 + 2];
        if (b > a) clk.push_back((double)c / (double)(b - a) * 0.1);
        first = std::min(first, a); last = std::max(last, b); cycSum += (double)c;
    }
    std::sort(clk.begin(), clk.end());
    Result r;
    r.clockGhz = clk.empty() ? 0.0 : clk[clk.size() / 2];
    const double extentS = (double)(last - first) * 1e-8;                     // 100 MHz ticks
    r.ms = extentS * 1e3;
    (void)ms;
    const double recs = (double)waves * activeLanes * hops;      // lane-level record fetches (SHARE > 1: several lanes fetch the same record)
    r.recPerCyclePerCu = recs / (extentS * r.clockGhz * 1e9 * numCus);
    r.cyclesPerHop = cycSum / waves / hops;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return r;
}

template <int TRANSPOSED>
static Result run_masked(const float4* dRecs, int waves, int hops, unsigned long long laneMask, uint32_t* dOut, unsigned long long* dStamps, int numCus, uint32_t* dMismatch)
{
    if (TRANSPOSED) {       // one short checking launch first: every quarter of every fetched record against the direct load
        hipMemset(dMismatch, 0, 8);
        chain_masked_kernel<TRANSPOSED, true><<<waves, 64>>>(dRecs, 8, laneMask, dOut, dStamps, dMismatch);
        uint32_t mm[2] = { 0, 0 }; hipMemcpy(mm, dMismatch, 8, hipMemcpyDeviceToHost);
        if (mm[0] != 0 || mm[1] == 0) { printf("TRANSPOSED FETCH SELF-CHECK FAILED: %u of %u quarters differ\n", mm[0], mm[1]); exit(1); }
    }
    chain_masked_kernel<TRANSPOSED, false><<<waves, 64>>>(dRecs, 16, laneMask, dOut, dStamps, nullptr);
    hipDeviceSynchronize();
    chain_masked_kernel<TRANSPOSED, false><<<waves, 64>>>(dRecs, hops, laneMask, dOut, dStamps, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st((size_t)waves * 3);
    hipMemcpy(st.data(), dStamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> clk; unsigned long long first = ~0ull, last = 0; double cycSum = 0;
    for (int w = 0; w < waves; ++w) {
        const unsigned long long a = st[(size_t)w * 3], b = st[(size_t)w * 3 + 1], c = st[(size_t)w * 3 + 2];
        if (b > a) clk.push_back((double)c / (double)(b - a) * 0.1);
        first = std::min(first, a); last = std::max(last, b); cycSum += (double)c;
    }
    std::sort(clk.begin(), clk.end());
    Result r;
    r.clockGhz = clk.empty() ? 0.0 : clk[clk.size() / 2];
    const double extentS = (double)(last - first) * 1e-8;
    r.ms = extentS * 1e3;
    const double recs = (double)waves * (double)__builtin_popcountll(laneMask) * hops;
    r.recPerCyclePerCu = recs / (extentS * r.clockGhz * 1e9 * numCus);
    r.cyclesPerHop = cycSum / waves / hops;
    return r;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int numCus = prop.multiProcessorCount;
    float4* dRecs; hipMalloc(&dRecs, (size_t)TOTAL_RECS * 64);
    uint32_t* dOut; unsigned long long* dStamps;
    const int maxWaves = numCus * 32;
    hipMalloc(&dOut, (size_t)maxWaves * 64 * 4); hipMalloc(&dStamps, (size_t)maxWaves * 3 * 8);
    uint32_t* dMismatch; hipMalloc(&dMismatch, 8);
    struct Mix { const char* name; float hot, warm; };
    // the trace kernel's hit mix: CHAIN_MIX="<L1 line hit rate>,<L2 hit rate>" (fractions, from the newest profiles/r*_summary.json: l1_hit_rate,
    // l2_hit_rate) replaces the round-2 figures: a record is 4 line accesses of which the last 3 always hit, so 4 x (1 - l1) of the RECORDS miss L1
    float l1 = 0.89f, l2 = 0.72f;
    if (const char* e = getenv("CHAIN_MIX")) { float a = 0, b = 0; if (sscanf(e, "%f,%f", &a, &b) == 2 && a > 0.75f && a <= 1.0f && b >= 0.0f && b <= 1.0f) { l1 = a; l2 = b; } }
    static char mixName[128];
    snprintf(mixName, sizeof mixName, "kernel mix (multi-1M: %.1f %% L1 line hits, %.1f %% L2 hits)", l1 * 100.0f, l2 * 100.0f);
    const float missRec = 4.0f * (1.0f - l1);
    const Mix mixes[] = { { mixName, 1.0f - missRec, missRec * l2 },
                          { "all records L1-resident", 1.0f, 0.0f },
                          { "all records from L2", 0.0f, 1.0f },
                          { "all records from the Infinity Cache", 0.0f, 0.0f } };
    std::vector<float> host((size_t)TOTAL_RECS * 16);
    // --brief: only the kernel's mix at 8 waves/SIMD (64 and 28 chasing lanes, 16 lanes per chain) -- the runs a PMC pass of this binary is
    // read for (tools/ubench_pmc.sh: is the vector-memory path saturated at the ceiling?)
    const bool brief = argc > 1 && strcmp(argv[1], "--brief") == 0;
    const bool onlyT = argc > 1 && strcmp(argv[1], "--transposed") == 0;       // ./chain --transposed [json]: only the round-4 pairs
    const bool briefT = argc > 1 && strcmp(argv[1], "--brief-transposed") == 0;
    const bool refillRows = argc > 1 && strcmp(argv[1], "--refill") == 0;
    FILE* js = (argc > 1 && !brief && !onlyT && !briefT && !refillRows) ? fopen(argv[1], "w") : ((onlyT || refillRows) && argc > 2 ? fopen(argv[2], "w") : nullptr);
    if (js) fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"runs\": [\n", prop.gcnArchName, numCus);
    bool firstJs = true;
    printf("%s, %d CUs; every wave 64-thread workgroup with 5 KiB LDS, launch_bounds(64, 8); hops per lane 512\n", prop.gcnArchName, numCus);
    for (const Mix& m : mixes) {
        Rng rng{ 0x9E3779B97F4A7C15ull };
        for (uint32_t i = 0; i < TOTAL_RECS; ++i) {
            float* r = &host[(size_t)i * 16];
            for (int b = 0; b < 2; ++b) {
                // boxes inside the unit cube, wide enough that about half the slab tests pass
                const float cx = rng.unit(), cy = 0.2f + 0.6f * rng.unit(), cz = 0.2f + 0.6f * rng.unit(), e = 0.15f + 0.35f * rng.unit();
                r[b * 8 + 0] = cx - e; r[b * 8 + 1] = cy - e; r[b * 8 + 2] = cz - e;
                r[b * 8 + 4] = cx + e; r[b * 8 + 5] = cy + e; r[b * 8 + 6] = cz + e; r[b * 8 + 7] = 0.0f;
                const uint32_t ref = pick(rng, m.hot, m.warm);
                memcpy(&r[b * 8 + 3], &ref, 4);
            }
        }
        hipMemcpy(dRecs, host.data(), host.size() * 4, hipMemcpyHostToDevice);
        printf("\n== %s (hot %.3f / warm %.3f / cold %.3f)\n", m.name, m.hot, m.warm, 1.0f - m.hot - m.warm);
        printf("%-26s %6s %6s %9s %10s %16s %12s\n", "variant", "w/SIMD", "lanes", "ms", "clock GHz", "records/cyc/CU", "cycles/hop");
        auto report = [&](const char* variant, int wps, uint32_t lanes, const Result& r) {
            printf("%-26s %6d %6u %9.3f %10.3f %16.4f %12.0f\n", variant, wps, lanes, r.ms, r.clockGhz, r.recPerCyclePerCu, r.cyclesPerHop);
            if (js) {
                fprintf(js, "%s{\"mix\": \"%s\", \"hot\": %.3f, \"warm\": %.3f, \"variant\": \"%s\", \"waves_per_simd\": %d, \"active_lanes\": %u, \"ms\": %.4f, \"clock_ghz\": %.4f, "
                            "\"records_per_cycle_per_cu\": %.5f, \"cycles_per_hop\": %.1f}", firstJs ? "" : ",\n", m.name, m.hot, m.warm, variant, wps, lanes, r.ms, r.clockGhz,
                        r.recPerCyclePerCu, r.cyclesPerHop);
                firstJs = false;
            }
        };
        // round 4: the row-transposed fetch against the per-lane fetch under the same lane masks (all 64 chasing; 28 scattered over the
        // four rows, 7 each -- the trace kernel's utilisation; 28 contiguous = rows 2,3 idle; 16 = one lane in four)
        auto transposed_rows = [&](const Mix&, auto& rep) {
            struct M { const char* name; unsigned long long mask; };
            const M masks[] = { { "64", ~0ull }, { "28 scattered", 0x2A952A952A952A95ull }, { "28 contiguous", (1ull << 28) - 1ull }, { "16 (every 4th)", 0x1111111111111111ull } };
            for (const M& k : masks) {
                char nm[96];
                const uint32_t lanes = (uint32_t)__builtin_popcountll(k.mask);
                for (int wps : { 8, 4 }) {
                    snprintf(nm, sizeof nm, "per-lane fetch, %s", k.name);
                    rep(nm, wps, lanes, run_masked<0>(dRecs, numCus * 4 * wps, 512, k.mask, dOut, dStamps, numCus, dMismatch));
                    snprintf(nm, sizeof nm, "ROW-TRANSPOSED, %s", k.name);
                    rep(nm, wps, lanes, run_masked<1>(dRecs, numCus * 4 * wps, 512, k.mask, dOut, dStamps, numCus, dMismatch));
                    snprintf(nm, sizeof nm, "QUAD-TRANSPOSED, %s", k.name);
                    rep(nm, wps, lanes, run_masked<2>(dRecs, numCus * 4 * wps, 512, k.mask, dOut, dStamps, numCus, dMismatch));
                }
            }
        };
        if (argc > 1 && strcmp(argv[1], "--transposed") == 0) { transposed_rows(m, report); continue; }
        if (argc > 1 && strcmp(argv[1], "--brief-transposed") == 0) {      // the four launches a PMC pass is read for (tools/ubench_pmc.sh <tag> --brief-transposed)
            if (&m != &mixes[0]) break;
            report("per-lane fetch, 64", 8, 64, run_masked<0>(dRecs, numCus * 32, 512, ~0ull, dOut, dStamps, numCus, dMismatch));
            report("QUAD-TRANSPOSED, 64", 8, 64, run_masked<2>(dRecs, numCus * 32, 512, ~0ull, dOut, dStamps, numCus, dMismatch));
            report("per-lane fetch, 28 scattered", 8, 28, run_masked<0>(dRecs, numCus * 32, 512, 0x2A952A952A952A95ull, dOut, dStamps, numCus, dMismatch));
            report("QUAD-TRANSPOSED, 28 scattered", 8, 28, run_masked<2>(dRecs, numCus * 32, 512, 0x2A952A952A952A95ull, dOut, dStamps, numCus, dMismatch));
            continue;
        }
        if (argc > 1 && strcmp(argv[1], "--refill") == 0) {
            // round 5: what would in-tile lane refill buy the memory side? The kernel's coherence (1.5 lanes per
            // record) at its lane count (28) and at the lane counts a refilled wave might reach, next to fully private chains
            if (&m != &mixes[0]) break;
            for (uint32_t lanes : { 28u, 34u, 40u, 46u, 52u, 58u, 64u }) report("4 x dwordx4, 1.5 lanes/chain", 8, lanes, run<4, 0>(dRecs, numCus * 32, 512, lanes, dOut, dStamps, numCus));
            for (uint32_t lanes : { 28u, 40u, 52u, 64u }) report("4 x dwordx4, 1 lane/chain", 8, lanes, run<4, 1>(dRecs, numCus * 32, 512, lanes, dOut, dStamps, numCus));
            for (uint32_t lanes : { 28u, 40u, 52u, 64u }) report("4 x dwordx4, 2 lanes/chain", 8, lanes, run<4, 2>(dRecs, numCus * 32, 512, lanes, dOut, dStamps, numCus));
            continue;
        }
        if (brief) {
            if (&m != &mixes[0]) break;
            report("4 x dwordx4 (the kernel's)", 8, 64, run<4, 1>(dRecs, numCus * 32, 512, 64, dOut, dStamps, numCus));
            report("4 x dwordx4 (the kernel's)", 8, 28, run<4, 1>(dRecs, numCus * 32, 512, 28, dOut, dStamps, numCus));
            report("4 x dwordx4, 16 lanes/chain", 8, 64, run<4, 16>(dRecs, numCus * 32, 512, 64, dOut, dStamps, numCus));
            continue;
        }
        for (int wps : { 8, 4, 2, 1 })
            for (uint32_t lanes : { 64u, 28u }) report("4 x dwordx4 (the kernel's)", wps, lanes, run<4, 1>(dRecs, numCus * 4 * wps, 512, lanes, dOut, dStamps, numCus));
        // diagnostics at full occupancy: fewer loads per record, shared chains
        report("2 x dwordx4", 8, 64, run<2, 1>(dRecs, numCus * 32, 512, 64, dOut, dStamps, numCus));
        report("1 x dwordx4", 8, 64, run<1, 1>(dRecs, numCus * 32, 512, 64, dOut, dStamps, numCus));
        report("4 x dwordx4, 4 lanes/chain", 8, 64, run<4, 4>(dRecs, numCus * 32, 512, 64, dOut, dStamps, numCus));
        report("4 x dwordx4, 16 lanes/chain", 8, 64, run<4, 16>(dRecs, numCus * 32, 512, 64, dOut, dStamps, numCus));
        report("4 x buffer_load_dwordx4", 8, 64, run<4, 1, 1>(dRecs, numCus * 32, 512, 64, dOut, dStamps, numCus));
        report("4 x buffer_load_dwordx4", 8, 28, run<4, 1, 1>(dRecs, numCus * 32, 512, 28, dOut, dStamps, numCus));
        report("4 x dwordx2", 8, 64, run<4, 1, 2>(dRecs, numCus * 32, 512, 64, dOut, dStamps, numCus));
        report("4 x dword", 8, 64, run<4, 1, 3>(dRecs, numCus * 32, 512, 64, dOut, dStamps, numCus));
        report("4 x dword, 16 lanes/chain", 8, 64, run<4, 16, 3>(dRecs, numCus * 32, 512, 64, dOut, dStamps, numCus));
        report("4 x dwordx4, 16 l/c, 28 lanes", 8, 28, run<4, 16>(dRecs, numCus * 32, 512, 28, dOut, dStamps, numCus));
        transposed_rows(m, report);
    }
    if (js) { fprintf(js, "\n]}\n"); fclose(js); }
    return 0;
}
