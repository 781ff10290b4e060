// crt_ldstop.h -- Trace with the HOT BVH TILES STAGED IN LDS (round 6; CRT_KERNEL=ldstop): north_star's "per-wavefront traversal stack and
// hot BVH tiles staged in LDS", on today's kernel. Opt-in, bit-identical to the default kernel, measured beside it.
//
// What is staged: the top levels of EVERY mesh's tree -- CRT_TOP_PAIRS = 252 sibling-pair records of 64 B (15.75 KiB), split evenly over the
// scene's meshes and filled breadth-first from each root (8 meshes: 31 records = 5 levels each; one mesh: 252 records = 7-8 levels) by
// crt_build_top_kernel whenever the BVH layout is rebuilt. Inside the table a child that is itself in the table is referenced as
// CRT_TOP_BIT | record; every other reference is the global one, so a traversal leaves the table exactly once per descent and never comes back
// (pops may return to it: pushed references keep their form). An instance record carries its root's table reference in r2.w.
// How it is used: a workgroup is FOUR waves (four 8x8 tiles next to each other in one XCD's tile row) that share one LDS copy of the table and
// keep 15 stack slots each in LDS (4 x 3.75 KiB + 15.75 KiB = 30.75 KiB: five workgroups = 20 waves per CU, against the default kernel's 32).
// Traversal::inner then has three fetch paths: every lane on the same record -> scalar load from the table's global copy (unchanged); every lane
// on SOME table record -> four ds_read_b128 per lane, no vector-memory instruction; anything else -> the four vector loads, table lanes reading
// the table's global copy. Per ray the sequence of visits, tests and stack contents is the default kernel's (a table record holds the same boxes
// and the same child order), so frames and work counters are bit-identical (tests/test_gpu_variants.py, test_gpu_fuzz.py).
// Loop served: kernel_main.cl:131-158.
#pragma once
#include "crt_kernels.h"

#ifndef CRT_TOP_PAIRS
#define CRT_TOP_PAIRS 252                    // (A/B builds override these three: profiles/r06_ldstop_ab.txt)
#endif
#ifndef CRT_TOP_WAVES
#define CRT_TOP_WAVES 4                      // waves (tiles) per workgroup
#endif
#define CRT_TOP_LDS_SLOTS 15                 // stack slots per wave in LDS; the other 17 of upstream's 32 in the overflow area (CRT_OVF_SLOTS_MAX)
static_assert(CRT_STACK_DEPTH - CRT_TOP_LDS_SLOTS <= CRT_OVF_SLOTS_MAX, "the overflow area is sized for 17 slots per wave");
static_assert(CRT_TOP_PAIRS <= 0xFFFF, "table references keep the record in 16 bits");

// One thread per mesh: breadth-first copy of the tree's top `perMesh` pair records into the mesh's range of the table, children that land in
// the table re-referenced. Runs once per BVH upload / device build (rebuild_bvh_layout); the trees are acyclic by then (crt_relayout_nodes).
__global__ void crt_build_top_kernel(const float4* __restrict__ pairs, const uint32_t* __restrict__ rootRefs, uint32_t numRoots, uint32_t perMesh,
                                     float4* __restrict__ topPairs, uint32_t* __restrict__ topRootRefs)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= CRT_MAX_MESHES) return;
    const uint32_t root = rootRefs[m];
    topRootRefs[m] = root;                                   // a leaf root, a mesh without a tree, no room: the global reference
    if (m >= numRoots || perMesh == 0 || (root & CRT_LEAF_BIT) || (m + 1) * perMesh > CRT_TOP_PAIRS) return;
    const uint32_t base = m * perMesh;
    uint32_t q[CRT_TOP_PAIRS];                               // global pair index of table record base + k (scratch: runs once per upload)
    uint32_t n = 1;
    q[0] = root;
    topRootRefs[m] = CRT_TOP_BIT | base;
    for (uint32_t h = 0; h < n; ++h) {
        const float4* p = pairs + (size_t)q[h] * 4;
        float4 lmin = p[0], lmax = p[1], rmin = p[2], rmax = p[3];
        uint32_t lref = __float_as_uint(lmin.w), rref = __float_as_uint(rmin.w);
        if (!(lref & CRT_LEAF_BIT) && n < perMesh) { q[n] = lref; lref = CRT_TOP_BIT | (base + n); ++n; }
        if (!(rref & CRT_LEAF_BIT) && n < perMesh) { q[n] = rref; rref = CRT_TOP_BIT | (base + n); ++n; }
        lmin.w = __uint_as_float(lref); rmin.w = __uint_as_float(rref);
        float4* o = topPairs + (size_t)(base + h) * 4;
        o[0] = lmin; o[1] = lmax; o[2] = rmin; o[3] = rmax;
    }
}

// The stack of one wave of a four-wave workgroup: CrtStackT's layout (slot s of lane l at lds[s * 64 + l], overflow block per WAVE, indexed by
// the wave's virtual block number) plus the workgroup's LDS copy of the tree tops.
struct CrtStackTop {
    static constexpr bool kTop = true;
    static constexpr int kLds = CRT_TOP_LDS_SLOTS;
    crt_lds_u32_ptr lds;          // this lane's slot 0 in its wave's block
    uint32_t* ovf;                // base of the launch's overflow area
    crt_lds_f32x4_ptr top;        // the workgroup's copy of the table
    uint32_t vblock;              // the wave's virtual block number (wave-uniform)
    __device__ __forceinline__ uint32_t* overflow_slot(int k) const
    {
        uint32_t lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));
        return ovf + ((size_t)vblock * CRT_OVF_SLOTS_MAX + (size_t)k) * CRT_BLOCK + lane;
    }
    __device__ __forceinline__ void write(int slot, uint32_t v) const
    {
        const int s = slot & (CRT_STACK_DEPTH - 1);
        if (__ballot(s >= kLds) == 0) { lds[s * 64] = v; return; }
        if (s < kLds) lds[s * 64] = v;
        else *overflow_slot(s - kLds) = v;
    }
    __device__ __forceinline__ uint32_t read(int slot) const
    {
        const int s = slot & (CRT_STACK_DEPTH - 1);
        if (__ballot(s >= kLds) == 0) return lds[s * 64];
        if (s < kLds) return lds[s * 64];
        return *overflow_slot(s - kLds);
    }
    __device__ __forceinline__ crt_lds_f32x4_ptr top_record(uint32_t k) const { return top + k * 4; }
};

// RayGen + Trace, both bounces, as crt_trace_kernel<COUNT> (plain instantiation: no shadow rays / refraction / instance tree / stamps / feedback
// lists), four tiles per workgroup. Workgroup g runs on XCD g % 8 and holds the tiles at positions (g / 8) * 4 + w of that XCD's plain list, so the
// tile-to-XCD mapping is the default kernel's (lane_pixel with the wave's virtual block number).
template <bool COUNT>
__global__ __launch_bounds__(CRT_BLOCK * CRT_TOP_WAVES) void crt_trace_ldstop_kernel(CrtDevScene S, CrtFrame F, float4* __restrict__ out, unsigned long long* __restrict__ counters)
{
    __shared__ crt_f32x4 s_top[CRT_TOP_PAIRS * 4];
    __shared__ uint32_t s_stack[CRT_TOP_WAVES * CRT_TOP_LDS_SLOTS * CRT_BLOCK];
    // stage the table: 1008 float4, four per thread (every wave of the workgroup takes part, also one whose tile lies outside the frame)
    {
        const crt_f32x4* __restrict__ src = reinterpret_cast<const crt_f32x4*>(S.topPairs);
        for (int k = (int)threadIdx.x; k < CRT_TOP_PAIRS * 4; k += CRT_BLOCK * CRT_TOP_WAVES) s_top[k] = src[k];
    }
    __syncthreads();
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int vb = (int)(((blockIdx.x >> 3) * CRT_TOP_WAVES + (uint32_t)wave) * 8u + (blockIdx.x & 7u));
    const CrtStackTop stack = { (crt_lds_u32_ptr)s_stack + wave * (CRT_TOP_LDS_SLOTS * CRT_BLOCK) + lane, S.stackOverflow, (crt_lds_f32x4_ptr)s_top, (uint32_t)vb };
    LaneCounters lc; zero_counters(lc);
    int px, py;
    const bool active = lane_pixel(F, px, py, nullptr, nullptr, vb, lane);
    if (active) {
        PathState ps;
        ps.o = mk3(F.camPos[0], F.camPos[1], F.camPos[2]);
        ps.d = raygen_dir(F, px, py);
        ps.result = mk3(0.0f, 0.0f, 0.0f);
        ps.energy = 1.0f;
        for (int bounce = 0; bounce < 2; ++bounce) {
            if (COUNT) { lc.rays++; if (bounce == 0) lc.primary++; else lc.secondary++; }
            Closest c = closest_hit<COUNT, false, false, false>(S, ps.o, ps.d, stack, lc);
            const int cont = shade_bounce(S, c, ps, bounce, F.lightY, F.lightZ);
            if (COUNT) { if (cont) lc.hits++; else lc.misses++; }
            if (!cont) break;
        }
        // the per-pixel stages that follow Trace upstream, on the value in registers (as crt_trace_kernel's epilogue)
        v3 rgb = ps.result;
        if (F.epilogue & CRT_EPILOGUE_QUANTIZE) rgb = mk3(quantize1(rgb.x), quantize1(rgb.y), quantize1(rgb.z));
        if (F.epilogue & CRT_EPILOGUE_POST) {
            rgb = post_pixel(rgb, px, py, F.width, F.height);
            if (F.epilogue & CRT_EPILOGUE_QUANTIZE) rgb = mk3(quantize1(rgb.x), quantize1(rgb.y), quantize1(rgb.z));
        }
        out[(size_t)py * (size_t)F.width + (size_t)px] = make_float4(rgb.x, rgb.y, rgb.z, 1.0f);
        if (F.packOut) F.packOut[(size_t)py * (size_t)F.width + (size_t)px] = unorm8(rgb.x) | (unorm8(rgb.y) << 8) | (unorm8(rgb.z) << 16) | 0xFF000000u;
    }
    if (COUNT) flush_counters(lc, counters);
}
