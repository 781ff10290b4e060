// crt_device.h -- device-side data layout and the per-ray traversal/shading code (gfx950).
//
// Reference semantics restated for CDNA4 (citations relative to upstream CLRayTracer/...):
//   kernels/kernel_main.cl:84-160   IntersectTriangle / IntersectAABB / IntersectBVH
//   kernels/kernel_main.cl:164-275  Trace          kernels/kernel_main.cl:277-287  RayGen
//   kernels/MathAndSTL.cl:100-119, 243-266  MatMul / Mat3Mul / reflect / colour / sampling
// Arithmetic contract (must match oracle/crt_oracle.h "Pinned builtin semantics"): fp32, no FMA
// contraction (-ffp-contract=off), IEEE divide/sqrt, per-ray operation order exactly upstream's.
//
// HBM layout (built by the kernels of crt_relayout.h from the reference-layout uploads):
//   pairs   : one 64-byte, 64-byte-aligned record per sibling pair {L.min,L.ref | L.max,- |
//             R.min,R.ref | R.max,-}; pair index = leftFirst >> 1 (siblings are adjacent upstream,
//             BVH.cpp:203-204). One inner-node visit = one aligned 64-byte fetch per lane.
//   ref     : 32-bit child descriptor. Inner: pair index. Leaf: bit31 | count<<24 | firstTri
//             (count 1..127; 0 = look up CrtDevScene::bigLeaf[firstTri]). A popped/descended node
//             needs no re-fetch to learn whether it is a leaf (upstream re-reads the 32-byte node).
//   triHot  : 36 bytes per triangle {v0, v1-v0, v2-v0} (edges precomputed with the same fp32
//             subtraction upstream performs per test -> bit-identical), read only by leaf tests.
//   triCold : the 32-byte attribute tail of the reference Tri (uv halfs, material, normal halfs),
//             read once per shaded hit.
//   texels  : RGBA8 (one dword per texel) expanded from the packed RGB8 pool.
//   instBounds : per instance, a conservative world-space bounding sphere of the mesh's root box
//             (computed in double on the host at upload time). Upstream has no TLAS: every ray
//             transforms into every instance and tests the root's two child boxes
//             (kernel_main.cl:198-217). A ray that misses the enlarged sphere cannot pass either
//             slab test, so the instance is skipped with identical results -- and the counters
//             record the one pop / one inner visit upstream would have spent on it.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/crt_types.h"

#define CRT_LEAF_BIT 0x80000000u
// CRT_KERNEL=ldstop only: an INNER reference with this bit names record (ref & 0xFFFF) of the tree-top table (crt_ldstop.h) instead of
// a pair index; such references exist only inside that table and in CrtDevInstance::r2.w, which no other kernel form reads
#define CRT_TOP_BIT 0x40000000u
#define CRT_BLOCK 64      // one wave64 per workgroup: a finished wave frees its LDS and wave slot at once
#ifndef CRT_WAVES_PER_SIMD
#define CRT_WAVES_PER_SIMD 8   // 32 waves per CU: 64 VGPRs (the trace kernel fits them without scratch) and 5 KiB of LDS each = the CU's 160 KiB
#endif
#define CRT_WAVES_PER_SIMD_COUNT 6   // instrumented instantiations (work counters, stamps, queries) carry 14 more registers: 80 VGPRs, no heavy spills
// Traversal stack: upstream's `int nodesToVisit[32]` (kernel_main.cl:126), slot indices wrapping modulo 32 where
// upstream's array would overflow. Slots 0..CRT_LDS_SLOTS-1 live in LDS -- slot s of lane l at lds[s * 64 + l], so a
// wave's ds_read/ds_write_b32 is conflict-free; 20 slots = 5 KiB per wave, which is what lets 32 waves share a CU's
// 160 KiB. Slots 20..31 (no scene here has passed depth 15; hand-built deep trees in tests/test_gpu_deep_stack.py do)
// live in a global overflow block owned by the WORKGROUP (blockIdx.x): no two waves of a launch share an entry, every
// frame slot / query has its own area, and an entry is always written (push) before it is read (pop) within one
// traversal, so the area needs no initialisation and costs nothing until a stack passes 20 entries.
// (Round 1 indexed the area by the hardware wave slot from HW_ID; that is only unique while no wave is context-saved
// and restored elsewhere. Round 1 also had two flavours, 32 LDS slots at 5 waves/SIMD and 25 at 6: the SLP vectoriser's
// packed-math splats cost 20+ VGPRs and spilled; with -fno-slp-vectorize one flavour at 8 waves/SIMD wins everywhere:
// 5.9 -> 6.6 (no SLP, 6 waves) -> 7.0 (7 waves) -> 7.6 Gray/s (8 waves) on multi-1M with frames in flight.)
// (Explicit LDS pointer type: through a generic pointer the compiler read the stack with flat_load.)
typedef uint32_t __attribute__((address_space(3))) * crt_lds_u32_ptr;
// loads through a constant-address-space pointer become scalar loads (s_load_*, scalar cache) when the address is wave-uniform
typedef float crt_f32x4 __attribute__((ext_vector_type(4)));
typedef const crt_f32x4 __attribute__((address_space(4)))* crt_const_f32x4_ptr;
typedef const crt_f32x4 __attribute__((address_space(3)))* crt_lds_f32x4_ptr;
#ifndef CRT_LDS_SLOTS
#define CRT_LDS_SLOTS 20
#endif
// PARK: the last PARK of the wave's LDS slots do not hold stack entries but per-lane values "parked" across the traversals
// (values that are live through them but not used in them, which would otherwise be spilled to scratch by the instantiations
// that carry more state: the instance tree's candidate list, the shadow ray's n.l). The stack then has CRT_LDS_SLOTS - PARK
// slots in LDS and the rest in the overflow block; the block is sized for the largest PARK.
#define CRT_MAX_PARK 5
#define CRT_OVF_SLOTS_MAX (CRT_STACK_DEPTH - (CRT_LDS_SLOTS - CRT_MAX_PARK))
#define CRT_OVF_WORDS_PER_BLOCK ((size_t)CRT_OVF_SLOTS_MAX * CRT_BLOCK)
template <int PARK>
struct CrtStackT {
    static constexpr bool kTop = false;      // no LDS-resident tree tops (crt_ldstop.h's stack type says true)
    static constexpr int kLds = CRT_LDS_SLOTS - PARK;
    static_assert(PARK >= 0 && PARK <= CRT_MAX_PARK && kLds >= 1 && kLds <= CRT_STACK_DEPTH, "LDS slots");
    crt_lds_u32_ptr lds;     // this lane's slot 0
    uint32_t* ovf;           // base of the launch's overflow area (wave-uniform)
    __device__ __forceinline__ uint32_t* overflow_slot(int k) const
    {
        uint32_t lane = threadIdx.x & 63;
        asm volatile("" : "+v"(lane));      // keep the address arithmetic inside this (rarely taken) branch: hoisted out of the
                                            // traversal loop it costs two VGPRs for the whole kernel
        return ovf + ((size_t)blockIdx.x * CRT_OVF_SLOTS_MAX + (size_t)k) * CRT_BLOCK + lane;
    }
    // The overflow test is made for the WAVE first (one compare + one scalar branch): no scene here passes kLds entries in the
    // common case, and the per-lane form costs every push and pop an exec-mask save / restore pair (the CU's scalar unit is
    // ~66 % busy with such bookkeeping).
    __device__ __forceinline__ void write(int slot, uint32_t v) const
    {
        const int s = slot & (CRT_STACK_DEPTH - 1);
        if (kLds >= CRT_STACK_DEPTH || __ballot(s >= kLds) == 0) { lds[s * 64] = v; return; }
        if (kLds >= CRT_STACK_DEPTH || s < kLds) lds[s * 64] = v;
        else *overflow_slot(s - kLds) = v;
    }
    __device__ __forceinline__ uint32_t read(int slot) const
    {
        const int s = slot & (CRT_STACK_DEPTH - 1);
        if (kLds >= CRT_STACK_DEPTH || __ballot(s >= kLds) == 0) return lds[s * 64];
        if (kLds >= CRT_STACK_DEPTH || s < kLds) return lds[s * 64];
        return *overflow_slot(s - kLds);
    }
    // parked value k (0 <= k < PARK) of this lane
    __device__ __forceinline__ crt_lds_f32x4_ptr top_record(uint32_t) const { return nullptr; }   // kTop stacks only (dead code here)
    __device__ __forceinline__ void park(int k, uint32_t v) const { lds[(kLds + k) * 64] = v; }
    __device__ __forceinline__ uint32_t parked(int k) const { return lds[(kLds + k) * 64]; }
};
typedef CrtStackT<0> CrtStack;
#ifndef CRT_SPLIT_BETA
#define CRT_SPLIT_BETA 1.2f
#endif
#ifndef CRT_SPLIT_BETA_ASYNC
#define CRT_SPLIT_BETA_ASYNC 2.0f
#endif
#define CRT_MAX_SPLIT 96   // per XCD and frame: heaviest tiles traced as four 4x4-pixel waves instead of one 8x8 wave
#ifndef CRT_MAX_SPLIT_PIPELINED
#define CRT_MAX_SPLIT_PIPELINED 8   // with frames in flight the tail is hidden by the next frame: split only the very heaviest
#endif
#define CRT_TILE 8        // 8x8 pixels per wave, Morton order inside

// Node of the instance tree (TLAS): a world-space bounding sphere and two children; child bit 31 set = leaf, low bits =
// instance index. Built on the host whenever instances are uploaded (crt_instances.h, rebuild_instance_master).
struct CrtTlasNode { float4 sphere; uint32_t left, right, pad0, pad1; };
#define CRT_TLAS_LEAF 0x80000000u
#define CRT_TLAS_MIN_INSTANCES 64      // with fewer instances the linear sphere loop of candidate_mask is cheaper
#define CRT_TLAS_LIST 8                // candidates a lane can hold; a wave with a lane that needs more uses the chunked loop

struct CrtDevScene {
    const float4* __restrict__ pairs;
    const float* __restrict__ triHot;
    const uint4* __restrict__ triCold;
    const uint32_t* __restrict__ bigLeaf;
    const uint32_t* __restrict__ rootRefs;
    uint32_t* stackOverflow;             // this launch's overflow area: CRT_OVF_WORDS_PER_BLOCK words per workgroup, see CrtStack
    const CrtMeshInstance* __restrict__ instances;
    const struct CrtDevInstance* __restrict__ devInstances;
    const float4* __restrict__ instBounds;   // world-space bounding sphere per instance (xyz, r); r < 0: never cull
    const CrtMaterial* __restrict__ materials;
    const CrtTexture* __restrict__ textures;
    const uint32_t* __restrict__ texels;
    int numTexels;
    uint32_t numInstances;
    const CrtTlasNode* __restrict__ tlas;    // bounding-sphere tree over the cullable instances (host-built), or null
    uint32_t tlasNodes;                      // 0: no tree (few instances, or none cullable)
    const uint32_t* __restrict__ alwaysList; // instances that are never culled (single-leaf meshes, unbounded), ascending
    uint32_t numAlways;
    const float4* __restrict__ topPairs;     // CRT_KERNEL=ldstop: the top levels of every mesh's tree, CRT_TOP_PAIRS records (crt_ldstop.h); else unused
};

struct CrtFrame {
    float invView[16];
    float invProj[16];
    float camPos[3];
    float lightY, lightZ;     // (float)sin((double)sunAngle), (float)cos((double)sunAngle), computed on the host
    int width, height;
    int tilesX;               // ceil(width / 16)
    int ownedTileRows;        // 16-row tile rows this rank renders
    int gridBlocks;           // ceil(ownedTileRows / 8) * 8 * tilesX
    int tileRowsPerBand;      // bandRows / 8
    int rank, nRanks;
    int slotsPerXcd;          // ceil(ownedTileRows / 8) * tilesX: tiles in each XCD's list
    const uint32_t* order;    // per-XCD dispatch list, heaviest first: slot | quadrant << 28 | split << 31; see crt_order_kernel
    const uint32_t* listLen;  // entries in each XCD's list (tiles + 3 extra entries per split tile)
    int listCap;              // capacity of one XCD's list = slotsPerXcd + 3 * CRT_MAX_SPLIT
    uint32_t* cost;           // per tile: shader cycles the wave spent on it this frame (feeds the next frame's order)
    uint32_t epilogue;        // crt_trace_kernel: per-pixel stages applied before the pixel is stored (CRT_EPILOGUE_*); 0 = the plain HDR value
    uint32_t* packOut;        // with CRT_EPILOGUE_QUANTIZE: also store the pixel's RGBA8 bytes here (the frame a read-back delivers); or null
};

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 add3(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 sub3(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 mul3(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 scale3(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 neg3(v3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ v3 cross3(v3 a, v3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ v3 normalize3(v3 v)
{
    float inv = 1.0f / sqrtf(dot3(v, v));
    return scale3(v, inv);
}
__device__ __forceinline__ v3 reflect3(v3 v, v3 n)
{
    float d = dot3(n, v);
    return sub3(v, scale3(scale3(n, d), 2.0f));
}
// (int) pinned: truncation, NaN -> 0, saturating
__device__ __forceinline__ int f2i(float x)
{
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}
__device__ __forceinline__ float h2f(uint32_t bits16) { return __half2float(__ushort_as_half((unsigned short)bits16)); }

struct CrtDevInstance;
struct Triout { float t, u, v; uint32_t tri; };

struct LaneCounters {
    uint32_t rays, primary, secondary, hits, misses;
    uint32_t traversals, pops, innerVisits, triTests, capHits, stackOverflows, maxStack;
    uint32_t shadowRays, shadowHits;      // CRT_RENDER_SHADOWS only
    uint32_t culled;                      // instance visits the sphere cull answered (counted in traversals/pops/innerVisits as upstream's one pop + one root visit)
};

// kernel_main.cl:108-117
__device__ __forceinline__ float intersect_aabb(v3 o, v3 inv, float4 bmin, float4 bmax, float minSoFar)
{
    float tminx = (bmin.x - o.x) * inv.x, tminy = (bmin.y - o.y) * inv.y, tminz = (bmin.z - o.z) * inv.z;
    float tmaxx = (bmax.x - o.x) * inv.x, tmaxy = (bmax.y - o.y) * inv.y, tmaxz = (bmax.z - o.z) * inv.z;
    float tnear = fmaxf(fmaxf(fminf(tminx, tmaxx), fminf(tminy, tmaxy)), fminf(tminz, tmaxz));
    float tfar  = fminf(fminf(fmaxf(tminx, tmaxx), fmaxf(tminy, tmaxy)), fmaxf(tminz, tmaxz));
    return (tnear < tfar && tnear > 0.0f && tnear < minSoFar) ? tnear : 1e30f;
}

// kernel_main.cl:84-106; hot = {v0, edge1, edge2}
__device__ __forceinline__ int intersect_triangle(v3 o, v3 d, v3 x, v3 edge1, v3 edge2, Triout& out, uint32_t i)
{
    const v3 h = cross3(d, edge2);
    const float a = dot3(edge1, h);
    const float f = 1.0f / a;
    const v3 s = sub3(o, x);
    const float u = f * dot3(s, h);
    const v3 q = cross3(s, edge1);
    const float v = f * dot3(d, q);
    const float t = f * dot3(edge2, q);
    int passed = (((int)(t > 0.0f) ^ (int)(t < out.t)) + (int)(u < 0.0f) + (int)(u > 1.0f) + (int)(v < 0.0f) + (int)(u + v > 1.0f)) == 0;
    int notPassed = 1 - passed;
    // arithmetic blend kept as upstream (NaN/inf propagate through the 0-weighted term)
    out.u = u * (float)passed + ((float)notPassed * out.u);
    out.v = v * (float)passed + ((float)notPassed * out.v);
    out.t = t * (float)passed + ((float)notPassed * out.t);
    out.tri = i * (uint32_t)passed + ((uint32_t)notPassed * out.tri);
    return passed;
}

// MathAndSTL.cl:100-102 with a row-major matrix in memory: ((m.x*v.x + m.y*v.y) + m.z*v.z) + m.w*v.w
__device__ __forceinline__ v3 matmul_xyz(const float* __restrict__ m, float vx, float vy, float vz, float vw)
{
    v3 r;
    r.x = ((m[0] * vx + m[4] * vy) + m[8] * vz) + m[12] * vw;
    r.y = ((m[1] * vx + m[5] * vy) + m[9] * vz) + m[13] * vw;
    r.z = ((m[2] * vx + m[6] * vy) + m[10] * vz) + m[14] * vw;
    return r;
}
__device__ __forceinline__ float matmul_w(const float* __restrict__ m, float vx, float vy, float vz, float vw)
{
    return ((m[3] * vx + m[7] * vy) + m[11] * vz) + m[15] * vw;
}
__device__ __forceinline__ v3 mat3mul(const float* __restrict__ m, v3 v)
{
    v3 r;
    r.x = (m[0] * v.x + m[4] * v.y) + m[8] * v.z;
    r.y = (m[1] * v.x + m[5] * v.y) + m[9] * v.z;
    r.z = (m[2] * v.x + m[6] * v.y) + m[10] * v.z;
    return r;
}

struct Closest { float distance; int hitInstance; int anyHit; Triout hit; };


// diagnostic (ITERS builds only): true in exactly one active lane, so summing over lanes counts wave-level loop trips
__device__ __forceinline__ bool first_active_lane() { return (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1; }

// Device-side instance record (64 B, built at upload): the 3 used columns of inverseTransform
// (MatMul(...).xyz and Mat3Mul never read column 3) with the root ref and materialStart in the
// spare lanes. One aligned 64-byte gather per lane when a lane enters an instance.
struct CrtDevInstance { float4 r0, r1, r2, r3; };   // rK = {m[K][0], m[K][1], m[K][2], extra}; r0.w = rootRef, r1.w = materialStart

__device__ __forceinline__ v3 xform_xyz(const CrtDevInstance& m, float vx, float vy, float vz, float vw)
{
    v3 r;   // MathAndSTL.cl:100-102: ((m.x*v.x + m.y*v.y) + m.z*v.z) + m.w*v.w
    r.x = ((m.r0.x * vx + m.r1.x * vy) + m.r2.x * vz) + m.r3.x * vw;
    r.y = ((m.r0.y * vx + m.r1.y * vy) + m.r2.y * vz) + m.r3.y * vw;
    r.z = ((m.r0.z * vx + m.r1.z * vy) + m.r2.z * vz) + m.r3.z * vw;
    return r;
}

__device__ __forceinline__ v3 mat3mul(const CrtDevInstance& m, v3 v)
{
    v3 r;   // MathAndSTL.cl:104-106 on ConvertToMatrix3(inverseTransform)
    r.x = (m.r0.x * v.x + m.r1.x * v.y) + m.r2.x * v.z;
    r.y = (m.r0.y * v.x + m.r1.y * v.y) + m.r2.y * v.z;
    r.z = (m.r0.z * v.x + m.r1.z * v.y) + m.r2.z * v.z;
    return r;
}

// The device record of instance `inst` (per lane). It comes through a SCALAR load (one s_load_dwordx16 through the scalar
// cache) when every lane that asks in this step asks for the same instance -- the common case for a coherent packet, whose
// lanes share their first candidates and mostly hit the same instance -- and through four per-lane vector loads otherwise:
// vector-memory instructions are what bounds the kernel, and unlike the node fetches (where the same test
// cost more than it saved, round 2) instance records are fetched rarely enough for one readfirstlane + compare + ballot
// per step to pay. (The table is written by crt_refresh_instances_kernel in an earlier launch, never by the reading kernel.)
__device__ __forceinline__ CrtDevInstance load_instance(const CrtDevInstance* __restrict__ table, uint32_t inst)
{
    CrtDevInstance I;
    const uint32_t inst0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)inst);
    if (__ballot(inst != inst0) == 0) {
        const crt_const_f32x4_ptr p = (crt_const_f32x4_ptr)(table + inst0);
        const crt_f32x4 a = p[0], b = p[1], c = p[2], e = p[3];
        I.r0 = make_float4(a.x, a.y, a.z, a.w); I.r1 = make_float4(b.x, b.y, b.z, b.w);
        I.r2 = make_float4(c.x, c.y, c.z, c.w); I.r3 = make_float4(e.x, e.y, e.z, e.w);
    } else {
        const CrtDevInstance* ip = table + inst;
        I.r0 = ip->r0; I.r1 = ip->r1; I.r2 = ip->r2; I.r3 = ip->r3;
    }
    return I;
}

// ------------------------------------------------------------------------------------------------
// Per-lane traversal state machine: the instance loop + IntersectBVH of kernel_main.cl:124-160,198-217.
//
// Upstream (and a literal port) runs `for instance { while pop { descend } }` in lock-step: a wave pays, for
// every instance, the slowest lane's traversal, and every lane waits for its neighbours' whole descents. Here
// every lane owns a `Traversal`: its own ascending list of candidate instances and its own position in the tree,
// advanced by three step kinds -- enter(next candidate), inner(one child pair), leaf(its triangles, then pop).
// Per ray the sequence of instances, node visits, triangle tests and the running best t is exactly upstream's,
// so results are bit-identical; only the interleaving between lanes differs.
// The stack lives in LDS/scratch (CRT_STACK_*); slot indices wrap modulo 32 where upstream's array would overflow.
// ------------------------------------------------------------------------------------------------
template <bool COUNT>
struct Traversal {
    v3 mo, md, inv;               // ray in the current instance's object space (direction not renormalised, hazard H6)
    Triout tr;                    // running best of the current instance (kernel_main.cl:200-202)
    int sp, prot, inters;         // stack pointer, pop counter (kernel_main.cl:131), OR of the `passed` flags
    uint32_t ref, curInst;
    bool active;                  // inside an instance

    __device__ __forceinline__ void reset()
    {
        mo = mk3(0.f, 0.f, 0.f); md = mo; inv = mo;
        tr.t = 0.f; tr.u = 0.f; tr.v = 0.f; tr.tri = 0;
        sp = 0; prot = 0; inters = 0; ref = 0; curInst = 0; active = false;
    }
    __device__ __forceinline__ bool at_inner() const { return active && !(ref & CRT_LEAF_BIT); }
    __device__ __forceinline__ bool at_leaf() const { return active && (ref & CRT_LEAF_BIT); }

    // ends the current instance: keep the hit if any triangle passed (kernel_main.cl:210-216)
    __device__ __forceinline__ void finish(Closest& c)
    {
        if (inters) { c.hitInstance = (int)curInst; c.hit = tr; c.distance = tr.t; c.anyHit = 1; }
        active = false;
    }
    // `while (currentNodeIndex > 0 && protection++ < 250) node = stack[--currentNodeIndex]` (kernel_main.cl:131-133)
    template <class STK>
    __device__ __forceinline__ void pop_next(const STK& stack, Closest& c, LaneCounters& lc)
    {
        // one level of branching: `protection++` happens exactly when the stack is not empty
        const bool nonEmpty = sp > 0, canPop = nonEmpty && prot < CRT_MAX_POPS;
        prot += nonEmpty ? 1 : 0;
        if (COUNT) { if (nonEmpty && !canPop) lc.capHits++; if (canPop) lc.pops++; }
        if (canPop) { --sp; ref = stack.read(sp); }
        else {
            const bool keep = inters != 0;                       // finish(): selects instead of a nested branch
            c.hitInstance = keep ? (int)curInst : c.hitInstance; c.distance = keep ? tr.t : c.distance; c.anyHit = keep ? 1 : c.anyHit;
            c.hit.t = keep ? tr.t : c.hit.t; c.hit.u = keep ? tr.u : c.hit.u; c.hit.v = keep ? tr.v : c.hit.v; c.hit.tri = keep ? tr.tri : c.hit.tri;
            active = false;
        }
    }
    // kernel_main.cl:200-210: transform the ray into instance `inst` and start at its root (record fetch: load_instance above)
    // TOP (crt_ldstop.h): start at the root's record in the tree-top table (CrtDevInstance::r2.w) instead of the global pair
    template <bool TOP = false>
    __device__ __forceinline__ void enter(const CrtDevScene& S, uint32_t inst, v3 o, v3 d, float bestSoFar, LaneCounters& lc)
    {
        curInst = inst;
        const CrtDevInstance I = load_instance(S.devInstances, inst);
        mo = xform_xyz(I, o.x, o.y, o.z, 1.0f);
        md = xform_xyz(I, d.x, d.y, d.z, 0.0f);
        inv = mk3(1.0f / md.x, 1.0f / md.y, 1.0f / md.z);       // native_recip pinned to IEEE
        tr.t = bestSoFar; tr.tri = 0; tr.u = 0.0f; tr.v = 0.0f;
        ref = __float_as_uint(TOP ? I.r2.w : I.r0.w);            // the root is popped at once: sp 1 -> 0, protection 0 -> 1
        sp = 0; prot = 1; inters = 0; active = true;
        if (COUNT) { lc.traversals++; lc.pops++; }
    }
    // kernel_main.cl:142-157: fetch the child pair, two slab tests, near child first, far child pushed
    template <class STK>
    __device__ __forceinline__ void inner(const CrtDevScene& S, const STK& stack, Closest& c, LaneCounters& lc)
    {
        // every lane of this step on the same node (the top of a tree under a coherent packet): one scalar load instead of four
        // vector loads, the boxes as scalar operands. (Round 2 measured the same idea as a wash -- the uniformity test costs every
        // step -- but with the instance bounds and records on the scalar path the node fetches are 3/4 of the vector-memory
        // instructions that bound the kernel: sponza-sibenik +6 %, multi-1M-dense +2 %, multi-1M and nanosuit-demo +-0, synchronous
        // frames +0...5 %. The same for leaves -- a uniform triangle through scalar loads -- lost 18 %: the 9 scalar operands push the
        // kernel into scratch, and small triangles are never shared by a whole packet.)
        // (Round 5: the two slab tests are written out in each branch, so that the scalar branch uses the record's SGPRs as operands
        // directly instead of copying 14 of them into VGPRs first.)
        float dist1, dist2;
        uint32_t nearRef, farRef;
        const uint32_t ref0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ref);
        if (__ballot(ref != ref0) == 0) {
            // (STK::kTop: a wave-uniform step on a tree-top record reads the table's GLOBAL copy through the scalar cache, as before)
            const crt_const_f32x4_ptr q = (STK::kTop && (ref0 & CRT_TOP_BIT)) ? (crt_const_f32x4_ptr)(S.topPairs + (size_t)(ref0 & 0xFFFFu) * 4)
                                                                             : (crt_const_f32x4_ptr)(S.pairs + (size_t)ref0 * 4);
            const crt_f32x4 a = q[0], b = q[1], c4 = q[2], e = q[3];
            dist1 = intersect_aabb(mo, inv, make_float4(a.x, a.y, a.z, a.w), make_float4(b.x, b.y, b.z, b.w), tr.t);
            dist2 = intersect_aabb(mo, inv, make_float4(c4.x, c4.y, c4.z, c4.w), make_float4(e.x, e.y, e.z, e.w), tr.t);
            nearRef = __float_as_uint(a.w); farRef = __float_as_uint(c4.w);
#ifndef CRT_TOP_NO_LDS_PATH
#define CRT_TOP_NO_LDS_PATH 0                // A/B builds: 1 = the four-wave form without its LDS reads (what the workgroup shape alone costs)
#endif
        } else if (STK::kTop && !CRT_TOP_NO_LDS_PATH && __ballot(!(ref & CRT_TOP_BIT)) == 0) {
            // crt_ldstop.h: EVERY lane of this step wants a record of the LDS-resident tree tops, and they differ: four ds_read_b128 per
            // lane instead of four vector-memory instructions for the wave (north_star's "hot BVH tiles staged in LDS")
            const auto q = stack.top_record(ref & 0xFFFFu);
            const crt_f32x4 a = q[0], b = q[1], c4 = q[2], e = q[3];
            dist1 = intersect_aabb(mo, inv, make_float4(a.x, a.y, a.z, a.w), make_float4(b.x, b.y, b.z, b.w), tr.t);
            dist2 = intersect_aabb(mo, inv, make_float4(c4.x, c4.y, c4.z, c4.w), make_float4(e.x, e.y, e.z, e.w), tr.t);
            nearRef = __float_as_uint(a.w); farRef = __float_as_uint(c4.w);
        } else {
            // one aligned 64-byte record (a step that mixes tree-top and other records takes the top ones from the table's global copy:
            // the four vector loads are issued for the wave either way)
            const float4* p = (STK::kTop && (ref & CRT_TOP_BIT)) ? S.topPairs + (size_t)(ref & 0xFFFFu) * 4 : S.pairs + (size_t)ref * 4;
            const float4 lmin = p[0], lmax = p[1], rmin = p[2], rmax = p[3];
            dist1 = intersect_aabb(mo, inv, lmin, lmax, tr.t);
            dist2 = intersect_aabb(mo, inv, rmin, rmax, tr.t);
            nearRef = __float_as_uint(lmin.w); farRef = __float_as_uint(rmin.w);
        }
        if (COUNT) lc.innerVisits++;
        {   // kernel_main.cl:148-151 as four selects on one compare (the branchy form costs an exec-mask save / restore)
            const bool sw = dist1 > dist2;
            const float d1 = sw ? dist2 : dist1, d2 = sw ? dist1 : dist2;
            const uint32_t r1 = sw ? farRef : nearRef, r2 = sw ? nearRef : farRef;
            dist1 = d1; dist2 = d2; nearRef = r1; farRef = r2;
        }
        if (dist2 != 1e30f) {                                   // both children hit (dist1 <= dist2): push the far one
            if (COUNT) { if (sp >= CRT_STACK_DEPTH) lc.stackOverflows++; }
            stack.write(sp, farRef);
            sp++;
            if (COUNT) { if ((uint32_t)sp > lc.maxStack) lc.maxStack = (uint32_t)sp; }
        }
        if (dist1 == 1e30f) pop_next(stack, c, lc);
        else ref = nearRef;
    }
    // kernel_main.cl:135-140: every triangle of the leaf, then the next pop.
    // ANYHIT (shadow rays): the traversal ends at the first triangle that passes.
    template <bool ANYHIT, class STK>
    __device__ __forceinline__ void leaf(const CrtDevScene& S, const STK& stack, Closest& c, LaneCounters& lc)
    {
        const uint32_t first = ref & 0x00FFFFFFu;
        uint32_t n = (ref >> 24) & 0x7Fu;
        if (n == 0) n = S.bigLeaf[first];
        for (uint32_t i = first, end = first + n; i < end; ++i) {
            if (COUNT) lc.triTests++;
            const float* __restrict__ hot = S.triHot + (size_t)i * 9;
            inters |= intersect_triangle(mo, md, mk3(hot[0], hot[1], hot[2]), mk3(hot[3], hot[4], hot[5]), mk3(hot[6], hot[7], hot[8]), tr, i);
            if (ANYHIT) { if (inters) break; }
        }
        if (ANYHIT && inters) finish(c);
        else pop_next(stack, c, lc);
    }
};

// ------------------------------------------------------------------------------------------------
// The conservative instance cull, and why its slack is enough.
//
// Claim: if the predicate below rejects instance i for a ray (o, d), the ray fails BOTH child slab tests of that instance's
// root as the traversal would compute them in fp32 (kernel_main.cl:203-207 transform + kernel_main.cl:108-117), so skipping the
// instance changes no hit record; the counters add the one pop + one inner visit upstream spends on it.
//
// Notation. u = 2^-24, g3 = 3u/(1-3u), g4 = 4u/(1-4u). M3 / T: the 3x3 part / the translation row of the instance's fp32
// inverseTransform, F3 = M3^-1 (double, host). kappa = |M3|_F |F3|_F (Frobenius condition number, >= 3; 3 for rigid +
// uniform scale), tau = |T|_2 |F3|_F. (c, r): exact centre / radius of the sphere around the world image of the union of the
// root's two child boxes; the table holds (fl(c), w) with w >= r (1 + 1e-4) + |c - fl(c)|, so its sphere contains the exact
// one. x = |c - o|, O = |o|.
//
// (1) What an fp32 pass means in world space. xform_xyz sums four products left to right:
//     |mo - o M|_2 <= g4 (|o| |M3|_F + |T|),  |md - d M3|_2 <= g3 |d| |M3|_F.
// In a slab test every t is (b - mo_k) * (1 / md_k): three roundings, t_computed = t_exact (1 + theta), |theta| <= g3. If the
// test passes (tnear < tfar, tnear > 0), the point p = mo + tnear md of the COMPUTED ray lies in the box grown along each
// axis by g3 max(|bmin_k - mo_k|, |bmax_k - mo_k|) (from lo_k (1 + theta) <= tnear < hi_k (1 + theta')); a zero / overflowing
// 1 / md_k or a 0 * inf NaN can only turn a pass into a miss or leave p inside the closed slab. Carried back to world space
// (a factor |F3|; t is the same parameter on both sides because directions are not renormalised, H6; |d| t <= x + r there), the
// exact world ray o + t d, t > 0, comes within
//     Delta(O, x) = g4 (kappa O + tau) + (1 + sqrt 3) g3 kappa (x + w)
// of the exact world image of the box, hence of the sphere (c, r): a pass implies dist(ray, c) <= r + Delta.
// (2) What a rejection means. With fp32 inputs (fl(c), w, o, d), oc = fl(c) - o is exact to u per component, oc2, dd and b =
// oc.d carry <= 5u, 3u (relative) and 4u |oc| |d| (absolute); `oc2 dd - b b > r2 dd` computed implies, in exact arithmetic,
//     dist(line, fl(c))^2 > 1.0201 (1 - 12u) w^2 + (4e-6 (1 - 12u) - 19u) x^2  >=  1.02 w^2 + 2.8e-6 x^2
// -- the 4e-6 |oc|^2 term is what outweighs the cancellation in oc2 dd - b b -- and the second clause (centre behind the origin,
// origin outside the grown sphere) gives the same bound for every t > 0 (b < 0 computed allows b <= 4u |oc| |d| exactly, which
// moves the closest approach by a factor 1 - 16u^2). Any NaN makes both comparisons false: not culled.
// (3) The slack dominates when sqrt(1.02 w^2 + 2.8e-6 x^2) >= w + Delta(O, x) for all x >= 0. The left side is convex in x, the
// right side affine; with c1 = (1 + sqrt 3) g3 kappa (4.9e-7 kappa) the minimum of the difference is at
// 2.8e-6 x = c1 sqrt(1.02 w^2 + 2.8e-6 x^2) and equals  w (sqrt(1.02 (1 - c1^2 / 2.8e-6)) - 1 - c1) - g4 (kappa O + tau), so
//     the cull is exact for ray origins with  |o| <= O_i = (w (sqrt(1.02 (1 - c1^2 / 2.8e-6)) - 1 - c1) - g4 tau) / (g4 kappa).
// Rigid + uniform scale, instance centre p: kappa = 3, tau = sqrt 3 |p|: O_i ~ 13,800 w - 0.58 |p| -- a 1-unit instance may be
// viewed from 13,800 units, a 1000-unit one from 1.4e7; kappa above ~470 (c1 > 2.3e-4) leaves no range at all.
// (4) Outside the range the host turns the cull off (crt_instances.h rebuild_instance_master / crt_frame.h crt1_render): an instance whose O_i is
// below the reach of bounce-ray origins (object-space hit points used as world origins, H6) is stored with r = -1 = never
// culled; a frame whose camera, or a query whose farthest origin, lies beyond the smallest O_i of the remaining instances runs
// with an all-never table (and the linear candidate loop). tests/test_gpu_cull_bound.py: instance scales 1e-3 ... 1e3,
// condition numbers to 300, origins out to 1e6 and to 0.95 O_i with grazing rays, hit records against the oracle.
// (5) Nodes of the instance tree (sphere_culls on CrtTlasNode::sphere, radius R around >= 2 instance spheres, so R >= sqrt 3 w_i
// and |C - c_i| + w_i <= R): a rejected node has dist(ray, c_i) >= dist(ray, C) - (R - w_i) > w_i + sqrt(1.02 R^2 + 2.8e-6 X^2) - R,
// and with x_i <= X + R the same minimisation gives R (0.00995 - 2 c1) - g4 (kappa O + tau) >= the instance's own margin
// w_i (0.00995 - c1) - ... for every c1 in the admitted range: a subtree is never rejected for a ray one of its instances admits.
// ------------------------------------------------------------------------------------------------
// The ray/sphere rejection of candidate_mask for the instance tree (same arithmetic; any NaN -> not culled).
__device__ __forceinline__ bool sphere_culls(const float4 bs, v3 o, v3 d, float dd)
{
    const v3 oc = mk3(bs.x - o.x, bs.y - o.y, bs.z - o.z);
    const float oc2 = dot3(oc, oc), b = dot3(oc, d);
    const float r2 = bs.w * bs.w * 1.0201f + 4e-6f * oc2;           // 1 % on the radius + slack growing with distance
    return (bs.w >= 0.0f) & ((oc2 * dd - b * b > r2 * dd) | ((b < 0.0f) & (oc2 > r2)));
}

// Conservative candidate mask for instances [base, base + cnt): bit k is cleared only when the ray provably misses
// instance base+k's bounding sphere (any NaN -> candidate). A culled instance costs upstream exactly one pop and one
// inner visit and changes nothing, which is what the counters record for it. Wave-uniform loop, scalar loads.
// The bounding spheres are read through a constant-address-space pointer: the index is wave-uniform, and only for that
// address space does the compiler turn a uniform load into a SCALAR load (s_load_dwordx4, scalar cache). Through the plain
// pointer -- a member of a by-value kernel argument carries no noalias/readonly information -- rounds 1-2 issued one VECTOR
// load per instance, bounce and wave here: 12 % of the trace kernel's vector-memory instructions, on the pipeline that
// bounds it. The table is written by the host before the launch and never by a kernel.
template <bool COUNT, bool DEFER_COUNT = false>
__device__ __forceinline__ unsigned long long candidate_mask(const CrtDevScene& S, v3 o, v3 d, uint32_t base, uint32_t cnt, LaneCounters& lc)
{
    const float dd = dot3(d, d);
    unsigned long long cand = 0;
    const crt_const_f32x4_ptr bounds = (crt_const_f32x4_ptr)S.instBounds;
    for (uint32_t k = 0; k < cnt; ++k) {
        const crt_f32x4 bs = bounds[base + k];
        const v3 oc = mk3(bs.x - o.x, bs.y - o.y, bs.z - o.z);
        const float oc2 = dot3(oc, oc), b = dot3(oc, d);
        const float r2 = bs.w * bs.w * 1.0201f + 4e-6f * oc2;       // 1 % on the radius + slack growing with distance
        const bool cull = (bs.w >= 0.0f) && ((oc2 * dd - b * b > r2 * dd) || (b < 0.0f && oc2 > r2));
        if (!cull) cand |= 1ull << k;
    }
    if (COUNT && !DEFER_COUNT) { const uint32_t culled = cnt - (uint32_t)__popcll(cand); lc.traversals += culled; lc.pops += culled; lc.innerVisits += culled; lc.culled += culled; }
    return cand;
}

// Candidate instances of one ray from the instance tree: up to CRT_TLAS_LIST indices (16 bits each, unordered, 0xFFFF =
// empty) in four words; returns false when the lane would need more. The traversal stack is idle at this point
// and serves as the tree stack. Scenes with hundreds of instances (upstream allows 401 and loops over all of them for
// every ray, kernel_main.cl:198) spend their time here otherwise: 401 instances, 1920x1080: 1.12 ms -> see DESIGN.md.
// The four words live in the stack's parked LDS slots 0..3 (CrtStackT<PARK >= 4>), not in registers: they are read once per
// instance entry, and as registers they were what the TLAS instantiations spilled to scratch (rounds 1-2: 48 B per lane).
#define CRT_TLAS_PARK (CRT_TLAS_LIST / 2)
template <class STK>
__device__ __forceinline__ bool candidate_list_add(const STK& stack, uint32_t& n, uint32_t idx)
{
    if (n >= (uint32_t)CRT_TLAS_LIST) return false;
    const uint32_t sh = (n & 1u) * 16u, m = ~(0xFFFFu << sh), v = idx << sh;
    const int k = (int)(n >> 1);
    stack.park(k, (stack.parked(k) & m) | v);
    n++;
    return true;
}
// smallest candidate index greater than `after` (after = -1 for the first), or 0xFFFF
template <class STK>
__device__ __forceinline__ uint32_t candidate_list_next(const STK& stack, int after)
{
    uint32_t best = 0xFFFFu;
#pragma unroll
    for (int k = 0; k < CRT_TLAS_PARK; ++k) {
        const uint32_t w = stack.parked(k);
        const uint32_t lo = w & 0xFFFFu, hi = w >> 16;
        if ((int)lo > after && lo < best) best = lo;
        if ((int)hi > after && hi < best) best = hi;
    }
    return best;
}
// fills the parked list; `n` receives the number of candidates
template <class STK>
__device__ __forceinline__ bool tlas_candidates(const CrtDevScene& S, v3 o, v3 d, const STK& stack, uint32_t& n)
{
#pragma unroll
    for (int k = 0; k < CRT_TLAS_PARK; ++k) stack.park(k, 0xFFFFFFFFu);
    n = 0;
    bool ok = true;
    for (uint32_t k = 0; k < S.numAlways; ++k) { const uint32_t i = S.alwaysList[k]; if (i < S.numInstances) ok = candidate_list_add(stack, n, i) && ok; }
    if (S.tlasNodes == 0) return ok;
    const float dd = dot3(d, d);
    uint32_t node = 0; int sp = 0;
    for (;;) {
        const CrtTlasNode nd = S.tlas[node];
        bool descend = false;
        if (!sphere_culls(nd.sphere, o, d, dd)) {
            // children: a leaf child is tested through its own sphere when it is popped as a one-node subtree
            if (nd.left & CRT_TLAS_LEAF) {
                const uint32_t i = nd.left & 0xFFFFu;             // a leaf NODE: left = leaf | instance, right unused
                if (i < S.numInstances) ok = candidate_list_add(stack, n, i) && ok;
            } else { stack.write(sp, nd.right); sp++; node = nd.left; descend = true; }
        }
        if (!descend) { if (sp == 0) break; --sp; node = stack.read(sp); }
    }
    return ok;
}

// Closest hit of one ray per lane over all instances (kernel_main.cl:198-217), driven in flat trips: every lane owns a
// Traversal and a trip of the wave's loop is  enter -> inner -> leaf -> inner  for whichever lanes are at that step (a lane
// may enter an instance, visit a node and test a leaf in the same trip; a lane that popped an inner node out of a leaf or a
// miss goes on at once). One ballot per trip decides when the wave is done. (Rounds 1-2 voted for ONE step kind per trip in
// packets with many working lanes -- three ballots, popcounts and the majority logic; at 8 waves/SIMD that lost 6 % and
// the second inner step gained another 10 %; the voted form was retired in round 3.)
// ANYHIT (shadow rays, CRT_RENDER_SHADOWS): a lane stops at the first triangle that passes -- inside the leaf, and
// for all later instances. `anyHit` is the same boolean the full closest-hit loop would return, because until the
// first passing triangle both visit the same nodes in the same order; only the work (and the counters) shrink.
// ITERS (stamped diagnostic launches): the counters record wave-level trips instead of per-ray work.
template <bool COUNT, bool ITERS, bool ANYHIT, class STK>
__device__ __forceinline__ void trip_steps(const CrtDevScene& S, const STK& stack, Traversal<COUNT>& T, Closest& c, LaneCounters& lc, bool done)
{
    if (!done && T.at_inner()) {
        if (ITERS) { lc.rays++; if (first_active_lane()) lc.innerVisits++; }
        T.inner(S, stack, c, lc);
    }
    if (!done && T.at_leaf()) {
        if (ITERS) {
            if (first_active_lane()) lc.triTests++;
            // wave-level triangle iterations of this leaf step = the largest count among the lanes taking it (3 vector loads each)
            uint32_t n = (T.ref >> 24) & 0x7Fu; if (n == 0) n = S.bigLeaf[T.ref & 0x00FFFFFFu];
            for (int off = 32; off > 0; off >>= 1) { const uint32_t o2 = (uint32_t)__shfl_xor((int)n, off, 64); n = o2 > n ? o2 : n; }
            if (first_active_lane()) lc.misses += n;
        }
        T.template leaf<ANYHIT>(S, stack, c, lc);
    }
    if (!done && T.at_inner()) {
        if (ITERS) { if (first_active_lane()) lc.hits++; }
        T.inner(S, stack, c, lc);      // lanes that just popped an inner node go on at once
    }
}

template <bool COUNT, bool ITERS = false, bool ANYHIT = false, bool TLAS = false, class STK = CrtStack>
__device__ __forceinline__ Closest closest_hit(const CrtDevScene& S, v3 o, v3 d, const STK& stack, LaneCounters& lc)
{
    Closest c;
    c.distance = 99999.0f; c.hitInstance = 0; c.anyHit = 0;
    c.hit.t = 0.0f; c.hit.u = 0.0f; c.hit.v = 0.0f; c.hit.tri = 0;
    Traversal<COUNT> T; T.reset();

    if constexpr (TLAS) {
        // Many instances: every lane collects its (few) candidates from the instance tree and walks them in ascending
        // order -- the same instances, in the same order, as the chunked loop below would. A wave in which some lane
        // has more than CRT_TLAS_LIST candidates takes the chunked loop instead.
        static_assert(!TLAS || STK::kLds <= CRT_LDS_SLOTS - CRT_TLAS_PARK, "the candidate list needs CRT_TLAS_PARK parked slots");
        uint32_t listed = 0;
        const bool fits = tlas_candidates(S, o, d, stack, listed);
        if (__ballot(!fits) == 0) {
            int prev = -1;                                        // last instance entered (ANYHIT + COUNT: culled ones in between)
            if (COUNT && !ANYHIT) { const uint32_t culled = S.numInstances - listed; lc.traversals += culled; lc.pops += culled; lc.innerVisits += culled; lc.culled += culled; }
            bool done = false;
            for (;;) {
                const bool wEnter = !done && !T.active;
                const bool wInner = !done && T.at_inner();
                const bool wLeaf = !done && T.at_leaf();
                if (__ballot(wEnter || wInner || wLeaf) == 0) break;      // (this exact form: `__ballot(!done)` makes the register allocator spill 30 VGPRs)
                if (wEnter) {
                    const uint32_t k = (ANYHIT && c.anyHit) ? 0xFFFFu : candidate_list_next(stack, prev);
                    if (k == 0xFFFFu) {
                        done = true;
                        if (COUNT && ANYHIT && !c.anyHit) { const uint32_t n = S.numInstances - (uint32_t)(prev + 1); lc.traversals += n; lc.pops += n; lc.innerVisits += n; lc.culled += n; }
                    } else {
                        if (COUNT && ANYHIT) { const uint32_t n = k - (uint32_t)(prev + 1); lc.traversals += n; lc.pops += n; lc.innerVisits += n; lc.culled += n; }
                        prev = (int)k;
                        T.template enter<STK::kTop>(S, k, o, d, c.distance, lc);
                    }
                }
                trip_steps<COUNT, false, ANYHIT>(S, stack, T, c, lc, done);
            }
            return c;
        }
    }

    for (uint32_t base = 0; base < S.numInstances; base += 64) {
        const uint32_t cnt = (S.numInstances - base) < 64u ? (S.numInstances - base) : 64u;
        unsigned long long cand = candidate_mask<COUNT, ANYHIT>(S, o, d, base, cnt, lc);
        // ANYHIT + COUNT: a culled instance is only "visited" (one pop, one inner visit upstream) if the ray gets that
        // far, so culled instances are counted when a later candidate is entered or the chunk ends without a hit
        unsigned long long culledLeft = (COUNT && ANYHIT) ? (~cand & (cnt == 64u ? ~0ull : ((1ull << cnt) - 1ull))) : 0ull;
        bool done = ANYHIT && c.anyHit;
        for (;;) {
            const bool wEnter = !done && !T.active;
            const bool wInner = !done && T.at_inner();
            const bool wLeaf = !done && T.at_leaf();
            if (__ballot(wEnter || wInner || wLeaf) == 0) break;
            if (ITERS) { if (first_active_lane()) lc.pops++; }
            if (wEnter) {
                if (ANYHIT && c.anyHit) done = true;           // occluded: later instances are never visited
                else if (cand == 0) {                          // this lane is finished with the chunk
                    done = true;
                    if (COUNT && ANYHIT) { const uint32_t n = (uint32_t)__popcll(culledLeft); lc.traversals += n; lc.pops += n; lc.innerVisits += n; lc.culled += n; culledLeft = 0; }
                } else {
                    if (ITERS) { if (first_active_lane()) lc.traversals++; }
                    const uint32_t k = (uint32_t)__ffsll((long long)cand) - 1u;
                    cand &= cand - 1;
                    if (COUNT && ANYHIT) {
                        const unsigned long long below = culledLeft & ((1ull << k) - 1ull);
                        const uint32_t n = (uint32_t)__popcll(below);
                        lc.traversals += n; lc.pops += n; lc.innerVisits += n; lc.culled += n; culledLeft &= ~below;
                    }
                    T.template enter<STK::kTop>(S, base + k, o, d, c.distance, lc);
                }
            }
            trip_steps<COUNT, ITERS, ANYHIT>(S, stack, T, c, lc, done);
        }
    }
    return c;
}

__device__ __forceinline__ int clamp_texel(int idx, int n) { return idx < 0 ? 0 : (idx >= n ? n - 1 : idx); }

// MathAndSTL.cl:253-258 (hard-wired to textures[2] and pool offset 2 upstream, hazard H9)
// Out of line on purpose: the double-precision atan2/acos expansion needs ~60 VGPRs of temporaries; inlined
// into the persistent kernel it forced the per-lane traversal state into scratch inside the hot loop.
__device__ __noinline__ int sample_skybox(v3 d, int texW, int texH)
{
    const double PI = 3.14159265358979323846;
    float at = (float)(atan2((double)d.x, (double)(-d.z)) / PI);
    float ac = (float)(acos((double)d.y) / PI);
    int theta = f2i((at * 0.5f) * (float)texW);
    int phi = f2i(ac * (float)texH);
    return (int)((uint32_t)phi * (uint32_t)texW + (uint32_t)(theta + 2));
}

// MathAndSTL.cl:260-266
__device__ __forceinline__ int sample_texture(const CrtTexture& tex, float u, float v)
{
    u = u - floorf(u);
    v = v - floorf(v);
    int uS = f2i((float)tex.width * u);
    int vS = f2i((float)tex.height * v);
    return (int)((uint32_t)vS * (uint32_t)tex.width + (uint32_t)tex.offset + (uint32_t)uS);
}

// kernel_main.cl:277-287
__device__ __forceinline__ v3 raygen_dir(const CrtFrame& F, int i, int j, int width, int height)
{
    float cx = (float)i / (float)width, cy = (float)j / (float)height;
    cx = cx * 2.0f - 1.0f;
    cy = cy * 2.0f - 1.0f;
    v3 t = matmul_xyz(F.invProj, cx, cy, 1.0f, 1.0f);
    float w = matmul_w(F.invProj, cx, cy, 1.0f, 1.0f);
    float tx = t.x / w, ty = t.y / w, tz = t.z / w, tw = w / w;
    v3 wv = matmul_xyz(F.invView, tx, ty, tz, tw);
    return normalize3(wv);
}

__device__ __forceinline__ v3 raygen_dir(const CrtFrame& F, int i, int j) { return raygen_dir(F, i, j, F.width, F.height); }

// One bounce of kernel_main.cl:187-272 after the closest hit is known. Returns false when the
// path terminated (miss -> skybox). On a hit, updates ray/energy/atmospheric/lightDir in place.
// Per-path state carried across the two bounces (kernel_main.cl:179-187). Upstream also carries
//   energy (float3): starts (1,1,1) and is only ever multiplied by a splat (kernel_main.cl:264,268) -> one float;
//   atmosphericLight: (0.255,0.25,0.27)*1 at bounce 0, that * 0.4 at bounce 1 (kernel_main.cl:185,269) -> from `bounce`;
//   lightDir: the sun at bounce 0, the bounce ray's direction at bounce 1 (kernel_main.cl:181,271) -> from `bounce`.
struct PathState { v3 o, d, result; float energy; };

// kernel_main.cl:264: specular = ((1 - roughness) * ndl * shadow) * specularColor * ndl, x component (a splat)
__device__ __forceinline__ float specular_x(float ndl, float shadow)
{
    const float sp = ((1.0f - 0.5f) * ndl) * shadow;
    return (sp * 0.2f) * ndl;
}

// DEFER_ENERGY (CRT_RENDER_SHADOWS): leave `energy *= specular` to the caller, which first traces the shadow ray
// from the new ray origin; `ndlOut` receives the clamped n.l that decides whether the shadow factor is observable.
// REFRACT (CRT_RENDER_REFRACTION, an extension defined by the oracle, oracle/crt_oracle.h): at the first hit of a material
// whose opacity (MTL `d`, kept in Material::roughness) is below 1 the continuing ray is the refracted one (Snell, index
// 1.5; total internal reflection keeps the reflection), 0.01 behind the surface, with (1 - opacity) of the energy.
// Returns 0: path ended (miss -> skybox); 1: continues with the reflected ray; 2: continues with the transmitted ray
// (energy already applied, no shadow ray wanted).
template <bool DEFER_ENERGY = false, bool REFRACT = false>
__device__ __forceinline__ int shade_bounce(const CrtDevScene& S, const Closest& c, PathState& ps, int bounce, float lightY, float lightZ,
                                            float* ndlOut = nullptr)
{
    const float UcharToFloat01 = 1.0f / 255.0f;
    if (c.distance > 99998.0f) {
        // textures[2] for every lane: a scalar load (constant address space, see crt_const_f32x4_ptr)
        const crt_f32x4 skyHdr = ((crt_const_f32x4_ptr)S.textures)[2];
        int idx = clamp_texel(sample_skybox(ps.d, __float_as_int(skyHdr.x), __float_as_int(skyHdr.y)), S.numTexels);
        uint32_t px = S.texels[idx];
        v3 skyc = scale3(mk3((float)(px & 0xffu), (float)((px >> 8) & 0xffu), (float)((px >> 16) & 0xffu)), UcharToFloat01);
        ps.result = add3(ps.result, scale3(skyc, ps.energy));
        return 0;
    }
    const v3 light = bounce == 0 ? mk3(0.0f, lightY, lightZ) : ps.d;     // lightDir = ray.direction after the first bounce
    const v3 atm0 = scale3(mk3(0.255f, 0.25f, 0.27f), 1.0f);
    const v3 atm = bounce == 0 ? atm0 : scale3(atm0, 0.4f);
    const CrtDevInstance I = load_instance(S.devInstances, (uint32_t)c.hitInstance);
    // meshRay of the winning instance, recomputed with the same arithmetic as in the loop
    const v3 mo = xform_xyz(I, ps.o.x, ps.o.y, ps.o.z, 1.0f);
    const v3 md = xform_xyz(I, ps.d.x, ps.d.y, ps.d.z, 0.0f);

    const uint4 c0 = S.triCold[(size_t)c.hit.tri * 2], c1 = S.triCold[(size_t)c.hit.tri * 2 + 1];
    // c0 = {uv0x|uv0y, uv1x|uv1y, uv2x|uv2y, mat|n0x}; c1 = {n0y|n0z, n1x|n1y, n1z|n2x, n2y|n2z}
    const uint32_t matIndex = c0.w & 0xffffu;
    uint32_t mi = __float_as_uint(I.r1.w) + matIndex;
    mi = mi < (uint32_t)CRT_MAX_MATERIALS ? mi : (uint32_t)CRT_MAX_MATERIALS - 1;
    const CrtMaterial mat = S.materials[mi];
    const float bx = (1.0f - c.hit.u) - c.hit.v, by = c.hit.u, bz = c.hit.v;

    const v3 n0 = mat3mul(I, mk3(h2f(c0.w >> 16), h2f(c1.x & 0xffffu), h2f(c1.x >> 16)));
    const v3 n1 = mat3mul(I, mk3(h2f(c1.y & 0xffffu), h2f(c1.y >> 16), h2f(c1.z & 0xffffu)));
    const v3 n2 = mat3mul(I, mk3(h2f(c1.z >> 16), h2f(c1.w & 0xffffu), h2f(c1.w >> 16)));
    const v3 normal = normalize3(add3(add3(scale3(n0, bx), scale3(n1, by)), scale3(n2, bz)));

    const float uvx = (h2f(c0.x & 0xffffu) * bx + h2f(c0.y & 0xffffu) * by) + h2f(c0.z & 0xffffu) * bz;
    const float uvy = (h2f(c0.x >> 16) * bx + h2f(c0.y >> 16) * by) + h2f(c0.z >> 16) * bz;

    uint32_t ti = mat.albedoTextureIndex;
    ti = ti < (uint32_t)CRT_MAX_TEXTURES ? ti : (uint32_t)CRT_MAX_TEXTURES - 1;
    const CrtTexture tex = S.textures[ti];
    const uint32_t px = S.texels[clamp_texel(sample_texture(tex, uvx, uvy), S.numTexels)];
    // the specular texel (kernel_main.cl:243) is fetched upstream but never used: not fetched here
    const uint32_t a = mat.color;
    const uint32_t cr = (((a & 0xffu) * (px & 0xffu)) >> 8) & 0xffu;
    const uint32_t cg = ((((a >> 8) & 0xffu) * ((px >> 8) & 0xffu)) >> 8) & 0xffu;
    const uint32_t cb = ((((a >> 16) & 0xffu) * ((px >> 16) & 0xffu)) >> 8) & 0xffu;
    const v3 color = scale3(mk3((float)cr, (float)cg, (float)cb), UcharToFloat01);
    const v3 point = add3(mo, scale3(md, c.hit.t));


    bool transmitted = false;
    float opacity = 1.0f;
    if (REFRACT) {
        if (bounce == 0) {
            opacity = h2f(mat.roughness);
            if (opacity < 1.0f) {
                const float dn = dot3(normal, ps.d);
                const bool entering = dn < 0.0f;
                const v3 nf = entering ? normal : neg3(normal);
                const float cosi = entering ? (0.0f - dn) : dn;
                const float eta = entering ? 0.6666667f : 1.5f;
                const float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
                if (k >= 0.0f) {
                    const float w = eta * cosi - sqrtf(k);
                    const v3 refr = add3(scale3(ps.d, eta), scale3(nf, w));
                    ps.o = sub3(point, scale3(nf, 0.01f));
                    ps.d = refr;
                    transmitted = true;
                }
            }
        }
    }
    if (!transmitted) {
        ps.o = add3(point, scale3(normal, 0.01f));
        ps.d = reflect3(ps.d, normal);
    }

    float ndl = dot3(normal, neg3(light));
    const v3 ambient = mul3(scale3(atm, fmaxf(0.0f - ndl, 0.1f)), color);
    ndl = fmaxf(ndl, 0.0f);
    // pow(x, shininess) with shininess == 1.0f (kernel_main.cl:250) is exactly x
    const float sl = (ndl * fmaxf(dot3(reflect3(neg3(light), normal), md), 0.0f)) * 0.2f;

    ps.result = add3(ps.result, add3(add3(scale3(scale3(color, ndl), ps.energy), ambient), mk3(sl, sl, sl)));
    if (REFRACT && transmitted) { ps.energy = ps.energy * (1.0f - opacity); return 2; }
    if (DEFER_ENERGY) *ndlOut = ndl;
    else ps.energy = ps.energy * specular_x(ndl, 1.0f);      // shadow = 1.0f (kernel_main.cl:258: no shadow ray upstream)
    return 1;
}
