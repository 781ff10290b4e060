// crt_relayout.h -- upload-time kernels: reference struct layouts -> the CDNA4 layouts of crt_device.h.
#pragma once
#include "crt_device.h"

__global__ void crt_relayout_tris(const CrtTri* __restrict__ raw, size_t first, size_t count,
                                  float* __restrict__ hot, uint4* __restrict__ cold)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const size_t i = first + k;
    const CrtTri t = raw[i];
    float* h = hot + i * 9;
    h[0] = t.v0[0]; h[1] = t.v0[1]; h[2] = t.v0[2];
    h[3] = t.v1[0] - t.v0[0]; h[4] = t.v1[1] - t.v0[1]; h[5] = t.v1[2] - t.v0[2];   // edge1 = y - x
    h[6] = t.v2[0] - t.v0[0]; h[7] = t.v2[1] - t.v0[1]; h[8] = t.v2[2] - t.v0[2];   // edge2 = z - x
    const uint4* tail = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(raw + i) + 48);
    cold[i * 2] = tail[0];
    cold[i * 2 + 1] = tail[1];
}

// ---- hot tiles: the top levels of every mesh's tree get the pair indices [0, CRT_HOT_PAIRS) ----
// `hotSlot[leftFirst >> 1]` = hot index of the sibling pair starting at node leftFirst, or CRT_NOT_HOT. Mesh m owns the
// slots [m * perMesh, (m + 1) * perMesh) in heap order (root pair 0; the pair under the left/right child of pair h is
// 2h+1 / 2h+2), perMesh = the largest power of two <= CRT_HOT_PAIRS / numRoots. Every pair also keeps its ordinary record at
// CRT_HOT_PAIRS + (leftFirst >> 1); child references point at the hot copy when there is one, so a kernel that stages
// pairs[0 .. CRT_HOT_PAIRS) in LDS serves the most visited nodes from there, and every other kernel just sees indices.
#define CRT_NOT_HOT 0xFFFFFFFFu

__global__ void crt_assign_hot_slots(const CrtBVHNode* __restrict__ raw, uint32_t nodeCount, const uint32_t* __restrict__ roots,
                                     uint32_t numRoots, uint32_t perMesh, uint32_t* __restrict__ hotSlot)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= numRoots || perMesh < 2) return;
    const uint32_t root = roots[m];
    if (root >= nodeCount) return;
    // breadth-first over heap indices 0 .. perMesh-2; node index of the PARENT node of pair h kept in a small local queue
    uint32_t parentOf[CRT_HOT_PAIRS];            // heap index -> node whose children form that pair (CRT_NOT_HOT = absent)
    for (uint32_t h = 0; h + 1 < perMesh; ++h) parentOf[h] = CRT_NOT_HOT;
    parentOf[0] = root;
    for (uint32_t h = 0; h + 1 < perMesh; ++h) {
        const uint32_t n = parentOf[h];
        if (n == CRT_NOT_HOT) continue;
        const CrtBVHNode node = raw[n];
        if (node.triCount > 0) continue;                                   // leaf: no pair below it
        const uint32_t l = node.leftFirst;
        if (l <= n || (uint64_t)l + 1 >= (uint64_t)nodeCount) continue;    // invalid link: flagged by crt_relayout_nodes
        hotSlot[l >> 1] = m * perMesh + h;
        if (2 * h + 1 < perMesh - 1) parentOf[2 * h + 1] = l;
        if (2 * h + 2 < perMesh - 1) parentOf[2 * h + 2] = l + 1;
    }
}

__device__ __forceinline__ uint32_t make_ref(const CrtBVHNode& n, uint32_t self, uint32_t nodeCount, uint32_t triCap,
                                             uint32_t* bigLeaf, const uint32_t* hotSlot, int* err)
{
    if (n.triCount > 0) {
        if ((uint64_t)n.leftFirst + (uint64_t)n.triCount > (uint64_t)triCap || n.leftFirst > 0x00FFFFFFu) { atomicOr(err, 1); return CRT_LEAF_BIT | (1u << 24); }
        if (n.triCount < 128u) return CRT_LEAF_BIT | (n.triCount << 24) | n.leftFirst;
        bigLeaf[n.leftFirst] = n.triCount;
        return CRT_LEAF_BIT | n.leftFirst;
    }
    // children are always allocated after their parent (BVH.cpp:203-204): enforces an acyclic graph
    if (n.leftFirst <= self || (uint64_t)n.leftFirst + 1 >= (uint64_t)nodeCount) { atomicOr(err, 2); return CRT_LEAF_BIT | (1u << 24); }
    const uint32_t hot = hotSlot[n.leftFirst >> 1];
    return hot != CRT_NOT_HOT ? hot : (uint32_t)CRT_HOT_PAIRS + (n.leftFirst >> 1);
}

__global__ void crt_relayout_nodes(const CrtBVHNode* __restrict__ raw, uint32_t nodeCount, uint32_t triCap,
                                   float4* __restrict__ pairs, uint32_t* __restrict__ bigLeaf, const uint32_t* __restrict__ hotSlot,
                                   int* __restrict__ err)
{
    uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nodeCount) return;
    const CrtBVHNode node = raw[n];
    if (node.triCount > 0) return;
    const uint32_t l = node.leftFirst;
    if (l <= n || (uint64_t)l + 1 >= (uint64_t)nodeCount) { atomicOr(err, 2); return; }
    const CrtBVHNode L = raw[l], R = raw[l + 1];
    const uint32_t lref = make_ref(L, l, nodeCount, triCap, bigLeaf, hotSlot, err);
    const uint32_t rref = make_ref(R, l + 1, nodeCount, triCap, bigLeaf, hotSlot, err);
    const float4 r0 = make_float4(L.aabbMin[0], L.aabbMin[1], L.aabbMin[2], __uint_as_float(lref));
    const float4 r1 = make_float4(L.aabbMax[0], L.aabbMax[1], L.aabbMax[2], 0.0f);
    const float4 r2 = make_float4(R.aabbMin[0], R.aabbMin[1], R.aabbMin[2], __uint_as_float(rref));
    const float4 r3 = make_float4(R.aabbMax[0], R.aabbMax[1], R.aabbMax[2], 0.0f);
    float4* p = pairs + ((size_t)CRT_HOT_PAIRS + (l >> 1)) * 4;
    p[0] = r0; p[1] = r1; p[2] = r2; p[3] = r3;
    const uint32_t hot = hotSlot[l >> 1];
    if (hot != CRT_NOT_HOT) { float4* q = pairs + (size_t)hot * 4; q[0] = r0; q[1] = r1; q[2] = r2; q[3] = r3; }
}

__global__ void crt_make_root_refs(const CrtBVHNode* __restrict__ raw, uint32_t nodeCount, uint32_t triCap,
                                   const uint32_t* __restrict__ roots, uint32_t numRoots,
                                   uint32_t* __restrict__ rootRefs, uint32_t* __restrict__ bigLeaf, const uint32_t* __restrict__ hotSlot,
                                   int* __restrict__ err)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= numRoots) return;
    const uint32_t r = roots[k];
    if (r >= nodeCount) { rootRefs[k] = CRT_LEAF_BIT | (1u << 24); return; } // not (yet) uploaded: harmless dummy, flagged at render
    rootRefs[k] = make_ref(raw[r], r, nodeCount, triCap, bigLeaf, hotSlot, err);
}

__global__ void crt_relayout_instances(const CrtMeshInstance* __restrict__ raw, const uint32_t* __restrict__ rootRefs,
                                       uint32_t count, CrtDevInstance* __restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const CrtMeshInstance m = raw[i];
    const uint32_t mesh = m.meshIndex < CRT_MAX_MESHES ? m.meshIndex : 0;
    CrtDevInstance d;
    d.r0 = make_float4(m.inverseTransform.m[0][0], m.inverseTransform.m[0][1], m.inverseTransform.m[0][2], __uint_as_float(rootRefs[mesh]));
    d.r1 = make_float4(m.inverseTransform.m[1][0], m.inverseTransform.m[1][1], m.inverseTransform.m[1][2], __uint_as_float((uint32_t)m.materialStart));
    d.r2 = make_float4(m.inverseTransform.m[2][0], m.inverseTransform.m[2][1], m.inverseTransform.m[2][2], 0.0f);
    d.r3 = make_float4(m.inverseTransform.m[3][0], m.inverseTransform.m[3][1], m.inverseTransform.m[3][2], 0.0f);
    out[i] = d;
}

__global__ void crt_relayout_texels(const uint8_t* __restrict__ raw, size_t firstTexel, size_t count, uint32_t* __restrict__ texels)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const size_t i = firstTexel + k;
    const uint8_t* p = raw + i * 3;
    texels[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}
