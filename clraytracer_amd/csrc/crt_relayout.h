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

// Largest squared distance of a vertex of triangles [first, first + count) from the object-space origin, as the bits of a non-negative float
// (atomicMax on the word orders them like the floats; NaN and overflow count as +inf). Hit points -- hence bounce-ray origins, hazard H6 -- lie on
// triangles, so this bounds how far out a bounce ray can start whatever boxes were uploaded around them (crt_instances.h: bounceOriginReach).
__global__ void crt_tri_reach_kernel(const CrtTri* __restrict__ raw, size_t first, size_t count, uint32_t* __restrict__ reachBits)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    if (k < count) {
        const CrtTri t = raw[first + k];
        const float* v[3] = { t.v0, t.v1, t.v2 };
        for (int c = 0; c < 3; ++c) {
            float d2 = (v[c][0] * v[c][0] + v[c][1] * v[c][1]) + v[c][2] * v[c][2];
            if (!(d2 == d2)) d2 = __uint_as_float(0x7F800000u);
            m = d2 > m ? d2 : m;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(reachBits, __float_as_uint(m));
}

// A reference that visits nothing: a leaf whose triangle count comes from bigLeaf[triCap], which crt_init sets to 0.
// Used where an upload is inconsistent (a root index beyond the node array, a mesh without a tree, a malformed node):
// such an instance or subtree renders as empty instead of testing whatever triangle 0 happens to be.
__device__ __host__ __forceinline__ uint32_t crt_empty_ref(uint32_t triCap) { return CRT_LEAF_BIT | triCap; }

__device__ __forceinline__ uint32_t make_ref(const CrtBVHNode& n, uint32_t self, uint32_t nodeCount, uint32_t triCap,
                                             uint32_t* bigLeaf, int* err)
{
    if (n.triCount > 0) {
        if ((uint64_t)n.leftFirst + (uint64_t)n.triCount > (uint64_t)triCap || n.leftFirst > 0x00FFFFFFu) { atomicOr(err, 1); return crt_empty_ref(triCap); }
        if (n.triCount < 128u) return CRT_LEAF_BIT | (n.triCount << 24) | n.leftFirst;
        bigLeaf[n.leftFirst] = n.triCount;
        return CRT_LEAF_BIT | n.leftFirst;
    }
    // children are always allocated after their parent (BVH.cpp:203-204): enforces an acyclic graph
    if (n.leftFirst <= self || (uint64_t)n.leftFirst + 1 >= (uint64_t)nodeCount) { atomicOr(err, 2); return crt_empty_ref(triCap); }
    return n.leftFirst >> 1;
}

__global__ void crt_relayout_nodes(const CrtBVHNode* __restrict__ raw, uint32_t nodeCount, uint32_t triCap,
                                   float4* __restrict__ pairs, uint32_t* __restrict__ bigLeaf, int* __restrict__ err)
{
    uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nodeCount) return;
    const CrtBVHNode node = raw[n];
    if (node.triCount > 0) return;
    const uint32_t l = node.leftFirst;
    if (l <= n || (uint64_t)l + 1 >= (uint64_t)nodeCount) { atomicOr(err, 2); return; }
    const CrtBVHNode L = raw[l], R = raw[l + 1];
    const uint32_t lref = make_ref(L, l, nodeCount, triCap, bigLeaf, err);
    const uint32_t rref = make_ref(R, l + 1, nodeCount, triCap, bigLeaf, err);
    float4* p = pairs + (size_t)(l >> 1) * 4;
    p[0] = make_float4(L.aabbMin[0], L.aabbMin[1], L.aabbMin[2], __uint_as_float(lref));
    p[1] = make_float4(L.aabbMax[0], L.aabbMax[1], L.aabbMax[2], 0.0f);
    p[2] = make_float4(R.aabbMin[0], R.aabbMin[1], R.aabbMin[2], __uint_as_float(rref));
    p[3] = make_float4(R.aabbMax[0], R.aabbMax[1], R.aabbMax[2], 0.0f);
}

__global__ void crt_make_root_refs(const CrtBVHNode* __restrict__ raw, uint32_t nodeCount, uint32_t triCap,
                                   const uint32_t* __restrict__ roots, uint32_t numRoots,
                                   uint32_t* __restrict__ rootRefs, uint32_t* __restrict__ bigLeaf, int* __restrict__ err)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= CRT_MAX_MESHES) return;
    const uint32_t r = k < numRoots ? roots[k] : 0xFFFFFFFFu;
    if (r >= nodeCount) { rootRefs[k] = crt_empty_ref(triCap); return; }      // no tree (yet) for this mesh: renders as empty
    rootRefs[k] = make_ref(raw[r], r, nodeCount, triCap, bigLeaf, err);
}

// Refresh of a frame slot's instance tables in ONE launch (r6): the slot's pinned staging block (raw instance records, bounding spheres, never-culled
// list, instance tree: crt_instances.h kStage*) is read over the host link by this kernel, copied into the slot's device block, and the 64-byte device
// records are built from the raw ones on the way -- where rounds 2-5 queued up to four hipMemcpyAsync (SDMA packets, ~10 us of stream latency each)
// and a relayout launch in front of every frame of an animated scene (upstream: one clEnqueueWriteBuffer, Renderer.cpp:312-320).
__global__ void crt_refresh_instances_kernel(const uint4* __restrict__ staging, uint4* __restrict__ block, uint32_t words16,
                                             const uint32_t* __restrict__ rootRefs, const uint32_t* __restrict__ topRootRefs,
                                             uint32_t count, CrtDevInstance* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint32_t k = i; k < words16; k += stride) block[k] = staging[k];
    if (i >= count) return;
    const CrtMeshInstance m = reinterpret_cast<const CrtMeshInstance*>(staging)[i];      // (kStageInst = 0: the raw records lead the block)
    const uint32_t mesh = m.meshIndex < CRT_MAX_MESHES ? m.meshIndex : 0;
    CrtDevInstance d;
    d.r0 = make_float4(m.inverseTransform.m[0][0], m.inverseTransform.m[0][1], m.inverseTransform.m[0][2], __uint_as_float(rootRefs[mesh]));
    d.r1 = make_float4(m.inverseTransform.m[1][0], m.inverseTransform.m[1][1], m.inverseTransform.m[1][2], __uint_as_float((uint32_t)m.materialStart));
    // (r2.w: the mesh's root as a reference into the tree-top table, or the same global reference as r0.w -- read by CRT_KERNEL=ldstop only)
    d.r2 = make_float4(m.inverseTransform.m[2][0], m.inverseTransform.m[2][1], m.inverseTransform.m[2][2], __uint_as_float(topRootRefs[mesh]));
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
