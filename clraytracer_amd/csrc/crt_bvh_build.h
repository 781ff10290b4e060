// crt_bvh_build.h -- BuildBVH (BVH.cpp:218-255) on the device: the reference's 8-bin SAH builder, level by level,
// producing the SAME bytes the sequential host builder produces -- triangle order, node numbering, bounds.
//
// Why that is possible although upstream is a depth-first recursion over an in-place partition:
//  * bounds, centroid ranges, bin counts and bin boxes are min/max/integer reductions -> order-free. (Where a sequential
//    `a < b ? a : b` fold can return either sign of zero, only *stored* node bounds are affected; they are fixed up
//    below by looking at the last zero in sequence order.)
//  * the 21-plane sweep of one node is 21 dependent steps -> one thread replays it literally.
//  * the partition loop `while (i <= j) { if (c[i] < split) i++; else swap(t[i], t[j--]); }` (BVH.cpp:185-192) has a
//    closed form. Examine order: elements come from the front while they are left-class (a right-class one ends the
//    front phase), then from the back while they are right-class (a left-class one ends the back phase), and so on;
//    left-class elements fill positions 0,1,.. and right-class ones n-1,n-2,.. in examine order. With L = number of
//    left-class elements, holes r_1<r_2<.. = right-class positions below L, and l_1<l_2<.. = back-order indices
//    (y = n-1-x) of the left-class elements at or above L, that gives
//        front left  x            -> x                      back left  #m  -> r_m
//        front right #m (also the right-class element AT L, as #G+1)       -> back slot l_{m-1}+1  (l_0 = -1)
//        back  right y            -> back slot y+1          (array position x-1)
//    which two prefix sums and two small tables evaluate in parallel. It also covers the degenerate partitions
//    (L == 0 or L == n) after which upstream keeps the node a leaf but leaves its triangles permuted.
//  * nodes are numbered by the recursion (children = next two free indices, then the whole left subtree, then the
//    right one): index(left) = base, index(right) = base+1, base(left) = base+2, base(right) = base+2+descendants(left).
//    A subtree with k leaves has 2k-1 nodes, and the leaves left of a node inside its mesh are exactly those of the left
//    siblings along its path, so with S = exclusive prefix count of leaf starts over the triangle pool
//        base(P) = index(root) + 1 + 2 depth(P) + 2 (S[first(P)] - S[first(root)]) - 2 rightTurns(P)
//        index(P) = base(P) - 2 for a left child, base(P) + 1 - 2 (S[first(P)] - S[first(left sibling)]) for a right child
//        index(root of mesh m) = firstNode + 2 S[first(root)] - m
//    -- one scan and one pass over the nodes once the tree is known (r2 ran two launches per level for this).
// Triangles move between two buffers, one level per pass; a finished leaf's segment is written to both.
#pragma once
#include "crt_device.h"

#define CRT_BVH_BINS 8
#define CRT_BVH_NONE 0xFFFFFFFFu

struct CrtBuildNode {
    uint32_t first, count;           // triangle range (absolute indices into the triangle pool)
    uint32_t left, right;            // build-node ids of the children, CRT_BVH_NONE for a leaf
    float bmin[3], bmax[3];          // UpdateNodeBounds
    float splitPos; int axis;
    uint32_t state;                  // 0 = undecided, 1 = split, 2 = leaf, 3 = leaf whose triangles the failed partition permuted
    uint32_t mesh;                   // mesh (root) this node belongs to
    uint32_t depth, rightTurns;      // of the path from the root: what the closed-form numbering needs
    uint32_t isRight;                // 1 = right child (its left sibling is the build node before it)
};

__device__ __forceinline__ const float* bvh_tri_f(const CrtTri* t, size_t i) { return reinterpret_cast<const float*>(t + i); }
__device__ __forceinline__ float bvh_centroid(const CrtTri* t, size_t i, int axis) { return bvh_tri_f(t, i)[3 + 4 * axis]; }
__device__ __forceinline__ uint32_t bvh_ordered(float f) { const uint32_t b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ float bvh_unordered(uint32_t o) { return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }
// BVH.cpp:41-46 + hsum_ps_sse3 (SIMDCommon.hpp:183-189): (e0*e0 + e1*e0) + (e2*e2 + 0)
__device__ __forceinline__ float bvh_area(const float* mn, const float* mx)
{
    const float e0 = mx[0] - mn[0], e1 = mx[1] - mn[1], e2 = mx[2] - mn[2];
    return (e0 * e0 + e1 * e0) + (e2 * e2 + 0.0f);
}

// BVH.cpp:232-234
__global__ void crt_bvh_centroids(CrtTri* __restrict__ tris, size_t first, size_t count)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    float* t = reinterpret_cast<float*>(tris + first + k);
    t[3] = ((t[0] + t[4]) + t[8]) * 0.333333f;
    t[7] = ((t[1] + t[5]) + t[9]) * 0.333333f;
    t[11] = ((t[2] + t[6]) + t[10]) * 0.333333f;
}

// The nodes of one level are kept in three compact id lists by size: BIG nodes (> CRT_BVH_SMALL triangles) are cut into chunks of
// CRT_BVH_CHUNK triangles, one workgroup per chunk (crt_bvh_big_*); MID nodes get one wave each (crt_bvh_bounds_wave, crt_bvh_mid);
// TINY nodes (<= CRT_BVH_TINY triangles -- the bulk of the deep levels: the builder splits down to one or two triangles per leaf) get
// one THREAD each that replays upstream's sequential code literally (crt_bvh_tiny).
#ifndef CRT_BVH_SMALL
#define CRT_BVH_SMALL 512             // (1024: 3-10 % slower, 2048: 5-20 %: the first wave-per-node levels are a few dozen long single-wave loops; 256: the same as 512)
#endif
#ifndef CRT_BVH_TINY
#define CRT_BVH_TINY 8
#endif
enum { CRT_BVH_CLASS_BIG = 0, CRT_BVH_CLASS_MID = 1, CRT_BVH_CLASS_TINY = 2 };
__host__ __device__ __forceinline__ int bvh_class(uint32_t count) { return count > CRT_BVH_SMALL ? CRT_BVH_CLASS_BIG : (count > CRT_BVH_TINY ? CRT_BVH_CLASS_MID : CRT_BVH_CLASS_TINY); }
// Children are appended to the node array and to next level's lists with ONE 64-bit atomic per splitting node (per wave
// in crt_bvh_tiny): the counter packs next level's three list sizes -- TINY in bits 0..23, MID in 24..43 (a MID node has
// more than 8 triangles: < 2^20 of them per level), BIG in 44..63 (more than CRT_BVH_SMALL = 512 triangles: < 2^13 of them in the 2.4 M-triangle pool) -- and because every
// child takes exactly one list slot, the number of children created so far is the sum of the three fields, which
// numbers the new nodes. (Three separate same-address atomics per node cost 20 of the 34 ms of a 1 M-triangle build.)
#define CRT_BVH_PACK_TINY(x) ((unsigned long long)(x))
#define CRT_BVH_PACK_MID(x) ((unsigned long long)(x) << 24)
#define CRT_BVH_PACK_BIG(x) ((unsigned long long)(x) << 44)
__host__ __device__ __forceinline__ uint32_t bvh_unpack(unsigned long long p, int cls)
{
    return cls == CRT_BVH_CLASS_TINY ? (uint32_t)(p & 0xFFFFFFull) : (cls == CRT_BVH_CLASS_MID ? (uint32_t)((p >> 24) & 0xFFFFFull) : (uint32_t)((p >> 44) & 0xFFFFFull));
}
__host__ __device__ __forceinline__ unsigned long long bvh_pack_one(int cls)
{
    return cls == CRT_BVH_CLASS_TINY ? CRT_BVH_PACK_TINY(1) : (cls == CRT_BVH_CLASS_MID ? CRT_BVH_PACK_MID(1) : CRT_BVH_PACK_BIG(1));
}
struct CrtBuildLists { uint32_t* list[3]; };   // next level's id lists (device pointers)
// writes the two child records of `node`; `before` = the packed counter before this node's two children were added
__device__ __forceinline__ void bvh_new_children(CrtBuildNode* nodes, CrtBuildNode& node, uint32_t first, uint32_t L, uint32_t n,
                                                 uint32_t levelEnd, unsigned long long before, const CrtBuildLists& next)
{
    const uint32_t id = levelEnd + bvh_unpack(before, 0) + bvh_unpack(before, 1) + bvh_unpack(before, 2);
    node.left = id; node.right = id + 1;
    CrtBuildNode c;
    c.left = c.right = CRT_BVH_NONE; c.splitPos = 0.0f; c.axis = 0; c.state = 0; c.mesh = node.mesh; c.depth = node.depth + 1; c.rightTurns = node.rightTurns; c.isRight = 0;
    for (int k = 0; k < 3; ++k) { c.bmin[k] = 1e30f; c.bmax[k] = -1e30f; }
    c.first = first; c.count = L; nodes[id] = c;
    c.first = first + L; c.count = n - L; c.rightTurns = node.rightTurns + 1; c.isRight = 1; nodes[id + 1] = c;
    const int cl = bvh_class(L), cr = bvh_class(n - L);
    next.list[cl][bvh_unpack(before, cl)] = id;
    next.list[cr][bvh_unpack(before + bvh_pack_one(cl), cr)] = id + 1;
}

// wave-level reductions on ordered images / indices
__device__ __forceinline__ uint32_t bvh_wave_min(uint32_t v) { for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64); v = o < v ? o : v; } return v; }
__device__ __forceinline__ uint32_t bvh_wave_max(uint32_t v) { for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64); v = o > v ? o : v; } return v; }
__device__ __forceinline__ int bvh_wave_max_i(int v) { for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off, 64); v = o > v ? o : v; } return v; }
// what one lane wrote (LDS or global) is read by another lane of the same wave after this
__device__ __forceinline__ void bvh_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// The sweep (BVH.cpp:131-160) by the first 21 lanes of a wave, one lane per candidate plane: lane = 7 * axis + plane. Growing the bin
// boxes in sweep order is a min/max over the non-empty bins on that side, so a lane can form its two boxes on its own;
// `planeCost < bestCost` taken in sequence order keeps the FIRST plane that reaches the minimum = the lexicographic minimum of
// (cost, lane); a NaN cost (an empty side: 0 x Inf) or one that is not below the initial 1e30 never wins. Bins: cnt[3][8],
// bmin/bmax[3][8][3] as ordered images, in LDS or global memory. All 64 lanes must call; the results are wave-uniform.
__device__ __forceinline__ void bvh_sweep_lanes(const uint32_t* cnt, const uint32_t* bmin, const uint32_t* bmax, const float* cmin, const float* cmax, uint32_t lane,
                                                int& axis, float& splitPos, float& bestCost)
{
    float cost = 1e30f, myPos = 0.0f; uint32_t idx = 64u + lane;
    if (lane < 3 * (CRT_BVH_BINS - 1)) {
        const int a = (int)lane / (CRT_BVH_BINS - 1), p = (int)lane % (CRT_BVH_BINS - 1);
        const float lo = a == 0 ? cmin[0] : (a == 1 ? cmin[1] : cmin[2]), hi = a == 0 ? cmax[0] : (a == 1 ? cmax[1] : cmax[2]);
        if (!(hi == lo)) {
            int leftSum = 0, rightSum = 0;
            float lmn[3] = { 1e30f, 1e30f, 1e30f }, lmx[3] = { -1e30f, -1e30f, -1e30f };
            float rmn[3] = { 1e30f, 1e30f, 1e30f }, rmx[3] = { -1e30f, -1e30f, -1e30f };
            for (int i = 0; i < CRT_BVH_BINS; ++i) {
                const int c_ = (int)cnt[a * CRT_BVH_BINS + i];
                float bmn[3], bmx[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { bmn[c] = bvh_unordered(bmin[(a * CRT_BVH_BINS + i) * 3 + c]); bmx[c] = bvh_unordered(bmax[(a * CRT_BVH_BINS + i) * 3 + c]); }
                const bool grow = bmn[0] != 1e30f;           // aabb::grow(aabb) (BVH.cpp:29-37): skipped for an empty box
                if (i <= p) {
                    leftSum += c_;
                    if (grow)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            lmn[c] = lmn[c] < bmn[c] ? lmn[c] : bmn[c]; lmx[c] = lmx[c] > bmn[c] ? lmx[c] : bmn[c];
                            lmn[c] = lmn[c] < bmx[c] ? lmn[c] : bmx[c]; lmx[c] = lmx[c] > bmx[c] ? lmx[c] : bmx[c];
                        }
                } else {
                    rightSum += c_;
                    if (grow)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            rmn[c] = rmn[c] < bmn[c] ? rmn[c] : bmn[c]; rmx[c] = rmx[c] > bmn[c] ? rmx[c] : bmn[c];
                            rmn[c] = rmn[c] < bmx[c] ? rmn[c] : bmx[c]; rmx[c] = rmx[c] > bmx[c] ? rmx[c] : bmx[c];
                        }
                }
            }
            const float planeCost = (float)leftSum * bvh_area(lmn, lmx) + (float)rightSum * bvh_area(rmn, rmx);
            const float scale = (hi - lo) / (float)CRT_BVH_BINS;
            myPos = lo + scale * (float)(p + 1);
            if (planeCost < 1e30f) { cost = planeCost; idx = lane; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float oc = __shfl_xor(cost, off, 64); const uint32_t oi = (uint32_t)__shfl_xor((int)idx, off, 64);
        if (oc < cost || (oc == cost && oi < idx)) { cost = oc; idx = oi; }
    }
    const bool found = idx < 3u * (CRT_BVH_BINS - 1);
    splitPos = found ? __shfl(myPos, (int)(found ? idx : 0u), 64) : 0.0f;
    axis = found ? (int)idx / (CRT_BVH_BINS - 1) : 0;
    bestCost = found ? cost : 1e30f;
}

// block-wide exclusive scan of one flag per thread; returns this thread's rank and the block total
__device__ __forceinline__ uint32_t bvh_block_scan(uint32_t flag, uint32_t* s_wave, uint32_t& total)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const unsigned long long m = __ballot(flag != 0);
    const uint32_t inWave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < nw; ++w) { const uint32_t c = s_wave[w]; if (w < wave) base += c; tot += c; }
    __syncthreads();
    total = tot;
    return base + inWave;
}

__device__ __forceinline__ void bvh_copy_tri(CrtTri* __restrict__ dst, size_t d, const CrtTri* __restrict__ src, size_t s)
{
    const uint4* a = reinterpret_cast<const uint4*>(src + s);
    uint4* b = reinterpret_cast<uint4*>(dst + d);
    const uint4 q0 = a[0], q1 = a[1], q2 = a[2], q3 = a[3], q4 = a[4];
    b[0] = q0; b[1] = q1; b[2] = q2; b[3] = q3; b[4] = q4;
}

// ---- BIG nodes (> CRT_BVH_SMALL triangles): CRT_BVH_CHUNK triangles per workgroup, as many workgroups as the level's BIG nodes need ----
// One workgroup per node (r2) left the top of the tree on 8-128 CUs: 4.9 of the 11.6 ms of a 1 M-triangle build went into the first
// eight levels. Here a level's BIG nodes are cut into chunks of CRT_BVH_CHUNK triangles; `chunkNode[c]` names the node (its slot in the
// level's BIG list) chunk c belongs to, and CrtBigScratch[slot] holds what the node's chunks reduce into with global atomics (bounds,
// centroid range, bins) or agree on (L, G). The closed-form partition needs ordered ranks across chunks: per-chunk counts, which every
// chunk sums for itself over its node's chunks (a node has at most a thousand), then ranks inside the chunk by a workgroup scan.
#define CRT_BVH_CHUNK 1024
#define CRT_BVH_BIG_THREADS 256
struct CrtBigScratch {
    uint32_t bmin[3], bmax[3];        // vertex bounds, ordered images
    uint32_t cmin[3], cmax[3];        // centroid range
    int last[6];                      // last zero in sequence order per bound (3 * triangle + vertex), -1 = none
    uint32_t chunkBase, nChunks;      // this node's chunks: chunkBase .. chunkBase + nChunks
    uint32_t L, G;
    uint32_t cnt[3 * CRT_BVH_BINS];
    uint32_t bbmin[3 * CRT_BVH_BINS * 3], bbmax[3 * CRT_BVH_BINS * 3];
};
struct CrtBuildCtl { unsigned long long packed; uint32_t nextChunks; uint32_t degenerate; };   // read back by the host after every level
// The read-back: the last launch of a level copies its control record into pinned host memory and then raises `seq`; the host, which needs
// the list sizes to shape the next level's launches, spins on `seq` instead of paying a copy + stream synchronisation per level.
struct CrtBuildCtlHost { CrtBuildCtl ctl; volatile uint32_t seq; uint32_t pad[3]; };
__global__ void crt_bvh_publish(const CrtBuildCtl* __restrict__ ctl, CrtBuildCtlHost* __restrict__ host, uint32_t seq)
{
    host->ctl.packed = ctl->packed; host->ctl.nextChunks = ctl->nextChunks; host->ctl.degenerate = ctl->degenerate;
    __threadfence_system();
    host->seq = seq;
}
__host__ __device__ __forceinline__ uint32_t bvh_chunks(uint32_t n) { return (n + CRT_BVH_CHUNK - 1) / CRT_BVH_CHUNK; }

// sum over the workgroup; every thread gets the total. s_red: one word per wave
__device__ __forceinline__ uint32_t bvh_block_sum(uint32_t v, uint32_t* s_red)
{
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
    const uint32_t nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
    for (uint32_t w = 0; w < nw; ++w) t += s_red[w];
    __syncthreads();
    return t;
}

// level 0: the roots; BIG ones get their slot (mesh order, as the host fills the BIG list) and their chunks
__global__ void crt_bvh_init_roots(CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ meshTriCounts, int numMeshes, uint32_t firstTri,
                                   CrtBigScratch* __restrict__ big, uint32_t* __restrict__ chunkNode)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t cur = firstTri, slot = 0, chunk = 0;
    for (int m = 0; m < numMeshes; ++m) {
        CrtBuildNode n;
        n.first = cur; n.count = meshTriCounts[m]; n.left = n.right = CRT_BVH_NONE;
        for (int c = 0; c < 3; ++c) { n.bmin[c] = 1e30f; n.bmax[c] = -1e30f; }
        n.splitPos = 0.0f; n.axis = 0; n.state = 0; n.mesh = (uint32_t)m; n.depth = 0; n.rightTurns = 0; n.isRight = 0;
        nodes[m] = n;
        cur += meshTriCounts[m];
        if (bvh_class(n.count) == CRT_BVH_CLASS_BIG) {
            const uint32_t nch = bvh_chunks(n.count);
            big[slot].chunkBase = chunk; big[slot].nChunks = nch;
            for (uint32_t c = 0; c < nch; ++c) chunkNode[chunk + c] = slot;
            chunk += nch; ++slot;
        }
    }
}

__global__ void crt_bvh_big_reset(CrtBigScratch* __restrict__ big, uint32_t count)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    CrtBigScratch& B = big[k];
    for (int c = 0; c < 3; ++c) { B.bmin[c] = B.cmin[c] = bvh_ordered(1e30f); B.bmax[c] = B.cmax[c] = bvh_ordered(-1e30f); }
    for (int c = 0; c < 6; ++c) B.last[c] = -1;
    B.L = 0; B.G = 0;
    for (int j = 0; j < 3 * CRT_BVH_BINS; ++j) B.cnt[j] = 0;
    for (int j = 0; j < 9 * CRT_BVH_BINS; ++j) { B.bbmin[j] = bvh_ordered(1e30f); B.bbmax[j] = bvh_ordered(-1e30f); }
}

// UpdateNodeBounds (BVH.cpp:54-74) + the centroid range of FindBestSplitPlane (BVH.cpp:108-113): min/max reductions, order-free.
__global__ void __launch_bounds__(CRT_BVH_BIG_THREADS) crt_bvh_big_bounds(const CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ list, CrtBigScratch* __restrict__ big,
                                                                          const uint32_t* __restrict__ chunkNode, const CrtTri* __restrict__ tris)
{
    const uint32_t k = chunkNode[blockIdx.x];
    CrtBigScratch& B = big[k];
    const CrtBuildNode& node = nodes[list[k]];
    const uint32_t first = node.first, n = node.count;
    const uint32_t lo = (blockIdx.x - B.chunkBase) * CRT_BVH_CHUNK, hi = (lo + CRT_BVH_CHUNK) < n ? (lo + CRT_BVH_CHUNK) : n;
    float mn[3] = { 1e30f, 1e30f, 1e30f }, mx[3] = { -1e30f, -1e30f, -1e30f };
    float cn[3] = { 1e30f, 1e30f, 1e30f }, cx[3] = { -1e30f, -1e30f, -1e30f };
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float* t = bvh_tri_f(tris, (size_t)first + i);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int v = 0; v < 3; ++v) { const float x = t[4 * v + c]; mn[c] = mn[c] < x ? mn[c] : x; mx[c] = mx[c] > x ? mx[c] : x; }
            const float ce = t[3 + 4 * c]; cn[c] = cn[c] < ce ? cn[c] : ce; cx[c] = cx[c] > ce ? cx[c] : ce;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t a = bvh_wave_min(bvh_ordered(mn[c])), b = bvh_wave_max(bvh_ordered(mx[c]));
        const uint32_t d = bvh_wave_min(bvh_ordered(cn[c])), e = bvh_wave_max(bvh_ordered(cx[c]));
        if ((threadIdx.x & 63) == 0) { atomicMin(&B.bmin[c], a); atomicMax(&B.bmax[c], b); atomicMin(&B.cmin[c], d); atomicMax(&B.cmax[c], e); }
    }
}

// The bins of FindBestSplitPlane (BVH.cpp:115-129): per workgroup in LDS, then one global atomic per non-empty bin word. Also the
// look-up behind the sign of a zero bound (see crt_bvh_bounds_wave), which needs the finished bounds and reads the same triangles.
__global__ void __launch_bounds__(CRT_BVH_BIG_THREADS) crt_bvh_big_bins(const CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ list, CrtBigScratch* __restrict__ big,
                                                                        const uint32_t* __restrict__ chunkNode, const CrtTri* __restrict__ tris)
{
    __shared__ uint32_t s_cnt[3 * CRT_BVH_BINS];
    __shared__ uint32_t s_bmin[9 * CRT_BVH_BINS], s_bmax[9 * CRT_BVH_BINS];
    const uint32_t k = chunkNode[blockIdx.x];
    CrtBigScratch& B = big[k];
    const CrtBuildNode& node = nodes[list[k]];
    const uint32_t first = node.first, n = node.count;
    const uint32_t lo = (blockIdx.x - B.chunkBase) * CRT_BVH_CHUNK, hi = (lo + CRT_BVH_CHUNK) < n ? (lo + CRT_BVH_CHUNK) : n;
    for (int j = threadIdx.x; j < 3 * CRT_BVH_BINS; j += blockDim.x) s_cnt[j] = 0;
    for (int j = threadIdx.x; j < 9 * CRT_BVH_BINS; j += blockDim.x) { s_bmin[j] = bvh_ordered(1e30f); s_bmax[j] = bvh_ordered(-1e30f); }
    float cmin[3], cmax[3], rmin[3], rmax[3];
    bool anyZero = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        cmin[a] = bvh_unordered(B.cmin[a]); cmax[a] = bvh_unordered(B.cmax[a]);
        rmin[a] = bvh_unordered(B.bmin[a]); rmax[a] = bvh_unordered(B.bmax[a]);
        anyZero = anyZero || rmin[a] == 0.0f || rmax[a] == 0.0f;
    }
    __syncthreads();
    int last[6] = { -1, -1, -1, -1, -1, -1 };
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float* t = bvh_tri_f(tris, (size_t)first + i);
        float tmn[3], tmx[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float l = 1e30f, h = -1e30f;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float x = t[4 * v + c]; l = l < x ? l : x; h = h > x ? h : x;
                if (anyZero && x == 0.0f) {
                    const int s = (int)(i * 3 + v);
                    if (rmin[c] == 0.0f) last[c] = last[c] > s ? last[c] : s;
                    if (rmax[c] == 0.0f) last[3 + c] = last[3 + c] > s ? last[3 + c] : s;
                }
            }
            tmn[c] = l; tmx[c] = h;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (cmax[a] == cmin[a]) continue;
            const float scale = (float)CRT_BVH_BINS / (cmax[a] - cmin[a]);
            int b = f2i((t[3 + 4 * a] - cmin[a]) * scale);
            b = (CRT_BVH_BINS - 1) < b ? (CRT_BVH_BINS - 1) : b;
            if (b < 0) b = 0;
            atomicAdd(&s_cnt[a * CRT_BVH_BINS + b], 1u);
#pragma unroll
            for (int c = 0; c < 3; ++c) { atomicMin(&s_bmin[(a * CRT_BVH_BINS + b) * 3 + c], bvh_ordered(tmn[c])); atomicMax(&s_bmax[(a * CRT_BVH_BINS + b) * 3 + c], bvh_ordered(tmx[c])); }
        }
    }
    if (anyZero)
#pragma unroll
        for (int c = 0; c < 6; ++c) { const int m = bvh_wave_max_i(last[c]); if ((threadIdx.x & 63) == 0 && m >= 0) atomicMax(&B.last[c], m); }
    __syncthreads();
    for (int j = threadIdx.x; j < 3 * CRT_BVH_BINS; j += blockDim.x) {
        if (s_cnt[j] == 0) continue;
        atomicAdd(&B.cnt[j], s_cnt[j]);
        for (int c = 0; c < 3; ++c) { atomicMin(&B.bbmin[j * 3 + c], s_bmin[j * 3 + c]); atomicMax(&B.bbmax[j * 3 + c], s_bmax[j * 3 + c]); }
    }
}

// The sweep and the split decision (BVH.cpp:131-163, 169-177), replayed by every chunk of the node for itself (the node's first chunk
// stores them), then the chunk's share of L = the number of left-class triangles. A node that stays a leaf by the cost test is
// copied to the other buffer here, in the same order.
__global__ void __launch_bounds__(CRT_BVH_BIG_THREADS) crt_bvh_big_sweep(CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ list, const CrtBigScratch* __restrict__ big,
                                                                         const uint32_t* __restrict__ chunkNode, const CrtTri* __restrict__ src, CrtTri* __restrict__ dst,
                                                                         uint32_t* __restrict__ chunkL)
{
    __shared__ uint32_t s_red[CRT_BVH_BIG_THREADS / 64];
    const uint32_t k = chunkNode[blockIdx.x];
    const CrtBigScratch& B = big[k];
    CrtBuildNode& node = nodes[list[k]];
    const uint32_t first = node.first, n = node.count;
    const uint32_t chunk = blockIdx.x - B.chunkBase;
    const uint32_t lo = chunk * CRT_BVH_CHUNK, hi = (lo + CRT_BVH_CHUNK) < n ? (lo + CRT_BVH_CHUNK) : n;
    float cmin[3], cmax[3], rmin[3], rmax[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        cmin[c] = bvh_unordered(B.cmin[c]); cmax[c] = bvh_unordered(B.cmax[c]);
        rmin[c] = bvh_unordered(B.bmin[c]); rmax[c] = bvh_unordered(B.bmax[c]);
        if (rmin[c] == 0.0f && B.last[c] >= 0) rmin[c] = bvh_tri_f(src, (size_t)first + B.last[c] / 3)[4 * (B.last[c] % 3) + c];
        if (rmax[c] == 0.0f && B.last[3 + c] >= 0) rmax[c] = bvh_tri_f(src, (size_t)first + B.last[3 + c] / 3)[4 * (B.last[3 + c] % 3) + c];
    }
    int axis; float splitPos, bestCost;                        // every wave replays the sweep: no broadcast needed
    bvh_sweep_lanes(B.cnt, B.bbmin, B.bbmax, cmin, cmax, threadIdx.x & 63, axis, splitPos, bestCost);
    const float nosplitCost = (float)n * bvh_area(rmin, rmax);
    const bool isLeaf = bestCost >= nosplitCost;
    if (chunk == 0 && threadIdx.x == 0) {
        for (int c = 0; c < 3; ++c) { node.bmin[c] = rmin[c]; node.bmax[c] = rmax[c]; }
        node.axis = axis; node.splitPos = splitPos; node.state = isLeaf ? 2u : 1u;
    }
    if (isLeaf) {
        for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) bvh_copy_tri(dst, (size_t)first + i, src, (size_t)first + i);
        return;
    }
    uint32_t c = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) c += bvh_centroid(src, (size_t)first + i, axis) < splitPos ? 1u : 0u;
    c = bvh_block_sum(c, s_red);
    if (threadIdx.x == 0) chunkL[blockIdx.x] = c;
}

// L, then the chunk's number of right-class triangles below L (the holes) and of left-class triangles at or above L.
__global__ void __launch_bounds__(CRT_BVH_BIG_THREADS) crt_bvh_big_count(const CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ list, CrtBigScratch* __restrict__ big,
                                                                         const uint32_t* __restrict__ chunkNode, const CrtTri* __restrict__ src, const uint32_t* __restrict__ chunkL,
                                                                         uint32_t* __restrict__ chunkFR, uint32_t* __restrict__ chunkBL)
{
    __shared__ uint32_t s_red[CRT_BVH_BIG_THREADS / 64];
    const uint32_t k = chunkNode[blockIdx.x];
    CrtBigScratch& B = big[k];
    const CrtBuildNode& node = nodes[list[k]];
    if (node.state != 1u) return;
    const uint32_t first = node.first, n = node.count;
    const uint32_t chunk = blockIdx.x - B.chunkBase;
    const uint32_t lo = chunk * CRT_BVH_CHUNK, hi = (lo + CRT_BVH_CHUNK) < n ? (lo + CRT_BVH_CHUNK) : n;
    const int axis = node.axis; const float splitPos = node.splitPos;
    uint32_t l = 0;
    for (uint32_t j = threadIdx.x; j < B.nChunks; j += blockDim.x) l += chunkL[B.chunkBase + j];
    const uint32_t L = bvh_block_sum(l, s_red);
    uint32_t fr = 0, bl = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const bool isLeft = bvh_centroid(src, (size_t)first + i, axis) < splitPos;
        fr += (i < L && !isLeft) ? 1u : 0u; bl += (i >= L && isLeft) ? 1u : 0u;
    }
    fr = bvh_block_sum(fr, s_red); bl = bvh_block_sum(bl, s_red);
    if (threadIdx.x == 0) { chunkFR[blockIdx.x] = fr; chunkBL[blockIdx.x] = bl; if (chunk == 0) B.L = L; }
}

// The tables of the closed form (see the header): rank of every right-class element of the front region among the holes and
// holes[m] = x; rank of every left-class element of the back region in back order (y = n-1-x) and backL[m] = y.
__global__ void __launch_bounds__(CRT_BVH_BIG_THREADS) crt_bvh_big_tables(const CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ list, CrtBigScratch* __restrict__ big,
                                                                          const uint32_t* __restrict__ chunkNode, const CrtTri* __restrict__ src, uint32_t poolFirst,
                                                                          const uint32_t* __restrict__ chunkFR, const uint32_t* __restrict__ chunkBL,
                                                                          uint32_t* __restrict__ rank, uint32_t* __restrict__ holes, uint32_t* __restrict__ backL)
{
    __shared__ uint32_t s_red[CRT_BVH_BIG_THREADS / 64];
    __shared__ uint32_t s_wave[16];
    const uint32_t k = chunkNode[blockIdx.x];
    CrtBigScratch& B = big[k];
    const CrtBuildNode& node = nodes[list[k]];
    if (node.state != 1u) return;
    const uint32_t first = node.first, n = node.count, L = B.L;
    const uint32_t chunk = blockIdx.x - B.chunkBase;
    const uint32_t lo = chunk * CRT_BVH_CHUNK, hi = (lo + CRT_BVH_CHUNK) < n ? (lo + CRT_BVH_CHUNK) : n;
    const int axis = node.axis; const float splitPos = node.splitPos;
    uint32_t* rk = rank + (first - poolFirst); uint32_t* hl = holes + (first - poolFirst); uint32_t* bl = backL + (first - poolFirst);
    uint32_t gb = 0, ga = 0, bb = 0;                           // holes in earlier chunks / in all chunks; back-order left-class elements in later chunks
    for (uint32_t j = threadIdx.x; j < B.nChunks; j += blockDim.x) {
        const uint32_t f = chunkFR[B.chunkBase + j];
        ga += f; if (j < chunk) gb += f;
        if (j > chunk) bb += chunkBL[B.chunkBase + j];
    }
    uint32_t G = bvh_block_sum(gb, s_red); const uint32_t Gall = bvh_block_sum(ga, s_red); uint32_t GB = bvh_block_sum(bb, s_red);
    if (chunk == 0 && threadIdx.x == 0) B.G = Gall;
    for (uint32_t base = lo; base < hi; base += blockDim.x) {
        const uint32_t x = base + threadIdx.x;
        const bool isR = x < hi && x < L && !(bvh_centroid(src, (size_t)first + x, axis) < splitPos);
        uint32_t tot; const uint32_t r = bvh_block_scan(isR ? 1u : 0u, s_wave, tot);
        if (isR) { rk[x] = G + r; hl[G + r] = x; }
        G += tot;
    }
    for (uint32_t base = 0; base < hi - lo; base += blockDim.x) {   // back order: x descending
        const uint32_t o = base + threadIdx.x;
        const uint32_t x = hi - 1 - o;                               // only meaningful while o < hi - lo
        const bool isL = o < hi - lo && x >= L && (bvh_centroid(src, (size_t)first + x, axis) < splitPos);
        uint32_t tot; const uint32_t r = bvh_block_scan(isL ? 1u : 0u, s_wave, tot);
        if (isL) { rk[x] = GB + r; bl[GB + r] = n - 1 - x; }
        GB += tot;
    }
}

// The move itself (closed form), then -- by the node's first chunk -- the children: two node records, their list slots, and for a BIG
// child its chunks. A partition that left one side empty (BVH.cpp:194) is only flagged here (state 3): its triangles are copied
// back by crt_bvh_big_degenerate, which the host launches when the level's control word says there was one.
__global__ void __launch_bounds__(CRT_BVH_BIG_THREADS) crt_bvh_big_scatter(CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ list, const CrtBigScratch* __restrict__ big,
                                                                           const uint32_t* __restrict__ chunkNode, const CrtTri* __restrict__ src, CrtTri* __restrict__ dst, uint32_t poolFirst,
                                                                           const uint32_t* __restrict__ rank, const uint32_t* __restrict__ holes, const uint32_t* __restrict__ backL,
                                                                           uint32_t levelEnd, CrtBuildCtl* __restrict__ ctl, CrtBuildLists next,
                                                                           CrtBigScratch* __restrict__ nextBig, uint32_t* __restrict__ nextChunkNode)
{
    __shared__ uint32_t s_fill[2][3];                          // per BIG child: slot, chunkBase, nChunks
    const uint32_t k = chunkNode[blockIdx.x];
    const CrtBigScratch& B = big[k];
    CrtBuildNode& node = nodes[list[k]];
    if (node.state != 1u) return;
    const uint32_t first = node.first, n = node.count, L = B.L, G = B.G;
    const uint32_t chunk = blockIdx.x - B.chunkBase;
    const uint32_t lo = chunk * CRT_BVH_CHUNK, hi = (lo + CRT_BVH_CHUNK) < n ? (lo + CRT_BVH_CHUNK) : n;
    const int axis = node.axis; const float splitPos = node.splitPos;
    const uint32_t* rk = rank + (first - poolFirst); const uint32_t* hl = holes + (first - poolFirst); const uint32_t* bl = backL + (first - poolFirst);
    for (uint32_t x = lo + threadIdx.x; x < hi; x += blockDim.x) {
        const bool isLeft = bvh_centroid(src, (size_t)first + x, axis) < splitPos;
        uint32_t dest;
        if (x < L) {
            if (isLeft) dest = x;
            else { const uint32_t m = rk[x]; const uint32_t slot = m == 0 ? 0u : bl[m - 1] + 1u; dest = n - 1 - slot; }
        } else if (isLeft) dest = hl[rk[x]];
        else if (x == L) { const uint32_t slot = G == 0 ? 0u : bl[G - 1] + 1u; dest = n - 1 - slot; }
        else dest = x - 1;
        bvh_copy_tri(dst, (size_t)first + dest, src, (size_t)first + x);
    }
    if (chunk != 0) return;
    if (L == 0 || L == n) {
        if (threadIdx.x == 0) { atomicAdd(&ctl->degenerate, 1u); }
        return;                                                // node.state is set by crt_bvh_big_degenerate (other chunks still read it here)
    }
    if (threadIdx.x == 0) {
        const int cl = bvh_class(L), cr = bvh_class(n - L);
        const unsigned long long before = atomicAdd(&ctl->packed, bvh_pack_one(cl) + bvh_pack_one(cr));
        bvh_new_children(nodes, node, first, L, n, levelEnd, before, next);
        const uint32_t slot[2] = { bvh_unpack(before, cl), bvh_unpack(before + bvh_pack_one(cl), cr) };
        const uint32_t cnt[2] = { L, n - L }; const int cls[2] = { cl, cr };
        for (int j = 0; j < 2; ++j) {
            s_fill[j][2] = 0;
            if (cls[j] != CRT_BVH_CLASS_BIG) continue;
            const uint32_t nch = bvh_chunks(cnt[j]);
            const uint32_t base = atomicAdd(&ctl->nextChunks, nch);
            nextBig[slot[j]].chunkBase = base; nextBig[slot[j]].nChunks = nch;
            s_fill[j][0] = slot[j]; s_fill[j][1] = base; s_fill[j][2] = nch;
        }
    }
    __syncthreads();
    for (int j = 0; j < 2; ++j)
        for (uint32_t c = threadIdx.x; c < s_fill[j][2]; c += blockDim.x) nextChunkNode[s_fill[j][1] + c] = s_fill[j][0];
}

// BVH.cpp:194 for BIG nodes: the node stays a leaf, its triangles stay permuted -> the partition's output goes to both buffers
__global__ void __launch_bounds__(CRT_BVH_BIG_THREADS) crt_bvh_big_degenerate(CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ list, const CrtBigScratch* __restrict__ big,
                                                                              const uint32_t* __restrict__ chunkNode, CrtTri* __restrict__ src, const CrtTri* __restrict__ dst)
{
    const uint32_t k = chunkNode[blockIdx.x];
    const CrtBigScratch& B = big[k];
    CrtBuildNode& node = nodes[list[k]];
    const uint32_t first = node.first, n = node.count;
    if (node.left != CRT_BVH_NONE || (node.state != 1u && node.state != 3u) || !(B.L == 0 || B.L == n)) return;
    const uint32_t chunk = blockIdx.x - B.chunkBase;
    const uint32_t lo = chunk * CRT_BVH_CHUNK, hi = (lo + CRT_BVH_CHUNK) < n ? (lo + CRT_BVH_CHUNK) : n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) bvh_copy_tri(src, (size_t)first + i, dst, (size_t)first + i);
}
__global__ void crt_bvh_big_degenerate_mark(CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ list, const CrtBigScratch* __restrict__ big, uint32_t count)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    CrtBuildNode& node = nodes[list[k]];
    if (node.left == CRT_BVH_NONE && node.state == 1u && (big[k].L == 0 || big[k].L == node.count)) node.state = 3u;
}

// ---- MID nodes (9 .. CRT_BVH_SMALL triangles), one WAVE per node and CRT_BVH_WAVES nodes per workgroup ----
// One workgroup per node (r2) made the deep levels dispatch- and atomic-bound: 45,000 one-wave workgroups per launch, three
// launches per level, and one same-address returning atomic per node (~6.7 ns each: 300 us of a 550 us partition launch).
// Here a wave does UpdateNodeBounds / FindBestSplitPlane + the partition of its node with wave-level reductions only, the
// 21 candidate planes are evaluated by 21 lanes instead of one thread, and the workgroup's waves share ONE atomic.
#define CRT_BVH_WAVES 8
#define CRT_BVH_LDS_TABLE 128         // partition tables of nodes up to this size live in LDS, larger ones in the global scratch
// UpdateNodeBounds (BVH.cpp:54-74) for the nodes list[0 .. count): min/max reductions per wave.
// (round 5: called at the top of crt_bvh_mid -- the wave reads its node's triangles for the bins anyway; as a kernel of its own it was 12
// launches of a 1 M-triangle build. All 64 lanes of the node's wave must call.)
__device__ __forceinline__ void bvh_bounds_wave_body(CrtBuildNode& node, const uint32_t first, const uint32_t n, const uint32_t lane, const CrtTri* __restrict__ tris)
{
    float mn[3] = { 1e30f, 1e30f, 1e30f }, mx[3] = { -1e30f, -1e30f, -1e30f };
    for (uint32_t i = lane; i < n; i += 64) {
        const float* t = bvh_tri_f(tris, (size_t)first + i);
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float x = t[4 * v + c]; mn[c] = mn[c] < x ? mn[c] : x; mx[c] = mx[c] > x ? mx[c] : x; }
    }
    float rmin[3], rmax[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { rmin[c] = bvh_unordered(bvh_wave_min(bvh_ordered(mn[c]))); rmax[c] = bvh_unordered(bvh_wave_max(bvh_ordered(mx[c]))); }
    bool anyZero = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) anyZero = anyZero || rmin[c] == 0.0f || rmax[c] == 0.0f;
    // sign of a zero bound: the sequential fold `acc < x ? acc : x` keeps the LAST zero it meets (tris in order, v0 v1 v2)
    if (anyZero) {
        int last[6] = { -1, -1, -1, -1, -1, -1 };
        for (uint32_t i = lane; i < n; i += 64) {
            const float* t = bvh_tri_f(tris, (size_t)first + i);
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (t[4 * v + c] == 0.0f) {
                        const int s = (int)(i * 3 + v);
                        if (rmin[c] == 0.0f) last[c] = last[c] > s ? last[c] : s;
                        if (rmax[c] == 0.0f) last[3 + c] = last[3 + c] > s ? last[3 + c] : s;
                    }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int a = bvh_wave_max_i(last[c]), b = bvh_wave_max_i(last[3 + c]);
            if (rmin[c] == 0.0f && a >= 0) rmin[c] = bvh_tri_f(tris, (size_t)first + a / 3)[4 * (a % 3) + c];
            if (rmax[c] == 0.0f && b >= 0) rmax[c] = bvh_tri_f(tris, (size_t)first + b / 3)[4 * (b % 3) + c];
        }
    }
    if (lane == 0)
        for (int c = 0; c < 3; ++c) { node.bmin[c] = rmin[c]; node.bmax[c] = rmax[c]; }
}

// UpdateNodeBounds + FindBestSplitPlane + the split decision + the partition (BVH.cpp:54-74, 103-163, 165-216) for the nodes list[0 .. count).
__global__ void __launch_bounds__(64 * CRT_BVH_WAVES) crt_bvh_mid(CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ list, uint32_t count,
                                                                  CrtTri* __restrict__ src, CrtTri* __restrict__ dst, uint32_t poolFirst,
                                                                  uint32_t* __restrict__ rank, uint32_t* __restrict__ holes, uint32_t* __restrict__ backL,
                                                                  uint32_t levelEnd, unsigned long long* __restrict__ packed, CrtBuildLists next)
{
    __shared__ uint32_t s_cnt[CRT_BVH_WAVES][3][CRT_BVH_BINS];
    __shared__ uint32_t s_bmin[CRT_BVH_WAVES][3][CRT_BVH_BINS][3], s_bmax[CRT_BVH_WAVES][3][CRT_BVH_BINS][3];
    __shared__ uint32_t s_tab[CRT_BVH_WAVES][3][CRT_BVH_LDS_TABLE];
    __shared__ unsigned long long s_inc[CRT_BVH_WAVES];
    __shared__ unsigned long long s_base;
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t k = blockIdx.x * CRT_BVH_WAVES + w;
    const bool live = k < count;                              // wave-uniform
    CrtBuildNode& node = nodes[list[live ? k : 0]];
    const uint32_t first = node.first, n = node.count;
    unsigned long long inc = 0; uint32_t L = 0;
    if (live) {
        bvh_bounds_wave_body(node, first, n, lane, src);       // lane 0 stores the bounds; the wave syncs below come before anybody reads them
        for (int j = lane; j < 3 * CRT_BVH_BINS; j += 64) (&s_cnt[w][0][0])[j] = 0;
        for (int j = lane; j < 9 * CRT_BVH_BINS; j += 64) { (&s_bmin[w][0][0][0])[j] = bvh_ordered(1e30f); (&s_bmax[w][0][0][0])[j] = bvh_ordered(-1e30f); }
        float cmin[3], cmax[3];
        {
            float mn[3] = { 1e30f, 1e30f, 1e30f }, mx[3] = { -1e30f, -1e30f, -1e30f };
            for (uint32_t i = lane; i < n; i += 64)
#pragma unroll
                for (int a = 0; a < 3; ++a) { const float v = bvh_centroid(src, (size_t)first + i, a); mn[a] = mn[a] < v ? mn[a] : v; mx[a] = mx[a] > v ? mx[a] : v; }
#pragma unroll
            for (int a = 0; a < 3; ++a) { cmin[a] = bvh_unordered(bvh_wave_min(bvh_ordered(mn[a]))); cmax[a] = bvh_unordered(bvh_wave_max(bvh_ordered(mx[a]))); }
        }
        bvh_wave_sync();
        for (uint32_t i = lane; i < n; i += 64) {
            const float* t = bvh_tri_f(src, (size_t)first + i);
            float tmn[3], tmx[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float lo = 1e30f, hi = -1e30f;
#pragma unroll
                for (int v = 0; v < 3; ++v) { const float x = t[4 * v + c]; lo = lo < x ? lo : x; hi = hi > x ? hi : x; }
                tmn[c] = lo; tmx[c] = hi;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (cmax[a] == cmin[a]) continue;
                const float scale = (float)CRT_BVH_BINS / (cmax[a] - cmin[a]);
                int b = f2i((t[3 + 4 * a] - cmin[a]) * scale);
                b = (CRT_BVH_BINS - 1) < b ? (CRT_BVH_BINS - 1) : b;
                if (b < 0) b = 0;
                atomicAdd(&s_cnt[w][a][b], 1u);
#pragma unroll
                for (int c = 0; c < 3; ++c) { atomicMin(&s_bmin[w][a][b][c], bvh_ordered(tmn[c])); atomicMax(&s_bmax[w][a][b][c], bvh_ordered(tmx[c])); }
            }
        }
        bvh_wave_sync();
        int axis; float splitPos, bestCost;
        bvh_sweep_lanes(&s_cnt[w][0][0], &s_bmin[w][0][0][0], &s_bmax[w][0][0][0], cmin, cmax, lane, axis, splitPos, bestCost);
        const float nosplitCost = (float)n * bvh_area(node.bmin, node.bmax);
        const bool isLeaf = bestCost >= nosplitCost;
        if (lane == 0) { node.axis = axis; node.splitPos = splitPos; node.state = isLeaf ? 2u : 1u; }
        if (isLeaf) {                                          // same order in both buffers
            for (uint32_t i = lane; i < n; i += 64) bvh_copy_tri(dst, (size_t)first + i, src, (size_t)first + i);
        } else {
            // the partition in its closed form (see the header), ballots instead of workgroup scans
            const bool inLds = n <= CRT_BVH_LDS_TABLE;
            uint32_t* rk = inLds ? s_tab[w][0] : rank + (first - poolFirst);
            uint32_t* hl = inLds ? s_tab[w][1] : holes + (first - poolFirst);
            uint32_t* bl = inLds ? s_tab[w][2] : backL + (first - poolFirst);
            const unsigned long long below = (1ull << lane) - 1ull;
            for (uint32_t base = 0; base < n; base += 64) {
                const uint32_t x = base + lane;
                L += (uint32_t)__popcll(__ballot(x < n && bvh_centroid(src, (size_t)first + x, axis) < splitPos));
            }
            uint32_t G = 0;
            for (uint32_t base = 0; base < L; base += 64) {
                const uint32_t x = base + lane;
                const bool isR = x < L && !(bvh_centroid(src, (size_t)first + x, axis) < splitPos);
                const unsigned long long m = __ballot(isR);
                const uint32_t r = G + (uint32_t)__popcll(m & below);
                if (isR) { rk[x] = r; hl[r] = x; }
                G += (uint32_t)__popcll(m);
            }
            uint32_t GB = 0;
            const uint32_t nb = n - L;
            for (uint32_t base = 0; base < nb; base += 64) {
                const uint32_t y = base + lane;
                const bool isL = y < nb && (bvh_centroid(src, (size_t)first + (n - 1 - y), axis) < splitPos);
                const unsigned long long m = __ballot(isL);
                const uint32_t r = GB + (uint32_t)__popcll(m & below);
                if (isL) { rk[n - 1 - y] = r; bl[r] = y; }
                GB += (uint32_t)__popcll(m);
            }
            bvh_wave_sync();                                  // tables complete (G == GB by counting)
            for (uint32_t x = lane; x < n; x += 64) {
                const bool isLeft = bvh_centroid(src, (size_t)first + x, axis) < splitPos;
                uint32_t dest;
                if (x < L) {
                    if (isLeft) dest = x;
                    else { const uint32_t m = rk[x]; const uint32_t slot = m == 0 ? 0u : bl[m - 1] + 1u; dest = n - 1 - slot; }
                } else if (isLeft) dest = hl[rk[x]];
                else if (x == L) { const uint32_t slot = G == 0 ? 0u : bl[G - 1] + 1u; dest = n - 1 - slot; }
                else dest = x - 1;
                bvh_copy_tri(dst, (size_t)first + dest, src, (size_t)first + x);
            }
            if (L == 0 || L == n) {                           // BVH.cpp:194: stays a leaf, triangles stay permuted -> both buffers
                bvh_wave_sync();
                for (uint32_t i = lane; i < n; i += 64) bvh_copy_tri(src, (size_t)first + i, dst, (size_t)first + i);
                if (lane == 0) node.state = 3u;
            } else inc = bvh_pack_one(bvh_class(L)) + bvh_pack_one(bvh_class(n - L));
        }
    }
    // one atomic per workgroup for the children of its CRT_BVH_WAVES nodes
    if (lane == 0) s_inc[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long total = 0;
        for (int j = 0; j < CRT_BVH_WAVES; ++j) total += s_inc[j];
        s_base = total ? atomicAdd(packed, total) : 0ull;
    }
    __syncthreads();
    if (inc != 0 && lane == 0) {
        unsigned long long before = s_base;
        for (uint32_t j = 0; j < w; ++j) before += s_inc[j];
        bvh_new_children(nodes, node, first, L, n, levelEnd, before, next);
    }
}

// UpdateNodeBounds for TINY nodes: one thread per node, the sequential fold itself. (round 5: called at the top of crt_bvh_tiny; as a
// kernel of its own it was 14 launches of a 1 M-triangle build)
__device__ __forceinline__ void bvh_bounds_tiny_body(CrtBuildNode& node, const CrtTri* __restrict__ tris)
{
    float mn[3] = { 1e30f, 1e30f, 1e30f }, mx[3] = { -1e30f, -1e30f, -1e30f };
    for (uint32_t i = 0; i < node.count; ++i) {
        const float* t = bvh_tri_f(tris, (size_t)node.first + i);
        for (int c = 0; c < 3; ++c)
            for (int v = 0; v < 3; ++v) { const float x = t[4 * v + c]; mn[c] = mn[c] < x ? mn[c] : x; mx[c] = mx[c] > x ? mx[c] : x; }
    }
    for (int c = 0; c < 3; ++c) { node.bmin[c] = mn[c]; node.bmax[c] = mx[c]; }
}

// One level of SubdivideBVH (BVH.cpp:165-216) for TINY nodes, one thread per node, everything in registers: the
// triangles' boxes and centroids are loaded once (loops over the at most CRT_BVH_TINY triangles are fully unrolled so
// the small arrays never reach scratch). Per candidate plane the left/right boxes are the min/max over the triangles
// binned left/right of it -- the same values upstream gets by growing bin boxes and merging them in sweep order
// (min/max are exact; an empty side keeps the (1e30, -1e30) box and its NaN cost, as upstream). The partition loop
// runs literally on a nibble-packed index permutation.
template <int N>
__device__ __forceinline__ void bvh_tiny_body(CrtBuildNode* __restrict__ nodes, CrtBuildNode& node, const bool live, const uint32_t first, const uint32_t n,
                                              CrtTri* __restrict__ src, CrtTri* __restrict__ dst, uint32_t levelEnd, unsigned long long* __restrict__ packed, const CrtBuildLists& next)
{
    // (the centroids are re-read where they are used -- per axis for the bins, once more for the partition: kept in registers next to the
    // 48 box floats they pushed the 8-triangle instantiation to 103 VGPRs + 160 B of scratch)
    float tmn[N][3], tmx[N][3];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if ((uint32_t)i < n) {
            const float* t = bvh_tri_f(src, (size_t)first + i);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float lo = 1e30f, hi = -1e30f;
#pragma unroll
                for (int v = 0; v < 3; ++v) { const float x = t[4 * v + c]; lo = lo < x ? lo : x; hi = hi > x ? hi : x; }
                tmn[i][c] = lo; tmx[i][c] = hi;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) { tmn[i][c] = 1e30f; tmx[i][c] = -1e30f; }
        }
    }
    float bestCost = 1e30f, splitPos = 0.0f; int bestAxis = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float boundsMin = 1e30f, boundsMax = -1e30f;
        float cen[N];                                          // this axis' centroids
#pragma unroll
        for (int i = 0; i < N; ++i) cen[i] = (uint32_t)i < n ? bvh_centroid(src, (size_t)first + i, a) : 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i)
            if ((uint32_t)i < n) { const float v = cen[i]; boundsMin = boundsMin < v ? boundsMin : v; boundsMax = boundsMax > v ? boundsMax : v; }
        if (boundsMax == boundsMin) continue;
        const float scale = (float)CRT_BVH_BINS / (boundsMax - boundsMin);
        int bin[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            int b = f2i((cen[i] - boundsMin) * scale);
            b = (CRT_BVH_BINS - 1) < b ? (CRT_BVH_BINS - 1) : b;
            if (b < 0) b = 0;
            bin[i] = (uint32_t)i < n ? b : CRT_BVH_BINS;          // absent triangles are on neither side
        }
        const float pscale = (boundsMax - boundsMin) / (float)CRT_BVH_BINS;
#pragma unroll
        for (int p = 0; p < CRT_BVH_BINS - 1; ++p) {
            int leftCount = 0, rightCount = 0;
            float lmn[3] = { 1e30f, 1e30f, 1e30f }, lmx[3] = { -1e30f, -1e30f, -1e30f };
            float rmn[3] = { 1e30f, 1e30f, 1e30f }, rmx[3] = { -1e30f, -1e30f, -1e30f };
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool isL = bin[i] <= p, isR = bin[i] > p && bin[i] < CRT_BVH_BINS;
                leftCount += isL ? 1 : 0; rightCount += isR ? 1 : 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if (isL) { lmn[c] = lmn[c] < tmn[i][c] ? lmn[c] : tmn[i][c]; lmx[c] = lmx[c] > tmx[i][c] ? lmx[c] : tmx[i][c]; }
                    if (isR) { rmn[c] = rmn[c] < tmn[i][c] ? rmn[c] : tmn[i][c]; rmx[c] = rmx[c] > tmx[i][c] ? rmx[c] : tmx[i][c]; }
                }
            }
            const float planeCost = (float)leftCount * bvh_area(lmn, lmx) + (float)rightCount * bvh_area(rmn, rmx);
            if (planeCost < bestCost) { splitPos = boundsMin + pscale * (float)(p + 1); bestAxis = a; bestCost = planeCost; }
        }
    }
    const float nosplitCost = (float)n * bvh_area(node.bmin, node.bmax);
    const bool isLeaf = bestCost >= nosplitCost;
    if (live) { node.axis = bestAxis; node.splitPos = splitPos; }
    if (live && isLeaf) {                                      // leaf: same order in both buffers
        node.state = 2u;
        for (uint32_t i = 0; i < n; ++i) bvh_copy_tri(dst, (size_t)first + i, src, (size_t)first + i);
    }
    // the partition loop (BVH.cpp:185-192) on an index permutation packed four bits per entry, then one copy per triangle
    uint32_t left = 0;                                         // bit i: centroid of triangle i is left of the plane
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if ((uint32_t)i < n && bvh_centroid(src, (size_t)first + i, bestAxis) < splitPos) left |= 1u << i;
    }
    static_assert(N <= 16 && N <= CRT_BVH_TINY, "the permutation is packed four bits per entry into 64 bits");
    unsigned long long perm = 0xFEDCBA9876543210ull;
    int i = 0, j = (live && !isLeaf) ? (int)n - 1 : -1;
    while (i <= j) {
        const unsigned long long pi = (perm >> (4 * i)) & 15ull;
        if ((left >> pi) & 1u) i++;
        else {
            const unsigned long long pj = (perm >> (4 * j)) & 15ull;
            perm = (perm & ~((15ull << (4 * i)) | (15ull << (4 * j)))) | (pj << (4 * i)) | (pi << (4 * j));
            j--;
        }
    }
    const uint32_t L = (uint32_t)i;
    const bool tried = live && !isLeaf, split = tried && L != 0 && L != n;
    if (tried) {
        for (uint32_t x = 0; x < n; ++x) bvh_copy_tri(dst, (size_t)first + x, src, (size_t)first + (size_t)((perm >> (4 * x)) & 15ull));
        if (!split) {                                          // BVH.cpp:194: stays a leaf, triangles stay permuted -> both buffers
            node.state = 3u;
            for (uint32_t x = 0; x < n; ++x) bvh_copy_tri(src, (size_t)first + x, dst, (size_t)first + x);
        } else node.state = 1u;
    }
    // one atomic per wave: both children of a TINY node are TINY
    const unsigned long long m = __ballot(split);
    if (m == 0) return;
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (lane == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(packed, CRT_BVH_PACK_TINY(2u * (uint32_t)__popcll(m)));
    base = (unsigned long long)__shfl((long long)base, __ffsll((long long)m) - 1, 64);
    if (split) bvh_new_children(nodes, node, first, L, n, levelEnd, base + CRT_BVH_PACK_TINY(2u * (uint32_t)__popcll(m & ((1ull << lane) - 1ull))), next);
}

// The fully unrolled body costs the same for 2 triangles as for 8, so a wave whose largest node has at most 4 (most waves of the
// deepest levels) runs the 4-triangle instantiation: half the instructions.
__global__ void crt_bvh_tiny(CrtBuildNode* __restrict__ nodes, const uint32_t* __restrict__ list, uint32_t count, CrtTri* __restrict__ src, CrtTri* __restrict__ dst,
                             uint32_t levelEnd, unsigned long long* __restrict__ packed, CrtBuildLists next)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = k < count;
    CrtBuildNode& node = nodes[list[live ? k : 0]];           // lanes past the end idle through the code (n = 0) so the wave stays converged
    const uint32_t first = node.first, n = live ? node.count : 0u;
    if (live) bvh_bounds_tiny_body(node, src);
    if (bvh_wave_max(n) <= 4u) bvh_tiny_body<4>(nodes, node, live, first, n, src, dst, levelEnd, packed, next);
    else bvh_tiny_body<CRT_BVH_TINY>(nodes, node, live, first, n, src, dst, levelEnd, packed, next);
}

// ---- numbering and emission (closed form, see the header) ----
__global__ void crt_bvh_leaf_flags(const CrtBuildNode* __restrict__ nodes, uint32_t count, uint32_t poolFirst, uint32_t* __restrict__ flags)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    if (nodes[k].left == CRT_BVH_NONE) flags[nodes[k].first - poolFirst] = 1u;
}

// exclusive prefix sum of n words into S[0 .. n] (S[n] = total): per-workgroup sums, one workgroup over the sums, then the scan proper.
// Launch crt_bvh_scan_sums and crt_bvh_scan_apply with n / CRT_BVH_SCAN_ITEMS + 1 workgroups (position n must be covered).
#define CRT_BVH_SCAN_THREADS 256
#define CRT_BVH_SCAN_PER_THREAD 16
#define CRT_BVH_SCAN_ITEMS (CRT_BVH_SCAN_THREADS * CRT_BVH_SCAN_PER_THREAD)
__global__ void __launch_bounds__(CRT_BVH_SCAN_THREADS) crt_bvh_scan_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t s_red[CRT_BVH_SCAN_THREADS / 64];
    const uint32_t base = blockIdx.x * CRT_BVH_SCAN_ITEMS;
    uint32_t v = 0;
    for (uint32_t i = threadIdx.x; i < CRT_BVH_SCAN_ITEMS; i += blockDim.x) if (base + i < n) v += in[base + i];
    v = bvh_block_sum(v, s_red);
    if (threadIdx.x == 0) sums[blockIdx.x] = v;
}
__global__ void __launch_bounds__(CRT_BVH_SCAN_THREADS) crt_bvh_scan_blocks(uint32_t* __restrict__ sums, uint32_t nb)
{
    __shared__ uint32_t s_part[CRT_BVH_SCAN_THREADS];
    const uint32_t per = (nb + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = threadIdx.x * per, hi = (lo + per) < nb ? (lo + per) : nb;
    uint32_t v = 0;
    for (uint32_t i = lo; i < hi; ++i) v += sums[i];
    s_part[threadIdx.x] = v;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t t = 0; t < threadIdx.x; ++t) before += s_part[t];
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t x = sums[i]; sums[i] = before; before += x; }
}
__global__ void __launch_bounds__(CRT_BVH_SCAN_THREADS) crt_bvh_scan_apply(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ sums, uint32_t* __restrict__ S)
{
    __shared__ uint32_t s_part[CRT_BVH_SCAN_THREADS];
    const uint32_t base = blockIdx.x * CRT_BVH_SCAN_ITEMS + threadIdx.x * CRT_BVH_SCAN_PER_THREAD;
    uint32_t x[CRT_BVH_SCAN_PER_THREAD], v = 0;
#pragma unroll
    for (int i = 0; i < CRT_BVH_SCAN_PER_THREAD; ++i) { x[i] = (base + i < n) ? in[base + i] : 0u; v += x[i]; }
    s_part[threadIdx.x] = v;
    __syncthreads();
    uint32_t before = sums[blockIdx.x];
    for (uint32_t t = 0; t < threadIdx.x; ++t) before += s_part[t];
#pragma unroll
    for (int i = 0; i < CRT_BVH_SCAN_PER_THREAD; ++i) { if (base + i < n) S[base + i] = before; before += x[i]; }
    if (base <= n && n < base + CRT_BVH_SCAN_PER_THREAD) {     // the thread whose range ends at or contains n writes the total
        uint32_t t = sums[blockIdx.x];
        for (uint32_t u = 0; u < threadIdx.x; ++u) t += s_part[u];
        for (uint32_t i = 0; base + i < n; ++i) t += x[i];
        S[n] = t;
    }
}

// every build node gets its final index and is written out as a reference-layout BVHNode; roots[m] and the node count too
__global__ void crt_bvh_emit(const CrtBuildNode* __restrict__ nodes, uint32_t count, int numMeshes, const uint32_t* __restrict__ S, uint32_t poolFirst, uint32_t poolCount,
                             uint32_t firstNode, CrtBVHNode* __restrict__ out, uint32_t* __restrict__ roots, uint32_t* __restrict__ nodesUsed, uint32_t* __restrict__ outOfRange)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const CrtBuildNode& n = nodes[k];
    const uint32_t rootFirst = nodes[n.mesh].first;           // the roots are build nodes 0 .. numMeshes
    const uint32_t rootIndex = firstNode + 2u * S[rootFirst - poolFirst] - n.mesh;
    const uint32_t base = rootIndex + 1u + 2u * n.depth + 2u * (S[n.first - poolFirst] - S[rootFirst - poolFirst]) - 2u * n.rightTurns;
    uint32_t index;
    if (n.depth == 0) index = rootIndex;
    else if (!n.isRight) index = base - 2u;
    else index = base + 1u - 2u * (S[n.first - poolFirst] - S[nodes[k - 1].first - poolFirst]);
    CrtBVHNode o;
    for (int c = 0; c < 3; ++c) { o.aabbMin[c] = n.bmin[c]; o.aabbMax[c] = n.bmax[c]; }
    if (n.left == CRT_BVH_NONE) { o.leftFirst = n.first; o.triCount = n.count; }
    else { o.leftFirst = base; o.triCount = 0; }
    // the closed form is trusted only as far as the node array it was sized for: a number outside [firstNode, firstNode + count) would mean
    // a damaged tree, and is reported instead of stored (the host then refuses the build)
    if (index - firstNode >= count) { atomicOr(outOfRange, 1u); return; }
    out[index] = o;
    if (n.depth == 0) roots[n.mesh] = rootIndex;
    if (k == 0) *nodesUsed = 2u * S[poolCount] - (uint32_t)numMeshes;
}
