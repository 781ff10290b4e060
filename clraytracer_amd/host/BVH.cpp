// BVH.cpp -- host BVH2 builder: 8-bin SAH over centroids, one tree per mesh, sibling nodes
// allocated in adjacent pairs, triangles partitioned in place (reference: BVH.cpp:9-255).
//
// Traversal results depend on the exact tree (SURVEY.md hazards H1/H2), so the arithmetic below
// keeps the reference's operation order lane for lane:
//   - "area" is (ex*ex + ey*ex) + ez*ez   (BVH.cpp:41-46 through hsum_ps_sse3, SIMDCommon.hpp:183-189)
//   - min/max are SSE-style `a < b ? a : b` / `a > b ? a : b`
//   - centroid = ((a + b) + c) * 0.333333f (BVH.cpp:232-234)
// Structure differs from the reference: recursion is replaced by an explicit work stack (no
// unbounded call depth) and meshes are built concurrently into private node arrays that are then
// spliced at the offsets a sequential build would have produced (node numbering is a pure
// function of per-mesh node counts, BVH.cpp:237-252).
#include "ResourceManager.hpp"
#include <thread>
#include <vector>

namespace {

uint g_totalNodesUsed = 0; // BVH.cpp:49
size_t g_nodeCapacity = 0;  // 0 = unchecked (the reference never checks; its 1.2 M-node arena overflows near 0.65 M triangles)
bool g_overflowed = false;

inline float lane_min(float a, float b) { return a < b ? a : b; }
inline float lane_max(float a, float b) { return a > b ? a : b; }

struct Box {
    float lo[3], hi[3];
    Box() { for (int c = 0; c < 3; ++c) { lo[c] = 1e30f; hi[c] = -1e30f; } }
    void grow(const Tri& t)
    {
        for (int c = 0; c < 3; ++c) {
            lo[c] = lane_min(lane_min(lane_min(lo[c], t.v0[c]), t.v1[c]), t.v2[c]);
            hi[c] = lane_max(lane_max(lane_max(hi[c], t.v0[c]), t.v1[c]), t.v2[c]);
        }
    }
    void grow(const Box& o) // BVH.cpp:30-39
    {
        if (o.lo[0] == 1e30f) return;
        for (int c = 0; c < 3; ++c) {
            lo[c] = lane_min(lo[c], o.lo[c]);
            hi[c] = lane_max(hi[c], o.lo[c]);
            lo[c] = lane_min(lo[c], o.hi[c]);
            hi[c] = lane_max(hi[c], o.hi[c]);
        }
    }
};

inline float half_area(const float lo[3], const float hi[3])
{
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return (ex * ex + ey * ex) + (ez * ez + 0.0f);
}

inline float centroid_of(const Tri& t, int axis) { return axis == 0 ? t.centroidx : (axis == 1 ? t.centroidy : t.centroidz); }

void fit_bounds(BVHNode& node, const Tri* tris) // BVH.cpp:54-74
{
    float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
    const Tri* t = tris + node.leftFirst;
    for (uint i = 0; i < node.triCount; ++i, ++t)
        for (int c = 0; c < 3; ++c) {
            lo[c] = lane_min(lane_min(lane_min(lo[c], t->v0[c]), t->v1[c]), t->v2[c]);
            hi[c] = lane_max(lane_max(lane_max(hi[c], t->v0[c]), t->v1[c]), t->v2[c]);
        }
    for (int c = 0; c < 3; ++c) { node.aabbMin[c] = lo[c]; node.aabbMax[c] = hi[c]; }
}

constexpr int kBins = 8;

// BVH.cpp:103-163
float best_split(const BVHNode& node, const Tri* tris, int& axisOut, float& posOut)
{
    float best = 1e30f;
    const Tri* first = tris + node.leftFirst;
    const uint n = node.triCount;
    for (int axis = 0; axis < 3; ++axis) {
        float cmin = 1e30f, cmax = -1e30f;
        for (uint i = 0; i < n; ++i) {
            const float c = centroid_of(first[i], axis);
            cmin = cmin < c ? cmin : c;
            cmax = cmax > c ? cmax : c;
        }
        if (cmax == cmin) continue;

        Box bounds[kBins];
        uint counts[kBins] = { 0 };
        float scale = (float)kBins / (cmax - cmin);
        for (uint i = 0; i < n; ++i) {
            int b = crtmath::TruncToInt((centroid_of(first[i], axis) - cmin) * scale);
            b = (kBins - 1) < b ? (kBins - 1) : b;
            if (b < 0) b = 0;
            counts[b]++;
            bounds[b].grow(first[i]);
        }

        float areaL[kBins - 1], areaR[kBins - 1];
        int countL[kBins - 1], countR[kBins - 1];
        Box accL, accR;
        int sumL = 0, sumR = 0;
        for (int i = 0; i < kBins - 1; ++i) {
            sumL += (int)counts[i];
            countL[i] = sumL;
            accL.grow(bounds[i]);
            areaL[i] = half_area(accL.lo, accL.hi);
            sumR += (int)counts[kBins - 1 - i];
            countR[kBins - 2 - i] = sumR;
            accR.grow(bounds[kBins - 1 - i]);
            areaR[kBins - 2 - i] = half_area(accR.lo, accR.hi);
        }

        scale = (cmax - cmin) / (float)kBins;
        for (int i = 0; i < kBins - 1; ++i) {
            const float cost = (float)countL[i] * areaL[i] + (float)countR[i] * areaR[i];
            if (cost < best) { posOut = cmin + scale * (float)(i + 1); axisOut = axis; best = cost; }
        }
    }
    return best;
}

// Builds one mesh's tree into `nodes` (numbered from `base`, root == base). Returns nodes used.
// Equivalent to BVH.cpp:241-250 + the recursion of SubdivideBVH (BVH.cpp:165-216): the explicit
// stack visits left subtrees first, so pair allocation order equals the recursive preorder.
uint build_one(Tri* tris, uint firstTri, uint triCount, BVHNode* nodes, uint base)
{
    uint used = base;
    const uint root = used++;
    nodes[root].leftFirst = firstTri;
    nodes[root].triCount = triCount;
    fit_bounds(nodes[root], tris);

    std::vector<uint> todo;
    todo.push_back(root);
    while (!todo.empty()) {
        const uint idx = todo.back();
        todo.pop_back();
        BVHNode& node = nodes[idx];
        const uint leftFirst = node.leftFirst, count = node.triCount;
        int axis = 0; float pos = 0.0f;
        const float splitCost = best_split(node, tris, axis, pos);
        const float leafCost = (float)count * half_area(node.aabbMin, node.aabbMax);
        if (splitCost >= leafCost) continue;

        int i = (int)leftFirst, j = i + (int)count - 1;
        while (i <= j) {
            if (centroid_of(tris[i], axis) < pos) ++i;
            else { Tri tmp = tris[i]; tris[i] = tris[j]; tris[j] = tmp; --j; }
        }
        const int leftCount = i - (int)leftFirst;
        if (leftCount == 0 || leftCount == (int)count) continue;

        const uint l = used++, r = used++;
        nodes[l].leftFirst = leftFirst; nodes[l].triCount = (uint)leftCount;
        nodes[r].leftFirst = (uint)i;   nodes[r].triCount = count - (uint)leftCount;
        node.leftFirst = l; node.triCount = 0;
        fit_bounds(nodes[l], tris);
        fit_bounds(nodes[r], tris);
        todo.push_back(r); // right is processed after the whole left subtree
        todo.push_back(l);
    }
    return used - base;
}

} // namespace

void ResetBVHNodeCounter() { g_totalNodesUsed = 0; g_overflowed = false; }
void AdvanceBVHNodeCounter(uint nodes) { g_totalNodesUsed += nodes; }
void SetBVHNodeCapacity(size_t nodes) { g_nodeCapacity = nodes; }
bool BVHBuildOverflowed() { return g_overflowed; }

// BVH.cpp:218-255. `nodes` is indexed exactly as upstream (absolute indices from this pointer,
// starting at the running counter); returns the number of nodes added by this call.
uint BuildBVH(Tri* tris, MeshInfo* meshes, int numMeshes, BVHNode* nodes, uint* bvhIndices)
{
    size_t numTriangles = 0;
    for (int i = 0; i < numMeshes; ++i) numTriangles += meshes[i].numTriangles;
    for (size_t i = 0; i < numTriangles; ++i) {
        Tri& t = tris[i];
        t.centroidx = ((t.v0[0] + t.v1[0]) + t.v2[0]) * 0.333333f;
        t.centroidy = ((t.v0[1] + t.v1[1]) + t.v2[1]) * 0.333333f;
        t.centroidz = ((t.v0[2] + t.v1[2]) + t.v2[2]) * 0.333333f;
    }

    const uint start = g_totalNodesUsed;
    // every mesh into its own zero-based array, concurrently (triangle ranges are disjoint)
    std::vector<std::vector<BVHNode>> local((size_t)numMeshes);
    std::vector<uint> used((size_t)numMeshes, 0);
    std::vector<uint> firstTri((size_t)numMeshes, 0);
    uint curr = 0;
    for (int i = 0; i < numMeshes; ++i) { firstTri[i] = curr; curr += meshes[i].numTriangles; }

    auto work = [&](int i) {
        const uint n = meshes[i].numTriangles;
        local[i].resize((size_t)(n ? 2 * n : 1) + 1);
        used[i] = build_one(tris, firstTri[i], n, local[i].data(), 0);
    };
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    if (numMeshes <= 1 || hw == 1) {
        for (int i = 0; i < numMeshes; ++i) work(i);
    } else {
        std::vector<std::thread> pool;
        int next = 0;
        while (next < numMeshes) {
            pool.clear();
            for (unsigned t = 0; t < hw && next < numMeshes; ++t, ++next) pool.emplace_back(work, next);
            for (auto& th : pool) th.join();
        }
    }

    // splice: child indices of inner nodes shift by the mesh's base index
    for (int i = 0; i < numMeshes; ++i) {
        const uint base = g_totalNodesUsed;
        if (g_nodeCapacity && (size_t)base + used[i] > g_nodeCapacity) { g_overflowed = true; break; }
        bvhIndices[i] = base;
        for (uint k = 0; k < used[i]; ++k) {
            BVHNode n = local[i][k];
            if (n.triCount == 0) n.leftFirst += base;
            nodes[base + k] = n;
        }
        g_totalNodesUsed += used[i];
    }
    return g_totalNodesUsed - start;
}
