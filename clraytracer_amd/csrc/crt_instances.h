// crt_instances.h -- everything derived from the instance table and the root nodes: device records, bounding spheres and the cull range, the instance tree, per-slot copies
// Part of the one translation unit crt_shim.hip (included there, in this order: crt_state.h, crt_instances.h, crt_upload.h,
// crt_bvh_driver.h, crt_frame.h, crt_multidev.h); everything here has internal linkage.
#pragma once
namespace {

int cache_root_nodes();
void rebuild_instance_master(uint32_t dirtyFirst = 0, uint32_t dirtyCount = CRT_MAX_INSTANCES, bool everything = true);

int rebuild_bvh_layout()
{
    HIPCHK(hipMemsetAsync(g.err, 0, sizeof(int), g.stream));
    if (g.nodeCount) {
        crt_relayout_nodes<<<(g.nodeCount + 255) / 256, 256, 0, g.stream>>>(g.rawNodes, g.nodeCount, (uint32_t)g.triCap, g.pairs, g.bigLeaf, g.err);
        HIPCHK(hipGetLastError());
    }
    crt_make_root_refs<<<(CRT_MAX_MESHES + 255) / 256, 256, 0, g.stream>>>(g.rawNodes, g.nodeCount, (uint32_t)g.triCap, g.roots, g.numRoots, g.rootRefs, g.bigLeaf, g.err);
    HIPCHK(hipGetLastError());
    {   // the tree-top table of CRT_KERNEL=ldstop (crt_ldstop.h): the records split evenly over the meshes that have a root. Built for every
        // session (one 128-thread launch per BVH upload); only the ldstop kernel reads it and the references into it (CrtDevInstance::r2.w)
        const uint32_t perMesh = g.numRoots ? (uint32_t)CRT_TOP_PAIRS / g.numRoots : 0u;
        crt_build_top_kernel<<<1, CRT_MAX_MESHES, 0, g.stream>>>(g.pairs, g.rootRefs, g.numRoots, perMesh, g.topPairs, g.topRootRefs);
        HIPCHK(hipGetLastError());
    }
    int err = 0;
    HIPCHK(hipMemcpyAsync(&err, g.err, sizeof(int), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    g.sceneValid = (err == 0);
    if (err) return CRT_E_BAD_ARGUMENT;
    RCCHK(cache_root_nodes());
    rebuild_instance_master();          // root references and root boxes feed the per-instance records
    return CRT_OK;
}

// World-space bounding spheres for the conservative instance cull (crt_device.h). Runs at upload
// time only. forward = inverse(inverseTransform) in double; sphere = image of the root box's corners.
bool invert4(const double m[16], double out[16])
{
    double a[4][8];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = m[r * 4 + c]; a[r][4 + c] = (r == c) ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r) if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        if (!(fabs(a[piv][col]) > 1e-300)) return false;
        if (piv != col) for (int c = 0; c < 8; ++c) { double t = a[col][c]; a[col][c] = a[piv][c]; a[piv][c] = t; }
        const double inv = 1.0 / a[col][col];
        for (int c = 0; c < 8; ++c) a[col][c] *= inv;
        for (int r = 0; r < 4; ++r) if (r != col) { const double f = a[r][col]; if (f != 0.0) for (int c = 0; c < 8; ++c) a[r][c] -= f * a[col][c]; }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[r * 4 + c] = a[r][4 + c];
    return true;
}

// Root node of every mesh, read back once per BVH upload (everything is quiescent then): instance uploads need the root
// boxes and must not touch the device.
int cache_root_nodes()
{
    for (uint32_t m = 0; m < CRT_MAX_MESHES; ++m) {
        g.hHaveRoot[m] = m < g.numRoots && g.hRoots[m] < g.nodeCount;
        if (g.hHaveRoot[m]) HIPCHK(hipMemcpyAsync(&g.hRootNodes[m], g.rawNodes + g.hRoots[m], sizeof(CrtBVHNode), hipMemcpyDeviceToHost, g.stream));
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    // the two children of every inner root (kernel_main.cl:144-145: leftFirst, leftFirst + 1): the cull's sphere goes around THEIR
    // boxes, which is what the claim "a ray that misses the sphere fails both slab tests" is about -- for a tree from BuildBVH
    // their union is the root box, for an arbitrary upload it need not be
    for (uint32_t m = 0; m < CRT_MAX_MESHES; ++m) {
        g.hHaveKids[m] = g.hHaveRoot[m] && g.hRootNodes[m].triCount == 0 && (unsigned long long)g.hRootNodes[m].leftFirst + 1ull < (unsigned long long)g.nodeCount;
        if (g.hHaveKids[m]) HIPCHK(hipMemcpyAsync(&g.hRootKids[m][0], g.rawNodes + g.hRootNodes[m].leftFirst, 2 * sizeof(CrtBVHNode), hipMemcpyDeviceToHost, g.stream));
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    return CRT_OK;
}

// Host master of the instance-derived tables: bounding spheres, the instance tree, the never-culled list. Pure host work; bumps the version the
// frame slots compare against.
// r6 -- incremental, because an animated scene calls this before every frame (upstream uploads the dirty range per frame, Renderer.cpp:312-320) and a
// synchronous caller pays it serially (401 instances: 37 us of bounds + 23 us of tree before):
//   * `everything` (a BVH / triangle upload: the root boxes or the reach of bounce origins may have changed) recomputes every instance; an instance upload
//     only the records it replaced, [dirtyFirst, dirtyFirst + dirtyCount) -- what the others yield depends on their own record, the root boxes and the reach only;
//   * the instance tree is REFITTED while the set of cullable instances is the one it was built for: node boxes bottom-up over the same partition, which
//     gives bit for bit the spheres a rebuild with that partition would (min / max are exact); a new median-split build when the set changed, when the
//     refitted tree's inner radii have grown by more than a quarter since the build, or after 256 refits. Any partition is a correct tree (a node's sphere
//     holds its instances' spheres, crt_device.h (5)); the topology only decides how many node tests a ray makes.
void rebuild_instance_master(uint32_t dirtyFirst, uint32_t dirtyCount, bool everything)
{
    float4* bounds = g.hBounds;
    const CrtBVHNode* rootNodes = g.hRootNodes;
    const bool* haveRoot = g.hHaveRoot;
    // Bounce, shadow and refraction rays start at object-space hit points of the hit instance used as world-space origins
    // (hazard H6): no farther from the world origin than the farthest corner of any mesh's root (or root children's) box, plus the
    // 0.01 offset along the normal. An instance that cannot be culled exactly for origins that far out is never culled.
    double reach = g.hReach;
    if (everything) {
        reach = 0.0;
        for (uint32_t m = 0; m < CRT_MAX_MESHES; ++m) {
            if (!haveRoot[m]) continue;
            const CrtBVHNode* boxes[3] = { &rootNodes[m], g.hHaveKids[m] ? &g.hRootKids[m][0] : nullptr, g.hHaveKids[m] ? &g.hRootKids[m][1] : nullptr };
            for (const CrtBVHNode* b : boxes) {
                if (!b) continue;
                double far2 = 0.0;
                for (int a = 0; a < 3; ++a) { const double v = fmax(fabs((double)b->aabbMin[a]), fabs((double)b->aabbMax[a])); far2 += v * v; }
                const double far = sqrt(far2) * (1.0 + 1e-5) + 0.02;
                if (far > reach || !(far == far)) reach = far;
            }
        }
        // ... and no farther than the farthest uploaded vertex: trees that arrive through crt_upload_bvh_nodes need not bound their triangles
        { const double far = sqrt(g.triReach2) * (1.0 + 1e-5) + 0.02; if (far > reach || !(far == far)) reach = far; }
        g.hReach = reach;
        g.bounceOriginReach = reach < 3.0e38 ? (float)reach : 3.0e38f;
    }
    const uint32_t iFirst = everything ? 0u : dirtyFirst;
    const uint32_t iEnd = everything ? (uint32_t)CRT_MAX_INSTANCES : (dirtyFirst + dirtyCount < (uint32_t)CRT_MAX_INSTANCES ? dirtyFirst + dirtyCount : (uint32_t)CRT_MAX_INSTANCES);
    // test hook (CRT_DEBUG_HOOKS=1 only): CRT_DEBUG_CULL_RANGE_SCALE=k multiplies every O_i -- tools/fuzz_cull.py uses it to measure how far
    // beyond the proven range the cull stays exact in practice (the derivation is a worst-case bound)
    double rangeScale = 1.0;
    { const char* h = getenv("CRT_DEBUG_HOOKS"); const char* k = getenv("CRT_DEBUG_CULL_RANGE_SCALE"); if (h && atoi(h) != 0 && k && atof(k) > 0.0) rangeScale = atof(k); }
    const double U = 5.9604644775390625e-8, G3 = 3.0 * U / (1.0 - 3.0 * U), G4 = 4.0 * U / (1.0 - 4.0 * U), K = 2.8e-6;
    for (uint32_t i = iFirst; i < iEnd; ++i) {
        bounds[i] = make_float4(0.f, 0.f, 0.f, -1.0f);
        g.hCullOriginLimit[i] = 0.0f;
        if (i >= g.instHigh) continue;
        const CrtMeshInstance& inst = g.hInstances[i];
        if (inst.meshIndex >= CRT_MAX_MESHES || !haveRoot[inst.meshIndex]) continue;
        const CrtBVHNode& root = rootNodes[inst.meshIndex];
        if (root.triCount > 0) continue;      // single-leaf mesh: its triangles are tested without any box test (hazard H3)
        if (!g.hHaveKids[inst.meshIndex]) continue;
        double inv[16], fwd[16];
        for (int k = 0; k < 16; ++k) inv[k] = (double)(&inst.inverseTransform.m[0][0])[k];
        // forward = inverse(inverseTransform). An affine record (fourth column 0 0 0 1: every matrix InverseTransform / PositionRotationScale produce)
        // inverts as a 3x3 by cofactors + the translation row -- a third of the general elimination's time, and this runs per instance and upload (r6)
        if (inv[3] == 0.0 && inv[7] == 0.0 && inv[11] == 0.0 && inv[15] == 1.0) {
            const double a = inv[0], b = inv[1], c = inv[2], d = inv[4], e = inv[5], f = inv[6], gg = inv[8], h = inv[9], k2 = inv[10];
            const double A = e * k2 - f * h, B = c * h - b * k2, Cc = b * f - c * e;
            const double det = a * A + d * B + gg * Cc;
            if (!(fabs(det) > 1e-300) || !(det == det)) continue;
            const double id = 1.0 / det;
            fwd[0] = A * id; fwd[1] = B * id; fwd[2] = Cc * id; fwd[3] = 0.0;
            fwd[4] = (f * gg - d * k2) * id; fwd[5] = (a * k2 - c * gg) * id; fwd[6] = (c * d - a * f) * id; fwd[7] = 0.0;
            fwd[8] = (d * h - e * gg) * id; fwd[9] = (b * gg - a * h) * id; fwd[10] = (a * e - b * d) * id; fwd[11] = 0.0;
            for (int col = 0; col < 3; ++col) fwd[12 + col] = -(inv[12] * fwd[col] + inv[13] * fwd[4 + col] + inv[14] * fwd[8 + col]);
            fwd[15] = 1.0;
        } else if (!invert4(inv, fwd)) continue;
        // the box around the root's two child boxes (= the root box for a tree from BuildBVH)
        const CrtBVHNode* kid = g.hRootKids[inst.meshIndex];
        double lo[3], hi[3];
        for (int a = 0; a < 3; ++a) { lo[a] = fmin((double)kid[0].aabbMin[a], (double)kid[1].aabbMin[a]); hi[a] = fmax((double)kid[0].aabbMax[a], (double)kid[1].aabbMax[a]); }
        // sphere = image of the box: centre = image of the box centre, radius = the farthest image of a corner. A corner is centre +- the half extents, so
        // its image lies at +- hx row0 +- hy row1 +- hz row2 of the 3x3 part from the centre's: four sign patterns cover the eight corners
        const double mid[3] = { 0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2]) }, hx = 0.5 * (hi[0] - lo[0]), hy = 0.5 * (hi[1] - lo[1]), hz = 0.5 * (hi[2] - lo[2]);
        double cw[3];
        for (int c = 0; c < 3; ++c) cw[c] = mid[0] * fwd[0 + c] + mid[1] * fwd[4 + c] + mid[2] * fwd[8 + c] + fwd[12 + c];
        double r = 0.0;
        for (int k = 0; k < 4; ++k) {
            const double sy = (k & 1) ? -hy : hy, sz = (k & 2) ? -hz : hz;
            double d2 = 0.0;
            for (int c = 0; c < 3; ++c) { const double v = hx * fwd[0 + c] + sy * fwd[4 + c] + sz * fwd[8 + c]; d2 += v * v; }
            if (d2 > r) r = d2;
        }
        r = sqrt(r);
        // the fp32 centre the kernel reads differs from the exact one: the radius takes the difference
        const float cf[3] = { (float)cw[0], (float)cw[1], (float)cw[2] };
        const double ex = cw[0] - (double)cf[0], ey = cw[1] - (double)cf[1], ez = cw[2] - (double)cf[2];
        const float rf = (float)((r * (1.0 + 1e-4) + sqrt(ex * ex + ey * ey + ez * ez)) * (1.0 + 1e-6));
        const float4 b = make_float4(cf[0], cf[1], cf[2], rf);
        if (!(isfinite(b.x) && isfinite(b.y) && isfinite(b.z) && isfinite(b.w)) || !(b.w < 1e18f) || !(b.w > 1e-18f)) continue;
        // O_i of the derivation in crt_device.h: kappa = |M3|_F |F3|_F, tau = |T| |F3|_F, c1 = (1 + sqrt 3) g3 kappa
        double m3 = 0.0, f3 = 0.0, t2 = 0.0;
        for (int rr = 0; rr < 3; ++rr) for (int c = 0; c < 3; ++c) { m3 += inv[rr * 4 + c] * inv[rr * 4 + c]; f3 += fwd[rr * 4 + c] * fwd[rr * 4 + c]; }
        for (int c = 0; c < 3; ++c) t2 += inv[12 + c] * inv[12 + c];
        const double kappa = sqrt(m3) * sqrt(f3), tau = sqrt(t2) * sqrt(f3), c1 = (1.0 + sqrt(3.0)) * G3 * kappa;
        const double inside = 1.02 * (1.0 - c1 * c1 / K);
        double limit = inside > 0.0 ? ((double)rf * (sqrt(inside) - 1.0 - c1) - G4 * tau) / (G4 * kappa) : -1.0;
        limit *= rangeScale;                  // 1 unless the test hook below stretches the range to find where the cull really starts to err
        if (!(limit >= reach)) continue;      // (also NaN) never culled: bounce rays alone would leave the proven range
        g.hCullOriginLimit[i] = (float)fmin(limit * (1.0 - 1e-6), 3e38);
        bounds[i] = b;
    }
    {   // the smallest O_i over the cullable instances (the conversion above is monotonic: the minimum of the stored values is the stored minimum)
        float lo = (float)fmin(1e30 * (1.0 - 1e-6), 3e38);
        for (uint32_t i = 0; i < g.instHigh; ++i) if (bounds[i].w >= 0.0f && g.hCullOriginLimit[i] < lo) lo = g.hCullOriginLimit[i];
        g.cullOriginLimit = lo;
    }
    // Instance tree for scenes with many instances (closest_hit<..., TLAS>): median-split binary tree over the cullable
    // instances' spheres, node sphere = centre and half diagonal of the box around its children's spheres. Instances
    // that are never culled go to a separate ascending list.
    {
        CrtTlasNode* nodes = g.hTlas;
        // node sphere from the box around what is below it (a node's sphere holds >= 2 instance spheres, so its radius is >= sqrt 3 x theirs and its
        // share of the slack covers their Delta: crt_device.h (5); the fp32 centre's rounding goes into the radius as for the instances)
        auto node_sphere = [](const double lo[3], const double hi[3]) {
            const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            const double nc[3] = { 0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2]) };
            const float ncf[3] = { (float)nc[0], (float)nc[1], (float)nc[2] };
            const double nex = nc[0] - (double)ncf[0], ney = nc[1] - (double)ncf[1], nez = nc[2] - (double)ncf[2];
            return make_float4(ncf[0], ncf[1], ncf[2], (float)((0.5 * sqrt(dx * dx + dy * dy + dz * dz) * (1.0 + 1e-5) + sqrt(nex * nex + ney * ney + nez * nez)) * (1.0 + 1e-6)));
        };
        // is the tree of the last build still a tree over exactly the cullable instances of now?
        bool sameSet = g.hTlasBuilt && g.hTlasBuiltHigh == g.instHigh && g.hTlasRefits < 256u;
        for (uint32_t k = 0; sameSet && k < g.instHigh; ++k) sameSet = g.hTlasMember[k] == (uint8_t)(bounds[k].w >= 0.0f);
        bool refitted = false;
        if (sameSet && g.hTlasNodes > 0) {
            // REFIT: the same partition, boxes bottom-up (children are numbered after their parent) -- the spheres a rebuild with this partition would give
            double blo[2 * CRT_MAX_INSTANCES][3], bhi[2 * CRT_MAX_INSTANCES][3];
            double sumR = 0.0;
            for (uint32_t n = g.hTlasNodes; n-- > 0;) {
                CrtTlasNode& nd = nodes[n];
                if (nd.left & CRT_TLAS_LEAF) {
                    const float4 b = bounds[nd.left & 0xFFFFu];
                    const double c[3] = { b.x, b.y, b.z };
                    for (int a = 0; a < 3; ++a) { blo[n][a] = c[a] - b.w; bhi[n][a] = c[a] + b.w; }
                    nd.sphere = b;
                } else {
                    for (int a = 0; a < 3; ++a) { blo[n][a] = fmin(blo[nd.left][a], blo[nd.right][a]); bhi[n][a] = fmax(bhi[nd.left][a], bhi[nd.right][a]); }
                    nd.sphere = node_sphere(blo[n], bhi[n]);
                    sumR += (double)nd.sphere.w;
                }
            }
            refitted = sumR <= 1.25 * g.hTlasBuiltRadii;            // (NaN: rebuild)
            if (refitted) g.hTlasRefits++;
        } else if (sameSet) refitted = true;                        // no cullable instance then and now: nothing to do
        if (!refitted) {
            uint32_t* always = g.hAlways;
            uint32_t nAlways = 0, nLeaves = 0, nNodes = 0;
            uint32_t leaves[CRT_MAX_INSTANCES];
            // only instances that were uploaded; a frame that asks for more (never-uploaded, all-zero records) uses the linear loop
            for (uint32_t k = 0; k < g.instHigh; ++k) { const bool cullable = bounds[k].w >= 0.0f; g.hTlasMember[k] = (uint8_t)cullable; if (!cullable) always[nAlways++] = k; else leaves[nLeaves++] = k; }
            struct Range { uint32_t lo, hi, node; };
            double sumR = 0.0;
            if (nLeaves) {
                Range stack[64]; int sp = 0;
                stack[sp++] = Range{ 0, nLeaves, nNodes++ };
                while (sp) {
                    const Range r = stack[--sp];
                    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 }, clo[3] = { 1e300, 1e300, 1e300 }, chi[3] = { -1e300, -1e300, -1e300 };
                    for (uint32_t k = r.lo; k < r.hi; ++k) {
                        const float4 b = bounds[leaves[k]]; const double c[3] = { b.x, b.y, b.z };
                        for (int a = 0; a < 3; ++a) {
                            if (c[a] - b.w < lo[a]) lo[a] = c[a] - b.w;
                            if (c[a] + b.w > hi[a]) hi[a] = c[a] + b.w;
                            if (c[a] < clo[a]) clo[a] = c[a];
                            if (c[a] > chi[a]) chi[a] = c[a];
                        }
                    }
                    CrtTlasNode& n = nodes[r.node];
                    n.pad0 = n.pad1 = 0;
                    if (r.hi - r.lo == 1) { n.sphere = bounds[leaves[r.lo]]; n.left = CRT_TLAS_LEAF | leaves[r.lo]; n.right = 0; continue; }
                    n.sphere = node_sphere(lo, hi);
                    sumR += (double)n.sphere.w;
                    int axis = 0;
                    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
                    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
                    const uint32_t mid = (r.lo + r.hi) / 2;
                    auto key = [&](uint32_t idx) { const float4 b = bounds[idx]; return axis == 0 ? b.x : (axis == 1 ? b.y : b.z); };
                    std::nth_element(leaves + r.lo, leaves + mid, leaves + r.hi, [&](uint32_t p, uint32_t q) { return key(p) < key(q) || (key(p) == key(q) && p < q); });
                    n.left = nNodes++; n.right = nNodes++;
                    stack[sp++] = Range{ mid, r.hi, n.right };
                    stack[sp++] = Range{ r.lo, mid, n.left };
                }
            }
            g.hTlasNodes = nNodes; g.hNumAlways = nAlways;
            g.hTlasBuilt = true; g.hTlasBuiltHigh = g.instHigh; g.hTlasBuiltRadii = sumR; g.hTlasRefits = 0; g.hTlasBuilds++;
        }
    }
    g.instVersion++;
}

// Offsets of the tables inside a slot's pinned staging block
// (r6: one block on the device with the same layout, the instance tree last, so a refresh is ONE copy of the used extent instead of four)
constexpr size_t kStageInst = 0;
constexpr size_t kStageBounds = (kStageInst + CRT_MAX_INSTANCES * sizeof(CrtMeshInstance) + 255) & ~(size_t)255;
constexpr size_t kStageAlways = (kStageBounds + CRT_MAX_INSTANCES * sizeof(float4) + 255) & ~(size_t)255;
constexpr size_t kStageTlas = (kStageAlways + CRT_MAX_INSTANCES * sizeof(uint32_t) + 255) & ~(size_t)255;
constexpr size_t kStageBytes = kStageTlas + 2 * CRT_MAX_INSTANCES * sizeof(CrtTlasNode);

// Brings a slot's instance tables up to the host master, on the slot's own stream, before a frame (or query) uses them.
int ensure_slot_instances(FrameSlot& fs)
{
    if (fs.instVersion == g.instVersion) return CRT_OK;
    HIPCHK(hipEventSynchronize(fs.staged));                        // the previous refresh no longer reads the staging block
    memcpy(fs.staging + kStageInst, g.hInstances, CRT_MAX_INSTANCES * sizeof(CrtMeshInstance));
    memcpy(fs.staging + kStageBounds, g.hBounds, CRT_MAX_INSTANCES * sizeof(float4));
    if (g.hNumAlways) memcpy(fs.staging + kStageAlways, g.hAlways, g.hNumAlways * sizeof(uint32_t));
    if (g.hTlasNodes) memcpy(fs.staging + kStageTlas, g.hTlas, g.hTlasNodes * sizeof(CrtTlasNode));
    // every frame of an animated scene pays this (upstream: Renderer.cpp:312-320, one clEnqueueWriteBuffer of the dirty range): ONE small launch that
    // reads the pinned block over the link, fills the device block and builds the device records (crt_refresh_instances_kernel)
    static_assert(kStageInst == 0 && kStageBytes % 16 == 0 && kStageTlas % 16 == 0 && sizeof(CrtTlasNode) % 16 == 0, "copied as 16-byte words");
    const uint32_t words16 = (uint32_t)((kStageTlas + g.hTlasNodes * sizeof(CrtTlasNode)) / 16);
    crt_refresh_instances_kernel<<<16, 256, 0, fs.stream>>>(reinterpret_cast<const uint4*>(fs.stagingDev), reinterpret_cast<uint4*>(fs.instBlock), words16,
                                                             g.rootRefs, g.topRootRefs, CRT_MAX_INSTANCES, fs.devInstances);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(fs.staged, fs.stream));
    fs.tlasNodes = g.hTlasNodes; fs.numAlways = g.hNumAlways; fs.instVersion = g.instVersion;
    return CRT_OK;
}

} // namespace
