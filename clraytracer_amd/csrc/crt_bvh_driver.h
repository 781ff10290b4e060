// crt_bvh_driver.h -- host driver of the device BuildBVH (kernels: crt_bvh_build.h; reference: BVH.cpp:218-255)
// Part of the one translation unit crt_shim.hip (included there, in this order: crt_state.h, crt_instances.h, crt_upload.h,
// crt_bvh_driver.h, crt_frame.h, crt_multidev.h); everything here has internal linkage.
#pragma once
namespace {

// BuildBVH on the device (crt_bvh_build.h): same triangle order, node numbering and bounds as the host builder.
int crt1_build_bvh(size_t firstTri, const uint32_t* meshTriCounts, int numMeshes, size_t firstNode, size_t firstMesh, uint32_t* nodesUsedOut)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!meshTriCounts || numMeshes < 1) return CRT_E_BAD_ARGUMENT;
    if (firstMesh + (size_t)numMeshes > CRT_MAX_MESHES) return CRT_E_OUT_OF_RANGE;
    size_t total = 0;
    for (int m = 0; m < numMeshes; ++m) { if (meshTriCounts[m] == 0) return CRT_E_BAD_ARGUMENT; total += meshTriCounts[m]; }
    if (firstTri + total > g.trisHigh) return CRT_E_BAD_ARGUMENT;                  // triangles must have been uploaded
    if (firstTri + total > 0x00FFFFFFu) return CRT_E_OUT_OF_RANGE;                 // leaf references carry 24-bit triangle indices
    if (firstNode + 2 * total > g.nodeCap) return CRT_E_OUT_OF_RANGE;              // a mesh of n triangles needs at most 2n-1 nodes
    if (total / CRT_BVH_SMALL >= (1u << 20) || total / CRT_BVH_TINY >= (1u << 20)) return CRT_E_OUT_OF_RANGE;   // field widths of the packed per-level counter (crt_bvh_build.h)
    {   // test hook (CRT_DEBUG_HOOKS=1 only): refuse, so that the caller's fall-back to the host BuildBVH can be exercised
        const char* h = getenv("CRT_DEBUG_HOOKS"); const char* f = getenv("CRT_DEBUG_FAIL_BVH_BUILD");
        // CRT_DEBUG_FAIL_BVH_BUILD: 1 = CRT_E_OUT_OF_RANGE; any other non-zero value is returned as it is -- 2 (hipErrorOutOfMemory: the builder's
        // scratch could not be allocated), -2 (CRT_E_BAD_ARGUMENT), 719 (hipErrorLaunchFailure: a sticky fault, the one kind the caller must not paper over)
        if (h && atoi(h) != 0 && f && atoi(f) != 0) return atoi(f) == 1 ? (int)CRT_E_OUT_OF_RANGE : atoi(f);
    }
    RCCHK(sync_all());
    g.buildLaunches = 0; g.buildLevels = 0;

    // second triangle pool (allocated on first use, indexed like rawTris) and scratch:
    // build nodes | rank, holes, backL | 2 x 3 id lists | 2 x BIG-node scratch | 2 x chunk->node + 3 per-chunk counts | mesh counts, roots | scalars
    if (!g.buildTris) HIPCHK(hipMalloc(&g.buildTris, g.triCap * sizeof(CrtTri)));
    if (!g.buildCtlHost) { HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&g.buildCtlHost), sizeof(CrtBuildCtlHost), hipHostMallocMapped | hipHostMallocCoherent)); g.buildCtlHost->seq = 0; g.buildSeq = 0; }
    const size_t maxNodes = 2 * total + (size_t)numMeshes;
    const size_t offNodes = 0;
    const size_t offRank = (offNodes + maxNodes * sizeof(CrtBuildNode) + 255) & ~(size_t)255;
    const size_t offLists = (offRank + 3 * total * sizeof(uint32_t) + 255) & ~(size_t)255;
    const size_t listCap = total + (size_t)numMeshes;                              // a level never has more nodes than triangles
    const size_t maxBig = total / CRT_BVH_SMALL + (size_t)numMeshes + 1;           // BIG nodes of one level (each has more than CRT_BVH_SMALL triangles)
    const size_t maxChunks = total / CRT_BVH_CHUNK + maxBig + 1;                   // sum of ceil(n / CRT_BVH_CHUNK) over them
    const size_t offBig = (offLists + 6 * listCap * sizeof(uint32_t) + 255) & ~(size_t)255;
    const size_t offChunks = (offBig + 2 * maxBig * sizeof(CrtBigScratch) + 255) & ~(size_t)255;
    const size_t offSmall = (offChunks + 5 * maxChunks * sizeof(uint32_t) + 255) & ~(size_t)255;
    const int kCtlLevels = 256;                                                    // one zeroed control record per level up to here (one memset); deeper levels reuse the last one
    const size_t need = offSmall + (2 * (size_t)numMeshes + 8) * sizeof(uint32_t) + kCtlLevels * sizeof(CrtBuildCtl) + 16;
    if (need > g.buildBytes) {
        if (g.buildBuf) (void)hipFree(g.buildBuf);
        g.buildBuf = nullptr; g.buildBytes = 0;
        HIPCHK(hipMalloc(&g.buildBuf, need));
        g.buildBytes = need;
    }
    char* base = static_cast<char*>(g.buildBuf);
    CrtTri* A = g.rawTris;
    CrtTri* B = g.buildTris;
    CrtBuildNode* bn = reinterpret_cast<CrtBuildNode*>(base + offNodes);
    uint32_t* rank = reinterpret_cast<uint32_t*>(base + offRank);
    uint32_t* holes = rank + total; uint32_t* backL = holes + total;
    uint32_t* listMem = reinterpret_cast<uint32_t*>(base + offLists);
    CrtBuildLists lists[2];
    for (int p = 0; p < 2; ++p) for (int c = 0; c < 3; ++c) lists[p].list[c] = listMem + ((size_t)p * 3 + (size_t)c) * listCap;
    CrtBigScratch* bigs[2] = { reinterpret_cast<CrtBigScratch*>(base + offBig), reinterpret_cast<CrtBigScratch*>(base + offBig) + maxBig };
    uint32_t* chunkMem = reinterpret_cast<uint32_t*>(base + offChunks);
    uint32_t* chunkNode[2] = { chunkMem, chunkMem + maxChunks };
    uint32_t* chunkL = chunkMem + 2 * maxChunks; uint32_t* chunkFR = chunkL + maxChunks; uint32_t* chunkBL = chunkFR + maxChunks;
    uint32_t* dCounts = reinterpret_cast<uint32_t*>(base + offSmall);
    uint32_t* dRoots = dCounts + numMeshes;
    uint32_t* dScal = dRoots + numMeshes;                                          // [0] nodes used
    CrtBuildCtl* dCtls = reinterpret_cast<CrtBuildCtl*>((reinterpret_cast<uintptr_t>(dScal + 2) + 15) & ~(uintptr_t)15);  // per level: next level's list sizes and chunk count (crt_bvh_build.h)
    hipStream_t st = g.stream;
    HIPCHK(hipMemsetAsync(dCtls, 0, kCtlLevels * sizeof(CrtBuildCtl), st));
    HIPCHK(hipMemcpyAsync(dCounts, meshTriCounts, (size_t)numMeshes * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    // level 0 = the roots, classified here
    uint32_t cnt[3] = { 0, 0, 0 };
    uint32_t chunks = 0;                                                           // chunks of the current level's BIG nodes
    {
        std::vector<uint32_t> ids[3];
        for (int m = 0; m < numMeshes; ++m) {
            const int cls = bvh_class(meshTriCounts[m]);
            ids[cls].push_back((uint32_t)m);
            if (cls == CRT_BVH_CLASS_BIG) chunks += bvh_chunks(meshTriCounts[m]);
        }
        for (int c = 0; c < 3; ++c) {
            cnt[c] = (uint32_t)ids[c].size();
            if (cnt[c]) HIPCHK(hipMemcpyAsync(lists[0].list[c], ids[c].data(), cnt[c] * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        }
        HIPCHK(hipStreamSynchronize(st));                                          // ids[] go out of scope
    }
    crt_bvh_centroids<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(A, firstTri, total); ++g.buildLaunches;
    crt_bvh_init_roots<<<1, 1, 0, st>>>(bn, dCounts, numMeshes, (uint32_t)firstTri, bigs[0], chunkNode[0]); ++g.buildLaunches;
    HIPCHK(hipGetLastError());

    // Measurement hook (CRT_DEBUG_HOOKS=1 + CRT_DEBUG_BVH_REPLAY=1): the builder is deterministic, so a build of the SAME input can
    // take every level's list sizes from a recording of the previous one and enqueue all its launches back to back -- exact grids, no publish
    // kernel, no host round trip per level. That is the floor of ANY one-submission scheme (a HIP graph re-launched with device-read grid sizes,
    // a persistent kernel with a device-side level counter): what `crt_build_bvh` would take if the level hand-shake cost nothing.
    // (The caller must rebuild the SAME triangles -- tools/bvh_build_time.py does; the key covers only the mesh sizes. A mismatch ends in
    // CRT_E_OUT_OF_RANGE at the final node-count check.)
    bool replaying = false, recording = false;
    {
        const char* h = getenv("CRT_DEBUG_HOOKS"); const char* r = getenv("CRT_DEBUG_BVH_REPLAY");
        if (h && atoi(h) != 0 && r && atoi(r) != 0) {
            unsigned long long key = 1469598103934665603ull;                        // FNV-1a over what determines the level structure besides the triangles themselves
            auto mix = [&](unsigned long long v) { for (int b = 0; b < 8; ++b) { key ^= (v >> (8 * b)) & 0xFFu; key *= 1099511628211ull; } };
            mix(firstTri); mix(total); mix((unsigned long long)numMeshes);
            for (int m = 0; m < numMeshes; ++m) mix(meshTriCounts[m]);
            replaying = !g.buildReplay.empty() && g.buildReplayKey == key;
            if (!replaying) { g.buildReplay.clear(); g.buildReplayKey = key; recording = true; }
        } else g.buildReplay.clear();
    }
    const unsigned W = CRT_BVH_WAVES, T = CRT_BVH_BIG_THREADS;
    auto bounds = [&](int p, const uint32_t n[3], uint32_t nChunks, const CrtTri* tris) {
        const CrtBuildLists& L = lists[p];
        if (n[0]) { crt_bvh_big_reset<<<(n[0] + 255) / 256, 256, 0, st>>>(bigs[p], n[0]); ++g.buildLaunches;
                    crt_bvh_big_bounds<<<nChunks, T, 0, st>>>(bn, L.list[0], bigs[p], chunkNode[p], tris); ++g.buildLaunches; }
        // (MID and TINY nodes compute their bounds at the top of crt_bvh_mid / crt_bvh_tiny: round 5, 26 launches fewer per 1 M-triangle build)
    };
    bounds(0, cnt, chunks, A);
    uint32_t begin = 0, end = (uint32_t)numMeshes;
    CrtTri* src = A; CrtTri* dst = B;
    int cur = 0, level = 0;
    while (end > begin) {
        const CrtBuildLists& L = lists[cur]; const CrtBuildLists& N = lists[cur ^ 1];
        CrtBuildCtl ctl = { 0, 0, 0 };
        CrtBuildCtl* dCtl = dCtls + (level < kCtlLevels ? level : kCtlLevels - 1);
        if (level >= kCtlLevels - 1) HIPCHK(hipMemsetAsync(dCtl, 0, sizeof ctl, st));   // the shared last record (zero already on its first use: harmless)
        ++level; ++g.buildLevels;
        if (cnt[0]) {
            CrtBigScratch* big = bigs[cur]; const uint32_t* cn = chunkNode[cur];
            crt_bvh_big_bins<<<chunks, T, 0, st>>>(bn, L.list[0], big, cn, src); ++g.buildLaunches;
            crt_bvh_big_sweep<<<chunks, T, 0, st>>>(bn, L.list[0], big, cn, src, dst, chunkL); ++g.buildLaunches;
            crt_bvh_big_count<<<chunks, T, 0, st>>>(bn, L.list[0], big, cn, src, chunkL, chunkFR, chunkBL); ++g.buildLaunches;
            crt_bvh_big_tables<<<chunks, T, 0, st>>>(bn, L.list[0], big, cn, src, (uint32_t)firstTri, chunkFR, chunkBL, rank, holes, backL); ++g.buildLaunches;
            crt_bvh_big_scatter<<<chunks, T, 0, st>>>(bn, L.list[0], big, cn, src, dst, (uint32_t)firstTri, rank, holes, backL, end, dCtl, N, bigs[cur ^ 1], chunkNode[cur ^ 1]); ++g.buildLaunches;
        }
        if (cnt[1]) { crt_bvh_mid<<<(cnt[1] + W - 1) / W, 64 * W, 0, st>>>(bn, L.list[1], cnt[1], src, dst, (uint32_t)firstTri, rank, holes, backL, end, &dCtl->packed, N); ++g.buildLaunches; }
        if (cnt[2]) { crt_bvh_tiny<<<(cnt[2] + 63) / 64, 64, 0, st>>>(bn, L.list[2], cnt[2], src, dst, end, &dCtl->packed, N); ++g.buildLaunches; }
        HIPCHK(hipGetLastError());
        if (replaying && (size_t)(level - 1) < g.buildReplay.size()) {
            // measurement hook (below): the level's list sizes are known from the recorded build of the same input -- no publish, no wait
            ctl = g.buildReplay[(size_t)(level - 1)];
        } else {
            // the level's list sizes: published into pinned memory behind the level's kernels; spin on the sequence number (a copy + stream
            // synchronisation per level cost ~40 us x 23 levels of a 1 M-triangle build), fall back to the stream if it does not arrive
            const uint32_t seq = ++g.buildSeq;
            crt_bvh_publish<<<1, 1, 0, st>>>(dCtl, g.buildCtlHost, seq); ++g.buildLaunches;
            HIPCHK(hipGetLastError());
            bool arrived = false;
            for (unsigned spin = 0; spin < (1u << 22) && !g.buildNoSpin; ++spin) {
                if (g.buildCtlHost->seq == seq) { arrived = true; break; }
                if ((spin & 0x3FFu) == 0x3FFu && hipStreamQuery(st) != hipErrorNotReady) {
                    // finished (or failed) between two looks at the flag: the record may have landed in that window -- look once more
                    // before blaming coherence; a failed stream is sorted out by the synchronisation below
                    __atomic_thread_fence(__ATOMIC_ACQUIRE);
                    arrived = g.buildCtlHost->seq == seq;
                    break;
                }
                crt_cpu_relax();
            }
            if (!arrived) {
                HIPCHK(hipStreamSynchronize(st)); if (g.buildCtlHost->seq != seq) return CRT_E_UNSUPPORTED;
                // Latched only when the record really was invisible while the level ran: the spin budget ran out with the stream still busy, or
                // the stream had drained and the record was STILL stale at the re-read above (host memory not coherent with running kernels).
                // From then on wait for the stream at once instead of burning the spin budget on every level.
                if (!g.buildNoSpin) { g.buildNoSpin = true; fprintf(stderr, "[crt] crt_build_bvh: the pinned control record is not visible before the stream drains; using stream synchronisation per level\n"); }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            ctl = g.buildCtlHost->ctl;
            if (recording) g.buildReplay.push_back(ctl);
        }
        if (ctl.degenerate) {                                                      // BVH.cpp:194 hit a BIG node: its permuted triangles go to both buffers
            crt_bvh_big_degenerate<<<chunks, T, 0, st>>>(bn, L.list[0], bigs[cur], chunkNode[cur], src, dst); ++g.buildLaunches;
            crt_bvh_big_degenerate_mark<<<(cnt[0] + 255) / 256, 256, 0, st>>>(bn, L.list[0], bigs[cur], cnt[0]); ++g.buildLaunches;
        }
        for (int c = 0; c < 3; ++c) cnt[c] = bvh_unpack(ctl.packed, c);
        chunks = ctl.nextChunks;
        const uint32_t newEnd = end + cnt[0] + cnt[1] + cnt[2];
        if (newEnd > (uint32_t)maxNodes || cnt[0] > maxBig || chunks > maxChunks) { (void)hipStreamSynchronize(st); return CRT_E_OUT_OF_RANGE; }   // nothing stays queued behind a refused build
        bounds(cur ^ 1, cnt, chunks, dst);
        begin = end; end = newEnd;
        CrtTri* t = src; src = dst; dst = t;
        cur ^= 1;
    }
    const uint32_t numBuilt = end;
    if (firstNode + numBuilt > g.nodeCap) { (void)hipStreamSynchronize(st); return CRT_E_OUT_OF_RANGE; }
    // numbering in closed form (crt_bvh_build.h): leaf starts -> exclusive prefix counts S (flags in `rank`, S in `holes`..: total + 1 words) -> one pass
    {
        uint32_t* flags = rank; uint32_t* S = holes; uint32_t* sums = chunkL;
        const uint32_t nb = (uint32_t)(total / CRT_BVH_SCAN_ITEMS) + 1;
        if (nb > maxChunks) { (void)hipStreamSynchronize(st); return CRT_E_OUT_OF_RANGE; }
        HIPCHK(hipMemsetAsync(flags, 0, total * sizeof(uint32_t), st));
        HIPCHK(hipMemsetAsync(dScal, 0, 2 * sizeof(uint32_t), st));                     // [0] nodes used, [1] "a node number fell outside the node array"
        crt_bvh_leaf_flags<<<(numBuilt + 255) / 256, 256, 0, st>>>(bn, numBuilt, (uint32_t)firstTri, flags); ++g.buildLaunches;
        crt_bvh_scan_sums<<<nb, CRT_BVH_SCAN_THREADS, 0, st>>>(flags, (uint32_t)total, sums); ++g.buildLaunches;
        crt_bvh_scan_blocks<<<1, CRT_BVH_SCAN_THREADS, 0, st>>>(sums, nb); ++g.buildLaunches;
        crt_bvh_scan_apply<<<nb, CRT_BVH_SCAN_THREADS, 0, st>>>(flags, (uint32_t)total, sums, S); ++g.buildLaunches;
        crt_bvh_emit<<<(numBuilt + 255) / 256, 256, 0, st>>>(bn, numBuilt, numMeshes, S, (uint32_t)firstTri, (uint32_t)total, (uint32_t)firstNode, g.rawNodes, dRoots, dScal, dScal + 1); ++g.buildLaunches;
        HIPCHK(hipGetLastError());
    }
    uint32_t used = 0, scal[2] = { 0, 0 };
    HIPCHK(hipMemcpyAsync(scal, dScal, sizeof scal, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(g.roots + firstMesh, dRoots, (size_t)numMeshes * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(g.hRoots + firstMesh, dRoots, (size_t)numMeshes * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    crt_relayout_tris<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(g.rawTris, firstTri, total, g.triHot, g.triCold); ++g.buildLaunches;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    used = scal[0];
    if (scal[1] != 0 || used != numBuilt) return CRT_E_OUT_OF_RANGE;                               // the closed form and the level loop disagree: never seen, would mean a damaged tree
    if (firstNode + used > g.nodeCount) g.nodeCount = (uint32_t)(firstNode + used);
    if (firstMesh + (size_t)numMeshes > g.numRoots) g.numRoots = (uint32_t)(firstMesh + (size_t)numMeshes);
    if (nodesUsedOut) *nodesUsedOut = used;
    return rebuild_bvh_layout();
}

} // namespace
