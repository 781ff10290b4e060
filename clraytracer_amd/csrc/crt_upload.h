// crt_upload.h -- session set-up / tear-down, uploads in the reference layouts (Renderer.cpp:122-193, ResourceManager.cpp:145-300) and read-backs of the pools
// Part of the one translation unit crt_shim.hip (included there, in this order: crt_state.h, crt_instances.h, crt_upload.h,
// crt_bvh_driver.h, crt_frame.h, crt_multidev.h); everything here has internal linkage.
#pragma once
namespace {
// ---- per-device implementation of the C-ABI entry points (current state = g) ----


const char* crt1_device_name(void) { return g.deviceName; }

int crt1_upload_texels(const void* rgb8, size_t byteOffset, size_t bytes);

static int init_impl(int device, int width, int height)
{
    if (g.initialized) return CRT_E_BAD_ARGUMENT;
    if (width < 16 || height < 16) return CRT_E_BAD_ARGUMENT;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CRT_E_NO_DEVICE;
    if (device < 0 || device >= n) return CRT_E_BAD_ARGUMENT;
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    snprintf(g.deviceName, sizeof g.deviceName, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    g.device = device;
    // frames in flight: 3 by default; more pays when a frame is small against its slowest tile (a rank's 1/8 share of a
    // frame: DESIGN.md 6). Each slot has its own stream; past four the runtime needs GPU_MAX_HW_QUEUES raised before
    // its first call, or it folds the streams onto four hardware queues (crt_init_devices does that when it still can).
    { const char* e = getenv("CRT_FRAMES_IN_FLIGHT"); g.nSlots = e ? atoi(e) : 3; if (g.nSlots < 1) g.nSlots = 1; if (g.nSlots > CRT_MAX_FRAMES_IN_FLIGHT) g.nSlots = CRT_MAX_FRAMES_IN_FLIGHT; }
    for (int si = 0; si < g.nSlots; ++si) {
        FrameSlot& fs = g.slot[si];
        HIPCHK(hipStreamCreateWithFlags(&fs.stream, hipStreamNonBlocking));
        for (EventSet& es : fs.es) for (int i = 0; i < 4; ++i) HIPCHK(hipEventCreate(&es.ev[i]));
        HIPCHK(hipMalloc(&fs.instBlock, kStageBytes));         // the slot's instance tables in the staging block's layout (crt_instances.h)
        fs.instances = reinterpret_cast<CrtMeshInstance*>(fs.instBlock + kStageInst); fs.instBounds = reinterpret_cast<float4*>(fs.instBlock + kStageBounds);
        fs.alwaysList = reinterpret_cast<uint32_t*>(fs.instBlock + kStageAlways); fs.tlas = reinterpret_cast<CrtTlasNode*>(fs.instBlock + kStageTlas);
        HIPCHK(hipMalloc(&fs.devInstances, CRT_MAX_INSTANCES * sizeof(CrtDevInstance)));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&fs.staging), kStageBytes, hipHostMallocDefault));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&fs.stagingDev), fs.staging, 0));     // the refresh kernel reads the pinned block itself
        HIPCHK(hipEventCreateWithFlags(&fs.staged, hipEventDisableTiming));
        HIPCHK(hipEventRecord(fs.staged, fs.stream));
        HIPCHK(hipEventCreateWithFlags(&fs.partDone, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&fs.slotDone, hipEventDisableTiming));
        fs.instVersion = 0;
    }
    HIPCHK(hipEventCreate(&g.statStart));
    g.stream = g.slot[0].stream; g.cur = 0; g.asyncSeq = 0; g.othersBusy = false;

    g.triCap = (size_t)CRT_MAX_TRIANGLES * 2;           // ResourceManager.cpp:158
    g.nodeCap = (size_t)CRT_MAX_TRIANGLES * 2;          // ResourceManager.cpp:159 (MAX_BVHMEMORY * 2)
    g.texelByteCap = CRT_MAX_TEXTURE_BYTES * 2;         // ResourceManager.cpp:163
    HIPCHK(hipMalloc(&g.rawTris, g.triCap * sizeof(CrtTri)));
    HIPCHK(hipMemsetAsync(g.rawTris, 0, g.triCap * sizeof(CrtTri), g.stream));   // crt_tri_reach_kernel may scan slots nobody has uploaded yet (an upload that leaves a gap)
    HIPCHK(hipMalloc(&g.rawNodes, g.nodeCap * sizeof(CrtBVHNode)));
    HIPCHK(hipMalloc(&g.roots, CRT_MAX_MESHES * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.rawTexels, g.texelByteCap + 16));
    HIPCHK(hipMalloc(&g.pairs, (g.nodeCap / 2 + 1) * 4 * sizeof(float4)));
    HIPCHK(hipMalloc(&g.triHot, g.triCap * 9 * sizeof(float)));
    HIPCHK(hipMalloc(&g.triCold, g.triCap * 2 * sizeof(uint4)));
    HIPCHK(hipMalloc(&g.bigLeaf, (g.triCap + 1) * sizeof(uint32_t)));
    HIPCHK(hipMemset(g.bigLeaf + g.triCap, 0, sizeof(uint32_t)));      // crt_empty_ref: a leaf of zero triangles
    HIPCHK(hipMalloc(&g.rootRefs, CRT_MAX_MESHES * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.topRootRefs, CRT_MAX_MESHES * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.topPairs, CRT_TOP_PAIRS * 4 * sizeof(float4)));
    HIPCHK(hipMemset(g.topPairs, 0, CRT_TOP_PAIRS * 4 * sizeof(float4)));
    HIPCHK(hipMalloc(&g.texels, (g.texelByteCap / 3 + 2) * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.materials, CRT_MAX_MATERIALS * sizeof(CrtMaterial)));
    HIPCHK(hipMalloc(&g.textures, CRT_MAX_TEXTURES * sizeof(CrtTexture)));
    HIPCHK(hipMalloc(&g.counters, CRT_NUM_COUNTERS * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&g.err, sizeof(int)));
    HIPCHK(hipMalloc(&g.triReachBits, sizeof(uint32_t)));
    HIPCHK(hipMemset(g.triReachBits, 0, sizeof(uint32_t)));
    {   // the "never cull" bounds table of frames whose rays start beyond the cull's proven range
        static float4 never[CRT_MAX_INSTANCES];
        for (float4& b : never) b = make_float4(0.f, 0.f, 0.f, -1.0f);
        HIPCHK(hipMalloc(&g.noCullBounds, sizeof never));
        HIPCHK(hipMemcpy(g.noCullBounds, never, sizeof never, hipMemcpyHostToDevice));
    }
    g.numCUs = prop.multiProcessorCount;
    {   // CRT_KERNEL: the Trace kernel structure of this session. Unset / "" / "default": the megakernel (faster);
        // "wavefront", "refill", "block": the opt-in compaction forms, "ldstop": tree tops staged in LDS. Anything else is a typo, not a wish for
        // the default: the session refuses to start (VERDICT r5 #3 -- a variant test must not pass because the name stopped matching).
        const char* e = getenv("CRT_KERNEL");
        g.wavefront = 0; g.refill = 0; g.ldstop = 0;
        if (e && *e && strcmp(e, "default") != 0) {
            if (strcmp(e, "wavefront") == 0) g.wavefront = 1;
            else if (strcmp(e, "ldstop") == 0) g.ldstop = 1;
            else if (strcmp(e, "refill") == 0) g.refill = 1;
            else if (strcmp(e, "block") == 0) g.refill = 2;
            else { fprintf(stderr, "crt_init: CRT_KERNEL=%s is not one of default, wavefront, refill, block, ldstop\n", e); return CRT_E_BAD_ARGUMENT; }
        }
    }
    { const char* e = getenv("CRT_SPLIT_BETA"); g.splitBeta = e ? (float)atof(e) : CRT_SPLIT_BETA; }
    { const char* e = getenv("CRT_SPLIT_BETA_ASYNC"); g.splitBetaAsync = e ? (float)atof(e) : CRT_SPLIT_BETA_ASYNC; }
    { const char* e = getenv("CRT_COST_SPREAD"); g.costSpread = e ? (float)atof(e) : 0.8f; }
    { const char* e = getenv("CRT_SPLIT");               // tuning knob: cap on quadrant-split tiles per XCD (both modes)
      if (e) { int v = atoi(e); v = v < 0 ? 0 : (v > CRT_MAX_SPLIT ? CRT_MAX_SPLIT : v); g.maxSplit = g.maxSplitPipelined = v; }
      else { g.maxSplit = CRT_MAX_SPLIT; g.maxSplitPipelined = CRT_MAX_SPLIT_PIPELINED; } }
    { const char* e = getenv("CRT_GATHER_RGBA8"); g.gather8 = !(e && atoi(e) == 0); }
    { const char* e = getenv("CRT_TLAS"); g.forceTlas = e ? (atoi(e) != 0 ? 1 : 0) : -1; }
    { const char* e = getenv("CRT_STAGGER_US"); g.staggerUs = e ? atoi(e) : -1; }
    { const char* e = getenv("CRT_FEEDBACK"); g.feedback = !(e && atoi(e) == 0); }
    { const char* e = getenv("CRT_FEEDBACK_ASYNC"); g.feedbackAsync = (e && atoi(e) != 0); }
    HIPCHK(hipMemset(g.roots, 0, CRT_MAX_MESHES * sizeof(uint32_t)));
    { std::vector<uint32_t> e(CRT_MAX_MESHES, crt_empty_ref((uint32_t)g.triCap)); HIPCHK(hipMemcpy(g.rootRefs, e.data(), e.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(g.topRootRefs, e.data(), e.size() * sizeof(uint32_t), hipMemcpyHostToDevice)); }
    HIPCHK(hipMemset(g.materials, 0, CRT_MAX_MATERIALS * sizeof(CrtMaterial)));
    HIPCHK(hipMemset(g.textures, 0, CRT_MAX_TEXTURES * sizeof(CrtTexture)));
    HIPCHK(hipMemset(g.texels, 0, 64));
    g.nodeCount = 0; g.numRoots = 0; g.texelBytesHigh = 0; g.trisHigh = 0; g.sceneValid = true; g.instHigh = 0;
    memset(g.hInstances, 0, sizeof g.hInstances); memset(g.hRoots, 0, sizeof g.hRoots);
    memset(g.hHaveRoot, 0, sizeof g.hHaveRoot); g.instVersion = 1;
    rebuild_instance_master();
    g.bandRows = 16; g.rank = 0; g.nRanks = 1;
    int rc = alloc_frame_buffers(width, height);
    if (rc) return rc;
    g.initialized = true;
    // default white / black texels (ResourceManager.cpp:168-177)
    const unsigned char def[6] = { 0xFF, 0xFF, 0xFF, 0, 0, 0 };
    return crt1_upload_texels(def, 0, 6);
}

// frees everything State holds (also after an init that failed half way) and resets it
static void release_all()
{
    for (FrameSlot& fs : g.slot) if (fs.stream) (void)hipStreamSynchronize(fs.stream);
    void* ptrs[] = { g.rawTris, g.rawNodes, g.roots, g.rawTexels, g.pairs, g.triHot, g.triCold, g.bigLeaf, g.rootRefs,
                     g.texels, g.materials, g.textures, g.rays, g.counters, g.err, g.triReachBits, g.topPairs, g.topRootRefs,
                     g.queryBuf, g.buildBuf, g.buildTris, g.stamps, g.noCullBounds };
    for (FrameSlot& fs : g.slot) {
        void* q[] = { fs.out, fs.aux, fs.blockQueue, fs.wfCount, fs.ovf, fs.order, fs.len, fs.cost, fs.mixOrder, fs.mixLen, fs.packBuf, fs.instBlock, fs.devInstances };
        for (void* p : q) if (p) (void)hipFree(p);
        if (fs.staging) (void)hipHostFree(fs.staging);
        if (fs.staged) (void)hipEventDestroy(fs.staged);
        if (fs.partDone) (void)hipEventDestroy(fs.partDone);
        if (fs.slotDone) (void)hipEventDestroy(fs.slotDone);
        if (fs.hostBuf) (void)hipHostFree(fs.hostBuf);
        if (fs.copied) (void)hipEventDestroy(fs.copied);
    }
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (g.statStart) (void)hipEventDestroy(g.statStart);
    if (g.buildCtlHost) (void)hipHostFree(g.buildCtlHost);
    for (FrameSlot& fs : g.slot) {
        for (EventSet& es : fs.es) for (int i = 0; i < 4; ++i) if (es.ev[i]) (void)hipEventDestroy(es.ev[i]);
        if (fs.stream) (void)hipStreamDestroy(fs.stream);
    }
    { State* me = G; *me = State(); }
}

int crt1_resize(int width, int height)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (width < 16 || height < 16) return CRT_OK; // Renderer.cpp:200
    RCCHK(sync_all());
    return alloc_frame_buffers(width, height);
}

int crt1_set_row_bands(int bandRows, int rank, int nRanks)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (bandRows < CRT_TILE || bandRows % CRT_TILE != 0 || nRanks < 1 || rank < 0 || rank >= nRanks) return CRT_E_BAD_ARGUMENT;
    RCCHK(sync_all());
    g.bandRows = bandRows; g.rank = rank; g.nRanks = nRanks;
    return CRT_OK;
}


int crt1_owned_rows(void)
{
    if (!g.initialized) return 0;
    int rows = 0;
    const int tpb = g.bandRows / CRT_TILE;
    for (int y = 0; y < g.height; ++y) if ((((y / CRT_TILE) / tpb) % g.nRanks) == g.rank) ++rows;
    return rows;
}

int crt1_upload_triangles(const void* tris, size_t byteOffset, size_t bytes)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (bytes == 0) return CRT_OK;
    if (!tris || byteOffset % sizeof(CrtTri) || bytes % sizeof(CrtTri)) return CRT_E_BAD_ARGUMENT;
    if (byteOffset + bytes > g.triCap * sizeof(CrtTri)) return CRT_E_OUT_OF_RANGE;
    RCCHK(quiesce());
    HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(g.rawTris) + byteOffset, tris, bytes, hipMemcpyHostToDevice, g.stream));
    const size_t first = byteOffset / sizeof(CrtTri), count = bytes / sizeof(CrtTri);
    crt_relayout_tris<<<(unsigned)((count + 255) / 256), 256, 0, g.stream>>>(g.rawTris, first, count, g.triHot, g.triCold);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g.stream));
    // How far from the object-space origin a vertex (hence a hit point, hence a bounce-ray origin, hazard H6) can lie: the cull's proven range is
    // checked against this as well as against the root boxes, so that nodes uploaded through crt_upload_bvh_nodes whose boxes do not bound their
    // triangles cannot take bounce origins beyond it. Reduced on the device from the pool itself (r6, ADVICE r5: it used to be a serial host
    // scan per upload and per device state, and could only grow): an upload that appends folds its own range in; one that overwrites triangles
    // uploaded before recomputes over the whole pool, so a far vertex that has been replaced stops counting.
    const bool overwrites = first < g.trisHigh;
    if (first + count > g.trisHigh) g.trisHigh = first + count;
    if (overwrites) HIPCHK(hipMemsetAsync(g.triReachBits, 0, sizeof(uint32_t), g.stream));
    {
        const size_t f0 = overwrites ? 0 : first, n0 = overwrites ? g.trisHigh : count;
        crt_tri_reach_kernel<<<(unsigned)((n0 + 255) / 256), 256, 0, g.stream>>>(g.rawTris, f0, n0, g.triReachBits);
        HIPCHK(hipGetLastError());
    }
    uint32_t bitsHost = 0;
    HIPCHK(hipMemcpyAsync(&bitsHost, g.triReachBits, sizeof bitsHost, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    {
        float f; memcpy(&f, &bitsHost, sizeof f);
        const double far2 = (f == f && f < 3.0e38f) ? (double)f * (1.0 + 1e-6) : 1e300;     // (fp32 sum of three squares: 3 roundings; inf / NaN: unbounded)
        if (far2 != g.triReach2) { g.triReach2 = far2; rebuild_instance_master(); }
    }
    return CRT_OK;
}

int crt1_upload_bvh_nodes(const void* nodes, size_t byteOffset, size_t bytes)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (bytes == 0) return CRT_OK;
    if (!nodes || byteOffset % sizeof(CrtBVHNode) || bytes % sizeof(CrtBVHNode)) return CRT_E_BAD_ARGUMENT;
    if (byteOffset + bytes > g.nodeCap * sizeof(CrtBVHNode)) return CRT_E_OUT_OF_RANGE;
    RCCHK(quiesce());
    HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(g.rawNodes) + byteOffset, nodes, bytes, hipMemcpyHostToDevice, g.stream));
    const uint32_t high = (uint32_t)((byteOffset + bytes) / sizeof(CrtBVHNode));
    if (high > g.nodeCount) g.nodeCount = high;
    return rebuild_bvh_layout();
}

int crt1_upload_bvh_roots(const uint32_t* roots, size_t firstMesh, size_t count)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (count == 0) return CRT_OK;
    if (!roots) return CRT_E_BAD_ARGUMENT;
    if (firstMesh + count > CRT_MAX_MESHES) return CRT_E_OUT_OF_RANGE;
    RCCHK(quiesce());
    HIPCHK(hipMemcpyAsync(g.roots + firstMesh, roots, count * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
    memcpy(g.hRoots + firstMesh, roots, count * sizeof(uint32_t));
    if (firstMesh + count > g.numRoots) g.numRoots = (uint32_t)(firstMesh + count);
    return rebuild_bvh_layout();
}

int crt1_upload_materials(const void* materials, size_t first, size_t count)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (count == 0) return CRT_OK;
    if (!materials) return CRT_E_BAD_ARGUMENT;
    if (first + count > CRT_MAX_MATERIALS) return CRT_E_OUT_OF_RANGE;
    RCCHK(quiesce());
    HIPCHK(hipMemcpyAsync(g.materials + first, materials, count * sizeof(CrtMaterial), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return CRT_OK;
}

int crt1_upload_texture_table(const void* textures, size_t count)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (count == 0) return CRT_OK;
    if (!textures) return CRT_E_BAD_ARGUMENT;
    if (count > CRT_MAX_TEXTURES) return CRT_E_OUT_OF_RANGE;
    RCCHK(quiesce());
    HIPCHK(hipMemcpyAsync(g.textures, textures, count * sizeof(CrtTexture), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return CRT_OK;
}

int crt1_upload_texels(const void* rgb8, size_t byteOffset, size_t bytes)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (bytes == 0) return CRT_OK;
    if (!rgb8) return CRT_E_BAD_ARGUMENT;
    if (byteOffset + bytes > g.texelByteCap) return CRT_E_OUT_OF_RANGE;
    RCCHK(quiesce());
    HIPCHK(hipMemcpyAsync(g.rawTexels + byteOffset, rgb8, bytes, hipMemcpyHostToDevice, g.stream));
    if (byteOffset + bytes > g.texelBytesHigh) g.texelBytesHigh = byteOffset + bytes;
    const size_t firstTexel = byteOffset / 3;
    const size_t endTexel = (byteOffset + bytes) / 3;       // whole texels only; a trailing partial texel waits for its bytes
    if (endTexel > firstTexel) {
        const size_t count = endTexel - firstTexel;
        crt_relayout_texels<<<(unsigned)((count + 255) / 256), 256, 0, g.stream>>>(g.rawTexels, firstTexel, count, g.texels);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    return CRT_OK;
}

int crt1_upload_instances(const void* instances, size_t first, size_t count)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (count == 0) return CRT_OK;
    if (!instances) return CRT_E_BAD_ARGUMENT;
    if (first + count > CRT_MAX_INSTANCES) return CRT_E_OUT_OF_RANGE;
    const CrtMeshInstance* in = static_cast<const CrtMeshInstance*>(instances);
    for (size_t i = 0; i < count; ++i) if (in[i].meshIndex >= CRT_MAX_MESHES) return CRT_E_BAD_ARGUMENT;
    // host only: frames already submitted keep the tables they were submitted with, every later frame (on whichever
    // slot) refreshes its slot's copy on its own stream first -- an animated scene stays pipelined
    memcpy(g.hInstances + first, instances, count * sizeof(CrtMeshInstance));
    if (first + count > g.instHigh) g.instHigh = (uint32_t)(first + count);
    rebuild_instance_master((uint32_t)first, (uint32_t)count, false);      // only the records that were replaced (+ a refit of the instance tree)
    return CRT_OK;
}

// Read back the reference-layout pools (after crt1_build_bvh: the reordered triangles with their centroids, the nodes,
// the roots), e.g. to keep host arenas in step with the device.
int crt1_download_triangles(void* dst, size_t byteOffset, size_t bytes)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (bytes == 0) return CRT_OK;
    if (!dst || byteOffset % sizeof(CrtTri) || bytes % sizeof(CrtTri)) return CRT_E_BAD_ARGUMENT;
    if (byteOffset + bytes > g.triCap * sizeof(CrtTri)) return CRT_E_OUT_OF_RANGE;
    RCCHK(sync_all());
    HIPCHK(hipMemcpy(dst, reinterpret_cast<const char*>(g.rawTris) + byteOffset, bytes, hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt1_download_bvh_nodes(void* dst, size_t byteOffset, size_t bytes)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (bytes == 0) return CRT_OK;
    if (!dst || byteOffset % sizeof(CrtBVHNode) || bytes % sizeof(CrtBVHNode)) return CRT_E_BAD_ARGUMENT;
    if (byteOffset + bytes > g.nodeCap * sizeof(CrtBVHNode)) return CRT_E_OUT_OF_RANGE;
    RCCHK(sync_all());
    HIPCHK(hipMemcpy(dst, reinterpret_cast<const char*>(g.rawNodes) + byteOffset, bytes, hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt1_download_bvh_roots(uint32_t* dst, size_t firstMesh, size_t count)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (count == 0) return CRT_OK;
    if (!dst) return CRT_E_BAD_ARGUMENT;
    if (firstMesh + count > CRT_MAX_MESHES) return CRT_E_OUT_OF_RANGE;
    RCCHK(sync_all());
    HIPCHK(hipMemcpy(dst, g.roots + firstMesh, count * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CRT_OK;
}

} // namespace
