// crt_state.h -- host state of one device session: frame slots, pools, the helpers every other part uses
// Part of the one translation unit crt_shim.hip (included there, in this order: crt_state.h, crt_instances.h, crt_upload.h,
// crt_bvh_driver.h, crt_frame.h, crt_multidev.h); everything here has internal linkage.
#pragma once
// ------------------------------------------------------------------------------------------------
// host state
// ------------------------------------------------------------------------------------------------
namespace {

#define CRT_MAX_FRAMES_IN_FLIGHT 8
#define CRT_MAX_DEVICES 16

struct EventSet {
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    bool pending = false; int flags = 0; bool evRaygen = false, evPost = false;   // timing not yet read back
    unsigned long long seq = 0;
};

struct FrameSlot {
    hipStream_t stream = nullptr;
    EventSet es[2];                            // two sets, so the host may queue a slot's next frame before reading the last one's timing
    unsigned frames = 0;
    float4* out = nullptr;
    float4* aux = nullptr; size_t auxPixels = 0;   // CRT_RENDER_FXAA: the unfiltered frame the filter reads (allocated on first use)
    CrtBounceRay* blockQueue = nullptr; size_t blockQueueCap = 0;   // CRT_KERNEL=block / wavefront: bounce-ray queue, one 64- or 128-record range per workgroup of the primary launch
    uint32_t* wfCount = nullptr; size_t wfCap = 0;                  // CRT_KERNEL=wavefront: per primary wave {continuing rays}, {offset within its XCD}, then the 8 per-XCD totals
    uint32_t* ovf = nullptr; size_t ovfBlocks = 0;   // traversal-stack overflow area of this slot's launches (CrtStack), one block per workgroup
    uint32_t* order = nullptr; uint32_t* len = nullptr; uint32_t* cost = nullptr;   // feedback launch lists
    size_t orderCap = 0; int orderSlots = -1; int orderKey[6] = { 0, 0, 0, 0, 0, 0 };
    bool listsReady = false;                   // the lists for the next frame were already sorted at the end of the last one
    // CRT_RENDER_READBACK: pinned host copy of this slot's frame, queued behind the frame on the slot's stream
    void* hostBuf = nullptr; size_t hostCap = 0, hostBytes = 0; uint32_t* packBuf = nullptr; size_t packCap = 0; hipEvent_t copied = nullptr;
    // This slot's copy of the instance tables (reference-layout records, device records, bounding spheres, instance tree,
    // never-culled list), refreshed on the slot's own stream from the host master when it is stale (ensure_slot_instances):
    // an instance upload never has to wait for the frames in flight, and those frames never see it.
    char* instBlock = nullptr;                 // one device allocation; the four pointers below point into it
    CrtMeshInstance* instances = nullptr; CrtDevInstance* devInstances = nullptr; float4* instBounds = nullptr;
    CrtTlasNode* tlas = nullptr; uint32_t* alwaysList = nullptr; uint32_t tlasNodes = 0, numAlways = 0;
    unsigned long long instVersion = 0;        // 0 = never filled (the master starts at 1)
    uint32_t* mixOrder = nullptr; uint32_t* mixLen = nullptr; size_t mixCap = 0; int mixSlots = -1;   // CRT_RENDER_DIAG_MIX3 launch lists
    char* stagingDev = nullptr;                // the staging block as the device sees it
    char* staging = nullptr; hipEvent_t staged = nullptr;   // pinned staging block and "its copies have been issued and done" event
    // in-process multi-GPU (crt_init_devices): a secondary device records `partDone` behind the copy of its bands into the
    // primary's frame; the primary records `slotDone` behind everything a frame queues on this slot (incl. a read-back)
    hipEvent_t partDone = nullptr, slotDone = nullptr;
    // RGBA8 gather (a multi-device session's CRT_RENDER_UNORM8 frames without FXAA): the WHOLE frame is in packBuf (every device's Trace
    // epilogue stores its pixels' four bytes, the secondaries copy 4 B per pixel into the primary's packBuf); `out` holds only this
    // device's bands until somebody asks for the float frame (expand_rgba8_frame: x = byte / 255, exactly what the epilogue stored)
    bool frameIs8 = false;
};

struct State {
    bool initialized = false;
    int device = -1;
    char deviceName[256] = { 0 };
    // Frame slots: a synchronous frame always uses slot 0; CRT_RENDER_ASYNC frames rotate over the first nSlots slots
    // (own stream, output buffer, launch lists and events each), so the tail of one frame overlaps the next.
    FrameSlot slot[CRT_MAX_FRAMES_IN_FLIGHT]; int nSlots = 3;
    hipStream_t stream = nullptr;              // == slot[0].stream: uploads, queries, diagnostics
    int cur = 0;                               // slot of the most recently submitted frame
    int readbackRing[CRT_MAX_FRAMES_IN_FLIGHT] = { -1, -1, -1, -1, -1, -1, -1, -1 }; unsigned readbackCount = 0;   // slots of the latest CRT_RENDER_READBACK frames
    unsigned asyncSeq = 0; bool othersBusy = false;   // frames possibly running on slots > 0
    // Start-up stagger of a burst of frames in flight: frames submitted to an idle device start together, run in lockstep and have
    // their long-ray tails at the same time -- exactly what frames in flight are there to avoid -- until the slots drift apart (three
    // frames take 1.17 ms to fill the pipeline where steady state delivers 4.3). The first frame a slot runs after the session was
    // idle is therefore held back on its stream by slot x (last frame latency / slots) by a one-wave timer kernel.
    float pipelinedLatencyMs = 0.0f;     // latency of the newest plain frame-in-flight timed so far (what the stagger is derived from; 0 = none yet)
    unsigned burstFrames = 0; int staggerUs = -1;   // frames submitted since the device was last known idle; CRT_STAGGER_US: -1 = automatic, 0 = off, n = n us per slot
    // Automatic = only for a caller that is known to stream: the burst before this one ran longer than the slot count. A caller that
    // submits two or three frames and then reads never reaches steady state and would only pay the hold-back as latency.
    unsigned prevBurstFrames = 0; unsigned long long staggeredFrames = 0;
    int width = 0, height = 0;
    int bandRows = 16, rank = 0, nRanks = 1;
    // raw (reference-layout) device copies
    CrtTri* rawTris = nullptr; CrtBVHNode* rawNodes = nullptr; uint32_t* roots = nullptr; uint8_t* rawTexels = nullptr;
    // CDNA4 layouts
    float4* pairs = nullptr; float* triHot = nullptr; uint4* triCold = nullptr; uint32_t* bigLeaf = nullptr;
    uint32_t* rootRefs = nullptr; uint32_t* texels = nullptr;
    CrtMaterial* materials = nullptr; CrtTexture* textures = nullptr;
    // host master of everything derived from the instance table (rebuild_instance_master); slots copy it when stale
    float4 hBounds[CRT_MAX_INSTANCES]; CrtTlasNode hTlas[2 * CRT_MAX_INSTANCES]; uint32_t hAlways[CRT_MAX_INSTANCES];
    uint32_t hTlasNodes = 0, hNumAlways = 0; unsigned long long instVersion = 1;
    // incremental rebuild_instance_master (r6): the reach the limits were checked against, and what the instance tree was last BUILT for (it is refitted while
    // the set of cullable instances stays the same)
    double hReach = 0.0; bool hTlasBuilt = false; uint32_t hTlasBuiltHigh = 0, hTlasRefits = 0; double hTlasBuiltRadii = 0.0; uint8_t hTlasMember[CRT_MAX_INSTANCES] = { 0 };
    unsigned long long hTlasBuilds = 0;        // median-split builds so far (crt_debug_tlas_stats)
    CrtBVHNode hRootNodes[CRT_MAX_MESHES]; bool hHaveRoot[CRT_MAX_MESHES];   // root node of every mesh, cached at BVH upload
    CrtBVHNode hRootKids[CRT_MAX_MESHES][2]; bool hHaveKids[CRT_MAX_MESHES];  // ... and the root's two children: their boxes are what an entering ray is tested against
    // Range of ray origins for which the instance cull is provably exact (derivation: crt_device.h above sphere_culls):
    // per instance and the smallest over the cullable ones; a frame / query whose origins lie beyond it runs with `noCullBounds`.
    float hCullOriginLimit[CRT_MAX_INSTANCES]; float cullOriginLimit = 0.0f; float bounceOriginReach = 0.0f;
    double triReach2 = 0.0;                    // largest squared distance of a vertex in the triangle pool from its object-space origin (crt1_upload_triangles)
    uint32_t* triReachBits = nullptr;          // device: that maximum as float bits (crt_tri_reach_kernel)
    float4* noCullBounds = nullptr;            // device: CRT_MAX_INSTANCES x (0, 0, 0, -1) = "never cull"
    unsigned long long noCullFrames = 0;       // frames and queries that ran without the cull for that reason
    CrtMeshInstance hInstances[CRT_MAX_INSTANCES]; uint32_t hRoots[CRT_MAX_MESHES]; uint32_t instHigh = 0;
    float* rays = nullptr;
    unsigned long long* counters = nullptr; int* err = nullptr;
    unsigned long long* stamps = nullptr; size_t stampBytes = 0, stampWaves = 0;
    int numCUs = 0;
    int forceTlas = -1;   // CRT_TLAS=0/1: force the linear / tree candidate search (tests); default: by instance count
    int feedbackAsync = 0; int feedback = 1; int maxSplit = CRT_MAX_SPLIT, maxSplitPipelined = CRT_MAX_SPLIT_PIPELINED;
    // feedback lists while the view changes: rank a tile by max(own cost, costSpread x heaviest of its 8 neighbours) -- next
    // frame's heavy tiles are this frame's or the ones next to them. A view that stood still for a frame is ranked by own cost.
    float costSpread = 0.8f;
    float lastView[35] = { 0 }; unsigned long long lastViewInst = 0; bool viewMoved = false;   // camera matrices + position / instance version of the last sorted frame
    float splitBeta = CRT_SPLIT_BETA, splitBetaAsync = CRT_SPLIT_BETA_ASYNC;   // split a tile whose wave would run longer than beta x the XCD's time for the frame
    int refill = 0;                            // CRT_KERNEL=refill / block: 1 = in-tile lane refill, 2 = phase-separated block compaction (crt_refill.h), every frame (one they cannot render is refused)
    int ldstop = 0;                            // CRT_KERNEL=ldstop: four-wave workgroups sharing an LDS copy of the tree tops (crt_ldstop.h)
    float4* topPairs = nullptr; uint32_t* topRootRefs = nullptr;   // the tree-top table (CRT_TOP_PAIRS records) and every mesh's entry into it, rebuilt with the BVH layout
    int wavefront = 0;                         // CRT_KERNEL=wavefront: one launch per bounce, ordered ballot compaction in between (crt_kernels.h)
    char lastKernel[128] = { 0 };              // crt_debug_last_kernel: the Trace launch(es) of the most recently submitted frame
    void* queryBuf = nullptr; size_t queryBytes = 0;
    void* buildBuf = nullptr; size_t buildBytes = 0;          // crt_build_bvh scratch
    std::vector<CrtBuildCtl> buildReplay; unsigned long long buildReplayKey = 0;   // CRT_DEBUG_BVH_REPLAY (crt_bvh_driver.h): the level records of the last build
    unsigned buildLaunches = 0, buildLevels = 0;   // crt_debug_build_stats: kernel launches and levels of the last crt_build_bvh (before the re-layout)
    bool buildNoSpin = false;                                   // the spin on buildCtlHost timed out once: synchronise the stream per level instead
    CrtBuildCtlHost* buildCtlHost = nullptr; uint32_t buildSeq = 0;   // pinned: the per-level control record the builder publishes (crt_bvh_publish)
    CrtTri* buildTris = nullptr;                               // crt_build_bvh: second triangle pool (same indexing as rawTris)
    size_t triCap = 0, nodeCap = 0, texelByteCap = 0;
    uint32_t nodeCount = 0, numRoots = 0; size_t texelBytesHigh = 0; size_t trisHigh = 0;
    bool sceneValid = true;
    double msSum[4] = { 0, 0, 0, 0 }; unsigned long long framesTimed = 0;   // crt_frame_time_stats
    float ms[4] = { 0, 0, 0, 0 }; unsigned long long msSeq = 0, frameSeq = 0;   // timing of the newest frame read back so far
    hipEvent_t statStart = nullptr; bool statStartArmed = true, statStartValid = false; unsigned long long statStartSeq = 0; double statExtent = 0, statFirstMs = 0;
    CrtCounters lastCounters; unsigned long long lastCulled = 0;
    double frameLog[512]; unsigned frameLogN = 0;      // crt_debug_read_frame_times: {start, end} ms after statStart of the frames since the last reset
    // in-process multi-GPU: this device renders band `rank` of `nRanks`; `primary` (rank 0) owns the frame that is read
    State* primary = nullptr; State* group[CRT_MAX_DEVICES] = { nullptr }; int groupSize = 1;
    int gather8 = 1;                           // CRT_GATHER_RGBA8=0: gather float4 bands even for CRT_RENDER_UNORM8 frames (A/B, tests)
};
// One State per device (crt_init: one; crt_init_devices: one per GPU). Every function below works on "the current
// device's state" through `g`; the dispatch layer at the end of the file selects it (and the HIP device) per call, on the
// calling thread or on a per-device worker thread.
thread_local State* G = nullptr;
#define g (*G)

// a spinning host thread's pause (the level read-back of crt_build_bvh, the per-device workers): x86 `pause`, otherwise a compiler barrier
static inline void crt_cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}

#define CRT_NUM_COUNTERS 15
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { return (int)e_; } } while (0)
#define RCCHK(x) do { int r_ = (x); if (r_ != CRT_OK) { return r_; } } while (0)

// Wait for frames still running on the second slot before anything touches shared device state.
int quiesce()
{
    if (g.othersBusy) {
        for (int i = 1; i < g.nSlots; ++i) HIPCHK(hipStreamSynchronize(g.slot[i].stream));
        g.othersBusy = false;
    }
    return CRT_OK;
}

int sync_all()
{
    HIPCHK(hipStreamSynchronize(g.slot[0].stream));
    RCCHK(quiesce());
    if (g.burstFrames) g.prevBurstFrames = g.burstFrames;
    g.burstFrames = 0;                          // every slot is idle: the next pipelined frames start a burst
    return CRT_OK;
}

int owned_tile_rows()
{
    const int totalTileRows = (g.height + CRT_TILE - 1) / CRT_TILE;
    const int tpb = g.bandRows / CRT_TILE;
    int n = 0;
    for (int r = 0; r < totalTileRows; ++r) if (((r / tpb) % g.nRanks) == g.rank) ++n;
    return n;
}

void fill_frame(CrtFrame& F, const CrtTraceArgs* args, const float* invView, const float* invProj)
{
    memset(&F, 0, sizeof F);
    if (invView) memcpy(F.invView, invView, 64);
    if (invProj) memcpy(F.invProj, invProj, 64);
    if (args) {
        memcpy(F.camPos, args->cameraPos, 12);
        F.lightY = (float)sin((double)args->sunAngle);
        F.lightZ = (float)cos((double)args->sunAngle);
    }
    F.width = g.width; F.height = g.height;
    F.tilesX = (g.width + CRT_TILE - 1) / CRT_TILE;
    F.ownedTileRows = owned_tile_rows();
    F.gridBlocks = ((F.ownedTileRows + 7) / 8) * 8 * F.tilesX;
    F.slotsPerXcd = F.gridBlocks / 8;
    F.order = nullptr; F.cost = nullptr; F.listLen = nullptr; F.listCap = F.slotsPerXcd;
    F.tileRowsPerBand = g.bandRows / CRT_TILE;
    F.rank = g.rank; F.nRanks = g.nRanks;
}

// noCull: the rays of this launch may start beyond the range the instance cull is proven for (State::cullOriginLimit): every
// instance is a candidate for every ray (all-never bounds table, no instance tree)
void fill_scene(CrtDevScene& S, uint32_t numInstances, const FrameSlot& fs, bool noCull = false)
{
    S.pairs = g.pairs; S.triHot = g.triHot; S.triCold = g.triCold; S.bigLeaf = g.bigLeaf; S.rootRefs = g.rootRefs; S.stackOverflow = fs.ovf;
    S.instances = fs.instances; S.devInstances = fs.devInstances; S.instBounds = fs.instBounds; S.materials = g.materials; S.textures = g.textures; S.texels = g.texels;
    S.numTexels = (int)((g.texelBytesHigh + 2) / 3);
    if (S.numTexels < 1) S.numTexels = 1;
    S.numInstances = numInstances;
    S.tlas = fs.tlas; S.tlasNodes = fs.tlasNodes; S.alwaysList = fs.alwaysList; S.numAlways = fs.numAlways;
    S.topPairs = g.topPairs;
    if (noCull) { S.instBounds = g.noCullBounds; S.tlas = nullptr; S.tlasNodes = 0; S.alwaysList = nullptr; S.numAlways = 0; g.noCullFrames++; }
}
// true when a ray origin this far from the world origin is outside the proven range (NaN counts as outside)
bool beyond_cull_range(double originNorm) { return !(originNorm <= (double)g.cullOriginLimit); }

// The traversal-stack overflow area of a slot must hold one block per workgroup of its largest launch.
int ensure_overflow(FrameSlot& fs, size_t blocks)
{
    if (blocks <= fs.ovfBlocks) return CRT_OK;
    HIPCHK(hipStreamSynchronize(fs.stream));
    if (fs.ovf) (void)hipFree(fs.ovf);
    fs.ovf = nullptr; fs.ovfBlocks = 0;
    HIPCHK(hipMalloc(&fs.ovf, blocks * CRT_OVF_WORDS_PER_BLOCK * sizeof(uint32_t)));   // never initialised: entries are written before they are read
    fs.ovfBlocks = blocks;
    return CRT_OK;
}

// New frame buffers are allocated first and swapped in only when every allocation succeeded: a failed resize leaves
// the old frame size fully usable (crt_resize returns the error).
int alloc_frame_buffers(int w, int h)
{
    const size_t pixels = (size_t)w * (size_t)h;
    float* rays = nullptr; float4* outs[CRT_MAX_FRAMES_IN_FLIGHT] = {};
    hipError_t e = hipMalloc(&rays, sizeof(float) * 3 * pixels);
    for (int i = 0; i < g.nSlots && e == hipSuccess; ++i) {   // slots past nSlots are never rendered into
        e = hipMalloc(&outs[i], sizeof(float4) * pixels);
        if (e == hipSuccess) e = hipMemsetAsync(outs[i], 0, sizeof(float4) * pixels, g.stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e != hipSuccess) {
        if (rays) (void)hipFree(rays);
        for (float4* o : outs) if (o) (void)hipFree(o);
        return (int)e;
    }
    if (g.rays) (void)hipFree(g.rays);
    g.rays = rays;
    for (int i = 0; i < CRT_MAX_FRAMES_IN_FLIGHT; ++i) {
        FrameSlot& fs = g.slot[i];
        if (fs.out) (void)hipFree(fs.out);
        fs.out = outs[i];
        if (fs.aux) { (void)hipFree(fs.aux); fs.aux = nullptr; fs.auxPixels = 0; }
    }
    g.width = w; g.height = h; g.readbackCount = 0; g.pipelinedLatencyMs = 0.0f;
    return CRT_OK;
}

// Copies the pixel rows this rank owns (16-row bands dealt round-robin, crt_set_row_bands) from one frame-shaped buffer to
// the same place in another: one strided 2-D copy (a band is contiguous, bands repeat every nRanks * bandRows rows) plus at
// most one partial band at the bottom. Used for band-only read-backs and for the in-process multi-GPU gather.
// The rows rank `rank` of `nRanks` owns, as one strided block list: `fullBands` bands of `bandRows` rows starting at row
// `firstRow` and repeating every bandRows * nRanks rows, plus `tailRows` rows of a last, partial band at row `tailRow`.
struct BandPlan { int firstRow, fullBands, tailRow, tailRows; };
BandPlan band_plan(int height, int bandRows, int rank, int nRanks)
{
    BandPlan p = { rank * bandRows, 0, 0, 0 };
    const int period = bandRows * nRanks;
    if (p.firstRow >= height) { p.tailRow = height; return p; }
    const int x = height - p.firstRow;
    p.fullBands = x / period + ((x % period) >= bandRows ? 1 : 0);
    p.tailRow = p.firstRow + p.fullBands * period;
    p.tailRows = p.tailRow < height ? height - p.tailRow : 0;
    return p;
}

int copy_owned_rows_async(void* dstFrame, const void* srcFrame, size_t bytesPerPixel, hipMemcpyKind kind, hipStream_t stream, bool allRows = false)
{
    const size_t rowBytes = (size_t)g.width * bytesPerPixel;
    if (g.nRanks == 1 || allRows) return (int)hipMemcpyAsync(dstFrame, srcFrame, rowBytes * (size_t)g.height, kind, stream);
    const BandPlan p = band_plan(g.height, g.bandRows, g.rank, g.nRanks);
    const size_t bandBytes = rowBytes * (size_t)g.bandRows, pitch = bandBytes * (size_t)g.nRanks;
    char* d = static_cast<char*>(dstFrame) + (size_t)p.firstRow * rowBytes; const char* sp = static_cast<const char*>(srcFrame) + (size_t)p.firstRow * rowBytes;
    if (p.fullBands > 0) HIPCHK(hipMemcpy2DAsync(d, pitch, sp, pitch, bandBytes, (size_t)p.fullBands, kind, stream));
    if (p.tailRows > 0) {
        const size_t off = (size_t)p.fullBands * pitch;
        HIPCHK(hipMemcpyAsync(d + off, sp + off, rowBytes * (size_t)p.tailRows, kind, stream));
    }
    return CRT_OK;
}

} // namespace
