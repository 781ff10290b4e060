// crt_shim.hip -- host state and the C-ABI (include/crt_api.h) of the MI355X ray-trace path; kernels in crt_kernels.h.
//
// Replaces the OpenCL side of the reference's Renderer.cpp / ResourceManager.cpp: device pools,
// uploads, the per-frame RayGen -> Trace -> PostProcess launch (Renderer.cpp:305-375). Uploads
// arrive in the reference's struct layouts and are re-laid-out on the device (crt_device.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/crt_api.h"
#include "crt_kernels.h"
#include "crt_relayout.h"
#include "crt_bvh_build.h"
#include <vector>
#include <algorithm>
#include <utility>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <atomic>

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
    uint32_t* ovf = nullptr; size_t ovfBlocks = 0;   // traversal-stack overflow area of this slot's launches (CrtStack), one block per workgroup
    uint32_t* order = nullptr; uint32_t* len = nullptr; uint32_t* cost = nullptr;   // feedback launch lists
    size_t orderCap = 0; int orderSlots = -1; int orderKey[6] = { 0, 0, 0, 0, 0, 0 };
    bool listsReady = false;                   // the lists for the next frame were already sorted at the end of the last one
    // CRT_RENDER_READBACK: pinned host copy of this slot's frame, queued behind the frame on the slot's stream
    void* hostBuf = nullptr; size_t hostCap = 0, hostBytes = 0; uint32_t* packBuf = nullptr; size_t packCap = 0; hipEvent_t copied = nullptr;
    // This slot's copy of the instance tables (reference-layout records, device records, bounding spheres, instance tree,
    // never-culled list), refreshed on the slot's own stream from the host master when it is stale (ensure_slot_instances):
    // an instance upload never has to wait for the frames in flight, and those frames never see it.
    CrtMeshInstance* instances = nullptr; CrtDevInstance* devInstances = nullptr; float4* instBounds = nullptr;
    CrtTlasNode* tlas = nullptr; uint32_t* alwaysList = nullptr; uint32_t tlasNodes = 0, numAlways = 0;
    unsigned long long instVersion = 0;        // 0 = never filled (the master starts at 1)
    uint32_t* mixOrder = nullptr; uint32_t* mixLen = nullptr; size_t mixCap = 0; int mixSlots = -1;   // CRT_RENDER_DIAG_MIX3 launch lists
    char* staging = nullptr; hipEvent_t staged = nullptr;   // pinned staging block and "its copies have been issued and done" event
    // in-process multi-GPU (crt_init_devices): a secondary device records `partDone` behind the copy of its bands into the
    // primary's frame; the primary records `slotDone` behind everything a frame queues on this slot (incl. a read-back)
    hipEvent_t partDone = nullptr, slotDone = nullptr;
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
    CrtBVHNode hRootNodes[CRT_MAX_MESHES]; bool hHaveRoot[CRT_MAX_MESHES];   // root node of every mesh, cached at BVH upload
    CrtBVHNode hRootKids[CRT_MAX_MESHES][2]; bool hHaveKids[CRT_MAX_MESHES];  // ... and the root's two children: their boxes are what an entering ray is tested against
    // Range of ray origins for which the instance cull is provably exact (derivation: crt_device.h above sphere_culls):
    // per instance and the smallest over the cullable ones; a frame / query whose origins lie beyond it runs with `noCullBounds`.
    float hCullOriginLimit[CRT_MAX_INSTANCES]; float cullOriginLimit = 0.0f; float bounceOriginReach = 0.0f;
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
    int wavefront = 0; CrtBounceRay* bounceQueue = nullptr; uint32_t* bounceCount = nullptr; size_t bounceCap = 0;
    void* queryBuf = nullptr; size_t queryBytes = 0;
    void* buildBuf = nullptr; size_t buildBytes = 0;          // crt_build_bvh scratch
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
};
// One State per device (crt_init: one; crt_init_devices: one per GPU). Every function below works on "the current
// device's state" through `g`; the dispatch layer at the end of the file selects it (and the HIP device) per call, on the
// calling thread or on a per-device worker thread.
thread_local State* G = nullptr;
#define g (*G)

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
    float* rays = nullptr; CrtBounceRay* queue = nullptr; float4* outs[CRT_MAX_FRAMES_IN_FLIGHT] = {};
    hipError_t e = hipMalloc(&queue, sizeof(CrtBounceRay) * pixels);
    if (e == hipSuccess) e = hipMalloc(&rays, sizeof(float) * 3 * pixels);
    for (int i = 0; i < g.nSlots && e == hipSuccess; ++i) {   // slots past nSlots are never rendered into
        e = hipMalloc(&outs[i], sizeof(float4) * pixels);
        if (e == hipSuccess) e = hipMemsetAsync(outs[i], 0, sizeof(float4) * pixels, g.stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e != hipSuccess) {
        if (queue) (void)hipFree(queue);
        if (rays) (void)hipFree(rays);
        for (float4* o : outs) if (o) (void)hipFree(o);
        return (int)e;
    }
    if (g.rays) (void)hipFree(g.rays);
    if (g.bounceQueue) (void)hipFree(g.bounceQueue);
    g.rays = rays; g.bounceQueue = queue; g.bounceCap = pixels;
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

int cache_root_nodes();
void rebuild_instance_master();

int rebuild_bvh_layout()
{
    HIPCHK(hipMemsetAsync(g.err, 0, sizeof(int), g.stream));
    if (g.nodeCount) {
        crt_relayout_nodes<<<(g.nodeCount + 255) / 256, 256, 0, g.stream>>>(g.rawNodes, g.nodeCount, (uint32_t)g.triCap, g.pairs, g.bigLeaf, g.err);
        HIPCHK(hipGetLastError());
    }
    crt_make_root_refs<<<(CRT_MAX_MESHES + 255) / 256, 256, 0, g.stream>>>(g.rawNodes, g.nodeCount, (uint32_t)g.triCap, g.roots, g.numRoots, g.rootRefs, g.bigLeaf, g.err);
    HIPCHK(hipGetLastError());
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

// Host master of the instance-derived tables: bounding spheres, the instance tree, the never-culled list. Pure host work
// (a few tens of microseconds for 401 instances); bumps the version the frame slots compare against.
void rebuild_instance_master()
{
    float4* bounds = g.hBounds;
    const CrtBVHNode* rootNodes = g.hRootNodes;
    const bool* haveRoot = g.hHaveRoot;
    // Bounce, shadow and refraction rays start at object-space hit points of the hit instance used as world-space origins
    // (hazard H6): no farther from the world origin than the farthest corner of any mesh's root (or root children's) box, plus the
    // 0.01 offset along the normal. An instance that cannot be culled exactly for origins that far out is never culled.
    double reach = 0.0;
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
    g.bounceOriginReach = (float)reach;
    double minLimit = 1e30;
    // test hook (CRT_DEBUG_HOOKS=1 only): CRT_DEBUG_CULL_RANGE_SCALE=k multiplies every O_i -- tools/fuzz_cull.py uses it to measure how far
    // beyond the proven range the cull stays exact in practice (the derivation is a worst-case bound)
    double rangeScale = 1.0;
    { const char* h = getenv("CRT_DEBUG_HOOKS"); const char* k = getenv("CRT_DEBUG_CULL_RANGE_SCALE"); if (h && atoi(h) != 0 && k && atof(k) > 0.0) rangeScale = atof(k); }
    const double U = 5.9604644775390625e-8, G3 = 3.0 * U / (1.0 - 3.0 * U), G4 = 4.0 * U / (1.0 - 4.0 * U), K = 2.8e-6;
    for (uint32_t i = 0; i < CRT_MAX_INSTANCES; ++i) {
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
        if (!invert4(inv, fwd)) continue;
        auto xform = [&](double x, double y, double z, double* o) {
            for (int c = 0; c < 3; ++c) o[c] = x * fwd[0 + c] + y * fwd[4 + c] + z * fwd[8 + c] + fwd[12 + c];
        };
        // the box around the root's two child boxes (= the root box for a tree from BuildBVH)
        const CrtBVHNode* kid = g.hRootKids[inst.meshIndex];
        double lo[3], hi[3];
        for (int a = 0; a < 3; ++a) { lo[a] = fmin((double)kid[0].aabbMin[a], (double)kid[1].aabbMin[a]); hi[a] = fmax((double)kid[0].aabbMax[a], (double)kid[1].aabbMax[a]); }
        double cw[3];
        xform(0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2]), cw);
        double r = 0.0;
        for (int k = 0; k < 8; ++k) {
            double p[3];
            xform((k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2], p);
            const double dx = p[0] - cw[0], dy = p[1] - cw[1], dz = p[2] - cw[2];
            const double dist = sqrt(dx * dx + dy * dy + dz * dz);
            if (dist > r) r = dist;
        }
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
        if (limit < minLimit) minLimit = limit;
        bounds[i] = b;
    }
    g.cullOriginLimit = (float)fmin(minLimit * (1.0 - 1e-6), 3e38);
    // Instance tree for scenes with many instances (closest_hit<..., TLAS>): median-split binary tree over the cullable
    // instances' spheres, node sphere = centre and half diagonal of the box around its children's spheres. Instances
    // that are never culled go to a separate ascending list.
    {
        CrtTlasNode* nodes = g.hTlas;
        uint32_t* always = g.hAlways;
        uint32_t nAlways = 0, nLeaves = 0, nNodes = 0;
        uint32_t leaves[CRT_MAX_INSTANCES];
        // only instances that were uploaded; a frame that asks for more (never-uploaded, all-zero records) uses the linear loop
        for (uint32_t i = 0; i < g.instHigh; ++i) { if (bounds[i].w < 0.0f) always[nAlways++] = i; else leaves[nLeaves++] = i; }
        struct Range { uint32_t lo, hi, node; };
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
                const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                // (a node's sphere holds >= 2 instance spheres, so its radius is >= sqrt 3 x theirs and its share of the slack covers their
                // Delta: crt_device.h (5); the fp32 centre's rounding goes into the radius as for the instances)
                const double nc[3] = { 0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2]) };
                const float ncf[3] = { (float)nc[0], (float)nc[1], (float)nc[2] };
                const double nex = nc[0] - (double)ncf[0], ney = nc[1] - (double)ncf[1], nez = nc[2] - (double)ncf[2];
                n.sphere = make_float4(ncf[0], ncf[1], ncf[2],
                                       (float)((0.5 * sqrt(dx * dx + dy * dy + dz * dz) * (1.0 + 1e-5) + sqrt(nex * nex + ney * ney + nez * nez)) * (1.0 + 1e-6)));
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
    }
    g.instVersion++;
}

// Offsets of the tables inside a slot's pinned staging block
constexpr size_t kStageInst = 0;
constexpr size_t kStageBounds = (kStageInst + CRT_MAX_INSTANCES * sizeof(CrtMeshInstance) + 255) & ~(size_t)255;
constexpr size_t kStageTlas = (kStageBounds + CRT_MAX_INSTANCES * sizeof(float4) + 255) & ~(size_t)255;
constexpr size_t kStageAlways = (kStageTlas + 2 * CRT_MAX_INSTANCES * sizeof(CrtTlasNode) + 255) & ~(size_t)255;
constexpr size_t kStageBytes = kStageAlways + CRT_MAX_INSTANCES * sizeof(uint32_t);

// Brings a slot's instance tables up to the host master, on the slot's own stream, before a frame (or query) uses them.
int ensure_slot_instances(FrameSlot& fs)
{
    if (fs.instVersion == g.instVersion) return CRT_OK;
    HIPCHK(hipEventSynchronize(fs.staged));                        // the previous refresh no longer reads the staging block
    memcpy(fs.staging + kStageInst, g.hInstances, CRT_MAX_INSTANCES * sizeof(CrtMeshInstance));
    memcpy(fs.staging + kStageBounds, g.hBounds, CRT_MAX_INSTANCES * sizeof(float4));
    if (g.hTlasNodes) memcpy(fs.staging + kStageTlas, g.hTlas, g.hTlasNodes * sizeof(CrtTlasNode));
    if (g.hNumAlways) memcpy(fs.staging + kStageAlways, g.hAlways, g.hNumAlways * sizeof(uint32_t));
    HIPCHK(hipMemcpyAsync(fs.instances, fs.staging + kStageInst, CRT_MAX_INSTANCES * sizeof(CrtMeshInstance), hipMemcpyHostToDevice, fs.stream));
    HIPCHK(hipMemcpyAsync(fs.instBounds, fs.staging + kStageBounds, CRT_MAX_INSTANCES * sizeof(float4), hipMemcpyHostToDevice, fs.stream));
    if (g.hTlasNodes) HIPCHK(hipMemcpyAsync(fs.tlas, fs.staging + kStageTlas, g.hTlasNodes * sizeof(CrtTlasNode), hipMemcpyHostToDevice, fs.stream));
    if (g.hNumAlways) HIPCHK(hipMemcpyAsync(fs.alwaysList, fs.staging + kStageAlways, g.hNumAlways * sizeof(uint32_t), hipMemcpyHostToDevice, fs.stream));
    HIPCHK(hipEventRecord(fs.staged, fs.stream));
    crt_relayout_instances<<<(CRT_MAX_INSTANCES + 255) / 256, 256, 0, fs.stream>>>(fs.instances, g.rootRefs, CRT_MAX_INSTANCES, fs.devInstances);
    HIPCHK(hipGetLastError());
    fs.tlasNodes = g.hTlasNodes; fs.numAlways = g.hNumAlways; fs.instVersion = g.instVersion;
    return CRT_OK;
}

// Event timing is read back lazily: when the slot is about to be reused (which also bounds the frames in flight to
// one per slot), or when somebody asks. Synchronous frames are complete by then, so this never blocks them.
int collect_set(EventSet& es)
{
    if (!es.pending) return CRT_OK;
    hipEvent_t* ev = es.ev;
    hipEvent_t traceStart = es.evRaygen ? ev[1] : ev[0], frameEnd = es.evPost ? ev[3] : ev[2];
    HIPCHK(hipEventSynchronize(frameEnd));
    float ms[4] = { 0, 0, 0, 0 };
    HIPCHK(hipEventElapsedTime(&ms[0], ev[0], frameEnd));
    if (es.evRaygen) HIPCHK(hipEventElapsedTime(&ms[1], ev[0], ev[1]));
    HIPCHK(hipEventElapsedTime(&ms[2], traceStart, ev[2]));
    if (es.evPost) HIPCHK(hipEventElapsedTime(&ms[3], ev[2], ev[3]));
    for (int k = 0; k < 4; ++k) g.msSum[k] += (double)ms[k];
    g.framesTimed++;
    if (g.statStartValid && es.seq >= g.statStartSeq) {
        float ext = 0;
        HIPCHK(hipEventElapsedTime(&ext, g.statStart, frameEnd));
        if ((double)ext > g.statExtent) g.statExtent = (double)ext;
        if (es.seq == g.statStartSeq) g.statFirstMs = (double)ext;      // the first frame of the extent: fill time of the pipeline
        if (g.frameLogN < 256) {
            float st = 0;
            if (hipEventElapsedTime(&st, g.statStart, ev[0]) == hipSuccess) { g.frameLog[2 * g.frameLogN] = (double)st; g.frameLog[2 * g.frameLogN + 1] = (double)ext; g.frameLogN++; }
        }
    }
    if (es.seq >= g.msSeq) { memcpy(g.ms, ms, sizeof ms); g.msSeq = es.seq; }
    if ((es.flags & CRT_RENDER_ASYNC) && !(es.flags & (CRT_RENDER_COUNTERS | CRT_RENDER_STAMPS | CRT_RENDER_WRITE_RAYS))) g.pipelinedLatencyMs = ms[0];
    if (es.flags & CRT_RENDER_COUNTERS) {
        unsigned long long c[CRT_NUM_COUNTERS];
        HIPCHK(hipMemcpy(c, g.counters, sizeof c, hipMemcpyDeviceToHost));
        CrtCounters& o = g.lastCounters;
        o.rays = c[0]; o.primary = c[1]; o.secondary = c[2]; o.hits = c[3]; o.misses = c[4]; o.traversals = c[5];
        o.pops = c[6]; o.innerVisits = c[7]; o.triTests = c[8]; o.capHits = c[9]; o.stackOverflows = c[10]; o.maxStack = c[11];
        o.shadowRays = c[12]; o.shadowHits = c[13]; g.lastCulled = c[14];
    }
    es.pending = false;
    return CRT_OK;
}

int collect_timing()
{
    for (int i = 0; i < g.nSlots; ++i) {
        FrameSlot& fs = g.slot[i];
        const int older = fs.es[0].seq <= fs.es[1].seq ? 0 : 1;
        RCCHK(collect_set(fs.es[older]));
        RCCHK(collect_set(fs.es[older ^ 1]));
    }
    return CRT_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
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
        HIPCHK(hipMalloc(&fs.instances, CRT_MAX_INSTANCES * sizeof(CrtMeshInstance)));
        HIPCHK(hipMalloc(&fs.devInstances, CRT_MAX_INSTANCES * sizeof(CrtDevInstance)));
        HIPCHK(hipMalloc(&fs.instBounds, CRT_MAX_INSTANCES * sizeof(float4)));
        HIPCHK(hipMalloc(&fs.tlas, 2 * CRT_MAX_INSTANCES * sizeof(CrtTlasNode)));
        HIPCHK(hipMalloc(&fs.alwaysList, CRT_MAX_INSTANCES * sizeof(uint32_t)));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&fs.staging), kStageBytes, hipHostMallocDefault));
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
    HIPCHK(hipMalloc(&g.rawNodes, g.nodeCap * sizeof(CrtBVHNode)));
    HIPCHK(hipMalloc(&g.roots, CRT_MAX_MESHES * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.rawTexels, g.texelByteCap + 16));
    HIPCHK(hipMalloc(&g.pairs, (g.nodeCap / 2 + 1) * 4 * sizeof(float4)));
    HIPCHK(hipMalloc(&g.triHot, g.triCap * 9 * sizeof(float)));
    HIPCHK(hipMalloc(&g.triCold, g.triCap * 2 * sizeof(uint4)));
    HIPCHK(hipMalloc(&g.bigLeaf, (g.triCap + 1) * sizeof(uint32_t)));
    HIPCHK(hipMemset(g.bigLeaf + g.triCap, 0, sizeof(uint32_t)));      // crt_empty_ref: a leaf of zero triangles
    HIPCHK(hipMalloc(&g.rootRefs, CRT_MAX_MESHES * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.texels, (g.texelByteCap / 3 + 2) * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.materials, CRT_MAX_MATERIALS * sizeof(CrtMaterial)));
    HIPCHK(hipMalloc(&g.textures, CRT_MAX_TEXTURES * sizeof(CrtTexture)));
    HIPCHK(hipMalloc(&g.counters, CRT_NUM_COUNTERS * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&g.err, sizeof(int)));
    HIPCHK(hipMalloc(&g.bounceCount, sizeof(uint32_t)));
    {   // the "never cull" bounds table of frames whose rays start beyond the cull's proven range
        static float4 never[CRT_MAX_INSTANCES];
        for (float4& b : never) b = make_float4(0.f, 0.f, 0.f, -1.0f);
        HIPCHK(hipMalloc(&g.noCullBounds, sizeof never));
        HIPCHK(hipMemcpy(g.noCullBounds, never, sizeof never, hipMemcpyHostToDevice));
    }
    g.numCUs = prop.multiProcessorCount;
    { const char* e = getenv("CRT_KERNEL"); g.wavefront = (e && strcmp(e, "wavefront") == 0); }  // default: megakernel (faster)
    { const char* e = getenv("CRT_SPLIT_BETA"); g.splitBeta = e ? (float)atof(e) : CRT_SPLIT_BETA; }
    { const char* e = getenv("CRT_SPLIT_BETA_ASYNC"); g.splitBetaAsync = e ? (float)atof(e) : CRT_SPLIT_BETA_ASYNC; }
    { const char* e = getenv("CRT_COST_SPREAD"); g.costSpread = e ? (float)atof(e) : 0.8f; }
    { const char* e = getenv("CRT_SPLIT");               // tuning knob: cap on quadrant-split tiles per XCD (both modes)
      if (e) { int v = atoi(e); v = v < 0 ? 0 : (v > CRT_MAX_SPLIT ? CRT_MAX_SPLIT : v); g.maxSplit = g.maxSplitPipelined = v; }
      else { g.maxSplit = CRT_MAX_SPLIT; g.maxSplitPipelined = CRT_MAX_SPLIT_PIPELINED; } }
    { const char* e = getenv("CRT_TLAS"); g.forceTlas = e ? (atoi(e) != 0 ? 1 : 0) : -1; }
    { const char* e = getenv("CRT_STAGGER_US"); g.staggerUs = e ? atoi(e) : -1; }
    { const char* e = getenv("CRT_FEEDBACK"); g.feedback = !(e && atoi(e) == 0); }
    { const char* e = getenv("CRT_FEEDBACK_ASYNC"); g.feedbackAsync = (e && atoi(e) != 0); }
    HIPCHK(hipMemset(g.roots, 0, CRT_MAX_MESHES * sizeof(uint32_t)));
    { std::vector<uint32_t> e(CRT_MAX_MESHES, crt_empty_ref((uint32_t)g.triCap)); HIPCHK(hipMemcpy(g.rootRefs, e.data(), e.size() * sizeof(uint32_t), hipMemcpyHostToDevice)); }
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
                     g.texels, g.materials, g.textures, g.rays, g.counters, g.err,
                     g.queryBuf, g.buildBuf, g.buildTris, g.stamps, g.bounceQueue, g.bounceCount, g.noCullBounds };
    for (FrameSlot& fs : g.slot) {
        void* q[] = { fs.out, fs.aux, fs.ovf, fs.order, fs.len, fs.cost, fs.mixOrder, fs.mixLen, fs.packBuf, fs.instances, fs.devInstances, fs.instBounds, fs.tlas, fs.alwaysList };
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
    if (first + count > g.trisHigh) g.trisHigh = first + count;
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
    rebuild_instance_master();
    return CRT_OK;
}

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
        if (h && atoi(h) != 0 && f && atoi(f) != 0) return CRT_E_OUT_OF_RANGE;
    }
    RCCHK(sync_all());

    // second triangle pool (allocated on first use, indexed like rawTris) and scratch:
    // build nodes | rank, holes, backL | 2 x 3 id lists | 2 x BIG-node scratch | 2 x chunk->node + 3 per-chunk counts | mesh counts, roots | scalars
    if (!g.buildTris) HIPCHK(hipMalloc(&g.buildTris, g.triCap * sizeof(CrtTri)));
    if (!g.buildCtlHost) { HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&g.buildCtlHost), sizeof(CrtBuildCtlHost), hipHostMallocDefault)); g.buildCtlHost->seq = 0; g.buildSeq = 0; }
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
    crt_bvh_centroids<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(A, firstTri, total);
    crt_bvh_init_roots<<<1, 1, 0, st>>>(bn, dCounts, numMeshes, (uint32_t)firstTri, bigs[0], chunkNode[0]);
    HIPCHK(hipGetLastError());

    const unsigned W = CRT_BVH_WAVES, T = CRT_BVH_BIG_THREADS;
    auto bounds = [&](int p, const uint32_t n[3], uint32_t nChunks, const CrtTri* tris) {
        const CrtBuildLists& L = lists[p];
        if (n[0]) { crt_bvh_big_reset<<<(n[0] + 255) / 256, 256, 0, st>>>(bigs[p], n[0]);
                    crt_bvh_big_bounds<<<nChunks, T, 0, st>>>(bn, L.list[0], bigs[p], chunkNode[p], tris); }
        if (n[1]) crt_bvh_bounds_wave<<<(n[1] + W - 1) / W, 64 * W, 0, st>>>(bn, L.list[1], n[1], tris);
        if (n[2]) crt_bvh_bounds_tiny<<<(n[2] + 63) / 64, 64, 0, st>>>(bn, L.list[2], n[2], tris);
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
        ++level;
        if (cnt[0]) {
            CrtBigScratch* big = bigs[cur]; const uint32_t* cn = chunkNode[cur];
            crt_bvh_big_bins<<<chunks, T, 0, st>>>(bn, L.list[0], big, cn, src);
            crt_bvh_big_sweep<<<chunks, T, 0, st>>>(bn, L.list[0], big, cn, src, dst, chunkL);
            crt_bvh_big_count<<<chunks, T, 0, st>>>(bn, L.list[0], big, cn, src, chunkL, chunkFR, chunkBL);
            crt_bvh_big_tables<<<chunks, T, 0, st>>>(bn, L.list[0], big, cn, src, (uint32_t)firstTri, chunkFR, chunkBL, rank, holes, backL);
            crt_bvh_big_scatter<<<chunks, T, 0, st>>>(bn, L.list[0], big, cn, src, dst, (uint32_t)firstTri, rank, holes, backL, end, dCtl, N, bigs[cur ^ 1], chunkNode[cur ^ 1]);
        }
        if (cnt[1]) crt_bvh_mid<<<(cnt[1] + W - 1) / W, 64 * W, 0, st>>>(bn, L.list[1], cnt[1], src, dst, (uint32_t)firstTri, rank, holes, backL, end, &dCtl->packed, N);
        if (cnt[2]) crt_bvh_tiny<<<(cnt[2] + 63) / 64, 64, 0, st>>>(bn, L.list[2], cnt[2], src, dst, end, &dCtl->packed, N);
        HIPCHK(hipGetLastError());
        {   // the level's list sizes: published into pinned memory behind the level's kernels; spin on the sequence number (a copy + stream
            // synchronisation per level cost ~40 us x 23 levels of a 1 M-triangle build), fall back to the stream if it does not arrive
            const uint32_t seq = ++g.buildSeq;
            crt_bvh_publish<<<1, 1, 0, st>>>(dCtl, g.buildCtlHost, seq);
            HIPCHK(hipGetLastError());
            bool arrived = false;
            for (unsigned spin = 0; spin < (1u << 22); ++spin) {
                if (g.buildCtlHost->seq == seq) { arrived = true; break; }
                if ((spin & 0x3FFu) == 0x3FFu && hipStreamQuery(st) != hipErrorNotReady) break;      // finished (or failed) without our flag: let the sync below sort it out
                __builtin_ia32_pause();
            }
            if (!arrived) { HIPCHK(hipStreamSynchronize(st)); if (g.buildCtlHost->seq != seq) return CRT_E_UNSUPPORTED; }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            ctl = g.buildCtlHost->ctl;
        }
        if (ctl.degenerate) {                                                      // BVH.cpp:194 hit a BIG node: its permuted triangles go to both buffers
            crt_bvh_big_degenerate<<<chunks, T, 0, st>>>(bn, L.list[0], bigs[cur], chunkNode[cur], src, dst);
            crt_bvh_big_degenerate_mark<<<(cnt[0] + 255) / 256, 256, 0, st>>>(bn, L.list[0], bigs[cur], cnt[0]);
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
        crt_bvh_leaf_flags<<<(numBuilt + 255) / 256, 256, 0, st>>>(bn, numBuilt, (uint32_t)firstTri, flags);
        crt_bvh_scan_sums<<<nb, CRT_BVH_SCAN_THREADS, 0, st>>>(flags, (uint32_t)total, sums);
        crt_bvh_scan_blocks<<<1, CRT_BVH_SCAN_THREADS, 0, st>>>(sums, nb);
        crt_bvh_scan_apply<<<nb, CRT_BVH_SCAN_THREADS, 0, st>>>(flags, (uint32_t)total, sums, S);
        crt_bvh_emit<<<(numBuilt + 255) / 256, 256, 0, st>>>(bn, numBuilt, numMeshes, S, (uint32_t)firstTri, (uint32_t)total, (uint32_t)firstNode, g.rawNodes, dRoots, dScal, dScal + 1);
        HIPCHK(hipGetLastError());
    }
    uint32_t used = 0, scal[2] = { 0, 0 };
    HIPCHK(hipMemcpyAsync(scal, dScal, sizeof scal, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(g.roots + firstMesh, dRoots, (size_t)numMeshes * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(g.hRoots + firstMesh, dRoots, (size_t)numMeshes * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    crt_relayout_tris<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(g.rawTris, firstTri, total, g.triHot, g.triCold);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    used = scal[0];
    if (scal[1] != 0 || used != numBuilt) return CRT_E_OUT_OF_RANGE;                               // the closed form and the level loop disagree: never seen, would mean a damaged tree
    if (firstNode + used > g.nodeCount) g.nodeCount = (uint32_t)(firstNode + used);
    if (firstMesh + (size_t)numMeshes > g.numRoots) g.numRoots = (uint32_t)(firstMesh + (size_t)numMeshes);
    if (nodesUsedOut) *nodesUsedOut = used;
    return rebuild_bvh_layout();
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

// Feedback launch lists for the megakernel (lane_pixel / crt_order_kernel). Buffers follow the frame geometry; a
// change of geometry resets to the identity order. The previous frame's per-tile costs are turned into this frame's
// lists (and the costs zeroed) by a sort that is queued right AFTER the previous frame's last kernel and its end
// event (sort_for_next_frame), so it runs while the host is between two crt1_render calls and is off the frame's
// critical path (it used to open every frame: 10 us + a launch gap of a 0.5 ms synchronous frame).
// this frame's per-tile costs -> the next frame's lists; with g.costSpread > 0 a tile is ranked by its neighbours' costs too
static void launch_order_kernel(const CrtFrame& F, FrameSlot& fs, bool pipelined)
{
    const uint32_t* key = fs.cost;
    if (g.costSpread > 0.0f && g.viewMoved) {
        uint32_t* k2 = fs.cost + fs.orderCap;                   // second half of the cost allocation
        crt_cost_spread_kernel<<<(8 * F.slotsPerXcd + 255) / 256, 256, 0, fs.stream>>>(fs.cost, k2, F.slotsPerXcd, F.tilesX, g.costSpread);
        key = k2;
    }
    crt_order_kernel<<<8, 1024, 0, fs.stream>>>(fs.cost, key, fs.order, fs.len, F.slotsPerXcd, F.listCap, (uint32_t)(pipelined ? g.maxSplitPipelined : g.maxSplit),
                                                 (pipelined ? g.splitBetaAsync : g.splitBeta) / (float)((g.numCUs / 8) * 4 * CRT_WAVES_PER_SIMD));
}

static int prepare_launch_lists(CrtFrame& F, unsigned& grid, FrameSlot& fs, bool pipelined)
{
    const int key[6] = { g.width, g.height, g.bandRows, g.rank, g.nRanks, F.slotsPerXcd };
    F.listCap = F.slotsPerXcd + 3 * CRT_MAX_SPLIT;
    const size_t need = (size_t)8 * (size_t)F.listCap;
    if (need > fs.orderCap) {
        if (fs.order) (void)hipFree(fs.order);
        if (fs.len) (void)hipFree(fs.len);
        if (fs.cost) (void)hipFree(fs.cost);
        fs.order = nullptr; fs.len = nullptr; fs.cost = nullptr; fs.orderCap = 0;
        HIPCHK(hipMalloc(&fs.order, sizeof(uint32_t) * need));
        HIPCHK(hipMalloc(&fs.len, sizeof(uint32_t) * 8));
        HIPCHK(hipMalloc(&fs.cost, sizeof(uint32_t) * need * 2));        // costs, then the sort keys derived from them
        fs.orderCap = need; fs.orderSlots = -1;
    }
    if (fs.orderSlots != F.slotsPerXcd || memcmp(key, fs.orderKey, sizeof key) != 0) {
        HIPCHK(hipMemsetAsync(fs.cost, 0, sizeof(uint32_t) * need, fs.stream));
        crt_identity_order_kernel<<<(8 * F.slotsPerXcd + 255) / 256, 256, 0, fs.stream>>>(fs.order, fs.len, F.slotsPerXcd, F.listCap);
        fs.orderSlots = F.slotsPerXcd; memcpy(fs.orderKey, key, sizeof key);
    } else if (!fs.listsReady) {
        launch_order_kernel(F, fs, pipelined);
    }
    fs.listsReady = false;
    HIPCHK(hipGetLastError());
    F.order = fs.order; F.listLen = fs.len; F.cost = fs.cost;
    grid = 8u * (unsigned)F.listCap;
    return CRT_OK;
}

// Queued behind a frame's last kernel: this frame's costs -> the next frame's lists (same geometry assumed; a change is
// caught by the key in prepare_launch_lists, which then starts from the identity order again).
static int sort_for_next_frame(const CrtFrame& F, FrameSlot& fs, bool pipelined)
{
    launch_order_kernel(F, fs, pipelined);
    HIPCHK(hipGetLastError());
    fs.listsReady = true;
    return CRT_OK;
}

// The Trace launch(es) of one frame, by kernel structure (default: megakernel with feedback launch lists).
// `out`: the frame the launch writes (the slot's frame, or its unfiltered copy when FXAA follows).
// *epilogueApplied: the launch was the default megakernel, which applies F.epilogue (RGBA8 target / PostProcess) itself
static int launch_trace(const CrtDevScene& S, const CrtFrame& F, int flags, unsigned grid, FrameSlot& fs, float4* out, bool* epilogueApplied)
{
    *epilogueApplied = false;
    const bool count = (flags & CRT_RENDER_COUNTERS) != 0;
    if (count) HIPCHK(hipMemsetAsync(g.counters, 0, CRT_NUM_COUNTERS * sizeof(unsigned long long), fs.stream));
    if (flags & CRT_RENDER_STAMPS) {                      // diagnostic launch with per-wave stamps
        const size_t need = (16 + (size_t)grid * 8) * sizeof(unsigned long long);
        if (need > g.stampBytes) {
            if (g.stamps) (void)hipFree(g.stamps);
            g.stamps = nullptr; g.stampBytes = 0;
            HIPCHK(hipMalloc(&g.stamps, need));
            g.stampBytes = need;
        }
        g.stampWaves = grid;
        HIPCHK(hipMemsetAsync(g.stamps, 0, need, fs.stream));
        crt_trace_kernel<false, true><<<grid, CRT_BLOCK, 0, fs.stream>>>(S, F, out, g.stamps);
        *epilogueApplied = true;                           // the same kernel template: F.epilogue is applied there
    } else if (g.wavefront) {                              // bounce 0, ballot compaction, bounce 1
        const unsigned ownedPixels = (unsigned)F.ownedTileRows * CRT_TILE * (unsigned)F.width;
        const unsigned grid2 = (ownedPixels + CRT_BLOCK - 1) / CRT_BLOCK;
        HIPCHK(hipMemsetAsync(g.bounceCount, 0, sizeof(uint32_t), fs.stream));
        if (count) {
            crt_primary_kernel<true><<<grid, CRT_BLOCK, 0, fs.stream>>>(S, F, out, g.counters, g.bounceQueue, g.bounceCount);
            crt_bounce_kernel<true><<<grid2, CRT_BLOCK, 0, fs.stream>>>(S, F, out, g.counters, g.bounceQueue, g.bounceCount);
        } else {
            crt_primary_kernel<false><<<grid, CRT_BLOCK, 0, fs.stream>>>(S, F, out, g.counters, g.bounceQueue, g.bounceCount);
            crt_bounce_kernel<false><<<grid2, CRT_BLOCK, 0, fs.stream>>>(S, F, out, g.counters, g.bounceQueue, g.bounceCount);
        }
    } else {
        // default megakernel: <COUNT, STAMP, SHADOW, TLAS, REFRACT>
        *epilogueApplied = true;
        const bool shadow = (flags & CRT_RENDER_SHADOWS) != 0, refract = (flags & CRT_RENDER_REFRACTION) != 0;
        // TLAS: more than CRT_TLAS_MIN_INSTANCES instances and an instance tree to walk (CRT_TLAS=0/1 forces)
        const bool tlas = S.tlasNodes > 0 && (g.forceTlas >= 0 ? (g.forceTlas != 0 && S.numInstances <= g.instHigh) : (S.numInstances > CRT_TLAS_MIN_INSTANCES && S.numInstances <= g.instHigh));      // (S.tlasNodes = 0: no tree, or a frame without the cull)
#define CRT_LAUNCH_TRACE3(C_, S_, T_, R_) crt_trace_kernel<C_, false, S_, T_, R_><<<grid, CRT_BLOCK, 0, fs.stream>>>(S, F, out, g.counters)
#define CRT_LAUNCH_TRACE2(C_, S_, T_) do { if (refract) CRT_LAUNCH_TRACE3(C_, S_, T_, true); else CRT_LAUNCH_TRACE3(C_, S_, T_, false); } while (0)
#define CRT_LAUNCH_TRACE(C_, S_) do { if (tlas) CRT_LAUNCH_TRACE2(C_, S_, true); else CRT_LAUNCH_TRACE2(C_, S_, false); } while (0)
        if (count) { if (shadow) CRT_LAUNCH_TRACE(true, true); else CRT_LAUNCH_TRACE(true, false); }
        else       { if (shadow) CRT_LAUNCH_TRACE(false, true); else CRT_LAUNCH_TRACE(false, false); }
#undef CRT_LAUNCH_TRACE
#undef CRT_LAUNCH_TRACE2
#undef CRT_LAUNCH_TRACE3
    }
    HIPCHK(hipGetLastError());
    return CRT_OK;
}

// one wave that occupies its stream for `ticks` periods of the 100 MHz real-time counter (start-up stagger, see State::burstFrames)
__global__ void crt_delay_kernel(unsigned long long ticks)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long guard = 0;
    while (__builtin_amdgcn_s_memrealtime() - r0 < ticks && guard < (1ull << 22)) { __builtin_amdgcn_s_sleep(16); ++guard; }
}

// In a multi-device session the dispatcher (crt_render) decides once per frame what every device must agree on and hands it
// to each device's crt1_render: the frame slot (so a device that owned no rows of some frame, or failed one, cannot fall out
// of step with the primary's slot rotation) and whether the call may return before the device has finished (secondaries
// never wait on the host: the primary's end-of-frame event waits for their partDone events, which is what gives a
// synchronous N-device frame the duration of the longest share instead of the sum of two).
struct RenderPlan { int slot; bool noHostWait; };

int crt1_render(const CrtTraceArgs* args, const float invView[16], const float invProj[16], int flags, const RenderPlan* plan = nullptr)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!args || !invView || !invProj) return CRT_E_BAD_ARGUMENT;
    if (args->numMeshes > CRT_MAX_INSTANCES) return CRT_E_OUT_OF_RANGE;
    if (!g.sceneValid) return CRT_E_BAD_ARGUMENT;
    int rc = CRT_OK;
    CrtFrame F; fill_frame(F, args, invView, invProj);
    if (F.gridBlocks == 0) {
        // a device that owns no rows of this frame still takes part in the frame's hand-shake: its "bands have arrived"
        // event is recorded on the planned slot so the primary's wait refers to this frame, not to an older one
        if (plan && g.groupSize > 1 && g.primary != G && plan->slot >= 0 && plan->slot < g.nSlots) {
            FrameSlot& efs = g.slot[plan->slot];
            if (plan->slot != 0) g.othersBusy = true;
            HIPCHK(hipEventRecord(efs.partDone, efs.stream));
        }
        return CRT_OK;
    }
    unsigned grid = (unsigned)F.gridBlocks;

    // Slot choice: plain ASYNC frames of the default kernel rotate over the frame slots so consecutive frames
    // overlap (each slot has its own stream, output buffer and launch lists). Everything else -- synchronous frames,
    // diagnostic flags, the opt-in kernel variants (they share queues / the ray buffer) -- runs on slot 0.
    const bool variant = g.wavefront != 0;
    if ((flags & (CRT_RENDER_SHADOWS | CRT_RENDER_REFRACTION)) && (variant || (flags & CRT_RENDER_STAMPS))) return CRT_E_UNSUPPORTED;   // default kernel only
    const bool fxaa = (flags & CRT_RENDER_FXAA) != 0;
    if (fxaa && g.groupSize <= 1 && g.nRanks > 1) return CRT_E_UNSUPPORTED;                  // the filter reads across band edges
    const bool pipelined = (flags & CRT_RENDER_ASYNC) && !variant
                        && !(flags & (CRT_RENDER_WRITE_RAYS | CRT_RENDER_COUNTERS | CRT_RENDER_STAMPS));
    int slot = 0;
    if (plan) {                              // multi-device session: the dispatcher chose the slot for every device
        slot = pipelined ? plan->slot : 0;
        if (slot < 0 || slot >= g.nSlots) return CRT_E_BAD_ARGUMENT;
        if (!pipelined) { rc = quiesce(); if (rc) return rc; }
    } else if (pipelined) slot = (int)(g.asyncSeq++ % (unsigned)g.nSlots);
    else { rc = quiesce(); if (rc) return rc; }
    FrameSlot& fs = g.slot[slot];
    EventSet& es = fs.es[fs.frames & 1u];
    rc = collect_set(es);                    // waits for the frame two back on this slot: at most two queued per slot
    if (rc) return rc;
    if (flags & CRT_RENDER_COUNTERS) { rc = collect_timing(); if (rc) return rc; }
    if (slot != 0) g.othersBusy = true;
    rc = ensure_slot_instances(fs);          // this slot's instance tables, refreshed on its stream if an upload happened since
    if (rc) return rc;
    CrtDevScene S;
    // Feedback launch lists serve synchronous frames, whose end is decided by their slowest waves. With frames in flight the
    // tail is hidden by the next frame and the lists only cost (cost atomics, the sort launch, quadrant waves at a quarter
    // of the lane utilisation): 7.58 with, 7.72 Gray/s without on multi-1M -> pipelined frames use the plain row-interleaved order.
    const bool mix3 = (flags & CRT_RENDER_DIAG_MIX3) != 0;
    if (mix3) {
        if (pipelined || variant || g.groupSize > 1 || (flags & (CRT_RENDER_STAMPS | CRT_RENDER_WRITE_RAYS | CRT_RENDER_FXAA))) return CRT_E_UNSUPPORTED;
        // three copies of the plain row-interleaved order, copy j starting a third of the XCD's list later: entry 3 i + j = tile (i + j S / 3) mod S
        const int S3 = 3 * F.slotsPerXcd;
        if ((size_t)8 * S3 > fs.mixCap) {
            HIPCHK(hipStreamSynchronize(fs.stream));
            if (fs.mixOrder) (void)hipFree(fs.mixOrder);
            if (fs.mixLen) (void)hipFree(fs.mixLen);
            fs.mixOrder = nullptr; fs.mixLen = nullptr; fs.mixCap = 0; fs.mixSlots = -1;
            HIPCHK(hipMalloc(&fs.mixOrder, sizeof(uint32_t) * 8 * (size_t)S3));
            HIPCHK(hipMalloc(&fs.mixLen, sizeof(uint32_t) * 8));
            fs.mixCap = (size_t)8 * S3;
        }
        if (fs.mixSlots != F.slotsPerXcd) {
            std::vector<uint32_t> h((size_t)8 * S3), len(8, (uint32_t)S3);
            for (int x = 0; x < 8; ++x)
                for (int i = 0; i < F.slotsPerXcd; ++i)
                    for (int j = 0; j < 3; ++j) h[(size_t)x * S3 + 3 * i + j] = (uint32_t)((i + j * (F.slotsPerXcd / 3)) % F.slotsPerXcd);
            HIPCHK(hipMemcpyAsync(fs.mixOrder, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, fs.stream));
            HIPCHK(hipMemcpyAsync(fs.mixLen, len.data(), 8 * sizeof(uint32_t), hipMemcpyHostToDevice, fs.stream));
            HIPCHK(hipStreamSynchronize(fs.stream));      // the host vectors go out of scope
            fs.mixSlots = F.slotsPerXcd;
        }
        F.order = fs.mixOrder; F.listLen = fs.mixLen; F.listCap = S3; F.cost = nullptr;
        grid = 8u * (unsigned)S3;
    } else
    if (g.feedback && !g.wavefront && (!pipelined || g.feedbackAsync)) { rc = prepare_launch_lists(F, grid, fs, pipelined); if (rc) return rc; }
    {   // overflow blocks: one per workgroup of the largest launch of this frame (wavefront: the bounce launch may be larger)
        size_t blocks = grid;
        if (g.wavefront) { const size_t g2 = ((size_t)F.ownedTileRows * CRT_TILE * (size_t)F.width + CRT_BLOCK - 1) / CRT_BLOCK; if (g2 > blocks) blocks = g2; }
        rc = ensure_overflow(fs, blocks); if (rc) return rc;
    }
    fill_scene(S, args->numMeshes, fs, beyond_cull_range(sqrt((double)args->cameraPos[0] * args->cameraPos[0] + (double)args->cameraPos[1] * args->cameraPos[1] + (double)args->cameraPos[2] * args->cameraPos[2])));

    // events: [0] frame start, [1] Trace start, [2] Trace end, [3] end of PostProcess = frame end.
    // A plain frame records only two (RayGen is fused, PostProcess off): [0] == [1], [2] == [3].
    if (g.statStartArmed) {                  // first frame since the statistics were reset: start of the extent
        HIPCHK(hipEventRecord(g.statStart, fs.stream));
        g.statStartArmed = false; g.statStartValid = true; g.statStartSeq = g.frameSeq + 1; g.statExtent = 0; g.statFirstMs = 0; g.frameLogN = 0;
    }
    if (pipelined) {
        // first frame of this slot in a burst that starts from an idle device: hold it back so the slots do not run in lockstep
        const unsigned k = g.burstFrames++;
        // (automatic only with up to three slots: with eight -- a rank's small share of a tiled frame, where one frame cannot fill the
        // GPU and the slots exist to run many at once -- the ramp costs more than the coinciding tails: 83.2 -> 74.6 Gray/s predicted at N = 8)
        if (k > 0 && k < (unsigned)g.nSlots && g.staggerUs != 0 && (g.staggerUs > 0 || (g.nSlots <= 3 && g.prevBurstFrames > (unsigned)g.nSlots))) {
            double step = g.staggerUs > 0 ? (double)g.staggerUs : (double)g.pipelinedLatencyMs * 1e3 / (double)g.nSlots;
            if (step > 500.0) step = 500.0;                      // a stale or foreign latency must not stall a burst
            const double us = step * k;
            if (us >= 5.0) { crt_delay_kernel<<<1, 64, 0, fs.stream>>>((unsigned long long)(us * 100.0)); HIPCHK(hipGetLastError()); g.staggeredFrames++; }
        }
    } else { if (g.burstFrames) g.prevBurstFrames = g.burstFrames; g.burstFrames = 0; }
    es.evRaygen = (flags & CRT_RENDER_WRITE_RAYS) != 0;
    es.evPost = (flags & (CRT_RENDER_POSTPROCESS | CRT_RENDER_UNORM8 | CRT_RENDER_FXAA)) != 0;
    HIPCHK(hipEventRecord(es.ev[0], fs.stream));
    if (es.evRaygen) {
        crt_raygen_kernel<<<grid, CRT_BLOCK, 0, fs.stream>>>(F, g.rays);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(es.ev[1], fs.stream));
    }
    // upstream's per-pixel stages behind Trace (its RGBA8 render target, PostProcess) ride in the Trace kernel's epilogue
    // unless a kernel variant without the epilogue is selected. FXAA sits between them and reads neighbours: on one device
    // Trace then writes the slot's second buffer (its pixels already through the RGBA8 target) and the filter writes the
    // frame, applying PostProcess and the final RGBA8 store in ITS epilogue -- two launches, no copy.
    const bool unorm = (flags & CRT_RENDER_UNORM8) != 0, post = (flags & CRT_RENDER_POSTPROCESS) != 0;
    const bool fxaaLocal = fxaa && g.groupSize <= 1;
    const size_t framePixels = (size_t)g.width * (size_t)g.height;
    if (fxaa && !(g.groupSize > 1 && g.primary != G) && fs.auxPixels < framePixels) {
        HIPCHK(hipStreamSynchronize(fs.stream));
        if (fs.aux) (void)hipFree(fs.aux);
        fs.aux = nullptr; fs.auxPixels = 0;
        HIPCHK(hipMalloc(&fs.aux, framePixels * sizeof(float4)));
        fs.auxPixels = framePixels;
    }
    if (!fxaa) F.epilogue = (unorm ? CRT_EPILOGUE_QUANTIZE : 0u) | (post ? CRT_EPILOGUE_POST : 0u);
    else if (fxaaLocal) F.epilogue = unorm ? CRT_EPILOGUE_QUANTIZE : 0u;
    // a read-back of the RGBA8 frame: the kernel that stores the final pixel stores its four bytes too (one device; a
    // multi-device session packs the gathered frame on its first device)
    const bool packInKernel = unorm && (flags & CRT_RENDER_READBACK) && g.groupSize <= 1;
    if (packInKernel && framePixels * 4 > fs.packCap) {
        HIPCHK(hipStreamSynchronize(fs.stream));
        if (fs.packBuf) (void)hipFree(fs.packBuf);
        fs.packBuf = nullptr; fs.packCap = 0;
        HIPCHK(hipMalloc(&fs.packBuf, framePixels * 4));
        fs.packCap = framePixels * 4;
    }
    if (packInKernel && !fxaa) F.packOut = fs.packBuf;
    bool fused = false;
    rc = launch_trace(S, F, flags, grid, fs, fxaaLocal ? fs.aux : fs.out, &fused);
    if (rc) return rc;
    // in-process multi-GPU, primary device: the frame is complete when every secondary's bands have arrived -- its last
    // event is recorded behind waits for their partDone events (recorded before this call: the dispatcher submits the
    // secondaries first)
    const bool isPrimary = g.groupSize > 1 && g.primary == G, isSecondary = g.groupSize > 1 && g.primary != G;
    auto wait_for_parts = [&]() -> int {
        for (int d = 1; d < g.groupSize; ++d) HIPCHK(hipStreamWaitEvent(fs.stream, g.group[d]->slot[slot].partDone, 0));
        return CRT_OK;
    };
    if (isPrimary && !es.evPost) RCCHK(wait_for_parts());
    HIPCHK(hipEventRecord(es.ev[2], fs.stream));
    if (es.evPost) {
        // upstream: Trace write_imagef's into an RGBA8 texture, PostProcess read_imagef's it back and write_imagef's again
        if (!fxaa) {
            if (!fused) {
                if (unorm) crt_quantize_kernel<<<grid, CRT_BLOCK, 0, fs.stream>>>(F, fs.out);
                if (post) crt_postprocess_kernel<<<grid, CRT_BLOCK, 0, fs.stream>>>(F, fs.out);
                if (unorm && post) crt_quantize_kernel<<<grid, CRT_BLOCK, 0, fs.stream>>>(F, fs.out);
                HIPCHK(hipGetLastError());
            }
            if (isPrimary) RCCHK(wait_for_parts());
        } else if (!isSecondary) {
            // FXAA reads up to 5 pixels around its own in the Trace result, so it runs on the whole frame: a multi-device
            // session gathers the raw bands first (the secondaries skip their per-pixel stages) and its first device filters
            if (isPrimary) RCCHK(wait_for_parts());
            CrtFrame FF = F;                                // every tile row, plain order
            FF.order = nullptr; FF.cost = nullptr; FF.listLen = nullptr;
            FF.rank = 0; FF.nRanks = 1;
            FF.ownedTileRows = (g.height + CRT_TILE - 1) / CRT_TILE;
            FF.gridBlocks = ((FF.ownedTileRows + 7) / 8) * 8 * FF.tilesX;
            FF.slotsPerXcd = FF.gridBlocks / 8; FF.listCap = FF.slotsPerXcd;
            const unsigned gridAll = (unsigned)FF.gridBlocks;
            if (fxaaLocal) {
                if (unorm && !fused) crt_quantize_kernel<<<gridAll, CRT_BLOCK, 0, fs.stream>>>(FF, fs.aux);
            } else {
                if (unorm) crt_quantize_kernel<<<gridAll, CRT_BLOCK, 0, fs.stream>>>(FF, fs.out);
                HIPCHK(hipMemcpyAsync(fs.aux, fs.out, framePixels * sizeof(float4), hipMemcpyDeviceToDevice, fs.stream));
            }
            FF.epilogue = (unorm ? CRT_EPILOGUE_QUANTIZE : 0u) | (post ? CRT_EPILOGUE_POST : 0u);
            FF.packOut = packInKernel ? fs.packBuf : nullptr;
            crt_fxaa_kernel<<<gridAll, CRT_BLOCK, 0, fs.stream>>>(FF, fs.aux, fs.out);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(es.ev[3], fs.stream));
    }
    if (isSecondary) {
        // gather: this device's bands go into the primary's frame of the same slot (peer copy over xGMI), once the primary
        // is done with whatever the slot's previous frame still had queued (its read-back)
        FrameSlot& pfs = g.primary->slot[slot];
        HIPCHK(hipStreamWaitEvent(fs.stream, pfs.slotDone, 0));
        RCCHK(copy_owned_rows_async(pfs.out, fs.out, 16, hipMemcpyDeviceToDevice, fs.stream));
        HIPCHK(hipEventRecord(fs.partDone, fs.stream));
    }
    g.cur = slot;
    es.pending = true; es.flags = flags; es.seq = ++g.frameSeq; fs.frames++;
    const bool sorted = F.order != nullptr && !mix3;
    if (sorted) {
        // did the view change since the last sorted frame? (camera matrices and position, instance tables)
        float view[35];
        memcpy(view, F.invView, 64); memcpy(view + 16, F.invProj, 64); memcpy(view + 32, F.camPos, 12);
        g.viewMoved = memcmp(view, g.lastView, sizeof view) != 0 || g.lastViewInst != g.instVersion;
        memcpy(g.lastView, view, sizeof view); g.lastViewInst = g.instVersion;
        rc = sort_for_next_frame(F, fs, pipelined); if (rc) return rc;
    }
    if (flags & CRT_RENDER_READBACK) {
        // the frame travels to pinned host memory behind its own kernels; the other slots' frames keep the GPU busy meanwhile
        const size_t pixels = (size_t)g.width * (size_t)g.height;
        const bool bytes8 = (flags & CRT_RENDER_UNORM8) != 0;
        const size_t bytes = pixels * (bytes8 ? 4 : 16);
        if (bytes > fs.hostCap) {
            if (fs.hostBuf) (void)hipHostFree(fs.hostBuf);
            fs.hostBuf = nullptr; fs.hostCap = 0;
            HIPCHK(hipHostMalloc(&fs.hostBuf, bytes, hipHostMallocDefault));
            fs.hostCap = bytes;
        }
        if (!fs.copied) HIPCHK(hipEventCreateWithFlags(&fs.copied, hipEventDisableTiming));
        const void* src = fs.out;
        if (bytes8) {
            if (pixels * 4 > fs.packCap) {
                if (fs.packBuf) (void)hipFree(fs.packBuf);
                fs.packBuf = nullptr; fs.packCap = 0;
                HIPCHK(hipMalloc(&fs.packBuf, pixels * 4));
                fs.packCap = pixels * 4;
            }
            const bool packed = packInKernel && (fxaa || fused);     // the Trace (or FXAA) kernel stored the bytes already
            if (!packed) crt_pack_unorm8_kernel<<<(unsigned)((pixels + 255) / 256), 256, 0, fs.stream>>>(fs.out, fs.packBuf, pixels);
            HIPCHK(hipGetLastError());
            src = fs.packBuf;
        }
        // only the rows this rank renders travel (the host buffer keeps the full-frame layout)
        RCCHK(copy_owned_rows_async(fs.hostBuf, src, bytes8 ? 4 : 16, hipMemcpyDeviceToHost, fs.stream, isPrimary));
        HIPCHK(hipEventRecord(fs.copied, fs.stream));
        fs.hostBytes = bytes; g.readbackRing[g.readbackCount++ % CRT_MAX_FRAMES_IN_FLIGHT] = slot;
    }
    if (isPrimary) HIPCHK(hipEventRecord(fs.slotDone, fs.stream));
    // the reference's clFinish (Renderer.cpp:367): wait for the frame's end event -- the sort for the next frame that is
    // queued behind it needs no waiting for
    if (!(flags & CRT_RENDER_ASYNC) && !(plan && plan->noHostWait)) HIPCHK(hipEventSynchronize(es.evPost ? es.ev[3] : es.ev[2]));
    return CRT_OK;
}

// Whether a frame with these flags rotates over the frame slots (the rule of crt1_render, for the dispatcher)
static bool frame_is_pipelined(int flags)
{
    return (flags & CRT_RENDER_ASYNC) && !g.wavefront && !(flags & (CRT_RENDER_WRITE_RAYS | CRT_RENDER_COUNTERS | CRT_RENDER_STAMPS));
}

// Diagnostic: the shader clock under whatever load the device carries right now. One wave per XCD spins for `micros`
// microseconds of the 100 MHz real-time counter and reports delta s_memtime / delta s_memrealtime (MI355X_MICROARCH.md, DVFS
// item 6); runs on a stream of its own, next to the frames in flight.
__global__ void crt_clock_probe_kernel(unsigned long long ticks, double* __restrict__ out)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0, guard = 0;
    while (r1 - r0 < ticks && guard < (1ull << 24)) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); ++guard; }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[blockIdx.x] = r1 > r0 ? (double)(c1 - c0) / (double)(r1 - r0) * 0.1 : 0.0;
}

int crt1_debug_measure_clock(int micros, double* ghz)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!ghz || micros < 1 || micros > 100000) return CRT_E_BAD_ARGUMENT;
    double* d = nullptr; hipStream_t st = nullptr;
    HIPCHK(hipMalloc(&d, 8 * sizeof(double)));
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double h[8] = { 0 };
    if (e == hipSuccess) {
        crt_clock_probe_kernel<<<8, 64, 0, st>>>((unsigned long long)micros * 100ull, d);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(d);
    if (e != hipSuccess) return (int)e;
    double sum = 0; int n = 0;
    for (double v : h) if (v > 0.0) { sum += v; ++n; }
    *ghz = n ? sum / n : 0.0;
    return CRT_OK;
}

int crt1_sync(void)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    return sync_all();
}

int crt1_query_hits(const float* origins, const float* dirs, int n, uint32_t numInstances, CrtRayHit* out)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (n <= 0) return CRT_OK;
    if (!origins || !dirs || !out || numInstances > CRT_MAX_INSTANCES) return CRT_E_BAD_ARGUMENT;
    if (!g.sceneValid) return CRT_E_BAD_ARGUMENT;
    int rc = collect_timing();
    if (rc) return rc;
    RCCHK(quiesce());
    const size_t rayBytes = sizeof(float) * 3 * (size_t)n, need = rayBytes * 2 + sizeof(CrtRayHit) * (size_t)n;
    if (need > g.queryBytes) {
        if (g.queryBuf) (void)hipFree(g.queryBuf);
        g.queryBuf = nullptr; g.queryBytes = 0;
        HIPCHK(hipMalloc(&g.queryBuf, need));
        g.queryBytes = need;
    }
    float* dO = static_cast<float*>(g.queryBuf);
    float* dD = dO + 3 * (size_t)n;
    CrtRayHit* dH = reinterpret_cast<CrtRayHit*>(dD + 3 * (size_t)n);
    HIPCHK(hipMemcpyAsync(dO, origins, rayBytes, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemcpyAsync(dD, dirs, rayBytes, hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipMemsetAsync(g.counters, 0, CRT_NUM_COUNTERS * sizeof(unsigned long long), g.stream));
    FrameSlot& fs = g.slot[0];
    RCCHK(ensure_slot_instances(fs));
    RCCHK(ensure_overflow(fs, (size_t)((n + CRT_BLOCK - 1) / CRT_BLOCK)));
    double farthest2 = 0.0;      // the cull is proven for origins up to State::cullOriginLimit from the world origin
    for (int k = 0; k < n; ++k) {
        const double x = origins[3 * k], y = origins[3 * k + 1], z = origins[3 * k + 2], d2 = x * x + y * y + z * z;
        if (!(d2 <= farthest2)) farthest2 = d2;      // (NaN sticks)
    }
    CrtDevScene S; fill_scene(S, numInstances, fs, beyond_cull_range(sqrt(farthest2)));
    const bool tlas = S.tlasNodes > 0 && (g.forceTlas >= 0 ? (g.forceTlas != 0 && numInstances <= g.instHigh) : (numInstances > CRT_TLAS_MIN_INSTANCES && numInstances <= g.instHigh));
    if (tlas) crt_query_kernel<true><<<(unsigned)((n + CRT_BLOCK - 1) / CRT_BLOCK), CRT_BLOCK, 0, g.stream>>>(S, dO, dD, n, dH, g.counters);
    else crt_query_kernel<false><<<(unsigned)((n + CRT_BLOCK - 1) / CRT_BLOCK), CRT_BLOCK, 0, g.stream>>>(S, dO, dD, n, dH, g.counters);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dH, sizeof(CrtRayHit) * (size_t)n, hipMemcpyDeviceToHost, g.stream));
    unsigned long long c[CRT_NUM_COUNTERS];
    HIPCHK(hipMemcpyAsync(c, g.counters, sizeof c, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    CrtCounters& o = g.lastCounters;
    o.rays = c[0]; o.primary = c[1]; o.secondary = c[2]; o.hits = c[3]; o.misses = c[4]; o.traversals = c[5];
    o.pops = c[6]; o.innerVisits = c[7]; o.triTests = c[8]; o.capHits = c[9]; o.stackOverflows = c[10]; o.maxStack = c[11];
    o.shadowRays = c[12]; o.shadowHits = c[13]; g.lastCulled = c[14];
    return CRT_OK;
}

int crt1_read_output(float* dst, size_t floats)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!dst || floats != (size_t)g.width * (size_t)g.height * 4) return CRT_E_BAD_ARGUMENT;
    RCCHK(sync_all());
    HIPCHK(hipMemcpy(dst, g.slot[g.cur].out, floats * sizeof(float), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt1_read_output_rows(float* dst, int row0, int rows)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!dst || row0 < 0 || rows < 0 || row0 + rows > g.height) return CRT_E_BAD_ARGUMENT;
    RCCHK(sync_all());
    HIPCHK(hipMemcpy(dst, g.slot[g.cur].out + (size_t)row0 * (size_t)g.width, (size_t)rows * (size_t)g.width * sizeof(float4), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt1_read_output_rgba8(uint8_t* dst, size_t bytes)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    const size_t pixels = (size_t)g.width * (size_t)g.height;
    if (!dst || bytes != pixels * 4) return CRT_E_BAD_ARGUMENT;
    RCCHK(sync_all());
    if (pixels * 4 > g.queryBytes) {                       // shares the query scratch buffer
        if (g.queryBuf) (void)hipFree(g.queryBuf);
        g.queryBuf = nullptr; g.queryBytes = 0;
        HIPCHK(hipMalloc(&g.queryBuf, pixels * 4));
        g.queryBytes = pixels * 4;
    }
    crt_pack_unorm8_kernel<<<(unsigned)((pixels + 255) / 256), 256, 0, g.stream>>>(g.slot[g.cur].out, static_cast<uint32_t*>(g.queryBuf), pixels);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dst, g.queryBuf, pixels * 4, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return CRT_OK;
}

int crt1_map_host_frame_back(int framesBack, const void** ptr, size_t* bytes)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    // a pipelined frame's copy lives in its slot until the slot is reused: the last nSlots READBACK frames are reachable
    if (!ptr || framesBack < 0 || (unsigned)framesBack >= g.readbackCount || framesBack >= g.nSlots) return CRT_E_BAD_ARGUMENT;
    const int slot = g.readbackRing[(g.readbackCount - 1u - (unsigned)framesBack) % CRT_MAX_FRAMES_IN_FLIGHT];
    for (int k = 0; k < framesBack; ++k)      // a later frame on the same slot (synchronous frames all use slot 0) has replaced it
        if (g.readbackRing[(g.readbackCount - 1u - (unsigned)k) % CRT_MAX_FRAMES_IN_FLIGHT] == slot) return CRT_E_BAD_ARGUMENT;
    FrameSlot& fs = g.slot[slot];
    HIPCHK(hipEventSynchronize(fs.copied));
    *ptr = fs.hostBuf;
    if (bytes) *bytes = fs.hostBytes;
    return CRT_OK;
}

int crt1_map_host_frame(const void** ptr, size_t* bytes) { return crt1_map_host_frame_back(0, ptr, bytes); }

int crt1_read_rays(float* dst, size_t floats)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!dst || floats != (size_t)g.width * (size_t)g.height * 3) return CRT_E_BAD_ARGUMENT;
    RCCHK(sync_all());
    HIPCHK(hipMemcpy(dst, g.rays, floats * sizeof(float), hipMemcpyDeviceToHost));
    return CRT_OK;
}

void* crt1_output_device_ptr(void) { return g.initialized ? (void*)g.slot[g.cur].out : nullptr; }

float crt1_last_kernel_ms(int which)
{
    if (!g.initialized || which < 0 || which > 3) return -1.0f;
    if (collect_timing() != CRT_OK) return -1.0f;
    return g.ms[which];
}

int crt1_frame_time_stats(CrtFrameStats* out, int reset)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    RCCHK(collect_timing());
    if (out) {
        out->frames = g.framesTimed;
        for (int k = 0; k < 4; ++k) out->sumMs[k] = g.msSum[k];
        out->extentMs = g.statExtent;
        out->firstFrameMs = g.statFirstMs;
    }
    if (reset) {
        for (int k = 0; k < 4; ++k) g.msSum[k] = 0.0;
        g.framesTimed = 0; g.statExtent = 0; g.statFirstMs = 0; g.statStartArmed = true; g.statStartValid = false;
    }
    return CRT_OK;
}

int crt1_debug_read_frame_times(double* dst, size_t maxFrames, size_t* numFrames)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!numFrames) return CRT_E_BAD_ARGUMENT;
    RCCHK(collect_timing());
    *numFrames = g.frameLogN;
    if (dst) memcpy(dst, g.frameLog, sizeof(double) * 2 * (maxFrames < g.frameLogN ? maxFrames : g.frameLogN));
    return CRT_OK;
}

int crt1_debug_read_stamps(uint64_t* dst, size_t maxWaves, size_t* numWaves)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!numWaves) return CRT_E_BAD_ARGUMENT;
    *numWaves = g.stampWaves;
    if (!dst || !g.stamps) return CRT_OK;
    const size_t n = maxWaves < g.stampWaves ? maxWaves : g.stampWaves;
    RCCHK(sync_all());
    HIPCHK(hipMemcpy(dst, g.stamps + 16, n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt1_get_culled_visits(uint64_t* out)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!out) return CRT_E_BAD_ARGUMENT;
    int rc = collect_timing();
    if (rc) return rc;
    *out = g.lastCulled;
    return CRT_OK;
}

// Diagnostic: the range of ray origins the instance cull is proven for. limits[i] = O_i of instance i (0: never culled),
// *sceneLimit = the smallest over the cullable instances (a frame whose camera is farther out runs without the cull),
// *bounceReach = how far from the world origin bounce-ray origins can lie, *noCullFrames = launches that ran without it so far.
int crt1_get_cull_range(float* limits, int n, float* sceneLimit, float* bounceReach, uint64_t* noCullFrames)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (n < 0 || n > CRT_MAX_INSTANCES || (n > 0 && !limits)) return CRT_E_BAD_ARGUMENT;
    for (int i = 0; i < n; ++i) limits[i] = g.hCullOriginLimit[i];
    if (sceneLimit) *sceneLimit = g.cullOriginLimit;
    if (bounceReach) *bounceReach = g.bounceOriginReach;
    if (noCullFrames) *noCullFrames = g.noCullFrames;
    return CRT_OK;
}

int crt1_get_counters(CrtCounters* out)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!out) return CRT_E_BAD_ARGUMENT;
    int rc = collect_timing();
    if (rc) return rc;
    *out = g.lastCounters;
    return CRT_OK;
}

} // namespace

extern "C" {

const char* crt_error_string(int code)
{
    switch (code) {
    case CRT_OK: return "ok";
    case CRT_E_NOT_INITIALIZED: return "crt: not initialized (crt_init failed or was not called)";
    case CRT_E_BAD_ARGUMENT: return "crt: bad argument or invalid scene data";
    case CRT_E_OUT_OF_RANGE: return "crt: upload exceeds a fixed device pool";
    case CRT_E_NO_DEVICE: return "crt: no usable HIP device";
    case CRT_E_UNSUPPORTED: return "crt: unsupported";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "crt: unknown error";
    }
}

int crt_row_owner(int row, int bandRows, int nRanks)
{
    if (row < 0 || bandRows < CRT_TILE || bandRows % CRT_TILE != 0 || nRanks < 1) return CRT_E_BAD_ARGUMENT;
    return (row / bandRows) % nRanks;
}


// ------------------------------------------------------------------------------------------------
// Dispatch: one device (crt_init) or several in one process (crt_init_devices / crt_init_gpus)
//
// The reference drives ONE OpenCL device from one thread (Renderer.cpp:134 asks clGetDeviceIDs for a single GPU). With
// several devices the same C-ABI is kept: the scene is replicated by every upload, the frame is cut into 16-row bands
// dealt round-robin to the devices (device d = rank d of n), every device traces its bands with its own streams and frame
// slots, copies them into the primary device's frame (hipMemcpy2DAsync peer copies, one strided copy per device and frame,
// over xGMI) and the primary's end-of-frame event waits for those copies -- so crt_render keeps upstream's
// Render() + clFinish meaning, crt_read_output / crt_map_host_frame / crt_output_device_ptr return the WHOLE frame, and
// frames in flight work as before. No collective and no host staging in the data path. Each secondary device has a
// host worker thread that submits its share of a frame, so the per-frame submission cost does not grow with the number of
// devices; the calling thread submits the primary's share last (its stream must wait on events the others have recorded).
// ------------------------------------------------------------------------------------------------
struct Worker {
    // Job hand-over by generation counters: the owner bumps `posted`, the worker bumps `finished`. Both sides spin briefly
    // (a frame's share is submitted in ~30 us, a condition-variable wake-up alone costs 5-10 us each way) and fall back
    // to the condition variable, so an idle session does not burn a core.
    std::thread th; std::mutex m; std::condition_variable cv;
    std::function<int()> job; std::atomic<unsigned> posted{0}, finished{0}; std::atomic<bool> quit{false}; int result = 0;
    State* st = nullptr; int device = 0;
    static constexpr int kSpins = 4000;
    void start(State* s, int dev)
    {
        st = s; device = dev;
        th = std::thread([this] {
            (void)hipSetDevice(device);
            G = st;
            unsigned seen = 0;
            for (;;) {
                int spins = 0;
                while (posted.load(std::memory_order_acquire) == seen && !quit.load(std::memory_order_acquire)) {
                    if (++spins < kSpins) { __builtin_ia32_pause(); continue; }
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return posted.load(std::memory_order_acquire) != seen || quit.load(std::memory_order_acquire); });
                }
                if (quit.load(std::memory_order_acquire)) return;
                seen = posted.load(std::memory_order_acquire);
                result = job();
                finished.store(seen, std::memory_order_release);
                { std::lock_guard<std::mutex> lk(m); }
                cv.notify_all();
            }
        });
    }
    void post(std::function<int()> f)
    {
        job = std::move(f);
        { std::lock_guard<std::mutex> lk(m); posted.fetch_add(1, std::memory_order_release); }
        cv.notify_all();
    }
    int wait()
    {
        const unsigned want = posted.load(std::memory_order_acquire);
        int spins = 0;
        while (finished.load(std::memory_order_acquire) != want) {
            if (++spins < kSpins) { __builtin_ia32_pause(); continue; }
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return finished.load(std::memory_order_acquire) == want; });
        }
        return result;
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(m); quit.store(true, std::memory_order_release); }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

struct Group {
    int n = 0;                                  // 0: no session; 1: crt_init; >1: crt_init_devices
    State* dev[CRT_MAX_DEVICES] = { nullptr };
    int hipDevice[CRT_MAX_DEVICES] = { 0 };
    Worker* worker[CRT_MAX_DEVICES] = { nullptr };
    // how device d's bands reach the primary's frame: 2 = same physical GPU as the primary (rehearsal), 1 = peer mapping
    // (hipDeviceEnablePeerAccess: xGMI), 0 = no peer access, the runtime stages the copy through host memory
    int peer[CRT_MAX_DEVICES] = { 0 };
    unsigned seq = 0;                           // frame-slot rotation of the session (crt_render)
    bool broken = false;                        // a resize failed on some device and could not be rolled back
    int injectFailure = -1;                     // crt_debug_inject_failure
} M;

// selects device d of the session for the calling thread; the primary is re-selected when the scope ends
struct Use {
    explicit Use(int d) { select(d); }
    ~Use() { if (M.n > 1) select(0); }
    static void select(int d) { if (M.n > d && M.dev[d]) { if (M.n > 1) (void)hipSetDevice(M.hipDevice[d]); G = M.dev[d]; } else G = nullptr; }
};
#define NEED_SESSION() do { if (M.n == 0) return CRT_E_NOT_INITIALIZED; } while (0)
// run `expr` on every device of the session (scene uploads, resize, ...); first error wins
#define ON_ALL(expr) do { NEED_SESSION(); int rc_ = CRT_OK; for (int d_ = 0; d_ < M.n; ++d_) { Use u_(d_); const int r_ = (expr); if (r_ != CRT_OK && rc_ == CRT_OK) rc_ = r_; } return rc_; } while (0)
#define ON_PRIMARY(expr) do { NEED_SESSION(); Use u_(0); return (expr); } while (0)

static void destroy_group()
{
    for (int d = 1; d < M.n; ++d) if (M.worker[d]) { M.worker[d]->stop(); delete M.worker[d]; M.worker[d] = nullptr; }
    for (int d = 0; d < M.n; ++d) {
        if (!M.dev[d]) continue;
        if (M.n > 1) (void)hipSetDevice(M.hipDevice[d]);
        G = M.dev[d];
        release_all();
        delete M.dev[d]; M.dev[d] = nullptr;
    }
    G = nullptr; M.n = 0; M.seq = 0; M.broken = false; M.injectFailure = -1;
    for (int& p : M.peer) p = 0;
}

// pure: the block list a rank's gather / read-back copies (needs no device; tests/test_distributed.py)
int crt_band_plan(int height, int bandRows, int rank, int nRanks, int out[4])
{
    if (!out || height < 0 || bandRows < CRT_TILE || bandRows % CRT_TILE != 0 || nRanks < 1 || rank < 0 || rank >= nRanks) return CRT_E_BAD_ARGUMENT;
    const BandPlan p = band_plan(height, bandRows, rank, nRanks);
    out[0] = p.firstRow; out[1] = p.fullBands; out[2] = p.tailRow; out[3] = p.tailRows;
    return CRT_OK;
}

// More than four frame slots want one hardware queue per slot stream; the HIP runtime reads GPU_MAX_HW_QUEUES when it starts
// (its first call in the process), so this must run before any HIP call: both initialisers call it first. A caller that
// has already used HIP (e.g. through another library) must export the variable itself.
static void raise_hw_queues()
{
    const char* e = getenv("CRT_FRAMES_IN_FLIGHT");
    if (e && atoi(e) > 4) (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

int crt_init_devices(const int* devices, int numDevices, int width, int height)
{
    if (M.n != 0) return CRT_E_BAD_ARGUMENT;
    if (!devices || numDevices < 1 || numDevices > CRT_MAX_DEVICES) return CRT_E_BAD_ARGUMENT;
    raise_hw_queues();
    int rc = CRT_OK;
    M.n = numDevices;
    for (int d = 0; d < numDevices && rc == CRT_OK; ++d) {
        M.hipDevice[d] = devices[d];
        M.dev[d] = new State();
        G = M.dev[d];
        rc = init_impl(devices[d], width, height);           // selects the HIP device
        M.peer[d] = 2;
        if (rc == CRT_OK && numDevices > 1) {
            g.bandRows = 16; g.rank = d; g.nRanks = numDevices;
            g.primary = M.dev[0]; g.groupSize = numDevices;
            if (d > 0 && devices[d] != devices[0]) {
                // the gather copies this device's bands into the primary's frame from this device's stream: it needs the
                // primary's memory mapped here (xGMI peer access). Without it the runtime stages every copy through host
                // memory -- correct but slow, so the state is recorded (crt_peer_access) and reported once.
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, devices[d], devices[0]);
                M.peer[d] = 0;
                if (can) {
                    const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                    if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) M.peer[d] = 1; else rc = (int)e;
                    (void)hipGetLastError();
                }
                if (M.peer[d] == 0 && rc == CRT_OK)
                    fprintf(stderr, "crt_init_devices: device %d has no peer access to device %d: its bands are gathered through host memory\n", devices[d], devices[0]);
            }
        }
    }
    if (rc == CRT_OK && numDevices > 1) {
        for (int d = 0; d < numDevices; ++d) for (int k = 0; k < numDevices; ++k) M.dev[d]->group[k] = M.dev[k];
        for (int d = 1; d < numDevices; ++d) { M.worker[d] = new Worker(); M.worker[d]->start(M.dev[d], devices[d]); }
    }
    if (rc != CRT_OK) { destroy_group(); return rc; }
    Use::select(0);
    return CRT_OK;
}

int crt_init(int device, int width, int height) { return crt_init_devices(&device, 1, width, height); }

int crt_init_gpus(int numGpus, int width, int height)
{
    raise_hw_queues();                          // before the first call that can start the HIP runtime
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return CRT_E_NO_DEVICE;
    if (numGpus < 1 || numGpus > have || numGpus > CRT_MAX_DEVICES) return CRT_E_BAD_ARGUMENT;
    int ids[CRT_MAX_DEVICES];
    for (int d = 0; d < numGpus; ++d) ids[d] = d;
    return crt_init_devices(ids, numGpus, width, height);
}

int crt_num_devices(void) { return M.n; }
int crt_peer_access(int device) { if (M.n == 0) return CRT_E_NOT_INITIALIZED; if (device < 0 || device >= M.n) return CRT_E_BAD_ARGUMENT; return M.peer[device]; }
const char* crt_gather_path(void)
{
    if (M.n <= 1) return "none (one device)";
    int lo = 2;
    for (int d = 1; d < M.n; ++d) if (M.peer[d] < lo) lo = M.peer[d];
    return lo == 2 ? "same-device copies (rehearsal: one GPU listed several times)" : (lo == 1 ? "xgmi-peer" : "host-staged");
}
// armed only in a process that asked for the test hooks (CRT_DEBUG_HOOKS=1): a production caller cannot make a frame fail by accident
int crt_debug_inject_failure(int device)
{
    NEED_SESSION();
    { const char* e = getenv("CRT_DEBUG_HOOKS"); if (!e || atoi(e) == 0) return CRT_E_UNSUPPORTED; }
    if (device < 0 || device >= M.n) return CRT_E_BAD_ARGUMENT;
    M.injectFailure = device;
    return CRT_OK;
}
int crt_debug_staggered_frames(uint64_t* out) { NEED_SESSION(); if (!out) return CRT_E_BAD_ARGUMENT; Use u_(0); *out = g.staggeredFrames; return CRT_OK; }
int crt_debug_measure_clock(int micros, double* ghz) { ON_PRIMARY(crt1_debug_measure_clock(micros, ghz)); }

int crt_shutdown(void)
{
    NEED_SESSION();
    destroy_group();
    return CRT_OK;
}

const char* crt_device_name(void) { if (M.n == 0) return ""; Use u(0); return crt1_device_name(); }

int crt_resize(int width, int height)
{
    NEED_SESSION();
    if (M.n == 1) { Use u(0); return crt1_resize(width, height); }
    // all devices or none: a device that cannot reallocate its frame buffers sends the others back to the old size; if that
    // fails too the devices disagree about the frame and the session refuses to render (crt_shutdown is what is left)
    int oldW, oldH;
    { Use u(0); oldW = g.width; oldH = g.height; }
    int rc = CRT_OK, done = 0;
    for (; done < M.n && rc == CRT_OK; ++done) { Use u(done); rc = (done == M.injectFailure) ? (int)CRT_E_UNSUPPORTED : crt1_resize(width, height); }
    if (M.injectFailure >= 0) M.injectFailure = -1;
    if (rc == CRT_OK) return CRT_OK;
    for (int d = 0; d < done - 1; ++d) { Use u(d); if (crt1_resize(oldW, oldH) != CRT_OK) M.broken = true; }
    return rc;
}
int crt_set_row_bands(int bandRows, int rank, int nRanks)
{
    NEED_SESSION();
    if (M.n > 1) return CRT_E_UNSUPPORTED;                  // the bands belong to the devices of this session
    ON_PRIMARY(crt1_set_row_bands(bandRows, rank, nRanks));
}
int crt_owned_rows(void) { if (M.n == 0) return 0; Use u(0); return M.n > 1 ? g.height : crt1_owned_rows(); }

int crt_upload_triangles(const void* tris, size_t byteOffset, size_t bytes) { ON_ALL(crt1_upload_triangles(tris, byteOffset, bytes)); }
int crt_upload_bvh_nodes(const void* nodes, size_t byteOffset, size_t bytes) { ON_ALL(crt1_upload_bvh_nodes(nodes, byteOffset, bytes)); }
int crt_upload_bvh_roots(const uint32_t* roots, size_t firstMesh, size_t count) { ON_ALL(crt1_upload_bvh_roots(roots, firstMesh, count)); }
int crt_upload_materials(const void* materials, size_t first, size_t count) { ON_ALL(crt1_upload_materials(materials, first, count)); }
int crt_upload_texture_table(const void* textures, size_t count) { ON_ALL(crt1_upload_texture_table(textures, count)); }
int crt_upload_texels(const void* rgb8, size_t byteOffset, size_t bytes) { ON_ALL(crt1_upload_texels(rgb8, byteOffset, bytes)); }
int crt_upload_instances(const void* instances, size_t first, size_t count) { ON_ALL(crt1_upload_instances(instances, first, count)); }
// every device builds its own copy (the builder is deterministic: same bytes everywhere); downloads read the primary's
int crt_build_bvh(size_t firstTri, const uint32_t* meshTriCounts, int numMeshes, size_t firstNode, size_t firstMesh, uint32_t* nodesUsedOut)
{ ON_ALL(crt1_build_bvh(firstTri, meshTriCounts, numMeshes, firstNode, firstMesh, nodesUsedOut)); }
int crt_download_triangles(void* dst, size_t byteOffset, size_t bytes) { ON_PRIMARY(crt1_download_triangles(dst, byteOffset, bytes)); }
int crt_download_bvh_nodes(void* dst, size_t byteOffset, size_t bytes) { ON_PRIMARY(crt1_download_bvh_nodes(dst, byteOffset, bytes)); }
int crt_download_bvh_roots(uint32_t* dst, size_t firstMesh, size_t count) { ON_PRIMARY(crt1_download_bvh_roots(dst, firstMesh, count)); }

int crt_render(const CrtTraceArgs* args, const float invView[16], const float invProj[16], int flags)
{
    NEED_SESSION();
    if (M.n == 1) { Use u(0); return crt1_render(args, invView, invProj, flags); }
    if (M.broken) return CRT_E_BAD_ARGUMENT;                 // a failed resize left the devices at different frame sizes
    if (!args || !invView || !invProj) return CRT_E_BAD_ARGUMENT;
    if (flags & (CRT_RENDER_WRITE_RAYS | CRT_RENDER_STAMPS)) return CRT_E_UNSUPPORTED;   // single-device diagnostics
    // The dispatcher decides the frame slot once for every device (the devices' own rotation counters are not used in a
    // session: a device that owns no rows of a short frame, or whose submission failed, stays in step).
    RenderPlan plan;
    { Use u(0); plan.slot = frame_is_pipelined(flags) ? (int)(M.seq++ % (unsigned)g.nSlots) : 0; }
    plan.noHostWait = true;
    // secondaries first, each on its own worker thread (they record the events the primary's stream then waits on) and
    // without waiting for their devices; their copy of the frame never leaves the device except through the gather, so
    // READBACK is the primary's business
    struct Job { CrtTraceArgs a; float iv[16], ip[16]; int flags; RenderPlan plan; } job;
    job.a = *args; memcpy(job.iv, invView, 64); memcpy(job.ip, invProj, 64); job.flags = flags & ~CRT_RENDER_READBACK; job.plan = plan;
    const int failDevice = M.injectFailure;                  // test hook (crt_debug_inject_failure): that device's submission fails once
    M.injectFailure = -1;
    for (int d = 1; d < M.n; ++d) {
        if (d == failDevice) M.worker[d]->post([]() { return (int)CRT_E_UNSUPPORTED; });
        else M.worker[d]->post([job]() { return crt1_render(&job.a, job.iv, job.ip, job.flags, &job.plan); });
    }
    int rc = CRT_OK;
    for (int d = 1; d < M.n; ++d) { const int r = M.worker[d]->wait(); if (r != CRT_OK && rc == CRT_OK) rc = r; }
    // A secondary failed: its bands will not arrive, so the primary must not queue a wait for them (it would wait on the
    // slot's previous frame's event and present a frame with stale bands as complete). The frame is abandoned: the error
    // is returned, nothing of it is readable, and the slot rotation has advanced on every device alike.
    if (rc != CRT_OK) return rc;
    if (failDevice == 0) return CRT_E_UNSUPPORTED;
    Use u(0);
    plan.noHostWait = false;
    return crt1_render(args, invView, invProj, flags, &plan);
}

int crt_sync(void) { ON_ALL(crt1_sync()); }

// reads go to the primary, which holds the gathered frame -- after every device has drained
static int drain_secondaries() { for (int d = 1; d < M.n; ++d) { Use u(d); const int r = crt1_sync(); if (r != CRT_OK) return r; } return CRT_OK; }
int crt_query_hits(const float* origins, const float* dirs, int n, uint32_t numInstances, CrtRayHit* out) { ON_PRIMARY(crt1_query_hits(origins, dirs, n, numInstances, out)); }
int crt_read_output(float* dst, size_t floats) { NEED_SESSION(); RCCHK(drain_secondaries()); ON_PRIMARY(crt1_read_output(dst, floats)); }
int crt_read_output_rows(float* dst, int row0, int rows) { NEED_SESSION(); RCCHK(drain_secondaries()); ON_PRIMARY(crt1_read_output_rows(dst, row0, rows)); }
int crt_read_output_rgba8(uint8_t* dst, size_t bytes) { NEED_SESSION(); RCCHK(drain_secondaries()); ON_PRIMARY(crt1_read_output_rgba8(dst, bytes)); }
int crt_map_host_frame_back(int framesBack, const void** ptr, size_t* bytes) { ON_PRIMARY(crt1_map_host_frame_back(framesBack, ptr, bytes)); }
int crt_map_host_frame(const void** ptr, size_t* bytes) { return crt_map_host_frame_back(0, ptr, bytes); }
int crt_read_rays(float* dst, size_t floats) { NEED_SESSION(); if (M.n > 1) return CRT_E_UNSUPPORTED; ON_PRIMARY(crt1_read_rays(dst, floats)); }
void* crt_output_device_ptr(void) { if (M.n == 0) return nullptr; Use u(0); return crt1_output_device_ptr(); }
float crt_last_kernel_ms(int which) { if (M.n == 0) return -1.0f; Use u(0); return crt1_last_kernel_ms(which); }
int crt_frame_time_stats(CrtFrameStats* out, int reset)
{
    NEED_SESSION();
    for (int d = 1; d < M.n; ++d) { Use u(d); RCCHK(crt1_frame_time_stats(nullptr, reset)); }     // keeps the secondaries' event sets collected
    ON_PRIMARY(crt1_frame_time_stats(out, reset));
}
int crt_debug_read_stamps(uint64_t* dst, size_t maxWaves, size_t* numWaves) { ON_PRIMARY(crt1_debug_read_stamps(dst, maxWaves, numWaves)); }
int crt_debug_read_frame_times(double* dst, size_t maxFrames, size_t* numFrames) { ON_PRIMARY(crt1_debug_read_frame_times(dst, maxFrames, numFrames)); }

// work counters of the last counted frame: the sum over the devices (maxStack: the maximum)
int crt_get_counters(CrtCounters* out)
{
    NEED_SESSION();
    if (!out) return CRT_E_BAD_ARGUMENT;
    CrtCounters total; memset(&total, 0, sizeof total);
    unsigned long long maxStack = 0;
    for (int d = 0; d < M.n; ++d) {
        Use u(d);
        CrtCounters c;
        RCCHK(crt1_get_counters(&c));
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&c);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(&total);
        for (size_t k = 0; k < sizeof(CrtCounters) / sizeof(unsigned long long); ++k) dst[k] += src[k];
        if (c.maxStack > maxStack) maxStack = c.maxStack;
    }
    total.maxStack = maxStack;
    *out = total;
    return CRT_OK;
}
int crt_get_cull_range(float* limits, int n, float* sceneLimit, float* bounceReach, uint64_t* noCullFrames) { ON_PRIMARY(crt1_get_cull_range(limits, n, sceneLimit, bounceReach, noCullFrames)); }
int crt_get_culled_visits(uint64_t* out)
{
    NEED_SESSION();
    if (!out) return CRT_E_BAD_ARGUMENT;
    uint64_t total = 0;
    for (int d = 0; d < M.n; ++d) { Use u(d); uint64_t c = 0; RCCHK(crt1_get_culled_visits(&c)); total += c; }
    *out = total;
    return CRT_OK;
}
} // extern "C"
