// crt_shim.hip -- the C-ABI (include/crt_api.h, include/crt_debug.h) of the MI355X ray-trace path: the extern "C" block only.
//
// Replaces the OpenCL side of the reference's Renderer.cpp / ResourceManager.cpp: device pools,
// uploads, the per-frame RayGen -> Trace -> PostProcess launch (Renderer.cpp:305-375). Uploads
// arrive in the reference's struct layouts and are re-laid-out on the device (crt_device.h).
// One translation unit; the parts (round 5 split what used to be one 2,000-line file):
//   kernels     crt_device.h (traversal + shading), crt_kernels.h (launches), crt_refill.h (opt-in in-wave compaction forms), crt_ldstop.h (opt-in: tree tops staged in LDS),
//               crt_relayout.h (upload-time layouts), crt_bvh_build.h (device BuildBVH)
//   host state  crt_state.h (State / FrameSlot, helpers), crt_instances.h (instance tables, cull bounds, instance tree)
//   entry impl  crt_upload.h (init, uploads, read-backs), crt_bvh_driver.h (crt_build_bvh), crt_frame.h (crt_render and what a frame
//               needs), crt_multidev.h (several devices behind the same calls)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/crt_api.h"
#include "../../include/crt_debug.h"
#include "crt_kernels.h"
#include "crt_refill.h"
#include "crt_ldstop.h"
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
#include "crt_state.h"
#include "crt_instances.h"
#include "crt_upload.h"
#include "crt_bvh_driver.h"
#include "crt_frame.h"

#include "crt_multidev.h"

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
// (r6: raised for every session, not only for more than four slots -- with the default of four queues two of three slot streams can end up on
// one queue, whose frames then serialise: 16.6 -> 12.3 Gray/s on nanosuit-demo, profiles/r06_hw_queues.txt. Never overrides the caller's setting.)
static void raise_hw_queues()
{
    (void)setenv("GPU_MAX_HW_QUEUES", "8", 0);
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
int crt_debug_tlas_stats(uint64_t* builds, uint32_t* refitsSinceBuild, uint32_t* nodes)
{
    NEED_SESSION();
    Use u_(0);
    if (builds) *builds = g.hTlasBuilds;
    if (refitsSinceBuild) *refitsSinceBuild = g.hTlasRefits;
    if (nodes) *nodes = g.hTlasNodes;
    return CRT_OK;
}
int crt_debug_build_stats(uint32_t* levels, uint32_t* launches)
{
    NEED_SESSION();
    Use u_(0);
    if (levels) *levels = g.buildLevels;
    if (launches) *launches = g.buildLaunches;
    return CRT_OK;
}
int crt_debug_last_gather(uint64_t* bytes, int* bytesPerPixel)
{
    NEED_SESSION();
    if (bytes) *bytes = M.n > 1 ? M.lastGatherBytes : 0;
    if (bytesPerPixel) *bytesPerPixel = M.n > 1 ? M.lastGatherBpp : 0;
    return CRT_OK;
}
int crt_debug_last_kernel(char* dst, size_t cap)
{
    NEED_SESSION();
    if (!dst || cap == 0) return CRT_E_BAD_ARGUMENT;
    Use u_(0);
    snprintf(dst, cap, "%s", g.lastKernel);
    return CRT_OK;
}
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
    // RGBA8 frames travel as bytes (frame_gathers_rgba8): every device's byte frames exist before a secondary copies into the primary's
    bool gather8 = false;
    { Use u(0); gather8 = frame_gathers_rgba8(flags); }
    if (gather8) {
        int w, h;
        { Use u(0); w = g.width; h = g.height; }
        if (M.pack8W != w || M.pack8H != h) {                   // once per frame size, not per frame (16 hipSetDevice calls at 8 devices otherwise)
            for (int d = 0; d < M.n; ++d) { Use u(d); RCCHK(crt1_prepare_gather8()); }
            M.pack8W = w; M.pack8H = h;
        }
    }
    {   // what the secondaries send into the primary for this frame (crt_debug_last_gather)
        Use u(0);
        unsigned long long rows = 0;
        for (int d = 1; d < M.n; ++d) { const BandPlan p = band_plan(g.height, g.bandRows, d, M.n); rows += (unsigned long long)p.fullBands * (unsigned)g.bandRows + (unsigned)p.tailRows; }
        M.lastGatherBpp = gather8 ? 4 : 16; M.lastGatherBytes = rows * (unsigned long long)g.width * (unsigned)M.lastGatherBpp;
    }
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
