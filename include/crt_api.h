/* crt_api.h -- C-ABI of the MI355X ray-trace path (libcrt_hip.so).
 *
 * This is the drop-in boundary: every entry point replaces one group of OpenCL call sites of the
 * reference's Renderer.cpp / ResourceManager.cpp (cited per function, paths relative to the
 * upstream tree CLRayTracer/...). Plain pointers and sizes only; buffers are passed in the
 * reference's own struct layouts (crt_types.h) and re-laid-out for CDNA4 on the device.
 *
 * Conventions
 *   - every function returns 0 on success, a positive hipError_t on a HIP failure, or a negative
 *     CRT_E_* code for argument/state errors; crt_error_string() explains either.
 *   - state is process-global and NOT thread-safe, like the reference's file-static state
 *     (Renderer.cpp:23-39, ResourceManager.cpp:49-90). One process drives one GPU.
 *   - host pointers only need to stay valid for the duration of the call (uploads are
 *     synchronous; the reference's CL_FALSE writes required the arenas to outlive the queue).
 *   - there is no CPU fallback: without a usable GPU crt_init fails and everything else returns
 *     CRT_E_NOT_INITIALIZED.
 */
#ifndef CRT_API_H
#define CRT_API_H

#include <stddef.h>
#include <stdint.h>
#include "crt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    CRT_OK = 0,
    CRT_E_NOT_INITIALIZED = -1,
    CRT_E_BAD_ARGUMENT    = -2,
    CRT_E_OUT_OF_RANGE    = -3,   /* upload beyond a fixed-size device pool */
    CRT_E_NO_DEVICE       = -4,
    CRT_E_UNSUPPORTED     = -5
};

/* crt_render flags */
enum {
    CRT_RENDER_POSTPROCESS = 1,   /* also run PostProcess (kernel_main.cl:342-359) on the output */
    CRT_RENDER_WRITE_RAYS  = 2,   /* materialise the RayGen buffer (kernel_main.cl:277-287) in HBM */
    CRT_RENDER_ASYNC       = 4,   /* do not wait for completion (the reference always clFinish()es); see crt_render */
    CRT_RENDER_COUNTERS    = 8,   /* instrumented launch that fills the work counters (slower) */
    CRT_RENDER_STAMPS      = 16,  /* diagnostic launch: per-wave start/end clock stamps (crt_debug_read_stamps) */
    CRT_RENDER_SHADOWS     = 32,  /* extension (kernel_main.cl:256-258 is a TODO upstream): one any-hit shadow ray from the first
                                     hit towards the sun sets the `shadow` factor of kernel_main.cl:264; see DESIGN.md */
    CRT_RENDER_UNORM8      = 64,  /* hazard H8: upstream renders into an RGBA8-UNORM texture (Renderer.cpp:63,192): quantise the
                                     Trace result like write_imagef/read_imagef before PostProcess and the final frame after it */
    CRT_RENDER_READBACK    = 128, /* also copy the finished frame to pinned host memory behind its kernels (float4, or RGBA8 bytes
                                     with CRT_RENDER_UNORM8); fetch it with crt_map_host_frame. Overlaps with the other frames in flight */
    CRT_RENDER_REFRACTION  = 256, /* extension (upstream README TODO "refraction / transculency", no upstream code): at the first hit
                                     of a material whose MTL `d` (opacity, Material::roughness) is below 1 the bounce ray is the
                                     refracted ray (index 1.5) with (1 - opacity) of the energy; defined by the oracle, see DESIGN.md */
    CRT_RENDER_DIAG_MIX3   = 1024, /* diagnostic (profiling aid): ONE Trace dispatch that traces the frame three times, tile lists
                                     interleaved a third of the frame apart -- the wave mix of three frames in flight in a dispatch a
                                     PMC pass can see (rocprofv3 serialises dispatches, so real frames in flight cannot be profiled).
                                     Same pixels (every copy stores the same value). One device, synchronous, default kernel only */
    CRT_RENDER_FXAA        = 512  /* extension: upstream's FXAA function (kernel_main.cl:289-340) is dead code -- its call is commented out
                                     (kernel_main.cl:349), it returns nothing and would read pixels PostProcess is rewriting. Run it
                                     as the first PostProcess stage (or alone, without CRT_RENDER_POSTPROCESS), reading the unmodified
                                     Trace result; semantics defined by the oracle (orc_fxaa). Whole frames only: refused while
                                     crt_set_row_bands leaves this device a share of the rows (an in-process multi-GPU session
                                     gathers first and filters on its first device) */
};

/* Device work counters of the last CRT_RENDER_COUNTERS / crt_query_hits launch. Same meaning as
 * the oracle's OrcStats so tests can require exact equality. */
typedef struct CrtCounters {
    uint64_t rays, primary, secondary, hits, misses;
    uint64_t traversals, pops, innerVisits, triTests, capHits, stackOverflows, maxStack;
    uint64_t shadowRays, shadowHits;   /* CRT_RENDER_SHADOWS: shadow rays traced (also in `rays`) / found occluded */
} CrtCounters;

/* Renderer.cpp:122-193 (InitializeOpenCL + buffer creation) and ResourceManager.cpp:145-178
 * (device pools, default white/black texels). `device` is the HIP ordinal this process owns.
 * Allocates every pool at its reference capacity once; nothing is resized later except by
 * crt_resize. */
int crt_init(int device, int width, int height);
/* Several GPUs of one node behind the same entry points (no reference counterpart: Renderer.cpp:134 asks
 * clGetDeviceIDs for ONE device). The scene is replicated by every upload; the frame is cut into 16-row
 * bands dealt round-robin to the devices; every device traces its bands on its own streams and copies them into the first
 * device's frame (peer copies over xGMI, no collective, no host staging). crt_render keeps its meaning -- without
 * CRT_RENDER_ASYNC it returns when the WHOLE frame is complete -- and crt_read_output / crt_map_host_frame /
 * crt_output_device_ptr deliver the whole frame from the first device. Counters are summed over the devices.
 * Not available in such a session: crt_set_row_bands (the bands are the library's), CRT_RENDER_WRITE_RAYS, CRT_RENDER_STAMPS.
 * `devices` may name the same GPU more than once (functional rehearsal of the multi-device path on a one-GPU box). */
int crt_init_devices(const int* devices, int numDevices, int width, int height);
int crt_init_gpus(int numGpus, int width, int height);        /* devices 0 .. numGpus-1 */
int crt_num_devices(void);                                    /* 0 = no session */
/* How device `device` of the session (0 = the primary) delivers its bands into the primary's frame: 2 = it is the same
 * physical GPU as the primary (a rehearsal session), 1 = peer mapping enabled (hipDeviceEnablePeerAccess: xGMI), 0 = no peer
 * access -- the HIP runtime stages every gather copy through host memory (correct, slow; crt_init_devices says so once on
 * stderr). crt_gather_path() names the slowest path any device of the session uses: "xgmi-peer", "host-staged", ... */
int crt_peer_access(int device);
const char* crt_gather_path(void);
/* Renderer.cpp:377-394, ResourceManager.cpp:303-319 */
int crt_shutdown(void);
/* Renderer.cpp:198-211: ignores sizes below 16 like the reference (returns CRT_OK, no change). */
int crt_resize(int width, int height);
/* Multi-GPU image tiling (no reference counterpart: upstream is single-device). The frame is cut
 * into horizontal bands of `bandRows` rows (a multiple of 8, the tile height); this process renders bands
 * rank, rank+nRanks, ... RayGen and Vignette still use full-frame coordinates. Default (16,0,1). */
int crt_set_row_bands(int bandRows, int rank, int nRanks);
/* Which rank renders frame row `row` under that tiling (pure function, needs no device). */
int crt_row_owner(int row, int bandRows, int nRanks);
/* The rows a rank owns as the block list its gather / read-back copies use (pure function): out = { firstRow, fullBands,
 * tailRow, tailRows }: `fullBands` bands of bandRows rows from firstRow, every bandRows * nRanks rows, then tailRows rows
 * of a last partial band at tailRow. */
int crt_band_plan(int height, int bandRows, int rank, int nRanks, int out[4]);

/* ResourceManager.cpp:286 -- triangles in the 80-byte reference layout, offsets in bytes. */
int crt_upload_triangles(const void* tris, size_t byteOffset, size_t bytes);
/* ResourceManager.cpp:293 -- BVH nodes (32 B each), offsets in bytes, indices as built on the host.
 * Trees from BuildBVH / crt_build_bvh bound their triangles; uploaded nodes need not (leaf triangles are tested without a box test,
 * kernel_main.cl:135-140). The instance cull stays exact either way: the range of bounce-ray origins it is proven for is taken from the
 * uploaded triangles' extents as well as from the root boxes (crt_get_cull_range: bounceReach). */
int crt_upload_bvh_nodes(const void* nodes, size_t byteOffset, size_t bytes);
/* ResourceManager.cpp:291 -- per-mesh root node indices (uint32). */
int crt_upload_bvh_roots(const uint32_t* roots, size_t firstMesh, size_t count);
/* ResourceManager.cpp:142,295 */
int crt_upload_materials(const void* materials, size_t first, size_t count);
/* ResourceManager.cpp:236 */
int crt_upload_texture_table(const void* textures, size_t count);
/* ResourceManager.cpp:177,203 -- packed RGB8 bytes at a byte offset into the texel pool. */
int crt_upload_texels(const void* rgb8, size_t byteOffset, size_t bytes);
/* Renderer.cpp:245,314. Host-side only: the instance table is versioned, frames already submitted keep the version they were
 * submitted with and every later frame refreshes its slot's device copy on its own stream -- no waiting for frames in flight. */
int crt_upload_instances(const void* instances, size_t first, size_t count);

/* BuildBVH (BVH.cpp:218-255; called from ResourceManager.cpp:282) on the device, for triangles already uploaded with
 * crt_upload_triangles: `numMeshes` meshes of meshTriCounts[m] triangles each, stored back to back from triangle
 * `firstTri`. Writes the triangle centroids, reorders the triangles, writes the nodes from node index `firstNode` and
 * the roots of meshes firstMesh.. -- byte for byte what the host BuildBVH produces for the same input (triangle
 * order, node numbering by the recursion's allocation order, bounds) -- and re-lays everything out for rendering.
 * *nodesUsedOut = number of nodes written. The crt_download_* calls read the reference-layout pools back. */
int crt_build_bvh(size_t firstTri, const uint32_t* meshTriCounts, int numMeshes, size_t firstNode, size_t firstMesh, uint32_t* nodesUsedOut);
int crt_download_triangles(void* dst, size_t byteOffset, size_t bytes);
int crt_download_bvh_nodes(void* dst, size_t byteOffset, size_t bytes);
int crt_download_bvh_roots(uint32_t* dst, size_t firstMesh, size_t count);

/* Renderer.cpp:337-367: RayGen + Trace (+ PostProcess) for one frame, then (unless ASYNC) wait.
 * invView / invProj are the camera's inverse matrices, row-major (hazard H10: taken as inputs).
 * Frames in flight (no reference counterpart): consecutive CRT_RENDER_ASYNC frames rotate over three frame
 * slots (CRT_FRAMES_IN_FLIGHT=1..8 in the environment, default 3; every slot's stream wants a hardware queue of its own, so
 * crt_init / crt_init_devices / crt_init_gpus set GPU_MAX_HW_QUEUES=8 before their first HIP call unless it is set -- effective only if
 * nothing in the process has started the HIP runtime before; otherwise export it yourself), each with its own HIP stream, output buffer and
 * launch lists, so frames run concurrently and the long-ray tail of one is hidden behind the others; the call
 * blocks only to keep at most two frames queued per slot. Mesh / texture / material uploads, resize, queries and reads
 * wait for every frame in flight first; instance uploads do not need to (see crt_upload_instances). Either way scene
 * edits between frames stay ordered. crt_read_output* and crt_output_device_ptr refer
 * to the most recently submitted frame. */
int crt_render(const CrtTraceArgs* args, const float invView[16], const float invProj[16], int flags);
int crt_sync(void);                                           /* wait for every frame in flight */

/* Closest-hit query for arbitrary world-space rays (host pointers, n rays) against the first
 * `numInstances` instances: the instance loop + IntersectBVH of kernel_main.cl:198-217 exposed for
 * hit-record parity tests. Also fills the work counters. */
int crt_query_hits(const float* origins, const float* dirs, int n, uint32_t numInstances, CrtRayHit* out);

/* Output: the HDR float4 frame (the reference writes a CL-GL RGBA8 texture, Renderer.cpp:63,192). */
int crt_read_output(float* dstRGBA, size_t floats);           /* full frame, width*height*4 floats */
int crt_read_output_rows(float* dstRGBA, int row0, int rows); /* rows [row0,row0+rows) */
int crt_read_output_rgba8(uint8_t* dstRGBA, size_t bytes);    /* the frame as RGBA8 (convert_uchar_sat_rte(x*255)), width*height*4 bytes */
/* Host copy of the most recent CRT_RENDER_READBACK frame: waits for that copy only; the pointer (pinned memory owned by
 * the library) stays valid until as many further READBACK frames as there are frame slots have been submitted. */
int crt_map_host_frame(const void** ptr, size_t* bytes);
/* The same for the READBACK frame submitted `framesBack` READBACK frames earlier (0 = the latest), while its slot has not
 * been reused: lets a consumer work on frame k-1 while frame k renders. */
int crt_map_host_frame_back(int framesBack, const void** ptr, size_t* bytes);
int crt_read_rays(float* dst, size_t floats);                 /* width*height*3, after WRITE_RAYS */
void* crt_output_device_ptr(void);
int crt_owned_rows(void);                                     /* rows this rank renders per frame */

/* Timing of the last crt_render measured with HIP events on the launch stream.
 * which: 0 = whole frame, 1 = RayGen (only with WRITE_RAYS), 2 = Trace, 3 = PostProcess (and the RGBA8 stores, FXAA) --
 * close to zero for the default kernel, which applies PostProcess and the RGBA8 target to the pixel in its registers
 * before storing it; the stages run as launches of their own behind FXAA and the opt-in kernel variants. */
float crt_last_kernel_ms(int which);
/* Event timing accumulated over every frame since the last reset. Read back lazily per frame slot, so this does
 * not serialise ASYNC frames the way asking crt_last_kernel_ms after each frame would. With frames in flight the
 * per-frame durations overlap: sumMs[2] / frames is the mean duration of one Trace launch (what a kernel trace
 * shows), extentMs / frames the device time the frames took per frame. */
typedef struct CrtFrameStats {
    uint64_t frames;
    double sumMs[4];      /* same four intervals as crt_last_kernel_ms */
    double extentMs;      /* start of the first frame -> end of the last one to finish */
    double firstFrameMs;  /* start -> end of the first frame since the reset: the pipeline's fill time. (extentMs - firstFrameMs) /
                             (frames - 1) is the steady-state device time per frame, independent of how many frames were timed */
} CrtFrameStats;
int crt_frame_time_stats(CrtFrameStats* out, int reset);
int crt_get_counters(CrtCounters* out);
/* Of the last counted launch's `traversals` (= pops = root visits, one per ray and instance as upstream spends them,
 * kernel_main.cl:198-217), how many the conservative instance cull answered without fetching anything: the device's
 * real child-pair fetches are innerVisits - this. (Measurement aid for bench.py's gather-rate figure.) */
int crt_get_culled_visits(uint64_t* out);
/* The range of ray origins for which the instance cull is provably exact (derivation: csrc/crt_device.h above sphere_culls):
 * limits[i] (i < n <= 401) = the largest |origin| instance i may be culled for, 0 = never culled; *sceneLimit = the smallest over
 * the cullable instances -- a crt_render whose camera, or a crt_query_hits whose farthest origin, lies beyond it runs without the
 * cull (same results, every instance entered as upstream does, kernel_main.cl:198); *bounceReach = how far out bounce-ray origins
 * can lie (object-space hit points, hazard H6); *noCullFrames = launches that ran without the cull so far. Any pointer may be NULL. */
int crt_get_cull_range(float* limits, int n, float* sceneLimit, float* bounceReach, uint64_t* noCullFrames);
const char* crt_error_string(int code);
const char* crt_device_name(void);

#ifdef __cplusplus
}
#endif
#endif
