/* crt_debug.h -- diagnostics and test hooks of libcrt_hip.so. NOT part of the drop-in surface (include/crt_api.h): nothing here
 * replaces a call site of the reference's Renderer.cpp / ResourceManager.cpp; the measurement tools under tools/, bench.py and
 * the tests use these entry points. Same library, same conventions (0 = ok). */
#ifndef CRT_DEBUG_H
#define CRT_DEBUG_H
#include "crt_api.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: per wave of the last CRT_RENDER_STAMPS launch, 8 x uint64 {start, end (s_memrealtime, 100 MHz),
 * shader cycles, XCC_ID | HW_ID << 32, wave-level trips of the outer loop | second-inner-step executions << 32,
 * enter-instance steps | wave-level triangle iterations << 32, first-inner-step executions, leaf steps << 32 | lane-level node
 * visits}. Pass dst = NULL to query the wave count. */
int crt_debug_read_stamps(uint64_t* dst, size_t maxWaves, size_t* numWaves);
/* Diagnostic: {start, end} in ms after the start of the first frame, for each of the (up to 256) frames timed since the last
 * crt_frame_time_stats(..., reset = 1), in the order their timing was collected: how a burst of frames in flight fills and drains. */
int crt_debug_read_frame_times(double* dst, size_t maxFrames, size_t* numFrames);
/* Diagnostic: the shader clock (GHz) the device holds under whatever load it carries while the call runs: one wave per XCD
 * watches s_memtime against the 100 MHz s_memrealtime for `micros` microseconds on a stream of its own (bench.py calls it
 * beside frames in flight so that cycle-based figures use the measured clock, not the 2.4 GHz nominal one). */
int crt_debug_measure_clock(int micros, double* ghz);
/* Test hook (multi-device sessions): the next crt_render / crt_resize fails on session device `device` as if that
 * device's submission had returned an error, once. Exercises the "a secondary failed" paths of the dispatcher. Refused with
 * CRT_E_UNSUPPORTED unless the process runs with CRT_DEBUG_HOOKS=1 in its environment. */
int crt_debug_inject_failure(int device);
/* Diagnostic: frames of this session that were held back by the start-up stagger of a burst of CRT_RENDER_ASYNC frames (a
 * caller that streams -- the burst before ran longer than the frame-slot count -- has the first frame of slots 1.. of a new burst
 * delayed by slot x latency / slots so that the slots do not run in lockstep; CRT_STAGGER_US=0 turns it off, =n forces n us). */
int crt_debug_staggered_frames(uint64_t* out);

/* Diagnostic: which Trace kernel rendered the most recently submitted frame (of the session's first device), under the name(s)
 * rocprofv3 prints: "crt_trace_kernel<COUNT,STAMP,SHADOW,TLAS,REFRACT>" for the default megakernel, "crt_trace_refill_kernel<..>",
 * "crt_trace_block_kernel<..>", or "crt_primary_kernel<..>+crt_wavefront_scan_kernel+crt_bounce_kernel<..>" for the opt-in forms
 * CRT_KERNEL selects (read by crt_init; an unknown value fails crt_init). A frame the selected form cannot render is refused with
 * CRT_E_UNSUPPORTED, never rendered by another kernel -- this query is how the tests know (tests/test_gpu_variants.py).
 * Writes a NUL-terminated string of at most cap bytes; "" before the first frame. */
int crt_debug_last_kernel(char* dst, size_t cap);
/* Diagnostic (crt_init_devices sessions): what the secondary devices copied into the first device for the most recently submitted
 * frame -- *bytes in total, *bytesPerPixel 16 (float4 bands) or 4: a CRT_RENDER_UNORM8 frame without FXAA is gathered as the bytes of
 * upstream's RGBA8 render target (Renderer.cpp:63,192), which every device's Trace epilogue stores beside the float pixel; the
 * float frame is rebuilt from them (x = byte / 255, bit for bit) only when crt_read_output / crt_output_device_ptr ask for it.
 * CRT_GATHER_RGBA8=0 in the environment keeps the float4 gather. 0 / 0 in a one-device session. */
int crt_debug_last_gather(uint64_t* bytes, int* bytesPerPixel);
/* Diagnostic: levels and kernel launches of the most recent crt_build_bvh on the session's first device (its own launches, before the
 * re-layout for rendering): the build is launch-bound, bench.py reports both in `bvh_build`. */
int crt_debug_build_stats(uint32_t* levels, uint32_t* launches);
/* Diagnostic: the host-built instance tree (the sphere tree scenes with more than 64 instances find their candidates in). An instance upload refits
 * it -- same partition, node spheres recomputed bottom-up, bit for bit what a rebuild with that partition gives -- while the set of cullable
 * instances is the one it was built for; a median-split build follows when the set changed, when the inner radii have grown by a quarter, or
 * after 256 refits. *builds = builds of this session so far, *refitsSinceBuild, *nodes = nodes of the current tree. */
int crt_debug_tlas_stats(uint64_t* builds, uint32_t* refitsSinceBuild, uint32_t* nodes);

#ifdef __cplusplus
}
#endif
#endif
