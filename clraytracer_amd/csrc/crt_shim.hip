// crt_shim.hip -- kernels and the C-ABI (include/crt_api.h) of the MI355X ray-trace path.
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
#include "crt_device.h"

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

__device__ __forceinline__ void flush_counters(const LaneCounters& lc, unsigned long long* g)
{
    // every lane of the wave must call this (inactive pixels contribute zeros)
    uint32_t s[11] = { lc.rays, lc.primary, lc.secondary, lc.hits, lc.misses, lc.traversals, lc.pops,
                       lc.innerVisits, lc.triTests, lc.capHits, lc.stackOverflows };
    uint32_t mx = wave_max(lc.maxStack);
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        uint32_t t = wave_sum(s[k]);
        if ((threadIdx.x & 63) == 0 && t) atomicAdd(&g[k], (unsigned long long)t);
    }
    if ((threadIdx.x & 63) == 0) atomicMax(&g[11], (unsigned long long)mx);
}

__device__ __forceinline__ void zero_counters(LaneCounters& lc)
{
    lc.rays = lc.primary = lc.secondary = lc.hits = lc.misses = 0;
    lc.traversals = lc.pops = lc.innerVisits = lc.triTests = lc.capHits = lc.stackOverflows = lc.maxStack = 0;
}

#include "crt_persistent.h"
#include "crt_ldstile.h"

// Pixel of this lane. One wave64 per workgroup owns an 8x8 pixel tile, lanes in Morton order
// (coherent ray packets). Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD
// b % 8), so block b takes tile row (b/8 / tilesX) * 8 + b % 8: every XCD (own 4 MiB L2) walks whole
// tile rows left to right -- neighbouring tiles share BVH subtrees in its L2 -- while the eight XCDs
// interleave row by row, which keeps them equally loaded when geometry is concentrated in one part
// of the frame (a contiguous slab per XCD left most XCDs idle: ~1.2 resident waves/SIMD measured).
// `slotOut` receives this workgroup's index into the per-tile cost array (or -1).
__device__ __forceinline__ bool lane_pixel(const CrtFrame& F, int& px, int& py, int* slotOut = nullptr)
{
    const int b = blockIdx.x;
    const int xcd = b & 7;
    int slot = b >> 3;
    int quadrant = -1;
    // Feedback scheduling (crt_order_kernel): each XCD's tiles are launched heaviest-first, by the cycles the same
    // tile cost in the previous frame, and the very heaviest are traced by four waves of one 4x4 quadrant each, so
    // that no single wave's serial chain outlasts the rest of the frame. Nothing is cached or skipped -- only the
    // launch order and the wave shape change. (Before: 0.45 ms with the machine full + 0.45 ms of tail.)
    if (F.order) {
        if ((uint32_t)slot >= F.listLen[xcd]) { if (slotOut) *slotOut = -1; return false; }
        const uint32_t e = F.order[xcd * F.listCap + slot];
        slot = (int)(e & 0x0FFFFFFFu);
        if (e & 0x80000000u) quadrant = (int)((e >> 28) & 3u);
    } else if (slot >= F.slotsPerXcd) { if (slotOut) *slotOut = -1; return false; }
    if (slotOut) *slotOut = xcd * F.slotsPerXcd + slot;
    const int round = slot / F.tilesX;
    const int tx = slot - round * F.tilesX;
    const int k = round * 8 + xcd;                     // index among the tile rows this rank owns
    if (k >= F.ownedTileRows) return false;
    const int bandK = k / F.tileRowsPerBand;
    const int tileRow = (F.rank + bandK * F.nRanks) * F.tileRowsPerBand + (k - bandK * F.tileRowsPerBand);
    const int lane = threadIdx.x & 63;
    if (quadrant >= 0 && (lane >> 4) != quadrant) return false;   // Morton order: lanes 16q..16q+15 are one 4x4 quadrant
    const int lx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4);
    const int ly = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    px = tx * CRT_TILE + lx;
    py = tileRow * CRT_TILE + ly;
    return px < F.width && py < F.height;
}

// Builds the next frame's launch lists: one workgroup per XCD, counting sort of that XCD's tiles by this frame's
// cost, descending (1024 linear bins up to the list's maximum). The tiles that cost at least half the maximum
// (at most CRT_MAX_SPLIT) are emitted as four quadrant entries each and come first.
__global__ __launch_bounds__(1024) void crt_order_kernel(uint32_t* __restrict__ cost, uint32_t* __restrict__ order,
                                                       uint32_t* __restrict__ listLen, int slotsPerXcd, int listCap)
{
    __shared__ uint32_t s_bins[1024];
    __shared__ uint32_t s_max, s_nSplit;
    const int x = blockIdx.x, tid = threadIdx.x;
    const uint32_t* c = cost + (size_t)x * slotsPerXcd;
    uint32_t* o = order + (size_t)x * listCap;
    s_bins[tid] = 0;
    if (tid == 0) s_max = 1;
    __syncthreads();
    uint32_t m = 0;
    for (int i = tid; i < slotsPerXcd; i += 1024) m = c[i] > m ? c[i] : m;
    atomicMax(&s_max, m);
    __syncthreads();
    const float scale = 1023.0f / (float)s_max;
    for (int i = tid; i < slotsPerXcd; i += 1024) {
        int bin = 1023 - (int)((float)c[i] * scale);     // heaviest -> bin 0
        bin = bin < 0 ? 0 : (bin > 1023 ? 1023 : bin);
        atomicAdd(&s_bins[bin], 1u);
    }
    __syncthreads();
    // exclusive scan of the 1024 bins: one bin per thread, wave scans + a scan of the 16 wave totals
    {
        const uint32_t n = s_bins[tid];
        uint32_t incl = n;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64); if ((tid & 63) >= off) incl += v; }
        __shared__ uint32_t s_wave[16];
        if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (int w = 0; w < (tid >> 6); ++w) wbase += s_wave[w];
        s_bins[tid] = wbase + incl - n;
        if (tid == 511) {
            // bins 0..511 hold cost > max/2; a frame of near-equal tiles (nothing stands out) splits nothing
            const uint32_t heavy = wbase + incl;
            s_nSplit = (heavy * 8u > (uint32_t)slotsPerXcd) ? 0u : (heavy < (uint32_t)CRT_MAX_SPLIT ? heavy : (uint32_t)CRT_MAX_SPLIT);
            listLen[x] = (uint32_t)slotsPerXcd + 3u * s_nSplit;
        }
    }
    __syncthreads();
    const uint32_t nSplit = s_nSplit;
    for (int i = tid; i < slotsPerXcd; i += 1024) {
        int bin = 1023 - (int)((float)c[i] * scale);
        bin = bin < 0 ? 0 : (bin > 1023 ? 1023 : bin);
        const uint32_t pos = atomicAdd(&s_bins[bin], 1u);
        if (pos < nSplit) { for (uint32_t q = 0; q < 4; ++q) o[4 * pos + q] = (uint32_t)i | (q << 28) | 0x80000000u; }
        else o[3 * nSplit + pos] = (uint32_t)i;
    }
    __syncthreads();
    for (int i = tid; i < slotsPerXcd; i += 1024) cost[(size_t)x * slotsPerXcd + i] = 0;    // this frame's waves add their cycles
}

__global__ void crt_identity_order_kernel(uint32_t* __restrict__ order, uint32_t* __restrict__ listLen, int slotsPerXcd, int listCap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 8) listLen[i] = (uint32_t)slotsPerXcd;
    if (i < 8 * slotsPerXcd) order[(i / slotsPerXcd) * listCap + (i % slotsPerXcd)] = (uint32_t)(i % slotsPerXcd);
}

// kernel Trace (kernel_main.cl:164-275) with RayGen (kernel_main.cl:277-287) fused: the ray
// direction is computed with the same arithmetic RayGen stores, so the 24.9 MB ray buffer
// round-trip disappears. One thread per pixel, both bounces.
// STAMP (diagnostic build only, CRT_RENDER_STAMPS): every wave records start/end s_memrealtime (100 MHz),
// its s_memtime cycle count and XCC/HW ids into a buffer nothing else reads.
template <bool COUNT, bool STAMP = false>
__global__ __launch_bounds__(CRT_BLOCK, CRT_WAVES_PER_SIMD) void crt_trace_kernel(CrtDevScene S, CrtFrame F, float4* __restrict__ out,
                                                              unsigned long long* __restrict__ counters)
{
    __shared__ uint32_t s_stack[CRT_STACK_DEPTH * CRT_BLOCK];
    crt_lds_u32_ptr stack = (crt_lds_u32_ptr)s_stack + threadIdx.x;
    LaneCounters lc; zero_counters(lc);
    unsigned long long t0rt = 0, t0c = 0;
    if (STAMP) { t0rt = __builtin_amdgcn_s_memrealtime(); t0c = __builtin_amdgcn_s_memtime(); }
    int px, py, costSlot = -1;
    const unsigned long long tc0 = F.cost ? __builtin_amdgcn_s_memtime() : 0ull;
    const bool active = lane_pixel(F, px, py, &costSlot);
    if (active) {
        PathState ps;
        ps.o = mk3(F.camPos[0], F.camPos[1], F.camPos[2]);
        ps.d = raygen_dir(F, px, py);
        ps.result = mk3(0.0f, 0.0f, 0.0f);
        ps.energy = 1.0f;
        for (int bounce = 0; bounce < 2; ++bounce) {
            if (COUNT) { lc.rays++; if (bounce == 0) lc.primary++; else lc.secondary++; }
            Closest c = closest_hit<COUNT, STAMP>(S, ps.o, ps.d, stack, lc);
            bool cont = shade_bounce(S, c, ps, bounce, F.lightY, F.lightZ);
            if (COUNT) { if (cont) lc.hits++; else lc.misses++; }
            if (!cont) break;
        }
        out[(size_t)py * (size_t)F.width + (size_t)px] = make_float4(ps.result.x, ps.result.y, ps.result.z, 1.0f);
    }
    if (F.cost && costSlot >= 0) {      // per-tile cost of this frame (wave-uniform value, one store)
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - tc0;
        if ((threadIdx.x & 63) == 0) atomicAdd(&F.cost[costSlot], dt > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)dt);   // the four waves of a split tile add up
    }
    if (COUNT) flush_counters(lc, counters);
    if (STAMP) {
        const unsigned long long t1c = __builtin_amdgcn_s_memtime(), t1rt = __builtin_amdgcn_s_memrealtime();
        const uint32_t wOuter = wave_sum(lc.pops), wEnter = wave_sum(lc.traversals), wDescent = wave_sum(lc.innerVisits),
                       wLeaf = wave_sum(lc.triTests), laneVisits = wave_sum(lc.rays);
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* st = counters + 16 + (size_t)blockIdx.x * 8;
            st[4] = wOuter; st[5] = wEnter; st[6] = wDescent; st[7] = ((unsigned long long)wLeaf << 32) | laneVisits;
            st[0] = t0rt; st[1] = t1rt; st[2] = t1c - t0c;
            st[3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
        }
    }
}

// ---- wavefront form of Trace: one launch per bounce with ballot compaction in between ----------------
// The megakernel above runs bounce 1 inside the same wave as bounce 0, at the lane density of the pixels
// that hit something (31 % on multi-1M) and on top of the wave's bounce-0 latency. Here bounce 0 writes the
// pixel's partial result and appends {origin, direction, energy, pixel} of every continuing path to a queue:
// the lanes of a wave that continue are found with one ballot, the wave reserves a contiguous queue range with
// ONE atomic, and every lane stores its 32-byte record at base + (rank among the continuing lanes). Bounce 1
// is then traced by dense 64-ray packets. Per-path arithmetic is unchanged: result = (partial) + (bounce-1
// terms) in the same order as kernel_main.cl:267, so pixels are bit-identical to the megakernel.
struct CrtBounceRay { float ox, oy, oz, energy, dx, dy, dz; uint32_t pixel; };   // 32 B

template <bool COUNT>
__global__ __launch_bounds__(CRT_BLOCK, CRT_WAVES_PER_SIMD) void crt_primary_kernel(CrtDevScene S, CrtFrame F, float4* __restrict__ out,
                                                                unsigned long long* __restrict__ counters,
                                                                CrtBounceRay* __restrict__ queue, uint32_t* __restrict__ queueCount)
{
    __shared__ uint32_t s_stack[CRT_STACK_DEPTH * CRT_BLOCK];
    crt_lds_u32_ptr stack = (crt_lds_u32_ptr)s_stack + threadIdx.x;
    LaneCounters lc; zero_counters(lc);
    int px, py;
    const bool active = lane_pixel(F, px, py);
    bool cont = false;
    PathState ps;
    ps.o = mk3(0.f, 0.f, 0.f); ps.d = ps.o; ps.result = ps.o; ps.energy = 1.0f;
    if (active) {
        ps.o = mk3(F.camPos[0], F.camPos[1], F.camPos[2]);
        ps.d = raygen_dir(F, px, py);
        if (COUNT) { lc.rays++; lc.primary++; }
        Closest c = closest_hit<COUNT>(S, ps.o, ps.d, stack, lc);
        cont = shade_bounce(S, c, ps, 0, F.lightY, F.lightZ);
        if (COUNT) { if (cont) lc.hits++; else lc.misses++; }
        out[(size_t)py * (size_t)F.width + (size_t)px] = make_float4(ps.result.x, ps.result.y, ps.result.z, 1.0f);
    }
    // wave-level compaction of the continuing paths
    const unsigned long long m = __ballot(cont);
    if (m != 0) {
        uint32_t base = 0;
        if ((threadIdx.x & 63) == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(queueCount, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1, 64);
        if (cont) {
            const uint32_t rank = (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
            CrtBounceRay r;
            r.ox = ps.o.x; r.oy = ps.o.y; r.oz = ps.o.z; r.energy = ps.energy;
            r.dx = ps.d.x; r.dy = ps.d.y; r.dz = ps.d.z; r.pixel = (uint32_t)py * (uint32_t)F.width + (uint32_t)px;
            queue[base + rank] = r;
        }
    }
    if (COUNT) flush_counters(lc, counters);
}

template <bool COUNT>
__global__ __launch_bounds__(CRT_BLOCK, CRT_WAVES_PER_SIMD) void crt_bounce_kernel(CrtDevScene S, CrtFrame F, float4* __restrict__ out,
                                                               unsigned long long* __restrict__ counters,
                                                               const CrtBounceRay* __restrict__ queue, const uint32_t* __restrict__ queueCount)
{
    __shared__ uint32_t s_stack[CRT_STACK_DEPTH * CRT_BLOCK];
    crt_lds_u32_ptr stack = (crt_lds_u32_ptr)s_stack + threadIdx.x;
    LaneCounters lc; zero_counters(lc);
    const uint32_t n = *queueCount;
    const uint32_t k = blockIdx.x * CRT_BLOCK + threadIdx.x;
    if (k < n) {
        const CrtBounceRay r = queue[k];
        PathState ps;
        ps.o = mk3(r.ox, r.oy, r.oz); ps.d = mk3(r.dx, r.dy, r.dz); ps.energy = r.energy;
        const float4 partial = out[r.pixel];
        ps.result = mk3(partial.x, partial.y, partial.z);
        if (COUNT) { lc.rays++; lc.secondary++; }
        Closest c = closest_hit<COUNT>(S, ps.o, ps.d, stack, lc);
        const bool cont = shade_bounce(S, c, ps, 1, F.lightY, F.lightZ);
        if (COUNT) { if (cont) lc.hits++; else lc.misses++; }
        out[r.pixel] = make_float4(ps.result.x, ps.result.y, ps.result.z, 1.0f);
    }
    if (COUNT) flush_counters(lc, counters);
}

// kernel RayGen as its own launch (only for CRT_RENDER_WRITE_RAYS)
__global__ __launch_bounds__(CRT_BLOCK) void crt_raygen_kernel(CrtFrame F, float* __restrict__ rays)
{
    int px, py;
    if (!lane_pixel(F, px, py)) return;
    v3 d = raygen_dir(F, px, py);
    float* o = rays + 3 * ((size_t)px + (size_t)py * (size_t)F.width);
    o[0] = d.x; o[1] = d.y; o[2] = d.z;
}

// kernel PostProcess (kernel_main.cl:342-359, MathAndSTL.cl:132-169) on the float frame
__global__ __launch_bounds__(CRT_BLOCK) void crt_postprocess_kernel(CrtFrame F, float4* __restrict__ img)
{
    int px, py;
    if (!lane_pixel(F, px, py)) return;
    const size_t idx = (size_t)py * (size_t)F.width + (size_t)px;
    const float uvx = (float)px / (float)F.width, uvy = (float)py / (float)F.height;
    float4 p = img[idx];
    v3 rgb = mk3(p.x, p.y, p.z);
    const float P = sqrtf((rgb.x * rgb.x) * 0.299f + ((rgb.y * rgb.y) * 0.587f) + ((rgb.z * rgb.z) * 0.114f));
    const v3 Pv = mk3(P, P, P);
    rgb = add3(Pv, scale3(sub3(rgb, Pv), 1.2f));
    const v3 lw = mk3(0.2126f, 0.7152f, 0.0722f);
    const float max_white_l = 0.8f;
    const float l_old = dot3(rgb, lw);
    const float numerator = l_old * (1.0f + (l_old / (max_white_l * max_white_l)));
    const float l_new = numerator / (1.0f + l_old);
    const float l_in = dot3(rgb, lw);
    rgb = scale3(rgb, l_new / l_in);
    const float ig = 1.0f / 1.55f;
    rgb = mk3(powf(rgb.x, ig), powf(rgb.y, ig), powf(rgb.z, ig));
    const float oneDivGamma = 1.0f / 1.2f;
    rgb = mk3(powf(rgb.x, oneDivGamma), powf(rgb.y, oneDivGamma), powf(rgb.z, oneDivGamma));
    const float vx = uvx * (1.0f - uvy), vy = uvy * (1.0f - uvx);
    float vig = (vx * vy) * 15.0f;
    vig = powf(vig, 0.15f);
    rgb = scale3(rgb, vig);
    img[idx] = make_float4(rgb.x, rgb.y, rgb.z, 1.0f);
}

// closest-hit query over explicit rays (hit-record parity)
__global__ __launch_bounds__(CRT_BLOCK, CRT_WAVES_PER_SIMD) void crt_query_kernel(CrtDevScene S, const float* __restrict__ origins,
                                                              const float* __restrict__ dirs, int n,
                                                              CrtRayHit* __restrict__ out, unsigned long long* __restrict__ counters)
{
    __shared__ uint32_t s_stack[CRT_STACK_DEPTH * CRT_BLOCK];
    crt_lds_u32_ptr stack = (crt_lds_u32_ptr)s_stack + threadIdx.x;
    LaneCounters lc; zero_counters(lc);
    const int k = blockIdx.x * CRT_BLOCK + threadIdx.x;
    if (k < n) {
        v3 o = mk3(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]);
        v3 d = mk3(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]);
        lc.rays++;
        Closest c = closest_hit<true>(S, o, d, stack, lc);
        CrtRayHit h;
        if (c.anyHit) { h.t = c.hit.t; h.u = c.hit.u; h.v = c.hit.v; h.triIndex = c.hit.tri; h.instance = c.hitInstance; lc.hits++; }
        else { h.t = c.distance; h.u = 0.0f; h.v = 0.0f; h.triIndex = 0; h.instance = -1; lc.misses++; }
        out[k] = h;
    }
    flush_counters(lc, counters);
}

// ---- relayout kernels (reference layout -> CDNA4 layout) ----
__global__ void crt_relayout_tris(const CrtTri* __restrict__ raw, size_t first, size_t count,
                                  float* __restrict__ hot, uint4* __restrict__ cold)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const size_t i = first + k;
    const CrtTri t = raw[i];
    float* h = hot + i * 9;
    h[0] = t.v0[0]; h[1] = t.v0[1]; h[2] = t.v0[2];
    h[3] = t.v1[0] - t.v0[0]; h[4] = t.v1[1] - t.v0[1]; h[5] = t.v1[2] - t.v0[2];   // edge1 = y - x
    h[6] = t.v2[0] - t.v0[0]; h[7] = t.v2[1] - t.v0[1]; h[8] = t.v2[2] - t.v0[2];   // edge2 = z - x
    const uint4* tail = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(raw + i) + 48);
    cold[i * 2] = tail[0];
    cold[i * 2 + 1] = tail[1];
}

// ---- hot tiles: the top levels of every mesh's tree get the pair indices [0, CRT_HOT_PAIRS) ----
// `hotSlot[leftFirst >> 1]` = hot index of the sibling pair starting at node leftFirst, or CRT_NOT_HOT. Mesh m owns the
// slots [m * perMesh, (m + 1) * perMesh) in heap order (root pair 0; the pair under the left/right child of pair h is
// 2h+1 / 2h+2), perMesh = the largest power of two <= CRT_HOT_PAIRS / numRoots. Every pair also keeps its ordinary record at
// CRT_HOT_PAIRS + (leftFirst >> 1); child references point at the hot copy when there is one, so a kernel that stages
// pairs[0 .. CRT_HOT_PAIRS) in LDS serves the most visited nodes from there, and every other kernel just sees indices.
#define CRT_NOT_HOT 0xFFFFFFFFu

__global__ void crt_assign_hot_slots(const CrtBVHNode* __restrict__ raw, uint32_t nodeCount, const uint32_t* __restrict__ roots,
                                     uint32_t numRoots, uint32_t perMesh, uint32_t* __restrict__ hotSlot)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= numRoots || perMesh < 2) return;
    const uint32_t root = roots[m];
    if (root >= nodeCount) return;
    // breadth-first over heap indices 0 .. perMesh-2; node index of the PARENT node of pair h kept in a small local queue
    uint32_t parentOf[CRT_HOT_PAIRS];            // heap index -> node whose children form that pair (CRT_NOT_HOT = absent)
    for (uint32_t h = 0; h + 1 < perMesh; ++h) parentOf[h] = CRT_NOT_HOT;
    parentOf[0] = root;
    for (uint32_t h = 0; h + 1 < perMesh; ++h) {
        const uint32_t n = parentOf[h];
        if (n == CRT_NOT_HOT) continue;
        const CrtBVHNode node = raw[n];
        if (node.triCount > 0) continue;                                   // leaf: no pair below it
        const uint32_t l = node.leftFirst;
        if (l <= n || (uint64_t)l + 1 >= (uint64_t)nodeCount) continue;    // invalid link: flagged by crt_relayout_nodes
        hotSlot[l >> 1] = m * perMesh + h;
        if (2 * h + 1 < perMesh - 1) parentOf[2 * h + 1] = l;
        if (2 * h + 2 < perMesh - 1) parentOf[2 * h + 2] = l + 1;
    }
}

__device__ __forceinline__ uint32_t make_ref(const CrtBVHNode& n, uint32_t self, uint32_t nodeCount, uint32_t triCap,
                                             uint32_t* bigLeaf, const uint32_t* hotSlot, int* err)
{
    if (n.triCount > 0) {
        if ((uint64_t)n.leftFirst + (uint64_t)n.triCount > (uint64_t)triCap || n.leftFirst > 0x00FFFFFFu) { atomicOr(err, 1); return CRT_LEAF_BIT | (1u << 24); }
        if (n.triCount < 128u) return CRT_LEAF_BIT | (n.triCount << 24) | n.leftFirst;
        bigLeaf[n.leftFirst] = n.triCount;
        return CRT_LEAF_BIT | n.leftFirst;
    }
    // children are always allocated after their parent (BVH.cpp:203-204): enforces an acyclic graph
    if (n.leftFirst <= self || (uint64_t)n.leftFirst + 1 >= (uint64_t)nodeCount) { atomicOr(err, 2); return CRT_LEAF_BIT | (1u << 24); }
    const uint32_t hot = hotSlot[n.leftFirst >> 1];
    return hot != CRT_NOT_HOT ? hot : (uint32_t)CRT_HOT_PAIRS + (n.leftFirst >> 1);
}

__global__ void crt_relayout_nodes(const CrtBVHNode* __restrict__ raw, uint32_t nodeCount, uint32_t triCap,
                                   float4* __restrict__ pairs, uint32_t* __restrict__ bigLeaf, const uint32_t* __restrict__ hotSlot,
                                   int* __restrict__ err)
{
    uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nodeCount) return;
    const CrtBVHNode node = raw[n];
    if (node.triCount > 0) return;
    const uint32_t l = node.leftFirst;
    if (l <= n || (uint64_t)l + 1 >= (uint64_t)nodeCount) { atomicOr(err, 2); return; }
    const CrtBVHNode L = raw[l], R = raw[l + 1];
    const uint32_t lref = make_ref(L, l, nodeCount, triCap, bigLeaf, hotSlot, err);
    const uint32_t rref = make_ref(R, l + 1, nodeCount, triCap, bigLeaf, hotSlot, err);
    const float4 r0 = make_float4(L.aabbMin[0], L.aabbMin[1], L.aabbMin[2], __uint_as_float(lref));
    const float4 r1 = make_float4(L.aabbMax[0], L.aabbMax[1], L.aabbMax[2], 0.0f);
    const float4 r2 = make_float4(R.aabbMin[0], R.aabbMin[1], R.aabbMin[2], __uint_as_float(rref));
    const float4 r3 = make_float4(R.aabbMax[0], R.aabbMax[1], R.aabbMax[2], 0.0f);
    float4* p = pairs + ((size_t)CRT_HOT_PAIRS + (l >> 1)) * 4;
    p[0] = r0; p[1] = r1; p[2] = r2; p[3] = r3;
    const uint32_t hot = hotSlot[l >> 1];
    if (hot != CRT_NOT_HOT) { float4* q = pairs + (size_t)hot * 4; q[0] = r0; q[1] = r1; q[2] = r2; q[3] = r3; }
}

__global__ void crt_make_root_refs(const CrtBVHNode* __restrict__ raw, uint32_t nodeCount, uint32_t triCap,
                                   const uint32_t* __restrict__ roots, uint32_t numRoots,
                                   uint32_t* __restrict__ rootRefs, uint32_t* __restrict__ bigLeaf, const uint32_t* __restrict__ hotSlot,
                                   int* __restrict__ err)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= numRoots) return;
    const uint32_t r = roots[k];
    if (r >= nodeCount) { rootRefs[k] = CRT_LEAF_BIT | (1u << 24); return; } // not (yet) uploaded: harmless dummy, flagged at render
    rootRefs[k] = make_ref(raw[r], r, nodeCount, triCap, bigLeaf, hotSlot, err);
}

__global__ void crt_relayout_instances(const CrtMeshInstance* __restrict__ raw, const uint32_t* __restrict__ rootRefs,
                                       uint32_t count, CrtDevInstance* __restrict__ out)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const CrtMeshInstance m = raw[i];
    const uint32_t mesh = m.meshIndex < CRT_MAX_MESHES ? m.meshIndex : 0;
    CrtDevInstance d;
    d.r0 = make_float4(m.inverseTransform.m[0][0], m.inverseTransform.m[0][1], m.inverseTransform.m[0][2], __uint_as_float(rootRefs[mesh]));
    d.r1 = make_float4(m.inverseTransform.m[1][0], m.inverseTransform.m[1][1], m.inverseTransform.m[1][2], __uint_as_float((uint32_t)m.materialStart));
    d.r2 = make_float4(m.inverseTransform.m[2][0], m.inverseTransform.m[2][1], m.inverseTransform.m[2][2], 0.0f);
    d.r3 = make_float4(m.inverseTransform.m[3][0], m.inverseTransform.m[3][1], m.inverseTransform.m[3][2], 0.0f);
    out[i] = d;
}

__global__ void crt_relayout_texels(const uint8_t* __restrict__ raw, size_t firstTexel, size_t count, uint32_t* __restrict__ texels)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const size_t i = firstTexel + k;
    const uint8_t* p = raw + i * 3;
    texels[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

// ------------------------------------------------------------------------------------------------
// host state
// ------------------------------------------------------------------------------------------------
namespace {

struct State {
    bool initialized = false;
    int device = -1;
    char deviceName[256] = { 0 };
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    int width = 0, height = 0;
    int bandRows = 16, rank = 0, nRanks = 1;
    // raw (reference-layout) device copies
    CrtTri* rawTris = nullptr; CrtBVHNode* rawNodes = nullptr; uint32_t* roots = nullptr; uint8_t* rawTexels = nullptr;
    // CDNA4 layouts
    float4* pairs = nullptr; float* triHot = nullptr; uint4* triCold = nullptr; uint32_t* bigLeaf = nullptr;
    uint32_t* rootRefs = nullptr; uint32_t* texels = nullptr;
    CrtMeshInstance* instances = nullptr; CrtMaterial* materials = nullptr; CrtTexture* textures = nullptr;
    float4* instBounds = nullptr; CrtDevInstance* devInstances = nullptr; uint32_t* hotSlot = nullptr;
    CrtMeshInstance hInstances[CRT_MAX_INSTANCES]; uint32_t hRoots[CRT_MAX_MESHES]; uint32_t instHigh = 0;
    float* rays = nullptr; float4* out = nullptr;
    unsigned long long* counters = nullptr; int* err = nullptr;
    unsigned long long* stamps = nullptr; size_t stampBytes = 0, stampWaves = 0;
    CrtQueues* queues = nullptr; int numCUs = 0; int persistent = 0; int wavesPerCU = 16;
    uint32_t* tileOrder[2] = { nullptr, nullptr }; uint32_t* tileLen[2] = { nullptr, nullptr }; uint32_t* tileCost = nullptr; size_t orderCap = 0; int orderSlots = -1; int orderKey[6] = { 0, 0, 0, 0, 0, 0 }; int feedback = 1;
    int ldsTiles = 0; uint32_t* listNext = nullptr;
    int wavefront = 0; CrtBounceRay* bounceQueue = nullptr; uint32_t* bounceCount = nullptr; size_t bounceCap = 0;
    void* queryBuf = nullptr; size_t queryBytes = 0;
    size_t triCap = 0, nodeCap = 0, texelByteCap = 0;
    uint32_t nodeCount = 0, numRoots = 0; size_t texelBytesHigh = 0; size_t trisHigh = 0;
    bool sceneValid = true;
    float ms[4] = { 0, 0, 0, 0 };
    bool timed[4] = { false, false, false, false };
    bool pendingTiming = false; int pendingFlags = 0; bool evRaygen = false, evPost = false;
    CrtCounters lastCounters;
};
State g;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { return (int)e_; } } while (0)

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

void fill_scene(CrtDevScene& S, uint32_t numInstances)
{
    S.pairs = g.pairs; S.triHot = g.triHot; S.triCold = g.triCold; S.bigLeaf = g.bigLeaf; S.rootRefs = g.rootRefs;
    S.instances = g.instances; S.devInstances = g.devInstances; S.instBounds = g.instBounds; S.materials = g.materials; S.textures = g.textures; S.texels = g.texels;
    S.numTexels = (int)((g.texelBytesHigh + 2) / 3);
    if (S.numTexels < 1) S.numTexels = 1;
    S.numInstances = numInstances;
}

int alloc_frame_buffers(int w, int h)
{
    if (g.rays) { (void)hipFree(g.rays); g.rays = nullptr; }
    if (g.out) { (void)hipFree(g.out); g.out = nullptr; }
    if (g.bounceQueue) { (void)hipFree(g.bounceQueue); g.bounceQueue = nullptr; }
    g.bounceCap = (size_t)w * (size_t)h;
    HIPCHK(hipMalloc(&g.bounceQueue, sizeof(CrtBounceRay) * g.bounceCap));
    HIPCHK(hipMalloc(&g.rays, sizeof(float) * 3 * (size_t)w * (size_t)h));
    HIPCHK(hipMalloc(&g.out, sizeof(float4) * (size_t)w * (size_t)h));
    HIPCHK(hipMemsetAsync(g.out, 0, sizeof(float4) * (size_t)w * (size_t)h, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    g.width = w; g.height = h;
    return CRT_OK;
}

int rebuild_instance_bounds();

int rebuild_bvh_layout()
{
    HIPCHK(hipMemsetAsync(g.err, 0, sizeof(int), g.stream));
    if (g.nodeCount) {
        HIPCHK(hipMemsetAsync(g.hotSlot, 0xFF, ((size_t)g.nodeCount / 2 + 1) * sizeof(uint32_t), g.stream));
        if (g.numRoots) {
            uint32_t perMesh = 1;
            while (perMesh * 2 * g.numRoots <= (uint32_t)CRT_HOT_PAIRS) perMesh *= 2;
            if (perMesh * g.numRoots > (uint32_t)CRT_HOT_PAIRS) perMesh = 0;
            crt_assign_hot_slots<<<(g.numRoots + 63) / 64, 64, 0, g.stream>>>(g.rawNodes, g.nodeCount, g.roots, g.numRoots, perMesh, g.hotSlot);
            HIPCHK(hipGetLastError());
        }
        crt_relayout_nodes<<<(g.nodeCount + 255) / 256, 256, 0, g.stream>>>(g.rawNodes, g.nodeCount, (uint32_t)g.triCap, g.pairs, g.bigLeaf, g.hotSlot, g.err);
        HIPCHK(hipGetLastError());
    }
    if (g.numRoots) {
        crt_make_root_refs<<<(g.numRoots + 255) / 256, 256, 0, g.stream>>>(g.rawNodes, g.nodeCount, (uint32_t)g.triCap, g.roots, g.numRoots, g.rootRefs, g.bigLeaf, g.hotSlot, g.err);
        HIPCHK(hipGetLastError());
    }
    int err = 0;
    HIPCHK(hipMemcpyAsync(&err, g.err, sizeof(int), hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    g.sceneValid = (err == 0);
    if (err) return CRT_E_BAD_ARGUMENT;
    return rebuild_instance_bounds();
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

int rebuild_instance_bounds()
{
    static float4 bounds[CRT_MAX_INSTANCES];
    static CrtBVHNode rootNodes[CRT_MAX_MESHES];
    static bool haveRoot[CRT_MAX_MESHES];
    for (uint32_t m = 0; m < CRT_MAX_MESHES; ++m) {
        haveRoot[m] = m < g.numRoots && g.hRoots[m] < g.nodeCount;
        if (haveRoot[m]) HIPCHK(hipMemcpyAsync(&rootNodes[m], g.rawNodes + g.hRoots[m], sizeof(CrtBVHNode), hipMemcpyDeviceToHost, g.stream));
    }
    HIPCHK(hipStreamSynchronize(g.stream));
    for (uint32_t i = 0; i < CRT_MAX_INSTANCES; ++i) {
        bounds[i] = make_float4(0.f, 0.f, 0.f, -1.0f);
        if (i >= g.instHigh) continue;
        const CrtMeshInstance& inst = g.hInstances[i];
        if (inst.meshIndex >= CRT_MAX_MESHES || !haveRoot[inst.meshIndex]) continue;
        const CrtBVHNode& root = rootNodes[inst.meshIndex];
        if (root.triCount > 0) continue;      // single-leaf mesh: its triangles are tested without any box test (hazard H3)
        double inv[16], fwd[16];
        for (int k = 0; k < 16; ++k) inv[k] = (double)(&inst.inverseTransform.m[0][0])[k];
        if (!invert4(inv, fwd)) continue;
        auto xform = [&](double x, double y, double z, double* o) {
            for (int c = 0; c < 3; ++c) o[c] = x * fwd[0 + c] + y * fwd[4 + c] + z * fwd[8 + c] + fwd[12 + c];
        };
        const double lo[3] = { root.aabbMin[0], root.aabbMin[1], root.aabbMin[2] }, hi[3] = { root.aabbMax[0], root.aabbMax[1], root.aabbMax[2] };
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
        const float rf = (float)(r * (1.0 + 1e-4)) ;
        const float4 b = make_float4((float)cw[0], (float)cw[1], (float)cw[2], rf);
        if (!(isfinite(b.x) && isfinite(b.y) && isfinite(b.z) && isfinite(b.w)) || !(b.w < 1e18f)) continue;
        bounds[i] = b;
    }
    HIPCHK(hipMemcpyAsync(g.instBounds, bounds, sizeof bounds, hipMemcpyHostToDevice, g.stream));
    crt_relayout_instances<<<(CRT_MAX_INSTANCES + 255) / 256, 256, 0, g.stream>>>(g.instances, g.rootRefs, CRT_MAX_INSTANCES, g.devInstances);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g.stream));
    return CRT_OK;
}

int collect_timing()
{
    if (!g.pendingTiming) return CRT_OK;
    hipEvent_t traceStart = g.evRaygen ? g.ev[1] : g.ev[0], frameEnd = g.evPost ? g.ev[3] : g.ev[2];
    HIPCHK(hipEventSynchronize(frameEnd));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, g.ev[0], frameEnd)); g.ms[0] = t;
    g.ms[1] = 0.0f; g.ms[3] = 0.0f;
    if (g.evRaygen) { HIPCHK(hipEventElapsedTime(&t, g.ev[0], g.ev[1])); g.ms[1] = t; }
    HIPCHK(hipEventElapsedTime(&t, traceStart, g.ev[2])); g.ms[2] = t;
    if (g.evPost) { HIPCHK(hipEventElapsedTime(&t, g.ev[2], g.ev[3])); g.ms[3] = t; }
    if (g.pendingFlags & CRT_RENDER_COUNTERS) {
        unsigned long long c[12];
        HIPCHK(hipMemcpy(c, g.counters, sizeof c, hipMemcpyDeviceToHost));
        CrtCounters& o = g.lastCounters;
        o.rays = c[0]; o.primary = c[1]; o.secondary = c[2]; o.hits = c[3]; o.misses = c[4]; o.traversals = c[5];
        o.pops = c[6]; o.innerVisits = c[7]; o.triTests = c[8]; o.capHits = c[9]; o.stackOverflows = c[10]; o.maxStack = c[11];
    }
    g.pendingTiming = false;
    return CRT_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
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

const char* crt_device_name(void) { return g.deviceName; }

int crt_init(int device, int width, int height)
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
    HIPCHK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    for (int i = 0; i < 5; ++i) HIPCHK(hipEventCreate(&g.ev[i]));

    g.triCap = (size_t)CRT_MAX_TRIANGLES * 2;           // ResourceManager.cpp:158
    g.nodeCap = (size_t)CRT_MAX_TRIANGLES * 2;          // ResourceManager.cpp:159 (MAX_BVHMEMORY * 2)
    g.texelByteCap = CRT_MAX_TEXTURE_BYTES * 2;         // ResourceManager.cpp:163
    HIPCHK(hipMalloc(&g.rawTris, g.triCap * sizeof(CrtTri)));
    HIPCHK(hipMalloc(&g.rawNodes, g.nodeCap * sizeof(CrtBVHNode)));
    HIPCHK(hipMalloc(&g.roots, CRT_MAX_MESHES * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.rawTexels, g.texelByteCap + 16));
    HIPCHK(hipMalloc(&g.pairs, (g.nodeCap / 2 + 1 + CRT_HOT_PAIRS) * 4 * sizeof(float4)));
    HIPCHK(hipMalloc(&g.hotSlot, (g.nodeCap / 2 + 1) * sizeof(uint32_t)));
    HIPCHK(hipMemset(g.pairs, 0, (size_t)CRT_HOT_PAIRS * 4 * sizeof(float4)));
    HIPCHK(hipMalloc(&g.triHot, g.triCap * 9 * sizeof(float)));
    HIPCHK(hipMalloc(&g.triCold, g.triCap * 2 * sizeof(uint4)));
    HIPCHK(hipMalloc(&g.bigLeaf, g.triCap * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.rootRefs, CRT_MAX_MESHES * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.texels, (g.texelByteCap / 3 + 2) * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.instances, CRT_MAX_INSTANCES * sizeof(CrtMeshInstance)));
    HIPCHK(hipMalloc(&g.instBounds, CRT_MAX_INSTANCES * sizeof(float4)));
    HIPCHK(hipMalloc(&g.devInstances, CRT_MAX_INSTANCES * sizeof(CrtDevInstance)));
    HIPCHK(hipMalloc(&g.materials, CRT_MAX_MATERIALS * sizeof(CrtMaterial)));
    HIPCHK(hipMalloc(&g.textures, CRT_MAX_TEXTURES * sizeof(CrtTexture)));
    HIPCHK(hipMalloc(&g.counters, 12 * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&g.err, sizeof(int)));
    HIPCHK(hipMalloc(&g.queues, sizeof(CrtQueues)));
    HIPCHK(hipMalloc(&g.bounceCount, sizeof(uint32_t)));
    HIPCHK(hipMalloc(&g.listNext, 8 * sizeof(uint32_t)));
    g.numCUs = prop.multiProcessorCount;
    { const char* e = getenv("CRT_KERNEL"); g.persistent = (e && strcmp(e, "persistent") == 0); g.wavefront = (e && strcmp(e, "wavefront") == 0); g.ldsTiles = (e && strcmp(e, "lds") == 0); }  // default: tile kernel (faster)
    { const char* e = getenv("CRT_FEEDBACK"); g.feedback = !(e && atoi(e) == 0); }
    { const char* e = getenv("CRT_WAVES_PER_CU"); g.wavesPerCU = e ? atoi(e) : 16; if (g.wavesPerCU < 1) g.wavesPerCU = 1; }
    HIPCHK(hipMemset(g.roots, 0, CRT_MAX_MESHES * sizeof(uint32_t)));
    HIPCHK(hipMemset(g.rootRefs, 0, CRT_MAX_MESHES * sizeof(uint32_t)));
    HIPCHK(hipMemset(g.instances, 0, CRT_MAX_INSTANCES * sizeof(CrtMeshInstance)));
    HIPCHK(hipMemset(g.materials, 0, CRT_MAX_MATERIALS * sizeof(CrtMaterial)));
    HIPCHK(hipMemset(g.textures, 0, CRT_MAX_TEXTURES * sizeof(CrtTexture)));
    HIPCHK(hipMemset(g.texels, 0, 64));
    g.nodeCount = 0; g.numRoots = 0; g.texelBytesHigh = 0; g.trisHigh = 0; g.sceneValid = true; g.instHigh = 0;
    memset(g.hInstances, 0, sizeof g.hInstances); memset(g.hRoots, 0, sizeof g.hRoots);
    { int rcb = rebuild_instance_bounds(); if (rcb) return rcb; }
    g.bandRows = 16; g.rank = 0; g.nRanks = 1;
    int rc = alloc_frame_buffers(width, height);
    if (rc) return rc;
    g.initialized = true;
    // default white / black texels (ResourceManager.cpp:168-177)
    const unsigned char def[6] = { 0xFF, 0xFF, 0xFF, 0, 0, 0 };
    return crt_upload_texels(def, 0, 6);
}

int crt_shutdown(void)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    (void)hipStreamSynchronize(g.stream);
    void* ptrs[] = { g.rawTris, g.rawNodes, g.roots, g.rawTexels, g.pairs, g.triHot, g.triCold, g.bigLeaf, g.rootRefs,
                     g.texels, g.instances, g.instBounds, g.devInstances, g.hotSlot, g.materials, g.textures, g.rays, g.out, g.counters, g.err, g.queryBuf, g.stamps, g.queues, g.bounceQueue, g.bounceCount, g.listNext, g.tileOrder[0], g.tileOrder[1], g.tileLen[0], g.tileLen[1], g.tileCost };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int i = 0; i < 5; ++i) if (g.ev[i]) (void)hipEventDestroy(g.ev[i]);
    if (g.stream) (void)hipStreamDestroy(g.stream);
    g = State();
    return CRT_OK;
}

int crt_resize(int width, int height)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (width < 16 || height < 16) return CRT_OK; // Renderer.cpp:200
    HIPCHK(hipStreamSynchronize(g.stream));
    return alloc_frame_buffers(width, height);
}

int crt_set_row_bands(int bandRows, int rank, int nRanks)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (bandRows < 16 || bandRows % 16 != 0 || nRanks < 1 || rank < 0 || rank >= nRanks) return CRT_E_BAD_ARGUMENT;
    g.bandRows = bandRows; g.rank = rank; g.nRanks = nRanks;
    return CRT_OK;
}

int crt_row_owner(int row, int bandRows, int nRanks)
{
    if (row < 0 || bandRows < 16 || bandRows % 16 != 0 || nRanks < 1) return CRT_E_BAD_ARGUMENT;
    return (row / bandRows) % nRanks;
}

int crt_owned_rows(void)
{
    if (!g.initialized) return 0;
    int rows = 0;
    const int tpb = g.bandRows / CRT_TILE;
    for (int y = 0; y < g.height; ++y) if ((((y / CRT_TILE) / tpb) % g.nRanks) == g.rank) ++rows;
    return rows;
}

int crt_upload_triangles(const void* tris, size_t byteOffset, size_t bytes)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (bytes == 0) return CRT_OK;
    if (!tris || byteOffset % sizeof(CrtTri) || bytes % sizeof(CrtTri)) return CRT_E_BAD_ARGUMENT;
    if (byteOffset + bytes > g.triCap * sizeof(CrtTri)) return CRT_E_OUT_OF_RANGE;
    HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(g.rawTris) + byteOffset, tris, bytes, hipMemcpyHostToDevice, g.stream));
    const size_t first = byteOffset / sizeof(CrtTri), count = bytes / sizeof(CrtTri);
    crt_relayout_tris<<<(unsigned)((count + 255) / 256), 256, 0, g.stream>>>(g.rawTris, first, count, g.triHot, g.triCold);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g.stream));
    if (first + count > g.trisHigh) g.trisHigh = first + count;
    return CRT_OK;
}

int crt_upload_bvh_nodes(const void* nodes, size_t byteOffset, size_t bytes)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (bytes == 0) return CRT_OK;
    if (!nodes || byteOffset % sizeof(CrtBVHNode) || bytes % sizeof(CrtBVHNode)) return CRT_E_BAD_ARGUMENT;
    if (byteOffset + bytes > g.nodeCap * sizeof(CrtBVHNode)) return CRT_E_OUT_OF_RANGE;
    HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(g.rawNodes) + byteOffset, nodes, bytes, hipMemcpyHostToDevice, g.stream));
    const uint32_t high = (uint32_t)((byteOffset + bytes) / sizeof(CrtBVHNode));
    if (high > g.nodeCount) g.nodeCount = high;
    return rebuild_bvh_layout();
}

int crt_upload_bvh_roots(const uint32_t* roots, size_t firstMesh, size_t count)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (count == 0) return CRT_OK;
    if (!roots) return CRT_E_BAD_ARGUMENT;
    if (firstMesh + count > CRT_MAX_MESHES) return CRT_E_OUT_OF_RANGE;
    HIPCHK(hipMemcpyAsync(g.roots + firstMesh, roots, count * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
    memcpy(g.hRoots + firstMesh, roots, count * sizeof(uint32_t));
    if (firstMesh + count > g.numRoots) g.numRoots = (uint32_t)(firstMesh + count);
    return rebuild_bvh_layout();
}

int crt_upload_materials(const void* materials, size_t first, size_t count)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (count == 0) return CRT_OK;
    if (!materials) return CRT_E_BAD_ARGUMENT;
    if (first + count > CRT_MAX_MATERIALS) return CRT_E_OUT_OF_RANGE;
    HIPCHK(hipMemcpyAsync(g.materials + first, materials, count * sizeof(CrtMaterial), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return CRT_OK;
}

int crt_upload_texture_table(const void* textures, size_t count)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (count == 0) return CRT_OK;
    if (!textures) return CRT_E_BAD_ARGUMENT;
    if (count > CRT_MAX_TEXTURES) return CRT_E_OUT_OF_RANGE;
    HIPCHK(hipMemcpyAsync(g.textures, textures, count * sizeof(CrtTexture), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    return CRT_OK;
}

int crt_upload_texels(const void* rgb8, size_t byteOffset, size_t bytes)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (bytes == 0) return CRT_OK;
    if (!rgb8) return CRT_E_BAD_ARGUMENT;
    if (byteOffset + bytes > g.texelByteCap) return CRT_E_OUT_OF_RANGE;
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

int crt_upload_instances(const void* instances, size_t first, size_t count)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (count == 0) return CRT_OK;
    if (!instances) return CRT_E_BAD_ARGUMENT;
    if (first + count > CRT_MAX_INSTANCES) return CRT_E_OUT_OF_RANGE;
    const CrtMeshInstance* in = static_cast<const CrtMeshInstance*>(instances);
    for (size_t i = 0; i < count; ++i) if (in[i].meshIndex >= CRT_MAX_MESHES) return CRT_E_BAD_ARGUMENT;
    HIPCHK(hipMemcpyAsync(g.instances + first, instances, count * sizeof(CrtMeshInstance), hipMemcpyHostToDevice, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    memcpy(g.hInstances + first, instances, count * sizeof(CrtMeshInstance));
    if (first + count > g.instHigh) g.instHigh = (uint32_t)(first + count);
    return rebuild_instance_bounds();
}

// Feedback launch lists for the megakernel (lane_pixel / crt_order_kernel). Buffers follow the frame geometry; a
// change of geometry resets to the identity order. Every frame starts by turning the previous frame's per-tile
// costs into this frame's lists (and zeroing the costs), so the sort is inside the frame but outside the Trace
// event pair.
static int prepare_launch_lists(CrtFrame& F, unsigned& grid)
{
    const int key[6] = { g.width, g.height, g.bandRows, g.rank, g.nRanks, F.slotsPerXcd };
    F.listCap = F.slotsPerXcd + 3 * CRT_MAX_SPLIT;
    const size_t need = (size_t)8 * (size_t)F.listCap;
    if (need > g.orderCap) {
        if (g.tileOrder[0]) (void)hipFree(g.tileOrder[0]);
        if (g.tileLen[0]) (void)hipFree(g.tileLen[0]);
        if (g.tileCost) (void)hipFree(g.tileCost);
        g.tileOrder[0] = nullptr; g.tileLen[0] = nullptr; g.tileCost = nullptr; g.orderCap = 0;
        HIPCHK(hipMalloc(&g.tileOrder[0], sizeof(uint32_t) * need));
        HIPCHK(hipMalloc(&g.tileLen[0], sizeof(uint32_t) * 8));
        HIPCHK(hipMalloc(&g.tileCost, sizeof(uint32_t) * need));
        g.orderCap = need; g.orderSlots = -1;
    }
    if (g.orderSlots != F.slotsPerXcd || memcmp(key, g.orderKey, sizeof key) != 0) {
        HIPCHK(hipMemsetAsync(g.tileCost, 0, sizeof(uint32_t) * need, g.stream));
        crt_identity_order_kernel<<<(8 * F.slotsPerXcd + 255) / 256, 256, 0, g.stream>>>(g.tileOrder[0], g.tileLen[0], F.slotsPerXcd, F.listCap);
        g.orderSlots = F.slotsPerXcd; memcpy(g.orderKey, key, sizeof key);
    } else {
        crt_order_kernel<<<8, 1024, 0, g.stream>>>(g.tileCost, g.tileOrder[0], g.tileLen[0], F.slotsPerXcd, F.listCap);
    }
    HIPCHK(hipGetLastError());
    F.order = g.tileOrder[0]; F.listLen = g.tileLen[0]; F.cost = g.tileCost;
    grid = 8u * (unsigned)F.listCap;
    return CRT_OK;
}

// The Trace launch(es) of one frame, by kernel structure (default: megakernel with feedback launch lists).
static int launch_trace(const CrtDevScene& S, const CrtFrame& F, int flags, unsigned grid)
{
    const bool count = (flags & CRT_RENDER_COUNTERS) != 0;
    if (count) HIPCHK(hipMemsetAsync(g.counters, 0, 12 * sizeof(unsigned long long), g.stream));
    if (flags & CRT_RENDER_STAMPS) {                      // diagnostic launch with per-wave stamps
        unsigned waves = grid;
        if (g.persistent) {
            const unsigned tiles = (unsigned)F.ownedTileRows * (unsigned)F.tilesX;
            waves = (unsigned)(g.numCUs * g.wavesPerCU);
            if (waves > tiles) waves = tiles;
        }
        const size_t need = (16 + (size_t)waves * 8) * sizeof(unsigned long long);
        if (need > g.stampBytes) {
            if (g.stamps) (void)hipFree(g.stamps);
            g.stamps = nullptr; g.stampBytes = 0;
            HIPCHK(hipMalloc(&g.stamps, need));
            g.stampBytes = need;
        }
        g.stampWaves = waves;
        HIPCHK(hipMemsetAsync(g.stamps, 0, need, g.stream));
        if (g.persistent) {
            HIPCHK(hipMemsetAsync(g.queues, 0, sizeof(CrtQueues), g.stream));
            crt_trace_persistent_kernel<false, true><<<waves, CRT_BLOCK, 0, g.stream>>>(S, F, g.out, g.stamps, g.queues);
        } else {
            crt_trace_kernel<false, true><<<grid, CRT_BLOCK, 0, g.stream>>>(S, F, g.out, g.stamps);
        }
    } else if (g.ldsTiles) {                               // resident 1024-thread workgroups, hot tiles in LDS (crt_ldstile.h)
        HIPCHK(hipMemsetAsync(g.listNext, 0, 8 * sizeof(uint32_t), g.stream));
        if (count) crt_trace_lds_kernel<true><<<g.numCUs, 64 * CRT_LDS_WAVES, 0, g.stream>>>(S, F, g.out, g.counters, g.listNext);
        else crt_trace_lds_kernel<false><<<g.numCUs, 64 * CRT_LDS_WAVES, 0, g.stream>>>(S, F, g.out, g.counters, g.listNext);
    } else if (g.persistent) {                             // resident waves pulling tiles from per-XCD queues
        const unsigned tiles = (unsigned)F.ownedTileRows * (unsigned)F.tilesX;
        unsigned waves = (unsigned)(g.numCUs * g.wavesPerCU);
        if (waves > tiles) waves = tiles;
        HIPCHK(hipMemsetAsync(g.queues, 0, sizeof(CrtQueues), g.stream));
        if (count) crt_trace_persistent_kernel<true><<<waves, CRT_BLOCK, 0, g.stream>>>(S, F, g.out, g.counters, g.queues);
        else crt_trace_persistent_kernel<false><<<waves, CRT_BLOCK, 0, g.stream>>>(S, F, g.out, g.counters, g.queues);
    } else if (g.wavefront) {                              // bounce 0, ballot compaction, bounce 1
        const unsigned ownedPixels = (unsigned)F.ownedTileRows * CRT_TILE * (unsigned)F.width;
        const unsigned grid2 = (ownedPixels + CRT_BLOCK - 1) / CRT_BLOCK;
        HIPCHK(hipMemsetAsync(g.bounceCount, 0, sizeof(uint32_t), g.stream));
        if (count) {
            crt_primary_kernel<true><<<grid, CRT_BLOCK, 0, g.stream>>>(S, F, g.out, g.counters, g.bounceQueue, g.bounceCount);
            crt_bounce_kernel<true><<<grid2, CRT_BLOCK, 0, g.stream>>>(S, F, g.out, g.counters, g.bounceQueue, g.bounceCount);
        } else {
            crt_primary_kernel<false><<<grid, CRT_BLOCK, 0, g.stream>>>(S, F, g.out, g.counters, g.bounceQueue, g.bounceCount);
            crt_bounce_kernel<false><<<grid2, CRT_BLOCK, 0, g.stream>>>(S, F, g.out, g.counters, g.bounceQueue, g.bounceCount);
        }
    } else if (count) {
        crt_trace_kernel<true><<<grid, CRT_BLOCK, 0, g.stream>>>(S, F, g.out, g.counters);
    } else {
        crt_trace_kernel<false><<<grid, CRT_BLOCK, 0, g.stream>>>(S, F, g.out, g.counters);
    }
    HIPCHK(hipGetLastError());
    return CRT_OK;
}

int crt_render(const CrtTraceArgs* args, const float invView[16], const float invProj[16], int flags)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!args || !invView || !invProj) return CRT_E_BAD_ARGUMENT;
    if (args->numMeshes > CRT_MAX_INSTANCES) return CRT_E_OUT_OF_RANGE;
    if (!g.sceneValid) return CRT_E_BAD_ARGUMENT;
    int rc = collect_timing();
    if (rc) return rc;
    CrtFrame F; fill_frame(F, args, invView, invProj);
    CrtDevScene S; fill_scene(S, args->numMeshes);
    if (F.gridBlocks == 0) return CRT_OK;
    unsigned grid = (unsigned)F.gridBlocks;
    if (g.feedback && !g.persistent && !g.wavefront) { rc = prepare_launch_lists(F, grid); if (rc) return rc; }

    // events: [0] frame start, [1] Trace start, [2] Trace end, [3] end of PostProcess = frame end.
    // A plain frame records only two (RayGen is fused, PostProcess off): [0] == [1], [2] == [3].
    g.evRaygen = (flags & CRT_RENDER_WRITE_RAYS) != 0;
    g.evPost = (flags & CRT_RENDER_POSTPROCESS) != 0;
    HIPCHK(hipEventRecord(g.ev[0], g.stream));
    if (g.evRaygen) {
        crt_raygen_kernel<<<grid, CRT_BLOCK, 0, g.stream>>>(F, g.rays);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(g.ev[1], g.stream));
    }
    rc = launch_trace(S, F, flags, grid);
    if (rc) return rc;
    HIPCHK(hipEventRecord(g.ev[2], g.stream));
    if (g.evPost) {
        crt_postprocess_kernel<<<grid, CRT_BLOCK, 0, g.stream>>>(F, g.out);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(g.ev[3], g.stream));
    }
    g.pendingTiming = true; g.pendingFlags = flags;
    if (!(flags & CRT_RENDER_ASYNC)) HIPCHK(hipStreamSynchronize(g.stream));   // the reference's clFinish (Renderer.cpp:367)
    return CRT_OK;
}

int crt_sync(void)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    HIPCHK(hipStreamSynchronize(g.stream));
    return CRT_OK;
}

int crt_query_hits(const float* origins, const float* dirs, int n, uint32_t numInstances, CrtRayHit* out)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (n <= 0) return CRT_OK;
    if (!origins || !dirs || !out || numInstances > CRT_MAX_INSTANCES) return CRT_E_BAD_ARGUMENT;
    if (!g.sceneValid) return CRT_E_BAD_ARGUMENT;
    int rc = collect_timing();
    if (rc) return rc;
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
    HIPCHK(hipMemsetAsync(g.counters, 0, 12 * sizeof(unsigned long long), g.stream));
    CrtDevScene S; fill_scene(S, numInstances);
    crt_query_kernel<<<(unsigned)((n + CRT_BLOCK - 1) / CRT_BLOCK), CRT_BLOCK, 0, g.stream>>>(S, dO, dD, n, dH, g.counters);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dH, sizeof(CrtRayHit) * (size_t)n, hipMemcpyDeviceToHost, g.stream));
    unsigned long long c[12];
    HIPCHK(hipMemcpyAsync(c, g.counters, sizeof c, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream));
    CrtCounters& o = g.lastCounters;
    o.rays = c[0]; o.primary = c[1]; o.secondary = c[2]; o.hits = c[3]; o.misses = c[4]; o.traversals = c[5];
    o.pops = c[6]; o.innerVisits = c[7]; o.triTests = c[8]; o.capHits = c[9]; o.stackOverflows = c[10]; o.maxStack = c[11];
    return CRT_OK;
}

int crt_read_output(float* dst, size_t floats)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!dst || floats != (size_t)g.width * (size_t)g.height * 4) return CRT_E_BAD_ARGUMENT;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(dst, g.out, floats * sizeof(float), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_read_output_rows(float* dst, int row0, int rows)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!dst || row0 < 0 || rows < 0 || row0 + rows > g.height) return CRT_E_BAD_ARGUMENT;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(dst, g.out + (size_t)row0 * (size_t)g.width, (size_t)rows * (size_t)g.width * sizeof(float4), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_read_rays(float* dst, size_t floats)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!dst || floats != (size_t)g.width * (size_t)g.height * 3) return CRT_E_BAD_ARGUMENT;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(dst, g.rays, floats * sizeof(float), hipMemcpyDeviceToHost));
    return CRT_OK;
}

void* crt_output_device_ptr(void) { return g.initialized ? (void*)g.out : nullptr; }

float crt_last_kernel_ms(int which)
{
    if (!g.initialized || which < 0 || which > 3) return -1.0f;
    if (collect_timing() != CRT_OK) return -1.0f;
    return g.ms[which];
}

int crt_debug_read_stamps(uint64_t* dst, size_t maxWaves, size_t* numWaves)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!numWaves) return CRT_E_BAD_ARGUMENT;
    *numWaves = g.stampWaves;
    if (!dst || !g.stamps) return CRT_OK;
    const size_t n = maxWaves < g.stampWaves ? maxWaves : g.stampWaves;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy(dst, g.stamps + 16, n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_get_counters(CrtCounters* out)
{
    if (!g.initialized) return CRT_E_NOT_INITIALIZED;
    if (!out) return CRT_E_BAD_ARGUMENT;
    int rc = collect_timing();
    if (rc) return rc;
    *out = g.lastCounters;
    return CRT_OK;
}

} // extern "C"
