// crt_refill.h -- Trace with IN-TILE LANE REFILL (round 5; CRT_KERNEL=refill): north_star's "wavefront ballot / compaction used to
// keep ray packets coherent", applied inside the wave instead of across waves.
//
// The default megakernel (crt_kernels.h) gives a wave one 8x8 tile and one pixel per lane: a lane whose path ends early (sky after
// the instance cull, a short traversal, a bounce ray that leaves the scene) idles until the slowest lane of the tile is done --
// 28 of 64 lanes work per vector-memory instruction on multi-1M (profiles/r04_summary.md, valu.lane_utilisation 0.442). The two
// compaction forms measured before both lost coherence: `persistent` (r1) refilled lanes from OTHER tiles, `wavefront` regroups
// rays across waves through memory (two tails, atomics, arrival order). Here a wave owns a BLOCK of CRT_REFILL_TILES 8x8 tiles
// side by side (16x8 pixels by default, Morton order inside each tile) and a lane whose path has ended takes the block's next
// pixel: one ballot of the lanes that want a pixel, their rank from mbcnt, a cursor in an SGPR. No memory traffic, no
// synchronisation between waves, and every ray of the wave stays within 16 pixels of the others.
//
// Per ray nothing changes -- the same RayGen, instance order, node visits, triangle tests, shading arithmetic as the megakernel
// (kernel_main.cl:164-287), so frames and work counters are bit-identical; what changes is which rays share a wave-level
// instruction. Shading (kernel_main.cl:219-272) moves INTO the traversal loop as a "service step": lanes whose ray is finished
// wait until CRT_REFILL_BATCH of them have gathered (or until they outnumber half the lanes still traversing), then shade
// together, start their bounce ray or store their pixel and draw the next one. The pixel index (and the bounce number in bit 31)
// and the path's energy wait in parked LDS slots (CrtStackT<2>), so the kernel carries no more registers across the traversal than the megakernel does.
//
// Limits: up to 64 instances (one candidate mask), no shadow rays / refraction / instance tree: those frames take the megakernel.
#pragma once
#include "crt_kernels.h"

#ifndef CRT_REFILL_TILES
#define CRT_REFILL_TILES 2        // 8x8 tiles per block, side by side
#endif
#ifndef CRT_REFILL_BATCH
#define CRT_REFILL_BATCH 32       // finished rays that gather before a service step is run for them
#endif
#ifndef CRT_REFILL_RATIO
#define CRT_REFILL_RATIO 1         // ... or when RATIO x (finished lanes) outnumber the lanes still traversing
#endif
#define CRT_REFILL_PIXELS (64 * CRT_REFILL_TILES)

// Block of workgroup `b`: the launch's CrtFrame counts BLOCKS where the megakernel's counts tiles (tilesX = blocks per tile row,
// slotsPerXcd, gridBlocks, the launch lists and the per-slot costs), so lane_pixel's dealing -- block b on XCD b % 8, XCD x owning
// the tile rows k with k % 8 == x, heaviest first when a feedback list exists -- carries over unchanged. Wave-uniform.
__device__ __forceinline__ bool refill_block(const CrtFrame& F, int b, int& tx0, int& tileRow, int& costSlot, int tiles = CRT_REFILL_TILES)
{
    const int xcd = b & 7;
    int slot = b >> 3;
    costSlot = -1;
    if (F.order) {
        if ((uint32_t)slot >= F.listLen[xcd]) return false;
        const uint32_t e = __builtin_amdgcn_readfirstlane(F.order[xcd * F.listCap + slot]);
        slot = (int)(e & 0x0FFFFFFFu);
    } else if (slot >= F.slotsPerXcd) return false;
    costSlot = xcd * F.slotsPerXcd + slot;
    const int round = slot / F.tilesX;
    const int bx = slot - round * F.tilesX;
    const int k = round * 8 + xcd;
    if (k >= F.ownedTileRows) return false;
    const int bandK = k / F.tileRowsPerBand;
    tileRow = (F.rank + bandK * F.nRanks) * F.tileRowsPerBand + (k - bandK * F.tileRowsPerBand);
    tx0 = bx * tiles;
    return true;
}

// pixel p of the block: tile p / 64, Morton position p % 64 inside it
__device__ __forceinline__ void refill_pixel(int tx0, int tileRow, uint32_t p, int& px, int& py)
{
    const int m = (int)(p & 63u);
    const int lx = (m & 1) | ((m >> 1) & 2) | ((m >> 2) & 4);
    const int ly = ((m >> 1) & 1) | ((m >> 2) & 2) | ((m >> 3) & 4);
    px = (tx0 + (int)(p >> 6)) * CRT_TILE + lx;
    py = tileRow * CRT_TILE + ly;
}

// STAMP (diagnostic, CRT_RENDER_STAMPS): per-wave time stamps and wave-level step counts in the megakernel's record layout
// (tools/wave_timeline.py), plus the number of service steps in the upper half of word 6.
template <bool COUNT, bool STAMP = false>
__global__ __launch_bounds__(CRT_BLOCK, (COUNT || STAMP) ? CRT_WAVES_PER_SIMD_COUNT : CRT_WAVES_PER_SIMD)
void crt_trace_refill_kernel(CrtDevScene S, CrtFrame F, float4* __restrict__ out, unsigned long long* __restrict__ counters)
{
    __shared__ uint32_t s_stack[CRT_LDS_SLOTS * CRT_BLOCK];
    typedef CrtStackT<2> Stack;                              // parked slots: [0] pixel index within the block | bounce << 31, [1] the path's energy
    const Stack stack = { (crt_lds_u32_ptr)s_stack + threadIdx.x, S.stackOverflow };
    LaneCounters lc; zero_counters(lc);
    unsigned long long t0rt = 0, t0c = 0;
    if (STAMP) { t0rt = __builtin_amdgcn_s_memrealtime(); t0c = __builtin_amdgcn_s_memtime(); }
    const unsigned long long tc0 = F.cost ? __builtin_amdgcn_s_memtime() : 0ull;
    int tx0 = 0, tileRow = 0, costSlot = -1;
    const bool valid = refill_block(F, blockIdx.x, tx0, tileRow, costSlot);
    if (valid) {
        uint32_t cursor = 0;                                 // next pixel of the block nobody has taken yet (wave-uniform)
        PathState ps;
        ps.o = mk3(0.f, 0.f, 0.f); ps.d = ps.o; ps.result = ps.o; ps.energy = 1.0f;
        Closest c;
        c.distance = 99999.0f; c.hitInstance = 0; c.anyHit = 0;
        c.hit.t = 0.0f; c.hit.u = 0.0f; c.hit.v = 0.0f; c.hit.tri = 0;
        Traversal<COUNT> T; T.reset();
        unsigned long long cand = 0;
        bool havePath = false;
        const uint32_t cnt = S.numInstances < 64u ? S.numInstances : 64u;
        for (;;) {
            // a lane's ray is finished when it is inside no instance and has no candidate left (kernel_main.cl:198-217 ran out)
            const bool rayDone = havePath && !T.active && cand == 0;
            const unsigned long long mDone = __ballot(rayDone);
            const unsigned long long mBusy = __ballot(havePath && !rayDone);
            if ((mDone | mBusy) == 0 && cursor >= (uint32_t)CRT_REFILL_PIXELS) break;       // every pixel of the block is stored
            const uint32_t nDone = (uint32_t)__popcll(mDone), nBusy = (uint32_t)__popcll(mBusy);
            if (STAMP) { if (first_active_lane()) lc.pops++; }
            // Service step: shade the finished rays, start bounce rays, store finished pixels, hand out new pixels. Run when enough
            // finished lanes have gathered to pay for the step's instruction stream, when they outnumber half of the lanes still
            // traversing (waiting would idle more lanes than it saves steps), or when nobody is traversing at all.
            if (nDone >= (uint32_t)CRT_REFILL_BATCH || (uint32_t)CRT_REFILL_RATIO * nDone > nBusy || nBusy == 0) {
                bool newRay = false;
                if (STAMP) { if (first_active_lane()) lc.capHits++; }
                if (rayDone) {
                    T.reset();                                // (the traversal is idle: nothing of it needs to survive the shading code)
                    const uint32_t st = stack.parked(0);
                    const int bounce = (int)(st >> 31);
                    ps.energy = __uint_as_float(stack.parked(1));      // (waits in LDS through the traversals, as in the megakernel's SHADOW instantiations)
                    const int cont = shade_bounce(S, c, ps, bounce, F.lightY, F.lightZ);
                    c.distance = 99999.0f; c.hitInstance = 0; c.anyHit = 0;
                    c.hit.t = 0.0f; c.hit.u = 0.0f; c.hit.v = 0.0f; c.hit.tri = 0;
                    if (COUNT) { if (cont) lc.hits++; else lc.misses++; }
                    if (cont != 0 && bounce == 0) {           // kernel_main.cl:187: the second iteration of the bounce loop
                        stack.park(0, st | 0x80000000u);
                        stack.park(1, __float_as_uint(ps.energy));
                        newRay = true;
                        if (COUNT) { lc.rays++; lc.secondary++; }
                    } else {
                        // the plain HDR value; upstream's per-pixel stages behind Trace are applied by the pass below the loop
                        int qx, qy;
                        refill_pixel(tx0, tileRow, st & 0x7FFFFFFFu, qx, qy);
                        out[(size_t)qy * (size_t)F.width + (size_t)qx] = make_float4(ps.result.x, ps.result.y, ps.result.z, 1.0f);
                        havePath = false;
                    }
                }
                // refill: the lanes without a path take the block's next pixels in lane order
                const unsigned long long mWant = __ballot(!havePath);
                if (mWant != 0 && cursor < (uint32_t)CRT_REFILL_PIXELS) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mWant >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mWant, 0u));
                    const uint32_t p = cursor + rank;
                    cursor += (uint32_t)__popcll(mWant);
                    if (!havePath && p < (uint32_t)CRT_REFILL_PIXELS) {
                        int px, py;
                        refill_pixel(tx0, tileRow, p, px, py);
                        if (px < F.width && py < F.height) {
                            havePath = true;
                            stack.park(0, p);
                            ps.o = mk3(F.camPos[0], F.camPos[1], F.camPos[2]);
                            // opaque copies: otherwise (float)width / (float)height are hoisted out of the loop and kept alive (spilled) through it
                            int w2 = F.width, h2 = F.height;
                            asm volatile("" : "+s"(w2), "+s"(h2));
                            ps.d = raygen_dir(F, px, py, w2, h2);
                            ps.result = mk3(0.0f, 0.0f, 0.0f);
                            stack.park(1, __float_as_uint(1.0f));
                            newRay = true;
                            if (COUNT) { lc.rays++; lc.primary++; }
                        }
                    }
                }
                if (newRay) cand = candidate_mask<COUNT>(S, ps.o, ps.d, 0u, cnt, lc);
            }
            // one traversal trip (closest_hit's: enter -> inner -> leaf -> inner) for the lanes that have something to traverse
            if (havePath && !T.active && cand != 0) {
                if (STAMP) { if (first_active_lane()) lc.traversals++; }
                const uint32_t k = (uint32_t)__ffsll((long long)cand) - 1u;
                cand &= cand - 1;
                T.enter(S, k, ps.o, ps.d, c.distance, lc);
            }
            trip_steps<COUNT, STAMP, false>(S, stack, T, c, lc, false);
        }
    }
    // The per-pixel stages that follow Trace upstream (its RGBA8 render target, PostProcess; the megakernel's epilogue): one pass over
    // the block's pixels at full lane occupancy, on the values this wave stored above (workgroup-scope release / acquire: the wave
    // reads its own stores). Kept out of the loop: seven powf per pixel inside the service step cost the traversal its registers.
    if (valid && (F.epilogue != 0 || F.packOut != nullptr)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t p = threadIdx.x & 63u; p < (uint32_t)CRT_REFILL_PIXELS; p += 64u) {
            int qx, qy;
            refill_pixel(tx0, tileRow, p, qx, qy);
            if (qx >= F.width || qy >= F.height) continue;
            const size_t idx = (size_t)qy * (size_t)F.width + (size_t)qx;
            const float4 v = out[idx];
            v3 rgb = mk3(v.x, v.y, v.z);
            if (F.epilogue & CRT_EPILOGUE_QUANTIZE) rgb = mk3(quantize1(rgb.x), quantize1(rgb.y), quantize1(rgb.z));
            if (F.epilogue & CRT_EPILOGUE_POST) {
                rgb = post_pixel(rgb, qx, qy, F.width, F.height);
                if (F.epilogue & CRT_EPILOGUE_QUANTIZE) rgb = mk3(quantize1(rgb.x), quantize1(rgb.y), quantize1(rgb.z));
            }
            if (F.epilogue != 0) out[idx] = make_float4(rgb.x, rgb.y, rgb.z, 1.0f);
            if (F.packOut) F.packOut[idx] = unorm8(rgb.x) | (unorm8(rgb.y) << 8) | (unorm8(rgb.z) << 16) | 0xFF000000u;
        }
    }
    if (F.cost && costSlot >= 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - tc0;
        if ((threadIdx.x & 63) == 0) atomicAdd(&F.cost[costSlot], dt > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)dt);
    }
    if (COUNT) flush_counters(lc, counters);
    if (STAMP) {
        const unsigned long long t1c = __builtin_amdgcn_s_memtime(), t1rt = __builtin_amdgcn_s_memrealtime();
        const uint32_t wOuter = wave_sum(lc.pops), wEnter = wave_sum(lc.traversals), wDescent = wave_sum(lc.innerVisits), wService = wave_sum(lc.capHits),
                       wLeaf = wave_sum(lc.triTests), laneVisits = wave_sum(lc.rays), wInner2 = wave_sum(lc.hits), wLeafIters = wave_sum(lc.misses);
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* st = counters + 16 + (size_t)blockIdx.x * 8;
            st[4] = wOuter | ((unsigned long long)wInner2 << 32); st[5] = wEnter | ((unsigned long long)wLeafIters << 32);
            st[6] = wDescent | ((unsigned long long)wService << 32); st[7] = ((unsigned long long)wLeaf << 32) | laneVisits;
            st[0] = t0rt; st[1] = t1rt; st[2] = t1c - t0c;
            st[3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// crt_trace_block_kernel (CRT_KERNEL=block) -- PHASE-SEPARATED compaction inside the wave.
//
// What the refill form above taught (profiles/r05_in_wave_compaction_timeline.txt, r05refill_summary.md): a wave that mixes rays at different stages executes every step kind
// (enter / inner / leaf) on every trip, so the wave-level instruction count falls far less than the lane utilisation rises, and
// the service steps cost what the packing saves. Here the rays of a block are regrouped BETWEEN the stages instead, so every
// traversal still runs as a packet of rays that start together:
//   phase 0  classify the block's pixels: RayGen + the instance cull for 64 pixels at a time. A primary ray without a candidate
//            instance is a sky pixel and is finished on the spot (all such lanes shade together); the others' pixel indices are
//            appended to a list in the wave's parked LDS slots (ballot + mbcnt: no atomics, the list is the wave's own).
//   phase 1  the listed primary rays, 64 per pass, as DENSE packets: closest hit, shading; a path that continues stores its
//            partial radiance in the frame and appends its bounce ray {origin, energy, direction, pixel} to the block's queue
//            (a 32-byte record in global memory that only this wave touches: written and read back through its own L2).
//   phase 2  the queued bounce rays, 64 per pass, again dense.
// Per ray nothing changes (the same closest_hit / shade_bounce as the megakernel; result = partial + bounce-1 terms in the order of
// kernel_main.cl:267), so frames and work counters are bit-identical. The cull of phase 0 is not counted; closest_hit counts it
// when the ray is traced (a culled instance is upstream's one pop + one root visit either way, so the totals are the same).
// ------------------------------------------------------------------------------------------------------------------------------
#ifndef CRT_BLOCK_TILES
#define CRT_BLOCK_TILES 2
#endif
#define CRT_BLOCK_PIXELS (64 * CRT_BLOCK_TILES)

template <bool COUNT, bool STAMP = false>
__global__ __launch_bounds__(CRT_BLOCK, (COUNT || STAMP) ? CRT_WAVES_PER_SIMD_COUNT : CRT_WAVES_PER_SIMD)
void crt_trace_block_kernel(CrtDevScene S, CrtFrame F, float4* __restrict__ out, unsigned long long* __restrict__ counters, CrtBounceRay* __restrict__ queue)
{
    __shared__ uint32_t s_stack[CRT_LDS_SLOTS * CRT_BLOCK];
    typedef CrtStackT<CRT_BLOCK_TILES> Stack;                // parked slots: the list of primary rays that have a candidate instance
    const Stack stack = { (crt_lds_u32_ptr)s_stack + threadIdx.x, S.stackOverflow };
    const crt_lds_u32_ptr list = (crt_lds_u32_ptr)s_stack + Stack::kLds * 64;
    const uint32_t lane = threadIdx.x & 63u;
    LaneCounters lc; zero_counters(lc);
    unsigned long long t0rt = 0, t0c = 0;
    if (STAMP) { t0rt = __builtin_amdgcn_s_memrealtime(); t0c = __builtin_amdgcn_s_memtime(); }
    const unsigned long long tc0 = F.cost ? __builtin_amdgcn_s_memtime() : 0ull;
    int tx0 = 0, tileRow = 0, costSlot = -1;
    const bool valid = refill_block(F, blockIdx.x, tx0, tileRow, costSlot, CRT_BLOCK_TILES);
    if (valid) {
        const uint32_t cnt = S.numInstances < 64u ? S.numInstances : 64u;
        CrtBounceRay* __restrict__ q = queue + (size_t)costSlot * CRT_BLOCK_PIXELS;
        uint32_t nList = 0, nQueue = 0;                      // wave-uniform
        // ---- phase 0: classify ----
        for (uint32_t t = 0; t < (uint32_t)CRT_BLOCK_TILES; ++t) {
            const uint32_t p = t * 64u + lane;
            int px, py;
            refill_pixel(tx0, tileRow, p, px, py);
            const bool inFrame = px < F.width && py < F.height;
            bool listed = false;
            if (inFrame) {
                PathState ps;
                ps.o = mk3(F.camPos[0], F.camPos[1], F.camPos[2]); ps.d = raygen_dir(F, px, py); ps.result = mk3(0.0f, 0.0f, 0.0f); ps.energy = 1.0f;
                LaneCounters none;
                const unsigned long long cand = candidate_mask<false>(S, ps.o, ps.d, 0u, cnt, none);
                if (cand == 0 && S.numInstances <= 64u) {
                    // no instance can be hit: closest_hit would return the initial miss (kernel_main.cl:219-224: skybox, break)
                    if (COUNT) { lc.rays++; lc.primary++; lc.misses++; lc.traversals += cnt; lc.pops += cnt; lc.innerVisits += cnt; lc.culled += cnt; }
                    Closest c;
                    c.distance = 99999.0f; c.hitInstance = 0; c.anyHit = 0; c.hit.t = 0.0f; c.hit.u = 0.0f; c.hit.v = 0.0f; c.hit.tri = 0;
                    (void)shade_bounce(S, c, ps, 0, F.lightY, F.lightZ);
                    out[(size_t)py * (size_t)F.width + (size_t)px] = make_float4(ps.result.x, ps.result.y, ps.result.z, 1.0f);
                } else listed = true;
            }
            const unsigned long long m = __ballot(listed);
            if (listed) list[nList + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = p;
            nList += (uint32_t)__popcll(m);
        }
        // ---- phase 1: the primary rays that may hit something, as dense packets ----
        // (what is not used inside the traversal is recomputed behind it rather than kept alive through it: the pixel index, as in
        // the megakernel; the opaque asm keeps the two computations apart)
        for (uint32_t base = 0; base < nList; base += 64u) {
            bool cont = false;
            PathState ps;
            ps.o = mk3(F.camPos[0], F.camPos[1], F.camPos[2]); ps.d = ps.o; ps.result = mk3(0.0f, 0.0f, 0.0f); ps.energy = 1.0f;
            if (base + lane < nList) {
                {
                    int px, py;
                    refill_pixel(tx0, tileRow, list[base + lane], px, py);
                    ps.d = raygen_dir(F, px, py);
                }
                if (COUNT) { lc.rays++; lc.primary++; }
                const Closest c = closest_hit<COUNT, STAMP>(S, ps.o, ps.d, stack, lc);
                cont = shade_bounce(S, c, ps, 0, F.lightY, F.lightZ) != 0;
                if (COUNT) { if (cont) lc.hits++; else lc.misses++; }
            }
            uint32_t base2 = base, lane2 = lane;
            asm volatile("" : "+s"(base2), "+v"(lane2));
            uint32_t pixel = 0;
            if (base2 + lane2 < nList) {
                int px, py;
                refill_pixel(tx0, tileRow, list[base2 + lane2], px, py);
                pixel = (uint32_t)py * (uint32_t)F.width + (uint32_t)px;
                out[pixel] = make_float4(ps.result.x, ps.result.y, ps.result.z, 1.0f);
            }
            const unsigned long long m = __ballot(cont);
            if (cont) {
                CrtBounceRay r;
                r.ox = ps.o.x; r.oy = ps.o.y; r.oz = ps.o.z; r.energy = ps.energy;
                r.dx = ps.d.x; r.dy = ps.d.y; r.dz = ps.d.z; r.pixel = pixel;
                q[nQueue + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = r;
            }
            nQueue += (uint32_t)__popcll(m);
        }
        // ---- phase 2: the bounce rays, as dense packets (the wave reads its own stores: workgroup-scope release / acquire) ----
        if (nQueue != 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        for (uint32_t base = 0; base < nQueue; base += 64u) {
            if (base + lane < nQueue) {
                PathState ps;
                {
                    const CrtBounceRay r = q[base + lane];
                    ps.o = mk3(r.ox, r.oy, r.oz); ps.d = mk3(r.dx, r.dy, r.dz); ps.energy = r.energy;
                }
                if (COUNT) { lc.rays++; lc.secondary++; }
                const Closest c = closest_hit<COUNT, STAMP>(S, ps.o, ps.d, stack, lc);
                // the partial radiance and the pixel index are fetched behind the traversal (nothing kept alive through it)
                uint32_t base2 = base, lane2 = lane;
                asm volatile("" : "+s"(base2), "+v"(lane2));
                const uint32_t pixel = q[base2 + lane2].pixel;
                const float4 partial = out[pixel];
                ps.result = mk3(partial.x, partial.y, partial.z);
                const bool cont = shade_bounce(S, c, ps, 1, F.lightY, F.lightZ) != 0;
                if (COUNT) { if (cont) lc.hits++; else lc.misses++; }
                out[pixel] = make_float4(ps.result.x, ps.result.y, ps.result.z, 1.0f);
            }
        }
    }
    // upstream's per-pixel stages behind Trace (see crt_trace_refill_kernel)
    if (valid && (F.epilogue != 0 || F.packOut != nullptr)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t p = lane; p < (uint32_t)CRT_BLOCK_PIXELS; p += 64u) {
            int qx, qy;
            refill_pixel(tx0, tileRow, p, qx, qy);
            if (qx >= F.width || qy >= F.height) continue;
            const size_t idx = (size_t)qy * (size_t)F.width + (size_t)qx;
            const float4 v = out[idx];
            v3 rgb = mk3(v.x, v.y, v.z);
            if (F.epilogue & CRT_EPILOGUE_QUANTIZE) rgb = mk3(quantize1(rgb.x), quantize1(rgb.y), quantize1(rgb.z));
            if (F.epilogue & CRT_EPILOGUE_POST) {
                rgb = post_pixel(rgb, qx, qy, F.width, F.height);
                if (F.epilogue & CRT_EPILOGUE_QUANTIZE) rgb = mk3(quantize1(rgb.x), quantize1(rgb.y), quantize1(rgb.z));
            }
            if (F.epilogue != 0) out[idx] = make_float4(rgb.x, rgb.y, rgb.z, 1.0f);
            if (F.packOut) F.packOut[idx] = unorm8(rgb.x) | (unorm8(rgb.y) << 8) | (unorm8(rgb.z) << 16) | 0xFF000000u;
        }
    }
    if (F.cost && costSlot >= 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - tc0;
        if ((threadIdx.x & 63) == 0) atomicAdd(&F.cost[costSlot], dt > 0x0FFFFFFFull ? 0x0FFFFFFFu : (uint32_t)dt);
    }
    if (COUNT) flush_counters(lc, counters);
    if (STAMP) {
        const unsigned long long t1c = __builtin_amdgcn_s_memtime(), t1rt = __builtin_amdgcn_s_memrealtime();
        const uint32_t wOuter = wave_sum(lc.pops), wEnter = wave_sum(lc.traversals), wDescent = wave_sum(lc.innerVisits),
                       wLeaf = wave_sum(lc.triTests), laneVisits = wave_sum(lc.rays), wInner2 = wave_sum(lc.hits), wLeafIters = wave_sum(lc.misses);
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* st = counters + 16 + (size_t)blockIdx.x * 8;
            st[4] = wOuter | ((unsigned long long)wInner2 << 32); st[5] = wEnter | ((unsigned long long)wLeafIters << 32);
            st[6] = wDescent; st[7] = ((unsigned long long)wLeaf << 32) | laneVisits;
            st[0] = t0rt; st[1] = t1rt; st[2] = t1c - t0c;
            st[3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
        }
    }
}
