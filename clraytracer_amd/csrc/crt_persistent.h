// crt_persistent.h -- the production Trace kernel: persistent wave64s with per-lane pixel refill.
//
// Why (measured on MI355X with the one-tile-per-wave kernel of crt_kernels.h in its first, lock-step form, multi-1M 1920x1080):
// only 16 of 64 lanes were active per node-visit trip and the slowest wave needed 1979 trips for a
// ray whose own path is <= 361 visits, because (a) lanes that finish their ray idle until the whole
// 8x8 tile is done, (b) the descend-then-leaf loop nest makes every lane wait for the longest descent
// segment of its neighbours. The kernel finished at 1.1 ms although the machine was full for only
// the first 0.43 ms (profiles/, tools/wave_timeline.py).
//
// Structure (per-ray arithmetic and order are untouched, so pixels stay bit-identical):
//   * one workgroup = one wave64 that stays resident and pulls 8x8 pixel tiles from 8 per-XCD queues
//     (tile rows interleaved over XCDs for L2 locality; a wave drains its own XCD's queue first and
//     then steals from the others), so there is no per-wave work assignment to go wrong;
//   * every lane owns one pixel at a time: {need pixel, need candidates, traversing, wait for shading};
//   * traversal advances in flat trips: in one trip a lane may enter its next candidate instance,
//     visit one inner node and test one leaf; no lane waits for another lane's descent;
//   * shading, ray generation, the candidate-instance mask and pixel refill run in a service section
//     entered when no lane is traversing or at least CRT_SERVICE_LANES lanes are waiting, so that the
//     long, branchy shading code always runs with many lanes.
// Wavefront ballots/popcounts drive every decision (which section runs, which lane takes which pending
// pixel); the traversal stack stays in LDS as in the tile kernel.
#pragma once
#include "crt_device.h"

#ifndef CRT_SERVICE_LANES
#define CRT_SERVICE_LANES 16
#endif

enum { CRT_ST_NEED_PIXEL = 0, CRT_ST_TRAVERSE = 1, CRT_ST_WAIT_SHADE = 2, CRT_ST_IDLE = 3, CRT_ST_NEED_CAND = 4 };

// position of the (r+1)-th set bit of m (r < popcount(m))
__device__ __forceinline__ uint32_t select_bit(unsigned long long m, uint32_t r)
{
    uint32_t pos = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const unsigned long long part = (m >> pos) & ((w == 32) ? 0xFFFFFFFFull : ((1ull << w) - 1ull));
        const uint32_t c = (uint32_t)__popcll(part);
        if (r >= c) { pos += (uint32_t)w; r -= c; }
    }
    return pos;
}

struct CrtQueues { uint32_t next[8]; };   // per-XCD tile counters, zeroed before every launch

template <bool COUNT, bool DIAG = false>
__global__ __launch_bounds__(CRT_BLOCK, 4) void crt_trace_persistent_kernel(CrtDevScene S, CrtFrame F, float4* __restrict__ out,
                                                                           unsigned long long* __restrict__ counters,
                                                                           CrtQueues* __restrict__ queues)
{
    __shared__ uint32_t s_stack[CRT_STACK_DEPTH * CRT_BLOCK];
    const CrtStack stack = { (crt_lds_u32_ptr)s_stack + threadIdx.x, S.stackOverflow };
    const uint32_t lane = threadIdx.x;
    LaneCounters lc; zero_counters(lc);
    // DIAG builds: lane 0 keeps {service passes, lanes served, inner trips, inner lanes, leaf trips, leaf lanes, enter trips, enter lanes}
    uint32_t dg[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long t0rt = 0, t0c = 0;
    if (DIAG) { t0rt = __builtin_amdgcn_s_memrealtime(); t0c = __builtin_amdgcn_s_memtime(); }

    // ---- wave-uniform work-queue state ----
    const int homeQ = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);   // HW_REG_XCC_ID
    int stolen = 0;                         // queues already found empty (home + 0 .. home + stolen - 1)
    unsigned long long pend = 0;            // not yet assigned pixels (Morton positions) of the pending tile
    int pendX = 0, pendY = 0;               // its pixel origin
    const int lx = (int)((lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4));
    const int ly = (int)(((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4));

    // ---- per-lane pixel / path state ----
    int state = CRT_ST_NEED_PIXEL;
    uint32_t pix = 0;                        // px | py << 16
    int bounce = 0;
    PathState ps;
    ps.o = mk3(0.f, 0.f, 0.f); ps.d = ps.o; ps.result = ps.o; ps.energy = 0.f;
    Closest c; c.distance = 99999.0f; c.hitInstance = 0; c.anyHit = 0; c.hit.t = 0.f; c.hit.u = 0.f; c.hit.v = 0.f; c.hit.tri = 0;
    // ---- per-lane traversal state (shared with the tile kernel: crt_device.h) ----
    uint32_t base = 0; unsigned long long cand = 0;
    Traversal<COUNT> T; T.reset();

    for (;;) {
        // =====================================================================================
        // service section (wave-uniform entry): shading, pixel refill, ray generation, candidates
        // =====================================================================================
        const uint32_t nTrav = (uint32_t)__popcll(__ballot(state == CRT_ST_TRAVERSE));
        const uint32_t nIdle = (uint32_t)__popcll(__ballot(state == CRT_ST_IDLE));
        if (nTrav == 0 || (64u - nTrav - nIdle) >= (uint32_t)CRT_SERVICE_LANES) {
            for (;;) {
                if (DIAG) { const uint32_t ns = (uint32_t)__popcll(__ballot(state == CRT_ST_WAIT_SHADE || state == CRT_ST_NEED_CAND || state == CRT_ST_NEED_PIXEL)); if (lane == 0) { dg[0]++; dg[1] += ns; } }
                // 1. shade finished rays (kernel_main.cl:219-271); two bounces at most (kernel_main.cl:187)
                if (state == CRT_ST_WAIT_SHADE) {
                    const bool cont = shade_bounce(S, c, ps, bounce, F.lightY, F.lightZ);
                    if (COUNT) { if (cont) lc.hits++; else lc.misses++; }
                    if (cont && bounce == 0) {
                        bounce = 1;
                        c.distance = 99999.0f; c.hitInstance = 0; c.anyHit = 0;
                        base = 0; state = CRT_ST_NEED_CAND;
                        if (COUNT) { lc.rays++; lc.secondary++; }
                    } else {
                        out[(size_t)(pix >> 16) * (size_t)F.width + (size_t)(pix & 0xFFFFu)] = make_float4(ps.result.x, ps.result.y, ps.result.z, 1.0f);
                        state = CRT_ST_NEED_PIXEL;
                    }
                }
                // 2. hand pending pixels to free lanes; fetch tiles while lanes are still free
                unsigned long long freeMask = __ballot(state == CRT_ST_NEED_PIXEL);
                while (freeMask != 0) {
                    if (pend == 0) {
                        // next tile: own XCD's queue first, then steal (wave-uniform; one atomic per fetch)
                        bool got = false;
                        while (!got && stolen < 8) {
                            const int q = (homeQ + stolen) & 7;
                            const int rowsQ = (F.ownedTileRows - q + 7) >> 3;       // owned tile rows k with k % 8 == q
                            const uint32_t lenQ = rowsQ > 0 ? (uint32_t)rowsQ * (uint32_t)F.tilesX : 0u;
                            uint32_t t = 0;
                            if (lenQ != 0) {
                                if (lane == 0) t = atomicAdd(&queues->next[q], 1u);
                                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                            }
                            if (lenQ != 0 && t < lenQ) {
                                const int r = (int)(t / (uint32_t)F.tilesX);
                                const int tx = (int)(t - (uint32_t)r * (uint32_t)F.tilesX);
                                const int k = q + 8 * r;
                                const int bandK = k / F.tileRowsPerBand;
                                const int tileRow = (F.rank + bandK * F.nRanks) * F.tileRowsPerBand + (k - bandK * F.tileRowsPerBand);
                                pendX = tx * CRT_TILE; pendY = tileRow * CRT_TILE;
                                pend = __ballot(pendX + lx < F.width && pendY + ly < F.height);
                                got = pend != 0;
                            } else {
                                ++stolen;
                            }
                        }
                        if (!got) break;    // every queue is empty
                    }
                    // the r-th free lane takes the r-th pending pixel
                    const uint32_t nAssign = min((uint32_t)__popcll(freeMask), (uint32_t)__popcll(pend));
                    const uint32_t rank = (uint32_t)__popcll(freeMask & ((1ull << lane) - 1ull));
                    const bool take = (state == CRT_ST_NEED_PIXEL) && rank < nAssign;
                    uint32_t pos = 0;
                    if (take) {
                        pos = select_bit(pend, rank);
                        const int px = pendX + (int)((pos & 1) | ((pos >> 1) & 2) | ((pos >> 2) & 4));
                        const int py = pendY + (int)(((pos >> 1) & 1) | ((pos >> 2) & 2) | ((pos >> 3) & 4));
                        pix = (uint32_t)px | ((uint32_t)py << 16);
                        // kernel_main.cl:179-185 + RayGen (kernel_main.cl:277-287)
                        ps.o = mk3(F.camPos[0], F.camPos[1], F.camPos[2]);
                        ps.d = raygen_dir(F, px, py);
                        ps.result = mk3(0.0f, 0.0f, 0.0f);
                        ps.energy = 1.0f;
                        bounce = 0;
                        c.distance = 99999.0f; c.hitInstance = 0; c.anyHit = 0;
                        base = 0; state = CRT_ST_NEED_CAND;
                        if (COUNT) { lc.rays++; lc.primary++; }
                    }
                    const unsigned long long taken = __ballot(take);
                    // clear the assigned pixels: the lowest nAssign set bits of pend
                    unsigned long long p2 = pend;
                    for (uint32_t i = 0; i < nAssign; ++i) p2 &= p2 - 1;
                    pend = p2;
                    freeMask &= ~taken;
                }
                if (state == CRT_ST_NEED_PIXEL) state = CRT_ST_IDLE;       // no work left for this lane
                // 3. candidate instances of the current 64-instance chunk (conservative sphere test, crt_device.h)
                const unsigned long long needCand = __ballot(state == CRT_ST_NEED_CAND);
                if (needCand != 0) {
                    // chunks are visited in ascending order; lanes on different chunks take turns
                    uint32_t myBase = (state == CRT_ST_NEED_CAND) ? base : 0xFFFFFFFFu;
                    uint32_t ubase = myBase;
                    for (int off = 32; off > 0; off >>= 1) { const uint32_t o2 = (uint32_t)__shfl_xor((int)ubase, off, 64); ubase = o2 < ubase ? o2 : ubase; }
                    ubase = (uint32_t)__builtin_amdgcn_readfirstlane((int)ubase);
                    if (state == CRT_ST_NEED_CAND && base == ubase) {
                        const uint32_t cnt = (S.numInstances - ubase) < 64u ? (S.numInstances - ubase) : 64u;
                        const unsigned long long m = ubase < S.numInstances ? candidate_mask<COUNT>(S, ps.o, ps.d, ubase, cnt, lc) : 0ull;
                        cand = m;
                        if (m != 0) { state = CRT_ST_TRAVERSE; T.active = false; }
                        else if (ubase + 64u < S.numInstances) base = ubase + 64u;           // stays NEED_CAND
                        else state = CRT_ST_WAIT_SHADE;                                     // nothing (more) to traverse
                    }
                }
                // repeat while some lane can still be served here
                const unsigned long long again = __ballot(state == CRT_ST_WAIT_SHADE || state == CRT_ST_NEED_CAND);
                if (again == 0) break;
            }
        }
        if (__ballot(state != CRT_ST_IDLE) == 0) break;

        // =====================================================================================
        // one traversal trip (kernel_main.cl:124-160, 198-217). Exactly ONE of the three step kinds
        // runs per trip -- the one most lanes are waiting for -- so that each section's code is issued
        // for many lanes: lanes pile up at the rarer steps (leaf, enter) while the common step (inner
        // node) keeps running, and are then served together.
        // =====================================================================================
        const bool trav = state == CRT_ST_TRAVERSE;
        const bool wEnter = trav && !T.active;
        const bool wInner = trav && T.at_inner();
        const bool wLeaf = trav && T.at_leaf();
        const uint32_t nE = (uint32_t)__popcll(__ballot(wEnter)), nI = (uint32_t)__popcll(__ballot(wInner)), nL = (uint32_t)__popcll(__ballot(wLeaf));
        if (nI > 0 && nI >= nE && nI >= nL) {
            if (DIAG) { if (lane == 0) { dg[2]++; dg[3] += nI; } }
            if (wInner) T.inner(S, GlobalPairLoader(), stack, c, lc);
        } else if (nL > 0 && nL >= nE) {
            if (DIAG) { if (lane == 0) { dg[4]++; dg[5] += nL; } }
            if (wLeaf) T.leaf(S, stack, c, lc);
        } else if (nE > 0) {
            if (DIAG) { if (lane == 0) { dg[6]++; dg[7] += nE; } }
            if (wEnter) {
                if (cand == 0) {
                    if (base + 64u < S.numInstances) { base += 64u; state = CRT_ST_NEED_CAND; }
                    else state = CRT_ST_WAIT_SHADE;
                } else {
                    const uint32_t k = (uint32_t)__ffsll((long long)cand) - 1u;
                    cand &= cand - 1;
                    T.enter(S, base + k, ps.o, ps.d, c.distance, lc);
                }
            }
        }
    }
    if (COUNT) flush_counters(lc, counters);
    if (DIAG) {
        const unsigned long long t1c = __builtin_amdgcn_s_memtime(), t1rt = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* st = counters + 16 + (size_t)blockIdx.x * 8;
            st[0] = t0rt; st[1] = t1rt; st[2] = t1c - t0c; st[3] = (unsigned long long)homeQ;
            st[4] = ((unsigned long long)dg[0] << 32) | dg[1]; st[5] = ((unsigned long long)dg[2] << 32) | dg[3];
            st[6] = ((unsigned long long)dg[4] << 32) | dg[5]; st[7] = ((unsigned long long)dg[6] << 32) | dg[7];
        }
    }
}
