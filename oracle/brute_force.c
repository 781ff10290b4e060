/* brute_force.c -- an all-triangles nearest-hit search, independent of the BVH and of the traversal code.
 *
 * TEST INFRASTRUCTURE ONLY (tests/test_brute_force.py). The oracle (crt_oracle.c) restates upstream's IntersectBVH
 * (kernel_main.cl:124-160); nothing upstream-held pins it (the reference ships no tests), so this file gives it an
 * independent cross-check: for every ray it runs the instance loop of kernel_main.cl:198-217 -- same ray transform, same
 * running `t` carried from instance to instance -- but tests EVERY triangle of the instance's mesh in index order with
 * its own Moeller-Trumbore evaluation (kernel_main.cl:84-106's formula, same operation order so `t,u,v` are comparable
 * bit for bit), never looking at a BVH node. Where the tree cannot hide a triangle (ray origin outside the mesh's root
 * box so that hazard H1 cannot bite, no 250-pop cap hit, no zero-thickness box on the path to the leaf) the oracle's
 * traversal must return exactly this search's winner.
 * Output per ray: the winner, plus `ties` = number of OTHER triangles of the winning instance whose t equals the
 * winner's t exactly (then the winner depends on test order, which legitimately differs between a tree and a list).
 */
#include <stdint.h>
#include <string.h>
#include "../include/crt_types.h"

typedef struct { float t, u, v; uint32_t triIndex; int32_t instance; uint32_t ties; } BruteHit;

static inline float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline void cross3(const float a[3], const float b[3], float r[3])
{
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}
/* MathAndSTL.cl:100-102 with a row-major matrix: ((m.x*v.x + m.y*v.y) + m.z*v.z) + m.w*v.w, xyz only */
static inline void xform(const CrtMatrix4* m, const float v[3], float w, float out[3])
{
    for (int c = 0; c < 3; ++c) out[c] = ((m->m[0][c] * v[0] + m->m[1][c] * v[1]) + m->m[2][c] * v[2]) + m->m[3][c] * w;
}

void brute_force_hits(const CrtTri* tris, const CrtMeshInstance* instances, uint32_t numInstances,
                      const uint32_t* meshTriStart, const uint32_t* meshTriCount,
                      const float* origins, const float* dirs, int n, BruteHit* out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
    for (int k = 0; k < n; ++k) {
        BruteHit best; memset(&best, 0, sizeof best);
        best.t = 99999.0f; best.instance = -1;
        for (uint32_t i = 0; i < numInstances; ++i) {
            const CrtMeshInstance* inst = instances + i;
            float o[3], d[3];
            xform(&inst->inverseTransform, origins + 3 * k, 1.0f, o);
            xform(&inst->inverseTransform, dirs + 3 * k, 0.0f, d);
            float runT = best.t, runU = 0.0f, runV = 0.0f; uint32_t runTri = 0, ties = 0; int found = 0;
            const uint32_t first = meshTriStart[inst->meshIndex], count = meshTriCount[inst->meshIndex];
            for (uint32_t j = first; j < first + count; ++j) {
                const CrtTri* tr = tris + j;
                float e1[3], e2[3], h[3], s[3], q[3];
                for (int c = 0; c < 3; ++c) { e1[c] = tr->v1[c] - tr->v0[c]; e2[c] = tr->v2[c] - tr->v0[c]; }
                cross3(d, e2, h);
                const float a = dot3(e1, h);
                const float f = 1.0f / a;
                for (int c = 0; c < 3; ++c) s[c] = o[c] - tr->v0[c];
                const float u = f * dot3(s, h);
                cross3(s, e1, q);
                const float v = f * dot3(d, q);
                const float t = f * dot3(e2, q);
                if (!(u < 0.0f) && !(u > 1.0f) && !(v < 0.0f) && !(u + v > 1.0f) && t > 0.0f) {
                    if (t < runT) { runT = t; runU = u; runV = v; runTri = j; found = 1; ties = 0; }
                    else if (found && t == runT) ++ties;
                }
            }
            if (found) { best.t = runT; best.u = runU; best.v = runV; best.triIndex = runTri; best.instance = (int32_t)i; best.ties = ties; }
        }
        out[k] = best;
    }
}
