/* crt_oracle.h -- CPU oracle for the CLRayTracer per-pixel ray-trace path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under clraytracer_amd/ (the product) may include, link or
 * call this. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer vectors for this
 * path, its host C++ is MSVC/Win32-only and its OpenCL kernels cannot run in this image (no
 * OpenCL device), so this restatement is pinned by reading the reference source, not by outputs
 * of the reference. See DESIGN.md "Oracle".
 *
 * What it restates (all citations relative to the upstream tree, CLRayTracer/...):
 *   kernels/kernel_main.cl:84-160   IntersectTriangle / IntersectAABB / IntersectBVH
 *   kernels/kernel_main.cl:164-275  kernel Trace
 *   kernels/kernel_main.cl:277-287  kernel RayGen
 *   kernels/kernel_main.cl:342-359  kernel PostProcess (+ MathAndSTL.cl:132-169)
 *   kernels/MathAndSTL.cl:100-119, 243-266  MatMul / Mat3Mul / reflect / colour / texture sampling
 *   BVH.cpp:9-255                   aabb, UpdateNodeBounds, FindBestSplitPlane, SubdivideBVH, BuildBVH
 *   CPURayTrace.cpp:186-249         CPU_RayCast
 *   Math/Math.hpp:154-201           half <-> float
 *   Math/Matrix.hpp:211-250,292-374 LookAtRH, PerspectiveFovRH, InverseTransform, Inverse
 *
 * Pinned builtin semantics (the OpenCL built-ins the reference calls are implementation-defined
 * at the ULP level; the oracle fixes them as follows, and the HIP path follows the same spec):
 *   - all arithmetic fp32, no FMA contraction, IEEE division and sqrt
 *   - native_recip(x)  := 1.0f / x
 *   - dot(a,b)         := (a.x*b.x + a.y*b.y) + a.z*b.z
 *   - cross(a,b)       := (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
 *   - normalize(v)     := v * (1.0f / sqrtf(dot(v,v)))
 *   - fmin/fmax        := IEEE minNum/maxNum (C fminf/fmaxf)
 *   - atan2pi(y,x)     := (float)(atan2((double)y,(double)x) / pi)
 *   - acospi(x)        := (float)(acos((double)x) / pi)
 *   - sin/cos(x)       := (float)sin((double)x), (float)cos((double)x)
 *   - pow(x,y)         := powf(x,y)  (pow(x,1.0f) == x)
 *   - (int)f           := truncation, NaN -> 0, saturating
 *   - uninitialised Triout.u/v (kernel_main.cl:200-202) := 0.0f
 *   - traversal stack beyond 32 entries (UB upstream) := slot index wraps modulo 32
 *   - texel index outside the pool (UB upstream, e.g. negative skybox theta at phi=0) := clamped
 */
#ifndef CRT_ORACLE_H
#define CRT_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/crt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcScene {
    const CrtTri* tris;
    const CrtBVHNode* nodes;
    const uint32_t* roots;          /* bvhIndices[meshIndex] */
    const CrtMaterial* materials;
    const CrtTexture* textures;
    const CrtRGB8* texels;
    int64_t numTexels;              /* pool size for the index clamp */
    const CrtMeshInstance* instances;
    uint32_t numInstances;
} OrcScene;

/* Work counters (layout-independent) used for the algorithmic-bytes roofline. */
typedef struct OrcStats {
    uint64_t rays;                  /* primary + traced secondary */
    uint64_t primary, secondary;
    uint64_t hits, misses;          /* per traced ray */
    uint64_t traversals;            /* (ray, instance) pairs */
    uint64_t pops;                  /* while-loop iterations, kernel_main.cl:131 */
    uint64_t innerVisits;           /* child-pair fetches, kernel_main.cl:144-145 */
    uint64_t triTests;              /* kernel_main.cl:138 */
    uint64_t capHits;               /* traversals stopped by the 250-pop cap */
    uint64_t stackOverflows;        /* pushes beyond 32 entries */
    uint64_t maxStack;
    uint64_t shadowRays, shadowHits; /* orc_trace_ex(shadows=1): shadow rays traced (also counted in `rays`) / occluded */
} OrcStats;

/* ---- scalar helpers / KATs ---- */
uint16_t orc_float_to_half(float v);                 /* Math.hpp:190-197 (round-half-up bit trick) */
float    orc_half_to_float(uint16_t h);              /* IEEE binary16 -> binary32 (vload_half) */
float    orc_half_to_float_ref(uint16_t h);          /* Math.hpp:156-164 (bit-hack variant) */
int      orc_intersect_triangle(const float o[3], const float d[3], const float x[3], const float y[3],
                                const float z[3], float tuv[3], uint32_t* triIndex, int i);
float    orc_intersect_aabb(const float o[3], const float invDir[3], const float bmin[3],
                            const float bmax[3], float minSoFar);
int      orc_sample_texture(const CrtTexture* tex, float u, float v);
int      orc_sample_skybox(const float d[3], const CrtTexture* tex);
void     orc_multiply_color(const uint8_t rgb[3], uint32_t color, float out[3]);

/* ---- matrices ---- */
void orc_inverse_transform(const float in[16], float out[16]);
void orc_inverse(const float in[16], float out[16]);
void orc_perspective_fov_rh(float fovRad, float width, float height, float zNear, float zFar, float out[16]);
void orc_look_at_rh(const float eye[3], const float front[3], const float up[3], float out[16]);

/* ---- BVH build (BVH.cpp:218-255) ----
 * `nodeCounter` is the file-static `totalNodesUsed` (BVH.cpp:49) made explicit; nodes are
 * indexed from the `nodes` pointer exactly as upstream. Returns the number of nodes added. */
uint32_t orc_build_bvh(CrtTri* tris, const uint32_t* meshTriCounts, int numMeshes,
                       CrtBVHNode* nodes, uint32_t* roots, uint32_t* nodeCounter);

/* ---- kernels ---- */
void orc_raygen(float* rays, int width, int height, const float invView[16], const float invProj[16]);
/* Trace rows [row0,row1) of a width x height frame; `rays` is the full-frame ray buffer;
 * `out` is the full-frame RGBA float buffer (only the requested rows are written). */
void orc_trace(const OrcScene* s, const CrtTraceArgs* args, const float* rays, int width, int height,
               int row0, int row1, float* out, OrcStats* stats, int nthreads);
/* orc_trace plus the shadow-ray EXTENSION (shadows != 0). Upstream has no shadow ray -- kernel_main.cl:256-258 is a
 * commented-out TODO with `shadow = 1.0f` threaded into kernel_main.cl:264 -- so this mode has no reference
 * behaviour to match; it is defined here (origin = the bounce ray's origin, direction = -lightDir, any-hit over the
 * same instance loop, traced at the first bounce when n.l > 0) and the HIP path must match it bit for bit. */
/* `extensions` = ORC_EXT_SHADOWS | ORC_EXT_REFRACTION (0 = upstream's Trace). ORC_EXT_REFRACTION: the other README TODO
 * of upstream ("refraction", "transculency"), equally without reference behaviour and equally defined here: at the first
 * hit of a material whose MTL `d` (opacity, kept in Material::roughness by the importer) is below 1, the continuing ray
 * is the refracted ray (Snell, index 1.5; total internal reflection keeps the reflected ray), starting 0.01 behind the
 * surface and carrying (1 - opacity) of the energy; no shadow ray is traced for such a hit (its factor is not used). */
#define ORC_EXT_SHADOWS 1
#define ORC_EXT_REFRACTION 2
#define ORC_EXT_PRIMARY_ONLY 4      /* analysis only: stop after the first bounce (per-pixel cost split, orc_trace_costs_ex) */
void orc_trace_ex(const OrcScene* s, const CrtTraceArgs* args, const float* rays, int width, int height,
                  int row0, int row1, float* out, OrcStats* stats, int nthreads, int extensions);
void orc_postprocess(float* rgba, int width, int height, int row0, int row1);
/* EXTENSION, parity unpinned: upstream's dead FXAA function (kernel_main.cl:289-340; call commented out at :349) made
 * runnable -- result returned, neighbours read from the unmodified `src`, reads clamped to the edge. Runs BEFORE
 * orc_postprocess in the chain upstream sketches (FXAA -> Saturation -> Reinhard -> Gamma -> Vignette). */
void orc_fxaa(const float* src_rgba, float* dst_rgba, int width, int height, int row0, int row1);
/* Hazard H8: the store + load through upstream's RGBA8-UNORM render target (write_imagef / read_imagef), in place on a
 * float frame; and the bytes themselves. Upstream's displayed frame = pack(postprocess(quantize(trace))). */
void orc_quantize_unorm8(float* rgba, int width, int height, int row0, int row1);
void orc_pack_unorm8(const float* rgba, uint8_t* out, int width, int height, int row0, int row1);
/* Analysis helper: per-pixel inner visits / triangle tests (both bounces) of a full frame. */
void orc_trace_costs(const OrcScene* s, const CrtTraceArgs* args, const float* rays, int width, int height,
                     uint32_t* innerOut, uint32_t* triOut, int nthreads);
void orc_trace_costs_ex(const OrcScene* s, const CrtTraceArgs* args, const float* rays, int width, int height,
                        uint32_t* innerOut, uint32_t* triOut, int nthreads, int extensions);
/* Analysis hook: count child-pair fetches per left-child index (NULL disables). */
void orc_set_visit_counts(uint32_t* counts);
/* Closest-hit query for arbitrary world-space rays: the instance loop of kernel_main.cl:198-217. */
void orc_closest_hits(const OrcScene* s, const float* origins, const float* dirs, int n,
                      CrtRayHit* out, OrcStats* stats, int nthreads);
/* The same, also flagging (capped[k] = 1) the rays whose traversal hit the 250-pop cap; `capped` may be NULL. */
void orc_closest_hits_ex(const OrcScene* s, const float* origins, const float* dirs, int n,
                         CrtRayHit* out, OrcStats* stats, int nthreads, uint8_t* capped);
/* CPU_RayCast (CPURayTrace.cpp:186-249), with _mm_rcp_ps pinned to IEEE 1/x. */
void orc_cpu_raycast(const OrcScene* s, const float* origins, const float* dirs, int n,
                     CrtHitRecord* out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
