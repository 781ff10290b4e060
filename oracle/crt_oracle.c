/* crt_oracle.c -- CPU restatement of the CLRayTracer ray-trace path. See crt_oracle.h.
 *
 * TEST INFRASTRUCTURE ONLY (never linked into the product). PARITY UNPINNED (see header).
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * All arithmetic is scalar fp32 in the exact operation order of the reference source.
 */
#include "crt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * small float3 helpers with the pinned operation order
 * ---------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } f3;

static inline f3 f3_make(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_mul(f3 a, f3 b) { return f3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 f3_scale(f3 a, float s) { return f3_make(a.x * s, a.y * s, a.z * s); }
static inline f3 f3_neg(f3 a) { return f3_make(-a.x, -a.y, -a.z); }
static inline float f3_dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline f3 f3_cross(f3 a, f3 b)
{
    return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* ---- Implementation-defined builtins -------------------------------------------------------------------------------------
 * Default: the pinned semantics of crt_oracle.h (what the HIP path is held to, bit for bit).
 * ORC_ALT_BUILTINS (oracle/Makefile target `sensitivity`, used ONLY by tests/test_oracle_sensitivity.py): the OTHER choices a
 * conforming OpenCL implementation may make for the same source -- native_recip / the reciprocal square root inside normalize off
 * by one ulp either way (a hardware estimate), single-precision atan2 / acos / sin / cos instead of correctly rounded ones -- so that
 * the distance between two legal readings of the reference can be MEASURED (what "parity unpinned" means in
 * pixels). Together with -ffp-contract=fast -mfma (OpenCL C's default FP_CONTRACT ON lets the compiler fuse a*b+c) that is the
 * "alt" library. Never linked into the product, never used as the checker of the HIP path. */
#ifdef ORC_ALT_BUILTINS
static inline float orc_wobble(float r)
{   /* +-1 ulp, the direction taken from the value's own bits: deterministic, unbiased */
    uint32_t u; memcpy(&u, &r, 4);
    if ((u & 0x7F800000u) == 0x7F800000u || (u & 0x7FFFFFFFu) == 0u) return r;      /* inf, NaN, zero: as they are */
    uint32_t h = u * 2654435761u; h ^= h >> 15;
    if ((h & 3u) == 0u) return r;                                                      /* a quarter of the values come out exact */
    u = (h & 4u) ? u + 1u : u - 1u;
    float q; memcpy(&q, &u, 4); return q;
}
#define ORC_RECIP(x) orc_wobble(1.0f / (x))
#define ORC_RSQRT(x) orc_wobble((float)(1.0 / sqrt((double)(x))))
#define ORC_ATAN2PI(y, x) (atan2f((y), (x)) / 3.14159265f)
#define ORC_ACOSPI(x) (acosf(x) / 3.14159265f)
#define ORC_SIN(x) sinf(x)
#define ORC_COS(x) cosf(x)
#else
#define ORC_RECIP(x) (1.0f / (x))
#define ORC_RSQRT(x) (1.0f / sqrtf(x))
#define ORC_ATAN2PI(y, x) ((float)(atan2((double)(y), (double)(x)) / ORC_PI))
#define ORC_ACOSPI(x) ((float)(acos((double)(x)) / ORC_PI))
#define ORC_SIN(x) ((float)sin((double)(x)))
#define ORC_COS(x) ((float)cos((double)(x)))
#endif
static inline f3 f3_normalize(f3 v)
{
    float inv = ORC_RSQRT(f3_dot(v, v));
    return f3_scale(v, inv);
}
/* MathAndSTL.cl:117-119  v - n * dot(n, v) * 2.0f */
static inline f3 f3_reflect(f3 v, f3 n)
{
    float d = f3_dot(n, v);
    return f3_sub(v, f3_scale(f3_scale(n, d), 2.0f));
}

/* (int) conversion pinned: truncation, NaN -> 0, saturating (what v_cvt_i32_f32 does). */
static inline int32_t f2i(float x)
{
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}

static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------------------------------
 * half <-> float
 * ---------------------------------------------------------------------------------------- */
/* Math.hpp:190-197 */
uint16_t orc_float_to_half(float Value)
{
    const uint32_t b = f2bits(Value) + 0x00001000u;
    const uint32_t e = (b & 0x7F800000u) >> 23;
    const uint32_t m = b & 0x007FFFFFu;
    uint32_t a = (b & 0x80000000u) >> 16 | (uint32_t)(e > 112) * ((((e - 112) << 10) & 0x7C00u) | m >> 13);
    /* shift count 125-e is only meaningful when the guard is true; keep it defined otherwise */
    uint32_t den = 0;
    if ((e < 113) & (e > 101)) den = (((0x007FF000u + m) >> (125 - e)) + 1) >> 1;
    return (uint16_t)(a | den | (uint32_t)(e > 143) * 0x7FFFu);
}

/* Math.hpp:156-164 */
float orc_half_to_float_ref(uint16_t x)
{
    const uint32_t e = (x & 0x7C00u) >> 10;
    const uint32_t m = (x & 0x03FFu) << 13;
    const uint32_t v = f2bits((float)m) >> 23;
    uint32_t a = (uint32_t)(x & 0x8000u) << 16 | (uint32_t)(e != 0) * ((e + 112) << 23 | m);
    if ((e == 0) & (m != 0)) a |= ((v - 37) << 23 | ((m << (150 - v)) & 0x007FE000u));
    return bits2f(a);
}

/* IEEE binary16 -> binary32, what vload_half does on the device (kernel_main.cl:232-240). */
float orc_half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu;
    uint32_t m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return bits2f(sign);
        /* subnormal: value = m * 2^-24 */
        float f = (float)m * (1.0f / 16777216.0f);
        return bits2f(f2bits(f) | sign);
    }
    if (e == 31) return bits2f(sign | 0x7F800000u | (m << 13));
    return bits2f(sign | ((e + 112) << 23) | (m << 13));
}

/* ------------------------------------------------------------------------------------------
 * kernel_main.cl:84-117  triangle and box tests
 * ---------------------------------------------------------------------------------------- */
typedef struct { float t, u, v; uint32_t triIndex; } Triout;
typedef struct { f3 origin, direction; } Ray;

static inline int intersect_triangle(Ray ray, const CrtTri* tri, Triout* o, int i)
{
    const f3 tx = f3_make(tri->v0[0], tri->v0[1], tri->v0[2]);
    const f3 ty = f3_make(tri->v1[0], tri->v1[1], tri->v1[2]);
    const f3 tz = f3_make(tri->v2[0], tri->v2[1], tri->v2[2]);
    const f3 edge1 = f3_sub(ty, tx);
    const f3 edge2 = f3_sub(tz, tx);
    const f3 h = f3_cross(ray.direction, edge2);
    const float a = f3_dot(edge1, h);
    const float f = ORC_RECIP(a);
    const f3 s = f3_sub(ray.origin, tx);
    const float u = f * f3_dot(s, h);
    const f3 q = f3_cross(s, edge1);
    const float v = f * f3_dot(ray.direction, q);
    const float t = f * f3_dot(edge2, q);
    int passed = (((t > 0.0000f) ^ (t < o->t)) + (u < 0.0f) + (u > 1.0f) + (v < 0.0f) + (u + v > 1.0f)) == 0;
    int notPassed = 1 - passed;
    /* arithmetic blend kept as upstream: int -> float conversion then multiply-add (NaN/inf propagate) */
    o->u = u * (float)passed + ((float)notPassed * o->u);
    o->v = v * (float)passed + ((float)notPassed * o->v);
    o->t = t * (float)passed + ((float)notPassed * o->t);
    o->triIndex = (uint32_t)i * (uint32_t)passed + ((uint32_t)notPassed * o->triIndex);
    return passed;
}

static inline float intersect_aabb(f3 origin, f3 invDir, f3 bmin, f3 bmax, float minSoFar)
{
    f3 tmin = f3_mul(f3_sub(bmin, origin), invDir);
    f3 tmax = f3_mul(f3_sub(bmax, origin), invDir);
    float tnear = fmaxf(fmaxf(fminf(tmin.x, tmax.x), fminf(tmin.y, tmax.y)), fminf(tmin.z, tmax.z));
    float tfar  = fminf(fminf(fmaxf(tmin.x, tmax.x), fmaxf(tmin.y, tmax.y)), fmaxf(tmin.z, tmax.z));
    if (tnear < tfar && tnear > 0.0f && tnear < minSoFar) return tnear;
    return 1e30f;
}

int orc_intersect_triangle(const float o[3], const float d[3], const float x[3], const float y[3],
                           const float z[3], float tuv[3], uint32_t* triIndex, int i)
{
    CrtTri tri; memset(&tri, 0, sizeof tri);
    memcpy(tri.v0, x, 12); memcpy(tri.v1, y, 12); memcpy(tri.v2, z, 12);
    Ray r = { f3_make(o[0], o[1], o[2]), f3_make(d[0], d[1], d[2]) };
    Triout out = { tuv[0], tuv[1], tuv[2], *triIndex };
    int p = intersect_triangle(r, &tri, &out, i);
    tuv[0] = out.t; tuv[1] = out.u; tuv[2] = out.v; *triIndex = out.triIndex;
    return p;
}

float orc_intersect_aabb(const float o[3], const float invDir[3], const float bmin[3],
                         const float bmax[3], float minSoFar)
{
    return intersect_aabb(f3_make(o[0], o[1], o[2]), f3_make(invDir[0], invDir[1], invDir[2]),
                          f3_make(bmin[0], bmin[1], bmin[2]), f3_make(bmax[0], bmax[1], bmax[2]), minSoFar);
}

/* ------------------------------------------------------------------------------------------
 * kernel_main.cl:124-160  IntersectBVH
 * ---------------------------------------------------------------------------------------- */
/* Analysis hook: when set, every child-pair fetch increments visitCounts[leftIndex]. */
static uint32_t* g_visit_counts = NULL;
void orc_set_visit_counts(uint32_t* counts) { g_visit_counts = counts; }

/* anyHit != 0 (shadow-ray extension, orc_trace_shadows): return at the first triangle that passes. */
static int intersect_bvh_ex(Ray ray, const CrtBVHNode* nodes, uint32_t rootNode, const CrtTri* tris,
                            Triout* out, OrcStats* st, int anyHit)
{
    int32_t nodesToVisit[CRT_STACK_DEPTH];
    memset(nodesToVisit, 0, sizeof nodesToVisit);
    nodesToVisit[0] = (int32_t)rootNode;
    int currentNodeIndex = 1;
    f3 invDir = f3_make(ORC_RECIP(ray.direction.x), ORC_RECIP(ray.direction.y), ORC_RECIP(ray.direction.z));
    int intersection = 0, protection = 0;
    st->traversals++;

    while (currentNodeIndex > 0) {
        if (!(protection++ < CRT_MAX_POPS)) { st->capHits++; break; }
        st->pops++;
        --currentNodeIndex;
        const CrtBVHNode* node = nodes + nodesToVisit[currentNodeIndex & (CRT_STACK_DEPTH - 1)];
        for (;;) { /* `traverse:` label */
            if (node->triCount > 0) {
                for (int i = (int)node->leftFirst, end = i + (int)node->triCount; i < end; ++i) {
                    st->triTests++;
                    intersection |= intersect_triangle(ray, tris + i, out, i);
                    if (anyHit && intersection) return 1;
                }
                break;
            }
            uint32_t leftIndex = node->leftFirst;
            uint32_t rightIndex = leftIndex + 1;
            const CrtBVHNode* l = nodes + leftIndex;
            const CrtBVHNode* r = nodes + rightIndex;
            st->innerVisits++;
            if (g_visit_counts) {
#pragma omp atomic
                g_visit_counts[leftIndex]++;
            }
            float dist1 = intersect_aabb(ray.origin, invDir, f3_make(l->aabbMin[0], l->aabbMin[1], l->aabbMin[2]),
                                         f3_make(l->aabbMax[0], l->aabbMax[1], l->aabbMax[2]), out->t);
            float dist2 = intersect_aabb(ray.origin, invDir, f3_make(r->aabbMin[0], r->aabbMin[1], r->aabbMin[2]),
                                         f3_make(r->aabbMax[0], r->aabbMax[1], r->aabbMax[2]), out->t);
            if (dist1 > dist2) {
                float tf = dist1; dist1 = dist2; dist2 = tf;
                uint32_t tu = leftIndex; leftIndex = rightIndex; rightIndex = tu;
            }
            if (dist1 == 1e30f) break;
            node = nodes + leftIndex;
            if (dist2 != 1e30f) {
                if (currentNodeIndex >= CRT_STACK_DEPTH) st->stackOverflows++;
                nodesToVisit[currentNodeIndex & (CRT_STACK_DEPTH - 1)] = (int32_t)rightIndex;
                currentNodeIndex++;
                if ((uint64_t)currentNodeIndex > st->maxStack) st->maxStack = (uint64_t)currentNodeIndex;
            }
        }
    }
    return intersection;
}

static int intersect_bvh(Ray ray, const CrtBVHNode* nodes, uint32_t rootNode, const CrtTri* tris,
                         Triout* out, OrcStats* st)
{
    return intersect_bvh_ex(ray, nodes, rootNode, tris, out, st, 0);
}

/* ------------------------------------------------------------------------------------------
 * MathAndSTL.cl:100-106  MatMul / Mat3Mul
 * ---------------------------------------------------------------------------------------- */
static inline void matmul4(const CrtMatrix4* m, float vx, float vy, float vz, float vw, float out[4])
{
    for (int c = 0; c < 4; ++c)
        out[c] = ((m->m[0][c] * vx + m->m[1][c] * vy) + m->m[2][c] * vz) + m->m[3][c] * vw;
}

static inline f3 mat3mul(const CrtMatrix4* m, f3 v)
{
    f3 r;
    r.x = (m->m[0][0] * v.x + m->m[1][0] * v.y) + m->m[2][0] * v.z;
    r.y = (m->m[0][1] * v.x + m->m[1][1] * v.y) + m->m[2][1] * v.z;
    r.z = (m->m[0][2] * v.x + m->m[1][2] * v.y) + m->m[2][2] * v.z;
    return r;
}

/* ------------------------------------------------------------------------------------------
 * MathAndSTL.cl:243-266  colour / texture sampling
 * ---------------------------------------------------------------------------------------- */
#define UCHAR_TO_FLOAT01 (1.0f / 255.0f)
#define ORC_PI 3.14159265358979323846

static inline int64_t clamp_texel(int64_t idx, int64_t n)
{
    if (idx < 0) return 0;
    if (idx >= n) return n - 1;
    return idx;
}

int orc_sample_skybox(const float d[3], const CrtTexture* tex)
{
    float at = ORC_ATAN2PI(d[0], -d[2]);
    float ac = ORC_ACOSPI(d[1]);
    int32_t theta = f2i((at * 0.5f) * (float)tex->width);
    int32_t phi = f2i(ac * (float)tex->height);
    /* mad24(phi, width, theta + 2) */
    return (int)((uint32_t)phi * (uint32_t)tex->width + (uint32_t)(theta + 2));
}

int orc_sample_texture(const CrtTexture* tex, float u, float v)
{
    u = u - floorf(u);
    v = v - floorf(v);
    int32_t uScaled = f2i((float)tex->width * u);
    int32_t vScaled = f2i((float)tex->height * v);
    return (int)((uint32_t)vScaled * (uint32_t)tex->width + (uint32_t)tex->offset + (uint32_t)uScaled);
}

void orc_multiply_color(const uint8_t rgb[3], uint32_t a, float out[3])
{
    uint8_t r = (uint8_t)(((a & 0xffu) * rgb[0]) >> 8);
    uint8_t g = (uint8_t)((((a >> 8) & 0xffu) * rgb[1]) >> 8);
    uint8_t b = (uint8_t)((((a >> 16) & 0xffu) * rgb[2]) >> 8);
    out[0] = (float)r * UCHAR_TO_FLOAT01;
    out[1] = (float)g * UCHAR_TO_FLOAT01;
    out[2] = (float)b * UCHAR_TO_FLOAT01;
}

/* ------------------------------------------------------------------------------------------
 * kernel_main.cl:277-287  RayGen
 * ---------------------------------------------------------------------------------------- */
static inline f3 raygen_dir(int i, int j, int width, int height, const CrtMatrix4* invView, const CrtMatrix4* invProj)
{
    float cx = (float)i / (float)width, cy = (float)j / (float)height;
    cx = cx * 2.0f - 1.0f;
    cy = cy * 2.0f - 1.0f;
    float target[4];
    matmul4(invProj, cx, cy, 1.0f, 1.0f, target);
    float w = target[3];
    target[0] /= w; target[1] /= w; target[2] /= w; target[3] /= w;
    float wv[4];
    matmul4(invView, target[0], target[1], target[2], target[3], wv);
    return f3_normalize(f3_make(wv[0], wv[1], wv[2]));
}

void orc_raygen(float* rays, int width, int height, const float invView[16], const float invProj[16])
{
    CrtMatrix4 iv, ip;
    memcpy(&iv, invView, 64); memcpy(&ip, invProj, 64);
    for (int j = 0; j < height; ++j)
        for (int i = 0; i < width; ++i) {
            f3 d = raygen_dir(i, j, width, height, &iv, &ip);
            float* o = rays + 3 * ((size_t)i + (size_t)j * (size_t)width);
            o[0] = d.x; o[1] = d.y; o[2] = d.z;
        }
}

/* ------------------------------------------------------------------------------------------
 * kernel_main.cl:198-217  instance loop (shared by Trace and the closest-hit query)
 * ---------------------------------------------------------------------------------------- */
typedef struct { float distance; int hitInstanceIndex; int anyHit; Triout hitOut; Ray meshRay; } Closest;

static inline Closest closest_hit(const OrcScene* s, Ray ray, uint32_t numMeshes, OrcStats* st)
{
    Closest c;
    memset(&c, 0, sizeof c);
    c.distance = 99999.0f; /* Infinite, MathAndSTL.cl:123 */
    for (uint32_t i = 0; i < numMeshes; ++i) {
        Triout triout;
        triout.t = c.distance;
        triout.triIndex = 0;
        triout.u = 0.0f; triout.v = 0.0f; /* uninitialised upstream; pinned to 0 */
        const CrtMeshInstance* instance = s->instances + i;
        float o4[4], d4[4];
        matmul4(&instance->inverseTransform, ray.origin.x, ray.origin.y, ray.origin.z, 1.0f, o4);
        matmul4(&instance->inverseTransform, ray.direction.x, ray.direction.y, ray.direction.z, 0.0f, d4);
        Ray mRay = { f3_make(o4[0], o4[1], o4[2]), f3_make(d4[0], d4[1], d4[2]) };
        if (intersect_bvh(mRay, s->nodes, s->roots[instance->meshIndex], s->tris, &triout, st)) {
            c.hitInstanceIndex = (int)i;
            c.hitOut = triout;
            c.distance = triout.t;
            c.meshRay = mRay;
            c.anyHit = 1;
        }
    }
    return c;
}

/* Shadow-ray EXTENSION (no upstream code: kernel_main.cl:256-258 is a commented-out TODO). The instance loop of
 * kernel_main.cl:198-217 for the ray (origin, dir) with t = Infinite, stopping at the first triangle that passes. */
static int occluded(const OrcScene* s, Ray ray, uint32_t numMeshes, OrcStats* st)
{
    for (uint32_t i = 0; i < numMeshes; ++i) {
        Triout triout;
        triout.t = 99999.0f; triout.triIndex = 0; triout.u = 0.0f; triout.v = 0.0f;
        const CrtMeshInstance* instance = s->instances + i;
        float o4[4], d4[4];
        matmul4(&instance->inverseTransform, ray.origin.x, ray.origin.y, ray.origin.z, 1.0f, o4);
        matmul4(&instance->inverseTransform, ray.direction.x, ray.direction.y, ray.direction.z, 0.0f, d4);
        Ray mRay = { f3_make(o4[0], o4[1], o4[2]), f3_make(d4[0], d4[1], d4[2]) };
        if (intersect_bvh_ex(mRay, s->nodes, s->roots[instance->meshIndex], s->tris, &triout, st, 1)) return 1;
    }
    return 0;
}

static void stats_add(OrcStats* a, const OrcStats* b)
{
    a->rays += b->rays; a->primary += b->primary; a->secondary += b->secondary;
    a->hits += b->hits; a->misses += b->misses; a->traversals += b->traversals;
    a->pops += b->pops; a->innerVisits += b->innerVisits; a->triTests += b->triTests;
    a->capHits += b->capHits; a->stackOverflows += b->stackOverflows;
    a->shadowRays += b->shadowRays; a->shadowHits += b->shadowHits;
    if (b->maxStack > a->maxStack) a->maxStack = b->maxStack;
}

/* ------------------------------------------------------------------------------------------
 * kernel_main.cl:164-275  Trace, one pixel
 * ---------------------------------------------------------------------------------------- */
static void trace_pixel(const OrcScene* s, const CrtTraceArgs* args, f3 rayDir, float lightY, float lightZ,
                        float out[4], OrcStats* st, int extensions)
{
    const int shadows = extensions & ORC_EXT_SHADOWS, refraction = extensions & ORC_EXT_REFRACTION;
    Ray ray = { f3_make(args->cameraPos[0], args->cameraPos[1], args->cameraPos[2]), rayDir };
    f3 lightDir = f3_make(0.0f, lightY, lightZ);
    f3 result = f3_make(0.0f, 0.0f, 0.0f);
    f3 energy = f3_make(1.0f, 1.0f, 1.0f);
    f3 atmosphericLight = f3_scale(f3_make(0.255f, 0.25f, 0.27f), 1.0f);

    const int bounces = (extensions & ORC_EXT_PRIMARY_ONLY) ? 1 : 2;     /* analysis only (orc_trace_costs) */
    for (int numBounces = 0; numBounces < bounces; ++numBounces) {
        st->rays++;
        if (numBounces == 0) st->primary++; else st->secondary++;
        Closest c = closest_hit(s, ray, args->numMeshes, st);

        if (c.distance > 99998.0f) { /* InfMinusOne */
            st->misses++;
            const float d[3] = { ray.direction.x, ray.direction.y, ray.direction.z };
            int64_t idx = clamp_texel((int64_t)orc_sample_skybox(d, &s->textures[2]), s->numTexels);
            CrtRGB8 px = s->texels[idx];
            f3 sky = f3_scale(f3_make((float)px.r, (float)px.g, (float)px.b), UCHAR_TO_FLOAT01);
            result = f3_add(result, f3_mul(sky, energy));
            break;
        }
        st->hits++;

        const CrtMeshInstance* hitInstance = s->instances + c.hitInstanceIndex;
        const CrtMatrix4* inv = &hitInstance->inverseTransform;
        const CrtTri* tri = s->tris + c.hitOut.triIndex;
        const CrtMaterial* material = s->materials + ((uint32_t)hitInstance->materialStart + (uint32_t)tri->materialIndex);
        float bx = (1.0f - c.hitOut.u) - c.hitOut.v, by = c.hitOut.u, bz = c.hitOut.v;

        f3 n0 = mat3mul(inv, f3_make(orc_half_to_float(tri->n0[0]), orc_half_to_float(tri->n0[1]), orc_half_to_float(tri->n0[2])));
        f3 n1 = mat3mul(inv, f3_make(orc_half_to_float(tri->n1[0]), orc_half_to_float(tri->n1[1]), orc_half_to_float(tri->n1[2])));
        f3 n2 = mat3mul(inv, f3_make(orc_half_to_float(tri->n2[0]), orc_half_to_float(tri->n2[1]), orc_half_to_float(tri->n2[2])));
        f3 normal = f3_normalize(f3_add(f3_add(f3_scale(n0, bx), f3_scale(n1, by)), f3_scale(n2, bz)));

        float uvx = (orc_half_to_float(tri->uv0[0]) * bx + orc_half_to_float(tri->uv1[0]) * by) + orc_half_to_float(tri->uv2[0]) * bz;
        float uvy = (orc_half_to_float(tri->uv0[1]) * bx + orc_half_to_float(tri->uv1[1]) * by) + orc_half_to_float(tri->uv2[1]) * bz;

        int64_t pidx = clamp_texel((int64_t)orc_sample_texture(&s->textures[material->albedoTextureIndex], uvx, uvy), s->numTexels);
        CrtRGB8 pixel = s->texels[pidx];
        /* the specular texel (kernel_main.cl:243) is fetched upstream but never used */
        const uint8_t prgb[3] = { pixel.r, pixel.g, pixel.b };
        float colf[3];
        orc_multiply_color(prgb, material->color, colf);
        f3 color = f3_make(colf[0], colf[1], colf[2]);
        f3 point = f3_add(c.meshRay.origin, f3_scale(c.meshRay.direction, c.hitOut.t));

        f3 specularColor = f3_make(0.2f, 0.2f, 0.2f);
        float roughness = 0.5f;
        float shininess = 1.0f;

        /* EXTENSION (refraction != 0; upstream: a README TODO with no code). A material whose MTL `d` (dissolve = opacity,
         * stored by the importer in Material::roughness, AssetManager.cpp:152-156) is below 1 transmits: at the FIRST hit the
         * continuing ray is the refracted ray instead of the reflected one -- Snell with a fixed index 1.5 (the importer does
         * not read `Ni`), entering when the ray runs against the normal, leaving otherwise; on total internal reflection the
         * reflected ray is kept. The origin steps 0.01 THROUGH the surface, and the next bounce carries (1 - opacity) of the
         * energy instead of upstream's `specular` term. Everything else in the shading of this hit is upstream's. */
        int transmitted = 0;
        float opacity = 1.0f;
        if (refraction && numBounces == 0) {
            opacity = orc_half_to_float(material->roughness);
            if (opacity < 1.0f) {
                const float dn = f3_dot(normal, ray.direction);
                const int entering = dn < 0.0f;
                const f3 nf = entering ? normal : f3_neg(normal);
                const float cosi = entering ? (0.0f - dn) : dn;
                const float eta = entering ? 0.6666667f : 1.5f;
                const float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
                if (k >= 0.0f) {
                    const float w = eta * cosi - sqrtf(k);
                    const f3 refr = f3_add(f3_scale(ray.direction, eta), f3_scale(nf, w));
                    ray.origin = f3_sub(point, f3_scale(nf, 0.01f));
                    ray.direction = refr;
                    transmitted = 1;
                }
            }
        }
        if (!transmitted) {
            ray.origin = point;
            ray.origin = f3_add(ray.origin, f3_scale(normal, 0.01f));
            ray.direction = f3_reflect(ray.direction, normal);
        }

        float shadow = 1.0f; /* upstream: "todo shadow for only directional light" (kernel_main.cl:258) */

        float ndl = f3_dot(normal, f3_neg(lightDir));
        f3 ambient = f3_mul(f3_scale(atmosphericLight, fmaxf(0.0f - ndl, 0.1f)), color);
        ndl = fmaxf(ndl, 0.0f);
        /* EXTENSION (shadows != 0): the commented-out `shadowRay = CreateRay(ray.origin, -lightDir)` of
         * kernel_main.cl:257. `shadow` only scales `specular`, i.e. the energy of the NEXT bounce, so the ray is traced
         * only where that is observable: at the first bounce and when ndl > 0 (else the product is 0 or NaN anyway). */
        if (shadows && !transmitted && numBounces == 0 && ndl > 0.0f) {   /* a transmitted ray's energy does not use `shadow` */
            Ray shadowRay = { ray.origin, f3_neg(lightDir) };
            st->rays++; st->shadowRays++;
            if (occluded(s, shadowRay, args->numMeshes, st)) { shadow = 0.0f; st->shadowHits++; }
        }
        float sp = ((1.0f - roughness) * ndl) * shadow;
        f3 specular = f3_scale(f3_mul(f3_make(sp, sp, sp), specularColor), ndl);
        float sl = (ndl * powf(fmaxf(f3_dot(f3_reflect(f3_neg(lightDir), normal), c.meshRay.direction), 0.0f), shininess)) * 0.2f;
        f3 specularLighting = f3_make(sl, sl, sl);

        result = f3_add(result, f3_add(f3_add(f3_mul(energy, f3_scale(color, ndl)), ambient), specularLighting));
        energy = transmitted ? f3_scale(energy, 1.0f - opacity) : f3_mul(energy, specular);
        atmosphericLight = f3_scale(atmosphericLight, 0.4f);

        lightDir = ray.direction;
    }
    out[0] = result.x; out[1] = result.y; out[2] = result.z; out[3] = 1.0f;
}

void orc_trace(const OrcScene* s, const CrtTraceArgs* args, const float* rays, int width, int height,
               int row0, int row1, float* out, OrcStats* stats, int nthreads)
{
    orc_trace_ex(s, args, rays, width, height, row0, row1, out, stats, nthreads, 0);
}

void orc_trace_ex(const OrcScene* s, const CrtTraceArgs* args, const float* rays, int width, int height,
                  int row0, int row1, float* out, OrcStats* stats, int nthreads, int extensions)
{
    (void)height;
    const float lightY = ORC_SIN(args->sunAngle);
    const float lightZ = ORC_COS(args->sunAngle);
    if (nthreads < 1) nthreads = 1;
    OrcStats total; memset(&total, 0, sizeof total);
#pragma omp parallel num_threads(nthreads)
    {
        OrcStats st; memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 4)
        for (int j = row0; j < row1; ++j) {
            for (int i = 0; i < width; ++i) {
                size_t idx = (size_t)j * (size_t)width + (size_t)i;
                f3 d = f3_make(rays[3 * idx], rays[3 * idx + 1], rays[3 * idx + 2]);
                trace_pixel(s, args, d, lightY, lightZ, out + 4 * idx, &st, extensions);
            }
        }
#pragma omp critical
        stats_add(&total, &st);
    }
    if (stats) stats_add(stats, &total);
}

/* Analysis helper: per-pixel work (inner visits + triangle tests over both bounces) of a frame. */
void orc_trace_costs(const OrcScene* s, const CrtTraceArgs* args, const float* rays, int width, int height,
                     uint32_t* innerOut, uint32_t* triOut, int nthreads)
{
    orc_trace_costs_ex(s, args, rays, width, height, innerOut, triOut, nthreads, 0);
}

void orc_trace_costs_ex(const OrcScene* s, const CrtTraceArgs* args, const float* rays, int width, int height,
                        uint32_t* innerOut, uint32_t* triOut, int nthreads, int extensions)
{
    const float lightY = ORC_SIN(args->sunAngle);
    const float lightZ = ORC_COS(args->sunAngle);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int j = 0; j < height; ++j) {
        for (int i = 0; i < width; ++i) {
            size_t idx = (size_t)j * (size_t)width + (size_t)i;
            OrcStats st; memset(&st, 0, sizeof st);
            float px[4];
            f3 d = f3_make(rays[3 * idx], rays[3 * idx + 1], rays[3 * idx + 2]);
            trace_pixel(s, args, d, lightY, lightZ, px, &st, extensions);
            innerOut[idx] = (uint32_t)st.innerVisits;
            triOut[idx] = (uint32_t)st.triTests;
        }
    }
}

void orc_closest_hits(const OrcScene* s, const float* origins, const float* dirs, int n,
                      CrtRayHit* out, OrcStats* stats, int nthreads)
{
    orc_closest_hits_ex(s, origins, dirs, n, out, stats, nthreads, NULL);
}

/* As orc_closest_hits; `capped` (optional, n bytes) is set to 1 for every ray whose traversals hit the 250-pop cap
 * (kernel_main.cl:131) in some instance, i.e. whose result may depend on where the traversal was cut off. */
void orc_closest_hits_ex(const OrcScene* s, const float* origins, const float* dirs, int n,
                         CrtRayHit* out, OrcStats* stats, int nthreads, uint8_t* capped)
{
    if (nthreads < 1) nthreads = 1;
    OrcStats total; memset(&total, 0, sizeof total);
#pragma omp parallel num_threads(nthreads)
    {
        OrcStats st; memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 256)
        for (int k = 0; k < n; ++k) {
            Ray ray = { f3_make(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]),
                        f3_make(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]) };
            st.rays++;
            const uint64_t capsBefore = st.capHits;
            Closest c = closest_hit(s, ray, s->numInstances, &st);
            if (capped) capped[k] = st.capHits != capsBefore;
            CrtRayHit h;
            if (c.anyHit) {
                h.t = c.hitOut.t; h.u = c.hitOut.u; h.v = c.hitOut.v;
                h.triIndex = c.hitOut.triIndex; h.instance = c.hitInstanceIndex;
                st.hits++;
            } else {
                h.t = c.distance; h.u = 0.0f; h.v = 0.0f; h.triIndex = 0; h.instance = -1;
                st.misses++;
            }
            out[k] = h;
        }
#pragma omp critical
        stats_add(&total, &st);
    }
    if (stats) stats_add(stats, &total);
}

/* ------------------------------------------------------------------------------------------
 * kernel_main.cl:342-359 + MathAndSTL.cl:132-169  PostProcess
 * ---------------------------------------------------------------------------------------- */
void orc_postprocess(float* rgba, int width, int height, int row0, int row1)
{
    const float oneDivGamma = 1.0f / 1.2f;
    const float max_white_l = 0.8f;
    for (int j = row0; j < row1; ++j) {
        for (int i = 0; i < width; ++i) {
            float* p = rgba + 4 * ((size_t)j * (size_t)width + (size_t)i);
            float uvx = (float)i / (float)width, uvy = (float)j / (float)height;
            f3 rgb = f3_make(p[0], p[1], p[2]);
            /* Saturation(rgb, 1.2f) */
            float P = sqrtf((rgb.x * rgb.x) * 0.299f + ((rgb.y * rgb.y) * 0.587f) + ((rgb.z * rgb.z) * 0.114f));
            f3 Pv = f3_make(P, P, P);
            rgb = f3_add(Pv, f3_scale(f3_sub(rgb, Pv), 1.2f));
            /* Reinhard */
            f3 lw = f3_make(0.2126f, 0.7152f, 0.0722f);
            float l_old = f3_dot(rgb, lw);
            float numerator = l_old * (1.0f + (l_old / (max_white_l * max_white_l)));
            float l_new = numerator / (1.0f + l_old);
            float l_in = f3_dot(rgb, lw);
            rgb = f3_scale(rgb, l_new / l_in);
            float ig = 1.0f / 1.55f;
            rgb = f3_make(powf(rgb.x, ig), powf(rgb.y, ig), powf(rgb.z, ig));
            /* GammaCorrect */
            rgb = f3_make(powf(rgb.x, oneDivGamma), powf(rgb.y, oneDivGamma), powf(rgb.z, oneDivGamma));
            /* Vignette: uv *= 1 - uv.yx */
            float vx = uvx * (1.0f - uvy), vy = uvy * (1.0f - uvx);
            float vig = (vx * vy) * 15.0f;
            vig = powf(vig, 0.15f);
            rgb = f3_scale(rgb, vig);
            p[0] = rgb.x; p[1] = rgb.y; p[2] = rgb.z; p[3] = 1.0f;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * kernel_main.cl:289-340  FXAA -- EXTENSION. Upstream lists FXAA in its README (README.md:9) but the function is dead:
 * its only call is commented out (kernel_main.cl:349), it has no return statement, and it would read neighbours of an
 * image PostProcess is rewriting in place. There is therefore no reference behaviour to match; what is restated here
 * is the function's arithmetic as written, with the three gaps closed the obvious way: the result is the `rgb` it
 * assigns last; neighbours are read from the UNMODIFIED frame (`src`) and written to `dst`; reads outside the image
 * clamp to the edge (sampler-less read_imagef is undefined there). `uv` is PostProcess's p / resolution
 * (kernel_main.cl:346) -- not the pixel centre, so the linear taps sit half a pixel up-left, as upstream's would.
 * The sampler is CLK_NORMALIZED_COORDS_TRUE | CLK_FILTER_LINEAR | CLK_ADDRESS_CLAMP_TO_EDGE, evaluated as the OpenCL
 * 1.2 specification defines it (8.2: i0 = floor(u - 0.5), a = frac(u - 0.5), weights (1-a)(1-b), a(1-b), (1-a)b, ab).
 * Parity for this stage is UNPINNED (nothing of upstream's ever ran it); the HIP kernel must match this bit for bit.
 * ---------------------------------------------------------------------------------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline f3 fxaa_texel(const float* img, int width, int height, int i, int j)
{
    const float* p = img + 4 * ((size_t)clampi(j, 0, height - 1) * (size_t)width + (size_t)clampi(i, 0, width - 1));
    return f3_make(p[0], p[1], p[2]);
}
static inline f3 fxaa_linear(const float* img, int width, int height, float s, float t)
{
    const float u = s * (float)width - 0.5f, v = t * (float)height - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float a = u - fu, b = v - fv;
    const int i0 = (int)fu, j0 = (int)fv;
    const f3 t00 = fxaa_texel(img, width, height, i0, j0), t10 = fxaa_texel(img, width, height, i0 + 1, j0);
    const f3 t01 = fxaa_texel(img, width, height, i0, j0 + 1), t11 = fxaa_texel(img, width, height, i0 + 1, j0 + 1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return f3_add(f3_add(f3_add(f3_scale(t00, w00), f3_scale(t10, w10)), f3_scale(t01, w01)), f3_scale(t11, w11));
}
void orc_fxaa(const float* src, float* dst, int width, int height, int row0, int row1)
{
    const float FXAA_SPAN_MAX = 8.0f, FXAA_REDUCE_MUL = 1.0f / 8.0f, FXAA_REDUCE_MIN = 1.0f / 128.0f;
    const f3 luma = f3_make(0.299f, 0.587f, 0.114f);
    const float resx = (float)width, resy = (float)height;
    for (int j = row0; j < row1; ++j) {
        for (int i = 0; i < width; ++i) {
            const float uvx = (float)i / resx, uvy = (float)j / resy;
            const f3 rgb = fxaa_texel(src, width, height, i, j);
            /* 1st stage: find the edge */
            const float lumaNW = f3_dot(fxaa_texel(src, width, height, i - 1, j - 1), luma);
            const float lumaNE = f3_dot(fxaa_texel(src, width, height, i + 1, j - 1), luma);
            const float lumaSW = f3_dot(fxaa_texel(src, width, height, i - 1, j + 1), luma);
            const float lumaSE = f3_dot(fxaa_texel(src, width, height, i + 1, j + 1), luma);
            const float lumaM = f3_dot(rgb, luma);
            float dirx = -((lumaNW + lumaNE) - (lumaSW + lumaSE));
            float diry = ((lumaNW + lumaSW) - (lumaNE + lumaSE));
            const float lumaSum = ((lumaNW + lumaNE) + lumaSW) + lumaSE;
            const float dirReduce = fmaxf(lumaSum * (0.25f * FXAA_REDUCE_MUL), FXAA_REDUCE_MIN);
            const float rcpDirMin = 1.0f / (fminf(fabsf(dirx), fabsf(diry)) + dirReduce);
            dirx = fminf(FXAA_SPAN_MAX, fmaxf(-FXAA_SPAN_MAX, dirx * rcpDirMin)) / resx;
            diry = fminf(FXAA_SPAN_MAX, fmaxf(-FXAA_SPAN_MAX, diry * rcpDirMin)) / resy;
            /* 2nd stage: blur along it */
            const f3 a0 = fxaa_linear(src, width, height, uvx + dirx * -0.166667f, uvy + diry * -0.166667f);
            const f3 a1 = fxaa_linear(src, width, height, uvx + dirx * 0.166667f, uvy + diry * 0.166667f);
            const f3 rgbA = f3_scale(f3_add(a0, a1), 0.5f);
            const f3 b0 = fxaa_linear(src, width, height, uvx + dirx * -0.5f, uvy + diry * -0.5f);
            const f3 b1 = fxaa_linear(src, width, height, uvx + dirx * 0.5f, uvy + diry * 0.5f);
            const f3 rgbB = f3_add(f3_scale(rgbA, 0.5f), f3_scale(f3_add(b0, b1), 0.25f));
            const float lumaB = f3_dot(rgbB, luma);
            const float lumaMin = fminf(lumaM, fminf(fminf(lumaNW, lumaNE), fminf(lumaSW, lumaSE)));
            const float lumaMax = fmaxf(lumaM, fmaxf(fmaxf(lumaNW, lumaNE), fmaxf(lumaSW, lumaSE)));
            const f3 out = ((lumaB < lumaMin) || (lumaB > lumaMax)) ? rgbA : rgbB;
            float* q = dst + 4 * ((size_t)j * (size_t)width + (size_t)i);
            q[0] = out.x; q[1] = out.y; q[2] = out.z; q[3] = 1.0f;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Hazard H8: upstream's render target is a GL_RGBA 8-bit UNORM texture (Renderer.cpp:63,192), so write_imagef
 * (kernel_main.cl:274,358) stores convert_uchar_sat_rte(x * 255) per channel and read_imagef (kernel_main.cl:347)
 * returns c / 255. orc_quantize_unorm8 applies that store+load to a float frame in place (OpenCL 1.2 spec 8.3.1.1:
 * NaN -> 0, round to nearest even, saturate); orc_pack_unorm8 emits the bytes.
 * ---------------------------------------------------------------------------------------- */
static inline uint8_t unorm8(float x)
{
    float v = x * 255.0f;
    if (!(v == v) || v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)rintf(v); /* default rounding mode: to nearest even */
}

void orc_quantize_unorm8(float* rgba, int width, int height, int row0, int row1)
{
    (void)height;
    for (int j = row0; j < row1; ++j)
        for (int i = 0; i < width; ++i) {
            float* p = rgba + 4 * ((size_t)j * (size_t)width + (size_t)i);
            for (int c = 0; c < 4; ++c) p[c] = (float)unorm8(p[c]) / 255.0f;
        }
}

void orc_pack_unorm8(const float* rgba, uint8_t* out, int width, int height, int row0, int row1)
{
    (void)height;
    for (int j = row0; j < row1; ++j)
        for (int i = 0; i < width; ++i) {
            const size_t k = 4 * ((size_t)j * (size_t)width + (size_t)i);
            for (int c = 0; c < 4; ++c) out[k + c] = unorm8(rgba[k + c]);
        }
}

/* ------------------------------------------------------------------------------------------
 * CPURayTrace.cpp:186-249  CPU_RayCast (one primary ray -> HitRecord, no lighting)
 * _mm_rcp_ps (vendor-specific 12-bit estimate) is pinned to IEEE 1/x.
 * ---------------------------------------------------------------------------------------- */
/* Math.hpp:53-90 polynomial ATan / ATan2 / ASin / ACos */
static inline float ref_atan(float x)
{
    const float x_sq = x * x;
    const float a1 = 0.99997726f, a3 = -0.33262347f, a5 = 0.19354346f, a7 = -0.11643287f, a9 = 0.05265332f, a11 = -0.01172120f;
    return x * (a1 + x_sq * (a3 + x_sq * (a5 + x_sq * (a7 + x_sq * (a9 + x_sq * a11)))));
}
static inline float ref_fabs(float x) { return x < 0.0f ? -x : x; }
static inline float ref_atan2(float y, float x)
{
    const float PI = 3.14159265358f, PI_2 = 1.5707963267f;
    float ay = ref_fabs(y), ax = ref_fabs(x);
    int invert = ay > ax;
    float z = invert ? ax / ay : ay / ax;
    float th = ref_atan(z);
    if (invert) th = PI_2 - th;
    if (x < 0) th = PI - th;
    return copysignf(th, y);
}
static inline float ref_acos(float x)
{
    const float PI = 3.14159265358f;
    const float PIDiv2 = PI / 2.0f;
    return PIDiv2 - ref_atan2(x, sqrtf(1.0f - (x * x)));
}
/* Math.hpp:41-44 Floor (truncation based) */
static inline float ref_floor(float x) { float whole = (float)f2i(x); return x - (x - whole); }

static int intersect_bvh_cpu(Ray ray, const CrtBVHNode* nodes, uint32_t rootNode, const CrtTri* tris, Triout* out)
{
    OrcStats st; memset(&st, 0, sizeof st);
    /* identical control flow (CPURayTrace.cpp:91-128); the only upstream difference is rcp */
    return intersect_bvh(ray, nodes, rootNode, tris, out, &st);
}

static CrtHitRecord cpu_raycast_one(const OrcScene* s, f3 origin, f3 direction)
{
    const float Miss = 1e30f;
    CrtHitRecord record; memset(&record, 0, sizeof record);
    record.normal[0] = 0.0f; record.normal[1] = 1.0f; record.normal[2] = 0.0f;
    record.distance = Miss;
    float bestDistance = Miss; uint32_t bestIndex = 0;
    Triout hitOut; memset(&hitOut, 0, sizeof hitOut);
    uint32_t hitInstanceIndex = 0;

    for (uint32_t i = 0; i < s->numInstances; ++i) {
        Triout triout; triout.t = bestDistance; triout.triIndex = 0; triout.u = 0.0f; triout.v = 0.0f;
        const CrtMeshInstance* instance = s->instances + i;
        /* Vector4Transform (Matrix.hpp:658-667): (v0 + v1) + (v2 + v3) */
        const CrtMatrix4* m = &instance->inverseTransform;
        float o4[3], d4[3];
        const float ov[4] = { origin.x, origin.y, origin.z, 1.0f }, dv[4] = { direction.x, direction.y, direction.z, 0.0f };
        for (int c = 0; c < 3; ++c) {
            o4[c] = (m->m[0][c] * ov[0] + m->m[1][c] * ov[1]) + (m->m[2][c] * ov[2] + m->m[3][c] * ov[3]);
            d4[c] = (m->m[0][c] * dv[0] + m->m[1][c] * dv[1]) + (m->m[2][c] * dv[2] + m->m[3][c] * dv[3]);
        }
        Ray meshRay = { f3_make(o4[0], o4[1], o4[2]), f3_make(d4[0], d4[1], d4[2]) };
        if (intersect_bvh_cpu(meshRay, s->nodes, s->roots[instance->meshIndex], s->tris, &triout)) {
            hitOut = triout;
            hitInstanceIndex = i;
            bestDistance = triout.t;
            bestIndex = instance->meshIndex;
        }
    }

    if (bestDistance == Miss) {
        const CrtTexture* tex = &s->textures[2];
        const float PI = 3.14159265358f;
        int32_t theta = f2i(((ref_atan2(direction.x, -direction.z) / PI) * 0.5f) * (float)tex->width);
        int32_t phi = f2i((ref_acos(direction.y) / PI) * (float)tex->height);
        int64_t idx = clamp_texel((int64_t)(int32_t)((uint32_t)phi * (uint32_t)tex->width + (uint32_t)theta + 2u), s->numTexels);
        CrtRGB8 px = s->texels[idx];
        record.color = (uint32_t)px.r | ((uint32_t)px.g << 8) | ((uint32_t)px.b << 16);
        return record;
    }

    const CrtMeshInstance* hitInstance = s->instances + hitInstanceIndex;
    const CrtTri* tri = s->tris + hitOut.triIndex;
    const CrtMaterial* material = s->materials + ((uint32_t)hitInstance->materialStart + (uint32_t)(int16_t)tri->materialIndex);
    float bx = (1.0f - hitOut.u) - hitOut.v, by = hitOut.u, bz = hitOut.v;
    const CrtMatrix4* inv = &hitInstance->inverseTransform;
    f3 n0 = mat3mul(inv, f3_make(orc_half_to_float_ref(tri->n0[0]), orc_half_to_float_ref(tri->n0[1]), orc_half_to_float_ref(tri->n0[2])));
    f3 n1 = mat3mul(inv, f3_make(orc_half_to_float_ref(tri->n1[0]), orc_half_to_float_ref(tri->n1[1]), orc_half_to_float_ref(tri->n1[2])));
    f3 n2 = mat3mul(inv, f3_make(orc_half_to_float_ref(tri->n2[0]), orc_half_to_float_ref(tri->n2[1]), orc_half_to_float_ref(tri->n2[2])));
    f3 nsum = f3_add(f3_add(f3_scale(n0, bx), f3_scale(n1, by)), f3_scale(n2, bz));
    /* Vector3::Normalize (Vector.hpp:142-144): a / Sqrt(Dot(a,a)), Dot = x*x + y*y + z*z */
    float len = sqrtf((nsum.x * nsum.x + nsum.y * nsum.y) + nsum.z * nsum.z);
    record.normal[0] = nsum.x / len; record.normal[1] = nsum.y / len; record.normal[2] = nsum.z / len;

    float uvx = (orc_half_to_float_ref(tri->uv0[0]) * bx + orc_half_to_float_ref(tri->uv1[0]) * by) + orc_half_to_float_ref(tri->uv2[0]) * bz;
    float uvy = (orc_half_to_float_ref(tri->uv0[1]) * bx + orc_half_to_float_ref(tri->uv1[1]) * by) + orc_half_to_float_ref(tri->uv2[1]) * bz;
    record.uv[0] = uvx; record.uv[1] = uvy;

    const CrtTexture* tex = &s->textures[material->albedoTextureIndex];
    float su = uvx - ref_floor(uvx), sv = uvy - ref_floor(uvy);
    int32_t uS = f2i((float)tex->width * su), vS = f2i((float)tex->height * sv);
    int64_t pidx = clamp_texel((int64_t)(int32_t)((uint32_t)vS * (uint32_t)tex->width + (uint32_t)tex->offset + (uint32_t)uS), s->numTexels);
    CrtRGB8 pixel = s->texels[pidx];
    uint32_t a = material->color, result = 0u;
    result |= ((a & 0xffu) * pixel.r) >> 8u;
    result |= ((((a >> 8u) & 0xffu) * pixel.g) >> 8u) << 8u;
    result |= ((((a >> 16u) & 0xffu) * pixel.b) >> 8u) << 16u;
    record.color = result;
    record.distance = bestDistance;
    record.index = bestIndex;
    return record;
}

void orc_cpu_raycast(const OrcScene* s, const float* origins, const float* dirs, int n, CrtHitRecord* out, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
    for (int k = 0; k < n; ++k)
        out[k] = cpu_raycast_one(s, f3_make(origins[3 * k], origins[3 * k + 1], origins[3 * k + 2]),
                                 f3_make(dirs[3 * k], dirs[3 * k + 1], dirs[3 * k + 2]));
}

/* ------------------------------------------------------------------------------------------
 * BVH.cpp:9-255  SAH-binned BVH2 builder
 * ---------------------------------------------------------------------------------------- */
static inline float sse_min(float a, float b) { return a < b ? a : b; } /* _mm_min_ps lane semantics */
static inline float sse_max(float a, float b) { return a > b ? a : b; }

typedef struct { float bmin[3], bmax[3]; } aabb_t;

static inline void aabb_init(aabb_t* b)
{
    for (int c = 0; c < 3; ++c) { b->bmin[c] = 1e30f; b->bmax[c] = -1e30f; }
}
static inline void aabb_grow_tri(aabb_t* b, const CrtTri* tri)
{
    for (int c = 0; c < 3; ++c) {
        b->bmin[c] = sse_min(b->bmin[c], tri->v0[c]);
        b->bmin[c] = sse_min(b->bmin[c], tri->v1[c]);
        b->bmin[c] = sse_min(b->bmin[c], tri->v2[c]);
        b->bmax[c] = sse_max(b->bmax[c], tri->v0[c]);
        b->bmax[c] = sse_max(b->bmax[c], tri->v1[c]);
        b->bmax[c] = sse_max(b->bmax[c], tri->v2[c]);
    }
}
static inline void aabb_grow_box(aabb_t* b, const aabb_t* o)
{
    if (o->bmin[0] != 1e30f) {
        for (int c = 0; c < 3; ++c) {
            b->bmin[c] = sse_min(b->bmin[c], o->bmin[c]);
            b->bmax[c] = sse_max(b->bmax[c], o->bmin[c]);
            b->bmin[c] = sse_min(b->bmin[c], o->bmax[c]);
            b->bmax[c] = sse_max(b->bmax[c], o->bmax[c]);
        }
    }
}
/* BVH.cpp:41-46 with hsum_ps_sse3 (SIMDCommon.hpp:183-189): (e0*e0 + e1*e0) + (e2*e2 + 0) */
static inline float extent_area(const float bmin[3], const float bmax[3])
{
    float e0 = bmax[0] - bmin[0], e1 = bmax[1] - bmin[1], e2 = bmax[2] - bmin[2];
    return (e0 * e0 + e1 * e0) + (e2 * e2 + 0.0f);
}

static inline float tri_centroid(const CrtTri* t, int axis)
{
    return axis == 0 ? t->centroidx : (axis == 1 ? t->centroidy : t->centroidz);
}

static void update_node_bounds(CrtBVHNode* nodes, const CrtTri* tris, uint32_t nodeIdx)
{
    CrtBVHNode* node = nodes + nodeIdx;
    float nmin[3] = { 1e30f, 1e30f, 1e30f }, nmax[3] = { -1e30f, -1e30f, -1e30f };
    const CrtTri* leaf = tris + node->leftFirst;
    for (uint32_t i = 0; i < node->triCount; i++, leaf++) {
        for (int c = 0; c < 3; ++c) {
            nmin[c] = sse_min(nmin[c], leaf->v0[c]);
            nmin[c] = sse_min(nmin[c], leaf->v1[c]);
            nmin[c] = sse_min(nmin[c], leaf->v2[c]);
            nmax[c] = sse_max(nmax[c], leaf->v0[c]);
            nmax[c] = sse_max(nmax[c], leaf->v1[c]);
            nmax[c] = sse_max(nmax[c], leaf->v2[c]);
        }
    }
    for (int c = 0; c < 3; ++c) { node->aabbMin[c] = nmin[c]; node->aabbMax[c] = nmax[c]; }
}

#define BINS 8
static float find_best_split_plane(const CrtBVHNode* node, const CrtTri* tris, int* outAxis, float* splitPos)
{
    float bestCost = 1e30f;
    uint32_t triCount = node->triCount, leftFirst = node->leftFirst;
    for (int axis = 0; axis < 3; ++axis) {
        float boundsMin = 1e30f, boundsMax = -1e30f;
        for (uint32_t i = 0; i < triCount; ++i) {
            float val = tri_centroid(tris + leftFirst + i, axis);
            boundsMin = boundsMin < val ? boundsMin : val;   /* Min(a,b) = a < b ? a : b */
            boundsMax = boundsMax > val ? boundsMax : val;   /* Max(a,b) = a > b ? a : b */
        }
        if (boundsMax == boundsMin) continue;

        aabb_t binBounds[BINS]; uint32_t binCount[BINS];
        for (int b = 0; b < BINS; ++b) { aabb_init(&binBounds[b]); binCount[b] = 0; }
        float scale = (float)BINS / (boundsMax - boundsMin);
        for (uint32_t i = 0; i < triCount; i++) {
            const CrtTri* triangle = tris + leftFirst + i;
            float centroid = tri_centroid(triangle, axis);
            int binIdx = f2i((centroid - boundsMin) * scale);
            binIdx = (BINS - 1) < binIdx ? (BINS - 1) : binIdx;
            if (binIdx < 0) binIdx = 0; /* unreachable for finite input; keeps the index defined */
            binCount[binIdx]++;
            aabb_grow_tri(&binBounds[binIdx], triangle);
        }

        float leftArea[BINS - 1], rightArea[BINS - 1];
        int leftCount[BINS - 1], rightCount[BINS - 1];
        int leftSum = 0, rightSum = 0;
        aabb_t leftBox, rightBox;
        aabb_init(&leftBox); aabb_init(&rightBox);
        for (int i = 0; i < BINS - 1; i++) {
            leftSum += (int)binCount[i];
            leftCount[i] = leftSum;
            aabb_grow_box(&leftBox, &binBounds[i]);
            leftArea[i] = extent_area(leftBox.bmin, leftBox.bmax);
            rightSum += (int)binCount[BINS - 1 - i];
            rightCount[BINS - 2 - i] = rightSum;
            aabb_grow_box(&rightBox, &binBounds[BINS - 1 - i]);
            rightArea[BINS - 2 - i] = extent_area(rightBox.bmin, rightBox.bmax);
        }

        scale = (boundsMax - boundsMin) / (float)BINS;
        for (int i = 0; i < BINS - 1; i++) {
            float planeCost = (float)leftCount[i] * leftArea[i] + (float)rightCount[i] * rightArea[i];
            if (planeCost < bestCost) {
                *splitPos = boundsMin + scale * (float)(i + 1);
                *outAxis = axis;
                bestCost = planeCost;
            }
        }
    }
    return bestCost;
}

static void subdivide(CrtBVHNode* nodes, CrtTri* tris, uint32_t nodeIdx, uint32_t* totalNodesUsed)
{
    CrtBVHNode* node = nodes + nodeIdx;
    uint32_t leftFirst = node->leftFirst, triCount = node->triCount;
    int axis = 0; float splitPos = 0.0f;
    float splitCost = find_best_split_plane(node, tris, &axis, &splitPos);
    float nosplitCost = (float)node->triCount * extent_area(node->aabbMin, node->aabbMax);
    if (splitCost >= nosplitCost) return;

    int i = (int)leftFirst;
    int j = i + (int)triCount - 1;
    while (i <= j) {
        if (tri_centroid(tris + i, axis) < splitPos) i++;
        else {
            CrtTri tmp = tris[i]; tris[i] = tris[j]; tris[j] = tmp;
            j--;
        }
    }
    int leftCount = i - (int)leftFirst;
    if (leftCount == 0 || leftCount == (int)triCount) return;
    uint32_t leftChildIdx = (*totalNodesUsed)++;
    uint32_t rightChildIdx = (*totalNodesUsed)++;
    nodes[leftChildIdx].leftFirst = leftFirst;
    nodes[leftChildIdx].triCount = (uint32_t)leftCount;
    nodes[rightChildIdx].leftFirst = (uint32_t)i;
    nodes[rightChildIdx].triCount = triCount - (uint32_t)leftCount;
    node->leftFirst = leftChildIdx;
    node->triCount = 0;
    update_node_bounds(nodes, tris, leftChildIdx);
    update_node_bounds(nodes, tris, rightChildIdx);
    subdivide(nodes, tris, leftChildIdx, totalNodesUsed);
    subdivide(nodes, tris, rightChildIdx, totalNodesUsed);
}

uint32_t orc_build_bvh(CrtTri* tris, const uint32_t* meshTriCounts, int numMeshes,
                       CrtBVHNode* nodes, uint32_t* roots, uint32_t* nodeCounter)
{
    int numTriangles = 0;
    for (int i = 0; i < numMeshes; ++i) numTriangles += (int)meshTriCounts[i];
    for (int i = 0; i < numTriangles; i++) {
        CrtTri* tri = tris + i;
        tri->centroidx = ((tri->v0[0] + tri->v1[0]) + tri->v2[0]) * 0.333333f;
        tri->centroidy = ((tri->v0[1] + tri->v1[1]) + tri->v2[1]) * 0.333333f;
        tri->centroidz = ((tri->v0[2] + tri->v1[2]) + tri->v2[2]) * 0.333333f;
    }
    uint32_t nodesUsedStart = *nodeCounter;
    int currTriangle = 0;
    for (int i = 0; i < numMeshes; ++i) {
        uint32_t rootNodeIndex = (*nodeCounter)++;
        roots[i] = rootNodeIndex;
        CrtBVHNode* root = nodes + rootNodeIndex;
        root->leftFirst = (uint32_t)currTriangle;
        root->triCount = meshTriCounts[i];
        update_node_bounds(nodes, tris, rootNodeIndex);
        subdivide(nodes, tris, rootNodeIndex, nodeCounter);
        currTriangle += (int)meshTriCounts[i];
    }
    return *nodeCounter - nodesUsedStart;
}

/* ------------------------------------------------------------------------------------------
 * Matrix.hpp  InverseTransform / Inverse / PerspectiveFovRH / LookAtRH
 * ---------------------------------------------------------------------------------------- */
typedef struct { float f[4]; } v4;
static inline v4 v4_set(float a, float b, float c, float d) { v4 r = { { a, b, c, d } }; return r; }
static inline v4 v4_add(v4 a, v4 b) { return v4_set(a.f[0] + b.f[0], a.f[1] + b.f[1], a.f[2] + b.f[2], a.f[3] + b.f[3]); }
static inline v4 v4_sub(v4 a, v4 b) { return v4_set(a.f[0] - b.f[0], a.f[1] - b.f[1], a.f[2] - b.f[2], a.f[3] - b.f[3]); }
static inline v4 v4_mul(v4 a, v4 b) { return v4_set(a.f[0] * b.f[0], a.f[1] * b.f[1], a.f[2] * b.f[2], a.f[3] * b.f[3]); }
static inline v4 v4_div(v4 a, v4 b) { return v4_set(a.f[0] / b.f[0], a.f[1] / b.f[1], a.f[2] / b.f[2], a.f[3] / b.f[3]); }
static inline v4 v4_swz(v4 a, int x, int y, int z, int w) { return v4_set(a.f[x], a.f[y], a.f[z], a.f[w]); }
/* VecShuffle(v1, v2, x,y,z,w) = (v1[x], v1[y], v2[z], v2[w]) */
static inline v4 v4_shuf(v4 a, v4 b, int x, int y, int z, int w) { return v4_set(a.f[x], a.f[y], b.f[z], b.f[w]); }
static inline v4 v4_load(const float* p) { return v4_set(p[0], p[1], p[2], p[3]); }
static inline void v4_store(float* p, v4 a) { memcpy(p, a.f, 16); }

/* Matrix.hpp:292-325 */
void orc_inverse_transform(const float in[16], float out[16])
{
    v4 r0 = v4_load(in), r1 = v4_load(in + 4), r2 = v4_load(in + 8), r3 = v4_load(in + 12);
    v4 t0 = v4_shuf(r0, r1, 0, 1, 0, 1);
    v4 t1 = v4_shuf(r0, r1, 2, 3, 2, 3);
    v4 o0 = v4_shuf(t0, r2, 0, 2, 0, 3);
    v4 o1 = v4_shuf(t0, r2, 1, 3, 1, 3);
    v4 o2 = v4_shuf(t1, r2, 0, 2, 2, 3);
    v4 sizeSqr = v4_mul(o0, o0);
    sizeSqr = v4_add(sizeSqr, v4_mul(o1, o1));
    sizeSqr = v4_add(sizeSqr, v4_mul(o2, o2));
    v4 rSizeSqr;
    for (int c = 0; c < 4; ++c) rSizeSqr.f[c] = (sizeSqr.f[c] < 1.e-8f) ? 1.0f : (1.0f / sizeSqr.f[c]);
    o0 = v4_mul(o0, rSizeSqr);
    o1 = v4_mul(o1, rSizeSqr);
    o2 = v4_mul(o2, rSizeSqr);
    v4 o3 = v4_mul(o0, v4_swz(r3, 0, 0, 0, 0));
    o3 = v4_add(o3, v4_mul(o1, v4_swz(r3, 1, 1, 1, 1)));
    o3 = v4_add(o3, v4_mul(o2, v4_swz(r3, 2, 2, 2, 2)));
    o3 = v4_sub(v4_set(0.f, 0.f, 0.f, 1.f), o3);
    v4_store(out, o0); v4_store(out + 4, o1); v4_store(out + 8, o2); v4_store(out + 12, o3);
}

/* Matrix.hpp:270-290 2x2 helpers */
static inline v4 mat2mul(v4 a, v4 b)
{
    return v4_add(v4_mul(a, v4_swz(b, 0, 3, 0, 3)), v4_mul(v4_swz(a, 1, 0, 3, 2), v4_swz(b, 2, 1, 2, 1)));
}
static inline v4 mat2adjmul(v4 a, v4 b)
{
    return v4_sub(v4_mul(v4_swz(a, 3, 3, 0, 0), b), v4_mul(v4_swz(a, 1, 1, 2, 2), v4_swz(b, 2, 3, 0, 1)));
}
static inline v4 mat2muladj(v4 a, v4 b)
{
    return v4_sub(v4_mul(a, v4_swz(b, 3, 0, 3, 0)), v4_mul(v4_swz(a, 1, 0, 3, 2), v4_swz(b, 2, 1, 2, 1)));
}

/* Matrix.hpp:327-374 */
void orc_inverse(const float in[16], float out[16])
{
    v4 r0 = v4_load(in), r1 = v4_load(in + 4), r2 = v4_load(in + 8), r3 = v4_load(in + 12);
    v4 A = v4_shuf(r0, r1, 0, 1, 0, 1);
    v4 B = v4_shuf(r0, r1, 2, 3, 2, 3);
    v4 C = v4_shuf(r2, r3, 0, 1, 0, 1);
    v4 D = v4_shuf(r2, r3, 2, 3, 2, 3);
    v4 detSub = v4_sub(v4_mul(v4_shuf(r0, r2, 0, 2, 0, 2), v4_shuf(r1, r3, 1, 3, 1, 3)),
                       v4_mul(v4_shuf(r0, r2, 1, 3, 1, 3), v4_shuf(r1, r3, 0, 2, 0, 2)));
    v4 detA = v4_swz(detSub, 0, 0, 0, 0), detB = v4_swz(detSub, 1, 1, 1, 1);
    v4 detC = v4_swz(detSub, 2, 2, 2, 2), detD = v4_swz(detSub, 3, 3, 3, 3);
    v4 D_C = mat2adjmul(D, C);
    v4 A_B = mat2adjmul(A, B);
    v4 X_ = v4_sub(v4_mul(detD, A), mat2mul(B, D_C));
    v4 W_ = v4_sub(v4_mul(detA, D), mat2mul(C, A_B));
    v4 detM = v4_mul(detA, detD);
    v4 Y_ = v4_sub(v4_mul(detB, C), mat2muladj(D, A_B));
    v4 Z_ = v4_sub(v4_mul(detC, B), mat2muladj(A, D_C));
    detM = v4_add(detM, v4_mul(detB, detC));
    v4 tr = v4_mul(A_B, v4_swz(D_C, 0, 2, 1, 3));
    /* _mm_hadd_ps(tr,tr) twice: ((t0+t1) + (t2+t3)) in every lane */
    float h = (tr.f[0] + tr.f[1]) + (tr.f[2] + tr.f[3]);
    detM = v4_sub(detM, v4_set(h, h, h, h));
    v4 rDetM = v4_div(v4_set(1.f, -1.f, -1.f, 1.f), detM);
    X_ = v4_mul(X_, rDetM); Y_ = v4_mul(Y_, rDetM); Z_ = v4_mul(Z_, rDetM); W_ = v4_mul(W_, rDetM);
    v4_store(out,      v4_shuf(X_, Y_, 3, 1, 3, 1));
    v4_store(out + 4,  v4_shuf(X_, Y_, 2, 0, 2, 0));
    v4_store(out + 8,  v4_shuf(Z_, W_, 3, 1, 3, 1));
    v4_store(out + 12, v4_shuf(Z_, W_, 2, 0, 2, 0));
}

/* Math.hpp:33-38, 92-112 FMod / Sin / Cos polynomials (pure fp32, restated exactly) */
static inline float ref_fmod(float x, float y)
{
    float quotient = x / y;
    float whole = (float)f2i(quotient);
    float remainder = x - whole * y;
    remainder += (float)(remainder < 0.0f) * y;
    return remainder;
}
static inline float ref_sin(float x)
{
    const float PI = 3.14159265358f, TwoPI = PI * 2.0f;
    x = ref_fmod(x + PI, TwoPI) - PI;
    float xx = x * x * x;
    float t = x - (xx * 0.16666666666f);
    t += (xx *= x * x) * 0.00833333333f;
    t -= (xx *= x * x) * 0.00019841269f;
    t += (xx * x * x) / 362880.0f;
    return t;
}
static inline float ref_cos(float x)
{
    const float PI = 3.14159265358f, TwoPI = PI * 2.0f;
    x = ref_fmod(x + PI, TwoPI) - PI;
    float xx = x * x;
    float t = 1.0f - (xx * 0.5f);
    t += (xx *= x * x) * 0.04166666666f;
    t -= (xx *= x * x) * 0.00138888888f;
    t += (xx * x * x) / 40320.0f;
    return t;
}

/* Matrix.hpp:237-250 */
void orc_perspective_fov_rh(float fov, float width, float height, float zNear, float zFar, float out[16])
{
    const float rad = fov;
    const float h = ref_cos(0.5f * rad) / ref_sin(0.5f * rad);
    const float w = h * height / width;
    memset(out, 0, 64);
    out[0] = 1.0f; out[5] = 1.0f; out[10] = 1.0f; out[15] = 1.0f;
    out[0] = w;
    out[5] = h;
    out[10] = -(zFar + zNear) / (zFar - zNear);
    out[11] = -1.0f;
    out[14] = -(2.0f * zFar * zNear) / (zFar - zNear);
    out[15] = 0.0f;
}

/* Matrix.hpp:211-235, with _mm_rsqrt_ps (hardware estimate, hazard H10) pinned to 1/sqrtf */
static inline f3 lookat_normalize(f3 v)
{
    float d = (v.x * v.x + v.y * v.y) + v.z * v.z; /* _mm_dp_ps 0x7f: (x+y)+(z+0) */
    float r = 1.0f / sqrtf(d);
    return f3_make(r * v.x, r * v.y, r * v.z);
}
/* SSEVector3Cross (SIMDCommon.hpp:121-129): tmp3 - tmp4 = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x) */
void orc_look_at_rh(const float eye[3], const float center[3], const float up[3], float out[16])
{
    f3 EyePosition = f3_make(eye[0], eye[1], eye[2]);
    f3 EyeDirection = f3_make(0.0f - center[0], 0.0f - center[1], 0.0f - center[2]);
    f3 UpDirection = f3_make(up[0], up[1], up[2]);
    f3 R0 = lookat_normalize(f3_cross(UpDirection, EyeDirection));
    f3 R1 = lookat_normalize(f3_cross(EyeDirection, R0));
    f3 NegEye = f3_make(0.0f - EyePosition.x, 0.0f - EyePosition.y, 0.0f - EyePosition.z);
    float D0 = f3_dot(R0, NegEye), D1 = f3_dot(R1, NegEye), D2 = f3_dot(EyeDirection, NegEye);
    /* rows (R0,D0), (R1,D1), (EyeDirection,D2), (0,0,0,1) then transposed */
    float M[4][4] = { { R0.x, R0.y, R0.z, D0 }, { R1.x, R1.y, R1.z, D1 },
                      { EyeDirection.x, EyeDirection.y, EyeDirection.z, D2 }, { 0.f, 0.f, 0.f, 1.f } };
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[r * 4 + c] = M[c][r];
}
