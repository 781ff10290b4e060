// CPURayTrace.cpp -- single-ray CPU pick ray over the host arenas (reference: CPURayTrace.cpp:1-249).
// This is the reference's own CPU function (config 1), not a fallback for the GPU path: it returns a
// HitRecord (albedo x material colour, no lighting, no bounce) and is never used by Render().
//
// Differences from upstream, both forced by hardware-defined instructions: _mm_rcp_ps (invDir and
// the triangle's 1/a, CPURayTrace.cpp:49,95) is a vendor-specific 12-bit estimate -> IEEE 1/x here;
// the host texel arena mirrors the device pool (see ResourceManager.cpp header).
#include "CPURayTrace.hpp"
#include <cmath>

namespace ResourceManager { size_t TexelBytesUsed(); }

namespace {

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 load3(const float* p) { return { p[0], p[1], p[2] }; }

struct Triout { float t, u, v; uint triIndex; };

// CPURayTrace.cpp:42-63
bool IntersectTriangle(V3 o, V3 d, const Tri* tri, Triout* out, int i)
{
    const V3 v0 = load3(tri->v0);
    const V3 edge1 = sub(load3(tri->v1), v0), edge2 = sub(load3(tri->v2), v0);
    const V3 h = cross(d, edge2);
    const float f = 1.0f / dot(edge1, h);
    const V3 s = sub(o, v0);
    const float u = f * dot(s, h);
    const V3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    const float t = f * dot(edge2, q);
    const int passed = (((t > 0.0f) ^ (t < out->t)) + (u < 0.0f) + (u > 1.0f) + (v < 0.0f) + (u + v > 1.0f)) == 0;
    const int notPassed = 1 - passed;
    out->u = u * (float)passed + ((float)notPassed * out->u);
    out->v = v * (float)passed + ((float)notPassed * out->v);
    out->t = t * (float)passed + ((float)notPassed * out->t);
    out->triIndex = (uint)i * (uint)passed + ((uint)notPassed * out->triIndex);
    return passed != 0;
}

// CPURayTrace.cpp:77-86
float IntersectAABB(V3 o, V3 inv, const float* bmin, const float* bmax, float minSoFar)
{
    const float t0x = (bmin[0] - o.x) * inv.x, t0y = (bmin[1] - o.y) * inv.y, t0z = (bmin[2] - o.z) * inv.z;
    const float t1x = (bmax[0] - o.x) * inv.x, t1y = (bmax[1] - o.y) * inv.y, t1z = (bmax[2] - o.z) * inv.z;
    const float tnear = std::fmax(std::fmax(std::fmin(t0x, t1x), std::fmin(t0y, t1y)), std::fmin(t0z, t1z));
    const float tfar = std::fmin(std::fmin(std::fmax(t0x, t1x), std::fmax(t0y, t1y)), std::fmax(t0z, t1z));
    return (tnear < tfar && tnear > 0.0f && tnear < minSoFar) ? tnear : RayacastMissDistance;
}

// CPURayTrace.cpp:91-128
bool IntersectBVH(V3 o, V3 d, const BVHNode* nodes, uint rootNode, const Tri* tris, Triout* out)
{
    uint stack[32] = { rootNode };
    int sp = 1, protection = 0;
    const V3 inv = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    bool intersection = false;
    while (sp > 0 && protection++ < 250) {
        const BVHNode* node = nodes + stack[--sp & 31];
        for (;;) {
            if (node->triCount > 0) {
                for (int i = (int)node->leftFirst, end = i + (int)node->triCount; i < end; ++i)
                    intersection |= IntersectTriangle(o, d, tris + i, out, i);
                break;
            }
            uint l = node->leftFirst, r = l + 1;
            float d1 = IntersectAABB(o, inv, nodes[l].aabbMin, nodes[l].aabbMax, out->t);
            float d2 = IntersectAABB(o, inv, nodes[r].aabbMin, nodes[r].aabbMax, out->t);
            if (d1 > d2) { float tf = d1; d1 = d2; d2 = tf; uint tu = l; l = r; r = tu; }
            if (d1 == RayacastMissDistance) break;
            node = nodes + l;
            if (d2 != RayacastMissDistance) { stack[sp & 31] = r; ++sp; }
        }
    }
    return intersection;
}

// Math.hpp:53-90
inline float ATanPoly(float x)
{
    const float xs = x * x;
    return x * (0.99997726f + xs * (-0.33262347f + xs * (0.19354346f + xs * (-0.11643287f + xs * (0.05265332f + xs * -0.01172120f)))));
}
inline float ATan2(float y, float x)
{
    const float PI_2 = 1.5707963267f;
    const float ay = y < 0.0f ? -y : y, ax = x < 0.0f ? -x : x;
    const int invert = ay > ax;
    const float z = invert ? ax / ay : ay / ax;
    float th = ATanPoly(z);
    if (invert) th = PI_2 - th;
    if (x < 0) th = crtmath::PI - th;
    return std::copysign(th, y);
}
inline float ACos(float x) { return (crtmath::PI / 2.0f) - ATan2(x, std::sqrt(1.0f - (x * x))); }
inline float FloorTrunc(float x) { const float whole = (float)crtmath::TruncToInt(x); return x - (x - whole); } // Math.hpp:41-44

inline RGB8 texel_at(long long idx)
{
    const long long n = (long long)((ResourceManager::TexelBytesUsed() + 2) / 3);
    if (idx < 0) idx = 0;
    if (idx >= n) idx = n - 1;
    return g_TexturePixels[idx];
}

#if defined(__x86_64__)
} // namespace
#include <immintrin.h>
namespace {
// The reference's own instruction mix (CPURayTrace.cpp:42-128): one ray in two __m128 registers, _mm_dp_ps dot products,
// the 12-bit _mm_rcp_ps estimates for 1/direction and for Moeller-Trumbore's 1/a. This flavour exists for TIMING
// (bench.py's cpu_baseline, tools/cpu_baseline.py): `rcpps` is implementation-defined (Intel and AMD return different
// bits), so its hit records can differ from the IEEE flavour above in the last places of t,u,v and -- rarely -- in which
// triangle wins; tests only require the two to agree on >= 99.9 % of the rays.
#define CRT_SSE41 __attribute__((target("sse4.1")))
CRT_SSE41 inline __m128 load3z(const float* p) { return _mm_set_ps(0.0f, p[2], p[1], p[0]); }
CRT_SSE41 inline __m128 cross_ps(__m128 a, __m128 b)
{
    const __m128 a_yzx = _mm_shuffle_ps(a, a, _MM_SHUFFLE(3, 0, 2, 1)), b_yzx = _mm_shuffle_ps(b, b, _MM_SHUFFLE(3, 0, 2, 1));
    const __m128 c = _mm_sub_ps(_mm_mul_ps(a, b_yzx), _mm_mul_ps(a_yzx, b));
    return _mm_shuffle_ps(c, c, _MM_SHUFFLE(3, 0, 2, 1));
}
CRT_SSE41 bool IntersectTriangleSSE(__m128 o, __m128 d, const Tri* tri, Triout* out, int i)
{
    const __m128 v0 = load3z(tri->v0);
    const __m128 edge1 = _mm_sub_ps(load3z(tri->v1), v0), edge2 = _mm_sub_ps(load3z(tri->v2), v0);
    const __m128 h = cross_ps(d, edge2);
    const __m128 f = _mm_rcp_ps(_mm_dp_ps(edge1, h, 0x7f));
    const __m128 s = _mm_sub_ps(o, v0);
    const float u = _mm_cvtss_f32(_mm_mul_ps(f, _mm_dp_ps(s, h, 0x7f)));
    const __m128 q = cross_ps(s, edge1);
    const float v = _mm_cvtss_f32(_mm_mul_ps(f, _mm_dp_ps(d, q, 0x7f)));
    const float t = _mm_cvtss_f32(_mm_mul_ps(f, _mm_dp_ps(edge2, q, 0x7f)));
    const int passed = (((t > 0.0f) ^ (t < out->t)) + (u < 0.0f) + (u > 1.0f) + (v < 0.0f) + (u + v > 1.0f)) == 0;
    const int notPassed = 1 - passed;
    out->u = u * (float)passed + ((float)notPassed * out->u);
    out->v = v * (float)passed + ((float)notPassed * out->v);
    out->t = t * (float)passed + ((float)notPassed * out->t);
    out->triIndex = (uint)i * (uint)passed + ((uint)notPassed * out->triIndex);
    return passed != 0;
}
CRT_SSE41 inline float hmax3(__m128 v) { const __m128 m = _mm_max_ps(_mm_shuffle_ps(v, v, _MM_SHUFFLE(0, 0, 0, 0)), _mm_shuffle_ps(v, v, _MM_SHUFFLE(1, 1, 1, 1))); return _mm_cvtss_f32(_mm_max_ps(m, _mm_shuffle_ps(v, v, _MM_SHUFFLE(2, 2, 2, 2)))); }
CRT_SSE41 inline float hmin3(__m128 v) { const __m128 m = _mm_min_ps(_mm_shuffle_ps(v, v, _MM_SHUFFLE(0, 0, 0, 0)), _mm_shuffle_ps(v, v, _MM_SHUFFLE(1, 1, 1, 1))); return _mm_cvtss_f32(_mm_min_ps(m, _mm_shuffle_ps(v, v, _MM_SHUFFLE(2, 2, 2, 2)))); }
CRT_SSE41 inline float IntersectAABBSSE(__m128 o, __m128 inv, const float* bmin, const float* bmax, float minSoFar)
{
    const __m128 t0 = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(bmin), o), inv), t1 = _mm_mul_ps(_mm_sub_ps(_mm_loadu_ps(bmax), o), inv);
    const float tnear = hmax3(_mm_min_ps(t0, t1)), tfar = hmin3(_mm_max_ps(t0, t1));
    return (tnear < tfar && tnear > 0.0f && tnear < minSoFar) ? tnear : RayacastMissDistance;
}
CRT_SSE41 bool IntersectBVHSSE(V3 o3, V3 d3, const BVHNode* nodes, uint rootNode, const Tri* tris, Triout* out)
{
    const __m128 o = _mm_set_ps(1.0f, o3.z, o3.y, o3.x), d = _mm_set_ps(0.0f, d3.z, d3.y, d3.x);
    uint stack[32] = { rootNode };
    int sp = 1, protection = 0;
    const __m128 inv = _mm_rcp_ps(d);
    bool intersection = false;
    while (sp > 0 && protection++ < 250) {
        const BVHNode* node = nodes + stack[--sp & 31];
        for (;;) {
            if (node->triCount > 0) {
                for (int i = (int)node->leftFirst, end = i + (int)node->triCount; i < end; ++i)
                    intersection |= IntersectTriangleSSE(o, d, tris + i, out, i);
                break;
            }
            uint l = node->leftFirst, r = l + 1;
            float d1 = IntersectAABBSSE(o, inv, nodes[l].aabbMin, nodes[l].aabbMax, out->t);
            float d2 = IntersectAABBSSE(o, inv, nodes[r].aabbMin, nodes[r].aabbMax, out->t);
            if (d1 > d2) { float tf = d1; d1 = d2; d2 = tf; uint tu = l; l = r; r = tu; }
            if (d1 == RayacastMissDistance) break;
            node = nodes + l;
            if (d2 != RayacastMissDistance) { stack[sp & 31] = r; ++sp; }
        }
    }
    return intersection;
}
#define CRT_HAVE_SSE_FLAVOUR 1
#endif

} // namespace

void CPU_RayTraceInitialize() {}

namespace {
typedef bool (*BvhFn)(V3, V3, const BVHNode*, uint, const Tri*, Triout*);
template <BvhFn INTERSECT>
HitRecord ray_cast(RaySSE ray) // CPURayTrace.cpp:186-249
{
    HitRecord record;
    std::memset(&record, 0, sizeof record);
    record.normal[1] = 1.0f;
    record.distance = RayacastMissDistance;
    float bestDistance = RayacastMissDistance; uint bestIndex = 0;
    Triout hitOut = { 0, 0, 0, 0 };
    uint hitInstanceIndex = 0u;
    const float ov[4] = { ray.origin[0], ray.origin[1], ray.origin[2], 1.0f };
    const float dv[4] = { ray.direction[0], ray.direction[1], ray.direction[2], 0.0f };

    for (uint i = 0; i < g_NumMeshInstances; ++i) {
        Triout triout = { bestDistance, 0.0f, 0.0f, 0u };
        const MeshInstance& instance = g_MeshInstances[i];
        const float (*m)[4] = instance.inverseTransform.m;
        float o3[3], d3[3];
        for (int c = 0; c < 3; ++c) { // Vector4Transform (Matrix.hpp:658-667)
            o3[c] = (m[0][c] * ov[0] + m[1][c] * ov[1]) + (m[2][c] * ov[2] + m[3][c] * ov[3]);
            d3[c] = (m[0][c] * dv[0] + m[1][c] * dv[1]) + (m[2][c] * dv[2] + m[3][c] * dv[3]);
        }
        if (INTERSECT(load3(o3), load3(d3), g_BVHNodes, g_BVHIndices[instance.meshIndex], g_Triangles, &triout)) {
            hitOut = triout; hitInstanceIndex = i; bestDistance = triout.t; bestIndex = instance.meshIndex;
        }
    }

    if (bestDistance == RayacastMissDistance) {
        const Texture& sky = g_Textures[2];
        const int theta = crtmath::TruncToInt(((ATan2(dv[0], -dv[2]) / crtmath::PI) * 0.5f) * (float)sky.width);
        const int phi = crtmath::TruncToInt((ACos(dv[1]) / crtmath::PI) * (float)sky.height);
        const RGB8 px = texel_at((long long)(int)((uint)phi * (uint)sky.width + (uint)theta + 2u));
        record.color = (uint)px.r | ((uint)px.g << 8) | ((uint)px.b << 16);
        return record;
    }

    const MeshInstance& hitInstance = g_MeshInstances[hitInstanceIndex];
    const Tri& tri = g_Triangles[hitOut.triIndex];
    const Material& material = g_Materials[(uint)hitInstance.materialStart + (uint)(short)tri.materialIndex];
    const float bx = (1.0f - hitOut.u) - hitOut.v, by = hitOut.u, bz = hitOut.v;
    const float (*m)[4] = hitInstance.inverseTransform.m;
    auto xform = [&](const half* h) {
        const float x = crtmath::ConvertHalfToFloat(h[0]), y = crtmath::ConvertHalfToFloat(h[1]), z = crtmath::ConvertHalfToFloat(h[2]);
        return V3{ (m[0][0] * x + m[1][0] * y) + m[2][0] * z, (m[0][1] * x + m[1][1] * y) + m[2][1] * z, (m[0][2] * x + m[1][2] * y) + m[2][2] * z };
    };
    const V3 n0 = xform(tri.n0), n1 = xform(tri.n1), n2 = xform(tri.n2);
    const V3 ns = { (n0.x * bx + n1.x * by) + n2.x * bz, (n0.y * bx + n1.y * by) + n2.y * bz, (n0.z * bx + n1.z * by) + n2.z * bz };
    const float len = std::sqrt((ns.x * ns.x + ns.y * ns.y) + ns.z * ns.z);
    record.normal[0] = ns.x / len; record.normal[1] = ns.y / len; record.normal[2] = ns.z / len;
    using crtmath::ConvertHalfToFloat;
    record.uv[0] = (ConvertHalfToFloat(tri.uv0[0]) * bx + ConvertHalfToFloat(tri.uv1[0]) * by) + ConvertHalfToFloat(tri.uv2[0]) * bz;
    record.uv[1] = (ConvertHalfToFloat(tri.uv0[1]) * bx + ConvertHalfToFloat(tri.uv1[1]) * by) + ConvertHalfToFloat(tri.uv2[1]) * bz;

    const Texture& tex = g_Textures[material.albedoTextureIndex];
    const float su = record.uv[0] - FloorTrunc(record.uv[0]), sv = record.uv[1] - FloorTrunc(record.uv[1]);
    const int uS = crtmath::TruncToInt((float)tex.width * su), vS = crtmath::TruncToInt((float)tex.height * sv);
    const RGB8 pixel = texel_at((long long)(int)((uint)vS * (uint)tex.width + (uint)tex.offset + (uint)uS));
    const uint a = material.color;
    uint packed = 0u;
    packed |= ((a & 0xffu) * pixel.r) >> 8u;
    packed |= ((((a >> 8u) & 0xffu) * pixel.g) >> 8u) << 8u;
    packed |= ((((a >> 16u) & 0xffu) * pixel.b) >> 8u) << 16u;
    record.color = packed;
    record.distance = bestDistance;
    record.index = bestIndex;
    return record;
}
} // namespace

HitRecord CPU_RayCast(RaySSE ray) { return ray_cast<IntersectBVH>(ray); }

// The reference's SSE instruction mix (approximate _mm_rcp_ps); falls back to CPU_RayCast where SSE4.1 is not available.
HitRecord CPU_RayCastSSE(RaySSE ray)
{
#ifdef CRT_HAVE_SSE_FLAVOUR
    if (__builtin_cpu_supports("sse4.1")) return ray_cast<IntersectBVHSSE>(ray);
#endif
    return ray_cast<IntersectBVH>(ray);
}
