/* crt_types.h -- plain-old-data contracts shared by the host mirror, the C-ABI shim and tests.
 *
 * Every struct is byte-identical to the reference's host<->device struct it replaces, so a
 * reference-side caller can hand its arenas to the C-ABI unchanged. Citations are relative to
 * the upstream tree (CLRayTracer/...):
 *   CrtTri         <- ResourceManager.hpp:54-67 (host `Tri`), kernels/kernel_main.cl:34-43 (`Triangle`)
 *   CrtBVHNode     <- ResourceManager.hpp:7-11, kernel_main.cl:54-56
 *   CrtMaterial    <- ResourceManager.hpp:44-50, kernel_main.cl:26-32
 *   CrtTexture     <- ResourceManager.hpp:27-29, kernels/MathAndSTL.cl:229-231
 *   CrtRGB8        <- ResourceManager.hpp:14-16, MathAndSTL.cl:233-236
 *   CrtMeshInstance<- Renderer.hpp:6-10, kernel_main.cl:49-52
 *   CrtTraceArgs   <- Renderer.cpp:326-331, kernel_main.cl:9-14
 *   CrtHitRecord   <- CPURayTrace.hpp:5-12
 */
#ifndef CRT_TYPES_H
#define CRT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t crt_half; /* IEEE binary16 bits; reference `typedef ushort half` (Math/Math.hpp:154) */

typedef struct CrtTri {
    float v0[3]; float centroidx;     /* vertex0 + centroid lane written by BuildBVH (BVH.cpp:232) */
    float v1[3]; float centroidy;
    float v2[3]; float centroidz;
    crt_half uv0[2], uv1[2], uv2[2];
    uint16_t materialIndex;           /* `short` on the host, `ushort` on the device */
    crt_half n0[3], n1[3], n2[3];
} CrtTri;

typedef struct CrtBVHNode {
    float aabbMin[3]; uint32_t leftFirst; /* inner: index of the left child (right = +1); leaf: first triangle */
    float aabbMax[3]; uint32_t triCount;  /* >0 marks a leaf */
} CrtBVHNode;

typedef struct CrtMaterial {
    uint32_t color;                   /* 0x00BBGGRR */
    uint32_t specularColor;
    uint16_t albedoTextureIndex;
    uint16_t specularTextureIndex;
    crt_half shininess, roughness;
} CrtMaterial;

typedef struct CrtTexture { int32_t width, height, offset, padd; } CrtTexture; /* offset in texels */

typedef struct CrtRGB8 { uint8_t r, g, b; } CrtRGB8;

typedef struct CrtMatrix4 { float m[4][4]; } CrtMatrix4; /* row-major, row-vector convention (MathAndSTL.cl:100-102) */

typedef struct CrtMeshInstance {
    CrtMatrix4 inverseTransform;
    uint16_t meshIndex, materialStart;
    uint8_t _pad[12];                 /* AX_ALIGNED(16) tail padding of the reference struct */
} CrtMeshInstance;

typedef struct CrtTraceArgs {
    float cameraPos[3];
    float time;
    uint32_t numMeshes;               /* number of registered mesh *instances* */
    float sunAngle;
} CrtTraceArgs;

typedef struct CrtHitRecord {
    float normal[3];
    float uv[2];
    float distance;
    uint32_t color;
    uint32_t index;
} CrtHitRecord;

/* Per-ray closest-hit record returned by the ray-query entry points (the reference's `Triout`
 * kernel_main.cl:45-47 plus the winning instance index kept by the loop at kernel_main.cl:198-217). */
typedef struct CrtRayHit {
    float t, u, v;
    uint32_t triIndex;
    int32_t instance;                 /* -1: miss */
} CrtRayHit;

/* Compile-time limits of the reference. */
enum {
    CRT_MAX_INSTANCES   = 401,        /* Renderer.hpp:16 */
    CRT_MAX_TRIANGLES   = 1200000,    /* ResourceManager.cpp:34 (device pool holds 2x) */
    CRT_MAX_TEXTURES    = 32,         /* ResourceManager.cpp:38 */
    CRT_MAX_MATERIALS   = 256,        /* ResourceManager.cpp:39 */
    CRT_MAX_MESHES      = 128,        /* ResourceManager.cpp:40 */
    CRT_STACK_DEPTH     = 32,         /* kernel_main.cl:126 */
    CRT_MAX_POPS        = 250         /* kernel_main.cl:131 */
};
#define CRT_MAX_TEXTURE_BYTES ((size_t)104900000u) /* ResourceManager.cpp:32, 1.049e7*10 */

#ifdef __cplusplus
}
#if __cplusplus >= 201103L
static_assert(sizeof(CrtTri) == 80, "Tri must be 80 B (ResourceManager.hpp:69)");
static_assert(sizeof(CrtBVHNode) == 32, "BVHNode must be 32 B");
static_assert(sizeof(CrtMaterial) == 16, "Material must be 16 B");
static_assert(sizeof(CrtTexture) == 16, "Texture must be 16 B");
static_assert(sizeof(CrtRGB8) == 3, "RGB8 must be 3 B");
static_assert(sizeof(CrtMeshInstance) == 80, "MeshInstance must be 80 B");
static_assert(sizeof(CrtTraceArgs) == 24, "TraceArgs must be 24 B");
#endif
#endif

#endif /* CRT_TYPES_H */
