// ResourceManager.hpp -- host mirror of the reference's scene/resource layer (ResourceManager.hpp:1-110).
// Same names, argument meaning and limits; OpenCL types are gone (Initialize takes no cl_context),
// device pools live behind the C-ABI in libcrt_hip.so.
#pragma once
#include "Math.hpp"

typedef CrtBVHNode BVHNode;
typedef CrtRGB8 RGB8;
typedef CrtTexture Texture;
typedef CrtMaterial Material;
typedef CrtTri Tri;

struct MeshInfo {
    uint numTriangles;
    uint triangleStart;
    ushort materialStart;
    ushort numMaterials;
    const char* path;
};
struct TextureInfo { char* path; char* name; uint glTextureIcon; };
struct MaterialInfo { char* name; };

typedef ushort TextureHandle;
typedef ushort MeshHandle;
typedef ushort MaterialHandle;

namespace ResourceManager
{
    // path: a JPEG (decoded to the same RGB8 bytes as the reference's stbi_load(path, .., 3), ResourceManager.cpp:193;
    // JpegDecode.hpp) or a binary PPM (P6, maxval 255; the synthetic scenes). The path is tried as written, then under
    // the asset root, ignoring case per component (the reference's assets rely on Windows path semantics).
    TextureHandle ImportTexture(const char* path);
    // Extension: where relative asset paths (texture paths inside .mtl / .clm files, e.g. "Assets/sponza/KAMEN.JPG")
    // resolve when they are not found relative to the working directory -- the folder that contains `Assets/`.
    void SetAssetRoot(const char* dir);
    // Extension: import from memory (tightly packed RGB8, width*height*3 bytes).
    TextureHandle ImportTextureRGB8(const char* name, int width, int height, const unsigned char* rgb);
    MeshHandle ImportMesh(const char* path);

    constexpr TextureHandle  WhiteTexture = 0;
    constexpr TextureHandle  BlackTexture = 1;
    constexpr MaterialHandle NoneMaterial = 0;
    constexpr MaterialHandle DefaultMaterial = 0xFFFF;

    Material* CreateMaterial(MaterialHandle* handle, int count = 1);
    Material& EditMaterial(MaterialHandle handle);

    void PrepareMeshes();
    void PushMeshesToGPU();
    void PushMaterialsToGPU();
    void PushTexturesToGPU();
    ushort GetNumMeshes();

    MeshInfo GetMeshInfo(MeshHandle handle);
    TextureInfo GetTextureInfo(TextureHandle handle);

    // deviceUploads=false keeps everything on the host (importer / BVH / CPU_RayCast only);
    // any attempt to render then fails loudly.
    void Initialize(bool deviceUploads = true);
    // BuildBVH on the device (crt_build_bvh: same bytes as the host builder) instead of on the host; the host arenas are
    // then filled from the device. Off by default, like upstream (ResourceManager.cpp:282 builds on the CPU).
    void SetDeviceBVHBuild(bool enabled);
    void Destroy();
    void Finalize();

    // last non-zero C-ABI error seen by any upload (0 = none); the reference asserts instead
    int LastError();
}

// host arenas, as the reference's extern globals (ResourceManager.cpp:49-55)
extern RGB8* g_TexturePixels;
extern BVHNode* g_BVHNodes;
extern Tri* g_Triangles;
extern Material* g_Materials;
extern Texture* g_Textures;
extern uint* g_BVHIndices;

uint BuildBVH(Tri* tris, MeshInfo* meshes, int numMeshes, BVHNode* nodes, uint* bvhIndices);
void SetBVHNodeCapacity(size_t nodes); // 0 = unchecked
bool BVHBuildOverflowed();
void AdvanceBVHNodeCounter(uint nodes); // nodes written by a builder other than BuildBVH (crt_build_bvh)
void ResetBVHNodeCounter(); // the reference's file-static totalNodesUsed (BVH.cpp:49) is never reset; Finalize() does
