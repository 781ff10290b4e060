// AssetManager.hpp -- OBJ/MTL importer of the reference (AssetManager.hpp:1-42, AssetManager.cpp:90-289).
// OBJ subset -> Tri[] + materials, and the `.clm` mesh cache with its QuickLZ stream (MeshCache.cpp).
#pragma once
#include "ResourceManager.hpp"

struct ObjMaterial {
    int name;                          // offset into mtlText
    unsigned diffuseColor, specularColor;
    half shininess, roughness;
    int diffusePath, specularPath;     // offsets into mtlText, 0 = none
};

struct ObjMesh {
    char* name;
    Tri* tris;
    int numTris;
    ObjMaterial materials[32];
    int numMaterials;
    char* mtlText;
    unsigned mtlSize;                  // bytes in mtlText (upstream passes it alongside, AssetManager.cpp:287)
};

// Returns nullptr (and logs to stderr) on failure where the reference calls exit(0).
ObjMesh* AssetManager_ImportMesh(const char* path, Tri* triArena, size_t maxTris = 1000000);   // maxTris: room left in the arena
void AssetManager_DestroyMesh(ObjMesh* mesh);
// `.clm` cache (AssetManager.cpp:291-361): ImportMesh prefers `<stem>.clm` when it is not older than the OBJ and writes
// it after an OBJ import, as upstream does (upstream does not compare dates). On by default; maxTris = room in the arena.
void AssetManager_SetMeshCache(bool enabled);
bool AssetManager_MeshCacheEnabled();
bool AssetManager_SaveMeshToDisk(const char* path, const ObjMesh* mesh);
ObjMesh* AssetManager_LoadMeshFromDisk(const char* path, Tri* triArena, size_t maxTris);
bool MeshCache_IsFresh(const char* cache, const char* source);
size_t MeshCache_QlzDecompress(const unsigned char* src, size_t srcLen, unsigned char* dst, size_t dstCap);
size_t MeshCache_QlzStore(const unsigned char* src, size_t size, unsigned char* dst);
size_t MeshCache_QlzCompress(const unsigned char* src, size_t size, unsigned char* dst);   // dst: size + 400 bytes
void AssetManager_Initialize();
void AssetManager_Destroy();

// Writer for the same subset (used by the synthetic scene generators). Indices are 0-based.
// faces: 9 ints per triangle {v,vt,vn}x3; faceMaterial: material slot per triangle (sorted runs
// become `usemtl` groups); materialNames[i] is the MTL name of slot i.
int AssetManager_WriteObj(const char* path, const float* positions, int numPositions, const float* uvs, int numUvs,
                          const float* normals, int numNormals, const int* faces, const int* faceMaterial, int numFaces,
                          const char* const* materialNames, int numMaterials);
