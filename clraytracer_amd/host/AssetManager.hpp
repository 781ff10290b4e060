// AssetManager.hpp -- OBJ/MTL importer of the reference (AssetManager.hpp:1-42, AssetManager.cpp:90-289).
// In scope: OBJ subset -> Tri[] + materials. The .clm/QuickLZ cache is a "next" row.
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
};

// Returns nullptr (and logs to stderr) on failure where the reference calls exit(0).
ObjMesh* AssetManager_ImportMesh(const char* path, Tri* triArena);
void AssetManager_DestroyMesh(ObjMesh* mesh);
void AssetManager_Initialize();
void AssetManager_Destroy();

// Writer for the same subset (used by the synthetic scene generators). Indices are 0-based.
// faces: 9 ints per triangle {v,vt,vn}x3; faceMaterial: material slot per triangle (sorted runs
// become `usemtl` groups); materialNames[i] is the MTL name of slot i.
int AssetManager_WriteObj(const char* path, const float* positions, int numPositions, const float* uvs, int numUvs,
                          const float* normals, int numNormals, const int* faces, const int* faceMaterial, int numFaces,
                          const char* const* materialNames, int numMaterials);
