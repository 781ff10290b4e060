// ResourceManager.cpp -- host arenas + uploads through the C-ABI (reference: ResourceManager.cpp:1-319).
//
// Call order is the reference's (Engine.cpp:56-80): PrepareMeshes -> ImportTexture(skybox, must be
// texture index 2) -> ImportMesh... -> PushMeshesToGPU (BuildBVH + uploads) -> PushTexturesToGPU.
// Deliberate differences, all in the "latent upload bug" class of SURVEY.md 8b (they only bite on
// a second PushMeshesToGPU upstream): offsets use sizeof(uint) for the root table, the triangle
// source pointer is advanced, counts are not double-added, and only meshes not yet pushed are
// built. With a single push the uploaded bytes are identical to the reference's.
// The host texel arena mirrors the device pool byte for byte (upstream copies to a wrong offset,
// ResourceManager.cpp:205, and never stores the two default texels on the host).
#include "ResourceManager.hpp"
#include "AssetManager.hpp"
#include "JpegDecode.hpp"
#include "../../include/crt_api.h"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <dirent.h>
#include <strings.h>
#include <sys/stat.h>

constexpr size_t MAX_TEXTURE_MEMORY = CRT_MAX_TEXTURE_BYTES;
constexpr size_t MAX_TRIANGLES = CRT_MAX_TRIANGLES;
// The reference's host arena holds MAX_TRIANGLES nodes (ResourceManager.cpp:35,151) although its builder
// emits ~1.8 nodes per triangle and the device pool is twice that (ResourceManager.cpp:159); the host
// arena here matches the device pool so that a ~1 M-triangle scene (config 4) fits, and overflow is an error.
constexpr size_t MAX_BVHNODES = MAX_TRIANGLES * 2;
constexpr size_t MaxTextures = CRT_MAX_TEXTURES;
constexpr size_t MaxMaterials = CRT_MAX_MATERIALS;
constexpr size_t MaxMeshes = CRT_MAX_MESHES;

RGB8* g_TexturePixels = nullptr;
BVHNode* g_BVHNodes = nullptr;
Tri* g_Triangles = nullptr;
Material* g_Materials = nullptr;
Texture* g_Textures = nullptr;
uint* g_BVHIndices = nullptr;

namespace {
    Material m_Materials[MaxMaterials];
    Texture m_Textures[MaxTextures];
    uint m_BVHIndices[MaxMeshes];

    ObjMesh* meshObjs[MaxMeshes];
    MeshInfo meshInfos[MaxMeshes];
    TextureInfo textureInfos[MaxTextures];
    MaterialInfo materialInfos[MaxMaterials];
    std::string meshDirs[MaxMeshes];

    size_t numTriangles = 0;
    size_t lastTextureOffset = 0; // bytes
    size_t lastTriangleCount = 0;
    size_t lastBVHIndex = 0;
    uint numberOfBVH = 0;
    int numTextures = 0, numMeshes = 0, numMaterials = 0;
    bool deviceOn = true;
    bool deviceBuild = false;
    bool initialized = false;
    int lastError = 0;

    void note(int rc, const char* what)
    {
        if (rc != 0) { lastError = rc; std::fprintf(stderr, "[ResourceManager] %s failed: %s\n", what, crt_error_string(rc)); }
    }

    TextureHandle store_texture(const char* path, int width, int height, const unsigned char* rgb)
    {
        if (numTextures >= (int)MaxTextures) { std::fprintf(stderr, "[ResourceManager] texture table full\n"); lastError = CRT_E_OUT_OF_RANGE; return 0; }
        const size_t numBytes = (size_t)width * (size_t)height * 3;
        if (lastTextureOffset + numBytes >= MAX_TEXTURE_MEMORY) { // ResourceManager.cpp:198
            std::fprintf(stderr, "[ResourceManager] texture importing failed! MAX_TEXTURE_MEMORY is not enough!\n");
            lastError = CRT_E_OUT_OF_RANGE; return 0;
        }
        Texture& texture = g_Textures[numTextures];
        TextureInfo& info = textureInfos[numTextures];
        info.path = strdup(path);
        const char* slash = std::strrchr(info.path, '/');
        info.name = slash ? const_cast<char*>(slash + 1) : info.path;
        info.glTextureIcon = 0;
        texture.width = width; texture.height = height; texture.padd = 0;
        if (deviceOn) note(crt_upload_texels(rgb, lastTextureOffset, numBytes), "crt_upload_texels");
        std::memcpy(reinterpret_cast<unsigned char*>(g_TexturePixels) + lastTextureOffset, rgb, numBytes);
        texture.offset = (int)(lastTextureOffset / 3);
        lastTextureOffset += numBytes;
        return (TextureHandle)numTextures++;
    }

    std::string assetRoot;           // SetAssetRoot: stands in for the reference's working directory (the folder that holds Assets/)

    // The reference runs on Windows: `map_Kd Assets/sponza/01_ST_KP.JPG` opens `01_St_kp.JPG` there. Resolve a path the
    // same way: every component that does not exist as written is matched ignoring case against its directory.
    bool exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
    bool resolve_ignoring_case(const std::string& path, std::string& out)
    {
        if (exists(path)) { out = path; return true; }
        std::string cur = (!path.empty() && path[0] == '/') ? "/" : "";
        size_t i = 0;
        while (i < path.size()) {
            while (i < path.size() && path[i] == '/') ++i;
            size_t j = i;
            while (j < path.size() && path[j] != '/') ++j;
            if (j == i) break;
            const std::string comp = path.substr(i, j - i);
            std::string next = cur + comp;
            if (!exists(next)) {
                DIR* d = ::opendir(cur.empty() ? "." : cur.c_str());
                if (!d) return false;
                bool found = false;
                while (struct dirent* e = ::readdir(d))
                    if (::strcasecmp(e->d_name, comp.c_str()) == 0) { next = cur + e->d_name; found = true; break; }
                ::closedir(d);
                if (!found) return false;
            }
            cur = next;
            if (j < path.size()) cur += '/';
            i = j;
        }
        out = cur;
        return exists(out);
    }
    // as written (the reference resolves against its CWD), then under the asset root, then next to the mesh file
    bool resolve_asset(const std::string& p, const std::string& meshDir, std::string& out)
    {
        if (resolve_ignoring_case(p, out)) return true;
        if (!assetRoot.empty() && resolve_ignoring_case(assetRoot + "/" + p, out)) return true;
        if (!meshDir.empty() && resolve_ignoring_case(meshDir + p, out)) return true;
        const size_t slash = p.find_last_of('/');
        if (!meshDir.empty() && slash != std::string::npos && resolve_ignoring_case(meshDir + p.substr(slash + 1), out)) return true;
        return false;
    }

    bool read_file(const char* path, std::vector<unsigned char>& bytes)
    {
        FILE* f = std::fopen(path, "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END);
        const long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        bool ok = n >= 0;
        if (ok) { bytes.resize((size_t)n); ok = n == 0 || std::fread(bytes.data(), 1, (size_t)n, f) == (size_t)n; }
        std::fclose(f);
        return ok;
    }

    bool read_ppm(const char* path, int& w, int& h, std::string& pixels)
    {
        FILE* f = std::fopen(path, "rb");
        if (!f) return false;
        char magic[3] = { 0 };
        int maxv = 0;
        auto next_int = [&](int& v) {
            int c = std::fgetc(f);
            for (;;) {
                while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = std::fgetc(f);
                if (c == '#') { while (c != '\n' && c != EOF) c = std::fgetc(f); continue; }
                break;
            }
            if (c < '0' || c > '9') return false;
            v = 0;
            while (c >= '0' && c <= '9') { v = v * 10 + (c - '0'); c = std::fgetc(f); }
            return true; // exactly one whitespace byte after the number has been consumed
        };
        bool ok = std::fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && magic[1] == '6' && next_int(w) && next_int(h) && next_int(maxv) && maxv == 255 && w > 0 && h > 0;
        if (ok) {
            pixels.resize((size_t)w * (size_t)h * 3);
            ok = std::fread(&pixels[0], 1, pixels.size(), f) == pixels.size();
        }
        std::fclose(f);
        return ok;
    }
}

namespace ResourceManager
{
    MeshInfo GetMeshInfo(MeshHandle handle) { return meshInfos[handle]; }
    TextureInfo GetTextureInfo(TextureHandle handle) { return textureInfos[handle]; }
    Material& EditMaterial(MaterialHandle handle) { return g_Materials[handle]; }
    ushort GetNumMeshes() { return (ushort)numMeshes; }
    int LastError() { return lastError; }
}

Material* ResourceManager::CreateMaterial(MaterialHandle* materialPtr, int count)
{
    if (numMaterials + count > (int)MaxMaterials) { lastError = CRT_E_OUT_OF_RANGE; return nullptr; }
    Material* ptr = g_Materials + numMaterials;
    if (materialPtr) *materialPtr = (MaterialHandle)numMaterials; // upstream leaves the handle unset ("Todo fix", ResourceManager.cpp:132)
    numMaterials += count;
    return ptr;
}

void ResourceManager::PushMaterialsToGPU()
{
    if (deviceOn) note(crt_upload_materials(g_Materials, 0, (size_t)numMaterials), "crt_upload_materials");
}

void ResourceManager::Initialize(bool deviceUploads)
{
    if (initialized) return;
    deviceOn = deviceUploads;
    g_Triangles = (Tri*)std::aligned_alloc(64, MAX_TRIANGLES * sizeof(Tri));
    g_BVHNodes = (BVHNode*)std::aligned_alloc(64, MAX_BVHNODES * sizeof(BVHNode));
    g_TexturePixels = (RGB8*)std::malloc(MAX_TEXTURE_MEMORY * 2);
    g_Materials = m_Materials; g_Textures = m_Textures; g_BVHIndices = m_BVHIndices;
    std::memset(m_Materials, 0, sizeof m_Materials);
    std::memset(m_Textures, 0, sizeof m_Textures);
    std::memset(m_BVHIndices, 0, sizeof m_BVHIndices);
    AssetManager_Initialize();
    numTriangles = 0; lastTriangleCount = 0; lastBVHIndex = 0; numberOfBVH = 0;
    numTextures = 0; numMeshes = 0; numMaterials = 0; lastError = 0;
    ResetBVHNodeCounter();
    SetBVHNodeCapacity(MAX_BVHNODES);

    // default textures: white, black (ResourceManager.cpp:168-177). crt_init already placed the
    // same six bytes in the device pool.
    g_Textures[0].width = 1; g_Textures[1].width = 1;
    g_Textures[0].height = 1; g_Textures[1].height = 1;
    g_Textures[0].offset = 0; g_Textures[1].offset = 3; // sic: upstream stores the BYTE offset here (texel 3 = 2nd skybox texel); kept, the black texture only feeds the unused specular fetch
    const unsigned char def[6] = { 0xFF, 0xFF, 0xFF, 0, 0, 0 };
    std::memcpy(g_TexturePixels, def, 6);
    lastTextureOffset = 6; numTextures = 2;
    initialized = true;
}

void ResourceManager::SetAssetRoot(const char* dir) { assetRoot = dir ? dir : ""; while (!assetRoot.empty() && assetRoot.back() == '/') assetRoot.pop_back(); }

// ResourceManager.cpp:180-222: stbi_load(path, &w, &h, &channels, 3) -> RGB8 into the texel arena. JPEG files go through
// JpegDecode.cpp (same bytes as the reference's stb_image); binary PPM (P6) stays accepted for the synthetic scenes.
TextureHandle ResourceManager::ImportTexture(const char* path)
{
    std::string resolved;
    if (!resolve_asset(path, std::string(), resolved)) {
        std::fprintf(stderr, "[ResourceManager] texture importing failed! file is not exist: %s\n", path);
        lastError = CRT_E_BAD_ARGUMENT; return 0;
    }
    std::vector<unsigned char> file;
    if (!read_file(resolved.c_str(), file) || file.size() < 4) {
        std::fprintf(stderr, "[ResourceManager] texture importing failed! cannot read: %s\n", resolved.c_str());
        lastError = CRT_E_BAD_ARGUMENT; return 0;
    }
    if (file[0] == 0xFF && file[1] == 0xD8) {
        JpegInfo info; std::vector<unsigned char> rgb; const char* why = nullptr;
        if (!JpegDecodeRGB8(file.data(), file.size(), &info, rgb, &why)) {
            std::fprintf(stderr, "[ResourceManager] texture importing failed! keep in mind texture must be .jpeg (%s: %s)\n", resolved.c_str(), why ? why : "?");
            lastError = CRT_E_BAD_ARGUMENT; return 0;
        }
        return store_texture(path, info.width, info.height, rgb.data());
    }
    int w = 0, h = 0; std::string px;
    if (!read_ppm(resolved.c_str(), w, h, px)) {
        std::fprintf(stderr, "[ResourceManager] texture importing failed! keep in mind texture must be .jpeg (or binary PPM): %s\n", resolved.c_str());
        lastError = CRT_E_BAD_ARGUMENT; return 0;
    }
    return store_texture(path, w, h, reinterpret_cast<const unsigned char*>(px.data()));
}

TextureHandle ResourceManager::ImportTextureRGB8(const char* name, int width, int height, const unsigned char* rgb)
{
    if (!rgb || width <= 0 || height <= 0) { lastError = CRT_E_BAD_ARGUMENT; return 0; }
    return store_texture(name ? name : "memory", width, height, rgb);
}

void ResourceManager::PrepareMeshes() // ResourceManager.cpp:224-232
{
    Material* firstMaterial = g_Materials + 0;
    firstMaterial->color = 0x00FF0000u | (80 << 16) | (55);
    firstMaterial->specularColor = 250 | (228 << 8) | (210 << 16);
    firstMaterial->shininess = crtmath::ConvertFloatToHalf(1.2f);
    firstMaterial->roughness = crtmath::ConvertFloatToHalf(0.8f);
    firstMaterial->albedoTextureIndex = 0u; firstMaterial->specularTextureIndex = 1u; numMaterials++;
}

void ResourceManager::PushTexturesToGPU()
{
    if (deviceOn) note(crt_upload_texture_table(g_Textures, MaxTextures), "crt_upload_texture_table");
}

MeshHandle ResourceManager::ImportMesh(const char* path) // ResourceManager.cpp:241-276
{
    if (numMeshes >= (int)MaxMeshes) { lastError = CRT_E_OUT_OF_RANGE; return 0; }
    MeshInfo& meshInfo = meshInfos[numMeshes];
    ObjMesh* mesh = AssetManager_ImportMesh(path, g_Triangles + numTriangles, MAX_TRIANGLES - numTriangles);
    if (!mesh) { lastError = CRT_E_BAD_ARGUMENT; return 0; }
    if (numTriangles + (size_t)mesh->numTris > MAX_TRIANGLES || numMaterials + mesh->numMaterials > (int)MaxMaterials) {
        std::fprintf(stderr, "[ResourceManager] mesh does not fit the arenas: %s\n", path);
        AssetManager_DestroyMesh(mesh); lastError = CRT_E_OUT_OF_RANGE; return 0;
    }
    meshInfo.materialStart = (ushort)(mesh->numMaterials ? numMaterials : 0);
    meshInfo.triangleStart = (uint)numTriangles;
    meshInfo.numTriangles = (uint)mesh->numTris;
    meshInfo.numMaterials = (ushort)mesh->numMaterials;
    meshInfo.path = strdup(path);
    meshObjs[numMeshes] = mesh;
    std::string dir(path);
    size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);

    for (int i = 0; i < mesh->numMaterials; ++i) {
        ObjMaterial& objMaterial = mesh->materials[i];
        materialInfos[numMaterials].name = mesh->mtlText + objMaterial.name;
        Material& material = g_Materials[numMaterials++];
        material.color = objMaterial.diffuseColor;
        material.specularColor = objMaterial.specularColor;
        material.shininess = objMaterial.shininess;
        material.roughness = objMaterial.roughness;
        auto import_map = [&](int offset) -> ushort {
            if (!offset) return 0;
            // upstream resolves map paths against the process CWD (ignoring case: Windows); also try the asset root and
            // the mesh file's directory
            std::string p(mesh->mtlText + offset), found;
            while (!p.empty() && (p.back() == ' ' || p.back() == '\t' || p.back() == '\r')) p.pop_back();
            if (resolve_asset(p, dir, found)) p = found;
            return ImportTexture(p.c_str());
        };
        material.albedoTextureIndex = import_map(objMaterial.diffusePath);
        material.specularTextureIndex = import_map(objMaterial.specularPath);
    }
    numTriangles += meshInfo.numTriangles;
    return (MeshHandle)numMeshes++;
}

void ResourceManager::SetDeviceBVHBuild(bool enabled) { deviceBuild = enabled; }

void ResourceManager::PushMeshesToGPU() // ResourceManager.cpp:280-300
{
    if (numMeshes == (int)numberOfBVH) return;
    if (deviceOn && deviceBuild) {
        // upload the triangles as imported, build on the device, read the reordered triangles / nodes / roots back so the
        // host arenas (CPU_RayCast, tools) hold exactly what the device renders from
        const int newMeshes = numMeshes - (int)numberOfBVH;
        uint counts[MaxMeshes];
        for (int m = 0; m < newMeshes; ++m) counts[m] = meshInfos[numberOfBVH + m].numTriangles;
        const size_t addedTriangleSize = (numTriangles - lastTriangleCount) * sizeof(Tri);
        uint numNodesUsed = 0;
        int rc = crt_upload_triangles(g_Triangles + lastTriangleCount, lastTriangleCount * sizeof(Tri), addedTriangleSize);
        int built = rc;
        if (rc == 0) built = crt_build_bvh(lastTriangleCount, counts, newMeshes, lastBVHIndex, numberOfBVH, &numNodesUsed);
        // Refusals: a size beyond the builder's scratch layout or a failed consistency check (OUT_OF_RANGE, UNSUPPORTED), an argument it does not
        // take (BAD_ARGUMENT: e.g. a mesh without triangles), or its own scratch / second triangle pool not fitting (hipErrorOutOfMemory = 2, not sticky:
        // the device is as usable as before) -- ADVICE r5. The host arenas still hold the triangles as imported, so the host BuildBVH below takes
        // over and its uploads replace whatever the device build left behind.
        const bool refused = built == CRT_E_OUT_OF_RANGE || built == CRT_E_UNSUPPORTED || built == CRT_E_BAD_ARGUMENT || built == 2 /* hipErrorOutOfMemory */;
        if (rc == 0 && refused) {
            std::fprintf(stderr, "[ResourceManager] crt_build_bvh refused (%d: %s): building on the host\n", built, crt_error_string(built));
        } else if (rc == 0 && built != 0) {
            // anything else (a sticky HIP error: a faulting kernel, a failed stream synchronisation) is reported where it happened, not papered over:
            // lastError is set and the arenas stay as they were (nothing of this push is marked as built)
            note(built, "crt_build_bvh");
            return;
        } else {
            if (rc == 0) rc = crt_download_triangles(g_Triangles + lastTriangleCount, lastTriangleCount * sizeof(Tri), addedTriangleSize);
            if (rc == 0) rc = crt_download_bvh_nodes(g_BVHNodes + lastBVHIndex, lastBVHIndex * sizeof(BVHNode), sizeof(BVHNode) * numNodesUsed);
            if (rc == 0) rc = crt_download_bvh_roots(g_BVHIndices + numberOfBVH, numberOfBVH, (size_t)newMeshes);
            note(rc, "crt_build_bvh");
            if (rc != 0) return;
            note(crt_upload_materials(g_Materials, 0, (size_t)numMaterials), "crt_upload_materials");
            AdvanceBVHNodeCounter(numNodesUsed);
            numberOfBVH = (uint)numMeshes;
            lastBVHIndex += numNodesUsed;
            lastTriangleCount = numTriangles;
            return;
        }
    }
    const uint numNodesUsed = BuildBVH(g_Triangles + lastTriangleCount, meshInfos + numberOfBVH, numMeshes - (int)numberOfBVH,
                                       g_BVHNodes, g_BVHIndices + numberOfBVH);
    if (BVHBuildOverflowed()) {
        std::fprintf(stderr, "[ResourceManager] BVH node arena (%zu nodes) exhausted\n", MAX_BVHNODES);
        lastError = CRT_E_OUT_OF_RANGE;
        return;
    }
    // BuildBVH works on a triangle pointer that starts at this push's first triangle: rebase leaves
    if (lastTriangleCount) {
        for (size_t k = lastBVHIndex; k < lastBVHIndex + numNodesUsed; ++k)
            if (g_BVHNodes[k].triCount > 0) g_BVHNodes[k].leftFirst += (uint)lastTriangleCount;
    }
    if (deviceOn) {
        const size_t addedTriangleSize = (numTriangles - lastTriangleCount) * sizeof(Tri);
        note(crt_upload_triangles(g_Triangles + lastTriangleCount, lastTriangleCount * sizeof(Tri), addedTriangleSize), "crt_upload_triangles");
        note(crt_upload_bvh_nodes(g_BVHNodes + lastBVHIndex, lastBVHIndex * sizeof(BVHNode), sizeof(BVHNode) * numNodesUsed), "crt_upload_bvh_nodes");
        note(crt_upload_bvh_roots(g_BVHIndices + numberOfBVH, numberOfBVH, (size_t)(numMeshes - (int)numberOfBVH)), "crt_upload_bvh_roots");
        note(crt_upload_materials(g_Materials, 0, (size_t)numMaterials), "crt_upload_materials");
    }
    numberOfBVH = (uint)numMeshes;
    lastBVHIndex += numNodesUsed;
    lastTriangleCount = numTriangles;
}

void ResourceManager::Destroy() {}

void ResourceManager::Finalize()
{
    if (!initialized) return;
    Destroy();
    while (numMeshes--) { AssetManager_DestroyMesh(meshObjs[numMeshes]); std::free(const_cast<char*>(meshInfos[numMeshes].path)); }
    while (numTextures-- > 2) std::free(textureInfos[numTextures].path);
    std::free(g_TexturePixels); std::free(g_Triangles); std::free(g_BVHNodes);
    g_TexturePixels = nullptr; g_Triangles = nullptr; g_BVHNodes = nullptr;
    numMeshes = 0; numTextures = 0; numMaterials = 0;
    initialized = false;
}

namespace ResourceManager { size_t TexelBytesUsed() { return lastTextureOffset; } int NumMaterials() { return numMaterials; } int NumTextures() { return numTextures; } size_t NumTriangles() { return numTriangles; } size_t NumNodes() { return lastBVHIndex; } }
