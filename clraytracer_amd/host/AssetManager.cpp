// AssetManager.cpp -- OBJ/MTL subset importer and writer (reference: AssetManager.cpp:10-35, 90-289).
//
// Grammar accepted (exactly what the reference's hand-rolled parser understands):
//   MTL: newmtl <name> | Ns f | d f | Kd f f f | Ks f f f | map_Kd <path> | map_Ks <path> | # ...
//   OBJ: v f f f | vt f f | vn f f f | usemtl <name> | f v/vt/vn v/vt/vn v/vt/vn | o/s/mtllib/# lines
//   floats are [-]digits[.digits] (no exponent, AssetManager.cpp:13-35); indices positive, 1-based.
// Semantics kept: v flipped to 1-v (AssetManager.cpp:271); float->half by the reference's
// round-half-up trick (Math.hpp:190); material lookup through the 512-slot name hash with
// WangHash seeding (AssetManager.cpp:140-147,239-243); Ns clamped to [0,100]/50; Kd packed with
// truncation (Math.hpp:237). Unlike the reference this parser never reads past the buffer, returns
// nullptr instead of exit(0), and does not write a .clm cache.
#include "AssetManager.hpp"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace {

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// AssetManager.cpp:13-35, accumulated in double then narrowed, as upstream
const char* parse_float(float* out, const char* p, const char* end)
{
    while (p < end && is_blank(*p)) ++p;
    double sign = 1.0;
    if (p < end && *p == '-') { sign = -1.0; ++p; }
    double num = 0.0;
    while (p < end && is_digit(*p)) num = 10.0 * num + (double)(*p++ - '0');
    if (p < end && *p == '.') ++p;
    double fra = 0.0, div = 1.0;
    while (p < end && is_digit(*p)) { fra = 10.0f * fra + (double)(*p++ - '0'); div *= 10.0f; }
    num += fra / div;
    *out = (float)(sign * num);
    return p;
}

inline const char* skip_line(const char* p, const char* end)
{
    while (p < end && *p != '\n') ++p;
    return p < end ? p + 1 : p;
}
inline const char* skip_space_and_newlines(const char* p, const char* end)
{
    while (p < end && (*p == '\n' || is_blank(*p))) ++p;
    return p;
}

// Random.hpp:24-30
inline unsigned wang_hash(unsigned s)
{
    s = (s ^ 61u) ^ (s >> 16u);
    s *= 9; s = s ^ (s >> 4u);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15u);
    return s;
}

// AssetManager.cpp:140-142 / 239-241: seed from the first three bytes, then the sdbm step per byte
unsigned material_name_hash(const char* p, const char* end, const char** after)
{
    auto at = [&](int k) -> unsigned { return (p + k < end) ? (unsigned)(unsigned char)p[k] : 0u; };
    unsigned hash = wang_hash(at(0) | (at(1) << 8) | (at(2) << 16));
    while (p < end && *p != '\n' && !is_blank(*p)) {
        hash = (unsigned)(unsigned char)*p++ + (hash << 6) + (hash << 16) - hash;
    }
    *after = p;
    return hash;
}

bool read_file(const std::string& path, std::vector<char>& out)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize((size_t)(sz > 0 ? sz : 0) + 1);
    size_t got = sz > 0 ? std::fread(out.data(), 1, (size_t)sz, f) : 0;
    std::fclose(f);
    out[got] = '\0';
    out.resize(got + 1);
    // skip a UTF-8 BOM (the reference's SkipBOM)
    if (got >= 3 && (unsigned char)out[0] == 0xEF && (unsigned char)out[1] == 0xBB && (unsigned char)out[2] == 0xBF)
        out.erase(out.begin(), out.begin() + 3);
    return true;
}

std::string with_extension(const char* path, const char* ext)
{
    std::string s(path);
    size_t dot = s.find_last_of('.');
    size_t slash = s.find_last_of('/');
    if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) s.resize(dot);
    s += ".";
    s += ext;
    return s;
}

bool parse_mtl(ObjMesh* mesh, size_t msz, unsigned char materialMap[512])
{
    char* text = mesh->mtlText;
    const char* end = text + msz;
    char* curr = text;
    ObjMaterial* mat = nullptr;
    while (curr < end && *curr) {
        if (*curr == '#') { curr = const_cast<char*>(skip_line(curr, end)); continue; }
        curr = const_cast<char*>(skip_space_and_newlines(curr, end));
        if (curr >= end || !*curr) break;
        if (curr[0] == 'n' && curr + 7 < end && curr[1] == 'e' && curr[2] == 'w') {
            curr += 7; // "newmtl "
            if (mesh->numMaterials >= 32) { std::fprintf(stderr, "[AssetManager] more than 32 materials in one mesh\n"); return false; }
            mat = mesh->materials + mesh->numMaterials;
            mat->specularColor = ~0u; mat->diffuseColor = ~0u;
            mat->shininess = crtmath::ConvertFloatToHalf(2.2f);
            mat->roughness = crtmath::ConvertFloatToHalf(0.6f);
            mat->diffusePath = 0; mat->specularPath = 0;
            mat->name = (int)(curr - text);
            const char* after;
            unsigned hash = material_name_hash(curr, end, &after);
            curr = const_cast<char*>(after);
            if (curr < end) *curr++ = '\0';
            if (materialMap[hash & 511]) { std::fprintf(stderr, "[AssetManager] material name hash collision\n"); return false; }
            materialMap[hash & 511] = (unsigned char)mesh->numMaterials++;
        }
        else if (curr[0] == 'N' && curr[1] == 's' && mat) {
            float f; curr = const_cast<char*>(parse_float(&f, curr + 2, end));
            f = (f < 0.0f ? 0.0f : (f > 100.0f ? 100.0f : f)) / 50.0f;
            mat->shininess = crtmath::ConvertFloatToHalf(f);
        }
        else if (curr[0] == 'd' && mat) {
            float f; curr = const_cast<char*>(parse_float(&f, curr + 2, end));
            mat->roughness = crtmath::ConvertFloatToHalf(f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f));
        }
        else if (curr[0] == 'K' && (curr[1] == 'd' || curr[1] == 's') && mat) {
            const bool diffuse = curr[1] == 'd';
            float c[3];
            curr = const_cast<char*>(parse_float(c + 0, curr + 2, end));
            curr = const_cast<char*>(parse_float(c + 1, curr, end));
            curr = const_cast<char*>(parse_float(c + 2, curr, end));
            (diffuse ? mat->diffuseColor : mat->specularColor) = crtmath::PackColorRGBU32(c);
        }
        else if (curr[0] == 'm' && curr + 7 < end && curr[4] == 'K' && (curr[5] == 'd' || curr[5] == 's') && mat) {
            const bool diffuse = curr[5] == 'd';
            curr += 7; // "map_Kd "
            (diffuse ? mat->diffusePath : mat->specularPath) = (int)(curr - text);
            while (curr < end && *curr != '\n' && *curr != '\r') ++curr;
            if (curr < end) *curr++ = '\0';
        }
        else curr = const_cast<char*>(skip_line(curr, end));
    }
    return true;
}

} // namespace

void AssetManager_Initialize() {}
void AssetManager_Destroy() {}

static ObjMesh* AssetManager_ImportObj(const char* path, Tri* triArena, size_t maxTris);

// AssetManager.cpp:363-381: prefer `<stem>.clm`, else parse the OBJ and leave a `.clm` behind for the next run
ObjMesh* AssetManager_ImportMesh(const char* path, Tri* triArena, size_t maxTris)
{
    const std::string clm = with_extension(path, "clm");
    if (AssetManager_MeshCacheEnabled() && MeshCache_IsFresh(clm.c_str(), path)) {
        if (ObjMesh* cached = AssetManager_LoadMeshFromDisk(clm.c_str(), triArena, maxTris)) return cached;
        std::fprintf(stderr, "[AssetManager] ignoring %s, importing %s\n", clm.c_str(), path);
    }
    ObjMesh* mesh = AssetManager_ImportObj(path, triArena, maxTris);
    if (mesh && AssetManager_MeshCacheEnabled() && !AssetManager_SaveMeshToDisk(clm.c_str(), mesh))
        std::fprintf(stderr, "[AssetManager] could not write %s (continuing without the cache)\n", clm.c_str());
    return mesh;
}

static ObjMesh* AssetManager_ImportObj(const char* path, Tri* triArena, size_t maxTris)
{
    std::vector<char> obj;
    if (!read_file(path, obj)) { std::fprintf(stderr, "[AssetManager] mesh file does not exist: %s\n", path); return nullptr; }

    ObjMesh* mesh = new ObjMesh;
    mesh->name = nullptr; mesh->tris = triArena; mesh->numTris = 0; mesh->numMaterials = 0; mesh->mtlText = nullptr; mesh->mtlSize = 0;
    std::memset(mesh->materials, 0, sizeof mesh->materials);

    unsigned char materialMap[512] = { 0 };
    std::vector<char> mtl;
    if (read_file(with_extension(path, "mtl"), mtl) && mtl.size() > 1) {
        mesh->mtlText = (char*)std::malloc(mtl.size());
        std::memcpy(mesh->mtlText, mtl.data(), mtl.size());
        mesh->mtlSize = (unsigned)mtl.size() - 1;               // the file's bytes, as upstream counts them (AssetManager.cpp:109,303); the buffer keeps a terminator behind them
        if (!parse_mtl(mesh, mtl.size() - 1, materialMap)) { AssetManager_DestroyMesh(mesh); return nullptr; }
    }

    std::vector<float> pos, uv, nrm;
    const char* curr = obj.data();
    const char* end = curr + obj.size() - 1;
    unsigned currentMaterial = 0;
    auto fail = [&](const char* why) { std::fprintf(stderr, "[AssetManager] %s: %s\n", path, why); AssetManager_DestroyMesh(mesh); return (ObjMesh*)nullptr; };

    while (curr < end && *curr) {
        if (*curr == '#') { curr = skip_line(curr, end); continue; }
        curr = skip_space_and_newlines(curr, end);
        if (curr >= end) break;
        if (curr[0] == 'v' && curr[1] == ' ') {
            float f[3]; curr += 2;
            for (int k = 0; k < 3; ++k) curr = parse_float(f + k, curr, end);
            pos.insert(pos.end(), f, f + 3);
            curr = skip_line(curr, end);
        }
        else if (curr[0] == 'v' && curr[1] == 't') {
            float f[2]; curr += 2;
            for (int k = 0; k < 2; ++k) curr = parse_float(f + k, curr, end);
            uv.insert(uv.end(), f, f + 2);
            curr = skip_line(curr, end);
        }
        else if (curr[0] == 'v' && curr[1] == 'n') {
            float f[3]; curr += 2;
            for (int k = 0; k < 3; ++k) curr = parse_float(f + k, curr, end);
            nrm.insert(nrm.end(), f, f + 3);
            curr = skip_line(curr, end);
        }
        else if (curr[0] == 'u' && curr[1] == 's' && curr[2] == 'e' && curr + 7 < end) {
            curr += 7; // "usemtl "
            const char* after;
            unsigned hash = material_name_hash(curr, end, &after);
            curr = skip_line(after, end);
            currentMaterial = materialMap[hash & 511];
        }
        else if (curr[0] == 'f' && curr[1] == ' ') {
            curr += 2;
            if (mesh->numTris + 1 >= 1000000) return fail("too many triangles for one mesh (>= 1,000,000)");
            if ((size_t)mesh->numTris + 1 > maxTris) return fail("the triangle arena is full");
            Tri* tri = mesh->tris + mesh->numTris++;
            float* vert[3] = { tri->v0, tri->v1, tri->v2 };
            half* uvp[3] = { tri->uv0, tri->uv1, tri->uv2 };
            half* np[3] = { tri->n0, tri->n1, tri->n2 };
            for (int k = 0; k < 3; ++k) {
                long idx[3] = { 0, 0, 0 };
                for (int q = 0; q < 3; ++q) {
                    while (curr < end && is_digit(*curr)) idx[q] = 10 * idx[q] + (*curr++ - '0');
                    if (curr < end && q < 2) { if (*curr != '/') return fail("face vertex is not v/vt/vn"); ++curr; }
                }
                while (curr < end && is_blank(*curr)) ++curr;
                const long p = idx[0] - 1, t = idx[1] - 1, n = idx[2] - 1;
                if (p < 0 || (size_t)p * 3 + 2 >= pos.size() + 0 || t < 0 || (size_t)t * 2 + 1 >= uv.size() + 0 || n < 0 || (size_t)n * 3 + 2 >= nrm.size() + 0)
                    return fail("face index out of range");
                std::memcpy(vert[k], &pos[(size_t)p * 3], 12);
                uvp[k][0] = crtmath::ConvertFloatToHalf(uv[(size_t)t * 2 + 0]);
                uvp[k][1] = crtmath::ConvertFloatToHalf(1.0f - uv[(size_t)t * 2 + 1]);
                np[k][0] = crtmath::ConvertFloatToHalf(nrm[(size_t)n * 3 + 0]);
                np[k][1] = crtmath::ConvertFloatToHalf(nrm[(size_t)n * 3 + 1]);
                np[k][2] = crtmath::ConvertFloatToHalf(nrm[(size_t)n * 3 + 2]);
            }
            tri->centroidx = tri->centroidy = tri->centroidz = 0.0f; // written by BuildBVH
            tri->materialIndex = (uint16_t)currentMaterial;
            curr = skip_line(curr, end);
        }
        else curr = skip_line(curr, end); // o / s / mtllib / anything unknown
    }
    if (mesh->numTris == 0) return fail("no triangles");
    return mesh;
}

void AssetManager_DestroyMesh(ObjMesh* mesh)
{
    if (!mesh) return;
    std::free(mesh->mtlText);
    delete mesh;
}

int AssetManager_WriteObj(const char* path, const float* positions, int numPositions, const float* uvs, int numUvs,
                          const float* normals, int numNormals, const int* faces, const int* faceMaterial, int numFaces,
                          const char* const* materialNames, int numMaterials)
{
    FILE* f = std::fopen(path, "wb");
    if (!f) return -1;
    std::vector<char> buf(1 << 22);
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    std::fprintf(f, "# synthetic scene, OBJ subset of CLRayTracer's importer\n");
    for (int i = 0; i < numPositions; ++i) std::fprintf(f, "v %.6f %.6f %.6f\n", positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
    for (int i = 0; i < numUvs; ++i) std::fprintf(f, "vt %.6f %.6f\n", uvs[2 * i], uvs[2 * i + 1]);
    for (int i = 0; i < numNormals; ++i) std::fprintf(f, "vn %.6f %.6f %.6f\n", normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
    int last = -1;
    for (int i = 0; i < numFaces; ++i) {
        const int m = faceMaterial ? faceMaterial[i] : 0;
        if (m != last && numMaterials > 0) {
            if (m < 0 || m >= numMaterials) { std::fclose(f); return -2; }
            std::fprintf(f, "usemtl %s\n", materialNames[m]);
            last = m;
        }
        const int* v = faces + 9 * (size_t)i;
        std::fprintf(f, "f %d/%d/%d %d/%d/%d %d/%d/%d\n", v[0] + 1, v[1] + 1, v[2] + 1, v[3] + 1, v[4] + 1, v[5] + 1, v[6] + 1, v[7] + 1, v[8] + 1);
    }
    return std::fclose(f) == 0 ? 0 : -3;
}
