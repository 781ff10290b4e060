// MeshCache.cpp -- the reference's `.clm` mesh cache (AssetManager.cpp:291-361) for the mirrored AssetManager.
//
// File layout (little endian): u32 version (0) | i32 numTris | i32 numMaterials | ObjMaterial[numMaterials] (24 B each) |
// u32 mtlBytes | mtlText | then `Tri[numTris]` raw when numTris < 1000, else u64 compressedBytes | QuickLZ stream.
//
// The stream is QuickLZ 1.5.0, compression level 1, no streaming buffer (quicklz.h:27,33 upstream). The decoder below
// is written from the format, not taken from quicklz.c:
//   header  byte 0 = 01SSLLHC: C compressed, H long header (sizes are 4 bytes instead of 1), LL level, SS streaming;
//           then compressed size (whole stream incl. header) and decompressed size.
//   body    32-bit control words, consumed LSB first, refilled when only the sentinel bit is left. Flag 1 = match:
//           the next 2 bytes hold a 12-bit hash (bits 4..15) and a 4-bit length-2 (0 = a third byte is the length);
//           the match source is the last position whose 3-byte hash ((v >> 12) ^ v) & 4095 equals it, so the decoder
//           keeps the same hash table as the compressor: after a literal run every position up to 3 bytes before the
//           write cursor is hashed, after a match every position up to the match start (the inside of a match is
//           never hashed). Flag 0 = literals: 1..4 at a time (as many as there are consecutive 0 flags in the low
//           nibble) while more than 10 bytes remain, then byte by byte to the end.
// Writing uses the format's own escape for incompressible data -- a header with C = 0 followed by the bytes -- which
// every QuickLZ decoder (upstream's included) accepts; the cache is bigger than upstream's but interchangeable.
#include "AssetManager.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr unsigned CMeshVersion = 0;             // AssetManager.cpp:291
bool g_cacheEnabled = true;                      // upstream always writes and prefers the cache (AssetManager.cpp:287,372)

inline unsigned rd(const unsigned char* p, int n) { unsigned v = 0; for (int i = 0; i < n; ++i) v |= (unsigned)p[i] << (8 * i); return v; }
inline unsigned hash3(const unsigned char* p) { const unsigned v = rd(p, 3); return ((v >> 12) ^ v) & 4095u; }

} // namespace

// Returns the number of bytes written to dst (0 on a malformed stream or when dstCap is too small).
size_t MeshCache_QlzDecompress(const unsigned char* src, size_t srcLen, unsigned char* dst, size_t dstCap)
{
    if (srcLen < 3) return 0;
    const int n = (src[0] & 2) ? 4 : 1;
    const size_t header = 2 * (size_t)n + 1;
    if (srcLen < header) return 0;
    const size_t compSize = rd(src + 1, n), size = rd(src + 1 + n, n);
    if (compSize > srcLen || size > dstCap) return 0;
    if (!(src[0] & 1)) {                                       // stored
        if (header + size > srcLen) return 0;
        std::memcpy(dst, src + header, size);
        return size;
    }
    if (((src[0] >> 2) & 3) != 1 || size == 0) return 0;         // only level 1 is ever written by upstream
    std::vector<size_t> table(4096, (size_t)-1);               // hash -> position in dst
    const unsigned char* s = src + header;
    const unsigned char* const sEnd = src + compSize;
    size_t d = 0;                                              // write cursor
    size_t hashed = 0;                                         // positions [0, hashed) are in the table
    unsigned cword = 1;
    const size_t lastMatchStart = size >= 11 ? size - 11 : 0;  // last byte index - 6 - 4
    auto hash_upto = [&](size_t maxPos) {                      // hash positions hashed .. maxPos
        while (hashed <= maxPos && hashed + 3 <= d) { table[hash3(dst + hashed)] = hashed; ++hashed; }
    };
    for (;;) {
        if (cword == 1) {
            if (s + 4 > sEnd) return 0;
            cword = rd(s, 4); s += 4;
        }
        if (s + 4 > sEnd) return 0;
        if (cword & 1) {                                       // match
            cword >>= 1;
            const unsigned fetch = rd(s, 4);
            const size_t from = table[(fetch >> 4) & 0xfffu];
            size_t len;
            if (fetch & 0xf) { len = (fetch & 0xf) + 2; s += 2; }
            else { len = s[2]; s += 3; }
            if (from == (size_t)-1 || from + 3 > d || len > size - d || size - d - len < 4) return 0;
            for (size_t k = 0; k < len; ++k) dst[d + k] = dst[from + k];   // forward, byte-wise: overlap repeats
            const size_t start = d;
            d += len;
            hash_upto(start);
            hashed = d;                                        // the inside of a match is never hashed
        } else if (d < lastMatchStart) {                       // 1..4 literals
            const unsigned k = (unsigned)__builtin_ctz((cword & 0xfu) | 0x10u);   // literal flags (zeros) at the bottom of the nibble: 0..4
            if (d + 4 > size) return 0;
            std::memcpy(dst + d, s, 4);                        // only k of them count; the rest is overwritten
            cword >>= k; d += k; s += k;
            if (d >= 3) hash_upto(d - 3);
        } else {                                               // tail: byte by byte
            while (d < size) {
                if (cword == 1) { s += 4; cword = 1u << 31; }
                if (s >= sEnd) return 0;
                dst[d++] = *s++;
                cword >>= 1;
            }
            return size;
        }
    }
}

// Stored ("incompressible") QuickLZ block: 9-byte long header with C = 0, then the bytes. Returns bytes written.
size_t MeshCache_QlzStore(const unsigned char* src, size_t size, unsigned char* dst)
{
    dst[0] = (unsigned char)((1u << 6) | (1u << 2) | 2u);       // 01 SS=00 LL=01 H=1 C=0
    const unsigned comp = (unsigned)(size + 9), dec = (unsigned)size;
    for (int i = 0; i < 4; ++i) { dst[1 + i] = (unsigned char)(comp >> (8 * i)); dst[5 + i] = (unsigned char)(dec >> (8 * i)); }
    std::memcpy(dst + 9, src, size);
    return size + 9;
}

void AssetManager_SetMeshCache(bool enabled) { g_cacheEnabled = enabled; }
bool AssetManager_MeshCacheEnabled() { return g_cacheEnabled; }

// AssetManager.cpp:294-322
bool AssetManager_SaveMeshToDisk(const char* path, const ObjMesh* mesh)
{
    // written under a private name and renamed into place: several processes may import the same mesh at once
    const std::string tmp = std::string(path) + ".tmp" + std::to_string((long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const unsigned version = CMeshVersion, msz = mesh->mtlText ? mesh->mtlSize : 0u;
    bool ok = std::fwrite(&version, 4, 1, f) == 1 && std::fwrite(&mesh->numTris, 4, 1, f) == 1 && std::fwrite(&mesh->numMaterials, 4, 1, f) == 1;
    ok = ok && (mesh->numMaterials == 0 || std::fwrite(mesh->materials, sizeof(ObjMaterial), (size_t)mesh->numMaterials, f) == (size_t)mesh->numMaterials);
    ok = ok && std::fwrite(&msz, 4, 1, f) == 1 && (msz == 0 || std::fwrite(mesh->mtlText, 1, msz, f) == msz);
    const size_t bytes = (size_t)mesh->numTris * sizeof(Tri);
    if (mesh->numTris < 1000) ok = ok && (bytes == 0 || std::fwrite(mesh->tris, 1, bytes, f) == bytes);
    else {
        std::vector<unsigned char> buf(bytes + 9);
        const unsigned long long comp = MeshCache_QlzStore(reinterpret_cast<const unsigned char*>(mesh->tris), bytes, buf.data());
        ok = ok && std::fwrite(&comp, 8, 1, f) == 1 && std::fwrite(buf.data(), 1, (size_t)comp, f) == (size_t)comp;
    }
    ok = (std::fclose(f) == 0) && ok;
    ok = ok && std::rename(tmp.c_str(), path) == 0;
    if (!ok) std::remove(tmp.c_str());
    return ok;
}

// AssetManager.cpp:324-361. maxTris = room left in the triangle arena.
ObjMesh* AssetManager_LoadMeshFromDisk(const char* path, Tri* triArena, size_t maxTris)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return nullptr;
    std::fseek(f, 0, SEEK_END);
    const long fileLen = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> file(fileLen > 0 ? (size_t)fileLen : 0);
    const bool readOk = fileLen > 0 && std::fread(file.data(), 1, file.size(), f) == file.size();
    std::fclose(f);
    auto fail = [&](const char* why) { std::fprintf(stderr, "[AssetManager] %s: %s\n", path, why); return (ObjMesh*)nullptr; };
    if (!readOk) return fail("cannot read mesh cache");
    size_t at = 0;
    auto take = [&](void* out, size_t n) { if (n > file.size() - at) return false; std::memcpy(out, file.data() + at, n); at += n; return true; };
    unsigned version = 0, msz = 0; int numTris = 0, numMaterials = 0;
    if (!take(&version, 4) || !take(&numTris, 4) || !take(&numMaterials, 4)) return fail("truncated mesh cache");
    if (version != CMeshVersion) return fail("mesh version is not same!");           // AssetManager.cpp:341 (exit(0) upstream)
    if (numTris < 0 || (size_t)numTris > maxTris || numMaterials < 0 || numMaterials > 32) return fail("mesh cache does not fit");
    ObjMesh* mesh = new ObjMesh;
    mesh->name = nullptr; mesh->tris = triArena; mesh->numTris = numTris; mesh->numMaterials = numMaterials; mesh->mtlText = nullptr; mesh->mtlSize = 0;
    std::memset(mesh->materials, 0, sizeof mesh->materials);
    bool ok = take(mesh->materials, sizeof(ObjMaterial) * (size_t)numMaterials) && take(&msz, 4) && at + msz <= file.size();
    if (ok && msz) {
        mesh->mtlText = (char*)std::malloc(msz);
        mesh->mtlSize = msz;
        ok = take(mesh->mtlText, msz);
        for (int m = 0; ok && m < numMaterials; ++m) {            // offsets into mtlText must stay inside it
            const ObjMaterial& mt = mesh->materials[m];
            ok = (unsigned)mt.name < msz && (unsigned)mt.diffusePath < msz && (unsigned)mt.specularPath < msz;
        }
        if (ok) mesh->mtlText[msz - 1] = '\0';
    } else if (ok && numMaterials) ok = false;
    const size_t bytes = (size_t)numTris * sizeof(Tri);
    if (ok && numTris < 1000) ok = take(triArena, bytes);
    else if (ok) {
        unsigned long long comp = 0;
        // `comp` is an untrusted 64-bit field: compare against what is left of the file (at <= file.size() after take), never at + comp
        ok = take(&comp, 8) && comp <= (unsigned long long)(file.size() - at)
             && MeshCache_QlzDecompress(file.data() + at, (size_t)comp, reinterpret_cast<unsigned char*>(triArena), bytes) == bytes;
    }
    if (!ok) { AssetManager_DestroyMesh(mesh); return fail("corrupt mesh cache"); }
    return mesh;
}

// true when `cache` exists and is not older than `source`
bool MeshCache_IsFresh(const char* cache, const char* source)
{
    struct stat c, s;
    if (stat(cache, &c) != 0) return false;
    if (stat(source, &s) != 0) return true;                       // only the cache is there (upstream's shipped assets)
    return c.st_mtime >= s.st_mtime;
}
