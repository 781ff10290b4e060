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
// Writing: MeshCache_QlzCompress produces the bytes upstream's qlz_compress produces (level 1, 64-bit x86 build:
// 4-byte fetches), so a cache written here is the file upstream would have written for the same triangles:
//   the compressor walks the input keeping, per 12-bit hash, the last position it stood on and the 4 bytes it saw there.
//   At each step: the slot's 3 bytes equal the current ones and the slot is set (position 0 counts as unset) and the
//   candidate is more than 2 bytes back -- or exactly 1 back inside a run of 7 equal bytes after >= 3 literals -- ->
//   a match (3 bytes if the 4th byte differs; else 4, 5, or as many as agree up to 255 and never into the last 4
//   bytes); otherwise a literal. The slot is overwritten either way; positions inside a match are not visited.
//   The last 10 bytes are literals (their first 7 still refresh the table). A control word is flushed after 31
//   tokens; at a flush past half of the input with less than 1/32 saved, the stream is abandoned and the data is
//   stored with C = 0 (MeshCache_QlzStore). Inputs under 216 bytes get the short header.
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

// QuickLZ 1.5.0 level-1 compressor (format and policy in the header comment). dst needs size + 400 bytes.
// Returns the stream length including its header; 0 for size 0.
size_t MeshCache_QlzCompress(const unsigned char* src, size_t size, unsigned char* dst)
{
    if (size == 0 || size > 0xffffffffull - 400) return 0;
    const size_t head = size < 216 ? 3 : 9;
    struct Slot { uint32_t pos, seen; };                         // last position with this hash (0 = none) and the 4 bytes there
    std::vector<Slot> table(4096, Slot{ 0u, 0u });
    auto hashOf = [](uint32_t v) { return ((v >> 12) ^ v) & 4095u; };
    auto rd4 = [&](size_t at) { return (uint32_t)src[at] | (uint32_t)src[at + 1] << 8 | (uint32_t)src[at + 2] << 16 | (uint32_t)src[at + 3] << 24; };

    unsigned char* const body = dst + head;
    const size_t last = size - 1;
    size_t in = 0, out = 4, ctlAt = 0;                           // `out`/`ctlAt` index body[]; the first control word sits at 0
    uint32_t ctl = 1u << 31;                                     // flags enter at bit 31; the sentinel reaching bit 0 = 31 tokens
    unsigned literalsInARow = 0;
    bool stored = false;
    auto putCtl = [&](uint32_t v) { for (int i = 0; i < 4; ++i) body[ctlAt + i] = (unsigned char)(v >> (8 * i)); };
    auto flushIfFull = [&]() {
        if ((ctl & 1u) == 0) return;
        putCtl((ctl >> 1) | (1u << 31));
        ctlAt = out; out += 4; ctl = 1u << 31;
    };

    // positions at which a match may still start: it must end before the last 4 bytes and a 6-byte look-ahead stays inside
    const bool hasMatchZone = size >= 11;
    const size_t lastMatchStart = hasMatchZone ? last - 10 : 0;
    while (hasMatchZone && in <= lastMatchStart) {
        if (ctl & 1u) {
            if (in > (size >> 1) && out > in - (in >> 5)) { stored = true; break; }   // less than 1/32 saved past the middle: give up
            flushIfFull();
        }
        const uint32_t now = rd4(in);
        const uint32_t h = hashOf(now);
        Slot& slot = table[h];
        const uint32_t diff = now ^ slot.seen;
        const size_t cand = slot.pos;
        slot.seen = now; slot.pos = (uint32_t)in;
        bool usable = (diff & 0xffffffu) == 0 && cand != 0;
        if (usable && !(in - cand > 2)) {
            // a candidate 1 back is a run: taken only after >= 3 literals, when the 7 bytes from in-3 are all equal
            usable = in == cand + 1 && literalsInARow >= 3 && in > 3;
            for (size_t k = 1; usable && k <= 6; ++k) usable = src[in - 3 + k] == src[in - 3];
        }
        if (!usable) {
            body[out++] = src[in++];
            ctl >>= 1;
            ++literalsInARow;
            continue;
        }
        ctl = (ctl >> 1) | (1u << 31);
        size_t len = 3;
        if (diff == 0) {                                         // the 4th byte agrees too
            len = 4;
            if (src[cand + 4] == src[in + 4]) {
                len = 5;
                if (src[cand + 5] == src[in + 5]) {
                    const size_t room = last - 4 - in + 1;       // never into the last 4 bytes
                    const size_t cap = room > 255 ? 255 : room;
                    len = 6;
                    while (src[cand + len] == src[in + len] && len < cap) ++len;
                }
            }
        }
        if (len < 18) { const uint32_t w = (uint32_t)(len - 2) | (h << 4); body[out++] = (unsigned char)w; body[out++] = (unsigned char)(w >> 8); }
        else { const uint32_t w = (uint32_t)(len << 16) | (h << 4); body[out++] = (unsigned char)w; body[out++] = (unsigned char)(w >> 8); body[out++] = (unsigned char)(w >> 16); }
        in += len;
        literalsInARow = 0;
    }
    if (!stored) {
        while (in <= last) {                                     // the tail: literals; positions with 4 bytes left still refresh the table
            flushIfFull();
            if (in + 3 <= last) { const uint32_t now = rd4(in); Slot& slot = table[hashOf(now)]; slot.pos = (uint32_t)in; slot.seen = now; }
            body[out++] = src[in++];
            ctl >>= 1;
        }
        while ((ctl & 1u) == 0) ctl >>= 1;
        putCtl((ctl >> 1) | (1u << 31));
        if (out < 9) out = 9;                                    // upstream's minimum body length (the bytes past the data are unspecified there)
    }
    size_t total;
    if (stored) { std::memcpy(body, src, size); total = head + size; }
    else total = head + out;
    dst[0] = (unsigned char)((1u << 6) | (1u << 2) | (head == 9 ? 2u : 0u) | (stored ? 0u : 1u));   // 01 SS=00 LL=01 H C
    if (head == 3) { dst[1] = (unsigned char)total; dst[2] = (unsigned char)size; }
    else for (int i = 0; i < 4; ++i) { dst[1 + i] = (unsigned char)((uint32_t)total >> (8 * i)); dst[5 + i] = (unsigned char)((uint32_t)size >> (8 * i)); }
    return total;
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
        std::vector<unsigned char> buf(bytes + 400);             // AssetManager.cpp:312-314
        const unsigned long long comp = MeshCache_QlzCompress(reinterpret_cast<const unsigned char*>(mesh->tris), bytes, buf.data());
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
        mesh->mtlText = (char*)std::malloc((size_t)msz + 1);     // upstream stores the file's bytes without a terminator (AssetManager.cpp:303)
        mesh->mtlSize = msz;
        ok = take(mesh->mtlText, msz);
        mesh->mtlText[msz] = '\0';
        for (int m = 0; ok && m < numMaterials; ++m) {            // offsets into mtlText must stay inside it
            const ObjMaterial& mt = mesh->materials[m];
            ok = (unsigned)mt.name < msz && (unsigned)mt.diffusePath < msz && (unsigned)mt.specularPath < msz;
        }
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
