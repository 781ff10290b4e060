// JpegDecode.cpp -- see JpegDecode.hpp. Replaces the reference's stbi_load call (ResourceManager.cpp:193) for JPEG files.
//
// Structure (own design): the whole file is in memory; segments are parsed by Decoder::run(); every scan decodes
// quantised coefficients into per-component int16 block arrays (baseline scans dequantise as they go, progressive
// scans after the last one -- which matters only if a file redefines a DQT between scans, and is what stb_image does);
// then all blocks are inverse-transformed into per-component planes, and the planes are upsampled and colour-converted
// row by row. The arithmetic after entropy decoding is pinned to stb_image v2.27's so the bytes match the reference's:
//   * IDCT: the 12-bit fixed-point LL&M "islow" factorisation with stb's constants, +512 >> 10 after the column pass,
//     +65536 + (128 << 17) >> 17 after the row pass (stb_image.h:2400-2487);
//   * upsampling: nearest for 1x, (3a + b + 2) >> 2 vertically, the 3:1 / 9:3:3:1 filters with stb's edge rules for
//     2x horizontally / 2x2 (stb_image.h:3411-3470), replication for other ratios; which two source rows feed an output
//     row follows stb's half-step phase (stb_image.h:3874-3887);
//   * YCbCr -> RGB in 20-bit fixed point with the green Cb term masked to its high 16 bits (stb_image.h:3606-3630);
//   * colour space: three components are RGB (no conversion) when their ids are 'R','G','B' or when an Adobe APP14
//     segment says transform 0 and there is no JFIF header; four components are CMYK / YCCK per APP14
//     (stb_image.h:3827,3896-3922).
#include "JpegDecode.hpp"
#include <cstdint>
#include <cstring>

namespace {

const unsigned char kZigzag[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct HuffTable {
    bool defined = false;
    unsigned char values[256];
    unsigned char lookLen[512];      // 9-bit prefix -> code length (0 = longer than 9 bits)
    unsigned char lookSym[512];      // 9-bit prefix -> symbol
    int maxCode[18];                 // per length: largest code + 1, left-aligned to 16 bits; [17] = sentinel
    int valOffset[17];               // per length: index of the first symbol minus the first code
    bool build(const int counts[16])
    {
        int code = 0, k = 0;
        std::memset(lookLen, 0, sizeof lookLen);
        for (int len = 1; len <= 16; ++len) {
            valOffset[len] = k - code;
            const int n = counts[len - 1];
            if (n) {
                if (code + n > (1 << len)) return false;           // more codes than the length allows
                if (len <= 9)
                    for (int i = 0; i < n; ++i) {
                        const int first = (code + i) << (9 - len);
                        for (int j = 0; j < (1 << (9 - len)); ++j) { lookLen[first + j] = (unsigned char)len; lookSym[first + j] = (unsigned char)(k + i); }
                    }
                code += n; k += n;
            }
            maxCode[len] = code << (16 - len);
            code <<= 1;
        }
        maxCode[17] = 0x7fffffff;
        defined = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dcPred = 0;
    int pixW = 0, pixH = 0;          // effective size in samples
    int blocksW = 0, blocksH = 0;    // allocated blocks (whole MCUs)
    std::vector<int16_t> coef;       // blocksW * blocksH * 64, natural (row-major) order inside a block
    std::vector<unsigned char> plane;// blocksW*8 x blocksH*8
};

struct Decoder {
    const unsigned char* p; const unsigned char* end;
    const char* err = nullptr;
    // frame
    int width = 0, height = 0, ncomp = 0; bool progressive = false, haveFrame = false;
    int hMax = 1, vMax = 1, mcuX = 0, mcuY = 0;
    Component comp[4];
    uint16_t quant[4][64];
    HuffTable dc[4], ac[4];
    int restartInterval = 0;
    bool jfif = false; int adobeTransform = -1; int rgbIds = 0;
    // scan
    int scanN = 0, order[4] = { 0, 0, 0, 0 }, specStart = 0, specEnd = 63, succHigh = 0, succLow = 0;
    // entropy-coded segment reader
    uint32_t bitBuf = 0; int bitCnt = 0; bool hitMarker = false; int pendingMarker = -1; int eobRun = 0;

    bool fail(const char* why) { if (!err) err = why; return false; }
    int get8() { return p < end ? *p++ : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }
    bool atEnd() const { return p >= end; }
    void skip(int n) { if (n < 0 || (size_t)n > (size_t)(end - p)) p = end; else p += n; }

    // ---- bit reader over the entropy-coded segment: byte stuffing removed, zeros once a marker is reached ----
    void fill()
    {
        while (bitCnt <= 24) {
            int b = 0;
            if (!hitMarker) {
                b = get8();
                if (b == 0xff) {
                    int c = get8();
                    while (c == 0xff) c = get8();
                    if (c != 0) { pendingMarker = c; hitMarker = true; b = 0; }
                }
            }
            bitBuf |= (uint32_t)b << (24 - bitCnt);
            bitCnt += 8;
        }
    }
    int bits(int n)                      // n in 0..16
    {
        if (n == 0) return 0;
        if (bitCnt < n) fill();
        const int v = (int)(bitBuf >> (32 - n));
        bitBuf <<= n; bitCnt -= n;
        return v;
    }
    int bit() { return bits(1); }
    int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }   // T.81 F.2.2.1
    int receiveExtend(int n) { return n ? extend(bits(n), n) : 0; }
    int symbol(const HuffTable& t)
    {
        if (bitCnt < 16) fill();
        const int look = (int)(bitBuf >> 23);
        int len = t.lookLen[look];
        if (len) { bitBuf <<= len; bitCnt -= len; return t.values[t.lookSym[look]]; }
        const int top = (int)(bitBuf >> 16);
        for (len = 10; len <= 16; ++len) if (top < t.maxCode[len]) break;
        if (len > 16) return -1;
        const int idx = (top >> (16 - len)) + t.valOffset[len];
        bitBuf <<= len; bitCnt -= len;
        return (idx >= 0 && idx < 256) ? t.values[idx] : -1;
    }
    void resetEntropy()
    {
        bitBuf = 0; bitCnt = 0; hitMarker = false; pendingMarker = -1; eobRun = 0;
        for (Component& c : comp) c.dcPred = 0;
    }

    // ---- block decoders ----
    bool blockBaseline(Component& c, int16_t* blk)
    {
        const uint16_t* q = quant[c.tq];
        const int t = symbol(dc[c.td]);
        if (t < 0 || t > 15) return fail("bad huffman code");
        std::memset(blk, 0, 64 * sizeof(int16_t));
        c.dcPred = (int)((uint32_t)c.dcPred + (uint32_t)receiveExtend(t));      // modulo 2^32, like the rest (damaged files)
        blk[0] = (int16_t)((uint32_t)c.dcPred * (uint32_t)q[0]);
        for (int k = 1; k < 64;) {
            const int rs = symbol(ac[c.ta]);
            if (rs < 0) return fail("bad huffman code");
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) { if (rs != 0xf0) break; k += 16; continue; }
            k += r;
            if (k > 63) { (void)bits(s); break; }           // corrupt run: past the block, nothing to store
            const int z = kZigzag[k++];
            blk[z] = (int16_t)(receiveExtend(s) * q[z]);
        }
        return true;
    }
    bool blockProgressiveDC(Component& c, int16_t* blk)
    {
        if (specEnd != 0) return fail("can't merge dc and ac");
        if (succHigh == 0) {
            std::memset(blk, 0, 64 * sizeof(int16_t));
            const int t = symbol(dc[c.td]);
            if (t < 0 || t > 15) return fail("bad huffman code");
            c.dcPred = (int)((uint32_t)c.dcPred + (uint32_t)receiveExtend(t));
            blk[0] = (int16_t)((uint32_t)c.dcPred << succLow);
        } else if (bit()) blk[0] = (int16_t)(blk[0] + (int16_t)(1 << succLow));
        return true;
    }
    void refine(int16_t& coef, int16_t one)
    {
        if (bit() && (coef & one) == 0) coef = (int16_t)(coef > 0 ? coef + one : coef - one);
    }
    bool blockProgressiveAC(Component& c, int16_t* blk)
    {
        if (specStart == 0) return fail("can't merge dc and ac");
        const HuffTable& t = ac[c.ta];
        if (succHigh == 0) {                              // first pass over this band
            if (eobRun) { --eobRun; return true; }
            for (int k = specStart; k <= specEnd;) {
                const int rs = symbol(t);
                if (rs < 0) return fail("bad huffman code");
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) { eobRun = (1 << r) - 1; if (r) eobRun += bits(r); break; }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) { (void)bits(s); break; }
                    blk[kZigzag[k++]] = (int16_t)(receiveExtend(s) * (1 << succLow));
                }
            }
            return true;
        }
        const int16_t one = (int16_t)(1 << succLow);       // refinement pass (T.81 G.1.2.3)
        if (eobRun) {
            --eobRun;
            for (int k = specStart; k <= specEnd; ++k) { int16_t& v = blk[kZigzag[k]]; if (v != 0) refine(v, one); }
            return true;
        }
        int k = specStart;
        do {
            const int rs = symbol(t);
            if (rs < 0) return fail("bad huffman code");
            int r = rs >> 4, s = rs & 15;
            int16_t fresh = 0;
            if (s == 0) {
                if (r < 15) { eobRun = (1 << r) - 1; if (r) eobRun += bits(r); r = 64; }   // rest of the band: refinements only
            } else {
                if (s != 1) return fail("bad huffman code");
                fresh = bit() ? one : (int16_t)-one;
            }
            while (k <= specEnd) {
                int16_t& v = blk[kZigzag[k++]];
                if (v != 0) refine(v, one);
                else { if (r == 0) { v = fresh; break; } --r; }
            }
        } while (k <= specEnd);
        return true;
    }

    // ---- one scan ----
    bool decodeOne(Component& c, int bx, int by)
    {
        int16_t* blk = c.coef.data() + ((size_t)by * c.blocksW + bx) * 64;
        if (!progressive) return blockBaseline(c, blk);
        return specStart == 0 ? blockProgressiveDC(c, blk) : blockProgressiveAC(c, blk);
    }
    // returns false when the scan should stop (restart marker missing); `todo` counts MCUs to the next restart
    bool afterMcu(int& todo)
    {
        if (--todo > 0) return true;
        if (bitCnt < 24) fill();
        if (!(pendingMarker >= 0xd0 && pendingMarker <= 0xd7)) return false;
        resetEntropy();
        todo = restartInterval ? restartInterval : 0x7fffffff;
        return true;
    }
    bool scan()
    {
        resetEntropy();
        int todo = restartInterval ? restartInterval : 0x7fffffff;
        for (int i = 0; i < scanN; ++i) {
            const Component& c = comp[order[i]];
            if (!progressive || specStart == 0) { if (!dc[c.td].defined && !(progressive && succHigh)) return fail("bad DC huff"); }
            if (!progressive || specStart != 0) { if (!ac[c.ta].defined) return fail("bad AC huff"); }
        }
        if (scanN == 1) {                                  // non-interleaved: the component's own blocks, row by row
            Component& c = comp[order[0]];
            const int w = (c.pixW + 7) >> 3, h = (c.pixH + 7) >> 3;
            for (int by = 0; by < h; ++by)
                for (int bx = 0; bx < w; ++bx) {
                    if (!decodeOne(c, bx, by)) return false;
                    if (!afterMcu(todo)) return true;
                }
            return true;
        }
        if (progressive && specStart != 0) return fail("can't merge dc and ac");   // interleaved scans carry DC only
        for (int my = 0; my < mcuY; ++my)
            for (int mx = 0; mx < mcuX; ++mx) {
                for (int i = 0; i < scanN; ++i) {
                    Component& c = comp[order[i]];
                    for (int y = 0; y < c.v; ++y)
                        for (int x = 0; x < c.h; ++x)
                            if (!decodeOne(c, mx * c.h + x, my * c.v + y)) return false;
                }
                if (!afterMcu(todo)) return true;
            }
        return true;
    }

    // ---- segments ----
    bool readDQT()
    {
        int len = get16() - 2;
        while (len > 0) {
            const int q = get8(), prec = q >> 4, t = q & 15;
            if (prec > 1) return fail("bad DQT type");
            if (t > 3) return fail("bad DQT table");
            for (int i = 0; i < 64; ++i) quant[t][kZigzag[i]] = (uint16_t)(prec ? get16() : get8());
            len -= prec ? 129 : 65;
        }
        return len == 0 ? true : fail("bad DQT len");
    }
    bool readDHT()
    {
        int len = get16() - 2;
        while (len > 0) {
            const int q = get8(), cls = q >> 4, id = q & 15;
            if (cls > 1 || id > 3) return fail("bad DHT header");
            int counts[16], n = 0;
            for (int i = 0; i < 16; ++i) { counts[i] = get8(); n += counts[i]; }
            if (n > 256) return fail("bad DHT header");
            HuffTable& t = cls ? ac[id] : dc[id];
            if (!t.build(counts)) return fail("bad code lengths");
            for (int i = 0; i < n; ++i) t.values[i] = (unsigned char)get8();
            len -= 17 + n;
        }
        return len == 0 ? true : fail("bad DHT len");
    }
    bool readApp(int marker)
    {
        int len = get16();
        if (len < 2) return fail(marker == 0xfe ? "bad COM len" : "bad APP len");
        len -= 2;
        if (marker == 0xe0 && len >= 5) {
            static const char tag[5] = { 'J', 'F', 'I', 'F', 0 };
            bool ok = true;
            for (int i = 0; i < 5; ++i) if (get8() != (unsigned char)tag[i]) ok = false;
            len -= 5;
            if (ok) jfif = true;
        } else if (marker == 0xee && len >= 12) {
            static const char tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
            bool ok = true;
            for (int i = 0; i < 6; ++i) if (get8() != (unsigned char)tag[i]) ok = false;
            len -= 6;
            if (ok) { get8(); get16(); get16(); adobeTransform = get8(); len -= 6; }
        }
        skip(len);
        return true;
    }
    bool readSOF(int marker)
    {
        if (haveFrame) return fail("multiple SOF");
        progressive = marker == 0xc2;
        const int len = get16();
        if (len < 11) return fail("bad SOF len");
        if (get8() != 8) return fail("only 8-bit");
        height = get16(); if (height == 0) return fail("no header height");
        width = get16(); if (width == 0) return fail("0 width");
        ncomp = get8();
        if (ncomp != 1 && ncomp != 3 && ncomp != 4) return fail("bad component count");
        if (len != 8 + 3 * ncomp) return fail("bad SOF len");
        rgbIds = 0;
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            static const unsigned char rgb[3] = { 'R', 'G', 'B' };
            c.id = get8();
            if (ncomp == 3 && c.id == rgb[i]) ++rgbIds;
            const int q = get8();
            c.h = q >> 4; c.v = q & 15;
            if (c.h < 1 || c.h > 4) return fail("bad H");
            if (c.v < 1 || c.v > 4) return fail("bad V");
            c.tq = get8(); if (c.tq > 3) return fail("bad TQ");
        }
        if ((uint64_t)width * (uint64_t)height * (uint64_t)ncomp > 0x7fffffffull) return fail("too large");
        hMax = vMax = 1;
        for (int i = 0; i < ncomp; ++i) { if (comp[i].h > hMax) hMax = comp[i].h; if (comp[i].v > vMax) vMax = comp[i].v; }
        for (int i = 0; i < ncomp; ++i) { if (hMax % comp[i].h) return fail("bad H"); if (vMax % comp[i].v) return fail("bad V"); }
        mcuX = (width + hMax * 8 - 1) / (hMax * 8);
        mcuY = (height + vMax * 8 - 1) / (vMax * 8);
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            c.pixW = (width * c.h + hMax - 1) / hMax;
            c.pixH = (height * c.v + vMax - 1) / vMax;
            c.blocksW = mcuX * c.h; c.blocksH = mcuY * c.v;
            const uint64_t blocks = (uint64_t)c.blocksW * (uint64_t)c.blocksH;
            if (blocks * 64 > 0x7fffffffull) return fail("too large");
            c.coef.assign((size_t)blocks * 64, 0);
        }
        haveFrame = true;
        return true;
    }
    bool readSOS()
    {
        if (!haveFrame) return fail("SOS before SOF");
        const int len = get16();
        scanN = get8();
        if (scanN < 1 || scanN > 4 || scanN > ncomp) return fail("bad SOS component count");
        if (len != 6 + 2 * scanN) return fail("bad SOS len");
        for (int i = 0; i < scanN; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            while (which < ncomp && comp[which].id != id) ++which;
            if (which == ncomp) return fail("bad SOS component");
            comp[which].td = q >> 4; if (comp[which].td > 3) return fail("bad DC huff");
            comp[which].ta = q & 15; if (comp[which].ta > 3) return fail("bad AC huff");
            order[i] = which;
        }
        specStart = get8(); specEnd = get8();
        const int a = get8();
        succHigh = a >> 4; succLow = a & 15;
        if (progressive) {
            if (specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13) return fail("bad SOS");
        } else {
            if (specStart != 0 || succHigh != 0 || succLow != 0) return fail("bad SOS");
            specEnd = 63;
        }
        return true;
    }
    // next marker code: a pending one from the entropy stream, else 0xff fill bytes then the code; -1 when none
    int nextMarker()
    {
        if (pendingMarker >= 0) { const int m = pendingMarker; pendingMarker = -1; return m; }
        int x = get8();
        if (x != 0xff) return -1;
        while (x == 0xff) x = get8();
        return x;
    }

    bool run()
    {
        std::memset(quant, 0, sizeof quant);
        if (nextMarker() != 0xd8) return fail("no SOI");
        int m = nextMarker();
        for (;;) {
            if (m == 0xd9) break;                                   // EOI
            if (m == 0xda) {                                        // SOS + entropy-coded data
                if (!readSOS() || !scan()) return false;
                if (pendingMarker < 0) {                            // junk after the scan: look for the next marker
                    while (!atEnd()) { if (get8() == 0xff) { pendingMarker = get8(); break; } }
                }
            } else if (m == 0xdc) {                                 // DNL
                const int ld = get16(), nl = get16();
                if (ld != 4) return fail("bad DNL len");
                if (nl != height) return fail("bad DNL height");
            } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) { if (!readSOF(m)) return false; }
            else if (m == 0xdb) { if (!readDQT()) return false; }
            else if (m == 0xc4) { if (!readDHT()) return false; }
            else if (m == 0xdd) { if (get16() != 4) return fail("bad DRI len"); restartInterval = get16(); }
            else if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) { if (!readApp(m)) return false; }
            else if (m < 0) {
                // before the frame header stray bytes between segments are skipped; afterwards they are an error
                if (haveFrame) return fail("expected marker");
                if (atEnd()) return fail("no SOF");
            } else return fail("unknown marker");
            m = nextMarker();
            if (m < 0 && atEnd()) return fail(haveFrame ? "expected marker" : "no SOF");
        }
        if (!haveFrame) return fail("no SOF");
        if (!progressive) return true;
        for (int i = 0; i < ncomp; ++i) {                           // dequantise with the tables in force at the end
            const uint16_t* q = quant[comp[i].tq];
            std::vector<int16_t>& cf = comp[i].coef;
            for (size_t b = 0; b < cf.size(); b += 64)
                for (int k = 0; k < 64; ++k) cf[b + k] = (int16_t)(cf[b + k] * q[k]);
        }
        return true;
    }
};

// ---- inverse DCT (pinned arithmetic) ----
// All sums and products are taken modulo 2^32 (unsigned), which is what the reference's `int` arithmetic does on every
// real file and what two's-complement hardware does when a damaged file overflows it -- without undefined behaviour here.
typedef uint32_t u32;
inline u32 mulc(u32 a, int c) { return a * (u32)c; }
inline int sar(u32 x, int n) { return (int)x >> n; }             // arithmetic shift of the two's-complement value
struct Idct1D { u32 e0, e1, e2, e3, o0, o1, o2, o3; };
inline Idct1D idct1d(u32 s0, u32 s1, u32 s2, u32 s3, u32 s4, u32 s5, u32 s6, u32 s7)
{
    Idct1D r;
    const u32 z = mulc(s2 + s6, 2217);
    const u32 evenB = z + mulc(s6, -7567), evenA = z + mulc(s2, 3135);
    const u32 sum = mulc(s0 + s4, 4096), dif = mulc(s0 - s4, 4096);
    r.e0 = sum + evenA; r.e3 = sum - evenA; r.e1 = dif + evenB; r.e2 = dif - evenB;
    const u32 a = s7 + s3, b = s5 + s1, c = s7 + s1, d = s5 + s3;
    const u32 w = mulc(a + b, 4816);
    const u32 c5 = w + mulc(c, -3685), d5 = w + mulc(d, -10497), a3 = mulc(a, -8034), b3 = mulc(b, -1597);
    r.o3 = mulc(s1, 6149) + (c5 + b3);
    r.o2 = mulc(s3, 12586) + (d5 + a3);
    r.o1 = mulc(s5, 8410) + (d5 + b3);
    r.o0 = mulc(s7, 1223) + (c5 + a3);
    return r;
}
inline unsigned char clamp255(int x) { return (unsigned char)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

void idctBlock(const int16_t* d, unsigned char* out, int stride)
{
    int tmp[64];
    for (int c = 0; c < 8; ++c) {
        const Idct1D r = idct1d((u32)d[c], (u32)d[8 + c], (u32)d[16 + c], (u32)d[24 + c], (u32)d[32 + c], (u32)d[40 + c], (u32)d[48 + c], (u32)d[56 + c]);
        tmp[c] = sar(r.e0 + 512 + r.o3, 10);      tmp[56 + c] = sar(r.e0 + 512 - r.o3, 10);
        tmp[8 + c] = sar(r.e1 + 512 + r.o2, 10);  tmp[48 + c] = sar(r.e1 + 512 - r.o2, 10);
        tmp[16 + c] = sar(r.e2 + 512 + r.o1, 10); tmp[40 + c] = sar(r.e2 + 512 - r.o1, 10);
        tmp[24 + c] = sar(r.e3 + 512 + r.o0, 10); tmp[32 + c] = sar(r.e3 + 512 - r.o0, 10);
    }
    const u32 bias = 65536u + (128u << 17);
    for (int y = 0; y < 8; ++y) {
        const int* v = tmp + y * 8;
        unsigned char* o = out + (size_t)y * stride;
        const Idct1D r = idct1d((u32)v[0], (u32)v[1], (u32)v[2], (u32)v[3], (u32)v[4], (u32)v[5], (u32)v[6], (u32)v[7]);
        o[0] = clamp255(sar(r.e0 + bias + r.o3, 17)); o[7] = clamp255(sar(r.e0 + bias - r.o3, 17));
        o[1] = clamp255(sar(r.e1 + bias + r.o2, 17)); o[6] = clamp255(sar(r.e1 + bias - r.o2, 17));
        o[2] = clamp255(sar(r.e2 + bias + r.o1, 17)); o[5] = clamp255(sar(r.e2 + bias - r.o1, 17));
        o[3] = clamp255(sar(r.e3 + bias + r.o0, 17)); o[4] = clamp255(sar(r.e3 + bias - r.o0, 17));
    }
}

// ---- upsampling of one row (pinned filters); `w` source samples -> w*hs output samples ----
void upsampleRow(unsigned char* out, const unsigned char* nearRow, const unsigned char* farRow, int w, int hs, int vs)
{
    if (hs == 1 && vs == 1) { std::memcpy(out, nearRow, (size_t)w); return; }
    if (hs == 1 && vs == 2) { for (int i = 0; i < w; ++i) out[i] = (unsigned char)((3 * nearRow[i] + farRow[i] + 2) >> 2); return; }
    if (hs == 2 && vs == 1) {
        const unsigned char* in = nearRow;
        if (w == 1) { out[0] = out[1] = in[0]; return; }
        out[0] = in[0];
        out[1] = (unsigned char)((in[0] * 3 + in[1] + 2) >> 2);
        for (int i = 1; i < w - 1; ++i) {
            const int n = 3 * in[i] + 2;
            out[2 * i] = (unsigned char)((n + in[i - 1]) >> 2);
            out[2 * i + 1] = (unsigned char)((n + in[i + 1]) >> 2);
        }
        out[2 * (w - 1)] = (unsigned char)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[2 * (w - 1) + 1] = in[w - 1];
        return;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (unsigned char)((3 * nearRow[0] + farRow[0] + 2) >> 2); return; }
        int cur = 3 * nearRow[0] + farRow[0];
        out[0] = (unsigned char)((cur + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            const int prev = cur;
            cur = 3 * nearRow[i] + farRow[i];
            out[2 * i - 1] = (unsigned char)((3 * prev + cur + 8) >> 4);
            out[2 * i] = (unsigned char)((3 * cur + prev + 8) >> 4);
        }
        out[2 * w - 1] = (unsigned char)((cur + 2) >> 2);
        return;
    }
    for (int i = 0; i < w; ++i) for (int j = 0; j < hs; ++j) out[i * hs + j] = nearRow[i];   // other ratios: replication
}

inline unsigned char blinn8(unsigned x, unsigned y) { const unsigned t = x * y + 128; return (unsigned char)((t + (t >> 8)) >> 8); }

inline void ycc2rgb(int y, int cb, int cr, unsigned char* out)
{
    const int yf = (y << 20) + (1 << 19);
    cr -= 128; cb -= 128;
    int r = yf + cr * 1470208;
    int g = yf + cr * -748800 + (int)((unsigned)(cb * -360960) & 0xffff0000u);
    int b = yf + cb * 1858048;
    r >>= 20; g >>= 20; b >>= 20;
    out[0] = clamp255(r); out[1] = clamp255(g); out[2] = clamp255(b);
}

} // namespace

bool JpegDecodeRGB8(const unsigned char* data, size_t size, JpegInfo* info, std::vector<unsigned char>& rgb, const char** error)
{
    static const char* kNoData = "no data";
    if (error) *error = nullptr;
    if (!data || size < 4) { if (error) *error = kNoData; return false; }
    Decoder* D = new Decoder;
    D->p = data; D->end = data + size;
    struct Free { Decoder* d; ~Free() { delete d; } } guard{ D };
    if (!D->run()) { if (error) *error = D->err ? D->err : "corrupt JPEG"; return false; }
    const int W = D->width, H = D->height, N = D->ncomp;
    if (info) { info->width = W; info->height = H; info->components = N >= 3 ? 3 : 1; info->progressive = D->progressive; }

    // planes
    for (int i = 0; i < N; ++i) {
        Component& c = D->comp[i];
        const int stride = c.blocksW * 8;
        c.plane.assign((size_t)stride * (size_t)c.blocksH * 8, 0);
        for (int by = 0; by < c.blocksH; ++by)
            for (int bx = 0; bx < c.blocksW; ++bx)
                idctBlock(c.coef.data() + ((size_t)by * c.blocksW + bx) * 64, c.plane.data() + (size_t)by * 8 * stride + (size_t)bx * 8, stride);
        std::vector<int16_t>().swap(c.coef);
    }

    const bool isRgb = N == 3 && (D->rgbIds == 3 || (D->adobeTransform == 0 && !D->jfif));
    rgb.assign((size_t)W * (size_t)H * 3, 0);
    std::vector<unsigned char> line[4];
    struct Phase { int hs, vs, step, lo, hi, srcRow, wLow; } ph[4];
    for (int i = 0; i < N; ++i) {
        const Component& c = D->comp[i];
        ph[i].hs = D->hMax / c.h; ph[i].vs = D->vMax / c.v;
        ph[i].step = ph[i].vs >> 1; ph[i].lo = ph[i].hi = 0; ph[i].srcRow = 0;
        ph[i].wLow = (W + ph[i].hs - 1) / ph[i].hs;
        line[i].assign((size_t)W + 8 * 4, 0);
    }
    for (int y = 0; y < H; ++y) {
        const unsigned char* row[4] = { nullptr, nullptr, nullptr, nullptr };
        for (int i = 0; i < N; ++i) {
            const Component& c = D->comp[i];
            Phase& f = ph[i];
            const size_t stride = (size_t)c.blocksW * 8;
            // the output row sits in the lower half of the source row `hi` or the upper half of `lo`: the nearer one
            // weighs 3, the other 1 (rows lo and hi are the same at the top and bottom edges)
            const bool lower = f.step >= (f.vs >> 1);
            const unsigned char* nearRow = c.plane.data() + stride * (size_t)(lower ? f.hi : f.lo);
            const unsigned char* farRow = c.plane.data() + stride * (size_t)(lower ? f.lo : f.hi);
            if (f.hs == 1 && f.vs == 1) row[i] = nearRow;
            else { upsampleRow(line[i].data(), nearRow, farRow, f.wLow, f.hs, f.vs); row[i] = line[i].data(); }
            if (++f.step >= f.vs) { f.step = 0; f.lo = f.hi; if (++f.srcRow < c.pixH) ++f.hi; }
        }
        unsigned char* out = rgb.data() + (size_t)y * (size_t)W * 3;
        if (N == 1) for (int x = 0; x < W; ++x) { out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = row[0][x]; }
        else if (N == 3 && isRgb) for (int x = 0; x < W; ++x) { out[3 * x] = row[0][x]; out[3 * x + 1] = row[1][x]; out[3 * x + 2] = row[2][x]; }
        else if (N == 3) for (int x = 0; x < W; ++x) ycc2rgb(row[0][x], row[1][x], row[2][x], out + 3 * x);
        else if (D->adobeTransform == 0) for (int x = 0; x < W; ++x) {            // CMYK
            const unsigned k = row[3][x];
            out[3 * x] = blinn8(row[0][x], k); out[3 * x + 1] = blinn8(row[1][x], k); out[3 * x + 2] = blinn8(row[2][x], k);
        } else if (D->adobeTransform == 2) for (int x = 0; x < W; ++x) {          // YCCK
            const unsigned k = row[3][x];
            unsigned char t[3]; ycc2rgb(row[0][x], row[1][x], row[2][x], t);
            out[3 * x] = blinn8(255u - t[0], k); out[3 * x + 1] = blinn8(255u - t[1], k); out[3 * x + 2] = blinn8(255u - t[2], k);
        } else for (int x = 0; x < W; ++x) ycc2rgb(row[0][x], row[1][x], row[2][x], out + 3 * x);   // YCbCr + an ignored fourth channel
    }
    return true;
}
