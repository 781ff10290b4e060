// JpegDecode.hpp -- JPEG texture decode for ResourceManager::ImportTexture.
//
// The reference decodes textures with `stbi_load(path, &w, &h, &channels, 3)` from its vendored stb_image v2.27
// (ResourceManager.cpp:6-11,193; STBI_ONLY_JPEG). This is an independent decoder that produces the same RGB8 bytes:
// baseline and progressive Huffman JPEG (SOF0/SOF1/SOF2), 8-bit, 1/3/4 components, restart intervals, and -- because
// byte parity needs them -- the same fixed-point arithmetic stb_image uses after entropy decoding (its integer IDCT,
// its "jfif-centered" 2x chroma upsampling filters, its 20-bit YCbCr->RGB, its Adobe APP14 / 'R','G','B' component-id
// colour-space rules). tests/test_jpeg.py compares it byte for byte with the reference's own stb_image.h compiled as
// it lies (oracle/_ref/libstb_image_ref.so) on every JPEG the reference ships, and with committed SHA-256 fixtures.
#pragma once
#include <cstddef>
#include <vector>

struct JpegInfo {
    int width = 0, height = 0;
    int components = 0;       // components in the file (1, 3 or 4); the output is always RGB8
    bool progressive = false;
};

// Decodes `size` bytes at `data` into tightly packed RGB8 (width*height*3 bytes, top row first).
// Returns false and sets *error (static string) on malformed input; never reads outside [data, data+size).
bool JpegDecodeRGB8(const unsigned char* data, size_t size, JpegInfo* info, std::vector<unsigned char>& rgb, const char** error);
