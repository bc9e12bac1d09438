// texture.hpp -- image decode for material textures (host side).
//
// The reference loads textures through DevIL with a lower-left origin and converts to RGBA8
// (reference: src/texture.cpp:16-41, src/main.cpp:69-71).  DevIL is not available here; PNG (the format of
// assets/egyptcat/*.png) is decoded with zlib: non-interlaced, bit depth 8 or 16, colour types 0/2/3/4/6.
// JPEG (assets/country_kitchen/textures/*.jpg) is decoded by jpeg.cpp with the three integer stages of
// libjpeg's default path restated exactly -- the decoded bytes are libjpeg's bytes.
#pragma once
#include <string>
#include <vector>
#include <cstdint>
#include "scene.hpp"

namespace fluctus {

// throws std::runtime_error on unsupported / corrupt files
Texture loadPNG(const std::string &path);
Texture loadJPEG(const std::string &path);
Texture loadTexture(const std::string &path);      // by file signature: JPEG (FF D8) or PNG
// top-left origin, tightly packed RGB8 (what libjpeg's jpeg_read_scanlines delivers)
void decodeJPEG(const uint8_t *data, size_t size, uint32_t *width, uint32_t *height, std::vector<uint8_t> &rgb);
bool fileExists(const std::string &path);

} // namespace fluctus
