// texture.hpp -- image decode for material textures (host side).
//
// The reference loads textures through DevIL with a lower-left origin and converts to RGBA8
// (reference: src/texture.cpp:16-41, src/main.cpp:69-71).  DevIL is not available here; PNG (the format of
// assets/egyptcat/*.png) is decoded with zlib: non-interlaced, bit depth 8 or 16, colour types 0/2/3/4/6.
// JPEG (Country Kitchen) is SURVEY 8(f) N2 "next".
#pragma once
#include <string>
#include "scene.hpp"

namespace fluctus {

// throws std::runtime_error on unsupported / corrupt files
Texture loadPNG(const std::string &path);
bool fileExists(const std::string &path);

} // namespace fluctus
