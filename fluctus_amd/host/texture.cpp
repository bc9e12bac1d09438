// texture.cpp -- see texture.hpp.
#include "texture.hpp"
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <stdexcept>
#include <vector>

namespace fluctus {

bool fileExists(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fclose(f);
    return true;
}

static uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

Texture loadPNG(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<unsigned char> file;
    unsigned char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("not a PNG: " + path);
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte, trns;
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const unsigned char *type = &file[pos + 4], *data = &file[pos + 8];
        if (pos + 12 + len > file.size()) throw std::runtime_error("truncated PNG: " + path);
        if (!memcmp(type, "IHDR", 4)) { w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (!w || !h || interlace || (depth != 8 && depth != 16) || (ctype == 3 && depth != 8))
        throw std::runtime_error("unsupported PNG variant (interlaced / bit depth): " + path);
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels) throw std::runtime_error("unsupported PNG colour type: " + path);
    const size_t bpp = (size_t)channels * depth / 8, stride = bpp * w;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size())
        throw std::runtime_error("PNG inflate failed: " + path);
    std::vector<unsigned char> img(stride * h), zero(stride, 0);
    for (uint32_t y = 0; y < h; y++) {                     // undo the per-scanline filters (PNG spec 9)
        const unsigned char *in = &raw[(stride + 1) * y + 1];
        unsigned char *out = &img[stride * y];
        const unsigned char *up = y ? &img[stride * (y - 1)] : zero.data();
        int ft = raw[(stride + 1) * y];
        for (size_t i = 0; i < stride; i++) {
            int a = i >= bpp ? out[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0, x = in[i];
            switch (ft) {
            case 0: break; case 1: x += a; break; case 2: x += b; break; case 3: x += (a + b) / 2; break; case 4: x += paeth(a, b, c); break;
            default: throw std::runtime_error("bad PNG filter: " + path);
            }
            out[i] = (unsigned char)x;
        }
    }
    Texture t; t.name = path; t.width = w; t.height = h; t.rgba.resize((size_t)w * h * 4);
    const size_t step = depth / 8;                          // 16-bit samples: keep the most significant byte
    for (uint32_t y = 0; y < h; y++) {
        const unsigned char *src = &img[stride * y];
        unsigned char *dst = &t.rgba[(size_t)(h - 1 - y) * w * 4];          // lower-left origin: row 0 = bottom scanline
        for (uint32_t x = 0; x < w; x++) {
            const unsigned char *s = src + x * bpp; unsigned char *d = dst + x * 4;
            switch (ctype) {
            case 0: d[0] = d[1] = d[2] = s[0]; d[3] = 255; break;
            case 2: d[0] = s[0]; d[1] = s[step]; d[2] = s[2 * step]; d[3] = 255; break;
            case 3: { size_t k = s[0]; if (3 * k + 2 >= plte.size()) throw std::runtime_error("bad PNG palette index"); d[0] = plte[3 * k]; d[1] = plte[3 * k + 1]; d[2] = plte[3 * k + 2]; d[3] = k < trns.size() ? trns[k] : 255; break; }
            case 4: d[0] = d[1] = d[2] = s[0]; d[3] = s[step]; break;
            default: d[0] = s[0]; d[1] = s[step]; d[2] = s[2 * step]; d[3] = s[3 * step]; break;
            }
        }
    }
    return t;
}

} // namespace fluctus
