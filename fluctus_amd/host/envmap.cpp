// envmap.cpp -- see envmap.hpp.
#include "envmap.hpp"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <stdexcept>
#include <utility>

namespace fluctus {

namespace {

// Radiance RGBE pixel -> float: mantissa * 2^(e-136) (reference: src/rgbe/rgbe.cpp:92-105)
inline void rgbe2float(const unsigned char *p, float *out)
{
    if (p[3]) {
        float f = (float)std::ldexp(1.0, (int)p[3] - (128 + 8));
        out[0] = p[0] * f; out[1] = p[1] * f; out[2] = p[2] * f;
    } else out[0] = out[1] = out[2] = 0.0f;
}

void readHdr(const std::string &filename, int &w, int &h, std::vector<float> &rgb)
{
    FILE *f = fopen(filename.c_str(), "rb");
    if (!f) throw std::runtime_error("Cannot open file '" + filename + "'");
    char buf[256]; w = h = 0;
    while (fgets(buf, sizeof(buf), f)) {                     // header lines until the resolution line
        if (sscanf(buf, "-Y %d +X %d", &h, &w) == 2) break;
    }
    if (w <= 0 || h <= 0) { fclose(f); throw std::runtime_error("bad .hdr header: " + filename); }
    rgb.resize((size_t)w * h * 3);
    std::vector<unsigned char> line((size_t)w * 4);
    for (int y = 0; y < h; y++) {
        unsigned char hd[4];
        if (fread(hd, 1, 4, f) != 4) { fclose(f); throw std::runtime_error("short .hdr"); }
        bool rle = w >= 8 && w <= 0x7fff && hd[0] == 2 && hd[1] == 2 && !(hd[2] & 0x80) && ((hd[2] << 8) | hd[3]) == w;
        if (!rle) {                                          // flat scanline
            memcpy(line.data(), hd, 4);
            if (fread(line.data() + 4, 1, (size_t)(w - 1) * 4, f) != (size_t)(w - 1) * 4) { fclose(f); throw std::runtime_error("short .hdr"); }
        } else {
            for (int c = 0; c < 4; c++) {                    // each channel run-length encoded separately
                int x = 0;
                while (x < w) {
                    unsigned char b[2];
                    if (fread(b, 1, 2, f) != 2) { fclose(f); throw std::runtime_error("short .hdr"); }
                    if (b[0] > 128) { int cnt = b[0] - 128; while (cnt-- > 0 && x < w) line[(size_t)x++ * 4 + c] = b[1]; }
                    else {
                        int cnt = b[0];
                        if (cnt == 0) { fclose(f); throw std::runtime_error("bad .hdr run"); }
                        line[(size_t)x++ * 4 + c] = b[1];
                        for (int k = 1; k < cnt && x < w; k++) { int ch = fgetc(f); line[(size_t)x++ * 4 + c] = (unsigned char)ch; }
                    }
                }
            }
        }
        for (int x = 0; x < w; x++) rgbe2float(&line[(size_t)x * 4], &rgb[((size_t)y * w + x) * 3]);
    }
    fclose(f);
}

} // namespace

EnvironmentMap::EnvironmentMap(const std::string &filename) : name(filename)
{
    readHdr(filename, width, height, data);
    computeProbabilities();
}

EnvironmentMap::EnvironmentMap(int w, int h, const float *rgb) : width(w), height(h), name("memory")
{
    data.assign(rgb, rgb + (size_t)w * h * 3);
    computeProbabilities();
}

// reference: src/envmap.cpp:31-114
void EnvironmentMap::computeProbabilities()
{
    const int n = width * height;
    std::vector<float> scalars(n);
    for (int v = 0; v < height; v++) {
        float sinTh = std::sin(3.14159265358979323846f * float(v + 0.5f) / float(height));
        for (int u = 0; u < width; u++) {
            const float *p = &data[3 * ((size_t)v * width + u)];
            float lum = 0.212671f * p[0] + 0.715160f * p[1] + 0.072169f * p[2];
            scalars[v * width + u] = lum * sinTh;
        }
    }
    pdfTable.resize(n);
    float I = 0.0f;                                     // fp32 accumulation in index order (:58-59)
    for (int i = 0; i < n; i++) I += scalars[i] / (float)n;
    if (I == 0) for (int i = 0; i < n; i++) pdfTable[i] = 1.0f / float(n);
    else for (int i = 0; i < n; i++) pdfTable[i] = scalars[i] / I;

    // Vose's alias method with two LIFO work lists, split at p < 1 (:69-113).  Entries that end
    // with probability 1 never consult their alias; we point it at the entry itself.
    probTable.assign(n, 1.0f);
    aliasTable.resize(n);
    for (int i = 0; i < n; i++) aliasTable[i] = i;
    std::vector<std::pair<float, int>> small, large;
    for (int i = 0; i < n; i++) {
        float p = pdfTable[i];
        if (p < 1.0f) small.push_back({p, i}); else large.push_back({p, i});
    }
    while (!small.empty() && !large.empty()) {
        std::pair<float, int> l = small.back(), g = large.back();
        small.pop_back(); large.pop_back();
        probTable[l.second] = l.first;
        aliasTable[l.second] = g.second;
        float pg = (g.first + l.first) - 1.0f;
        if (pg < 1.0f) small.push_back({pg, g.second}); else large.push_back({pg, g.second});
    }
    for (auto &g : large) probTable[g.second] = 1.0f;
    for (auto &l : small) probTable[l.second] = 1.0f;
}

} // namespace fluctus
