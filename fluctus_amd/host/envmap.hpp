// envmap.hpp -- HDR environment map + alias tables for importance sampling (host side).
//
// Mirrors the reference's EnvironmentMap (reference: src/envmap.hpp:10-62): Radiance RGBE (.hdr,
// RLE or flat) -> float RGB, then computeProbabilities() builds the 1-D pdf over texels
// (luminance * sin(theta)) and Vose alias/probability tables (reference: src/envmap.cpp:31-114).
#pragma once
#include <string>
#include <vector>

namespace fluctus {

class EnvironmentMap {
public:
    EnvironmentMap() {}
    explicit EnvironmentMap(const std::string &filename);          // reads .hdr, builds tables
    EnvironmentMap(int w, int h, const float *rgb);                 // from memory, builds tables

    std::string getName() const { return name; }
    float *getData() { return data.data(); }
    float *getProbTable() { return probTable.data(); }
    int *getAliasTable() { return aliasTable.data(); }
    float *getPdfTable() { return pdfTable.data(); }
    int getWidth() const { return width; }
    int getHeight() const { return height; }
    bool valid() const { return !data.empty() && !probTable.empty() && width * height > 0; }

private:
    void computeProbabilities();
    int width = 0, height = 0;
    std::string name;
    std::vector<float> data, pdfTable, probTable;
    std::vector<int> aliasTable;
};

} // namespace fluctus
