// jpeg.cpp -- JPEG decode for material textures, see texture.hpp.
//
// The reference decodes textures through DevIL -> libjpeg with its defaults (reference: src/texture.cpp:16-41):
// accurate integer IDCT (JDCT_ISLOW), "fancy" triangle-filter chroma upsampling, fixed-point YCbCr->RGB.  Those three
// stages are integer algorithms published in the JPEG standard (ITU-T T.81) and the IJG documentation, restated here so
// that the decoded RGBA8 bytes are IDENTICAL to libjpeg's (tests/test_host.py compares against PIL = libjpeg-turbo,
// whose SIMD paths are bit-exact with the C ones).  Supported: baseline / extended sequential and progressive Huffman
// (SOF0/1/2), 8-bit, 1 or 3 components, any integral sampling factors (h2v1 / h2v2 fancy upsampling, replication
// otherwise), restart intervals, non-interleaved scans.  Not supported (throw): arithmetic coding, 12-bit, CMYK, lossless.
#include "texture.hpp"
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace fluctus {
namespace {

const int ZIGZAG[64 + 16] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                             28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                             47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};   // 16 guard entries for corrupt runs

struct Huff {
    bool present = false;
    uint8_t vals[256];
    int mincode[17], maxcode[18], valptr[17];
    void build(const uint8_t *bits /*[1..16]*/, const uint8_t *v, int n)
    {
        memcpy(vals, v, (size_t)n);
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k; mincode[l] = code;
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Bits {                       // entropy-coded segment reader: 0xFF00 unstuffing, zero-fill once a marker is hit
    const uint8_t *p, *end;
    uint32_t buf = 0; int cnt = 0;
    bool marker = false;
    int getbit()
    {
        if (cnt == 0) {
            int b = 0;
            if (!marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int b2 = p < end ? *p : 0xD9;
                    if (b2 == 0) p++;
                    else { marker = true; p--; b = 0; }      // leave the marker for the scan loop; feed zeros
                }
            } else marker = true;
            buf = (uint32_t)b; cnt = 8;
        }
        cnt--;
        return (int)((buf >> cnt) & 1u);
    }
    int receive(int n) { int v = 0; while (n--) v = (v << 1) | getbit(); return v; }
    void align() { cnt = 0; }
};

inline int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

int decodeSym(Bits &br, const Huff &h)
{
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.getbit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return 0;                        // corrupt data: libjpeg warns and uses 0
}

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int wBlocks = 0, hBlocks = 0;              // allocated (padded to whole MCUs)
    int realW = 0, realH = 0;                  // downsampled_width / downsampled_height in samples
    int pred = 0;
    bool quantLatched = false;
    uint16_t q[64];
    std::vector<int16_t> coef;                 // wBlocks * hBlocks * 64, natural order
    std::vector<uint8_t> plane;                // (wBlocks*8) x (hBlocks*8)
};

// jidctint.c (accurate integer IDCT): 13-bit constants, 2 extra bits kept between the passes
void idctISlow(const int16_t *in, const uint16_t *q, uint8_t *out, int stride)
{
    const int CB = 13, P1 = 2;
    const int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
              F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    auto descale = [](long x, int n) { return (int)((x + (1L << (n - 1))) >> n); };
    int ws[64];
    for (int c = 0; c < 8; c++) {
        const int16_t *ip = in + c; const uint16_t *qp = q + c; int *wp = ws + c;
        if (ip[8] == 0 && ip[16] == 0 && ip[24] == 0 && ip[32] == 0 && ip[40] == 0 && ip[48] == 0 && ip[56] == 0) {
            int dc = (ip[0] * qp[0]) << P1;
            for (int r = 0; r < 8; r++) wp[8 * r] = dc;
            continue;
        }
        long z2 = ip[16] * qp[16], z3 = ip[48] * qp[48];
        long z1 = (z2 + z3) * F0_541;
        long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        z2 = ip[0] * qp[0]; z3 = ip[32] * qp[32];
        long tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = ip[56] * qp[56]; tmp1 = ip[40] * qp[40]; tmp2 = ip[24] * qp[24]; tmp3 = ip[8] * qp[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
        long z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        wp[0] = descale(tmp10 + tmp3, CB - P1); wp[56] = descale(tmp10 - tmp3, CB - P1);
        wp[8] = descale(tmp11 + tmp2, CB - P1); wp[48] = descale(tmp11 - tmp2, CB - P1);
        wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
        wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
    }
    // range_limit[x & 1023] of the IDCT: x taken modulo 1024 as a signed value, + 128, clamped to 0..255
    auto limit = [](int x) { int s = ((x + 512) & 1023) - 512 + 128; return (uint8_t)(s < 0 ? 0 : s > 255 ? 255 : s); };
    for (int r = 0; r < 8; r++) {
        const int *wp = ws + 8 * r; uint8_t *op = out + (size_t)r * stride;
        if (wp[1] == 0 && wp[2] == 0 && wp[3] == 0 && wp[4] == 0 && wp[5] == 0 && wp[6] == 0 && wp[7] == 0) {
            uint8_t dc = limit(descale(wp[0], P1 + 3));
            for (int c = 0; c < 8; c++) op[c] = dc;
            continue;
        }
        long z2 = wp[2], z3 = wp[6];
        long z1 = (z2 + z3) * F0_541;
        long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        long tmp0 = ((long)wp[0] + wp[4]) << CB, tmp1 = ((long)wp[0] - wp[4]) << CB;
        long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
        long z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int S = CB + P1 + 3;
        op[0] = limit(descale(tmp10 + tmp3, S)); op[7] = limit(descale(tmp10 - tmp3, S));
        op[1] = limit(descale(tmp11 + tmp2, S)); op[6] = limit(descale(tmp11 - tmp2, S));
        op[2] = limit(descale(tmp12 + tmp1, S)); op[5] = limit(descale(tmp12 - tmp1, S));
        op[3] = limit(descale(tmp13 + tmp0, S)); op[4] = limit(descale(tmp13 - tmp0, S));
    }
}

struct Decoder {
    std::vector<uint8_t> file;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1;
    bool progressive = false, sawSOF = false, sawJFIF = false, sawAdobe = false;
    int adobeTransform = 0, restartInterval = 0;
    Comp comp[4];
    uint16_t qt[4][64]; bool qtPresent[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int eobrun = 0;

    [[noreturn]] void fail(const std::string &m) { throw std::runtime_error("JPEG: " + m); }

    void parseDQT(const uint8_t *p, int len)
    {
        while (len > 0) {
            int pq = p[0] >> 4, tq = p[0] & 15;
            if (tq > 3 || pq > 1 || len < 1 + 64 * (pq + 1)) fail("bad DQT");
            for (int i = 0; i < 64; i++) qt[tq][ZIGZAG[i]] = pq ? (uint16_t)((p[1 + 2 * i] << 8) | p[2 + 2 * i]) : p[1 + i];
            qtPresent[tq] = true;
            p += 1 + 64 * (pq + 1); len -= 1 + 64 * (pq + 1);
        }
    }
    void parseDHT(const uint8_t *p, int len)
    {
        while (len > 0) {
            if (len < 17) fail("bad DHT");
            int tc = p[0] >> 4, th = p[0] & 15;
            if (tc > 1 || th > 3) fail("bad DHT table id");
            uint8_t bits[17]; bits[0] = 0; int n = 0;
            for (int i = 1; i <= 16; i++) { bits[i] = p[i]; n += p[i]; }
            if (n > 256 || len < 17 + n) fail("bad DHT counts");
            (tc ? ac[th] : dc[th]).build(bits, p + 17, n);
            p += 17 + n; len -= 17 + n;
        }
    }
    void parseSOF(const uint8_t *p, int len, int marker)
    {
        if (sawSOF) fail("multiple frames");
        if (len < 6 || p[0] != 8) fail("only 8-bit samples are supported");
        H = (p[1] << 8) | p[2]; W = (p[3] << 8) | p[4]; ncomp = p[5];
        if (W <= 0 || H <= 0) fail("empty image");
        if (ncomp != 1 && ncomp != 3) fail("only grayscale and 3-component images are supported");
        if (len < 6 + 3 * ncomp) fail("bad SOF");
        progressive = marker == 0xC2;
        for (int i = 0; i < ncomp; i++) {
            Comp &c = comp[i];
            c.id = p[6 + 3 * i]; c.h = p[7 + 3 * i] >> 4; c.v = p[7 + 3 * i] & 15; c.tq = p[8 + 3 * i];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) fail("bad component spec");
            if (c.h > hmax) hmax = c.h;
            if (c.v > vmax) vmax = c.v;
        }
        const int mcuW = (W + 8 * hmax - 1) / (8 * hmax), mcuH = (H + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < ncomp; i++) {
            Comp &c = comp[i];
            c.wBlocks = mcuW * c.h; c.hBlocks = mcuH * c.v;
            c.realW = (W * c.h + hmax - 1) / hmax; c.realH = (H * c.v + vmax - 1) / vmax;
            c.coef.assign((size_t)c.wBlocks * c.hBlocks * 64, 0);
        }
        sawSOF = true;
    }

    // ---- one block of one scan (T.81 F.2.2 / G.2; progressive refinement as in IJG jdphuff.c)
    void decodeBlock(Bits &br, Comp &c, int16_t *blk, int Ss, int Se, int Ah, int Al)
    {
        if (!progressive) {
            int s = decodeSym(br, dc[c.td]);
            c.pred += extend(br.receive(s), s);
            blk[0] = (int16_t)c.pred;
            for (int k = 1; k < 64; k++) {
                int rs = decodeSym(br, ac[c.ta]), r = rs >> 4; s = rs & 15;
                if (s) { k += r; blk[ZIGZAG[k]] = (int16_t)extend(br.receive(s), s); }
                else { if (r != 15) break; k += 15; }
            }
            return;
        }
        if (Ss == 0) {                                       // DC scan
            if (Ah == 0) { int s = decodeSym(br, dc[c.td]); c.pred += extend(br.receive(s), s); blk[0] = (int16_t)(c.pred * (1 << Al)); }
            else if (br.getbit()) blk[0] |= (int16_t)(1 << Al);
            return;
        }
        if (Ah == 0) {                                       // AC first pass
            if (eobrun > 0) { eobrun--; return; }
            for (int k = Ss; k <= Se; k++) {
                int rs = decodeSym(br, ac[c.ta]), r = rs >> 4, s = rs & 15;
                if (s) { k += r; blk[ZIGZAG[k]] = (int16_t)(extend(br.receive(s), s) * (1 << Al)); }
                else if (r == 15) k += 15;
                else { eobrun = 1 << r; if (r) eobrun += br.receive(r); eobrun--; break; }
            }
            return;
        }
        const int p1 = 1 << Al, m1 = -(1 << Al);              // AC refinement
        int k = Ss;
        if (eobrun == 0) {
            for (; k <= Se; k++) {
                int rs = decodeSym(br, ac[c.ta]), r = rs >> 4, s = rs & 15;
                if (s) s = br.getbit() ? p1 : m1;
                else if (r != 15) { eobrun = 1 << r; if (r) eobrun += br.receive(r); break; }
                do {
                    int16_t &co = blk[ZIGZAG[k]];
                    if (co != 0) { if (br.getbit() && (co & p1) == 0) co = (int16_t)(co + (co >= 0 ? p1 : m1)); }
                    else if (--r < 0) break;
                    k++;
                } while (k <= Se);
                if (s && k < 64) blk[ZIGZAG[k]] = (int16_t)s;
            }
        }
        if (eobrun > 0) {
            for (; k <= Se; k++) {
                int16_t &co = blk[ZIGZAG[k]];
                if (co != 0 && br.getbit() && (co & p1) == 0) co = (int16_t)(co + (co >= 0 ? p1 : m1));
            }
            eobrun--;
        }
    }

    // returns the position of the marker that ended the scan
    size_t decodeScan(size_t pos, const uint8_t *hdr, int len)
    {
        if (!sawSOF) fail("SOS before SOF");
        int ns = hdr[0];
        if (ns < 1 || ns > ncomp || len < 4 + 2 * ns) fail("bad SOS");
        Comp *sc[4];
        for (int i = 0; i < ns; i++) {
            int id = hdr[1 + 2 * i], t = hdr[2 + 2 * i];
            sc[i] = nullptr;
            for (int j = 0; j < ncomp; j++) if (comp[j].id == id) sc[i] = &comp[j];
            if (!sc[i]) fail("SOS names an unknown component");
            sc[i]->td = t >> 4; sc[i]->ta = t & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) fail("bad table selector");
            if (!sc[i]->quantLatched) {                       // the table in force at the component's first scan
                if (!qtPresent[sc[i]->tq]) fail("missing quantisation table");
                memcpy(sc[i]->q, qt[sc[i]->tq], sizeof(sc[i]->q)); sc[i]->quantLatched = true;
            }
        }
        const int Ss = hdr[1 + 2 * ns], Se = hdr[2 + 2 * ns], Ah = hdr[3 + 2 * ns] >> 4, Al = hdr[3 + 2 * ns] & 15;
        if (progressive) { if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13) fail("bad progressive scan parameters"); }
        else if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) fail("bad sequential scan parameters");
        for (int i = 0; i < ns; i++) {
            const bool needDC = !progressive || (Ss == 0 && Ah == 0), needAC = !progressive || Ss > 0;
            if (needDC && !dc[sc[i]->td].present) fail("missing DC Huffman table");
            if (needAC && !ac[sc[i]->ta].present) fail("missing AC Huffman table");
        }
        Bits br; br.p = file.data() + pos; br.end = file.data() + file.size();
        for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
        eobrun = 0;
        int mcusX, mcusY;
        if (ns == 1) { mcusX = (sc[0]->realW + 7) / 8; mcusY = (sc[0]->realH + 7) / 8; }      // non-interleaved: only the real blocks
        else { mcusX = (W + 8 * hmax - 1) / (8 * hmax); mcusY = (H + 8 * vmax - 1) / (8 * vmax); }
        int toRestart = restartInterval, nextRst = 0;
        for (int my = 0; my < mcusY; my++)
            for (int mx = 0; mx < mcusX; mx++) {
                if (restartInterval && toRestart == 0) {
                    br.align();
                    // expect RSTn at br.p (skip fill bytes); tolerate a missing one as libjpeg's resync does for the common case
                    const uint8_t *q = br.p;
                    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF)) q++;
                    if (q + 1 < br.end && q[1] == 0xD0 + nextRst) { br.p = q + 2; br.marker = false; }
                    else if (q + 1 < br.end && q[1] >= 0xD0 && q[1] <= 0xD7) { br.p = q + 2; br.marker = false; }
                    nextRst = (nextRst + 1) & 7;
                    for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
                    eobrun = 0; toRestart = restartInterval;
                }
                if (ns == 1) {
                    Comp &c = *sc[0];
                    decodeBlock(br, c, &c.coef[((size_t)my * c.wBlocks + mx) * 64], Ss, Se, Ah, Al);
                } else {
                    for (int i = 0; i < ns; i++) {
                        Comp &c = *sc[i];
                        for (int by = 0; by < c.v; by++)
                            for (int bx = 0; bx < c.h; bx++)
                                decodeBlock(br, c, &c.coef[((size_t)(my * c.v + by) * c.wBlocks + (mx * c.h + bx)) * 64], Ss, Se, Ah, Al);
                    }
                }
                if (restartInterval) toRestart--;
            }
        // find the marker that follows the entropy-coded data
        const uint8_t *q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
        return (size_t)(q - file.data());
    }

    void reconstruct(std::vector<uint8_t> &rgb)
    {
        for (int i = 0; i < ncomp; i++) {
            Comp &c = comp[i];
            if (!c.quantLatched) fail("a component never appeared in a scan");
            const int stride = c.wBlocks * 8;
            c.plane.assign((size_t)stride * c.hBlocks * 8, 0);
            for (int by = 0; by < c.hBlocks; by++)
                for (int bx = 0; bx < c.wBlocks; bx++)
                    idctISlow(&c.coef[((size_t)by * c.wBlocks + bx) * 64], c.q, &c.plane[(size_t)by * 8 * stride + bx * 8], stride);
        }
        // upsample every component to full resolution (jdsample.c: fullsize copy, h2v1 / h2v2 "fancy" triangle filters,
        // integral replication otherwise)
        std::vector<uint8_t> full[3];
        for (int i = 0; i < ncomp; i++) {
            Comp &c = comp[i];
            const int stride = c.wBlocks * 8, hexp = hmax / c.h, vexp = vmax / c.v;
            if (hmax % c.h || vmax % c.v) fail("fractional sampling ratios are not supported");
            const int ow = c.realW * hexp, oh = c.realH * vexp;
            if (ow < W || oh < H) fail("internal: upsampled plane too small");
            full[i].assign((size_t)ow * oh, 0);
            auto row = [&](int y) { if (y < 0) y = 0; if (y >= c.realH) y = c.realH - 1; return &c.plane[(size_t)y * stride]; };
            if (hexp == 1 && vexp == 1) {
                for (int y = 0; y < oh; y++) memcpy(&full[i][(size_t)y * ow], row(y), (size_t)ow);
            } else if (hexp == 2 && vexp == 1 && c.realW > 2) {
                for (int y = 0; y < oh; y++) {
                    const uint8_t *in = row(y); uint8_t *out = &full[i][(size_t)y * ow];
                    const int n = c.realW;
                    out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                    for (int x = 1; x < n - 1; x++) { int v3 = in[x] * 3; out[2 * x] = (uint8_t)((v3 + in[x - 1] + 1) >> 2); out[2 * x + 1] = (uint8_t)((v3 + in[x + 1] + 2) >> 2); }
                    out[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2); out[2 * n - 1] = in[n - 1];
                }
            } else if (hexp == 2 && vexp == 2 && c.realW > 2) {
                const int n = c.realW;
                for (int y = 0; y < c.realH; y++)
                    for (int v = 0; v < 2; v++) {
                        const uint8_t *in0 = row(y), *in1 = row(v == 0 ? y - 1 : y + 1);
                        uint8_t *out = &full[i][(size_t)(2 * y + v) * ow];
                        int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
                        out[0] = (uint8_t)((thiscol * 4 + 8) >> 4); out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                        lastcol = thiscol; thiscol = nextcol;
                        for (int x = 1; x < n - 1; x++) {
                            nextcol = in0[x + 1] * 3 + in1[x + 1];
                            out[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); out[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                            lastcol = thiscol; thiscol = nextcol;
                        }
                        out[2 * n - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); out[2 * n - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
                    }
            } else {
                for (int y = 0; y < oh; y++) {
                    const uint8_t *in = row(y / vexp); uint8_t *out = &full[i][(size_t)y * ow];
                    for (int x = 0; x < ow; x++) out[x] = in[x / hexp];
                }
            }
        }
        // colour conversion (jdcolor.c: 16-bit fixed point tables)
        rgb.assign((size_t)W * H * 3, 0);
        bool ycc = ncomp == 3;
        if (ncomp == 3) {
            if (sawAdobe) ycc = adobeTransform != 0;
            else if (!sawJFIF && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') ycc = false;
        }
        auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
        const int ow0 = comp[0].realW * (hmax / comp[0].h);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                uint8_t *d = &rgb[((size_t)y * W + x) * 3];
                const int Y = full[0][(size_t)y * ow0 + x];
                if (ncomp == 1) { d[0] = d[1] = d[2] = (uint8_t)Y; continue; }
                const int ow1 = comp[1].realW * (hmax / comp[1].h), ow2 = comp[2].realW * (hmax / comp[2].h);
                const int c1 = full[1][(size_t)y * ow1 + x], c2 = full[2][(size_t)y * ow2 + x];
                if (!ycc) { d[0] = (uint8_t)Y; d[1] = (uint8_t)c1; d[2] = (uint8_t)c2; continue; }
                const int cb = c1 - 128, cr = c2 - 128;
                const int half = 1 << 15;
                const int crR = (int)((91881L * cr + half) >> 16);            // FIX(1.40200)
                const int cbB = (int)((116130L * cb + half) >> 16);           // FIX(1.77200)
                const int g = (int)((-22554L * cb + half - 46802L * cr) >> 16);   // FIX(0.34414), FIX(0.71414)
                d[0] = clamp(Y + crR); d[1] = clamp(Y + g); d[2] = clamp(Y + cbB);
            }
    }

    void run(std::vector<uint8_t> &rgb)
    {
        if (file.size() < 4 || file[0] != 0xFF || file[1] != 0xD8) fail("not a JPEG (no SOI)");
        size_t pos = 2;
        bool done = false;
        while (!done) {
            while (pos < file.size() && file[pos] != 0xFF) pos++;
            while (pos < file.size() && file[pos] == 0xFF) pos++;
            if (pos >= file.size()) break;
            const int m = file[pos++];
            if (m == 0xD9) break;                                   // EOI
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;    // parameterless
            if (pos + 2 > file.size()) fail("truncated marker segment");
            const int len = ((file[pos] << 8) | file[pos + 1]) - 2;
            const uint8_t *p = &file[pos + 2];
            if (len < 0 || pos + 2 + (size_t)len > file.size()) fail("truncated marker segment");
            pos += 2 + (size_t)len;
            switch (m) {
            case 0xC0: case 0xC1: case 0xC2: parseSOF(p, len, m); break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                fail("unsupported JPEG process (lossless / hierarchical / arithmetic)");
            case 0xC4: parseDHT(p, len); break;
            case 0xDB: parseDQT(p, len); break;
            case 0xDD: if (len >= 2) restartInterval = (p[0] << 8) | p[1]; break;
            case 0xE0: if (len >= 5 && !memcmp(p, "JFIF", 5)) sawJFIF = true; break;
            case 0xEE: if (len >= 12 && !memcmp(p, "Adobe", 5)) { sawAdobe = true; adobeTransform = p[11]; } break;
            case 0xDA: pos = decodeScan(pos, p, len); break;
            default: break;                                          // APPn, COM, DNL...
            }
        }
        if (!sawSOF) fail("no frame header");
        reconstruct(rgb);
    }
};

} // namespace

void decodeJPEG(const uint8_t *data, size_t size, uint32_t *w, uint32_t *h, std::vector<uint8_t> &rgb)
{
    Decoder d;
    d.file.assign(data, data + size);
    d.run(rgb);
    *w = (uint32_t)d.W; *h = (uint32_t)d.H;
}

Texture loadJPEG(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> file;
    unsigned char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(f);
    uint32_t w = 0, h = 0; std::vector<uint8_t> rgb;
    decodeJPEG(file.data(), file.size(), &w, &h, rgb);
    Texture t; t.name = path; t.width = w; t.height = h; t.rgba.resize((size_t)w * h * 4);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *src = &rgb[(size_t)y * w * 3];
        uint8_t *dst = &t.rgba[(size_t)(h - 1 - y) * w * 4];            // lower-left origin: row 0 = bottom scanline
        for (uint32_t x = 0; x < w; x++) { dst[4 * x] = src[3 * x]; dst[4 * x + 1] = src[3 * x + 1]; dst[4 * x + 2] = src[3 * x + 2]; dst[4 * x + 3] = 255; }
    }
    return t;
}

Texture loadTexture(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    unsigned char sig[2] = {0, 0};
    size_t n = fread(sig, 1, 2, f);
    fclose(f);
    if (n == 2 && sig[0] == 0xFF && sig[1] == 0xD8) return loadJPEG(path);
    return loadPNG(path);
}

} // namespace fluctus
