// rgbe_driver.cpp -- C entry point around the reference's own Radiance .hdr reader (src/rgbe/rgbe.cpp,
// compiled unmodified from /root/reference).  TEST INFRASTRUCTURE, oracle/_ref build only.
#include <cstdio>
#include "rgbe/rgbe.hpp"
extern "C" int ref_read_hdr(const char *path, int *w, int *h, float *out /* may be null: query size */)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    if (RGBE_ReadHeader(f, w, h, NULL) != 0) { fclose(f); return 2; }
    int rc = 0;
    if (out) rc = RGBE_ReadPixels_RLE(f, out, *w, *h);
    fclose(f);
    return rc;
}
