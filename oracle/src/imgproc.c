/*
 * ORACLE (test infrastructure only -- see ../vo_oracle.h).
 *
 * CPU restatements of the small image-processing cv2 calls on the path:
 *   cv2.cvtColor(BGR2GRAY)      reference stereo_camera.py:44-47
 *   cv2.remap(INTER_LINEAR)     reference stereo_camera.py:29-33
 *   resize(INTER_LINEAR_EXACT)  used inside ORB's pyramid (features2d/src/orb.cpp)
 * following OpenCV 4.x imgproc/src/color_rgb.simd.hpp, imgwarp.cpp, resize.cpp.
 * Parity unpinned (no reference fixture exists for these stages).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../vo_oracle.h"

/* RGB2Gray<uchar>, OpenCV 4.x: 15-bit coefficients BY15=3735, GY15=19235, RY15=9798 */
void vo_ref_bgr2gray(const uint8_t* bgr, int w, int h, uint8_t* gray)
{
    for (size_t i = 0; i < (size_t)w * h; i++) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
    }
}

/* remapBilinear, fixed-point maps: map1 = integer source coords (x,y), map2 = 5+5 bit
 * fraction index; weights = (32-fx)(32-fy)*32 ... (INTER_REMAP_COEF_SCALE = 2^15, exact for
 * the linear table so no sum correction applies); BORDER_CONSTANT value 0. */
void vo_ref_remap_bilinear(const uint8_t* src, int sw, int sh, const int16_t* map1,
                           const uint16_t* map2, int w, int h, uint8_t* dst)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            size_t i = (size_t)y * w + x;
            int sx = map1[2 * i], sy = map1[2 * i + 1];
            int f = map2[i] & 1023, fx = f & 31, fy = f >> 5;
            int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
            int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            int p00 = 0, p01 = 0, p10 = 0, p11 = 0;
            if ((unsigned)sy < (unsigned)sh) {
                if ((unsigned)sx < (unsigned)sw) p00 = src[(size_t)sy * sw + sx];
                if ((unsigned)(sx + 1) < (unsigned)sw) p01 = src[(size_t)sy * sw + sx + 1];
            }
            if ((unsigned)(sy + 1) < (unsigned)sh) {
                if ((unsigned)sx < (unsigned)sw) p10 = src[(size_t)(sy + 1) * sw + sx];
                if ((unsigned)(sx + 1) < (unsigned)sw) p11 = src[(size_t)(sy + 1) * sw + sx + 1];
            }
            int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
            dst[i] = (uint8_t)(v > 255 ? 255 : v);
        }
}

/* ---- resize INTER_LINEAR_EXACT, 8-bit, 1 channel (resize_bitExact<uint8_t, interpolationLinear>) ---- */
static int round_half_even(double v) { return (int)nearbyint(v); }

typedef struct { int* ofs; uint16_t* c; int mn, mx; } lin_coeffs;

static void make_coeffs(int srcsize, int dstsize, lin_coeffs* lc)
{
    /* scale = 1/inv_scale with inv_scale = (double)dst/src, each an IEEE double op (softdouble) */
    double inv_scale = (double)dstsize / srcsize;
    double scale = 1.0 / inv_scale;
    lc->ofs = (int*)calloc(dstsize, sizeof(int));
    lc->c = (uint16_t*)calloc(2 * (size_t)dstsize, sizeof(uint16_t));
    lc->mn = 0; lc->mx = dstsize;
    for (int v = 0; v < dstsize; v++) {
        volatile double t = scale * ((double)v + 0.5); /* volatile: forbid FMA contraction */
        double fval = t - 0.5;
        int ival = (int)floor(fval);
        if (ival >= 0 && srcsize > 1) {
            if (ival < srcsize - 1) {
                lc->ofs[v] = ival;
                int c1 = round_half_even((fval - (double)ival) * 256.0);
                lc->c[2 * v + 1] = (uint16_t)c1;
                lc->c[2 * v] = (uint16_t)(256 - c1);
            } else {
                lc->ofs[v] = srcsize - 1;
                if (v < lc->mx) lc->mx = v;
            }
        } else if (v + 1 > lc->mn)
            lc->mn = v + 1;
    }
}

static void hresize_row(const uint8_t* s, int sw, const lin_coeffs* cx, int dw, uint16_t* out)
{
    int i = 0;
    for (; i < cx->mn; i++) out[i] = (uint16_t)(s[0] << 8);
    for (; i < cx->mx; i++) {
        const uint8_t* px = s + cx->ofs[i];
        out[i] = (uint16_t)(cx->c[2 * i] * px[0] + cx->c[2 * i + 1] * px[1]);
    }
    for (; i < dw; i++) out[i] = (uint16_t)(s[sw - 1] << 8);
}

void vo_ref_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst,
                                int dw, int dh, int dstride)
{
    lin_coeffs cx, cy;
    make_coeffs(sw, dw, &cx);
    make_coeffs(sh, dh, &cy);
    uint16_t* r0 = (uint16_t*)malloc(dw * sizeof(uint16_t));
    uint16_t* r1 = (uint16_t*)malloc(dw * sizeof(uint16_t));
    for (int dy = 0; dy < dh; dy++) {
        uint8_t* d = dst + (size_t)dy * dstride;
        if (dy < cy.mn || dy >= cy.mx) {
            const uint8_t* s = src + (size_t)(dy < cy.mn ? 0 : sh - 1) * sstride;
            hresize_row(s, sw, &cx, dw, r0);
            for (int i = 0; i < dw; i++) d[i] = (uint8_t)((r0[i] + 128) >> 8);
        } else {
            hresize_row(src + (size_t)cy.ofs[dy] * sstride, sw, &cx, dw, r0);
            hresize_row(src + (size_t)(cy.ofs[dy] + 1) * sstride, sw, &cx, dw, r1);
            uint32_t m0 = cy.c[2 * dy], m1 = cy.c[2 * dy + 1];
            for (int i = 0; i < dw; i++) {
                uint32_t v = r0[i] * m0 + r1[i] * m1;
                d[i] = (uint8_t)((v + 32768u) >> 16);
            }
        }
    }
    free(r0); free(r1); free(cx.ofs); free(cx.c); free(cy.ofs); free(cy.c);
}
