/*
 * ORACLE (test infrastructure only -- see ../vo_oracle.h).
 *
 * CPU restatement of cv2.ORB_create(nfeatures).detectAndCompute(img, mask)
 * (reference stereo_odometer.py:22,117) with OpenCV's defaults scaleFactor=1.2f,
 * nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2, HARRIS_SCORE, patchSize=31,
 * fastThreshold=20.  Follows OpenCV 4.x features2d/src/orb.cpp (ORB_Impl::
 * detectAndCompute, computeKeyPoints, HarrisResponses, ICAngles,
 * computeOrbDescriptors), fast.cpp / fast_score.cpp (FAST_t<16>, cornerScore<16>),
 * keypoint.cpp (runByImageBorder, runByPixelsMask, retainBest),
 * core mathfuncs_core.simd.hpp (fastAtan2), imgproc filter (sepFilter2D 8-bit path).
 * Keypoint ORDER is canonical (octave, y, x) -- OpenCV's is nth_element-dependent.
 * Parity unpinned (no reference fixture exists for this stage).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../vo_oracle.h"

#define NLEVELS 8
#define EDGE_THRESHOLD 31
#define PATCH_SIZE 31
#define HALF_PATCH 15
#define FAST_THRESHOLD 20
#define HARRIS_K 0.04f

static const int8_t bit_pattern_31[256 * 4] = {
#include "../../include/vo_orb_pattern.inc"
};

static int cv_round_f(float v) { return (int)nearbyintf(v); }
static int cv_round_d(double v) { return (int)nearbyint(v); }

static float level_scale(int level) { return (float)pow((double)1.2f, (double)level); }

/* Two details of orb.cpp are recalled, not read (no OpenCV source or binary on this box); both are
 * switchable so that their weight can be MEASURED (tests/test_orb_variants.py, tests/orb_variants.py):
 *   bit 0: level size as cvRound(cols / scale) instead of cvRound(cols * (1.f / scale))
 *   bit 1: cosf / sinf (what `cos(float)` resolves to under libstdc++) instead of (float)cos((double)) */
static int g_variant = 0;
void vo_ref_orb_set_variant(int flags) { g_variant = flags; }
int vo_ref_orb_get_variant(void) { return g_variant; }

int vo_ref_orb_level_size(int w, int h, int level, int* lw, int* lh)
{
    float scale = level_scale(level);
    if (g_variant & 1) {
        *lw = cv_round_f((float)w / scale);
        *lh = cv_round_f((float)h / scale);
    } else {
        float inv = 1.0f / scale;
        *lw = cv_round_f((float)w * inv);
        *lh = cv_round_f((float)h * inv);
    }
    return 0;
}

/* ---------- FAST-9/16 ---------- */
static const int ring_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int ring_dy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

static int fast_is_corner(const uint8_t* p, const int* off, int t)
{
    int v = p[0], vt0 = v - t, vt1 = v + t, c0 = 0, c1 = 0;
    for (int k = 0; k < 25; k++) {
        int x = p[off[k & 15]];
        if (x < vt0) { if (++c0 > 8) return 1; } else c0 = 0;
        if (x > vt1) { if (++c1 > 8) return 1; } else c1 = 0;
    }
    return 0;
}

static int fast_corner_score(const uint8_t* p, const int* off, int threshold)
{
    int d[25], v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[off[k & 15]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 8; j++) if (d[k + j] < a) a = d[k + j];
        int m = a < d[k] ? a : d[k]; if (m > a0) a0 = m;
        m = a < d[k + 9] ? a : d[k + 9]; if (m > a0) a0 = m;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 8; j++) if (d[k + j] > b) b = d[k + j];
        int m = b > d[k] ? b : d[k]; if (m < b0) b0 = m;
        m = b > d[k + 9] ? b : d[k + 9]; if (m < b0) b0 = m;
    }
    return -b0 - 1;
}

/* score map after 3x3 non-max suppression (strict >), 0 elsewhere; scores are stored in
 * uchar as OpenCV does.  Detection rows/cols 3..dim-4. */
void vo_ref_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold,
                           uint8_t* out)
{
    uint8_t* sc = (uint8_t*)calloc((size_t)w * h, 1);
    int off[16];
    for (int k = 0; k < 16; k++) off[k] = ring_dy[k] * stride + ring_dx[k];
    memset(out, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (fast_is_corner(p, off, threshold))
                sc[(size_t)y * w + x] = (uint8_t)fast_corner_score(p, off, threshold);
        }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = sc[(size_t)y * w + x];
            if (!s) continue;
            const uint8_t* q = sc + (size_t)y * w + x;
            if (s > q[-1] && s > q[1] && s > q[-w - 1] && s > q[-w] && s > q[-w + 1] &&
                s > q[w - 1] && s > q[w] && s > q[w + 1])
                out[(size_t)y * w + x] = (uint8_t)s;
        }
    free(sc);
}

/* ---------- helpers ---------- */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

/* GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8-bit.
 * mode 0: sepFilter2D with the float kernel scaled by 256 and rounded -> [18,34,49,55,49,34,18]
 *         (the path taken when src is a sub-matrix without BORDER_ISOLATED, as in ORB);
 * mode 1: ufixedpoint16 bit-exact kernel (error-diffused, sums to 256) [18,34,48,56,48,34,18].
 * Both: int row pass, column pass, (v + 2^15) >> 16, saturate. */
static void gauss7(const uint8_t* src, int w, int h, int stride, int mode, uint8_t* dst)
{
    static const int k0[7] = { 18, 34, 49, 55, 49, 34, 18 };
    static const int k1[7] = { 18, 34, 48, 56, 48, 34, 18 };
    const int* k = mode ? k1 : k0;
    int* tmp = (int*)malloc((size_t)w * h * sizeof(int));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = -3; i <= 3; i++) s += k[i + 3] * src[(size_t)y * stride + reflect101(x + i, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = -3; i <= 3; i++) s += k[i + 3] * tmp[(size_t)reflect101(y + i, h) * w + x];
            s = (s + (1 << 15)) >> 16;
            dst[(size_t)y * w + x] = (uint8_t)(s > 255 ? 255 : s);
        }
    free(tmp);
}

static float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

typedef struct { int x, y; float resp; } cand_t;

/* KeyPointsFilter::retainBest as a SET: keep everything whose response is >= the n-th
 * largest response (ties kept); order left untouched (row-major). */
static int retain_best(cand_t* c, int n, int keep)
{
    if (keep < 0 || n <= keep) return n;
    if (keep == 0) return 0;
    int m = 0;
    uint8_t* ok = (uint8_t*)malloc(n);
    for (int i = 0; i < n; i++) {
        int greater = 0;
        for (int j = 0; j < n; j++) greater += c[j].resp > c[i].resp;
        ok[i] = greater < keep;
    }
    for (int i = 0; i < n; i++) if (ok[i]) c[m++] = c[i];
    free(ok);
    return m;
}

int vo_ref_orb_detect_and_compute(const uint8_t* img, int w, int h, int stride,
                                  const uint8_t* mask, int mask_stride, int nfeatures,
                                  int blur_mode, float* kp_xy, float* kp_size, float* kp_angle,
                                  float* kp_response, int32_t* kp_octave, uint8_t* desc, int cap,
                                  int* n_out)
{
    *n_out = 0;
    /* per-level quotas */
    int quota[NLEVELS];
    {
        float factor = (float)(1.0 / (double)1.2f);
        float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)NLEVELS));
        int sum = 0;
        for (int l = 0; l < NLEVELS - 1; l++) {
            quota[l] = cv_round_f(nd);
            sum += quota[l];
            nd *= factor;
        }
        quota[NLEVELS - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    }
    /* umax: row half-widths of the circular patch */
    int umax[HALF_PATCH + 2];
    {
        int vmax = (int)floor(HALF_PATCH * sqrt(2.f) / 2 + 1);
        int vmin = (int)ceil(HALF_PATCH * sqrt(2.f) / 2);
        for (int v = 0; v <= vmax; v++) umax[v] = cv_round_d(sqrt((double)HALF_PATCH * HALF_PATCH - v * v));
        for (int v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }
    /* pyramid: level 0 = input, level l = resize(level l-1, INTER_LINEAR_EXACT); mask likewise
     * + threshold(254, TOZERO) */
    uint8_t* lv[NLEVELS];
    uint8_t* mk[NLEVELS];
    int lw[NLEVELS], lh[NLEVELS];
    float lscale[NLEVELS];
    for (int l = 0; l < NLEVELS; l++) {
        lscale[l] = level_scale(l);
        vo_ref_orb_level_size(w, h, l, &lw[l], &lh[l]);
        lv[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l] + 16);
        mk[l] = mask ? (uint8_t*)malloc((size_t)lw[l] * lh[l] + 16) : NULL;
        if (l == 0) {
            for (int y = 0; y < h; y++) {
                memcpy(lv[0] + (size_t)y * w, img + (size_t)y * stride, w);
                if (mask) memcpy(mk[0] + (size_t)y * w, mask + (size_t)y * mask_stride, w);
            }
        } else {
            vo_ref_resize_linear_exact(lv[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], lv[l], lw[l], lh[l], lw[l]);
            if (mask) {
                vo_ref_resize_linear_exact(mk[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], mk[l], lw[l], lh[l], lw[l]);
                for (size_t i = 0; i < (size_t)lw[l] * lh[l]; i++) mk[l][i] = mk[l][i] > 254 ? mk[l][i] : 0;
            }
        }
    }

    int total = 0;
    const float hscale = 1.f / ((1 << 2) * 7 * 255.f);
    const float hscale4 = hscale * hscale * hscale * hscale;
    for (int l = 0; l < NLEVELS; l++) {
        const int W = lw[l], H = lh[l];
        const uint8_t* I = lv[l];
        if (W <= 2 * EDGE_THRESHOLD || H <= 2 * EDGE_THRESHOLD) continue; /* runByImageBorder clears */
        uint8_t* sm = (uint8_t*)malloc((size_t)W * H);
        vo_ref_fast_score_map(I, W, H, W, FAST_THRESHOLD, sm);
        int nc = 0, capc = 1024;
        cand_t* c = (cand_t*)malloc(capc * sizeof(cand_t));
        for (int y = EDGE_THRESHOLD; y < H - EDGE_THRESHOLD; y++)
            for (int x = EDGE_THRESHOLD; x < W - EDGE_THRESHOLD; x++) {
                int s = sm[(size_t)y * W + x];
                if (!s) continue;
                if (mk[l] && mk[l][(size_t)y * W + x] == 0) continue;
                if (nc == capc) { capc *= 2; c = (cand_t*)realloc(c, capc * sizeof(cand_t)); }
                c[nc].x = x; c[nc].y = y; c[nc].resp = (float)s; nc++;
            }
        free(sm);
        nc = retain_best(c, nc, 2 * quota[l]);
        /* Harris response, 7x7 block of Sobel-like gradients on the (unblurred) level */
        for (int i = 0; i < nc; i++) {
            int a = 0, b = 0, cc = 0;
            for (int dy = -3; dy <= 3; dy++)
                for (int dx = -3; dx <= 3; dx++) {
                    const uint8_t* p = I + (size_t)(c[i].y + dy) * W + c[i].x + dx;
                    int Ix = (p[1] - p[-1]) * 2 + (p[-W + 1] - p[-W - 1]) + (p[W + 1] - p[W - 1]);
                    int Iy = (p[W] - p[-W]) * 2 + (p[W - 1] - p[-W - 1]) + (p[W + 1] - p[-W + 1]);
                    a += Ix * Ix; b += Iy * Iy; cc += Ix * Iy;
                }
            volatile float t1 = (float)a * (float)b;
            volatile float t2 = (float)cc * (float)cc;
            volatile float s = (float)a + (float)b;
            volatile float t3 = HARRIS_K * s;
            volatile float t4 = t3 * s;
            volatile float t5 = t1 - t2;
            volatile float t6 = t5 - t4;
            c[i].resp = t6 * hscale4;
        }
        nc = retain_best(c, nc, quota[l]);
        /* blurred level for the descriptor */
        uint8_t* B = (uint8_t*)malloc((size_t)W * H);
        gauss7(I, W, H, W, blur_mode, B);
        for (int i = 0; i < nc; i++) {
            if (total >= cap) break;
            const uint8_t* ctr = I + (size_t)c[i].y * W + c[i].x;
            int m01 = 0, m10 = 0;
            for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m10 += u * ctr[u];
            for (int v = 1; v <= HALF_PATCH; ++v) {
                int vs = 0, d = umax[v];
                for (int u = -d; u <= d; ++u) {
                    int vp = ctr[u + v * W], vm = ctr[u - v * W];
                    vs += vp - vm;
                    m10 += u * (vp + vm);
                }
                m01 += v * vs;
            }
            float angle = fast_atan2_deg((float)m01, (float)m10);
            float px = (float)c[i].x * lscale[l], py = (float)c[i].y * lscale[l];
            kp_xy[2 * total] = px; kp_xy[2 * total + 1] = py;
            kp_size[total] = PATCH_SIZE * lscale[l];
            kp_angle[total] = angle;
            kp_response[total] = c[i].resp;
            kp_octave[total] = l;
            /* descriptor */
            float iscale = 1.f / lscale[l];
            float ar = angle * (float)(3.1415926535897932384626433832795 / 180.f);
            float ca, sa;
            if (g_variant & 2) { ca = cosf(ar); sa = sinf(ar); }
            else { ca = (float)cos((double)ar); sa = (float)sin((double)ar); }
            int cx = cv_round_f(px * iscale), cy = cv_round_f(py * iscale);
            const uint8_t* bc = B + (size_t)cy * W + cx;
            uint8_t* dsc = desc + (size_t)total * 32;
            for (int j = 0; j < 32; j++) {
                int val = 0;
                for (int k = 0; k < 8; k++) {
                    const int8_t* pt = bit_pattern_31 + (j * 8 + k) * 4;
                    volatile float xa = pt[0] * ca, xb = pt[1] * sa, ya = pt[0] * sa, yb = pt[1] * ca;
                    int ix0 = cv_round_f(xa - xb), iy0 = cv_round_f(ya + yb);
                    volatile float xc = pt[2] * ca, xd = pt[3] * sa, yc = pt[2] * sa, yd = pt[3] * ca;
                    int ix1 = cv_round_f(xc - xd), iy1 = cv_round_f(yc + yd);
                    int t0 = bc[iy0 * W + ix0], t1 = bc[iy1 * W + ix1];
                    val |= (t0 < t1) << k;
                }
                dsc[j] = (uint8_t)val;
            }
            total++;
        }
        free(B); free(c);
    }
    for (int l = 0; l < NLEVELS; l++) { free(lv[l]); if (mk[l]) free(mk[l]); }
    *n_out = total;
    return 0;
}
