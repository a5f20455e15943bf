/*
 * ORACLE (test infrastructure only -- see ../vo_oracle.h).
 *
 * 3-D and pose arithmetic of the path:
 *   cv2.reprojectImageTo3D        reference stereo_camera.py:52   (calib3d/src/calibration.cpp)
 *   bilinear_interpolate_pixels   reference stereo_odometer.py:50-79  (reference's own numpy; PINNED
 *                                 by tests/golden/g2_bilinear.npz)
 *   rigid_body_filter             reference stereo_odometer.py:82-105 (own numpy; PINNED by g3)
 *   cv2.estimateAffine3D(force_rotation=True)  reference stereo_odometer.py:190,204
 *                                 (calib3d/src/ptsetreg.cpp, Umeyama overload)
 *   cv2.Rodrigues                 reference stereo_odometer.py:212 (calib3d/src/calibration.cpp)
 * cv2 stages: parity unpinned.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../vo_oracle.h"

/* one pixel of reprojectImageTo3D (OpenCV 4.x): homg = Q*[x y d 1] accumulated left to right
 * in double; out = (float)homg[i], then out[i] = (float)(out[i] * (1/homg[3])). */
static void reproject_px(const double* Q, int x, int y, float dflt, float* out)
{
    double v[4] = { (double)x, (double)y, (double)dflt, 1.0 }, hg[4];
    for (int i = 0; i < 4; i++) {
        volatile double s = 0;
        for (int k = 0; k < 4; k++) { volatile double t = Q[i * 4 + k] * v[k]; s = s + t; }
        hg[i] = s;
    }
    double ialpha = 1.0 / hg[3];
    for (int i = 0; i < 3; i++) {
        float f = (float)hg[i];
        out[i] = (float)((double)f * ialpha);
    }
}

void vo_ref_reproject_to_3d(const float* disp, int w, int h, const double* Q, float* xyz)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            reproject_px(Q, x, y, disp[(size_t)y * w + x], xyz + ((size_t)y * w + x) * 3);
}

/* openVO bilinear_interpolate_pixels (reference stereo_odometer.py:50-79), exact numpy
 * semantics: taps in the order p00, p01 (y+1), p10 (x+1), p11; a tap is used iff it exists and
 * has no +-inf component (NaN is not excluded); weight computed in double, rounded to float32,
 * multiplied and accumulated in float32; den accumulated in double, rounded to float32 for the
 * final float32 division.  status 0 ok, 1 NaN in result, 2 no tap used (ZeroDivisionError). */
typedef void (*tap_fn)(const void* ctx, int x, int y, float* p);

static void bilinear_core(tap_fn tap, const void* ctx, int cw, int ch, const float* xy, int n,
                          float* xyz, uint8_t* status)
{
    for (int i = 0; i < n; i++) {
        double x = (double)xy[2 * i], y = (double)xy[2 * i + 1];
        int fx = (int)x, fy = (int)y;
        double rx = x - fx, ry = y - fy;
        const int tx[4] = { fx, fx, fx + 1, fx + 1 }, ty[4] = { fy, fy + 1, fy, fy + 1 };
        const double wt[4] = { (1 - rx) * (1 - ry), (1 - rx) * ry, rx * (1 - ry), rx * ry };
        float num[3] = { 0, 0, 0 };
        double den = 0;
        int used = 0;
        for (int k = 0; k < 4; k++) {
            if (tx[k] >= cw || ty[k] >= ch) continue;
            float p[3];
            tap(ctx, tx[k], ty[k], p);
            if (isinf(p[0]) || isinf(p[1]) || isinf(p[2])) continue;
            float wf = (float)wt[k];
            for (int c = 0; c < 3; c++) {
                volatile float t = wf * p[c];
                num[c] = num[c] + t; /* python int 0 + array == 0.0f + t */
            }
            den += wt[k];
            used++;
        }
        if (!used) {
            status[i] = 2;
            xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = NAN;
            continue;
        }
        float denf = (float)den;
        int nan = 0;
        for (int c = 0; c < 3; c++) {
            xyz[3 * i + c] = num[c] / denf;
            nan |= isnan(xyz[3 * i + c]);
        }
        status[i] = nan ? 1 : 0;
    }
}

typedef struct { const float* img; int w; } img3_ctx;
static void tap_img3(const void* c, int x, int y, float* p)
{
    const img3_ctx* k = (const img3_ctx*)c;
    memcpy(p, k->img + ((size_t)y * k->w + x) * 3, 3 * sizeof(float));
}

void vo_ref_bilinear_at(const float* img3d, int w, int h, const float* xy, int n, float* out,
                        uint8_t* status)
{
    img3_ctx c = { img3d, w };
    bilinear_core(tap_img3, &c, w, h, xy, n, out, status);
}

typedef struct { const int16_t* disp16; int w, x0, y0; const double* Q; } disp_ctx;
static void tap_disp(const void* c, int x, int y, float* p)
{
    const disp_ctx* k = (const disp_ctx*)c;
    int gx = x + k->x0, gy = y + k->y0;
    float d = (float)k->disp16[(size_t)gy * k->w + gx] / 16.0f;
    reproject_px(k->Q, gx, gy, d, p);
}

void vo_ref_points3d_at(const int16_t* disp16, int w, int h, const double* Q, int x0, int y0,
                        int x1, int y1, const float* xy, int n, float* xyz, uint8_t* status)
{
    /* numpy slice semantics of crop_to_valid_region_left (stereo_camera.py:35-37) */
    if (x1 > w) x1 = w;
    if (y1 > h) y1 = h;
    disp_ctx c = { disp16, w, x0, y0, Q };
    bilinear_core(tap_disp, &c, x1 - x0, y1 - y0, xy, n, xyz, status);
}

/* ---- 3x3 SVD by one-sided Jacobi; singular values descending, U and V orthonormal ---- */
static void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

void vo_ref_svd3(const double* A, double* U, double* w, double* Vt)
{
    double G[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(G, A, sizeof(G));
    for (int sweep = 0; sweep < 60; sweep++) {
        int rotated = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; i++) {
                    al += G[i * 3 + p] * G[i * 3 + p];
                    be += G[i * 3 + q] * G[i * 3 + q];
                    ga += G[i * 3 + p] * G[i * 3 + q];
                }
                if (fabs(ga) <= 1e-300 || fabs(ga) <= 2.2204460492503131e-16 * sqrt(al * be)) continue;
                rotated = 1;
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; i++) {
                    double gp = G[i * 3 + p], gq = G[i * 3 + q];
                    G[i * 3 + p] = c * gp - s * gq;
                    G[i * 3 + q] = s * gp + c * gq;
                    double vp = V[i * 3 + p], vq = V[i * 3 + q];
                    V[i * 3 + p] = c * vp - s * vq;
                    V[i * 3 + q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sv[3];
    int ord[3] = { 0, 1, 2 };
    for (int j = 0; j < 3; j++)
        sv[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (sv[ord[j]] > sv[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    double Uc[3][3], Vc[3][3];
    for (int j = 0; j < 3; j++) {
        int o = ord[j];
        w[j] = sv[o];
        for (int i = 0; i < 3; i++) {
            Vc[j][i] = V[i * 3 + o];
            Uc[j][i] = sv[o] > 0 ? G[i * 3 + o] / sv[o] : 0.0;
        }
    }
    /* complete U for (near-)zero singular values */
    const double tiny = w[0] * 1e-300 + 1e-300;
    if (w[1] <= tiny) {
        double a[3] = { 1, 0, 0 };
        if (fabs(Uc[0][0]) > 0.9) { a[0] = 0; a[1] = 1; }
        cross3(Uc[0], a, Uc[1]);
        double nn = sqrt(Uc[1][0] * Uc[1][0] + Uc[1][1] * Uc[1][1] + Uc[1][2] * Uc[1][2]);
        for (int i = 0; i < 3; i++) Uc[1][i] /= nn;
    }
    if (w[2] <= tiny || w[2] <= 1e-14 * w[0]) {
        cross3(Uc[0], Uc[1], Uc[2]);
        double nn = sqrt(Uc[2][0] * Uc[2][0] + Uc[2][1] * Uc[2][1] + Uc[2][2] * Uc[2][2]);
        if (nn > 0) for (int i = 0; i < 3; i++) Uc[2][i] /= nn;
    }
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) {
            U[i * 3 + j] = Uc[j][i];
            Vt[j * 3 + i] = Vc[j][i];
        }
}

static double det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}

int vo_ref_umeyama(const float* src, const float* dst, int m, int force_rotation, double* T,
                   double* scale_out)
{
    if (m < 3) return -1;
    const double one_over_n = 1.0 / m;
    double ms[3] = { 0, 0, 0 }, md[3] = { 0, 0, 0 };
    for (int i = 0; i < m; i++)
        for (int c = 0; c < 3; c++) { ms[c] += (double)src[3 * i + c]; md[c] += (double)dst[3 * i + c]; }
    for (int c = 0; c < 3; c++) { ms[c] *= one_over_n; md[c] *= one_over_n; }
    double cov[9] = { 0 }, var_from = 0;
    for (int i = 0; i < m; i++) {
        double s[3], d[3];
        for (int c = 0; c < 3; c++) { s[c] = (double)src[3 * i + c] - ms[c]; d[c] = (double)dst[3 * i + c] - md[c]; }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) cov[r * 3 + c] += d[r] * s[c];
        var_from += s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
    }
    for (int k = 0; k < 9; k++) cov[k] *= one_over_n;
    double U[9], w[3], Vt[9];
    vo_ref_svd3(cov, U, w, Vt);
    int nz = (w[0] != 0) + (w[1] != 0) + (w[2] != 0);
    if (nz < 2) return -2;
    double S[3] = { 1, 1, 1 };
    if (force_rotation && det3(U) * det3(Vt) < 0) S[2] = -1;
    double R[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += U[r * 3 + k] * S[k] * Vt[k * 3 + c];
            R[r * 3 + c] = a;
        }
    double scale = (w[0] * S[0] + w[1] * S[1] + w[2] * S[2]) * ((double)m / var_from);
    for (int r = 0; r < 3; r++) {
        double nt = 0;
        for (int c = 0; c < 3; c++) { T[r * 4 + c] = R[r * 3 + c]; nt += R[r * 3 + c] * ms[c]; }
        T[r * 4 + 3] = md[r] - scale * nt;
    }
    if (scale_out) *scale_out = scale;
    return 0;
}

void vo_ref_rodrigues(const double* Rin, double* r)
{
    double U[9], w[3], Vt[9], R[9];
    vo_ref_svd3(Rin, U, w, Vt);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += U[i * 3 + k] * Vt[k * 3 + j];
            R[i * 3 + j] = a;
        }
    r[0] = R[7] - R[5]; r[1] = R[2] - R[6]; r[2] = R[3] - R[1];
    double s = sqrt((r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t = (R[0] + 1) * 0.5;
        r[0] = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5;
        r[1] = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1. : 1.);
        t = (R[8] + 1) * 0.5;
        r[2] = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1. : 1.);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && (R[5] > 0) != (r[1] * r[2] > 0)) r[2] = -r[2];
        theta /= sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        for (int i = 0; i < 3; i++) r[i] *= theta;
    } else {
        double vth = 1 / (2 * s) * theta;
        for (int i = 0; i < 3; i++) r[i] *= vth;
    }
}

/* rigid_body_filter (reference stereo_odometer.py:82-105).  Inputs are float32 as produced by
 * point_clouds, so numpy evaluates the pairwise norms and the threshold compare in float32. */
void vo_ref_rigid_clique(const float* prev, const float* cur, int m, double thr, int64_t* clique)
{
    if (m <= 0) return;
    const float thr_f = (float)thr;
    uint8_t* cons = (uint8_t*)malloc((size_t)m * m);
    int* ncons = (int*)calloc(m, sizeof(int));
    uint8_t* compat = (uint8_t*)malloc(m);
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) {
            volatile float ax = cur[3 * i] - cur[3 * j], ay = cur[3 * i + 1] - cur[3 * j + 1], az = cur[3 * i + 2] - cur[3 * j + 2];
            volatile float bx = prev[3 * i] - prev[3 * j], by = prev[3 * i + 1] - prev[3 * j + 1], bz = prev[3 * i + 2] - prev[3 * j + 2];
            volatile float a2 = ax * ax, a3 = ay * ay, a4 = az * az, b2 = bx * bx, b3 = by * by, b4 = bz * bz;
            volatile float sa = a2 + a3, sb = b2 + b3;
            sa = sa + a4; sb = sb + b4;
            float na = sqrtf(sa), nb = sqrtf(sb);
            float dd = fabsf(na - nb);
            cons[(size_t)i * m + j] = dd < thr_f;
        }
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) ncons[j] += cons[(size_t)i * m + j];
    int seed = 0;
    for (int j = 1; j < m; j++) if (ncons[j] > ncons[seed]) seed = j;
    for (int j = 0; j < m; j++) { clique[j] = 0; compat[j] = cons[(size_t)seed * m + j]; }
    clique[seed] = 1;
    int csize = 1;
    for (int it = 0; it < m; it++) {
        long sum = 0;
        int sel = 0;
        long best = 0;
        /* candidates = compatible - clique (may be -1 where a clique member is not compatible);
         * np.sum(candidates)==0 stops; selected = argmax(num_consistent * candidates) */
        for (int j = 0; j < m; j++) sum += (long)compat[j] - (long)clique[j];
        if (sum == 0) break;
        best = (long)ncons[0] * ((long)compat[0] - (long)clique[0]);
        for (int j = 1; j < m; j++) {
            long v = (long)ncons[j] * ((long)compat[j] - (long)clique[j]);
            if (v > best) { best = v; sel = j; }
        }
        clique[sel] = 1;
        csize = 0; /* the reference compares against sum(clique), not clique_size */
        for (int j = 0; j < m; j++) csize += (int)clique[j];
        for (int i = 0; i < m; i++) {
            long dot = 0;
            for (int j = 0; j < m; j++) dot += cons[(size_t)i * m + j] * clique[j];
            compat[i] = dot >= csize;
        }
    }
    free(cons); free(ncons); free(compat);
}
