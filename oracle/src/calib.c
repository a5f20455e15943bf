/* TEST INFRASTRUCTURE ONLY (see vo_oracle.h).  CPU restatement of the one-off rectification setup behind
 * StereoCamera.__init__ (reference stereo_camera.py:17-22): cv2.stereoRectify with its Python defaults (flags =
 * CALIB_ZERO_DISPARITY, alpha = -1, newImageSize = imageSize) and cv2.initUndistortRectifyMap(..., CV_16SC2).
 *
 * Follows OpenCV 4.x (not on this box; restated from the published algorithm):
 *   modules/calib3d/src/calibration.cpp   cvStereoRectify, icvGetRectangles, cvRodrigues2
 *   modules/calib3d/src/undistort.dispatch.cpp (imgproc in 4.x)   cvUndistortPointsInternal, initUndistortRectifyMap
 *   modules/imgproc/src/imgwarp.cpp       convertMaps (CV_32FC1 x2 -> CV_16SC2 + CV_16UC1, INTER_BITS = 5)
 * Written scalar, pixel by pixel and step by step in the order of those functions -- on purpose NOT sharing structure with
 * openvo_amd/calib.py (vectorised numpy), which it checks.  Parity unpinned against a real cv2 (none here); the cv2 leg is
 * tests/test_cv2_crosscheck.py.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../vo_oracle.h"

void vo_ref_svd3(const double* A, double* U, double* w, double* Vt);   /* geom.c */
void vo_ref_rodrigues(const double* R, double* r);                    /* geom.c: matrix -> vector */

static void mat3_mul(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j];
            c[i * 3 + j] = s;
        }
}
static void mat3_vec(const double* a, const double* v, double* o)
{
    for (int i = 0; i < 3; i++) o[i] = a[i * 3] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
}
static void mat3_t(const double* a, double* t)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = a[j * 3 + i];
}

/* cvRodrigues2, vector -> matrix: R = cos(theta) I + (1 - cos(theta)) r r^T + sin(theta) [r]x */
static void rodrigues_vec(const double* om, double* R)
{
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    if (theta < 2.220446049250313e-16) {
        memset(R, 0, 9 * sizeof(double));
        R[0] = R[4] = R[8] = 1;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    const double rx = om[0] * it, ry = om[1] * it, rz = om[2] * it;
    const double rrt[9] = { rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz };
    const double rxm[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
    const double I[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * rxm[k];
}

static void dist14(const double* d, int n, double* k)
{
    memset(k, 0, 14 * sizeof(double));
    for (int i = 0; i < n && i < 14; i++) k[i] = d[i];
}

/* cvUndistortPointsInternal (5 fixed-point iterations, no termination criteria): pixel -> normalised, then optional
 * rotation R and projection P (fx, fy, cx, cy of a 3x4 matrix) */
static void undistort_point(double u, double v, const double* K, const double* k, const double* R, const double* P,
                            double* xo, double* yo)
{
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double x0 = (u - cx) / fx, y0 = (v - cy) / fy;
    double x = x0, y = y0;
    int any = 0;
    for (int i = 0; i < 14; i++) any |= k[i] != 0;
    if (any)
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) icdist = 1;          /* "test: undistortPoints.regression_14583" */
            const double dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
    if (R) {
        const double xx = R[0] * x + R[1] * y + R[2], yy = R[3] * x + R[4] * y + R[5], ww = 1. / (R[6] * x + R[7] * y + R[8]);
        x = xx * ww;
        y = yy * ww;
    }
    if (P) {
        x = x * P[0] + P[2];       /* fx', cx' of the row-major 3x4 */
        y = y * P[5] + P[6];
    }
    *xo = x;
    *yo = y;
}

/* icvGetRectangles: 9 x 9 grid of source pixels through undistortPoints(R, newCameraMatrix), kept as CvPoint2D32f; the inner
 * rectangle's edges are the max / min over the grid's border points, cv::Rect_<float>(x0, y0, x1 - x0, y1 - y0) */
static void inner_rectangle(const double* K, const double* k, const double* R, const double* P, int w, int h, float* rect)
{
    const int N = 9;
    float px[81], py[81];
    for (int y = 0, i = 0; y < N; y++)
        for (int x = 0; x < N; x++, i++) {
            const float sx = (float)((double)x * (w - 1) / (N - 1)), sy = (float)((double)y * (h - 1) / (N - 1));
            double ox, oy;
            undistort_point(sx, sy, K, k, R, P, &ox, &oy);
            px[i] = (float)ox;
            py[i] = (float)oy;
        }
    float ix0 = -3.4e38f, ix1 = 3.4e38f, iy0 = -3.4e38f, iy1 = 3.4e38f;
    for (int y = 0, i = 0; y < N; y++)
        for (int x = 0; x < N; x++, i++) {
            if (x == 0 && px[i] > ix0) ix0 = px[i];
            if (x == N - 1 && px[i] < ix1) ix1 = px[i];
            if (y == 0 && py[i] > iy0) iy0 = py[i];
            if (y == N - 1 && py[i] < iy1) iy1 = py[i];
        }
    rect[0] = ix0; rect[1] = iy0; rect[2] = ix1 - ix0; rect[3] = iy1 - iy0;
}

/* K1, K2: 3x3 row-major; d1, d2: n1 / n2 distortion coefficients (NULL / 0 = none); R 3x3, T 3.
 * Outputs: R1, R2 (9), P1, P2 (12), Q (16), roi1, roi2 (x, y, w, h). */
void vo_ref_stereo_rectify(const double* K1, const double* d1, int n1, const double* K2, const double* d2, int n2, int w, int h,
                           const double* R, const double* T, double* R1, double* R2, double* P1, double* P2, double* Q,
                           int* roi1, int* roi2)
{
    double om[3], r_r[9], t[3], uu[3] = { 0, 0, 0 }, ww[3], wR[9], r_rT[9];
    /* rotate both cameras by half the relative rotation: r_r = Rodrigues(-0.5 om) */
    vo_ref_rodrigues(R, om);
    for (int i = 0; i < 3; i++) om[i] *= -0.5;
    rodrigues_vec(om, r_r);
    mat3_vec(r_r, T, t);
    const int idx = fabs(t[0]) > fabs(t[1]) ? 0 : 1;
    const double c = t[idx], nt = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    uu[idx] = c > 0 ? 1 : -1;
    /* global rotation that makes the baseline horizontal (vertical): ww = t x uu scaled to the angle between them */
    ww[0] = t[1] * uu[2] - t[2] * uu[1];
    ww[1] = t[2] * uu[0] - t[0] * uu[2];
    ww[2] = t[0] * uu[1] - t[1] * uu[0];
    const double nw = sqrt(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
    if (nw > 0.0) {
        const double sc = acos(fabs(c) / nt) / nw;
        for (int i = 0; i < 3; i++) ww[i] *= sc;
    }
    rodrigues_vec(ww, wR);
    mat3_t(r_r, r_rT);
    mat3_mul(wR, r_rT, R1);
    mat3_mul(wR, r_r, R2);
    mat3_vec(R2, T, t);

    /* new focal length: the average of the two cameras' focal lengths along the OTHER axis (OpenCV >= 3.4.8), times
     * newImgSize / imageSize = 1 */
    const double fc_new = (K1[(idx ^ 1) * 4] + K2[(idx ^ 1) * 4]) * 0.5;
    double k1[14], k2[14], cc[2][2];
    dist14(d1, n1, k1);
    dist14(d2, n2, k2);
    for (int cam = 0; cam < 2; cam++) {
        const double* K = cam ? K2 : K1;
        const double* kk = cam ? k2 : k1;
        const double* Rk = cam ? R2 : R1;
        /* the four image corners: undistorted (CvPoint2D32f), made homogeneous, rotated and projected with fc_new
         * (cvProjectPoints2 into CvPoint2D32f), averaged in double */
        double ax = 0, ay = 0;
        for (int i = 0; i < 4; i++) {
            const float sx = (float)((i % 2) * (w - 1)), sy = (float)((i / 2) * (h - 1));
            double ux, uy;
            undistort_point(sx, sy, K, kk, 0, 0, &ux, &uy);
            const double x = (float)ux, y = (float)uy;
            const double X = Rk[0] * x + Rk[1] * y + Rk[2], Y = Rk[3] * x + Rk[4] * y + Rk[5], Z = Rk[6] * x + Rk[7] * y + Rk[8];
            ax += (float)(fc_new * X / Z);
            ay += (float)(fc_new * Y / Z);
        }
        cc[cam][0] = (w - 1) / 2. - ax * 0.25;
        cc[cam][1] = (h - 1) / 2. - ay * 0.25;
    }
    /* CALIB_ZERO_DISPARITY: the two principal points become their mean */
    cc[0][0] = cc[1][0] = (cc[0][0] + cc[1][0]) * 0.5;
    cc[0][1] = cc[1][1] = (cc[0][1] + cc[1][1]) * 0.5;
    memset(P1, 0, 12 * sizeof(double));
    memset(P2, 0, 12 * sizeof(double));
    P1[0] = P1[5] = P2[0] = P2[5] = fc_new;
    P1[2] = cc[0][0]; P1[6] = cc[0][1]; P1[10] = 1;
    P2[2] = cc[1][0]; P2[6] = cc[1][1]; P2[10] = 1;
    P2[idx * 4 + 3] = t[idx] * fc_new;

    /* alpha < 0: principal points and focal length stay; the valid ROIs are the inner rectangles, shifted by
     * (cx - cx0) = 0 and scaled by s = 1, rounded outwards-in (cvCeil of the origin, cvFloor of the extent) and
     * intersected with the image */
    float in1[4], in2[4];
    inner_rectangle(K1, k1, R1, P1, w, h, in1);
    inner_rectangle(K2, k2, R2, P2, w, h, in2);
    for (int cam = 0; cam < 2; cam++) {
        const float* in = cam ? in2 : in1;
        int* roi = cam ? roi2 : roi1;
        const double cx0 = cc[cam][0], cy0 = cc[cam][1];
        const int x = (int)ceil(((double)in[0] - cx0) * 1.0 + cx0), y = (int)ceil(((double)in[1] - cy0) * 1.0 + cy0);
        const int ew = (int)floor((double)in[2] * 1.0), eh = (int)floor((double)in[3] * 1.0);
        const int x0 = x > 0 ? x : 0, y0 = y > 0 ? y : 0;
        const int x1 = x + ew < w ? x + ew : w, y1 = y + eh < h ? y + eh : h;
        if (x1 <= x0 || y1 <= y0) roi[0] = roi[1] = roi[2] = roi[3] = 0;
        else { roi[0] = x0; roi[1] = y0; roi[2] = x1 - x0; roi[3] = y1 - y0; }
    }
    /* Q: [1 0 0 -cx; 0 1 0 -cy; 0 0 0 f; 0 0 -1/Tx (cx - cx')/Tx] */
    memset(Q, 0, 16 * sizeof(double));
    Q[0] = 1; Q[3] = -cc[0][0];
    Q[5] = 1; Q[7] = -cc[0][1];
    Q[11] = fc_new;
    Q[14] = -1. / t[idx];
    Q[15] = (idx == 0 ? cc[0][0] - cc[1][0] : cc[0][1] - cc[1][1]) / t[idx];
}

static void inv3(const double* m, double* o)
{
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C, id = 1. / det;
    o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
    o[3] = B * id; o[4] = (a * i - c * g) * id; o[5] = -(a * f - c * d) * id;
    o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
}

static int sat_round(double v)      /* cv::saturate_cast<int>(double): cvRound (nearest even), clamped */
{
    if (v > 2147483647.) v = 2147483647.;
    if (v < -2147483647.) v = -2147483647.;
    return (int)nearbyint(v);
}

/* initUndistortRectifyMap(K, dist, R, P, size, CV_16SC2): for every destination pixel the source position through the
 * inverse of (P[:3,:3] R), the lens model, K; then fixed point with 5 fractional bits:
 * map1 = (iu >> 5, iv >> 5) as int16, map2 = (iv & 31) * 32 + (iu & 31) */
void vo_ref_init_undistort_rectify_map(const double* K, const double* d, int nd, const double* R, const double* P, int w, int h,
                                       int16_t* map1, uint16_t* map2)
{
    double k[14], A[9], AR[9], ir[9];
    dist14(d, nd, k);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) A[i * 3 + j] = P[i * 4 + j];
    mat3_mul(A, R, AR);
    inv3(AR, ir);
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    for (int i = 0; i < h; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; j++) {
            /* (OpenCV advances _x += ir[0] per column; the products below are that sum without its accumulated rounding --
             * the fixed-point result agrees except on isolated rounding ties, which the comparison allows for) */
            const double xs = _x + j * ir[0], ys = _y + j * ir[3], ws = _w + j * ir[6];
            const double iw = 1. / ws, x = xs * iw, y = ys * iw;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2);
            const double xd = x * kr + k[2] * _2xy + k[3] * (r2 + 2 * x2) + k[8] * r2 + k[9] * r2 * r2;
            const double yd = y * kr + k[2] * (r2 + 2 * y2) + k[3] * _2xy + k[10] * r2 + k[11] * r2 * r2;
            const double u = fx * xd + u0, v = fy * yd + v0;
            const int iu = sat_round(u * 32), iv = sat_round(v * 32);
            map1[((size_t)i * w + j) * 2] = (int16_t)(iu >> 5);
            map1[((size_t)i * w + j) * 2 + 1] = (int16_t)(iv >> 5);
            map2[(size_t)i * w + j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}
