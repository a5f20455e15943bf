/*
 * ORACLE (test infrastructure only -- see ../vo_oracle.h).
 *
 * RANSAC solvePnP hypothesis generation + reprojection scoring (the north star's "solvePnP
 * hypothesis-scoring loop"; BASELINE config 2 names "ORB+SGBM+PnP").
 * There is NO openVO counterpart (the reference fits its pose with a closed-form 3-D/3-D Umeyama,
 * SURVEY M1): this is the build's own definition, restated here so that the HIP kernels can be
 * checked bit for bit.  "Parity unpinned" by construction; pinned instead by geometric known
 * answers (a synthetic pose must come back).
 *
 *   sample(h, j)  the same counter-based hash RNG as the essential-matrix loop, 4 distinct indices
 *   hypothesis    P3P on the first three correspondences in float64, formulated on the depths
 *                 l1..l3 along the unit bearings:  li^2 + lj^2 - 2 bij li lj = aij.
 *                 The pencil D1 + g D2 of the two homogeneous conics (D1 = a23 M12 - a12 M23,
 *                 D2 = a23 M13 - a13 M23) is made singular (real root of a cubic, by bisection:
 *                 only + - * / sqrt, so it restates exactly), the singular conic splits into two
 *                 lines, each line cuts D1 in <= 2 depth ratios -> <= 4 poses, polished by three
 *                 Gauss-Newton steps on the depths.  The fourth sample picks among them (lowest
 *                 normalised reprojection error, positive depth).
 *   score         P = K [R|t] rounded to float32; a point is an inlier iff it lies in front of the
 *                 camera and (xc - u zc)^2 + (yc - v zc)^2 < thr^2 zc^2, float32, fixed order
 *   winner        most inliers, ties -> lowest hypothesis index
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../vo_oracle.h"

static uint32_t lowbias32(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

void vo_ref_ransac_sample4(uint32_t seed, int h, int n, int* idx)
{
    for (int j = 0; j < 4; j++) {
        uint32_t attempt = 0;
        for (;;) {
            uint32_t r = lowbias32(seed ^ lowbias32((uint32_t)h * 0x9E3779B9u + (uint32_t)j * 0x85EBCA6Bu + attempt * 0xC2B2AE35u));
            int cand = (int)(r % (uint32_t)n), dup = 0;
            for (int k = 0; k < j; k++) dup |= idx[k] == cand;
            if (!dup || attempt >= 64) { idx[j] = cand; break; }
            attempt++;
        }
    }
}

static double det3(const double m[3][3])
{
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

/* det of A with column c taken from B */
static double det3_col(const double A[3][3], const double B[3][3], int c)
{
    double m[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = j == c ? B[i][j] : A[i][j];
    return det3(m);
}

static double cubic_eval(const double* c, double g) { return ((c[3] * g + c[2]) * g + c[1]) * g + c[0]; }

/* a real root of c3 g^3 + c2 g^2 + c1 g + c0 (c3 != 0): bisection on the Cauchy bound, 2 Newton steps */
static double cubic_root(const double* c)
{
    double m = fabs(c[2]);
    if (fabs(c[1]) > m) m = fabs(c[1]);
    if (fabs(c[0]) > m) m = fabs(c[0]);
    double hi = 1.0 + m / fabs(c[3]), lo = -hi;
    double flo = cubic_eval(c, lo);
    for (int it = 0; it < 80; it++) {
        const double mid = 0.5 * (lo + hi), fm = cubic_eval(c, mid);
        if ((fm < 0.0) == (flo < 0.0)) { lo = mid; flo = fm; } else hi = mid;
    }
    double g = 0.5 * (lo + hi);
    for (int it = 0; it < 2; it++) {
        const double d = (3.0 * c[3] * g + 2.0 * c[2]) * g + c[1];
        if (d != 0.0) g -= cubic_eval(c, g) / d;
    }
    return g;
}

static void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* unit null vector of the (rank-2) symmetric matrix M - s I: the longest cross product of two rows */
static int null_vec(const double M[3][3], double s, double* e)
{
    double r[3][3], c[3][3], best = -1.0;
    int bi = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i][j] = M[i][j] - (i == j ? s : 0.0);
    cross3(r[0], r[1], c[0]);
    cross3(r[0], r[2], c[1]);
    cross3(r[1], r[2], c[2]);
    for (int k = 0; k < 3; k++) {
        const double n2 = c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2];
        if (n2 > best) { best = n2; bi = k; }
    }
    if (!(best > 0.0)) return 0;
    const double inv = 1.0 / sqrt(best);
    for (int k = 0; k < 3; k++) e[k] = c[bi][k] * inv;
    return 1;
}

/* P3P: y = 3 unit bearings (rows), x = 3 points (rows) -> up to 4 (R row-major 9, t 3) */
int vo_ref_p3p(const double* y, const double* x, double* R_out, double* t_out)
{
    const double *y1 = y, *y2 = y + 3, *y3 = y + 6, *x1 = x, *x2 = x + 3, *x3 = x + 6;
    const double b12 = y1[0] * y2[0] + y1[1] * y2[1] + y1[2] * y2[2];
    const double b13 = y1[0] * y3[0] + y1[1] * y3[1] + y1[2] * y3[2];
    const double b23 = y2[0] * y3[0] + y2[1] * y3[1] + y2[2] * y3[2];
    double d12[3], d13[3], d23[3], dx[3];
    for (int k = 0; k < 3; k++) { d12[k] = x1[k] - x2[k]; d13[k] = x1[k] - x3[k]; d23[k] = x2[k] - x3[k]; }
    const double a12 = d12[0] * d12[0] + d12[1] * d12[1] + d12[2] * d12[2];
    const double a13 = d13[0] * d13[0] + d13[1] * d13[1] + d13[2] * d13[2];
    const double a23 = d23[0] * d23[0] + d23[1] * d23[1] + d23[2] * d23[2];
    cross3(d12, d13, dx);
    const double area2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
    if (!(a12 > 0.0) || !(a13 > 0.0) || !(a23 > 0.0) || !(area2 > 1e-24 * a12 * a13)) return 0;

    const double D1[3][3] = { { a23, -a23 * b12, 0.0 }, { -a23 * b12, a23 - a12, a12 * b23 }, { 0.0, a12 * b23, -a12 } };
    const double D2[3][3] = { { a23, 0.0, -a23 * b13 }, { 0.0, -a13, a13 * b23 }, { -a23 * b13, a13 * b23, a23 - a13 } };
    double c[4];
    c[0] = det3(D1);
    c[3] = det3(D2);
    c[1] = (det3_col(D1, D2, 0) + det3_col(D1, D2, 1)) + det3_col(D1, D2, 2);
    c[2] = (det3_col(D2, D1, 0) + det3_col(D2, D1, 1)) + det3_col(D2, D1, 2);
    double D0[3][3];
    if (fabs(c[3]) >= fabs(c[0])) {
        if (c[3] == 0.0) return 0;
        const double g = cubic_root(c);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D0[i][j] = D1[i][j] + g * D2[i][j];
    } else {
        const double cr[4] = { c[3], c[2], c[1], c[0] };   /* g' = 1/g : D0 = g' D1 + D2 */
        const double g = cubic_root(cr);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D0[i][j] = g * D1[i][j] + D2[i][j];
    }
    /* the two non-zero eigenvalues of the singular D0 */
    const double tr = (D0[0][0] + D0[1][1]) + D0[2][2];
    const double m2 = ((D0[0][0] * D0[1][1] - D0[0][1] * D0[0][1]) + (D0[0][0] * D0[2][2] - D0[0][2] * D0[0][2])) +
                      (D0[1][1] * D0[2][2] - D0[1][2] * D0[1][2]);
    if (!(m2 < 0.0)) return 0;                       /* same sign: the conic has no real lines */
    const double disc = tr * tr - 4.0 * m2;
    const double s1 = 0.5 * (tr + (tr >= 0.0 ? 1.0 : -1.0) * sqrt(disc));
    const double s2 = m2 / s1;
    double e1[3], e2[3];
    if (!null_vec(D0, s1, e1) || !null_vec(D0, s2, e2)) return 0;
    const double s = sqrt(-s2 / s1);

    int ns = 0;
    for (int sg = 0; sg < 2; sg++) {
        double p[3];
        for (int k = 0; k < 3; k++) p[k] = e1[k] + (sg ? -s : s) * e2[k];
        if (!(fabs(p[0]) > 1e-12)) continue;
        const double w0 = -p[1] / p[0], w1 = -p[2] / p[0];
        const double A = a23 * w1 * w1 - a12;
        const double B = (2.0 * a23 * w0 * w1 - 2.0 * a23 * b12 * w1) + 2.0 * a12 * b23;
        const double C = ((a23 * w0 * w0 - 2.0 * a23 * b12 * w0) + a23) - a12;
        double tau[2];
        int nt = 0;
        if (fabs(A) > 1e-14 * (fabs(B) + fabs(C))) {
            const double dq = B * B - 4.0 * A * C;
            if (dq >= 0.0) {
                const double sq = sqrt(dq);
                tau[0] = (-B + sq) / (2.0 * A);
                tau[1] = (-B - sq) / (2.0 * A);
                nt = 2;
            }
        } else if (B != 0.0) {
            tau[0] = -C / B;
            nt = 1;
        }
        for (int q = 0; q < nt; q++) {
            const double t = tau[q];
            if (!(t > 0.0)) continue;
            const double den = (1.0 + t * t) - 2.0 * b23 * t;
            if (!(den > 0.0)) continue;
            double l2 = sqrt(a23 / den), l3 = t * l2, l1 = (w0 + w1 * t) * l2;
            if (!(l1 > 0.0)) continue;
            for (int it = 0; it < 3; it++) {         /* Gauss-Newton on the three distance equations */
                const double r0 = ((l1 * l1 + l2 * l2) - 2.0 * b12 * l1 * l2) - a12;
                const double r1 = ((l1 * l1 + l3 * l3) - 2.0 * b13 * l1 * l3) - a13;
                const double r2 = ((l2 * l2 + l3 * l3) - 2.0 * b23 * l2 * l3) - a23;
                const double J[3][3] = { { 2.0 * l1 - 2.0 * b12 * l2, 2.0 * l2 - 2.0 * b12 * l1, 0.0 },
                                         { 2.0 * l1 - 2.0 * b13 * l3, 0.0, 2.0 * l3 - 2.0 * b13 * l1 },
                                         { 0.0, 2.0 * l2 - 2.0 * b23 * l3, 2.0 * l3 - 2.0 * b23 * l2 } };
                const double dj = det3(J);
                if (dj == 0.0) break;
                const double rr[3][3] = { { r0, 0, 0 }, { r1, 0, 0 }, { r2, 0, 0 } };
                const double n0 = det3_col(J, rr, 0);
                double Jc[3][3];
                for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Jc[i][j] = J[i][j];
                Jc[0][1] = r0; Jc[1][1] = r1; Jc[2][1] = r2;
                const double n1 = det3(Jc);
                for (int i = 0; i < 3; i++) Jc[i][1] = J[i][1];
                Jc[0][2] = r0; Jc[1][2] = r1; Jc[2][2] = r2;
                const double n2 = det3(Jc);
                l1 -= n0 / dj; l2 -= n1 / dj; l3 -= n2 / dj;
            }
            if (!(l1 > 0.0) || !(l2 > 0.0) || !(l3 > 0.0)) continue;
            /* R = Y X^-1 with X = [d12 d13 d12 x d13], Y = [l1 y1 - l2 y2, l1 y1 - l3 y3, cross] (columns) */
            double ya[3], yb[3], yc[3];
            for (int k = 0; k < 3; k++) { ya[k] = l1 * y1[k] - l2 * y2[k]; yb[k] = l1 * y1[k] - l3 * y3[k]; }
            cross3(ya, yb, yc);
            const double X[3][3] = { { d12[0], d13[0], dx[0] }, { d12[1], d13[1], dx[1] }, { d12[2], d13[2], dx[2] } };
            const double dX = det3(X);
            if (dX == 0.0) continue;
            double Xi[3][3];                          /* inverse by the adjugate */
            Xi[0][0] = (X[1][1] * X[2][2] - X[1][2] * X[2][1]) / dX;
            Xi[0][1] = (X[0][2] * X[2][1] - X[0][1] * X[2][2]) / dX;
            Xi[0][2] = (X[0][1] * X[1][2] - X[0][2] * X[1][1]) / dX;
            Xi[1][0] = (X[1][2] * X[2][0] - X[1][0] * X[2][2]) / dX;
            Xi[1][1] = (X[0][0] * X[2][2] - X[0][2] * X[2][0]) / dX;
            Xi[1][2] = (X[0][2] * X[1][0] - X[0][0] * X[1][2]) / dX;
            Xi[2][0] = (X[1][0] * X[2][1] - X[1][1] * X[2][0]) / dX;
            Xi[2][1] = (X[0][1] * X[2][0] - X[0][0] * X[2][1]) / dX;
            Xi[2][2] = (X[0][0] * X[1][1] - X[0][1] * X[1][0]) / dX;
            double* R = R_out + 9 * ns;
            double* tt = t_out + 3 * ns;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) R[i * 3 + j] = (ya[i] * Xi[0][j] + yb[i] * Xi[1][j]) + yc[i] * Xi[2][j];
            for (int i = 0; i < 3; i++) tt[i] = l1 * y1[i] - ((R[i * 3] * x1[0] + R[i * 3 + 1] * x1[1]) + R[i * 3 + 2] * x1[2]);
            ns++;
        }
    }
    return ns;
}

/* one hypothesis: Rt (3x4 row-major float64) and its float32 projection matrix P = K [R|t]; 0 if none */
int vo_ref_pnp_hypothesis(const float* X, const float* uv, const int* idx, const double* K4, double* Rt, float* P)
{
    double y[9], x[9], Rs[36], ts[12];
    for (int s = 0; s < 3; s++) {
        const int i = idx[s];
        const double a = ((double)uv[2 * i] - K4[2]) / K4[0], b = ((double)uv[2 * i + 1] - K4[3]) / K4[1];
        const double inv = 1.0 / sqrt((a * a + b * b) + 1.0);
        y[3 * s] = a * inv; y[3 * s + 1] = b * inv; y[3 * s + 2] = inv;
        for (int k = 0; k < 3; k++) x[3 * s + k] = (double)X[3 * i + k];
    }
    const int ns = vo_ref_p3p(y, x, Rs, ts);
    const int i4 = idx[3];
    const double u4 = ((double)uv[2 * i4] - K4[2]) / K4[0], v4 = ((double)uv[2 * i4 + 1] - K4[3]) / K4[1];
    const double x4[3] = { (double)X[3 * i4], (double)X[3 * i4 + 1], (double)X[3 * i4 + 2] };
    int best = -1;
    double beste = 0.0;
    for (int s = 0; s < ns; s++) {
        const double* R = Rs + 9 * s;
        const double* t = ts + 3 * s;
        const double xc = ((R[0] * x4[0] + R[1] * x4[1]) + R[2] * x4[2]) + t[0];
        const double yc = ((R[3] * x4[0] + R[4] * x4[1]) + R[5] * x4[2]) + t[1];
        const double zc = ((R[6] * x4[0] + R[7] * x4[1]) + R[8] * x4[2]) + t[2];
        if (!(zc > 0.0)) continue;
        const double du = xc / zc - u4, dv = yc / zc - v4, e = du * du + dv * dv;
        if (best < 0 || e < beste) { best = s; beste = e; }
    }
    if (best < 0) {
        for (int k = 0; k < 12; k++) { Rt[k] = 0.0; P[k] = 0.0f; }
        return 0;
    }
    const double* R = Rs + 9 * best;
    const double* t = ts + 3 * best;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rt[r * 4 + c] = R[r * 3 + c]; Rt[r * 4 + 3] = t[r]; }
    for (int c = 0; c < 4; c++) {
        P[c] = (float)(K4[0] * Rt[c] + K4[2] * Rt[8 + c]);
        P[4 + c] = (float)(K4[1] * Rt[4 + c] + K4[3] * Rt[8 + c]);
        P[8 + c] = (float)Rt[8 + c];
    }
    return 1;
}

static inline int reproj_inlier(const float* P, float X, float Y, float Z, float u, float v, float thr2)
{
    volatile float a0 = P[0] * X, a1 = P[1] * Y, a2 = P[2] * Z;
    volatile float b0 = P[4] * X, b1 = P[5] * Y, b2 = P[6] * Z;
    volatile float c0 = P[8] * X, c1 = P[9] * Y, c2 = P[10] * Z;
    volatile float xc = ((a0 + a1) + a2) + P[3];
    volatile float yc = ((b0 + b1) + b2) + P[7];
    volatile float zc = ((c0 + c1) + c2) + P[11];
    volatile float uz = u * zc, vz = v * zc;
    volatile float du = xc - uz, dv = yc - vz;
    volatile float q0 = du * du, q1 = dv * dv;
    volatile float e = q0 + q1;
    volatile float zz = zc * zc;
    volatile float lim = thr2 * zz;
    return zc > 0.0f && e < lim;
}

int vo_ref_reproj_count(const float* P, const float* X, const float* uv, int n, float thr, uint8_t* mask)
{
    volatile float thr2 = thr * thr;
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        int in = reproj_inlier(P, X[3 * i], X[3 * i + 1], X[3 * i + 2], uv[2 * i], uv[2 * i + 1], thr2);
        if (mask) mask[i] = (uint8_t)in;
        cnt += in;
    }
    return cnt;
}

int vo_ref_ransac_pnp(const float* X, const float* uv, int n, const double* K4, int iters, float thr, uint32_t seed,
                      double* Rt_best, uint8_t* mask, int32_t* counts, int* best_iter)
{
    if (n < 4 || iters <= 0) return -1;
    int best = -1, best_h = -1;
    float Pb[12];
    memset(Pb, 0, sizeof(Pb));
    for (int h = 0; h < iters; h++) {
        int idx[4];
        double Rt[12];
        float P[12];
        vo_ref_ransac_sample4(seed, h, n, idx);
        vo_ref_pnp_hypothesis(X, uv, idx, K4, Rt, P);
        int c = vo_ref_reproj_count(P, X, uv, n, thr, NULL);
        if (counts) counts[h] = c;
        if (c > best) { best = c; best_h = h; memcpy(Rt_best, Rt, sizeof(Rt)); memcpy(Pb, P, sizeof(P)); }
    }
    vo_ref_reproj_count(Pb, X, uv, n, thr, mask);
    if (best_iter) *best_iter = best_h;
    return best;
}
