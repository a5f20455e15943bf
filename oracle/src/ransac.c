/*
 * ORACLE (test infrastructure only -- see ../vo_oracle.h).
 *
 * RANSAC essential-matrix hypothesis generation + Sampson scoring (BASELINE config 5).
 * There is NO openVO counterpart (the reference has no RANSAC anywhere): this is the
 * build's own definition, restated here so that the HIP kernels can be checked bit for bit.
 * "Parity unpinned" by construction.
 *
 *   sample(h, j)  counter-based hash RNG (seed, hypothesis h, slot j, retry) -> 8 distinct indices
 *   hypothesis    8-point algorithm on K^-1-normalised points: null vector of A (8x9) as the
 *                 eigenvector of the smallest eigenvalue of A^T A (cyclic Jacobi, float64), then
 *                 projection on the essential manifold (3x3 SVD, singular values 1,1,0)
 *   score         F = K^-T E K^-1 rounded to float32; Sampson distance in float32, fixed
 *                 operation order; inlier iff d < thr^2
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

void vo_ref_ransac_sample8(uint32_t seed, int h, int n, int* idx)
{
    for (int j = 0; j < 8; j++) {
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

/* smallest-eigenvalue eigenvector of a symmetric 9x9 matrix (cyclic Jacobi) */
static void jacobi9_min(double a[9][9], double* vec)
{
    double v[9][9];
    for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 50; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < 9; p++) { diag += a[p][p] * a[p][p]; for (int q = p + 1; q < 9; q++) off += a[p][q] * a[p][q]; }
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < 8; p++)
            for (int q = p + 1; q < 9; q++) {
                const double apq = a[p][q];
                if (apq == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 9; k++) { const double x = a[k][p], y = a[k][q]; a[k][p] = c * x - s * y; a[k][q] = s * x + c * y; }
                for (int k = 0; k < 9; k++) { const double x = a[p][k], y = a[q][k]; a[p][k] = c * x - s * y; a[q][k] = s * x + c * y; }
                for (int k = 0; k < 9; k++) { const double x = v[k][p], y = v[k][q]; v[k][p] = c * x - s * y; v[k][q] = s * x + c * y; }
            }
    }
    int m = 0;
    for (int i = 1; i < 9; i++) if (a[i][i] < a[m][m]) m = i;
    for (int k = 0; k < 9; k++) vec[k] = v[k][m];
}

/* E (row-major 9) from 8 correspondences given in pixels; K = fx, fy, cx, cy */
void vo_ref_essential_8pt(const float* p1, const float* p2, const int* idx, const double* K4, double* E)
{
    double ata[9][9];
    memset(ata, 0, sizeof(ata));
    for (int s = 0; s < 8; s++) {
        const int i = idx[s];
        const double x1 = ((double)p1[2 * i] - K4[2]) / K4[0], y1 = ((double)p1[2 * i + 1] - K4[3]) / K4[1];
        const double x2 = ((double)p2[2 * i] - K4[2]) / K4[0], y2 = ((double)p2[2 * i + 1] - K4[3]) / K4[1];
        const double r[9] = { x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0 };
        for (int a = 0; a < 9; a++) for (int b = 0; b < 9; b++) ata[a][b] += r[a] * r[b];
    }
    double e0[9], U[9], w[3], Vt[9];
    jacobi9_min(ata, e0);
    vo_ref_svd3(e0, U, w, Vt);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) E[r * 3 + c] = U[r * 3 + 0] * Vt[0 * 3 + c] + U[r * 3 + 1] * Vt[1 * 3 + c];
}

/* F = K^-T E K^-1 in float32 */
void vo_ref_fundamental_f32(const double* E, const double* K4, float* F)
{
    const double ifx = 1.0 / K4[0], ify = 1.0 / K4[1], cx = K4[2], cy = K4[3];
    /* K^-1 = [ifx 0 -cx*ifx; 0 ify -cy*ify; 0 0 1] */
    double Ki[9] = { ifx, 0, -cx * ifx, 0, ify, -cy * ify, 0, 0, 1 }, T[9], Fd[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += E[r * 3 + k] * Ki[k * 3 + c]; T[r * 3 + c] = s; }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += Ki[k * 3 + r] * T[k * 3 + c]; Fd[r * 3 + c] = s; }
    /* scale so that the largest entry has magnitude 1 (keeps float32 well conditioned) */
    double mx = 0; for (int k = 0; k < 9; k++) if (fabs(Fd[k]) > mx) mx = fabs(Fd[k]);
    const double sc = mx > 0 ? 1.0 / mx : 1.0;
    for (int k = 0; k < 9; k++) F[k] = (float)(Fd[k] * sc);
}

static inline int sampson_inlier(const float* F, float u1, float v1, float u2, float v2, float thr2)
{
    volatile float a0 = F[0] * u1, a1 = F[1] * v1, b0 = F[3] * u1, b1 = F[4] * v1, c0 = F[6] * u1, c1 = F[7] * v1;
    volatile float fx0 = (a0 + a1) + F[2], fx1 = (b0 + b1) + F[5], fx2 = (c0 + c1) + F[8];
    volatile float d0 = F[0] * u2, d1 = F[3] * v2, e0 = F[1] * u2, e1 = F[4] * v2;
    volatile float ft0 = (d0 + d1) + F[6], ft1 = (e0 + e1) + F[7];
    volatile float g0 = u2 * fx0, g1 = v2 * fx1;
    volatile float num = (g0 + g1) + fx2;
    volatile float q0 = fx0 * fx0, q1 = fx1 * fx1, q2 = ft0 * ft0, q3 = ft1 * ft1;
    volatile float den = ((q0 + q1) + q2) + q3;
    volatile float nn = num * num;
    float d = nn / den;
    return d < thr2;
}

int vo_ref_sampson_count(const float* F, const float* p1, const float* p2, int n, float thr, uint8_t* mask)
{
    volatile float thr2 = thr * thr;
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        int in = sampson_inlier(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], thr2);
        if (mask) mask[i] = (uint8_t)in;
        cnt += in;
    }
    return cnt;
}

int vo_ref_ransac_essential(const float* p1, const float* p2, int n, const double* K4, int iters, float thr,
                            uint32_t seed, double* E_best, uint8_t* mask, int32_t* counts, int* best_iter)
{
    if (n < 8 || iters <= 0) return -1;
    int best = -1, best_h = -1;
    for (int h = 0; h < iters; h++) {
        int idx[8];
        double E[9];
        float F[9];
        vo_ref_ransac_sample8(seed, h, n, idx);
        vo_ref_essential_8pt(p1, p2, idx, K4, E);
        vo_ref_fundamental_f32(E, K4, F);
        int c = vo_ref_sampson_count(F, p1, p2, n, thr, NULL);
        if (counts) counts[h] = c;
        if (c > best) { best = c; best_h = h; memcpy(E_best, E, sizeof(E)); }
    }
    float F[9];
    vo_ref_fundamental_f32(E_best, K4, F);
    vo_ref_sampson_count(F, p1, p2, n, thr, mask);
    if (best_iter) *best_iter = best_h;
    return best;
}
