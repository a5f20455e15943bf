/*
 * ORACLE (test infrastructure only -- see ../vo_oracle.h).
 *
 * Five-point minimal solver for the essential matrix + the RANSAC loop around it (BASELINE config 5).
 * There is NO openVO counterpart (the reference has no RANSAC): this is the build's own definition,
 * restated here so that the HIP kernels (openvo_amd/csrc/ransac.hip) can be checked bit for bit; the mathematics
 * is pinned by known answers (tests/test_oracle_known_answers.py: exact synthetic motions, planar scenes).
 * "Parity unpinned" by construction.
 *
 * Algorithm (D. Nister, "An efficient solution to the five-point relative pose problem", PAMI 2004):
 *   1. the 5 epipolar constraints x2^T E x1 = 0 span a 5x9 system; E = x X + y Y + z Z + W over its 4-dimensional
 *      null space (Gauss-Jordan with full pivoting)
 *   2. det(E) = 0 and 2 E E^T E - tr(E E^T) E = 0 are ten cubics in (x, y, z): a 10x20 matrix over the monomials
 *      [x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1]; Gauss-Jordan on the first ten
 *   3. rows (x^2z, x^2), (y^2z, y^2), (xyz, xy) combine to B(z) [x y 1]^T = 0 with polynomial entries of degree 3, 3, 4;
 *      det B(z) is a 10th-degree polynomial whose real roots (isolated between the roots of successive derivatives on z and on 1/z, bisection)
 *      give z, then (x, y) from the cross product of two rows of B(z)
 *   4. a sixth correspondence picks among the <= 10 candidates (smallest Sampson error)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../vo_oracle.h"

static const int T11[4][4] = { { 0, 3, 4, 6 }, { 3, 1, 5, 7 }, { 4, 5, 2, 8 }, { 6, 7, 8, 9 } };
static const int T21[10][4] = { { 0, 3, 4, 10 }, { 5, 1, 6, 11 }, { 7, 8, 2, 12 }, { 3, 5, 9, 13 }, { 4, 9, 7, 14 },
                                { 9, 6, 8, 15 }, { 10, 13, 14, 16 }, { 13, 11, 15, 17 }, { 14, 15, 12, 18 }, { 16, 17, 18, 19 } };
static const int ORDER[20] = { 0, 1, 3, 5, 4, 10, 6, 11, 9, 13, 7, 14, 16, 8, 15, 17, 2, 12, 18, 19 };

/* out (10) += s * a (4: x y z 1) * b (4) */
static void mul11_acc(const double* a, const double* b, double s, double* out)
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[T11[i][j]] += s * (a[i] * b[j]);
}
/* out (20) += s * a (10) * b (4) */
static void mul21_acc(const double* a, const double* b, double s, double* out)
{
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 4; j++) out[T21[i][j]] += s * (a[i] * b[j]);
}

static double poly_eval(const double* c, int deg, double z)
{
    double v = c[deg];
    for (int k = deg - 1; k >= 0; k--) v = v * z + c[k];
    return v;
}

/* 1-D polynomial product: r (da+db+1) = a (da+1) * b (db+1) */
static void pmul(const double* a, int da, const double* b, int db, double* r)
{
    for (int k = 0; k <= da + db; k++) r[k] = 0.0;
    for (int i = 0; i <= da; i++)
        for (int j = 0; j <= db; j++) r[i + j] += a[i] * b[j];
}

/* all real roots of a degree-10 polynomial inside [-1, 1], increasing.  Roots of p lie one per interval between
 * consecutive roots of p', so the chain p^(9) (linear) -> ... -> p' -> p isolates every simple root however close two
 * of them are; each interval is bisected a fixed number of times (24 for the derivatives, 52 for p itself). */
static int roots_unit(const double* c, double* out)
{
    double d[10][11];
    for (int i = 0; i <= 10; i++) d[0][i] = c[i];
    for (int k = 1; k < 10; k++)
        for (int i = 0; i <= 10 - k; i++) d[k][i] = d[k - 1][i + 1] * (double)(i + 1);
    double brk[12], nxt[12];
    int nb = 0;                                     /* interior break points = roots of the next-higher derivative */
    for (int k = 9; k >= 0; k--) {
        const int deg = 10 - k, iters = k ? 24 : 52;
        int nn = 0;
        double a = -1.0, fa = poly_eval(d[k], deg, a);
        for (int s = 0; s <= nb; s++) {
            const double b = s < nb ? brk[s] : 1.0, fb = poly_eval(d[k], deg, b);
            if ((fa < 0.0) != (fb < 0.0) && nn < deg) {
                double lo = a, hi = b;
                const int neg = fa < 0.0;
                for (int it = 0; it < iters; it++) {
                    const double mid = 0.5 * (lo + hi), fm = poly_eval(d[k], deg, mid);
                    if ((fm < 0.0) == neg) lo = mid; else hi = mid;
                }
                nxt[nn++] = 0.5 * (lo + hi);
            }
            a = b; fa = fb;
        }
        nb = nn;
        for (int s = 0; s < nn; s++) brk[s] = nxt[s];
    }
    for (int s = 0; s < nb; s++) out[s] = brk[s];
    return nb;
}

/* x1, x2: 5 normalised points each (x, y); E_out: up to 10 matrices (row-major 9, unit Frobenius norm); returns count */
int vo_ref_poly10_roots_unit(const double* c11, double* out10) { return roots_unit(c11, out10); }

int vo_ref_essential_5pt(const double* x1, const double* x2, double* E_out)
{
    /* 1. null space of the 5x9 constraint matrix */
    double Q[5][9];
    for (int s = 0; s < 5; s++) {
        const double a = x1[2 * s], b = x1[2 * s + 1], c = x2[2 * s], d = x2[2 * s + 1];
        const double r[9] = { c * a, c * b, c, d * a, d * b, d, a, b, 1.0 };
        for (int k = 0; k < 9; k++) Q[s][k] = r[k];
    }
    int perm[9];
    for (int k = 0; k < 9; k++) perm[k] = k;
    for (int i = 0; i < 5; i++) {
        int pr = i, pc = i;
        double best = -1.0;
        for (int r = i; r < 5; r++)
            for (int c = i; c < 9; c++)
                if (fabs(Q[r][c]) > best) { best = fabs(Q[r][c]); pr = r; pc = c; }
        if (!(best > 1e-300)) return 0;
        for (int c = 0; c < 9; c++) { const double t = Q[i][c]; Q[i][c] = Q[pr][c]; Q[pr][c] = t; }
        for (int r = 0; r < 5; r++) { const double t = Q[r][i]; Q[r][i] = Q[r][pc]; Q[r][pc] = t; }
        { const int t = perm[i]; perm[i] = perm[pc]; perm[pc] = t; }
        const double inv = 1.0 / Q[i][i];
        for (int c = 0; c < 9; c++) Q[i][c] *= inv;
        for (int r = 0; r < 5; r++)
            if (r != i) {
                const double f = Q[r][i];
                for (int c = 0; c < 9; c++) Q[r][c] -= f * Q[i][c];
            }
    }
    double N[4][9];                                 /* X, Y, Z, W */
    for (int j = 0; j < 4; j++) {
        for (int k = 0; k < 9; k++) N[j][k] = 0.0;
        N[j][perm[5 + j]] = 1.0;
        for (int i = 0; i < 5; i++) N[j][perm[i]] = -Q[i][5 + j];
    }
    /* 2. the ten cubic constraints: entries of E as degree-1 polynomials (coefficients of x, y, z, 1) */
    double Ep[9][4];
    for (int k = 0; k < 9; k++)
        for (int j = 0; j < 4; j++) Ep[k][j] = N[j][k];
    double A[10][20];
    memset(A, 0, sizeof(A));
    {   /* det(E) */
        double m01[10], m02[10], m12[10];           /* 2x2 minors of rows 1, 2 */
        memset(m01, 0, sizeof(m01)); memset(m02, 0, sizeof(m02)); memset(m12, 0, sizeof(m12));
        mul11_acc(Ep[4], Ep[8], 1.0, m12); mul11_acc(Ep[5], Ep[7], -1.0, m12);   /* e11 e22 - e12 e21 */
        mul11_acc(Ep[3], Ep[8], 1.0, m02); mul11_acc(Ep[5], Ep[6], -1.0, m02);   /* e10 e22 - e12 e20 */
        mul11_acc(Ep[3], Ep[7], 1.0, m01); mul11_acc(Ep[4], Ep[6], -1.0, m01);   /* e10 e21 - e11 e20 */
        mul21_acc(m12, Ep[0], 1.0, A[0]);
        mul21_acc(m02, Ep[1], -1.0, A[0]);
        mul21_acc(m01, Ep[2], 1.0, A[0]);
    }
    {   /* (E E^T - 1/2 tr(E E^T) I) E */
        double G[3][3][10];
        memset(G, 0, sizeof(G));
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++) {
                for (int k = 0; k < 3; k++) mul11_acc(Ep[i * 3 + k], Ep[j * 3 + k], 1.0, G[i][j]);
                if (j != i) memcpy(G[j][i], G[i][j], sizeof(G[i][j]));
            }
        double tr[10];
        for (int k = 0; k < 10; k++) tr[k] = 0.5 * ((G[0][0][k] + G[1][1][k]) + G[2][2][k]);
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 10; k++) G[i][i][k] -= tr[k];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) mul21_acc(G[i][k], Ep[k * 3 + j], 1.0, A[1 + i * 3 + j]);
    }
    /* reorder the columns, Gauss-Jordan on the first ten */
    double M[10][20];
    for (int r = 0; r < 10; r++)
        for (int c = 0; c < 20; c++) M[r][c] = A[r][ORDER[c]];
    for (int c = 0; c < 10; c++) {
        int pr = c;
        double best = fabs(M[c][c]);
        for (int r = c + 1; r < 10; r++)
            if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); pr = r; }
        if (!(best > 1e-300)) return 0;
        if (pr != c)
            for (int k = 0; k < 20; k++) { const double t = M[c][k]; M[c][k] = M[pr][k]; M[pr][k] = t; }
        const double inv = 1.0 / M[c][c];
        for (int k = 0; k < 20; k++) M[c][k] *= inv;
        for (int r = 0; r < 10; r++)
            if (r != c) {
                const double f = M[r][c];
                if (f != 0.0)
                    for (int k = 0; k < 20; k++) M[r][k] -= f * M[c][k];
            }
    }
    /* 3. B(z): rows from (e,f) = (4,5), (g,h) = (6,7), (i,j) = (8,9); remaining columns 10..19 =
     *    [xz^2 xz x | yz^2 yz y | z^3 z^2 z 1] */
    double B[3][3][5];
    for (int r = 0; r < 3; r++) {
        const double* e = M[4 + 2 * r] + 10;
        const double* f = M[5 + 2 * r] + 10;
        for (int v = 0; v < 2; v++) {               /* x-part (v = 0), y-part (v = 1): cubic in z */
            const int o = 3 * v;
            B[r][v][0] = e[o + 2];
            B[r][v][1] = e[o + 1] - f[o + 2];
            B[r][v][2] = e[o] - f[o + 1];
            B[r][v][3] = -f[o];
            B[r][v][4] = 0.0;
        }
        B[r][2][0] = e[9];
        B[r][2][1] = e[8] - f[9];
        B[r][2][2] = e[7] - f[8];
        B[r][2][3] = e[6] - f[7];
        B[r][2][4] = -f[6];
    }
    double det[11], t6[7], t7[8], t10[11];
    for (int k = 0; k <= 10; k++) det[k] = 0.0;
    /* det = B00 (B11 B22 - B12 B21) - B01 (B10 B22 - B12 B20) + B02 (B10 B21 - B11 B20) */
    {
        double p[8], q[8], m[8];
        pmul(B[1][1], 3, B[2][2], 4, p); pmul(B[1][2], 4, B[2][1], 3, q);
        for (int k = 0; k <= 7; k++) m[k] = p[k] - q[k];
        pmul(B[0][0], 3, m, 7, t10);
        for (int k = 0; k <= 10; k++) det[k] += t10[k];
        pmul(B[1][0], 3, B[2][2], 4, p); pmul(B[1][2], 4, B[2][0], 3, q);
        for (int k = 0; k <= 7; k++) m[k] = p[k] - q[k];
        pmul(B[0][1], 3, m, 7, t10);
        for (int k = 0; k <= 10; k++) det[k] -= t10[k];
        pmul(B[1][0], 3, B[2][1], 3, t6); pmul(B[1][1], 3, B[2][0], 3, t7);
        double m6[7];
        for (int k = 0; k <= 6; k++) m6[k] = t6[k] - t7[k];
        pmul(B[0][2], 4, m6, 6, t10);
        for (int k = 0; k <= 10; k++) det[k] += t10[k];
    }
    double mx = 0.0;
    for (int k = 0; k <= 10; k++) if (fabs(det[k]) > mx) mx = fabs(det[k]);
    if (!(mx > 0.0)) return 0;
    double rev[11];
    for (int k = 0; k <= 10; k++) { det[k] /= mx; }
    for (int k = 0; k <= 10; k++) rev[k] = det[10 - k];
    /* real roots: |z| <= 1 on det, |z| > 1 as z = 1/w with |w| < 1 on the reversed polynomial */
    double roots[10], rt[10];
    int nr = 0;
    {
        const int n0 = roots_unit(det, rt);
        for (int k = 0; k < n0 && nr < 10; k++) roots[nr++] = rt[k];
        const int n1 = roots_unit(rev, rt);
        for (int k = 0; k < n1 && nr < 10; k++)
            if (fabs(rt[k]) > 1e-12 && fabs(rt[k]) < 1.0) roots[nr++] = 1.0 / rt[k];
    }
    /* 4. back-substitution */
    int ne = 0;
    for (int k = 0; k < nr; k++) {
        const double z = roots[k];
        double b[3][3];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) b[r][c] = poly_eval(B[r][c], 4, z);
        /* [x y 1] is orthogonal to every row: cross product of the pair of rows with the largest third component */
        double bestc[3] = { 0, 0, 0 };
        for (int p = 0; p < 3; p++) {
            const int q = (p + 1) % 3;
            const double c0 = b[p][1] * b[q][2] - b[p][2] * b[q][1];
            const double c1 = b[p][2] * b[q][0] - b[p][0] * b[q][2];
            const double c2 = b[p][0] * b[q][1] - b[p][1] * b[q][0];
            if (fabs(c2) > fabs(bestc[2])) { bestc[0] = c0; bestc[1] = c1; bestc[2] = c2; }
        }
        if (!(fabs(bestc[2]) > 1e-300)) continue;
        const double x = bestc[0] / bestc[2], y = bestc[1] / bestc[2];
        double E[9], nn = 0.0;
        for (int i = 0; i < 9; i++) {
            E[i] = ((x * N[0][i] + y * N[1][i]) + z * N[2][i]) + N[3][i];
            nn += E[i] * E[i];
        }
        if (!(nn > 0.0) || nn != nn) continue;
        const double inv = 1.0 / sqrt(nn);
        for (int i = 0; i < 9; i++) E_out[ne * 9 + i] = E[i] * inv;
        ne++;
    }
    return ne;
}

static uint32_t lowbias32(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

void vo_ref_ransac_sample6(uint32_t seed, int h, int n, int* idx)
{
    for (int j = 0; j < 6; j++) {
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

/* one hypothesis: 5 correspondences solve, the sixth picks; E = 0 when nothing real comes out */
void vo_ref_essential_5pt_hyp(const float* p1, const float* p2, const int* idx6, const double* K4, double* E)
{
    double a[12], b[12], cand[90];
    for (int s = 0; s < 6; s++) {
        const int i = idx6[s];
        a[2 * s] = ((double)p1[2 * i] - K4[2]) / K4[0]; a[2 * s + 1] = ((double)p1[2 * i + 1] - K4[3]) / K4[1];
        b[2 * s] = ((double)p2[2 * i] - K4[2]) / K4[0]; b[2 * s + 1] = ((double)p2[2 * i + 1] - K4[3]) / K4[1];
    }
    const int ne = vo_ref_essential_5pt(a, b, cand);
    for (int k = 0; k < 9; k++) E[k] = 0.0;
    double best = 1e300;
    const double u1 = a[10], v1 = a[11], u2 = b[10], v2 = b[11];
    for (int c = 0; c < ne; c++) {
        const double* F = cand + 9 * c;
        const double fx0 = (F[0] * u1 + F[1] * v1) + F[2], fx1 = (F[3] * u1 + F[4] * v1) + F[5], fx2 = (F[6] * u1 + F[7] * v1) + F[8];
        const double ft0 = (F[0] * u2 + F[3] * v2) + F[6], ft1 = (F[1] * u2 + F[4] * v2) + F[7];
        const double num = (u2 * fx0 + v2 * fx1) + fx2;
        const double den = ((fx0 * fx0 + fx1 * fx1) + ft0 * ft0) + ft1 * ft1;
        const double d = (num * num) / den;
        if (d < best) { best = d; for (int k = 0; k < 9; k++) E[k] = F[k]; }
    }
}

int vo_ref_ransac_essential5(const float* p1, const float* p2, int n, const double* K4, int iters, float thr,
                             uint32_t seed, double* E_best, uint8_t* mask, int32_t* counts, int* best_iter)
{
    if (n < 6 || iters <= 0) return -1;
    int best = -1, best_h = -1;
    for (int h = 0; h < iters; h++) {
        int idx[6];
        double E[9];
        float F[9];
        vo_ref_ransac_sample6(seed, h, n, idx);
        vo_ref_essential_5pt_hyp(p1, p2, idx, K4, E);
        vo_ref_fundamental_f32(E, K4, F);
        int c = 0;
        {
            int any = 0;
            for (int k = 0; k < 9; k++) any |= F[k] != 0.0f;
            if (any) c = vo_ref_sampson_count(F, p1, p2, n, thr, NULL);
        }
        if (counts) counts[h] = c;
        if (c > best) { best = c; best_h = h; memcpy(E_best, E, sizeof(E)); }
    }
    float F[9];
    vo_ref_fundamental_f32(E_best, K4, F);
    vo_ref_sampson_count(F, p1, p2, n, thr, mask);
    if (best_iter) *best_iter = best_h;
    return best;
}
