/*
 * ORACLE (test infrastructure only -- see ../vo_oracle.h).
 *
 * CPU restatement of cv::StereoSGBM::compute for 8-bit single-channel input,
 * modes MODE_SGBM (5 paths; what the reference uses, stereo_camera.py:23-27 leaves
 * `mode` at its default) and MODE_HH (8 paths).
 * Follows OpenCV 4.x modules/calib3d/src/stereosgbm.cpp: calcPixelCostBT,
 * computeDisparitySGBM, StereoSGBMImpl::compute; modules/imgproc/src/median_blur
 * (3x3 sorting network, replicate border) and calib3d filterSpeckles.
 * Parity unpinned: the reference holds no fixture for this stage.
 */
#include <stdlib.h>
#include <string.h>
#include "../vo_oracle.h"

#define DISP_SHIFT 4
#define DISP_SCALE 16
#define MAX_COST 32767

typedef int16_t cost_t;

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int16_t sat16(int v) { return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }

typedef struct {
    int minD, maxD, D, ur, d12, P1, P2, SW2, SH2, ftzero, minX1, maxX1, width1, invalid16;
} sgbm_eff;

static void effective(const vo_ref_sgbm_params* p, int w, sgbm_eff* e)
{
    e->minD = p->minDisparity;
    e->D = p->numDisparities;
    e->maxD = e->minD + e->D;
    e->ur = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    e->d12 = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    e->P1 = p->P1 > 0 ? p->P1 : 2;
    e->P2 = imax(p->P2 > 0 ? p->P2 : 5, e->P1 + 1);
    int bs = p->blockSize > 0 ? p->blockSize : 5;
    e->SW2 = e->SH2 = bs / 2;
    e->ftzero = imax(p->preFilterCap, 15) | 1;
    e->minX1 = imax(e->maxD, 0);
    e->maxX1 = w + imin(e->minD, 0);
    e->width1 = e->maxX1 - e->minX1;
    e->invalid16 = (e->minD - 1) * DISP_SCALE;
}

/* calcPixelCostBT for one row y: cost[(x-minX1)*D + (d-minD)], two pseudo-channels
 * (x-Sobel prefiltered and raw intensity), Birchfield-Tomasi both ways. */
static void pixel_cost_row(const uint8_t* L, const uint8_t* R, int w, int h, int y,
                           const sgbm_eff* e, uint8_t* pl, uint8_t* pr, uint8_t* v0b, uint8_t* v1b,
                           cost_t* cost)
{
    const int D = e->D, ft = e->ftzero;
    const uint8_t* l0 = L + (size_t)y * w;
    const uint8_t* r0 = R + (size_t)y * w;
    const uint8_t* ln = y > 0 ? l0 - w : l0;
    const uint8_t* ls = y < h - 1 ? l0 + w : l0;
    const uint8_t* rn = y > 0 ? r0 - w : r0;
    const uint8_t* rs = y < h - 1 ? r0 + w : r0;
    /* pl/pr: [channel][x]; both channels take tab[0] = ftzero at x = 0 and x = w-1 */
    for (int c = 0; c < 2; c++) {
        pl[c * w] = pl[c * w + w - 1] = (uint8_t)ft;
        pr[c * w] = pr[c * w + w - 1] = (uint8_t)ft;
    }
    for (int x = 1; x < w - 1; x++) {
        int gl = (l0[x + 1] - l0[x - 1]) * 2 + ln[x + 1] - ln[x - 1] + ls[x + 1] - ls[x - 1];
        int gr = (r0[x + 1] - r0[x - 1]) * 2 + rn[x + 1] - rn[x - 1] + rs[x + 1] - rs[x - 1];
        pl[x] = (uint8_t)(imin(imax(gl, -ft), ft) + ft);
        pr[x] = (uint8_t)(imin(imax(gr, -ft), ft) + ft);
        pl[w + x] = l0[x];
        pr[w + x] = r0[x];
    }
    memset(cost, 0, (size_t)e->width1 * D * sizeof(cost_t));
    for (int c = 0; c < 2; c++) {
        const int diff_scale = c == 0 ? 0 : 2;
        const uint8_t* u_ = pl + c * w;
        const uint8_t* v_ = pr + c * w;
        for (int x = 0; x < w; x++) {
            int v = v_[x];
            int vl = x > 0 ? (v + v_[x - 1]) / 2 : v;
            int vr = x < w - 1 ? (v + v_[x + 1]) / 2 : v;
            v0b[x] = (uint8_t)imin(imin(vl, vr), v);
            v1b[x] = (uint8_t)imax(imax(vl, vr), v);
        }
        for (int x = e->minX1; x < e->maxX1; x++) {
            int u = u_[x];
            int ul = x > 0 ? (u + u_[x - 1]) / 2 : u;
            int ur = x < w - 1 ? (u + u_[x + 1]) / 2 : u;
            int u0 = imin(imin(ul, ur), u), u1 = imax(imax(ul, ur), u);
            cost_t* cp = cost + (size_t)(x - e->minX1) * D;
            for (int d = e->minD; d < e->maxD; d++) {
                int v = v_[x - d], v0 = v0b[x - d], v1 = v1b[x - d];
                int c0 = imax(imax(0, u - v1), v0 - u);
                int c1 = imax(imax(0, v - u1), u0 - v);
                cp[d - e->minD] = (cost_t)(cp[d - e->minD] + (imin(c0, c1) >> diff_scale));
            }
        }
    }
}

/* C[y][x][d] = P2 + box-sum over (2*SW2+1)x(2*SH2+1) of the pixel cost, indices clamped to
 * [0,width1-1] x [0,h-1] (OpenCV's sliding hsumAdd/hsumSub with its edge scales). */
static cost_t* build_cost_volume(const uint8_t* L, const uint8_t* R, int w, int h, const sgbm_eff* e)
{
    const int D = e->D, W1 = e->width1;
    const size_t row = (size_t)W1 * D;
    cost_t* C = (cost_t*)malloc(row * h * sizeof(cost_t));
    cost_t* Hs = (cost_t*)malloc(row * h * sizeof(cost_t));
    cost_t* pix = (cost_t*)malloc(row * sizeof(cost_t));
    uint8_t* pl = (uint8_t*)malloc(2 * w);
    uint8_t* pr = (uint8_t*)malloc(2 * w);
    uint8_t* v0b = (uint8_t*)malloc(w);
    uint8_t* v1b = (uint8_t*)malloc(w);
    for (int k = 0; k < h; k++) {
        pixel_cost_row(L, R, w, h, k, e, pl, pr, v0b, v1b, pix);
        cost_t* hs = Hs + row * k;
        for (int d = 0; d < D; d++) {
            int s = pix[d] * (e->SW2 + 1);
            for (int i = 1; i <= e->SW2; i++) s += pix[(size_t)imin(i, W1 - 1) * D + d];
            hs[d] = (cost_t)s;
        }
        for (int x = 1; x < W1; x++) {
            const cost_t* add = pix + (size_t)imin(x + e->SW2, W1 - 1) * D;
            const cost_t* sub = pix + (size_t)imax(x - e->SW2 - 1, 0) * D;
            for (int d = 0; d < D; d++)
                hs[(size_t)x * D + d] = (cost_t)(hs[(size_t)(x - 1) * D + d] + add[d] - sub[d]);
        }
    }
    for (size_t i = 0; i < row; i++) {
        int s = e->P2 + Hs[i] * (e->SH2 + 1);
        for (int k = 1; k <= e->SH2; k++) s += Hs[row * imin(k, h - 1) + i];
        C[i] = (cost_t)s;
    }
    for (int y = 1; y < h; y++) {
        const cost_t* add = Hs + row * imin(y + e->SH2, h - 1);
        const cost_t* sub = Hs + row * imax(y - e->SH2 - 1, 0);
        const cost_t* prev = C + row * (y - 1);
        cost_t* cur = C + row * y;
        for (size_t i = 0; i < row; i++) cur[i] = (cost_t)(prev[i] + add[i] - sub[i]);
    }
    free(Hs); free(pix); free(pl); free(pr); free(v0b); free(v1b);
    return C;
}

int vo_ref_sgbm_cost_volume(const uint8_t* L, const uint8_t* R, int w, int h,
                            const vo_ref_sgbm_params* p, int16_t* Cout)
{
    sgbm_eff e;
    effective(p, w, &e);
    if (e.width1 <= 0) return -1;
    cost_t* C = build_cost_volume(L, R, w, h, &e);
    memcpy(Cout, C, (size_t)e.width1 * e.D * h * sizeof(cost_t));
    free(C);
    return 0;
}

/* one Lr update: L(d) = C(d) + min(Lp(d), Lp(d-1)+P1, Lp(d+1)+P1, delta) - delta, where
 * delta = minLp + P2 and C already carries +P2.  Lp has sentinels MAX_COST at d=-1 and d=D. */
static inline int lr_update(const cost_t* Cp, const cost_t* Lp, int delta, int P1, int D,
                            cost_t* Lout, int* Sacc)
{
    int minL = MAX_COST;
    for (int d = 0; d < D; d++) {
        int L = Cp[d] + imin((int)Lp[d], imin(Lp[d - 1] + P1, imin(Lp[d + 1] + P1, delta))) - delta;
        Lout[d] = (cost_t)L;
        minL = imin(minL, L);
        Sacc[d] += L;
    }
    return minL;
}

static void wta_row(const cost_t* Srow, const sgbm_eff* e, int w, int16_t* disp1, int16_t* disp2,
                    cost_t* disp2cost, const short* best_in, const cost_t* minS_in)
{
    /* Srow final; best_in/minS_in already hold first-minimum d and its cost per x */
    const int D = e->D, W1 = e->width1;
    for (int x = 0; x < w; x++) {
        disp1[x] = disp2[x] = (int16_t)e->invalid16;
        disp2cost[x] = MAX_COST;
    }
    for (int x = W1 - 1; x >= 0; x--) {
        const cost_t* Sp = Srow + (size_t)x * D;
        int minS = minS_in[x], bestDisp = best_in[x], d;
        for (d = 0; d < D; d++)
            if (Sp[d] * (100 - e->ur) < minS * 100 && abs(bestDisp - d) > 1) break;
        if (d < D) continue;
        d = bestDisp;
        int x2 = x + e->minX1 - d - e->minD;
        if (disp2cost[x2] > minS) {
            disp2cost[x2] = (cost_t)minS;
            disp2[x2] = (int16_t)(d + e->minD);
        }
        if (0 < d && d < D - 1) {
            int denom2 = imax(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
            d = d * DISP_SCALE + ((Sp[d - 1] - Sp[d + 1]) * DISP_SCALE + denom2) / (denom2 * 2);
        } else
            d *= DISP_SCALE;
        disp1[x + e->minX1] = (int16_t)(d + e->minD * DISP_SCALE);
    }
    for (int x = e->minX1; x < e->maxX1; x++) {
        int d1 = disp1[x];
        if (d1 == e->invalid16) continue;
        int _d = d1 >> DISP_SHIFT, d_ = (d1 + DISP_SCALE - 1) >> DISP_SHIFT;
        int _x = x - _d, x_ = x - d_;
        if (0 <= _x && _x < w && disp2[_x] >= e->minD && abs(disp2[_x] - _d) > e->d12 &&
            0 <= x_ && x_ < w && disp2[x_] >= e->minD && abs(disp2[x_] - d_) > e->d12)
            disp1[x] = (int16_t)e->invalid16;
    }
}

static void sgbm_raw(const uint8_t* Limg, const uint8_t* Rimg, int w, int h,
                     const vo_ref_sgbm_params* p, int16_t* disp)
{
    sgbm_eff e;
    effective(p, w, &e);
    if (e.width1 <= 0) {
        for (size_t i = 0; i < (size_t)w * h; i++) disp[i] = (int16_t)e.invalid16;
        return;
    }
    const int D = e.D, W1 = e.width1, D2 = D + 2;
    const int npasses = p->mode == 1 ? 2 : 1;
    const size_t row = (size_t)W1 * D;
    cost_t* C = build_cost_volume(Limg, Rimg, w, h, &e);
    /* S: one row (MODE_SGBM) or the full volume (MODE_HH, accumulated over both passes) */
    cost_t* Sfull = (cost_t*)calloc(npasses == 2 ? row * h : row, sizeof(cost_t));
    int* Sacc = (int*)malloc(D * sizeof(int));
    /* Lr[rowbuf][dir][x+1][d+1] with zeroed x-borders; minLr[rowbuf][dir][x+1] */
    const size_t lrsz = (size_t)4 * (W1 + 2) * D2;
    cost_t* Lr[2];
    cost_t* minLr[2];
    for (int i = 0; i < 2; i++) {
        Lr[i] = (cost_t*)malloc(lrsz * sizeof(cost_t));
        minLr[i] = (cost_t*)malloc((size_t)4 * (W1 + 2) * sizeof(cost_t));
    }
#define LR(b, dir, x) (Lr[b] + ((size_t)(dir) * (W1 + 2) + (x) + 1) * D2 + 1)
#define MINLR(b, dir, x) (minLr[b][(size_t)(dir) * (W1 + 2) + (x) + 1])
    int16_t* disp2 = (int16_t*)malloc(w * sizeof(int16_t));
    cost_t* disp2cost = (cost_t*)malloc(w * sizeof(cost_t));
    short* best = (short*)malloc(W1 * sizeof(short));
    cost_t* minSx = (cost_t*)malloc(W1 * sizeof(cost_t));

    for (int pass = 1; pass <= npasses; pass++) {
        int y1, y2, dy, x1, x2, dx;
        if (pass == 1) { y1 = 0; y2 = h; dy = 1; x1 = 0; x2 = W1; dx = 1; }
        else { y1 = h - 1; y2 = -1; dy = -1; x1 = W1 - 1; x2 = -1; dx = -1; }
        int lrID = 0;
        for (int i = 0; i < 2; i++) {
            memset(Lr[i], 0, lrsz * sizeof(cost_t));
            memset(minLr[i], 0, (size_t)4 * (W1 + 2) * sizeof(cost_t));
        }
        for (int y = y1; y != y2; y += dy) {
            const cost_t* Crow = C + row * y;
            cost_t* S = npasses == 2 ? Sfull + row * y : Sfull;
            if (pass == 1) memset(S, 0, row * sizeof(cost_t));
            /* directions: 0 (x-dx, y) 1 (x-1, y-dy) 2 (x, y-dy) 3 (x+1, y-dy) */
            for (int x = x1; x != x2; x += dx) {
                cost_t* Lp[4] = { LR(lrID, 0, x - dx), LR(1 - lrID, 1, x - 1), LR(1 - lrID, 2, x),
                                  LR(1 - lrID, 3, x + 1) };
                int delta[4] = { e.P2 + MINLR(lrID, 0, x - dx), e.P2 + MINLR(1 - lrID, 1, x - 1),
                                 e.P2 + MINLR(1 - lrID, 2, x), e.P2 + MINLR(1 - lrID, 3, x + 1) };
                const cost_t* Cp = Crow + (size_t)x * D;
                cost_t* Sp = S + (size_t)x * D;
                for (int d = 0; d < D; d++) Sacc[d] = Sp[d];
                for (int k = 0; k < 4; k++) {
                    Lp[k][-1] = Lp[k][D] = MAX_COST;
                    MINLR(lrID, k, x) = (cost_t)lr_update(Cp, Lp[k], delta[k], e.P1, D, LR(lrID, k, x), Sacc);
                }
                for (int d = 0; d < D; d++) Sp[d] = sat16(Sacc[d]);
            }
            if (pass == npasses) {
                /* MODE_SGBM: fifth path (x+1, y) swept right-to-left, fused with the WTA */
                for (int x = W1 - 1; x >= 0; x--) {
                    cost_t* Sp = S + (size_t)x * D;
                    int minS = MAX_COST, bestDisp = -1;
                    if (npasses == 1) {
                        cost_t* Lp0 = LR(lrID, 0, x + 1);
                        Lp0[-1] = Lp0[D] = MAX_COST;
                        cost_t* Lo = LR(lrID, 0, x);
                        const cost_t* Cp = Crow + (size_t)x * D;
                        int delta0 = e.P2 + MINLR(lrID, 0, x + 1), minL0 = MAX_COST;
                        for (int d = 0; d < D; d++) {
                            int L0 = Cp[d] + imin((int)Lp0[d], imin(Lp0[d - 1] + e.P1, imin(Lp0[d + 1] + e.P1, delta0))) - delta0;
                            Lo[d] = (cost_t)L0;
                            minL0 = imin(minL0, L0);
                            int Sval = Sp[d] = sat16(Sp[d] + L0);
                            if (Sval < minS) { minS = Sval; bestDisp = d; }
                        }
                        MINLR(lrID, 0, x) = (cost_t)minL0;
                    } else {
                        for (int d = 0; d < D; d++)
                            if (Sp[d] < minS) { minS = Sp[d]; bestDisp = d; }
                    }
                    best[x] = (short)bestDisp;
                    minSx[x] = (cost_t)minS;
                }
                wta_row(S, &e, w, disp + (size_t)y * w, disp2, disp2cost, best, minSx);
            }
            lrID = 1 - lrID;
        }
    }
    free(C); free(Sfull); free(Sacc);
    for (int i = 0; i < 2; i++) { free(Lr[i]); free(minLr[i]); }
    free(disp2); free(disp2cost); free(best); free(minSx);
#undef LR
#undef MINLR
}

/* medianBlur ksize 3 on CV_16S: sorting network, replicate border */
static inline void srt(int16_t* a, int16_t* b) { if (*a > *b) { int16_t t = *a; *a = *b; *b = t; } }
void vo_ref_median3x3_s16(const int16_t* src, int w, int h, int16_t* dst)
{
    for (int y = 0; y < h; y++) {
        const int16_t* r0 = src + (size_t)imax(y - 1, 0) * w;
        const int16_t* r1 = src + (size_t)y * w;
        const int16_t* r2 = src + (size_t)imin(y + 1, h - 1) * w;
        for (int x = 0; x < w; x++) {
            int xl = imax(x - 1, 0), xr = imin(x + 1, w - 1);
            int16_t p0 = r0[xl], p1 = r0[x], p2 = r0[xr], p3 = r1[xl], p4 = r1[x], p5 = r1[xr],
                    p6 = r2[xl], p7 = r2[x], p8 = r2[xr];
            srt(&p1, &p2); srt(&p4, &p5); srt(&p7, &p8); srt(&p0, &p1);
            srt(&p3, &p4); srt(&p6, &p7); srt(&p1, &p2); srt(&p4, &p5);
            srt(&p7, &p8); srt(&p0, &p3); srt(&p5, &p8); srt(&p4, &p7);
            srt(&p3, &p6); srt(&p1, &p4); srt(&p2, &p5); srt(&p4, &p7);
            srt(&p4, &p2); srt(&p6, &p4); srt(&p4, &p2);
            dst[(size_t)y * w + x] = p4;
        }
    }
}

/* filterSpeckles: 4-connected flood fill, neighbours joined when both != newVal and
 * |a-b| <= maxDiff; regions of <= maxSpeckleSize pixels are set to newVal. */
void vo_ref_filter_speckles(int16_t* img, int w, int h, int newVal, int maxSpeckleSize, int maxDiff)
{
    const size_t n = (size_t)w * h;
    int* labels = (int*)calloc(n, sizeof(int));
    int* stack = (int*)malloc(n * sizeof(int));
    uint8_t* rtype = (uint8_t*)calloc(n + 1, 1);
    int curlabel = 0;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            size_t idx = (size_t)i * w + j;
            if (img[idx] == newVal) continue;
            if (labels[idx]) {
                if (rtype[labels[idx]]) img[idx] = (int16_t)newVal;
                continue;
            }
            int sp = 0, count = 0;
            curlabel++;
            labels[idx] = curlabel;
            stack[sp++] = (int)idx;
            while (sp > 0) {
                int p = stack[--sp];
                int py = p / w, px = p % w;
                int dp = img[p];
                count++;
                if (py < h - 1 && !labels[p + w] && img[p + w] != newVal && abs(dp - img[p + w]) <= maxDiff) { labels[p + w] = curlabel; stack[sp++] = p + w; }
                if (py > 0 && !labels[p - w] && img[p - w] != newVal && abs(dp - img[p - w]) <= maxDiff) { labels[p - w] = curlabel; stack[sp++] = p - w; }
                if (px < w - 1 && !labels[p + 1] && img[p + 1] != newVal && abs(dp - img[p + 1]) <= maxDiff) { labels[p + 1] = curlabel; stack[sp++] = p + 1; }
                if (px > 0 && !labels[p - 1] && img[p - 1] != newVal && abs(dp - img[p - 1]) <= maxDiff) { labels[p - 1] = curlabel; stack[sp++] = p - 1; }
            }
            if (count <= maxSpeckleSize) {
                rtype[curlabel] = 1;
                img[idx] = (int16_t)newVal;
            } else
                rtype[curlabel] = 0;
        }
    free(labels); free(stack); free(rtype);
}

int vo_ref_sgbm_compute(const uint8_t* L, const uint8_t* R, int w, int h,
                        const vo_ref_sgbm_params* p, int16_t* disp_raw, int16_t* disp_median,
                        int16_t* disp_final)
{
    if (p->numDisparities <= 0 || p->numDisparities % 16 != 0) return -1;
    const size_t n = (size_t)w * h;
    int16_t* raw = (int16_t*)malloc(n * sizeof(int16_t));
    sgbm_raw(L, R, w, h, p, raw);
    if (disp_raw) memcpy(disp_raw, raw, n * sizeof(int16_t));
    vo_ref_median3x3_s16(raw, w, h, disp_final);
    if (disp_median) memcpy(disp_median, disp_final, n * sizeof(int16_t));
    if (p->speckleWindowSize > 0)
        vo_ref_filter_speckles(disp_final, w, h, (p->minDisparity - 1) * DISP_SCALE,
                               p->speckleWindowSize, DISP_SCALE * p->speckleRange);
    free(raw);
    return 0;
}
