// ORB on gfx950: replaces cv2.ORB_create(nfeatures).detectAndCompute(img, mask)
// [reference stereo_odometer.py:22,117] with the feature_mask of [stereo_odometer.py:38-41]
// fused into the mask read.  Arithmetic definition: OpenCV 4.x features2d/src/orb.cpp,
// fast.cpp, fast_score.cpp, keypoint.cpp with ORB_create's defaults (1.2f, 8 levels,
// edgeThreshold 31, HARRIS_SCORE, patchSize 31, fastThreshold 20).  Integer stages are exact;
// float stages (Harris response, IC angle, rotation) follow OpenCV's operation order and are
// compiled without FMA contraction.  Keypoints are emitted in canonical (octave, y, x) order.
//
// Pyramid layout in HBM: one byte buffer per kind (image, blurred, mask, FAST score), level l
// stored tightly (stride = level width) at lv[l].off.
#include <math.h>
#include <algorithm>
#include "vo_internal.h"

#define NL VO_ORB_LEVELS
#define EDGE VO_ORB_EDGE
#define HALF_PATCH VO_ORB_HALF_PATCH
#define FAST_T 20
#ifndef ORB_FAST_GRID_CAP
#define ORB_FAST_GRID_CAP 2048   // workgroups of k_orb_fast_nms (they walk the tiles of all levels)
#endif
#ifndef ORB_CONE_W
#define ORB_CONE_W 64            // level-0 footprint of one pyramid cone (k_orb_pyramid)
#define ORB_CONE_H 48
#endif
#ifndef ORB_FAST_TH
#define ORB_FAST_TH 32
#endif
// rows per tile of k_orb_fast_nms (a multiple of 4: each of the block's four waves judges TH / 4 rows)
static_assert(ORB_FAST_TH % 4 == 0 && ORB_FAST_TH >= 4 && (64 + 2) * (ORB_FAST_TH + 2) < 65536, "k_orb_fast_nms: tile height");

struct LevelDev {
    int w, h;
    unsigned off;       // byte offset in pyramid buffers
    float scale;
    int quota;
    int xt, yt;         // offsets into the resize tables (entries) for x and y
    int min_x, max_x, min_y, max_y;
    int cand_off;       // offset into candidate arrays
};
struct LevelsDev { LevelDev l[NL]; int umax[HALF_PATCH + 2]; };

__constant__ int8_t c_pattern[1024] = {
#include "../../include/vo_orb_pattern.inc"
};

// counters layout (int32)
#define CNT_CAND 0
#define CNT_A 8
#define CNT_FIN 16
#define CNT_TOTAL 24
#define CNT_HIST 32

// ---------------------------------------------------------------------------------------
// host-side tables
// ---------------------------------------------------------------------------------------
static int cv_round_f(float v) { return (int)nearbyintf(v); }

static void make_lin_coeffs(int srcsize, int dstsize, int32_t* ofs, uint16_t* coef, int* mn, int* mx)
{
    const double inv_scale = (double)dstsize / srcsize;
    const double scale = 1.0 / inv_scale;
    *mn = 0; *mx = dstsize;
    for (int v = 0; v < dstsize; v++) {
        volatile double t = scale * ((double)v + 0.5);
        double fval = t - 0.5;
        int ival = (int)floor(fval);
        ofs[v] = 0; coef[2 * v] = 256; coef[2 * v + 1] = 0;
        if (ival >= 0 && srcsize > 1) {
            if (ival < srcsize - 1) {
                ofs[v] = ival;
                int c1 = (int)nearbyint((fval - (double)ival) * 256.0);
                coef[2 * v + 1] = (uint16_t)c1;
                coef[2 * v] = (uint16_t)(256 - c1);
            } else {
                ofs[v] = srcsize - 1;
                if (v < *mx) *mx = v;
            }
        } else if (v + 1 > *mn)
            *mn = v + 1;
    }
}


int orb_prepare_tables(vo_ctx* ctx, int w, int h)
{
    if (ctx->orb_w == w && ctx->orb_h == h) return VO_OK;
    LevelsDev L;
    memset(&L, 0, sizeof(L));
    std::vector<int32_t> ofs;
    std::vector<uint16_t> coef;
    size_t off = 0;
    int cand_off = 0;
    for (int l = 0; l < NL; l++) {
        float scale = (float)pow((double)1.2f, (double)l);
        float inv = 1.0f / scale;
        LevelDev& d = L.l[l];
        d.w = cv_round_f((float)w * inv);
        d.h = cv_round_f((float)h * inv);
        d.scale = scale;
        d.off = (unsigned)off;
        off += ((size_t)d.w * d.h + 255) & ~(size_t)255;
        d.cand_off = cand_off;
        cand_off += ((d.w + 1) / 2) * ((d.h + 1) / 2);
        ctx->lv[l].w = d.w; ctx->lv[l].h = d.h; ctx->lv[l].off = d.off; ctx->lv[l].scale = scale;
        if (l > 0) {
            const LevelDev& p = L.l[l - 1];
            d.xt = (int)ofs.size();
            ofs.resize(ofs.size() + d.w); coef.resize(coef.size() + 2 * (size_t)d.w);
            make_lin_coeffs(p.w, d.w, ofs.data() + d.xt, coef.data() + 2 * (size_t)d.xt, &d.min_x, &d.max_x);
            d.yt = (int)ofs.size();
            ofs.resize(ofs.size() + d.h); coef.resize(coef.size() + 2 * (size_t)d.h);
            make_lin_coeffs(p.h, d.h, ofs.data() + d.yt, coef.data() + 2 * (size_t)d.yt, &d.min_y, &d.max_y);
        }
    }
    if (off > ctx->pyr_bytes) return vo_fail(ctx, VO_E_CAP, "pyramid of %dx%d exceeds the context capacity", w, h);
    // k_orb_pyramid's cones: level l is cut into nbx x nby parts of equal share (part i owns columns [i w_l / nbx, (i+1) w_l / nbx));
    // a cone also computes, per level, the margin the next level's interval reads (bilinear taps ofs[v], ofs[v] + 1), top-down
    std::vector<int32_t> rects;
    int pyr_nbx, pyr_nby, pyr_buf[2] = { 0, 0 }, pyr_tab = 0;
    {
        const LevelDev& top = L.l[NL - 1];
        pyr_nbx = std::max(1, std::min({ div_up(w, ORB_CONE_W), top.w / 8, 48 }));
        pyr_nby = std::max(1, std::min({ div_up(h, ORB_CONE_H), top.h / 8, 48 }));
        auto intervals = [&](int nb, bool xaxis, std::vector<int32_t>& out) {     // out[l][b][2]
            out.assign((size_t)NL * nb * 2, 0);
            for (int b = 0; b < nb; b++) {
                int nlo = 0, nhi = -1;                               // interval needed at the level above (empty at the top)
                for (int l = NL - 1; l >= 0; l--) {
                    const LevelDev& d = L.l[l];
                    const int size = xaxis ? d.w : d.h;
                    int lo = (int)((int64_t)b * size / nb), hi = (int)((int64_t)(b + 1) * size / nb) - 1;   // own part
                    if (l < NL - 1 && nlo <= nhi) {
                        const LevelDev& u = L.l[l + 1];
                        const int32_t* o = ofs.data() + (xaxis ? u.xt : u.yt);
                        const int mn = xaxis ? u.min_x : u.min_y, mx = xaxis ? u.max_x : u.max_y;
                        const int slo = nlo < mn ? 0 : (nlo >= mx ? size - 1 : o[nlo]);
                        const int shi = nhi < mn ? 0 : (nhi >= mx ? size - 1 : std::min(o[nhi] + 1, size - 1));
                        if (lo > hi) { lo = slo; hi = shi; }
                        else { lo = std::min(lo, slo); hi = std::max(hi, shi); }
                    }
                    out[((size_t)l * nb + b) * 2] = lo; out[((size_t)l * nb + b) * 2 + 1] = hi;
                    nlo = lo; nhi = hi;
                }
            }
        };
        std::vector<int32_t> rx, ry;
        intervals(pyr_nbx, true, rx);
        intervals(pyr_nby, false, ry);
        for (int l = 0; l < NL; l++) {
            int mw = 0, mh = 0;
            for (int b = 0; b < pyr_nbx; b++) mw = std::max(mw, rx[((size_t)l * pyr_nbx + b) * 2 + 1] - rx[((size_t)l * pyr_nbx + b) * 2] + 1);
            for (int b = 0; b < pyr_nby; b++) mh = std::max(mh, ry[((size_t)l * pyr_nby + b) * 2 + 1] - ry[((size_t)l * pyr_nby + b) * 2] + 1);
            pyr_buf[l & 1] = std::max(pyr_buf[l & 1], ((mw * mh + 15) & ~15));
            if (l > 0) pyr_tab = std::max(pyr_tab, std::max(mw, mh));
        }
        rects = rx;
        rects.insert(rects.end(), ry.begin(), ry.end());
        if (rects.size() > 4096 / 4 * 4) return vo_fail(ctx, VO_E_CAP, "pyramid cone table exceeds its capacity");
    }
    if ((size_t)cand_off > (size_t)ctx->cand_cap * 4) return vo_fail(ctx, VO_E_CAP, "candidate capacity exceeded");
    if (ofs.size() > (size_t)(ctx->max_w + ctx->max_h) * 2 * NL) return vo_fail(ctx, VO_E_CAP, "resize tables exceed capacity");
    // umax: row half-widths of the circular patch (orb.cpp computeKeyPoints)
    {
        int vmax = (int)floor(HALF_PATCH * sqrt(2.f) / 2 + 1);
        int vmin = (int)ceil(HALF_PATCH * sqrt(2.f) / 2);
        for (int v = 0; v <= vmax; v++) L.umax[v] = (int)nearbyint(sqrt((double)HALF_PATCH * HALF_PATCH - v * v));
        for (int v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
            while (L.umax[v0] == L.umax[v0 + 1]) ++v0;
            L.umax[v] = v0;
            ++v0;
        }
    }
    // The resize tables are one device array shared by the main and all look-ahead streams: extractions still
    // in flight anywhere read them, so everything queued so far must finish before they are rewritten for a
    // new image size (rare: once per size).  What those extractions produced stays valid.
    for (int k = 0; k < vo_ctx::MAX_ENGINES; k++)
        if (ctx->la_stream[k]) VO_HIP(ctx, hipStreamSynchronize(ctx->la_stream[k]));
    for (int k = 0; k < vo_ctx::N_POSE_ALT; k++)
        if (ctx->pose_alt[k].stream) VO_HIP(ctx, hipStreamSynchronize(ctx->pose_alt[k].stream));
    for (int k = 0; k < vo_ctx::N_MONO_ALT; k++)
        if (ctx->mono_alt[k].stream) VO_HIP(ctx, hipStreamSynchronize(ctx->mono_alt[k].stream));
    VO_HIP(ctx, hipMemcpyAsync(ctx->rs_ofs, ofs.data(), ofs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(ctx, hipMemcpyAsync(ctx->rs_coef, coef.data(), coef.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(ctx, hipMemcpyAsync(ctx->pyr_rects, rects.data(), rects.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pyr_nbx = pyr_nbx; ctx->pyr_nby = pyr_nby; ctx->pyr_buf[0] = pyr_buf[0]; ctx->pyr_buf[1] = pyr_buf[1]; ctx->pyr_tab = pyr_tab;
    static_assert(sizeof(LevelsDev) <= sizeof(ctx->rs_meta_host), "level table too large");
    memcpy(ctx->rs_meta_host, &L, sizeof(L));
    ctx->orb_quota_nfeatures = -1;
    ctx->orb_w = w; ctx->orb_h = h;
    return VO_OK;
}

// ---------------------------------------------------------------------------------------
// pyramid
// ---------------------------------------------------------------------------------------
// The whole pyramid in ONE launch (was: a copy kernel + seven resize launches, each level waiting for the one before).
// resize(prev, cur, INTER_LINEAR_EXACT) makes level l a function of level l - 1, but only locally: output column v reads the
// columns ofs[v], ofs[v] + 1 of the level below.  So the pyramid is cut into cones: workgroup (bx, by) owns part (bx, by) of
// EVERY level and, per level, also computes the margin the next level's part reads (1 pixel at the top, growing by the scale
// factor per level down: <= 13 pixels at level 0) -- the intervals are prepared by the host (orb_prepare_tables).  A cone's
// levels live in two ping-pong LDS rectangles per image kind; the owned part of every level is written to the pyramid in
// HBM.  Values are those of the level-by-level kernels bit for bit: the same fixed-point taps in the same order; border
// columns / rows (v < min, v >= max: the clamped taps of the exact-bilinear table) become taps with weights (256, 0).
// The mask pyramid (feature_mask of stereo_odometer.py:38-41 at level 0, or an explicit mask) is resized the same way and
// then thresholded (threshold(254, THRESH_TOZERO)), as orb.cpp does.
struct PyrArgs {
    const uint8_t* img; const int16_t* disp16; const uint8_t* mask;
    int img_stride, disp_stride, mask_stride, mask_mode, min_d16, max_d16, nbx, nby, bufA, bufB, tab;
};

template <bool MASK>
__global__ void __launch_bounds__(256) k_orb_pyramid(const LevelsDev L, const PyrArgs A, const int32_t* __restrict__ rects,
                                                     const int32_t* __restrict__ ofs, const uint16_t* __restrict__ coef,
                                                     uint8_t* __restrict__ pimg, uint8_t* __restrict__ pmask, int32_t* __restrict__ cnt)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_pyr_generic[];
    // (explicit LDS address space: a pointer picked from an array by a run-time index is a generic pointer to the compiler, and
    // every access through it a flat_load / flat_store)
    typedef __attribute__((address_space(3))) uint8_t lds8;
    typedef __attribute__((address_space(3))) uint16_t lds16;
    typedef __attribute__((address_space(3))) int lds32;
    lds8* const s_pyr = (lds8*)s_pyr_generic;
    // [img even | img odd | mask even | mask odd | taps of levels 1 .. 7: per level x then y entries {o0, o1, c0, c1} (u16)]
    const unsigned o_img[2] = { 0u, (unsigned)A.bufA }, o_msk[2] = { (unsigned)(A.bufA + A.bufB), (unsigned)(2 * A.bufA + A.bufB) };
    lds16* const taps = (lds16*)(s_pyr + (MASK ? 2 : 1) * (A.bufA + A.bufB));           // [NL - 1][2 * tab][4]
    lds32* const s_rect = (lds32*)(taps + (size_t)(NL - 1) * 2 * A.tab * 4);            // [NL][4] (no static LDS: the dynamic region may be all 160 KB)
#define RECT(l, k) s_rect[(l) * 4 + (k)]
    const int bx = blockIdx.x, by = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (bx == 0 && by == 0 && threadIdx.x < CNT_HIST) cnt[threadIdx.x] = 0;           // the run's counters start here
    // every level's rectangle, then every level's taps: two batches of independent global loads instead of one dependent
    // round trip per level
    if (threadIdx.x < NL * 4) {
        const int l = threadIdx.x >> 2, k = threadIdx.x & 3;
        RECT(l, k) = k < 2 ? rects[(l * A.nbx + bx) * 2 + k] : rects[NL * A.nbx * 2 + (l * A.nby + by) * 2 + (k - 2)];
    }
    __syncthreads();
    // (all of a thread's table reads first, then the LDS writes: one global round trip for the lot instead of one per entry)
    {
        constexpr int TB = 8;
        const int ntap = (NL - 1) * 2 * A.tab;
        for (int ib = threadIdx.x; ib < ntap; ib += TB * blockDim.x) {
            int o0[TB], c0[TB], c1[TB], org[TB], kind[TB];          // kind: 0 = skip, 1 = clamped tap, 2 = table tap
#pragma unroll
            for (int u = 0; u < TB; u++) {
                const int i = ib + u * blockDim.x;
                kind[u] = 0; o0[u] = 0; c0[u] = 256; c1[u] = 0; org[u] = 0;
                if (i >= ntap) continue;
                const int l = 1 + i / (2 * A.tab), j = i - (l - 1) * 2 * A.tab;
                const bool isx = j < A.tab;
                const int e = isx ? j : j - A.tab;
                const int lo = RECT(l, isx ? 0 : 2), hi = RECT(l, isx ? 1 : 3);
                if (e > hi - lo) continue;
                const LevelDev d = L.l[l];
                const LevelDev p = L.l[l - 1];
                const int v = lo + e;
                const int mn = isx ? d.min_x : d.min_y, mx = isx ? d.max_x : d.max_y, t0 = isx ? d.xt : d.yt;
                const int ssz = isx ? p.w : p.h;
                org[u] = RECT(l - 1, isx ? 0 : 2);
                if (v < mn) { kind[u] = 1; o0[u] = 0; }
                else if (v >= mx) { kind[u] = 1; o0[u] = ssz - 1; }
                else { kind[u] = 2; o0[u] = ofs[t0 + v]; c0[u] = coef[2 * (size_t)(t0 + v)]; c1[u] = coef[2 * (size_t)(t0 + v) + 1]; }
            }
#pragma unroll
            for (int u = 0; u < TB; u++) {
                if (!kind[u]) continue;
                lds16* t = taps + (size_t)(ib + u * blockDim.x) * 4;
                t[0] = (uint16_t)(o0[u] - org[u]); t[1] = (uint16_t)(o0[u] + (kind[u] == 2 ? 1 : 0) - org[u]); t[2] = (uint16_t)c0[u]; t[3] = (uint16_t)c1[u];
            }
        }
    }
    // level 0: the cropped left image and its mask (feature_mask, or the caller's)
    {
        const LevelDev d = L.l[0];
        const int x0 = RECT(0, 0), x1 = RECT(0, 1), y0 = RECT(0, 2), y1 = RECT(0, 3);
        const int nw = x1 - x0 + 1, nh = y1 - y0 + 1;
        lds8* const b0i = s_pyr + o_img[0];
        lds8* const b0m = s_pyr + o_msk[0];
        const int ox0 = (int)((long long)bx * d.w / A.nbx), ox1 = (int)((long long)(bx + 1) * d.w / A.nbx);     // owned part [ox0, ox1)
        const int oy0 = (int)((long long)by * d.h / A.nby), oy1 = (int)((long long)(by + 1) * d.h / A.nby);
        // (twelve rows of a column per lane in flight: a load per iteration would be one dependent global round trip per row)
        constexpr int RB0 = 12;
        for (int xx = lane; xx < nw; xx += 64) {
            const int x = x0 + xx;
            const bool ownx = x >= ox0 && x < ox1;
            for (int yb = wv; yb < nh; yb += 4 * RB0) {
                uint8_t v[RB0], m[RB0];
#pragma unroll
                for (int u = 0; u < RB0; u++) {
                    const int yy = min(yb + 4 * u, nh - 1), y = y0 + yy;
                    v[u] = A.img[(size_t)y * A.img_stride + x];
                    if (MASK) {
                        if (A.mask_mode == 1) {
                            const int dd = A.disp16[(size_t)y * A.disp_stride + x];
                            m[u] = (dd >= A.min_d16 && dd <= A.max_d16) ? 255 : 0;
                        } else
                            m[u] = A.mask[(size_t)y * A.mask_stride + x];
                    }
                }
#pragma unroll
                for (int u = 0; u < RB0; u++) {
                    const int yy = yb + 4 * u, y = y0 + yy;
                    if (yy >= nh) break;
                    const bool own = ownx && y >= oy0 && y < oy1;
                    b0i[yy * nw + xx] = v[u];
                    if (own) pimg[(size_t)y * d.w + x] = v[u];
                    if (MASK) {
                        b0m[yy * nw + xx] = m[u];
                        if (own) pmask[(size_t)y * d.w + x] = m[u];
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int l = 1; l < NL; l++) {
        const LevelDev d = L.l[l];
        const int x0 = RECT(l, 0), x1 = RECT(l, 1), y0 = RECT(l, 2), y1 = RECT(l, 3);
        const int nw = x1 - x0 + 1, nh = y1 - y0 + 1;
        const int pnw = RECT(l - 1, 1) - RECT(l - 1, 0) + 1;                          // pitch of the rectangle below
        const int ox0 = (int)((long long)bx * d.w / A.nbx), ox1 = (int)((long long)(bx + 1) * d.w / A.nbx);
        const int oy0 = (int)((long long)by * d.h / A.nby), oy1 = (int)((long long)(by + 1) * d.h / A.nby);
        lds8* const di = s_pyr + o_img[l & 1];
        lds8* const dm = s_pyr + o_msk[l & 1];
        const lds8* const si = s_pyr + o_img[(l - 1) & 1];
        const lds8* const sm = s_pyr + o_msk[(l - 1) & 1];
        const lds16* const tx = taps + (size_t)(l - 1) * 2 * A.tab * 4;
        const lds16* const ty = tx + (size_t)A.tab * 4;
        for (int xx = lane; xx < nw; xx += 64) {            // a lane keeps its column's taps in registers while it walks the rows
            const int x = x0 + xx;
            const unsigned a0 = tx[4 * xx], a1 = tx[4 * xx + 1], xc0 = tx[4 * xx + 2], xc1 = tx[4 * xx + 3];
            const bool ownx = x >= ox0 && x < ox1;
            for (int yb = wv; yb < nh; yb += 16) {          // four rows per lane in flight (LDS reads first, then the arithmetic)
                unsigned r0[4], r1[4], yc0[4], yc1[4], ti[4][4], tm[4][4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int yy = min(yb + 4 * u, nh - 1);
                    r0[u] = ty[4 * yy] * (unsigned)pnw; r1[u] = ty[4 * yy + 1] * (unsigned)pnw; yc0[u] = ty[4 * yy + 2]; yc1[u] = ty[4 * yy + 3];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    ti[u][0] = si[r0[u] + a0]; ti[u][1] = si[r0[u] + a1]; ti[u][2] = si[r1[u] + a0]; ti[u][3] = si[r1[u] + a1];
                    if (MASK) { tm[u][0] = sm[r0[u] + a0]; tm[u][1] = sm[r0[u] + a1]; tm[u][2] = sm[r1[u] + a0]; tm[u][3] = sm[r1[u] + a1]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int yy = yb + 4 * u, y = y0 + yy;
                    if (yy >= nh) break;
                    const bool own = ownx && y >= oy0 && y < oy1;
                    {
                        const unsigned h0 = xc0 * ti[u][0] + xc1 * ti[u][1], h1 = xc0 * ti[u][2] + xc1 * ti[u][3];
                        const unsigned out = (h0 * yc0[u] + h1 * yc1[u] + 32768u) >> 16;
                        di[yy * nw + xx] = (uint8_t)out;
                        if (own) pimg[d.off + (size_t)y * d.w + x] = (uint8_t)out;
                    }
                    if (MASK) {
                        const unsigned h0 = xc0 * tm[u][0] + xc1 * tm[u][1], h1 = xc0 * tm[u][2] + xc1 * tm[u][3];
                        unsigned out = (h0 * yc0[u] + h1 * yc1[u] + 32768u) >> 16;
                        out = out > 254u ? out : 0u;
                        dm[yy * nw + xx] = (uint8_t)out;
                        if (own) pmask[d.off + (size_t)y * d.w + x] = (uint8_t)out;
                    }
                }
            }
        }
        __syncthreads();
    }
#undef RECT
}

// ---------------------------------------------------------------------------------------
// FAST-9/16
// ---------------------------------------------------------------------------------------
// cornerScore<16> of OpenCV's FAST: the largest threshold for which the pixel is still a corner; d[k] = centre - ring pixel k
__device__ __forceinline__ int fast_corner_score(const int* d)
{
    int a0 = FAST_T;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        int a = min(d[(k + 1) & 15], d[(k + 2) & 15]);
#pragma unroll
        for (int j = 3; j <= 8; j++) a = min(a, d[(k + j) & 15]);
        a0 = max(a0, min(a, d[k & 15]));
        a0 = max(a0, min(a, d[(k + 9) & 15]));
    }
    int b0 = -a0;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        int b = max(d[(k + 1) & 15], d[(k + 2) & 15]);
#pragma unroll
        for (int j = 3; j <= 8; j++) b = max(b, d[(k + j) & 15]);
        b0 = min(b0, max(b, d[k & 15]));
        b0 = min(b0, max(b, d[(k + 9) & 15]));
    }
    return -b0 - 1;
}

// FAST score + 3x3 non-max suppression (strict >) + pixel mask + image border in one pass; survivors are
// appended per level.  Block = 4 waves, tile = 64 columns x ORB_FAST_TH rows of the border-free interior: the scores
// of the tile and its 1-pixel halo go through LDS only (no score image), wave w then judges rows w TH/4 .. (w + 1) TH/4 - 1.
// The survivors of the whole tile reserve their slots with ONE returning global atomic (a per-wave atomic
// on a single counter serialises at ~12 ns each).
// Scoring is three passes over shrinking LDS lists instead of one divergent function per pixel (round 4: the kernel was
// 10 % of the pipeline's vector instructions -- a wave paid for the 16-pixel ring and the 200-instruction corner score as soon
// as ONE of its 64 pixels needed them, which in a textured image is nearly every wave):
//   A  every pixel of the tile + halo: the four compass points of the ring (a run of 9 covers at least two of them: fewer than
//      two darker and fewer than two brighter => not a corner); the survivors (10-25 %) are listed
//   B  the listed pixels, densely packed over the lanes: the ring's dark / bright masks and the run-of-9 test; survivors listed
//   C  the few real corners: cornerScore
// The tile's pixels (+ the ring's reach) are staged in LDS once: a ring read is one ds_read_u8 at a constant offset.
__global__ void __launch_bounds__(256) k_orb_fast_nms(const LevelsDev L, const uint8_t* __restrict__ pimg,
                                                      const uint8_t* __restrict__ pmask, int with_mask, int32_t* __restrict__ cand_pos,
                                                      float* __restrict__ cand_resp, int32_t* __restrict__ cnt)
{
    constexpr int TW = 64, TH = ORB_FAST_TH, RW = TH / 4, SW = TW + 2, SH = TH + 2, SP = SW + 2;   // padded LDS row
    constexpr int IW = SW + 6, IH = SH + 6, IP = IW + 4;                    // staged pixels: the scored region + 3 on each side
    constexpr int NPX = SW * SH;
    __shared__ uint8_t s_sc[SH * SP];
    __shared__ uint8_t s_img[IH * IP];
    __shared__ uint16_t s_listA[NPX], s_listB[NPX];
    __shared__ int s_nA, s_nB;
    __shared__ int s_cnt[TH];
    __shared__ int s_base;
    // ring offsets (dx,dy), OpenCV order
    constexpr int rx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
    constexpr int ry[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
    // A compact one-dimensional grid walks the tiles of all levels (level-major): a (tiles_x, tiles_y, levels) grid sized for
    // level 0 launched 6720 workgroups at config 2 of which 60 % returned at once -- and every workgroup of a short kernel has
    // to win a slot from the dispatcher against the other pairs' sweeps.
    int tiles_before[NL + 1];
    tiles_before[0] = 0;
#pragma unroll
    for (int l = 0; l < NL; l++) {
        const int iw = L.l[l].w - 2 * EDGE, ih = L.l[l].h - 2 * EDGE;
        tiles_before[l + 1] = tiles_before[l] + (iw > 0 && ih > 0 ? ((iw + TW - 1) / TW) * ((ih + TH - 1) / TH) : 0);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int tile = blockIdx.x; tile < tiles_before[NL]; tile += gridDim.x) {
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < NL; l++) lvl += tile >= tiles_before[l];
    const LevelDev d = L.l[lvl];
    const int tpr = (d.w - 2 * EDGE + TW - 1) / TW, tl = tile - tiles_before[lvl];
    const int x0 = (tl % tpr) * TW + EDGE, y0 = (tl / tpr) * TH + EDGE;
    const int xlim = d.w - EDGE + 1, ylim = d.h - EDGE + 1;   // scores are needed up to one pixel past the interior
    const uint8_t* img = pimg + d.off;
    // stage rows y0 - 4 .. y0 + TH + 4, columns x0 - 4 .. x0 + TW + 4 (x0, y0 >= EDGE: the low side is inside the image; the
    // high side is clamped -- a pixel whose ring would leave the image is beyond xlim / ylim and never scored)
    for (int i = threadIdx.x; i < IH * IW; i += blockDim.x) {
        const int ty = i / IW, tx = i - ty * IW;
        const int x = min(x0 - 4 + tx, d.w - 1), y = min(y0 - 4 + ty, d.h - 1);
        s_img[ty * IP + tx] = img[(size_t)y * d.w + x];
    }
    for (int i = threadIdx.x; i < SH * SP; i += blockDim.x) s_sc[i] = 0;
    if (threadIdx.x == 0) { s_nA = 0; s_nB = 0; }
    __syncthreads();
    // pass A
    for (int i0 = 0; i0 < NPX; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        const int ty = i / SW, tx = i - ty * SW;
        bool pass = false;
        if (i < NPX && x0 - 1 + tx < xlim && y0 - 1 + ty < ylim) {   // x, y >= EDGE - 1 >= 3 always
            const uint8_t* c = s_img + (ty + 3) * IP + tx + 3;
            const int v = c[0];
            int nd = 0, nb = 0;
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                const int q = c[ry[k] * IP + rx[k]];
                nd += q < v - FAST_T;
                nb += q > v + FAST_T;
            }
            pass = nd >= 2 || nb >= 2;
        }
        const unsigned long long bal = __ballot(pass);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_nA, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (pass) s_listA[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
    }
    __syncthreads();
    // pass B
    const int nA = s_nA;
    for (int e0 = 0; e0 < nA; e0 += blockDim.x) {
        const int e = e0 + threadIdx.x;
        bool pass = false;
        int i = 0;
        if (e < nA) {
            i = s_listA[e];
            const int ty = i / SW, tx = i - ty * SW;
            const uint8_t* c = s_img + (ty + 3) * IP + tx + 3;
            const int v = c[0];
            unsigned dark = 0, bright = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int q = c[ry[k] * IP + rx[k]];
                dark |= (unsigned)(q < v - FAST_T) << k;
                bright |= (unsigned)(q > v + FAST_T) << k;
            }
            // >= 9 contiguous set bits on the 16-ring
            auto run9 = [](unsigned m) -> bool {
                unsigned r = m | (m << 16);
                unsigned a = r & (r >> 1);      // runs of 2
                a = a & (a >> 2);               // runs of 4
                a = a & (a >> 4);               // runs of 8
                a = a & (r >> 8);               // runs of 9
                return (a & 0xFFFFu) != 0;
            };
            pass = run9(dark) || run9(bright);
        }
        const unsigned long long bal = __ballot(pass);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_nB, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (pass) s_listB[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
    }
    __syncthreads();
    // pass C
    const int nB = s_nB;
    for (int e = threadIdx.x; e < nB; e += blockDim.x) {
        const int i = s_listB[e];
        const int ty = i / SW, tx = i - ty * SW;
        const uint8_t* c = s_img + (ty + 3) * IP + tx + 3;
        const int v = c[0];
        int dd[16];
#pragma unroll
        for (int k = 0; k < 16; k++) dd[k] = v - (int)c[ry[k] * IP + rx[k]];
        s_sc[ty * SP + tx] = (uint8_t)fast_corner_score(dd);
    }
    __syncthreads();
    const int x = x0 + lane;
    int sc[RW];
    unsigned long long bal[RW];
#pragma unroll
    for (int r = 0; r < RW; r++) {
        const int ty = wv * RW + r, y = y0 + ty;
        int s = 0;
        if (x < d.w - EDGE && y < d.h - EDGE) {
            const uint8_t* q = s_sc + (ty + 1) * SP + lane + 1;
            s = q[0];
            if (s && !(s > q[-1] && s > q[1] && s > q[-SP - 1] && s > q[-SP] && s > q[-SP + 1] && s > q[SP - 1] && s > q[SP] && s > q[SP + 1])) s = 0;
            if (s && with_mask && pmask[d.off + (size_t)y * d.w + x] == 0) s = 0;
        }
        sc[r] = s;
        bal[r] = __ballot(s != 0);
        if (lane == 0) s_cnt[wv * RW + r] = __popcll(bal[r]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < TH; k++) { const int c = s_cnt[k]; s_cnt[k] = tot; tot += c; }  // exclusive prefix
        s_base = tot ? atomicAdd(&cnt[CNT_CAND + lvl], tot) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RW; r++) {
        if (sc[r]) {
            const int y = y0 + wv * RW + r;
            const int slot = s_base + s_cnt[wv * RW + r] + __popcll(bal[r] & ((1ull << lane) - 1ull));
            cand_pos[d.cand_off + slot] = y * d.w + x;
            cand_resp[d.cand_off + slot] = (float)sc[r];
        }
    }
    __syncthreads();                                 // (the next tile reuses the LDS arrays)
    }
}

// Walks a 256-bin histogram (LDS) from bin 255 downwards until the running count reaches `need`: returns that bin and the
// count ABOVE it.  By the first wave of the block, all 64 lanes (lane l sums bins 255 - 4 l .. 252 - 4 l, one wave-wide
// inclusive prefix, the lane where the prefix crosses `need` resolves its four bins): ~40 instructions instead of a 256-step
// loop of dependent LDS reads on one thread (8 us each, five of them per level in the selection).  Returns bin -1 when the
// whole histogram holds fewer than `need` entries.
__device__ __forceinline__ void hist_scan_desc(const int* hist, int need, int lane, int& bin, int& above)
{
    const int b0 = 255 - 4 * lane;
    const int h0 = hist[b0], h1 = hist[b0 - 1], h2 = hist[b0 - 2], h3 = hist[b0 - 3];
    const int s = (h0 + h1) + (h2 + h3);
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const unsigned long long hit = __ballot(inc >= need);
    bin = -1; above = 0;
    if (hit) {
        const int f = __ffsll((long long)hit) - 1;
        int c = __shfl(inc - s, f, 64);
        const int g0 = __shfl(h0, f, 64), g1 = __shfl(h1, f, 64), g2 = __shfl(h2, f, 64);
        const int fb = 255 - 4 * f;
        if (c + g0 >= need) { bin = fb; above = c; }
        else if (c + g0 + g1 >= need) { bin = fb - 1; above = c + g0; }
        else if (c + g0 + g1 + g2 >= need) { bin = fb - 2; above = c + g0 + g1; }
        else { bin = fb - 3; above = c + g0 + g1 + g2; }
    }
}

// retainBest(2*quota) by FAST score: keep every candidate whose score >= the n-th largest.  The survivors go to `lds_pos`
// when all of them fit (lds_cap entries; the fused kernel), to outA otherwise; s_inlds says which.
__device__ __forceinline__ void orb_fast_select(const LevelsDev& L, const int32_t* cand_pos,
                                                const float* cand_resp, int32_t* outA,
                                                int32_t* cnt, int* s_hist, int& s_thr, int& s_n, int32_t* lds_pos, int lds_cap, int& s_inlds)
{
    const int lvl = blockIdx.x;
    const LevelDev d = L.l[lvl];
    const int n = cnt[CNT_CAND + lvl], keep = 2 * d.quota;
    const int32_t* pos = cand_pos + d.cand_off;
    const float* resp = cand_resp + d.cand_off;
    const int nt = blockDim.x;
    // histogram of the (integer) FAST scores in LDS
    for (int b = threadIdx.x; b < 256; b += nt) s_hist[b] = 0;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (keep > 0 && n > keep)
        for (int i0 = threadIdx.x; i0 < n; i0 += 4 * nt) {         // four loads in flight per thread: the list is long, the block alone
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = i0 + k * nt < n ? resp[i0 + k * nt] : -1.f;
#pragma unroll
            for (int k = 0; k < 4; k++) if (r[k] >= 0.f) atomicAdd(&s_hist[(int)r[k] & 255], 1);
        }
    __syncthreads();
    if (threadIdx.x < 64) {
        int thr = 0, kept = n;
        if (keep == 0) { thr = 1 << 30; kept = 0; }
        else if (n > keep) {
            int bin, above;
            hist_scan_desc(s_hist, keep, threadIdx.x, bin, above);
            thr = max(bin, 0);
            kept = above + s_hist[thr];
        }
        if (threadIdx.x == 0) {
            s_thr = thr;
            s_inlds = lds_pos != nullptr && kept <= lds_cap;
        }
    }
    __syncthreads();
    const int thr = s_thr;
    int32_t* const dst = s_inlds ? lds_pos : outA + d.cand_off;
    const int lane = threadIdx.x & 63;
    for (int i0 = 0; i0 < n; i0 += 4 * nt) {                        // four candidates per thread in flight
        float r[4];
        int p[4];
        unsigned long long bal[4];
        int tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + k * nt + threadIdx.x;
            r[k] = i < n ? resp[i] : -1.f;
            p[k] = i < n ? pos[i] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            bal[k] = __ballot(r[k] >= 0.f && (int)r[k] >= thr);
            tot += __popcll(bal[k]);
        }
        int base = 0;
        if (lane == 0 && tot) base = atomicAdd(&s_n, tot);
        base = __shfl(base, 0, 64);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((bal[k] >> lane) & 1ull) dst[base + __popcll(bal[k] & ((1ull << lane) - 1ull))] = p[k];
            base += __popcll(bal[k]);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) cnt[CNT_A + lvl] = s_n;
}

__device__ __forceinline__ int wave_sum_i32(int v);
#define ORB_RANK_LDS 8192      // positions the canonical-order pass of k_orb_select_harris keeps in LDS (32 KB)

__device__ __forceinline__ unsigned f2key(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Harris response of one candidate by one wave: lane = one of the 7x7 pixels.  Integer sums (the order of the additions
// does not matter); the float tail is evaluated by every lane alike, lane 0's copy is the one stored.
__device__ __forceinline__ float orb_harris_at(const uint8_t* __restrict__ lvl_img, int w, int pos, int lane)
{
    const int dy = lane / 7 - 3, dx = lane % 7 - 3;
    int a = 0, b = 0, cc = 0;
    if (lane < 49) {
        const uint8_t* p = lvl_img + pos + dy * w + dx;
        const int Ix = (p[1] - p[-1]) * 2 + (p[-w + 1] - p[-w - 1]) + (p[w + 1] - p[w - 1]);
        const int Iy = (p[w] - p[-w]) * 2 + (p[w - 1] - p[-w - 1]) + (p[w + 1] - p[-w + 1]);
        a = Ix * Ix; b = Iy * Iy; cc = Ix * Iy;
    }
    a = wave_sum_i32(a); b = wave_sum_i32(b); cc = wave_sum_i32(cc);
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale4 = scale * scale * scale * scale;
    const float fa = (float)a, fb = (float)b, fc = (float)cc;
    const float t1 = fa * fb, t2 = fc * fc, sm = fa + fb;
    const float t4 = (0.04f * sm) * sm;
    return ((t1 - t2) - t4) * scale4;
}

// retainBest(quota) by Harris response (radix select of the quota-th largest, ties kept), then
// sort the survivors by position and stage them per level.  pos / resp: the level's candidate list (n entries), tp / tr:
// scratch for the survivors -- global arrays, or LDS ones (scratch_in_lds: the rank pass then reads tp itself).
// (PI / PF / SI / SF: pointer types of the candidate list and of the scratch -- global, or LDS with its address space spelled
// out: through generic pointers every access would be a flat_load / flat_store)
template <typename PI, typename PF, typename SI, typename SF>
__device__ __forceinline__ void orb_harris_select(const LevelsDev& L, PI pos, PF resp, int n,
                                                  int32_t* fin_pos, float* fin_resp, SI tp, SF tr, bool scratch_in_lds,
                                                  int32_t* cnt, int* hist, int* s_tp, int rank_cap,
                                                  unsigned& s_prefix, unsigned& s_mask, int& s_remaining, int& s_nf)
{
    const int lvl = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const LevelDev d = L.l[lvl];
    const int keep = d.quota;
    unsigned thr_key = 0;
    if (keep <= 0) thr_key = 0xFFFFFFFFu;  // keep nothing (n_points == 0 clears)
    else if (n > keep) {
        if (tid == 0) { s_prefix = 0; s_mask = 0; s_remaining = keep; }
        __syncthreads();
        for (int pass = 3; pass >= 0; pass--) {
            const int shift = pass * 8;
            for (int b = tid; b < 256; b += nt) hist[b] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix, mask = s_mask;
            for (int i = tid; i < n; i += nt) {
                unsigned k = f2key(resp[i]);
                if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid < 64) {
                int bin, above;
                const int rem = s_remaining;
                hist_scan_desc(hist, rem, tid, bin, above);
                if (tid == 0) {
                    if (bin >= 0) { s_prefix = prefix | ((unsigned)bin << shift); s_remaining = rem - above; }
                    s_mask = mask | (255u << shift);
                }
            }
            __syncthreads();
        }
        thr_key = s_prefix;
    }
    if (tid == 0) s_nf = 0;
    __syncthreads();
    for (int i = tid; i < n; i += nt)
        if (keep > 0 && f2key(resp[i]) >= thr_key) {
            int slot = atomicAdd(&s_nf, 1);
            tp[slot] = pos[i];
            tr[slot] = resp[i];
        }
    __syncthreads();
    const int nf = s_nf;
    // canonical order: ascending position (rank by counting; positions are unique).  Every thread reads every position:
    // from LDS when they fit
    const bool in_lds = scratch_in_lds || nf <= rank_cap;
    if (in_lds && !scratch_in_lds)
        for (int i = tid; i < nf; i += nt) s_tp[i] = tp[i];
    __syncthreads();
    for (int i = tid; i < nf; i += nt) {
        const int pi = tp[i];
        int rank = 0;
        if (scratch_in_lds) for (int j = 0; j < nf; j++) rank += tp[j] < pi;
        else if (in_lds) for (int j = 0; j < nf; j++) rank += s_tp[j] < pi;
        else for (int j = 0; j < nf; j++) rank += tp[j] < pi;
        fin_pos[d.cand_off + rank] = pi;
        fin_resp[d.cand_off + rank] = tr[i];
    }
    if (tid == 0) cnt[CNT_FIN + lvl] = nf;
}

// The selection in three launches (nfeatures > 2000): one block per pyramid level keeps retainBest(2*quota) by FAST score;
// the Harris responses of the survivors are one WAVE per candidate across the whole device (lane = one of the 7x7 pixels;
// at 8000 features there are ~16000 of them and one block per level spent 0.5 ms on this step alone); one block per level
// keeps retainBest(quota) by Harris and writes the canonical order.
// (fin_* alias cand_*: the final lists replace the NMS lists, which are dead by then -- no __restrict__ there)
__global__ void __launch_bounds__(1024) k_orb_select_fast(const LevelsDev L, const int32_t* __restrict__ cand_pos, const float* __restrict__ cand_resp,
                                                          int32_t* __restrict__ candA_pos, int32_t* __restrict__ cnt)
{
    __shared__ int s_hist[256];
    __shared__ int s_thr, s_n, s_inlds;
    orb_fast_select(L, cand_pos, cand_resp, candA_pos, cnt, s_hist, s_thr, s_n, nullptr, 0, s_inlds);
}

__global__ void __launch_bounds__(256) k_orb_harris(const LevelsDev L, const uint8_t* __restrict__ pimg, const int32_t* __restrict__ candA_pos,
                                                    float* __restrict__ candA_resp, const int32_t* __restrict__ cnt)
{
    const int lvl = blockIdx.y;
    const LevelDev d = L.l[lvl];
    const int nA = cnt[CNT_A + lvl];
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwv = gridDim.x * (blockDim.x >> 6);
    for (int i = wv; i < nA; i += nwv) {
        const float r = orb_harris_at(pimg + d.off, d.w, candA_pos[d.cand_off + i], lane);
        if (lane == 0) candA_resp[d.cand_off + i] = r;
    }
}

__global__ void __launch_bounds__(1024) k_orb_select_harris(const LevelsDev L, const int32_t* candA_pos, const float* candA_resp,
                                                            int32_t* fin_pos, float* fin_resp, int32_t* tmp_pos, float* tmp_resp, int32_t* cnt)
{
    __shared__ int s_hist[256];
    __shared__ int s_remaining, s_nf;
    __shared__ unsigned s_prefix, s_mask;
    extern __shared__ int s_tp[];
    const LevelDev d = L.l[blockIdx.x];
    orb_harris_select(L, candA_pos + d.cand_off, candA_resp + d.cand_off, cnt[CNT_A + blockIdx.x], fin_pos, fin_resp, tmp_pos + d.cand_off,
                      tmp_resp + d.cand_off, false, cnt, s_hist, s_tp, ORB_RANK_LDS, s_prefix, s_mask, s_remaining, s_nf);
}

// The same selection as ONE launch (nfeatures <= 2000: the odometer's 500): a block per level runs the three steps back to
// back and keeps the lists between them in LDS -- the FAST survivors (2 x quota + score ties), their Harris responses (a wave
// per candidate, 16 waves), the Harris survivors -- so nothing but the NMS list is read from and nothing but the final list
// written to HBM.  As three launches each step was a chain of dependent global round trips by eight lonely blocks
// (35 + 7 + 42 us at config 2).  A level whose survivors do not fit the LDS lists (ORB_SEL_CAP: a flood of score ties) falls
// back to the global scratch arrays, inside the same launch.
#define ORB_SEL_CAP 2048
__global__ void __launch_bounds__(1024) k_orb_select(const LevelsDev L, const uint8_t* __restrict__ pimg, const int32_t* cand_pos, const float* cand_resp,
                                                     int32_t* candA_pos, float* candA_resp, int32_t* fin_pos, float* fin_resp,
                                                     int32_t* tmp_pos, float* tmp_resp, int32_t* cnt)
{
    __shared__ int s_hist[256];
    __shared__ int s_thr, s_n, s_inlds, s_remaining, s_nf;
    __shared__ unsigned s_prefix, s_mask;
    __shared__ int32_t s_posA[ORB_SEL_CAP], s_posB[ORB_SEL_CAP];
    __shared__ float s_respA[ORB_SEL_CAP], s_respB[ORB_SEL_CAP];
    const int lvl = blockIdx.x;
    const LevelDev d = L.l[lvl];
    orb_fast_select(L, cand_pos, cand_resp, candA_pos, cnt, s_hist, s_thr, s_n, s_posA, ORB_SEL_CAP, s_inlds);
    __syncthreads();
    const int nA = s_n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    auto harris = [&](auto posA, auto respA) {
        for (int i = wv; i < nA; i += 4 * nwv) {                     // four candidates per wave in flight
            int p[4];
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) p[k] = posA[min(i + k * nwv, nA - 1)];
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = orb_harris_at(pimg + d.off, d.w, p[k], lane);
#pragma unroll
            for (int k = 0; k < 4; k++) if (lane == 0 && i + k * nwv < nA) respA[i + k * nwv] = r[k];
        }
    };
    typedef __attribute__((address_space(3))) int32_t lds_i32;
    typedef __attribute__((address_space(3))) float lds_f32;
    if (s_inlds) {                                                   // (block-uniform)
        lds_i32* const pA = (lds_i32*)s_posA;
        lds_f32* const rA = (lds_f32*)s_respA;
        harris(pA, rA);
        __syncthreads();
        orb_harris_select(L, (const lds_i32*)pA, (const lds_f32*)rA, nA, fin_pos, fin_resp, (lds_i32*)s_posB, (lds_f32*)s_respB, true, cnt, s_hist,
                          s_posB, ORB_SEL_CAP, s_prefix, s_mask, s_remaining, s_nf);
    } else {
        harris((const int32_t*)(candA_pos + d.cand_off), candA_resp + d.cand_off);
        __syncthreads();
        orb_harris_select(L, (const int32_t*)(candA_pos + d.cand_off), (const float*)(candA_resp + d.cand_off), nA, fin_pos, fin_resp,
                          tmp_pos + d.cand_off, tmp_resp + d.cand_off, false, cnt, s_hist, s_posB, ORB_SEL_CAP, s_prefix, s_mask, s_remaining, s_nf);
    }
}

// ---------------------------------------------------------------------------------------
// GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) as the sepFilter2D 8-bit path evaluates it:
// kernel round(256*g) = [18,34,49,55,49,34,18], row pass exact in 16 bits, column pass
// (sum + 2^15) >> 16 saturated.
// ---------------------------------------------------------------------------------------
__constant__ int c_gk[7] = { 18, 34, 49, 55, 49, 34, 18 };

// ---------------------------------------------------------------------------------------
// orientation (intensity centroid) + rotated BRIEF, one wave per keypoint
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__device__ __forceinline__ int wave_sum_i32(int v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The descriptor samples the 7x7-Gaussian-blurred level (GaussianBlur(7,7,2,2), 8-bit fixed point, row pass
// to 16 bit then column pass) at rotated pattern points within 18 pixels of the keypoint.  Blurring the whole
// pyramid for a few hundred keypoints wasted two launches and ~40 M multiply-adds: each wave blurs just its
// own 39x39 patch into LDS (same two passes, same rounding; the 22-pixel reach stays inside the 31-pixel
// keypoint border, so no reflection is ever needed).
#define DESC_R 19
#define DESC_W (2 * DESC_R + 1)        // 39
#define DESC_HR (DESC_W + 6)           // 45 rows after the row pass
#define DESC_LD 40
__global__ void __launch_bounds__(256) k_orb_describe(const LevelsDev L, const uint8_t* __restrict__ pimg,
                                                     const int32_t* __restrict__ fin_pos, const float* __restrict__ fin_resp,
                                                     int32_t* __restrict__ cnt, int cap, float* __restrict__ kp_xy,
                                                     float* __restrict__ kp_size, float* __restrict__ kp_resp,
                                                     int32_t* __restrict__ kp_oct, float* __restrict__ kp_angle,
                                                     uint8_t* __restrict__ desc, int32_t* n_kp_host)
{
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // the levels' final lists are concatenated in level order: find this wave's keypoint
    int lvl = 0, base = 0, total = 0;
#pragma unroll
    for (int l = 0; l < NL; l++) {
        const int c = cnt[CNT_FIN + l];
        if (k >= total + c) { lvl = l + 1; base = total + c; }
        total += c;
    }
    if (k == 0 && lane == 0) {
        cnt[CNT_TOTAL] = total;
        // the keypoint count goes straight into the slot's pinned host word (a 4-byte copy command behind this kernel was one
        // more queue entry per pair)
        __hip_atomic_store(n_kp_host, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (k >= min(total, cap)) return;
    const LevelDev d = L.l[lvl];
    const int w = d.w, pos = fin_pos[d.cand_off + (k - base)];
    if (lane == 0) {
        const int px = pos % w, py = pos / w;
        kp_xy[2 * k] = (float)px * d.scale;
        kp_xy[2 * k + 1] = (float)py * d.scale;
        kp_size[k] = 31 * d.scale;
        kp_resp[k] = fin_resp[d.cand_off + (k - base)];
        kp_oct[k] = lvl;
    }
    // intensity centroid over the circular patch: lanes 0..30 <-> u = lane-15
    const uint8_t* ctr = pimg + d.off + pos;
    int m10 = 0, m01 = 0;
    const int u = lane - HALF_PATCH;
    if (lane <= 2 * HALF_PATCH) {
        const int au = abs(u);
        for (int v = -HALF_PATCH; v <= HALF_PATCH; v++) {
            if (au <= L.umax[abs(v)]) {
                int val = ctr[v * w + u];
                m10 += u * val;
                m01 += v * val;
            }
        }
    }
    m10 = wave_sum_i32(m10);
    m01 = wave_sum_i32(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    if (lane == 0) kp_angle[k] = angle;
    // rotated BRIEF on the blurred level; lane j < 32 produces descriptor byte j
    const float iscale = 1.f / d.scale;
    const float ar = angle * (float)(3.1415926535897932384626433832795 / 180.f);
    const float ca = (float)cos((double)ar), sa = (float)sin((double)ar);
    // (centre exactly as OpenCV derives it: cvRound(kpt.pt * (1 / scale)) of the float32 point written above)
    const int cx = __float2int_rn(((float)(pos % w) * d.scale) * iscale), cy = __float2int_rn(((float)(pos / w) * d.scale) * iscale);
    __shared__ uint16_t s_row[4][DESC_HR * DESC_LD];
    __shared__ uint8_t s_blur[4][DESC_W * DESC_LD];
    uint16_t* rowp = s_row[threadIdx.x >> 6];
    uint8_t* blr = s_blur[threadIdx.x >> 6];
    const uint8_t* src = pimg + d.off + (size_t)(cy - DESC_R - 3) * w + (cx - DESC_R);
    for (int i = lane; i < DESC_HR * DESC_W; i += 64) {          // row pass
        const int r = i / DESC_W, c = i - r * DESC_W;
        const uint8_t* q = src + (size_t)r * w + c;
        int sum = 0;
#pragma unroll
        for (int t = -3; t <= 3; t++) sum += c_gk[t + 3] * q[t];
        rowp[r * DESC_LD + c] = (uint16_t)sum;
    }
    __builtin_amdgcn_wave_barrier();   // LDS accesses of one wave are served in order
    for (int i = lane; i < DESC_W * DESC_W; i += 64) {           // column pass
        const int r = i / DESC_W, c = i - r * DESC_W;
        int sum = 0;
#pragma unroll
        for (int t = 0; t < 7; t++) sum += c_gk[t] * (int)rowp[(r + t) * DESC_LD + c];
        sum = (sum + (1 << 15)) >> 16;
        blr[r * DESC_LD + c] = (uint8_t)min(sum, 255);
    }
    __builtin_amdgcn_wave_barrier();
    const uint8_t* bc = blr + DESC_R * DESC_LD + DESC_R;
    if (lane < 32) {
        int val = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const int8_t* pt = c_pattern + (lane * 8 + b) * 4;
            const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
            const int ix0 = __float2int_rn(x0 * ca - y0 * sa), iy0 = __float2int_rn(x0 * sa + y0 * ca);
            const int ix1 = __float2int_rn(x1 * ca - y1 * sa), iy1 = __float2int_rn(x1 * sa + y1 * ca);
            const int t0 = bc[iy0 * DESC_LD + ix0], t1 = bc[iy1 * DESC_LD + ix1];
            val |= (t0 < t1) << b;
        }
        desc[(size_t)k * 32 + lane] = (uint8_t)val;
    }
}

// ---------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------
// Enqueue one extraction on ctx->stream (scratch: (*ctx->orbws)).  The keypoint count lands in the slot's
// pinned word; orb_finish() reads it once the stream (or the slot's `ready` event) has been waited for.
static int orb_enqueue(vo_ctx* ctx, FrameSlot* fs, const uint8_t* d_img, int img_stride, int w, int h, int nfeatures, int mask_mode,
                       const int16_t* d_disp16, int disp_stride, int min_d16, int max_d16, const uint8_t* d_mask, int mask_stride)
{
    fs->has_kp = false;
    if (w <= 2 * EDGE || h <= 2 * EDGE) {
        // level 0 has no pixel inside the border, and no smaller level has one: runByImageBorder clears every level
        *fs->n_kp_host = 0;
        return VO_OK;
    }
    int rc = orb_prepare_tables(ctx, w, h);
    if (rc) return rc;
    // per-level quotas (computeKeyPoints)
    LevelsDev* Lh = (LevelsDev*)ctx->rs_meta_host;
    {
        float factor = (float)(1.0 / (double)1.2f);
        float nd = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)NL));
        int sum = 0;
        for (int l = 0; l < NL - 1; l++) {
            Lh->l[l].quota = cv_round_f(nd);
            sum += Lh->l[l].quota;
            nd *= factor;
        }
        Lh->l[NL - 1].quota = nfeatures - sum > 0 ? nfeatures - sum : 0;
    }
    // The level table (geometry + this call's quotas, 452 bytes) travels BY VALUE in every launch: extractions
    // queued on other engines' streams with other nfeatures keep the quotas they were enqueued with.
    const LevelsDev dL = *Lh;
    const int with_mask = mask_mode != 0;
    StageTimer t(ctx, VO_T_ORB);
    {
        PyrArgs pa;
        pa.img = d_img; pa.disp16 = d_disp16; pa.mask = d_mask;
        pa.img_stride = img_stride; pa.disp_stride = disp_stride; pa.mask_stride = mask_stride; pa.mask_mode = mask_mode;
        pa.min_d16 = min_d16; pa.max_d16 = max_d16; pa.nbx = ctx->pyr_nbx; pa.nby = ctx->pyr_nby;
        pa.bufA = ctx->pyr_buf[0]; pa.bufB = ctx->pyr_buf[1]; pa.tab = ctx->pyr_tab;
        const size_t lds = (size_t)(with_mask ? 2 : 1) * (pa.bufA + pa.bufB) + (size_t)(NL - 1) * 2 * pa.tab * 8 + NL * 16;
        if (lds > 150 * 1024) return vo_fail(ctx, VO_E_CAP, "pyramid cones of %dx%d need %zu bytes of LDS", w, h, lds);
        auto kp = with_mask ? k_orb_pyramid<true> : k_orb_pyramid<false>;
        static unsigned long long attr_set[2] = { 0, 0 };     // per instantiation and device: allow more than 64 KB of dynamic LDS
        if (!((attr_set[with_mask] >> (ctx->device & 63)) & 1ull)) {
            VO_HIP(ctx, hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[with_mask] |= 1ull << (ctx->device & 63);
        }
        hipLaunchKernelGGL(kp, dim3(pa.nbx, pa.nby), dim3(256), lds, ctx->stream, dL, pa, ctx->pyr_rects, ctx->rs_ofs, ctx->rs_coef,
                           ctx->orbws->pyr_img, ctx->orbws->pyr_mask, ctx->orbws->counters);
    }
    int fast_tiles = 0;
    for (int l = 0; l < NL; l++)
        if (Lh->l[l].w > 2 * EDGE && Lh->l[l].h > 2 * EDGE) fast_tiles += div_up(Lh->l[l].w - 2 * EDGE, 64) * div_up(Lh->l[l].h - 2 * EDGE, ORB_FAST_TH);
    hipLaunchKernelGGL(k_orb_fast_nms, dim3(std::max(1, std::min(fast_tiles, ORB_FAST_GRID_CAP))), dim3(256), 0, ctx->stream, dL,
                       ctx->orbws->pyr_img, ctx->orbws->pyr_mask, with_mask, ctx->orbws->cand_pos, ctx->orbws->cand_resp, ctx->orbws->counters);
    // after the select, cand_* hold the per-level final lists; candB_* are scratch
    if (nfeatures <= 2000) {
        hipLaunchKernelGGL(k_orb_select, dim3(NL), dim3(1024), 0, ctx->stream, dL, ctx->orbws->pyr_img, ctx->orbws->cand_pos, ctx->orbws->cand_resp,
                           ctx->orbws->candA_pos, ctx->orbws->candA_resp, ctx->orbws->cand_pos, ctx->orbws->cand_resp, ctx->orbws->candB_pos,
                           ctx->orbws->candB_resp, ctx->orbws->counters);
    } else {
        hipLaunchKernelGGL(k_orb_select_fast, dim3(NL), dim3(1024), 0, ctx->stream, dL, ctx->orbws->cand_pos, ctx->orbws->cand_resp, ctx->orbws->candA_pos,
                           ctx->orbws->counters);
        hipLaunchKernelGGL(k_orb_harris, dim3(256, NL), dim3(256), 0, ctx->stream, dL, ctx->orbws->pyr_img, ctx->orbws->candA_pos,
                           ctx->orbws->candA_resp, ctx->orbws->counters);
        hipLaunchKernelGGL(k_orb_select_harris, dim3(NL), dim3(1024), (size_t)ORB_RANK_LDS * 4, ctx->stream, dL, ctx->orbws->candA_pos, ctx->orbws->candA_resp,
                           ctx->orbws->cand_pos, ctx->orbws->cand_resp, ctx->orbws->candB_pos, ctx->orbws->candB_resp, ctx->orbws->counters);
    }
    hipLaunchKernelGGL(k_orb_describe, dim3(div_up(ctx->kp_cap, 4)), dim3(256), 0, ctx->stream, dL, ctx->orbws->pyr_img, ctx->orbws->cand_pos,
                       ctx->orbws->cand_resp, ctx->orbws->counters, ctx->kp_cap, fs->kp_xy, fs->kp_size, fs->kp_resp, fs->kp_oct, fs->kp_angle,
                       fs->desc, fs->n_kp_host);
    VO_CHECK_LAUNCH(ctx);
    return VO_OK;
}

static int orb_finish(vo_ctx* ctx, FrameSlot* fs)
{
    const int total = *fs->n_kp_host;
    if (total > ctx->kp_cap)
        return vo_fail(ctx, VO_E_CAP, "%d keypoints (response ties included) exceed capacity %d; raise max_kp", total, ctx->kp_cap);
    fs->n_kp = total;
    fs->has_kp = true;
    return VO_OK;
}

int orb_run(vo_ctx* ctx, FrameSlot* fs, const uint8_t* d_img, int img_stride, int w, int h, int nfeatures, int mask_mode,
            const int16_t* d_disp16, int disp_stride, int min_d16, int max_d16, const uint8_t* d_mask, int mask_stride)
{
    int rc = orb_enqueue(ctx, fs, d_img, img_stride, w, h, nfeatures, mask_mode, d_disp16, disp_stride, min_d16, max_d16, d_mask, mask_stride);
    if (rc) return rc;
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return orb_finish(ctx, fs);
}

// extraction on a slot's left image inside the valid ROI with the fused disparity mask (mask_mode 1) or
// none (0); enqueue only
int orb_slot_enqueue(vo_ctx* ctx, FrameSlot& f, int nfeatures, int mask_mode, int min_disp16, int max_disp16)
{
    int x0 = 0, y0 = 0, x1 = f.w, y1 = f.h;
    if (ctx->has_roi) { x0 = ctx->roi[0]; y0 = ctx->roi[1]; x1 = ctx->roi[2] < f.w ? ctx->roi[2] : f.w; y1 = ctx->roi[3] < f.h ? ctx->roi[3] : f.h; }
    const int cw = x1 - x0, ch = y1 - y0;
    if (cw <= 0 || ch <= 0) { f.has_kp = false; *f.n_kp_host = 0; return VO_OK; }
    return orb_enqueue(ctx, &f, f.left + (size_t)y0 * f.w + x0, f.w, cw, ch, nfeatures, mask_mode,
                       f.disp16 + (size_t)y0 * f.w + x0, f.w, min_disp16, max_disp16, nullptr, 0);
}

static int download_kps(vo_ctx* ctx, FrameSlot& f, float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                        int32_t* kp_octave, uint8_t* desc, int cap, int* n_out)
{
    const int n = f.n_kp;
    if (n_out) *n_out = n;
    if (n == 0) return VO_OK;
    const bool any = kp_xy || kp_size || kp_angle || kp_response || kp_octave || desc;
    if (any && n > cap) return vo_fail(ctx, VO_E_CAP, "%d keypoints exceed the output capacity %d", n, cap);
    int rc = VO_OK;
    if (kp_xy && !rc) rc = xfer_d2h(ctx, kp_xy, f.kp_xy, (size_t)n * 8);
    if (kp_size && !rc) rc = xfer_d2h(ctx, kp_size, f.kp_size, (size_t)n * 4);
    if (kp_angle && !rc) rc = xfer_d2h(ctx, kp_angle, f.kp_angle, (size_t)n * 4);
    if (kp_response && !rc) rc = xfer_d2h(ctx, kp_response, f.kp_resp, (size_t)n * 4);
    if (kp_octave && !rc) rc = xfer_d2h(ctx, kp_octave, f.kp_oct, (size_t)n * 4);
    if (desc && !rc) rc = xfer_d2h(ctx, desc, f.desc, (size_t)n * 32);
    if (rc) return rc;
    if (any) return xfer_flush(ctx);
    return VO_OK;
}

extern "C" int vo_orb_detect_and_compute(vo_ctx* ctx, int slot, int nfeatures, int mask_mode, int min_disp16, int max_disp16,
                                         float* kp_xy, float* kp_size, float* kp_angle, float* kp_response, int32_t* kp_octave,
                                         uint8_t* desc, int cap, int* n_out)
{
    if (!ctx || slot < 0 || slot >= VO_NUM_SLOTS) return vo_fail(ctx, VO_E_ARG, "vo_orb_detect_and_compute: bad slot");
    if (nfeatures < 0 || nfeatures > ctx->max_kp) return vo_fail(ctx, VO_E_CAP, "nfeatures %d exceeds max_kp %d", nfeatures, ctx->max_kp);
    if (mask_mode != 0 && mask_mode != 1) return vo_fail(ctx, VO_E_ARG, "mask_mode must be 0 or 1");
    FrameSlot& f = ctx->slots[slot];
    if (!f.has_pair) return vo_fail(ctx, VO_E_STATE, "slot %d holds no image", slot);
    if (mask_mode == 1 && !f.has_disp) return vo_fail(ctx, VO_E_STATE, "slot %d holds no disparity for the fused mask", slot);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    const int params[4] = { nfeatures, mask_mode, min_disp16, max_disp16 };
    const bool same = !memcmp(params, f.kp_params, sizeof(params));
    const bool prefetched = f.kp_pending && same;
    f.kp_pending = false;
    int rc;
    if (f.has_kp && same) {
        // the slot already holds exactly this extraction (an earlier call, e.g. ahead of a pose step)
        if ((rc = slot_wait(ctx, f))) return rc;
        return download_kps(ctx, f, kp_xy, kp_size, kp_angle, kp_response, kp_octave, desc, cap, n_out);
    }
    if (prefetched) {
        // the look-ahead engine extracted these keypoints behind the SGBM: wait for it, nothing to launch
        VO_HIP(ctx, hipEventSynchronize(f.ready));
        if (f.pending && ctx->inflight > 0) ctx->inflight--;
        f.pending = false;
    } else {
        if ((rc = slot_wait(ctx, f))) return rc;
        if ((rc = orb_slot_enqueue(ctx, f, nfeatures, mask_mode, min_disp16, max_disp16))) return rc;
        VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(f.kp_params, params, sizeof(params));
    }
    // (the stream or the slot's `ready` event has been waited for: the SGBM run that produced the fused mask is over)
    if (mask_mode == 1 && (rc = slot_health(ctx, f, slot))) { f.has_kp = false; return rc; }
    if ((rc = orb_finish(ctx, &f))) return rc;
    return download_kps(ctx, f, kp_xy, kp_size, kp_angle, kp_response, kp_octave, desc, cap, n_out);
}

// keypoints + descriptors a slot already holds (no recomputation); any output may be NULL
extern "C" int vo_download_keypoints(vo_ctx* ctx, int slot, float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                                     int32_t* kp_octave, uint8_t* desc, int cap, int* n_out)
{
    if (!ctx || slot < 0 || slot >= VO_NUM_SLOTS) return vo_fail(ctx, VO_E_ARG, "vo_download_keypoints: bad slot");
    FrameSlot& f = ctx->slots[slot];
    if (!f.has_kp) return vo_fail(ctx, VO_E_STATE, "slot %d holds no keypoints", slot);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    { int rcw = slot_wait(ctx, f); if (rcw) return rcw; }
    return download_kps(ctx, f, kp_xy, kp_size, kp_angle, kp_response, kp_octave, desc, cap, n_out);
}

extern "C" int vo_orb_detect_and_compute_host(vo_ctx* ctx, const uint8_t* img, int w, int h, int stride, const uint8_t* mask,
                                              int mask_stride, int nfeatures, float* kp_xy, float* kp_size, float* kp_angle,
                                              float* kp_response, int32_t* kp_octave, uint8_t* desc, int cap, int* n_out)
{
    if (!ctx || !img || w <= 0 || h <= 0 || stride < w) return vo_fail(ctx, VO_E_ARG, "vo_orb_detect_and_compute_host: bad argument");
    if (w > ctx->max_w || h > ctx->max_h) return vo_fail(ctx, VO_E_CAP, "image %dx%d exceeds context", w, h);
    if (nfeatures < 0 || nfeatures > ctx->max_kp) return vo_fail(ctx, VO_E_CAP, "nfeatures %d exceeds max_kp %d", nfeatures, ctx->max_kp);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    FrameSlot& f = ctx->slots[VO_NUM_SLOTS];
    VO_HIP(ctx, hipMemcpy2DAsync(f.left, w, img, stride, w, h, hipMemcpyHostToDevice, ctx->stream));
    if (mask) VO_HIP(ctx, hipMemcpy2DAsync(ctx->host_mask_dev, w, mask, mask_stride, w, h, hipMemcpyHostToDevice, ctx->stream));
    int rc = orb_run(ctx, &f, f.left, w, w, h, nfeatures, mask ? 2 : 0, nullptr, 0, 0, 0, ctx->host_mask_dev, w);
    if (rc) return rc;
    return download_kps(ctx, f, kp_xy, kp_size, kp_angle, kp_response, kp_octave, desc, cap, n_out);
}
