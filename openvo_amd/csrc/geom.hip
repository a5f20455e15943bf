// 3-D lookup and pose arithmetic on gfx950.
//   cv2.reprojectImageTo3D                       [reference stereo_camera.py:52]
//   bilinear_interpolate_pixels / point_clouds   [reference stereo_odometer.py:50-79,162-175]
//   rigid_body_filter                            [reference stereo_odometer.py:82-105]
//   cv2.estimateAffine3D(force_rotation=True)    [reference stereo_odometer.py:190,204]
//   cv2.Rodrigues                                [reference stereo_odometer.py:212]
// Floating point here follows the reference operation by operation (float32 / float64 as numpy
// and OpenCV evaluate them); this file is compiled with -ffp-contract=off so no FMA is formed.
#include <math.h>
#include "vo_internal.h"

struct QMat { double q[16]; };

// one pixel of reprojectImageTo3D (OpenCV 4.x): homg = Q*[x y d 1] summed left to right in
// double; out_i = (float)homg_i, then out_i = (float)((double)out_i * (1/homg_3))
__device__ __forceinline__ void reproject_px(const QMat& Q, int x, int y, float d, float* out)
{
    const double v[4] = { (double)x, (double)y, (double)d, 1.0 };
    double hg[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) s = s + Q.q[i * 4 + k] * v[k];
        hg[i] = s;
    }
    const double ialpha = 1.0 / hg[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float f = (float)hg[i];
        out[i] = (float)((double)f * ialpha);
    }
}

__global__ void k_reproject(const float* __restrict__ disp, int w, int h, QMat Q, float* __restrict__ xyz)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    float p[3];
    reproject_px(Q, x, y, disp[(size_t)y * w + x], p);
    float* o = xyz + ((size_t)y * w + x) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

__global__ void k_reproject_disp16(const int16_t* __restrict__ disp16, int w, int h, QMat Q, float* __restrict__ xyz)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    float p[3];
    reproject_px(Q, x, y, (float)disp16[(size_t)y * w + x] / 16.0f, p);
    float* o = xyz + ((size_t)y * w + x) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

// openVO's bilinear lookup; TAP supplies the float3 at cropped coords (x, y)
template <typename TAP>
__device__ __forceinline__ void bilinear_one(const TAP& tap, int cw, int ch, float xf, float yf, float* out, uint8_t* status)
{
    const double x = (double)xf, y = (double)yf;
    const int fx = (int)x, fy = (int)y;
    const double rx = x - fx, ry = y - fy;
    const int tx[4] = { fx, fx, fx + 1, fx + 1 }, ty[4] = { fy, fy + 1, fy, fy + 1 };
    const double wt[4] = { (1 - rx) * (1 - ry), (1 - rx) * ry, rx * (1 - ry), rx * ry };
    float num[3] = { 0.f, 0.f, 0.f };
    double den = 0.0;
    int used = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (tx[k] >= cw || ty[k] >= ch) continue;
        float p[3];
        tap(tx[k], ty[k], p);
        if (isinf(p[0]) || isinf(p[1]) || isinf(p[2])) continue;
        const float wf = (float)wt[k];
        for (int c = 0; c < 3; c++) num[c] = num[c] + wf * p[c];
        den += wt[k];
        used++;
    }
    if (!used) {
        *status = 2;
        out[0] = out[1] = out[2] = __builtin_nanf("");
        return;
    }
    const float denf = (float)den;
    bool nan = false;
    for (int c = 0; c < 3; c++) {
        out[c] = num[c] / denf;
        nan |= isnan(out[c]);
    }
    *status = nan ? 1 : 0;
}

struct TapDisp {
    const int16_t* disp16; int w, x0, y0; QMat Q;
    __device__ void operator()(int x, int y, float* p) const
    {
        int gx = x + x0, gy = y + y0;
        reproject_px(Q, gx, gy, (float)disp16[(size_t)gy * w + gx] / 16.0f, p);
    }
};
struct TapImg {
    const float* img; int w;
    __device__ void operator()(int x, int y, float* p) const
    {
        const float* s = img + ((size_t)y * w + x) * 3;
        p[0] = s[0]; p[1] = s[1]; p[2] = s[2];
    }
};

__global__ void k_points3d(TapDisp tap, int cw, int ch, const float* __restrict__ xy, int n, float* __restrict__ xyz,
                           uint8_t* __restrict__ status)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bilinear_one(tap, cw, ch, xy[2 * i], xy[2 * i + 1], xyz + 3 * (size_t)i, status + i);
}
__global__ void k_bilinear_img(TapImg tap, int cw, int ch, const float* __restrict__ xy, int n, float* __restrict__ xyz,
                               uint8_t* __restrict__ status)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bilinear_one(tap, cw, ch, xy[2 * i], xy[2 * i + 1], xyz + 3 * (size_t)i, status + i);
}

static QMat make_q(const double* Q)
{
    QMat q;
    memcpy(q.q, Q, sizeof(q.q));
    return q;
}

int points3d_launch(vo_ctx* ctx, const int16_t* d_disp16, int w, int h, const float* d_xy, int n, float* d_xyz,
                    uint8_t* d_status)
{
    if (!ctx->has_Q) return vo_fail(ctx, VO_E_STATE, "vo_set_Q has not been called");
    int x0 = 0, y0 = 0, x1 = w, y1 = h;
    if (ctx->has_roi) { x0 = ctx->roi[0]; y0 = ctx->roi[1]; x1 = ctx->roi[2] < w ? ctx->roi[2] : w; y1 = ctx->roi[3] < h ? ctx->roi[3] : h; }
    if (n <= 0) return VO_OK;
    TapDisp tap{ d_disp16, w, x0, y0, make_q(ctx->Q) };
    hipLaunchKernelGGL(k_points3d, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, tap, x1 - x0, y1 - y0, d_xy, n, d_xyz, d_status);
    VO_CHECK_LAUNCH(ctx);
    return VO_OK;
}

extern "C" int vo_points3d_at(vo_ctx* ctx, int slot, const float* xy, int n, float* xyz_out, uint8_t* status_out)
{
    if (!ctx || slot < 0 || slot >= VO_NUM_SLOTS) return vo_fail(ctx, VO_E_ARG, "vo_points3d_at: bad slot");
    FrameSlot& f = ctx->slots[slot];
    if (!f.has_disp) return vo_fail(ctx, VO_E_STATE, "slot %d holds no disparity", slot);
    if (n < 0 || n > ctx->kp_cap) return vo_fail(ctx, VO_E_CAP, "n=%d exceeds keypoint capacity %d", n, ctx->kp_cap);
    if (n == 0) return VO_OK;
    if (!xy || !xyz_out || !status_out) return vo_fail(ctx, VO_E_ARG, "vo_points3d_at: null pointer");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    { int rcw = slot_wait(ctx, f); if (rcw) return rcw; }
    // keypoints must lie inside the cropped image, as img[floor_y, floor_x] requires
    int cw = (ctx->has_roi ? (ctx->roi[2] < f.w ? ctx->roi[2] : f.w) - ctx->roi[0] : f.w);
    int chh = (ctx->has_roi ? (ctx->roi[3] < f.h ? ctx->roi[3] : f.h) - ctx->roi[1] : f.h);
    for (int i = 0; i < n; i++)
        if (!(xy[2 * i] >= 0 && xy[2 * i + 1] >= 0 && (int)xy[2 * i] < cw && (int)xy[2 * i + 1] < chh))
            return vo_fail(ctx, VO_E_ARG, "keypoint %d (%g,%g) outside the %dx%d cropped image", i, xy[2 * i], xy[2 * i + 1], cw, chh);
    int rc = xfer_h2d(ctx, ctx->mw->xy_a, xy, (size_t)n * 8);
    if (rc) return rc;
    rc = points3d_launch(ctx, f.disp16, f.w, f.h, ctx->mw->xy_a, n, ctx->mw->pts_a, ctx->mw->st_a);
    if (rc) return rc;
    rc = xfer_d2h(ctx, xyz_out, ctx->mw->pts_a, (size_t)n * 12);
    if (!rc) rc = xfer_d2h(ctx, status_out, ctx->mw->st_a, (size_t)n);
    if (rc) return rc;
    if ((rc = xfer_flush(ctx))) return rc;
    return slot_health(ctx, f, slot);
}

static int ensure_ws(vo_ctx* ctx, float** p, size_t* have, size_t bytes)
{
    if (*have >= bytes) return VO_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *have = 0;
    VO_HIP(ctx, hipMalloc((void**)p, bytes));
    *have = bytes;
    return VO_OK;
}

extern "C" int vo_bilinear_at(vo_ctx* ctx, const float* img3d, int w, int h, const float* xy, int n, float* out,
                              uint8_t* status_out)
{
    if (!ctx || !img3d || !xy || !out || !status_out || w <= 0 || h <= 0) return vo_fail(ctx, VO_E_ARG, "vo_bilinear_at: bad argument");
    if (n < 0 || n > ctx->kp_cap) return vo_fail(ctx, VO_E_CAP, "n=%d exceeds keypoint capacity", n);
    if (n == 0) return VO_OK;
    for (int i = 0; i < n; i++)
        if (!(xy[2 * i] >= 0 && xy[2 * i + 1] >= 0 && (int)xy[2 * i] < w && (int)xy[2 * i + 1] < h))
            return vo_fail(ctx, VO_E_ARG, "point %d outside the image", i);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_ws(ctx, &ctx->img3_ws, &ctx->img3_ws_bytes, (size_t)w * h * 12);
    if (rc) return rc;
    VO_HIP(ctx, hipMemcpyAsync(ctx->img3_ws, img3d, (size_t)w * h * 12, hipMemcpyHostToDevice, ctx->stream));
    rc = xfer_h2d(ctx, ctx->mw->xy_a, xy, (size_t)n * 8);
    if (rc) return rc;
    TapImg tap{ ctx->img3_ws, w };
    hipLaunchKernelGGL(k_bilinear_img, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, tap, w, h, ctx->mw->xy_a, n, ctx->mw->pts_a, ctx->mw->st_a);
    VO_CHECK_LAUNCH(ctx);
    rc = xfer_d2h(ctx, out, ctx->mw->pts_a, (size_t)n * 12);
    if (!rc) rc = xfer_d2h(ctx, status_out, ctx->mw->st_a, (size_t)n);
    if (rc) return rc;
    return xfer_flush(ctx);
}

extern "C" int vo_reproject_to_3d(vo_ctx* ctx, const float* disp, int w, int h, const double* Q16, float* xyz)
{
    if (!ctx || !disp || !Q16 || !xyz || w <= 0 || h <= 0) return vo_fail(ctx, VO_E_ARG, "vo_reproject_to_3d: bad argument");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)w * h;
    int rc = ensure_ws(ctx, &ctx->img3_ws, &ctx->img3_ws_bytes, n * 16);
    if (rc) return rc;
    float* d_disp = ctx->img3_ws + n * 3;
    VO_HIP(ctx, hipMemcpyAsync(d_disp, disp, n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_reproject, dim3(div_up(w, 256), h), dim3(256), 0, ctx->stream, d_disp, w, h, make_q(Q16), ctx->img3_ws);
    VO_CHECK_LAUNCH(ctx);
    VO_HIP(ctx, hipMemcpyAsync(xyz, ctx->img3_ws, n * 12, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_download_xyz(vo_ctx* ctx, int slot, float* out)
{
    if (!ctx || slot < 0 || slot >= VO_NUM_SLOTS || !out) return vo_fail(ctx, VO_E_ARG, "vo_download_xyz: bad argument");
    FrameSlot& f = ctx->slots[slot];
    if (!f.has_disp) return vo_fail(ctx, VO_E_STATE, "slot %d holds no disparity", slot);
    if (!ctx->has_Q) return vo_fail(ctx, VO_E_STATE, "vo_set_Q has not been called");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    { int rcw = slot_wait(ctx, f); if (rcw) return rcw; }
    const size_t n = (size_t)f.w * f.h;
    int rc = ensure_ws(ctx, &ctx->img3_ws, &ctx->img3_ws_bytes, n * 12);
    if (rc) return rc;
    hipLaunchKernelGGL(k_reproject_disp16, dim3(div_up(f.w, 256), f.h), dim3(256), 0, ctx->stream, f.disp16, f.w, f.h,
                       make_q(ctx->Q), ctx->img3_ws);
    VO_CHECK_LAUNCH(ctx);
    VO_HIP(ctx, hipMemcpyAsync(out, ctx->img3_ws, n * 12, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return slot_health(ctx, f, slot);
}

// ---- ratio test + fused point_clouds ----------------------------------------------------
extern "C" int vo_ratio_filter(const int32_t* idx, const int32_t* dist, int nq, double ratio, int32_t* q_out,
                               int32_t* t_out, int* m_out)
{
    if (!idx || !dist || !q_out || !t_out || !m_out || nq < 0) return VO_E_ARG;
    int m = 0;
    for (int i = 0; i < nq; i++) {
        if (idx[2 * i + 1] < 0) return VO_E_ARG;  // the reference would raise IndexError (m[1] missing)
        double a = (double)(float)dist[2 * i], b = (double)(float)dist[2 * i + 1];
        if (a < ratio * b) { q_out[m] = i; t_out[m] = idx[2 * i]; m++; }
    }
    *m_out = m;
    return VO_OK;
}

// ratio test + ordered compaction on the device: one block (any multiple of 64 threads up to 1024).  Batches of up to eight
// elements per thread (element i0 + k * blockDim + thread): all their loads are in flight together, the keep flags become
// ballots, the per-(batch row, wave) counts go through LDS where the first wave turns them into exclusive offsets (one
// wave-wide prefix over 128 counts), and the survivors' coordinates are gathered and stored.  Three global round trips per
// 8192 queries (the one-element-per-thread loop this replaces paid three per 1024: 22 us at 8000 queries, now 6).
__global__ void __launch_bounds__(1024) k_ratio_compact(const int32_t* __restrict__ idx, const int32_t* __restrict__ dist, int nq, double ratio,
                                                        const float* __restrict__ xy_q, const float* __restrict__ xy_t, int32_t* __restrict__ q_out,
                                                        int32_t* __restrict__ t_out, float* __restrict__ xyq_out, float* __restrict__ xyt_out,
                                                        int32_t* __restrict__ m_out)
{
    constexpr int K = 8;
    __shared__ int s_cnt[K * 16 + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6, nt = blockDim.x;
    int base = 0;
    for (int i0 = 0; i0 < nq; i0 += K * nt) {
        int2 id[K], ds[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int i = i0 + k * nt + threadIdx.x;
            id[k] = make_int2(-1, -1); ds[k] = make_int2(0, 0);
            if (i < nq) { id[k] = ((const int2*)idx)[i]; ds[k] = ((const int2*)dist)[i]; }
        }
        unsigned long long bal[K];
        float2 pq[K], pt[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int i = i0 + k * nt + threadIdx.x;
            const double a = (double)(float)ds[k].x, b = (double)(float)ds[k].y;
            const bool keep = i < nq && id[k].y >= 0 && a < ratio * b;
            bal[k] = __ballot(keep);
            if (lane == 0) s_cnt[k * nw + wv] = __popcll(bal[k]);
            pq[k] = make_float2(0.f, 0.f); pt[k] = pq[k];
            if (keep) { pq[k] = ((const float2*)xy_q)[i]; pt[k] = ((const float2*)xy_t)[id[k].x]; }    // (in flight across the barriers)
        }
        __syncthreads();
        if (wv == 0) {                                   // exclusive prefix over the K * nw <= 128 counts, in (batch row, wave) order
            const int n = K * nw;
            const int c0 = 2 * lane < n ? s_cnt[2 * lane] : 0, c1 = 2 * lane + 1 < n ? s_cnt[2 * lane + 1] : 0;
            int inc = c0 + c1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
            const int ex = inc - (c0 + c1);
            if (2 * lane < n) s_cnt[2 * lane] = ex;
            if (2 * lane + 1 < n) s_cnt[2 * lane + 1] = ex + c0;
            if (lane == 63) s_cnt[K * 16] = inc;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++)
            if ((bal[k] >> lane) & 1ull) {
                const int pos = base + s_cnt[k * nw + wv] + __popcll(bal[k] & ((1ull << lane) - 1ull));
                q_out[pos] = i0 + k * nt + threadIdx.x; t_out[pos] = id[k].x;
                ((float2*)xyq_out)[pos] = pq[k]; ((float2*)xyt_out)[pos] = pt[k];
            }
        base += s_cnt[K * 16];
        __syncthreads();
    }
    if (threadIdx.x == 0) *m_out = base;
}

extern "C" int vo_point_clouds(vo_ctx* ctx, int slot_a, int slot_b, double ratio, int32_t* q_idx, int32_t* t_idx,
                               float* pts_a, float* pts_b, uint8_t* status_a, uint8_t* status_b, int cap, int* m_out)
{
    if (!ctx || slot_a < 0 || slot_a >= VO_NUM_SLOTS || slot_b < 0 || slot_b >= VO_NUM_SLOTS || !m_out)
        return vo_fail(ctx, VO_E_ARG, "vo_point_clouds: bad argument");
    FrameSlot& a = ctx->slots[slot_a];
    FrameSlot& b = ctx->slots[slot_b];
    if (!a.has_kp || !b.has_kp || !a.has_disp || !b.has_disp) return vo_fail(ctx, VO_E_STATE, "slots need disparity and keypoints");
    { int rcw = slot_wait(ctx, a); if (!rcw) rcw = slot_wait(ctx, b); if (rcw) return rcw; }
    *m_out = 0;
    if (a.n_kp == 0) return VO_OK;
    if (b.n_kp < 2) return vo_fail(ctx, VO_E_ARG, "train set has fewer than 2 descriptors (reference raises IndexError)");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    {
        StageTimer t(ctx, VO_T_MATCH);
        rc = match_knn2(ctx, a.desc, a.n_kp, b.desc, b.n_kp, ctx->mw->m_idx, ctx->mw->m_dist);
        if (rc) return rc;
        hipLaunchKernelGGL(k_ratio_compact, dim3(1), dim3(a.n_kp > 512 ? 1024 : 256), 0, ctx->stream, ctx->mw->m_idx, ctx->mw->m_dist, a.n_kp, ratio, a.kp_xy,
                           b.kp_xy, ctx->mw->mq_idx, ctx->mw->mt_idx, ctx->mw->xy_a, ctx->mw->xy_b, ctx->mw->m_count);
        VO_CHECK_LAUNCH(ctx);
        // 3-D lookups for every query slot position (n_kp upper bound); only the first M are meaningful
        VO_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctx->mw->m_count, 4, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if ((rc = slot_health(ctx, a, slot_a)) || (rc = slot_health(ctx, b, slot_b))) return rc;   // (both slots' producers are behind that synchronisation)
    const int m = *(int32_t*)ctx->pinned;
    *m_out = m;
    if (m == 0) return VO_OK;
    {
        StageTimer t(ctx, VO_T_POSE);
        rc = points3d_launch(ctx, a.disp16, a.w, a.h, ctx->mw->xy_a, m, ctx->mw->pts_a, ctx->mw->st_a);
        if (rc) return rc;
        rc = points3d_launch(ctx, b.disp16, b.w, b.h, ctx->mw->xy_b, m, ctx->mw->pts_b, ctx->mw->st_b);
        if (rc) return rc;
    }
    if (m > cap && (q_idx || t_idx || pts_a || pts_b || status_a || status_b))
        return vo_fail(ctx, VO_E_CAP, "%d matches exceed output capacity %d", m, cap);
    rc = VO_OK;
    if (q_idx && !rc) rc = xfer_d2h(ctx, q_idx, ctx->mw->mq_idx, (size_t)m * 4);
    if (t_idx && !rc) rc = xfer_d2h(ctx, t_idx, ctx->mw->mt_idx, (size_t)m * 4);
    if (pts_a && !rc) rc = xfer_d2h(ctx, pts_a, ctx->mw->pts_a, (size_t)m * 12);
    if (pts_b && !rc) rc = xfer_d2h(ctx, pts_b, ctx->mw->pts_b, (size_t)m * 12);
    if (status_a && !rc) rc = xfer_d2h(ctx, status_a, ctx->mw->st_a, (size_t)m);
    if (status_b && !rc) rc = xfer_d2h(ctx, status_b, ctx->mw->st_b, (size_t)m);
    if (rc) return rc;
    return xfer_flush(ctx);
}

// ---- Umeyama: two-pass float64 reductions on the device, 3x3 SVD on the host -------------
__device__ __forceinline__ double wave_sum_f64(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// out[0..2] = sum src, out[3..5] = sum dst
__global__ void k_umeyama_sums(const float* __restrict__ src, const float* __restrict__ dst, int m, double* __restrict__ out)
{
    __shared__ double sh[6][16];
    double a[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = threadIdx.x; i < m; i += blockDim.x)
        for (int c = 0; c < 3; c++) { a[c] += (double)src[3 * i + c]; a[3 + c] += (double)dst[3 * i + c]; }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = 0; c < 6; c++) { double s = wave_sum_f64(a[c]); if (lane == 0) sh[c][wv] = s; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double s = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) s += sh[threadIdx.x][k];
        out[threadIdx.x] = s;
    }
}
// out[6..14] = sum (dst-md)(src-ms)^T, out[15] = sum |src-ms|^2; means = out[0..5]/m
__global__ void k_umeyama_cov(const float* __restrict__ src, const float* __restrict__ dst, int m, double* __restrict__ out)
{
    __shared__ double sh[10][16];
    const double inv = 1.0 / m;
    double ms[3], md[3];
    for (int c = 0; c < 3; c++) { ms[c] = out[c] * inv; md[c] = out[3 + c] * inv; }
    double a[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        double s[3], d[3];
        for (int c = 0; c < 3; c++) { s[c] = (double)src[3 * i + c] - ms[c]; d[c] = (double)dst[3 * i + c] - md[c]; }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) a[r * 3 + c] += d[r] * s[c];
        a[9] += s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = 0; c < 10; c++) { double s = wave_sum_f64(a[c]); if (lane == 0) sh[c][wv] = s; }
    __syncthreads();
    if (threadIdx.x < 10) {
        double s = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) s += sh[threadIdx.x][k];
        out[6 + threadIdx.x] = s;
    }
}

static void cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// one-sided Jacobi SVD of a 3x3 matrix (host); singular values descending
void host_svd3(const double* A, double* U, double* w, double* Vt)
{
    double G[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    memcpy(G, A, sizeof(G));
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; i++) {
                    al += G[i * 3 + p] * G[i * 3 + p];
                    be += G[i * 3 + q] * G[i * 3 + q];
                    ga += G[i * 3 + p] * G[i * 3 + q];
                }
                if (fabs(ga) <= 1e-300 || fabs(ga) <= 2.2204460492503131e-16 * sqrt(al * be)) continue;
                rotated = true;
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
    for (int j = 0; j < 3; j++) sv[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
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
        for (int i = 0; i < 3; i++) { U[i * 3 + j] = Uc[j][i]; Vt[j * 3 + i] = Vc[j][i]; }
}

static double det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// finish Umeyama from the device sums (red[0..15])
static int umeyama_finish(vo_ctx* ctx, const double* red, int m, int force_rotation, double* T, double* scale_out)
{
    const double inv = 1.0 / m;
    double ms[3], md[3], cov[9];
    for (int c = 0; c < 3; c++) { ms[c] = red[c] * inv; md[c] = red[3 + c] * inv; }
    for (int k = 0; k < 9; k++) cov[k] = red[6 + k] * inv;
    const double var_from = red[15];
    double U[9], w[3], Vt[9];
    host_svd3(cov, U, w, Vt);
    // NaN inputs propagate to T (the reference then reports skip_cause "nan")
    int nz = (w[0] != 0) + (w[1] != 0) + (w[2] != 0);
    if (nz < 2) return vo_fail(ctx, VO_E_NUMERIC, "Points cannot be colinear");
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
    return VO_OK;
}

extern "C" int vo_umeyama(vo_ctx* ctx, const float* src, const float* dst, int m, int force_rotation, double* T12,
                          double* scale_out)
{
    if (!ctx || !src || !dst || !T12) return vo_fail(ctx, VO_E_ARG, "vo_umeyama: bad argument");
    if (m < 3) return vo_fail(ctx, VO_E_NUMERIC, "Umeyama algorithm needs at least 3 points for affine transformation estimation.");
    if (m > ctx->kp_cap) return vo_fail(ctx, VO_E_CAP, "m=%d exceeds capacity %d", m, ctx->kp_cap);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    StageTimer t(ctx, VO_T_POSE);
    int rc = xfer_h2d(ctx, ctx->mw->pts_a, src, (size_t)m * 12);
    if (!rc) rc = xfer_h2d(ctx, ctx->mw->pts_b, dst, (size_t)m * 12);
    if (rc) return rc;
    hipLaunchKernelGGL(k_umeyama_sums, dim3(1), dim3(256), 0, ctx->stream, ctx->mw->pts_a, ctx->mw->pts_b, m, ctx->red);
    hipLaunchKernelGGL(k_umeyama_cov, dim3(1), dim3(256), 0, ctx->stream, ctx->mw->pts_a, ctx->mw->pts_b, m, ctx->red);
    VO_CHECK_LAUNCH(ctx);
    VO_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctx->red, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = xfer_flush(ctx))) return rc;
    return umeyama_finish(ctx, (const double*)ctx->pinned, m, force_rotation, T12, scale_out);
}

extern "C" int vo_rodrigues(const double* Rin, double* r)
{
    if (!Rin || !r) return VO_E_ARG;
    double U[9], w[3], Vt[9], R[9];
    host_svd3(Rin, U, w, Vt);
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
        if (c > 0) { r[0] = r[1] = r[2] = 0; return VO_OK; }
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
    return VO_OK;
}

// ---- rigid_body_filter: consistency matrix + greedy clique, one workgroup -----------------
// cons[i][j] = | ||cur_i-cur_j|| - ||prev_i-prev_j|| | < thr, all in float32 as numpy evaluates it
__global__ void k_rigid_cons(const float* __restrict__ prev, const float* __restrict__ cur, int m, float thr,
                             uint8_t* __restrict__ cons, int* __restrict__ ncons)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= m) return;
    float ax = cur[3 * i] - cur[3 * j], ay = cur[3 * i + 1] - cur[3 * j + 1], az = cur[3 * i + 2] - cur[3 * j + 2];
    float bx = prev[3 * i] - prev[3 * j], by = prev[3 * i + 1] - prev[3 * j + 1], bz = prev[3 * i + 2] - prev[3 * j + 2];
    float na = sqrtf((ax * ax + ay * ay) + az * az), nb = sqrtf((bx * bx + by * by) + bz * bz);
    uint8_t c = fabsf(na - nb) < thr;
    cons[(size_t)i * m + j] = c;
    if (c) atomicAdd(&ncons[j], 1);  // column sums
}

// greedy clique on ONE wave (no block barriers): lanes stride over the m candidates; the
// compatibility test (cons[i,:] . clique >= |clique|) is kept incrementally -- adding `sel` to the
// clique adds column `sel` of the consistency matrix to the running dot products.
__global__ void __launch_bounds__(64) k_rigid_clique(const uint8_t* __restrict__ cons, const int* __restrict__ ncons, int m,
                                                     long long* __restrict__ mask_out)
{
    extern __shared__ int s_mem[];
    int* clique = s_mem;
    int* compat = s_mem + m;
    int* dots = s_mem + 2 * m;
    const int lane = threadIdx.x;
    // wave argmax with numpy's first-max tie rule: key = value << 32 | (0x7fffffff - index)
    auto wave_argmax = [&](long long key) -> int {
        for (int o = 32; o > 0; o >>= 1) { long long other = __shfl_xor(key, o, 64); key = other > key ? other : key; }
        return 0x7fffffff - (int)(key & 0x7fffffffLL);
    };
    long long key = -0x7fffffffffffffffLL;
    for (int j = lane; j < m; j += 64) {
        long long k = ((long long)ncons[j] << 32) | (long long)(0x7fffffff - j);
        key = k > key ? k : key;
    }
    const int seed = wave_argmax(key);          // argmax(num_consistent)
    for (int j = lane; j < m; j += 64) {
        clique[j] = j == seed;
        compat[j] = cons[(size_t)seed * m + j];
        dots[j] = cons[(size_t)seed * m + j];   // the matrix is symmetric: read rows (coalesced), not columns
    }
    int csize = 1;
    for (int it = 0; it < m; it++) {
        // candidates = compatible - clique; stop when they sum to 0; selected = argmax(ncons * cand)
        int psum = 0;
        key = -0x7fffffffffffffffLL;
        for (int j = lane; j < m; j += 64) {
            const int cand = compat[j] - clique[j];
            psum += cand;
            const long long v = (long long)ncons[j] * cand;
            const long long k = (v << 32) | (long long)(0x7fffffff - j);
            key = k > key ? k : key;
        }
        for (int o = 32; o > 0; o >>= 1) psum += __shfl_xor(psum, o, 64);
        if (psum == 0) break;
        const int sel = wave_argmax(key);
        const bool fresh = clique[sel] == 0;   // wave-uniform read
        __builtin_amdgcn_wave_barrier();
        if (fresh) {
            csize++;
            for (int j = lane; j < m; j += 64) dots[j] += cons[(size_t)sel * m + j];
            if (lane == 0) clique[sel] = 1;
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < m; j += 64) compat[j] = dots[j] >= csize;
    }
    for (int j = lane; j < m; j += 64) mask_out[j] = clique[j];
}

extern "C" int vo_rigid_clique(vo_ctx* ctx, const float* prev, const float* cur, int m, double thr, int64_t* mask_out)
{
    if (!ctx || !prev || !cur || !mask_out || m < 0) return vo_fail(ctx, VO_E_ARG, "vo_rigid_clique: bad argument");
    if (m == 0) return VO_OK;
    if (m > ctx->kp_cap || (size_t)m * 12 > 150 * 1024) return vo_fail(ctx, VO_E_CAP, "m=%d exceeds capacity", m);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t need = (size_t)m * m + (size_t)m * 4 * 3 + (size_t)m * 8 + 1024;
    if (ctx->mw->clique_ws_bytes < need) {
        if (ctx->mw->clique_ws) (void)hipFree(ctx->mw->clique_ws);
        ctx->mw->clique_ws = nullptr; ctx->mw->clique_ws_bytes = 0;
        VO_HIP(ctx, hipMalloc((void**)&ctx->mw->clique_ws, need));
        ctx->mw->clique_ws_bytes = need;
    }
    StageTimer t(ctx, VO_T_POSE);
    // carve: mask(8m) | ncons(4m) | clique(4m) | compat(4m) | cons(m*m)
    long long* d_mask = (long long*)ctx->mw->clique_ws;
    int* d_ncons = (int*)(ctx->mw->clique_ws + (size_t)m * 8);
    int* d_clique = d_ncons + m;
    int* d_compat = d_clique + m;
    uint8_t* d_cons = (uint8_t*)(d_compat + m);
    int rc = xfer_h2d(ctx, ctx->mw->pts_a, prev, (size_t)m * 12);
    if (!rc) rc = xfer_h2d(ctx, ctx->mw->pts_b, cur, (size_t)m * 12);
    if (rc) return rc;
    VO_HIP(ctx, hipMemsetAsync(d_ncons, 0, (size_t)m * 4, ctx->stream));
    hipLaunchKernelGGL(k_rigid_cons, dim3(div_up(m, 256), m), dim3(256), 0, ctx->stream, ctx->mw->pts_a, ctx->mw->pts_b, m, (float)thr, d_cons, d_ncons);
    hipLaunchKernelGGL(k_rigid_clique, dim3(1), dim3(64), (size_t)m * 12, ctx->stream, d_cons, d_ncons, m, d_mask);
    VO_CHECK_LAUNCH(ctx);
    if ((rc = xfer_d2h(ctx, mask_out, d_mask, (size_t)m * 8))) return rc;
    return xfer_flush(ctx);
}

// =========================================================================================
// Fused pair step: point_clouds + point_cloud_transform of two device-resident frames
// [reference stereo_odometer.py:162-175,177-205] with ONE host synchronisation.  Every count
// (matches M, survivors of the clique filter n1, of the outlier pass n2) stays on the device;
// kernels are launched for the upper bound and read the live count.  The host only applies the
// motion gates [:207-221] to the returned transform.
// =========================================================================================
struct PoseOut {
    int M, n1, n2, flags;   // flags bit0: a 3-D lookup had no usable tap (ZeroDivisionError); bit1: NaN residual
    int rc1, rc2, pad0, pad1;  // Umeyama status of the first / final fit: 0 ok, -1 < 3 points, -2 colinear
    double T1[12], s1, T2[12], s2;
};

// consistency matrix as bit rows: bits[i][w] bit b = | ||cur_i-cur_j|| - ||prev_i-prev_j|| | < thr, j = 64 w + b
// one 64-lane block per row i of the consistency matrix: the row's bit words (ballots), its popcount, and -- up to 512
// matches -- one byte per lane holding the row's bits for elements lane + 64 k (what the greedy loop in k_pose_solve reads)
__global__ void __launch_bounds__(64) k_pose_cons_bits(const float* __restrict__ prev, const float* __restrict__ cur, const int* __restrict__ m_dev,
                                                       float thr, unsigned long long* __restrict__ bits, int words_cap, int* __restrict__ ncons,
                                                       uint8_t* __restrict__ lanebytes)
{
    const int m = *m_dev;
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= m) return;
    const int words = (m + 63) >> 6;
    const float cix = cur[3 * i], ciy = cur[3 * i + 1], ciz = cur[3 * i + 2];
    const float pix = prev[3 * i], piy = prev[3 * i + 1], piz = prev[3 * i + 2];
    unsigned byte = 0;
    int cnt = 0;
    for (int k = 0; k < words; k++) {
        const int j = 64 * k + lane;
        bool c = false;
        if (j < m) {
            float ax = cix - cur[3 * j], ay = ciy - cur[3 * j + 1], az = ciz - cur[3 * j + 2];
            float bx = pix - prev[3 * j], by = piy - prev[3 * j + 1], bz = piz - prev[3 * j + 2];
            float na = sqrtf((ax * ax + ay * ay) + az * az), nb = sqrtf((bx * bx + by * by) + bz * bz);
            c = fabsf(na - nb) < thr;
        }
        const unsigned long long bal = __ballot(c);
        if (lane == 0) bits[(size_t)i * words_cap + k] = bal;
        cnt += __popcll(bal);
        byte |= (unsigned)c << (k & 7);
    }
    if (words <= 8) lanebytes[(size_t)i * 64 + lane] = (uint8_t)byte;
    if (lane == 0) ncons[i] = cnt;
}

#ifdef VO_POSE_STAMPS
__device__ long long g_pose_stamps[32];
#define STAMP(k) do { if (threadIdx.x == 0) g_pose_stamps[k] = (long long)wall_clock64(); } while (0)
__device__ void g_stamp_aux(int a, int b) { g_pose_stamps[20] = a; g_pose_stamps[21] = b; }
extern "C" int vo_debug_pose_stamps(long long* out32) { return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_pose_stamps), 32 * 8); }
#else
#define STAMP(k) do {} while (0)
__device__ __forceinline__ void g_stamp_aux(int, int) {}
#endif

__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#define DPP_MAX(ctrl, rmask) v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false))
    DPP_MAX(0x111, 0xf); DPP_MAX(0x112, 0xf); DPP_MAX(0x114, 0xf); DPP_MAX(0x118, 0xf);
    DPP_MAX(0x142, 0xa); DPP_MAX(0x143, 0xc);
#undef DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_sum_i32_dpp(int v)
{
    // inclusive DPP scan inside each row, then carry rows: lanes without a source add 0
#define DPP_ADD(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false)
    DPP_ADD(0x111, 0xf); DPP_ADD(0x112, 0xf); DPP_ADD(0x114, 0xf); DPP_ADD(0x118, 0xf);
    DPP_ADD(0x142, 0xa); DPP_ADD(0x143, 0xc);
#undef DPP_ADD
    return __builtin_amdgcn_readlane(v, 63);
}

// float64 wave sum on DPP moves (two per step); every lane of the result is NOT valid -- lane 63 is, and it is broadcast
__device__ __forceinline__ double wave_sum_f64_dpp(double v)
{
#define DPP_ADD64(ctrl, rmask)                                                                             \
    {                                                                                                      \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, false);         \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, false);         \
        v += __hiloint2double(hi, lo);                                                                     \
    }
    DPP_ADD64(0x111, 0xf) DPP_ADD64(0x112, 0xf) DPP_ADD64(0x114, 0xf) DPP_ADD64(0x118, 0xf)
    DPP_ADD64(0x142, 0xa) DPP_ADD64(0x143, 0xc)
#undef DPP_ADD64
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// register-resident greedy clique (m <= 64 * KREG <= 512), one wave: lane holds elements j = lane + 64 k; the clique and
// the set of nodes adjacent to every member are K-bit masks per lane.  A node stays in that set exactly when it is
// adjacent to the member just added, so one byte per (row, lane) -- the row's bits for this lane's K elements, laid out
// by k_pose_cons_bits -- is all an iteration reads: cp &= lanebits[sel][lane].
template <int KREG>
__device__ __forceinline__ void clique_fast(const uint8_t* lanebits, const int* __restrict__ ncons, int m, int lane, int* clique)
{
    unsigned keyk[KREG];
    unsigned vmask = 0, key = 0;
#pragma unroll
    for (int k = 0; k < KREG; k++) {
        const int j = lane + 64 * k;
        const bool valid = j < m;
        vmask |= (unsigned)valid << k;
        keyk[k] = valid ? (((unsigned)(ncons[j] + m) << 16) | (unsigned)(0xFFFF - j)) : 0u;   // most consistent, then lowest index
        key = max(key, keyk[k]);
    }
    const int seed = 0xFFFF - (int)(wave_max_u32(key) & 0xFFFFu);
    unsigned cp = (unsigned)lanebits[seed * 64 + lane] & vmask;
    unsigned cl = lane == (seed & 63) ? 1u << (seed >> 6) : 0u;
    STAMP(10);
    int csize = 1;
    for (int it = 0; it < m; it++) {
        const unsigned cand = cp & ~cl;               // adjacent to every member and not a member (unit diagonal: members stay in cp)
        unsigned t[KREG];
#pragma unroll
        for (int k = 0; k < KREG; k++) t[k] = keyk[k] & (unsigned)__builtin_amdgcn_sbfe((int)cand, k, 1);
#pragma unroll
        for (int w = KREG / 2; w > 0; w >>= 1)
#pragma unroll
            for (int k = 0; k < w; k++) t[k] = max(t[k], t[k + w]);
        const unsigned kmax = wave_max_u32(t[0]);
        if (kmax == 0u) break;                        // (a candidate's key is at least (m + 1) << 16)
        const int sel = 0xFFFF - (int)(kmax & 0xFFFFu);
        cp &= (unsigned)lanebits[sel * 64 + lane];
        cl |= lane == (sel & 63) ? 1u << (sel >> 6) : 0u;
        csize++;
    }
    STAMP(11);
    if (threadIdx.x == 0) g_stamp_aux(csize, m);
#pragma unroll
    for (int k = 0; k < KREG; k++)
        if ((vmask >> k) & 1u) clique[lane + 64 * k] = (int)((cl >> k) & 1u);
}

// greedy clique on one wave over the bit matrix, then ordered compaction of the kept pairs.
// 32-bit argmax keys ((value + m) << 16 | (0xFFFF - index)) need m < 32768.
__device__ void pose_clique_wave(int* s_mem, const unsigned long long* __restrict__ bits, int words_cap,
                                 const int* __restrict__ ncons, const int* __restrict__ m_dev, int use_filter,
                                 const float* __restrict__ pa, const float* __restrict__ pb,
                                 float* __restrict__ qa, float* __restrict__ qb, int* __restrict__ n1_out,
                                 int lds_m_cap, size_t lds_bits_cap, bool fast)
{
    // s_mem: 4 * lds_m_cap ints, then lds_bits_cap bytes for the bit matrix (fast: the per-lane row bytes); called by
    // ONE wave (threads 0..63)
    const int m = *m_dev;
    int* clique = s_mem;
    int* compat = s_mem + lds_m_cap;
    int* dots = s_mem + 2 * lds_m_cap;
    int* nc = s_mem + 3 * lds_m_cap;
    const int lane = threadIdx.x;
    if (fast) {
        const uint8_t* lanebits = (const uint8_t*)(s_mem + 4 * lds_m_cap);
        if (m <= 128) clique_fast<2>(lanebits, ncons, m, lane, clique);
        else if (m <= 256) clique_fast<4>(lanebits, ncons, m, lane, clique);
        else clique_fast<8>(lanebits, ncons, m, lane, clique);
        __builtin_amdgcn_wave_barrier();
    } else if (use_filter && m > 0) {
        for (int j = lane; j < m; j += 64) nc[j] = ncons[j];
        // the greedy loop is a chain of dependent row reads: keep the bit matrix in LDS when it fits
        const int mw = (m + 63) >> 6;
        const unsigned long long* rows = bits;
        int rstride = words_cap;
        if (lds_bits_cap >= (size_t)m * mw * 8) {
            unsigned long long* lb = (unsigned long long*)(s_mem + 4 * lds_m_cap);
            for (int k = lane; k < m * mw; k += 64) lb[k] = bits[(size_t)(k / mw) * words_cap + (k % mw)];
            __builtin_amdgcn_wave_barrier();
            rows = lb; rstride = mw;
        }
        auto bit = [&](int row, int j) -> int { return (int)((rows[(size_t)row * rstride + (j >> 6)] >> (j & 63)) & 1ull); };
        unsigned key = 0;
        for (int j = lane; j < m; j += 64) key = max(key, ((unsigned)(nc[j] + m) << 16) | (unsigned)(0xFFFF - j));
        const int seed = 0xFFFF - (int)(wave_max_u32(key) & 0xFFFFu);
        for (int j = lane; j < m; j += 64) {
            const int b = bit(seed, j);
            clique[j] = j == seed; compat[j] = b; dots[j] = b;
        }
        int csize = 1;
        for (int it = 0; it < m; it++) {
            int psum = 0;
            key = 0;
            for (int j = lane; j < m; j += 64) {
                const int cand = compat[j] - clique[j];
                psum += cand;
                key = max(key, ((unsigned)(nc[j] * cand + m) << 16) | (unsigned)(0xFFFF - j));
            }
            psum = wave_sum_i32_dpp(psum);
            if (psum == 0) break;
            const int sel = 0xFFFF - (int)(wave_max_u32(key) & 0xFFFFu);
            const bool fresh = clique[sel] == 0;
            __builtin_amdgcn_wave_barrier();
            if (fresh) {
                csize++;
                for (int j = lane; j < m; j += 64) dots[j] += bit(sel, j);
                if (lane == 0) clique[sel] = 1;
            }
            __builtin_amdgcn_wave_barrier();
            for (int j = lane; j < m; j += 64) compat[j] = dots[j] >= csize;
        }
    }
    // ordered compaction
    int base = 0;
    for (int j0 = 0; j0 < m; j0 += 64) {
        const int j = j0 + lane;
        const bool keep = j < m && (!use_filter || clique[j] > 0);
        const unsigned long long bal = __ballot(keep);
        if (keep) {
            const int o = base + __popcll(bal & ((1ull << lane) - 1ull));
            for (int c = 0; c < 3; c++) { qa[3 * o + c] = pa[3 * j + c]; qb[3 * o + c] = pb[3 * j + c]; }
        }
        base += __popcll(bal);
    }
    if (lane == 0) *n1_out = base;
}

__device__ void dev_cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// the same one-sided Jacobi SVD as host_svd3, on one device thread
// one Jacobi rotation of columns P, Q (compile-time indices: G and V stay in registers)
template <int P, int Q>
__device__ __forceinline__ bool svd3_rotate(double* G, double* V)
{
    double al = 0, be = 0, ga = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        al += G[i * 3 + P] * G[i * 3 + P];
        be += G[i * 3 + Q] * G[i * 3 + Q];
        ga += G[i * 3 + P] * G[i * 3 + Q];
    }
    if (fabs(ga) <= 1e-300 || fabs(ga) <= 2.2204460492503131e-16 * sqrt(al * be)) return false;
    const double zeta = (be - al) / (2.0 * ga);
    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double gp = G[i * 3 + P], gq = G[i * 3 + Q];
        G[i * 3 + P] = c * gp - s * gq;
        G[i * 3 + Q] = s * gp + c * gq;
        const double vp = V[i * 3 + P], vq = V[i * 3 + Q];
        V[i * 3 + P] = c * vp - s * vq;
        V[i * 3 + Q] = s * vp + c * vq;
    }
    return true;
}
__device__ __forceinline__ double pick3(double a, double b, double c, int o) { return o == 0 ? a : (o == 1 ? b : c); }

// one-sided Jacobi SVD of a 3x3 matrix; every index is a compile-time constant or a select, so nothing lives in scratch
__device__ void dev_svd3(const double* A, double* U, double* w, double* Vt)
{
    double G[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
#pragma unroll
    for (int k = 0; k < 9; k++) G[k] = A[k];
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = svd3_rotate<0, 1>(G, V);
        rotated |= svd3_rotate<0, 2>(G, V);
        rotated |= svd3_rotate<1, 2>(G, V);
        if (!rotated) break;
    }
    double sv[3];
#pragma unroll
    for (int j = 0; j < 3; j++) sv[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
    int o0 = 0, o1 = 1, o2 = 2;                    // descending order of the singular values (same swaps as a selection sort)
    if (pick3(sv[0], sv[1], sv[2], o1) > pick3(sv[0], sv[1], sv[2], o0)) { const int t = o0; o0 = o1; o1 = t; }
    if (pick3(sv[0], sv[1], sv[2], o2) > pick3(sv[0], sv[1], sv[2], o0)) { const int t = o0; o0 = o2; o2 = t; }
    if (pick3(sv[0], sv[1], sv[2], o2) > pick3(sv[0], sv[1], sv[2], o1)) { const int t = o1; o1 = o2; o2 = t; }
    const int ord[3] = { o0, o1, o2 };
    double Uc[3][3], Vc[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int o = ord[j];
        const double svo = pick3(sv[0], sv[1], sv[2], o);
        w[j] = svo;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            Vc[j][i] = pick3(V[i * 3], V[i * 3 + 1], V[i * 3 + 2], o);
            Uc[j][i] = svo > 0 ? pick3(G[i * 3], G[i * 3 + 1], G[i * 3 + 2], o) / svo : 0.0;
        }
    }
    const double tiny = w[0] * 1e-300 + 1e-300;
    if (w[1] <= tiny) {
        double a[3] = { 1, 0, 0 };
        if (fabs(Uc[0][0]) > 0.9) { a[0] = 0; a[1] = 1; }
        dev_cross3(Uc[0], a, Uc[1]);
        double nn = sqrt(Uc[1][0] * Uc[1][0] + Uc[1][1] * Uc[1][1] + Uc[1][2] * Uc[1][2]);
#pragma unroll
        for (int i = 0; i < 3; i++) Uc[1][i] /= nn;
    }
    if (w[2] <= tiny || w[2] <= 1e-14 * w[0]) {
        dev_cross3(Uc[0], Uc[1], Uc[2]);
        double nn = sqrt(Uc[2][0] * Uc[2][0] + Uc[2][1] * Uc[2][1] + Uc[2][2] * Uc[2][2]);
        if (nn > 0) {
#pragma unroll
            for (int i = 0; i < 3; i++) Uc[2][i] /= nn;
        }
    }
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) { U[i * 3 + j] = Uc[j][i]; Vt[j * 3 + i] = Vc[j][i]; }
}

__device__ double dev_det3(const double* m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}


// block-wide float64 sums of up to NV values per thread; result valid in thread 0's copy via sh
template <int NV>
__device__ void block_sum_f64(double* a, double (*sh)[16])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int c = 0; c < NV; c++) { double s = wave_sum_f64(a[c]); if (lane == 0) sh[c][wv] = s; }
    __syncthreads();
    for (int c = 0; c < NV; c++) {
        double s = 0;
        for (int k = 0; k < nw; k++) s += sh[c][k];
        a[c] = s;
    }
    __syncthreads();
}

// Umeyama fit of n (device count) pairs in one block: out T[12], scale, rc
__device__ void dev_umeyama_block(const float* __restrict__ src, const float* __restrict__ dst, int n, int min_n,
                                  double* T, double* scale, int* rc, double (*sh)[16])
{
    if (n < min_n || n < 3) { if (threadIdx.x == 0) *rc = n < 3 ? -1 : 1; return; }   // 1 = not attempted
    // up to 1024 pairs the sums are one wave's work (DPP reductions, no block barrier); the other waves wait at the end
    const bool one_wave = n <= 1024;
    if (one_wave && threadIdx.x >= 64) { __syncthreads(); return; }
    const int stride = one_wave ? 64 : (int)blockDim.x;
    double a[10];
#pragma unroll
    for (int c = 0; c < 10; c++) a[c] = 0;
    for (int i = threadIdx.x; i < n; i += stride)
#pragma unroll
        for (int c = 0; c < 3; c++) { a[c] += (double)src[3 * i + c]; a[3 + c] += (double)dst[3 * i + c]; }
    if (one_wave) {
#pragma unroll
        for (int c = 0; c < 6; c++) a[c] = wave_sum_f64_dpp(a[c]);
    } else
        block_sum_f64<6>(a, sh);
    STAMP(12);
    const double inv = 1.0 / n;
    double ms[3], md[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { ms[c] = a[c] * inv; md[c] = a[3 + c] * inv; }
#pragma unroll
    for (int c = 0; c < 10; c++) a[c] = 0;
    for (int i = threadIdx.x; i < n; i += stride) {
        double s[3], d[3];
#pragma unroll
        for (int c = 0; c < 3; c++) { s[c] = (double)src[3 * i + c] - ms[c]; d[c] = (double)dst[3 * i + c] - md[c]; }
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) a[r * 3 + c] += d[r] * s[c];
        a[9] += s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
    }
    if (one_wave) {
#pragma unroll
        for (int c = 0; c < 10; c++) a[c] = wave_sum_f64_dpp(a[c]);
    } else
        block_sum_f64<10>(a, sh);
    STAMP(13);
    if (threadIdx.x == 0) {
        double cov[9], U[9], w[3], Vt[9];
        for (int k = 0; k < 9; k++) cov[k] = a[k] * inv;
        dev_svd3(cov, U, w, Vt);
        STAMP(14);
        const int nz = (w[0] != 0) + (w[1] != 0) + (w[2] != 0);
        if (nz < 2) { *rc = -2; }
        else {
            double S[3] = { 1, 1, 1 };
            if (dev_det3(U) * dev_det3(Vt) < 0) S[2] = -1;
            double R[9];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    double acc = 0;
                    for (int k = 0; k < 3; k++) acc += U[r * 3 + k] * S[k] * Vt[k * 3 + c];
                    R[r * 3 + c] = acc;
                }
            const double sc = (w[0] * S[0] + w[1] * S[1] + w[2] * S[2]) * ((double)n / a[9]);
            for (int r = 0; r < 3; r++) {
                double nt = 0;
                for (int c = 0; c < 3; c++) { T[r * 4 + c] = R[r * 3 + c]; nt += R[r * 3 + c] * ms[c]; }
                T[r * 4 + 3] = md[r] - sc * nt;
            }
            *scale = sc;
            *rc = 0;
        }
    }
    __syncthreads();
}

// first fit + single-pass outlier rejection [reference :188-197] + final fit [:204], one block.
// errs: scratch of capacity >= n1 doubles; qa/qb are compacted in place.
__device__ void pose_fit_block(float* __restrict__ qa, float* __restrict__ qb, double outlier_thr, int min_matches,
                               double* errs, float* __restrict__ ra, float* __restrict__ rb,
                               PoseOut* __restrict__ out)
{
    __shared__ double sh[10][16];
    __shared__ double s_errs[512];
    __shared__ int s_rank[512];
    __shared__ double s_T[12], s_scale, s_med;
    __shared__ int s_rc, s_n2, s_nan;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n1 = out->n1;
    int n2 = n1;
    const float* fa = qa;
    const float* fb = qb;
    if (tid == 0) { s_rc = 1; s_nan = 0; }
    __syncthreads();
    if (n1 <= 512) errs = s_errs;        // the O(n^2) rank pass below reads every residual n times: keep them in LDS
    if (outlier_thr > 0 && n1 >= 10) {
        STAMP(2);
        dev_umeyama_block(qa, qb, n1, 3, s_T, &s_scale, &s_rc, sh);
        STAMP(3);
        if (tid == 0) { out->rc1 = s_rc; for (int k = 0; k < 12; k++) out->T1[k] = s_T[k]; out->s1 = s_scale; }
        __syncthreads();
        if (s_rc == 0) {
            // relative residual on homogeneous 4-vectors
            for (int i = tid; i < n1; i += nt) {
                const double x = qa[3 * i], y = qa[3 * i + 1], z = qa[3 * i + 2];
                const double X = qb[3 * i], Y = qb[3 * i + 1], Z = qb[3 * i + 2];
                double r[4];
                for (int k = 0; k < 3; k++) r[k] = ((s_T[k * 4] * x + s_T[k * 4 + 1] * y) + s_T[k * 4 + 2] * z) + s_T[k * 4 + 3];
                const double dx = X - r[0], dy = Y - r[1], dz = Z - r[2], dw = 1.0 - (((0.0 * x + 0.0 * y) + 0.0 * z) + 1.0);
                const double e = sqrt(((dx * dx + dy * dy) + dz * dz) + dw * dw) / sqrt(((X * X + Y * Y) + Z * Z) + 1.0);
                errs[i] = e;
                if (e != e) s_nan = 1;
            }
            __syncthreads();
            STAMP(4);
            // np.median: mean of the two middle order statistics (NaN if any NaN)
            if (tid == 0) s_med = 0;
            __syncthreads();
            if (!s_nan && n1 <= 512) {
                // rank of every residual (ties by index): the block splits each residual's comparisons over `parts` threads
                const int parts = max(1, min(nt / n1, 8)), chunk = (n1 + parts - 1) / parts;
                for (int i = tid; i < n1; i += nt) s_rank[i] = 0;
                __syncthreads();
                if (tid < n1 * parts) {
                    const int i = tid % n1, part = tid / n1;
                    const double e = s_errs[i];
                    const int j1 = min((part + 1) * chunk, n1);
                    int rank = 0;
                    for (int j = part * chunk; j < j1; j++) { const double f = s_errs[j]; rank += (f < e) || (f == e && j < i); }
                    atomicAdd(&s_rank[i], rank);
                }
                __syncthreads();
                const int k_hi = n1 / 2, k_lo = (n1 - 1) / 2;
                if (tid < n1) {
                    const double e = s_errs[tid];
                    if (s_rank[tid] == k_hi) atomicAdd(&s_med, 0.5 * e);
                    if (s_rank[tid] == k_lo) atomicAdd(&s_med, 0.5 * e);
                }
            } else if (!s_nan) {
                const int k_hi = n1 / 2, k_lo = (n1 - 1) / 2;
                for (int i = tid; i < n1; i += nt) {
                    const double e = errs[i];
                    int rank = 0;
                    for (int j = 0; j < n1; j++) { const double f = errs[j]; rank += (f < e) || (f == e && j < i); }
                    if (rank == k_hi) atomicAdd(&s_med, 0.5 * e);
                    if (rank == k_lo) atomicAdd(&s_med, 0.5 * e);
                }
            }
            __syncthreads();
            STAMP(5);
            const double thrv = s_nan ? __builtin_nan("") : outlier_thr + s_med;
            // ordered compaction of errors < threshold into ra/rb
            if (tid == 0) s_n2 = 0;
            __syncthreads();
            if (tid < 64) {
                int base = 0;
                for (int j0 = 0; j0 < n1; j0 += 64) {
                    const int j = j0 + tid;
                    const bool keep = j < n1 && errs[j] < thrv;
                    const unsigned long long bal = __ballot(keep);
                    if (keep) {
                        const int o = base + __popcll(bal & ((1ull << tid) - 1ull));
                        for (int c = 0; c < 3; c++) { ra[3 * o + c] = qa[3 * j + c]; rb[3 * o + c] = qb[3 * j + c]; }
                    }
                    base += __popcll(bal);
                }
                if (tid == 0) s_n2 = base;
            }
            __syncthreads();
            n2 = s_n2;
            fa = ra; fb = rb;
        }
    }
    STAMP(6);
    if (tid == 0) { out->n2 = n2; if (s_nan) out->flags |= 2; s_rc = 1; }
    __syncthreads();
    if (n2 >= min_matches) {
        dev_umeyama_block(fa, fb, n2, 3, s_T, &s_scale, &s_rc, sh);
        if (tid == 0) { out->rc2 = s_rc; for (int k = 0; k < 12; k++) out->T2[k] = s_T[k]; out->s2 = s_scale; }
    } else if (tid == 0)
        out->rc2 = 1;
}

// First half of the fused pose step in ONE block (was five launches): wave 0 runs the ratio test with ordered
// compaction, then the whole block looks up the matched keypoints' 3-D positions in both frames and clears the
// consistency counters.  A short chain of launches is what the step costs under load, not its arithmetic.
__global__ void __launch_bounds__(256) k_pose_prep(const int32_t* __restrict__ idx, const int32_t* __restrict__ dist, int nq, double ratio,
                                                   const float* __restrict__ xy_q, const float* __restrict__ xy_t,
                                                   int32_t* __restrict__ q_out, int32_t* __restrict__ t_out, float* __restrict__ xyq_out,
                                                   float* __restrict__ xyt_out, int32_t* __restrict__ m_out, PoseOut* __restrict__ out,
                                                   int* __restrict__ flags_dev, TapDisp ta, TapDisp tb, int cw, int ch,
                                                   float* __restrict__ pts_a, float* __restrict__ pts_b, uint8_t* __restrict__ st_a,
                                                   uint8_t* __restrict__ st_b, int* __restrict__ ncons)
{
    __shared__ int s_m;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int base = 0;
        for (int i0 = 0; i0 < nq; i0 += 64) {
            int i = i0 + lane;
            bool keep = false;
            int t = -1;
            if (i < nq) {
                t = idx[2 * i];
                double a = (double)(float)dist[2 * i], b = (double)(float)dist[2 * i + 1];
                keep = idx[2 * i + 1] >= 0 && a < ratio * b;
            }
            unsigned long long bal = __ballot(keep);
            if (keep) {
                int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                q_out[pos] = i; t_out[pos] = t;
                xyq_out[2 * pos] = xy_q[2 * i]; xyq_out[2 * pos + 1] = xy_q[2 * i + 1];
                xyt_out[2 * pos] = xy_t[2 * t]; xyt_out[2 * pos + 1] = xy_t[2 * t + 1];
            }
            base += __popcll(bal);
        }
        if (lane == 0) {
            s_m = base;
            *m_out = base;
            out->M = base; out->n1 = 0; out->n2 = 0; out->flags = 0; out->rc1 = 1; out->rc2 = 1;
            *flags_dev = 0;
        }
    }
    __syncthreads();
    const int m = s_m;
    int bad = 0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        bilinear_one(ta, cw, ch, xyq_out[2 * i], xyq_out[2 * i + 1], pts_a + 3 * (size_t)i, st_a + i);
        bilinear_one(tb, cw, ch, xyt_out[2 * i], xyt_out[2 * i + 1], pts_b + 3 * (size_t)i, st_b + i);
        bad |= st_a[i] == 2 || st_b[i] == 2;
    }
    if (bad) atomicOr(flags_dev, 1);
    for (int i = threadIdx.x; i < nq; i += blockDim.x) ncons[i] = 0;
}

// Second half in ONE block (was three launches): wave 0 picks the rigid clique, then all 1024 threads run the
// first fit, the outlier pass and the final fit.
__global__ void __launch_bounds__(1024) k_pose_solve(const unsigned long long* __restrict__ bits, int words_cap,
                                                     const int* __restrict__ ncons, const int* __restrict__ m_dev, int use_filter,
                                                     const float* __restrict__ pa, const float* __restrict__ pb, float* __restrict__ qa,
                                                     float* __restrict__ qb, int lds_m_cap, size_t lds_bits_cap,
                                                     const int* __restrict__ flags_dev, double outlier_thr, int min_matches,
                                                     double* __restrict__ errs, float* __restrict__ ra, float* __restrict__ rb,
                                                     PoseOut* __restrict__ out, const uint8_t* __restrict__ lanebytes, int* __restrict__ g_sets,
                                                     unsigned long long* __restrict__ host_rec, int cons_inline, float cons_thr)
{
    extern __shared__ __attribute__((aligned(16))) int s_mem[];
    STAMP(0);
    const int m = *m_dev;
    const bool fast = use_filter && m > 0 && m <= 512 && lds_bits_cap >= (size_t)m * 64;
    if (fast && cons_inline) {
        // up to 512 matches the consistency rows are computed HERE, straight into the LDS bytes the greedy loop reads (row i by
        // wave i mod 16: the arithmetic of k_pose_cons_bits, value for value) -- one launch and one global round trip less on
        // the pose stream; the row counts go into the `nc` quarter of the set arrays, which the register-resident loop leaves free
        uint8_t* const lb = (uint8_t*)(s_mem + 4 * lds_m_cap);
        int* const nc = s_mem + 3 * lds_m_cap;
        // both point sets into LDS first (behind the row bytes): a row reads every point, and from global memory each of its
        // steps waited for a round trip (29 us for 250 matches; 8 us from LDS)
        float* const s_pa = (float*)((uint8_t*)(s_mem + 4 * lds_m_cap) + lds_bits_cap);
        float* const s_pb = s_pa + 3 * lds_m_cap;
        for (int k = threadIdx.x; k < 3 * m; k += blockDim.x) { s_pa[k] = pa[k]; s_pb[k] = pb[k]; }
        __syncthreads();
        const float* const pa = s_pa;                     // (shadow the global pointers for the rows below)
        const float* const pb = s_pb;
        const int lane = threadIdx.x & 63, words = (m + 63) >> 6;
        for (int i = threadIdx.x >> 6; i < m; i += (int)(blockDim.x >> 6)) {
            const float cix = pb[3 * i], ciy = pb[3 * i + 1], ciz = pb[3 * i + 2];
            const float pix = pa[3 * i], piy = pa[3 * i + 1], piz = pa[3 * i + 2];
            unsigned byte = 0;
            int cnt = 0;
            for (int k = 0; k < words; k++) {
                const int j = 64 * k + lane;
                bool c = false;
                if (j < m) {
                    float ax = cix - pb[3 * j], ay = ciy - pb[3 * j + 1], az = ciz - pb[3 * j + 2];
                    float bx = pix - pa[3 * j], by = piy - pa[3 * j + 1], bz = piz - pa[3 * j + 2];
                    float na = sqrtf((ax * ax + ay * ay) + az * az), nb = sqrtf((bx * bx + by * by) + bz * bz);
                    c = fabsf(na - nb) < cons_thr;
                }
                cnt += __popcll(__ballot(c));
                byte |= (unsigned)c << (k & 7);
            }
            lb[(size_t)i * 64 + lane] = (uint8_t)byte;
            if (lane == 0) nc[i] = cnt;
        }
    } else if (fast) {
        // lanebits[row][lane] = bit k set when row is consistent with element lane + 64 k (written by k_pose_cons_bits):
        // into LDS, 16 bytes per thread
        uint4* dst = (uint4*)(s_mem + 4 * lds_m_cap);
        const uint4* src = (const uint4*)lanebytes;
        for (int idx = threadIdx.x; idx < m * 4; idx += blockDim.x) dst[idx] = src[idx];
    }
    __syncthreads();
    // the four per-match int arrays of the greedy loop: LDS, or (more matches than 56 KB of LDS hold) the global workspace
    if (threadIdx.x < 64)
        pose_clique_wave(g_sets ? g_sets : s_mem, bits, words_cap, (fast && cons_inline) ? s_mem + 3 * lds_m_cap : ncons, m_dev, use_filter, pa, pb, qa, qb,
                         &out->n1, lds_m_cap, lds_bits_cap, fast);
    if (threadIdx.x == 0) out->flags |= *flags_dev;
    STAMP(1);
    __syncthreads();
    pose_fit_block(qa, qb, outlier_thr, min_matches, errs, ra, rb, out);
    STAMP(9);
    // the finished record goes straight into the caller's PINNED host record (the copy command that used to follow this kernel
    // was one more entry on the pose stream's queue: 0.1 - 0.2 ms of latency beside 16 busy engines).  The record's fields were
    // written by several threads of this block: the stores are complete behind the barrier, the loads bypass this CU's L1.
    __syncthreads();
    if (host_rec && threadIdx.x < sizeof(PoseOut) / 8)
        host_rec[threadIdx.x] = __hip_atomic_load((const unsigned long long*)out + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// enqueue the whole fused step for two slots on ctx->stream with the scratch currently installed in ctx; the
size_t pose_ws_bytes(int nq)
{
    const int words = (nq + 63) / 64;
    return (size_t)nq * words * 8 + (size_t)nq * 4 + (size_t)nq * 12 * 4 + (size_t)nq * 8 + (size_t)nq * 64 + ((size_t)nq + 2) * 16 + 4096;
}

// k_pose_solve writes the finished PoseOut record into host_out (pinned host memory the device can address) itself.  No host
// synchronisation, no copy command.
static int pose_enqueue(vo_ctx* ctx, FrameSlot& a, FrameSlot& b, double ratio, int min_matches, double rigidity_thr,
                        double outlier_thr, void* host_out)
{
    const int nq = a.n_kp;
    // workspace: bit matrix + ncons + filtered point sets + residuals + result.  vo_create / pose_alt_prepare size it
    // for kp_cap query keypoints, so the branch below (a device-wide synchronisation) is never taken on the hot path
    const int words = (nq + 63) / 64;
    const size_t need = pose_ws_bytes(nq);
    if (ctx->mw->clique_ws_bytes < need) {
        if (ctx->mw->clique_ws) { VO_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->mw->clique_ws); }
        ctx->mw->clique_ws = nullptr; ctx->mw->clique_ws_bytes = 0;
        VO_HIP(ctx, hipMalloc((void**)&ctx->mw->clique_ws, need));
        ctx->mw->clique_ws_bytes = need;
    }
    uint8_t* wsp = ctx->mw->clique_ws;
    PoseOut* d_out = (PoseOut*)wsp; wsp += 1024;
    int* d_flags = (int*)wsp; wsp += 256;
    unsigned long long* d_bits = (unsigned long long*)wsp; wsp += (size_t)nq * words * 8;
    double* d_errs = (double*)wsp; wsp += (size_t)nq * 8;
    int* d_ncons = (int*)wsp; wsp += (size_t)nq * 4;
    float* d_qa = (float*)wsp; wsp += (size_t)nq * 12;
    float* d_qb = (float*)wsp; wsp += (size_t)nq * 12;
    float* d_ra = (float*)wsp; wsp += (size_t)nq * 12;
    float* d_rb = (float*)wsp; wsp += (size_t)nq * 12;
    uint8_t* d_lanebytes = (uint8_t*)(((uintptr_t)wsp + 15) & ~(uintptr_t)15);
    int* d_sets = (int*)(d_lanebytes + (size_t)nq * 64);          // 4 x m_cap ints (only used beyond 3584 keypoints)
    int* d_m = ctx->mw->m_count;  // k_pose_prep writes the match count M here
    int rc;
    {
        StageTimer t(ctx, VO_T_MATCH);
        rc = match_knn2(ctx, a.desc, a.n_kp, b.desc, b.n_kp, ctx->mw->m_idx, ctx->mw->m_dist);
        if (rc) return rc;
    }
    {
        StageTimer t(ctx, VO_T_POSE);
        int x0 = 0, y0 = 0, x1 = a.w, y1 = a.h;
        if (ctx->has_roi) { x0 = ctx->roi[0]; y0 = ctx->roi[1]; x1 = ctx->roi[2] < a.w ? ctx->roi[2] : a.w; y1 = ctx->roi[3] < a.h ? ctx->roi[3] : a.h; }
        TapDisp ta{ a.disp16, a.w, x0, y0, make_q(ctx->Q) };
        TapDisp tb{ b.disp16, b.w, x0, y0, make_q(ctx->Q) };
        hipLaunchKernelGGL(k_pose_prep, dim3(1), dim3(256), 0, ctx->stream, ctx->mw->m_idx, ctx->mw->m_dist, a.n_kp, ratio, a.kp_xy, b.kp_xy,
                           ctx->mw->mq_idx, ctx->mw->mt_idx, ctx->mw->xy_a, ctx->mw->xy_b, d_m, d_out, d_flags, ta, tb, x1 - x0, y1 - y0, ctx->mw->pts_a,
                           ctx->mw->pts_b, ctx->mw->st_a, ctx->mw->st_b, d_ncons);
        const int use_filter = rigidity_thr > 0;
        // LDS: 4 int arrays of nq (rounded to even so the bit matrix stays 8-byte aligned) + bit matrix if <= 48 KB
        const int m_cap = (nq + 1) & ~1;
        size_t bits_cap = (size_t)nq * words * 8;
        if (nq <= 512 && bits_cap < (size_t)nq * 64) bits_cap = (size_t)nq * 64;   // one byte per (row, lane): the register-resident greedy loop
        if ((size_t)m_cap * 16 + bits_cap > 56 * 1024) bits_cap = 0;
        const bool sets_global = (size_t)m_cap * 16 > 56 * 1024;      // > 3584 keypoints: the sets move to the workspace, LDS stays empty
        // up to 512 query keypoints (M <= nq) k_pose_solve computes the consistency rows itself, into LDS: no launch of their own
        // (needs both point sets in LDS beside the sets and the row bytes: 24 bytes per match more)
        const bool cons_inline = use_filter && nq <= 512 && !sets_global && bits_cap >= (size_t)nq * 64 &&
                                 (size_t)m_cap * 16 + bits_cap + (size_t)m_cap * 24 <= 60 * 1024;
        if (use_filter && !cons_inline)
            hipLaunchKernelGGL(k_pose_cons_bits, dim3(nq), dim3(64), 0, ctx->stream, ctx->mw->pts_a, ctx->mw->pts_b, d_m, (float)rigidity_thr,
                               d_bits, words, d_ncons, d_lanebytes);
        hipLaunchKernelGGL(k_pose_solve, dim3(1), dim3(1024), sets_global ? 0 : (size_t)m_cap * 16 + bits_cap + (cons_inline ? (size_t)m_cap * 24 : 0), ctx->stream, d_bits, words, d_ncons, d_m,
                           use_filter, ctx->mw->pts_a, ctx->mw->pts_b, d_qa, d_qb, m_cap, bits_cap, d_flags, outlier_thr, min_matches, d_errs,
                           d_ra, d_rb, d_out, d_lanebytes, sets_global ? d_sets : nullptr, (unsigned long long*)host_out, cons_inline ? 1 : 0, (float)rigidity_thr);
        VO_CHECK_LAUNCH(ctx);
    }
    return VO_OK;
}

static int pose_check(vo_ctx* ctx, int slot_a, int slot_b)
{
    if (!ctx || slot_a < 0 || slot_a >= VO_NUM_SLOTS || slot_b < 0 || slot_b >= VO_NUM_SLOTS) return vo_fail(ctx, VO_E_ARG, "vo_pose_pair: bad argument");
    FrameSlot& a = ctx->slots[slot_a];
    FrameSlot& b = ctx->slots[slot_b];
    if (!a.has_kp || !b.has_kp || !a.has_disp || !b.has_disp) return vo_fail(ctx, VO_E_STATE, "slots need disparity and keypoints");
    if (!ctx->has_Q) return vo_fail(ctx, VO_E_STATE, "vo_set_Q has not been called");
    if (a.n_kp > 0 && b.n_kp < 2) return vo_fail(ctx, VO_E_ARG, "train set has fewer than 2 descriptors (reference raises IndexError)");
    if (a.n_kp >= 32768 || (size_t)a.n_kp * 16 > 60 * 1024) return vo_fail(ctx, VO_E_CAP, "%d query keypoints exceed the fused pose path", a.n_kp);
    return VO_OK;
}

static void pose_unpack(const void* rec, int32_t* counts4, int32_t* rc2, double* T1_12, double* T2_12)
{
    const PoseOut* o = (const PoseOut*)rec;
    counts4[0] = o->M; counts4[1] = o->n1; counts4[2] = o->n2; counts4[3] = o->flags;
    rc2[0] = o->rc1; rc2[1] = o->rc2;
    if (T1_12) memcpy(T1_12, o->T1, sizeof(o->T1));
    memcpy(T2_12, o->T2, sizeof(o->T2));
}

extern "C" int vo_pose_pair(vo_ctx* ctx, int slot_a, int slot_b, double ratio, int min_matches, double rigidity_thr,
                            double outlier_thr, int32_t* counts4 /*M,n1,n2,flags*/, int32_t* rc2 /*first,final*/,
                            double* T1_12, double* T2_12)
{
    if (!counts4 || !rc2 || !T2_12) return vo_fail(ctx, VO_E_ARG, "vo_pose_pair: bad argument");
    int rc = pose_check(ctx, slot_a, slot_b);
    if (rc) return rc;
    FrameSlot& a = ctx->slots[slot_a];
    FrameSlot& b = ctx->slots[slot_b];
    VO_HIP(ctx, hipSetDevice(ctx->device));
    { int rcw = slot_wait(ctx, a); if (!rcw) rcw = slot_wait(ctx, b); if (rcw) return rcw; }
    counts4[0] = counts4[1] = counts4[2] = counts4[3] = 0;
    rc2[0] = rc2[1] = 1;
    if (a.n_kp == 0) return VO_OK;
    if ((rc = pose_enqueue(ctx, a, b, ratio, min_matches, rigidity_thr, outlier_thr, ctx->pinned))) return rc;
    if ((rc = xfer_flush(ctx))) return rc;
    if ((rc = slot_health(ctx, a, slot_a)) || (rc = slot_health(ctx, b, slot_b))) return rc;   // never a pose from an undefined disparity
    pose_unpack(ctx->pinned, counts4, rc2, T1_12, T2_12);
    return VO_OK;
}

// The context works on alternate k's stream and in its match scratch for the lifetime of the object, whatever leaves the scope
// (the stream handle changes places with the main one, `mw` is retargeted; no member of a workspace is copied)
struct PoseScope {
    vo_ctx* c;
    int k;
    PoseScope(vo_ctx* c_, int k_) : c(c_), k(k_)
    {
        std::swap(c->stream, c->pose_alt[k].stream);
        c->mw = &c->pose_alt[k].mw;
    }
    ~PoseScope()
    {
        c->mw = &c->main_mw;
        std::swap(c->stream, c->pose_alt[k].stream);
    }
    PoseScope(const PoseScope&) = delete;
    PoseScope& operator=(const PoseScope&) = delete;
};

static int pose_alt_prepare(vo_ctx* ctx, int k)
{
    vo_ctx::PoseAlt& p = ctx->pose_alt[k];
    if (p.ready) return VO_OK;
    const size_t cap = (size_t)ctx->kp_cap;
    hipStream_t& shared = ctx->pose_streams[k % ctx->n_pose_streams];
    if (!shared) VO_HIP(ctx, hipStreamCreateWithFlags(&shared, hipStreamNonBlocking));
    p.stream = shared;                            // (not owned by the alternate)
    VO_HIP(ctx, hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
    VO_HIP(ctx, hipHostMalloc(&p.result, 1024, hipHostMallocDefault));
    VO_HIP(ctx, hipMalloc((void**)&p.mw.m_idx, cap * 8 + 256));
    if (match_dist_alloc(ctx, &p.mw.m_dist)) return vo_fail(ctx, VO_E_HIP, "asynchronous pose step: allocation failed");
    VO_HIP(ctx, hipMalloc((void**)&p.mw.m_count, 256));
    VO_HIP(ctx, hipMalloc((void**)&p.mw.mq_idx, cap * 4 + 256)); VO_HIP(ctx, hipMalloc((void**)&p.mw.mt_idx, cap * 4 + 256));
    VO_HIP(ctx, hipMalloc((void**)&p.mw.pts_a, cap * 12 + 256)); VO_HIP(ctx, hipMalloc((void**)&p.mw.pts_b, cap * 12 + 256));
    VO_HIP(ctx, hipMalloc((void**)&p.mw.xy_a, cap * 8 + 256)); VO_HIP(ctx, hipMalloc((void**)&p.mw.xy_b, cap * 8 + 256));
    VO_HIP(ctx, hipMalloc((void**)&p.mw.st_a, cap + 256)); VO_HIP(ctx, hipMalloc((void**)&p.mw.st_b, cap + 256));
    p.mw.clique_ws_bytes = pose_ws_bytes(ctx->kp_cap);
    VO_HIP(ctx, hipMalloc((void**)&p.mw.clique_ws, p.mw.clique_ws_bytes));
    p.ready = true;
    return VO_OK;
}

void pose_alt_free(vo_ctx* ctx)
{
    for (int k = 0; k < vo_ctx::N_POSE_ALT; k++) {
        vo_ctx::PoseAlt& p = ctx->pose_alt[k];
        if (p.stream) (void)hipStreamSynchronize(p.stream);
        void* ps[] = { p.mw.m_idx, p.mw.m_dist, p.mw.m_count, p.mw.mq_idx, p.mw.mt_idx, p.mw.pts_a, p.mw.pts_b, p.mw.xy_a, p.mw.xy_b, p.mw.st_a, p.mw.st_b, p.mw.clique_ws };
        for (void* q : ps) if (q) (void)hipFree(q);
        if (p.result) (void)hipHostFree(p.result);
        if (p.done) (void)hipEventDestroy(p.done);
        p = vo_ctx::PoseAlt();
    }
    for (hipStream_t& st : ctx->pose_streams) {
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
    }
}

extern "C" int vo_pose_pair_begin(vo_ctx* ctx, int slot_a, int slot_b, double ratio, int min_matches, double rigidity_thr,
                                  double outlier_thr, int* ticket_out)
{
    if (!ticket_out) return vo_fail(ctx, VO_E_ARG, "vo_pose_pair_begin: bad argument");
    int rc = pose_check(ctx, slot_a, slot_b);
    if (rc) return rc;
    VO_HIP(ctx, hipSetDevice(ctx->device));
    int k = -1;                                   // the first free alternate from the round-robin position on (tickets need not end in order)
    for (int i = 0; i < vo_ctx::N_POSE_ALT && k < 0; i++)
        if (!ctx->pose_alt[(ctx->pose_next + i) % vo_ctx::N_POSE_ALT].busy) k = (ctx->pose_next + i) % vo_ctx::N_POSE_ALT;
    if (k < 0) return vo_fail(ctx, VO_E_STATE, "vo_pose_pair_begin: every asynchronous pose step is still open (end one first)");
    vo_ctx::PoseAlt& p = ctx->pose_alt[k];
    // the first step begun builds EVERY alternate (a dozen allocations, a pinned record and an event each: ~1 ms apiece): built one
    // by one as the round-robin first reaches them, the later ones fell into whatever the caller was timing by then (the first two
    // of bench.py's five cold windows read 10 % low)
    for (int i = 0; i < vo_ctx::N_POSE_ALT; i++)
        if ((rc = pose_alt_prepare(ctx, (k + i) % vo_ctx::N_POSE_ALT))) return rc;
    FrameSlot& a = ctx->slots[slot_a];
    FrameSlot& b = ctx->slots[slot_b];
    PoseOut* rec = (PoseOut*)p.result;
    memset(rec, 0, sizeof(PoseOut));
    rec->rc1 = rec->rc2 = 1;
    // the step runs on the alternate's own stream: order it behind whatever still produces the two slots
    // (look-ahead engines) and behind the main stream's work on them
    VO_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    {
        PoseScope on_alt(ctx, k);
        hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->ev0, 0);
        if (e == hipSuccess && a.pending) e = hipStreamWaitEvent(ctx->stream, a.ready, 0);
        if (e == hipSuccess && b.pending) e = hipStreamWaitEvent(ctx->stream, b.ready, 0);
        rc = e == hipSuccess ? VO_OK : vo_fail(ctx, VO_E_HIP, "hipStreamWaitEvent failed: %s", hipGetErrorString(e));
        if (!rc && a.n_kp > 0) rc = pose_enqueue(ctx, a, b, ratio, min_matches, rigidity_thr, outlier_thr, rec);
        if (!rc && hipEventRecord(p.done, ctx->stream) != hipSuccess) rc = vo_fail(ctx, VO_E_HIP, "hipEventRecord failed");
    }
    if (rc) return rc;
    // both slots are read on this alternate's stream until p.done: whoever refills one of them waits for it first
    for (FrameSlot* f : { &a, &b }) {
        slot_add_reader(*f, p.done);
    }
    p.busy = true; p.slot_a = slot_a; p.slot_b = slot_b;
    p.gen_a = a.disp_gen; p.gen_b = b.disp_gen;
    p.params[0] = ratio; p.params[1] = min_matches; p.params[2] = rigidity_thr; p.params[3] = outlier_thr;
    ctx->pose_next = (k + 1) % vo_ctx::N_POSE_ALT;
    *ticket_out = k;
    return VO_OK;
}

extern "C" int vo_pose_pair_end(vo_ctx* ctx, int ticket, int32_t* counts4, int32_t* rc2, double* T1_12, double* T2_12)
{
    if (!ctx || ticket < 0 || ticket >= vo_ctx::N_POSE_ALT || !counts4 || !rc2 || !T2_12) return vo_fail(ctx, VO_E_ARG, "vo_pose_pair_end: bad argument");
    vo_ctx::PoseAlt& p = ctx->pose_alt[ticket];
    if (!p.busy) return vo_fail(ctx, VO_E_STATE, "vo_pose_pair_end: ticket %d is not open", ticket);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    p.busy = false;
    VO_HIP(ctx, hipEventSynchronize(p.done));
    // the step read the disparities the two slots held when it was begun: were those runs healthy?  (a slot refilled since
    // then carries another generation and another word value)
    const int32_t gens[2] = { p.gen_a, p.gen_b };
    const int slots[2] = { p.slot_a, p.slot_b };
    for (int i = 0; i < 2; i++) {
        const FrameSlot& f = ctx->slots[slots[i]];
        if (gens[i] != 0 && *(volatile int32_t*)f.sweep_word == gens[i])
            return vo_fail(ctx, VO_E_SWEEP, "pose step %d: the aggregation sweep of the pair in slot %d gave up a strip hand-off: its "
                                            "disparity is undefined and no pose is derived from it", ticket, slots[i]);
    }
    pose_unpack(p.result, counts4, rc2, T1_12, T2_12);
    return VO_OK;
}
