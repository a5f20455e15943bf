// RANSAC essential-matrix hypothesis generation + Sampson inlier scoring on gfx950
// (BASELINE config 5: monocular 1920x1080, 8000 keypoints, 5000 hypotheses).
//
// The reference has NO RANSAC (its pose is a closed-form Umeyama fit): this stage has
// no openVO counterpart and is defined by this build; the CPU restatement lives with the other
// test infrastructure and the two must agree bit for bit (hypothesis order, inlier counts, masks).
//
//   k_ransac_hyp    one lane per hypothesis: counter-based hash sampling of 8 distinct matches,
//                   A^T A (9x9) in float64, cyclic Jacobi for the smallest eigenvector, 3x3 SVD
//                   projection on the essential manifold, F = K^-T E K^-1 rounded to float32
//   k_ransac_score  one wave per hypothesis (4 per block); lanes stride over the matches held in
//                   LDS-free registers, Sampson test in float32 with a fixed operation order, inlier
//                   count by __ballot + popcount -- 40 M residuals at config 5
//   k_ransac_best   argmax (ties -> lowest hypothesis) with a DPP-free shuffle reduction, then the
//                   winner's inlier mask
#include <math.h>
#include "vo_internal.h"

__device__ __forceinline__ uint32_t lowbias32(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

__device__ void rs_cross3(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// one-sided Jacobi SVD of a 3x3 matrix (same algorithm as the pose path)
__device__ void rs_svd3(const double* A, double* U, double* w, double* Vt)
{
    double G[9], V[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int k = 0; k < 9; k++) G[k] = A[k];
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
        const int o = ord[j];
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
        rs_cross3(Uc[0], a, Uc[1]);
        double nn = sqrt(Uc[1][0] * Uc[1][0] + Uc[1][1] * Uc[1][1] + Uc[1][2] * Uc[1][2]);
        for (int i = 0; i < 3; i++) Uc[1][i] /= nn;
    }
    if (w[2] <= tiny || w[2] <= 1e-14 * w[0]) {
        rs_cross3(Uc[0], Uc[1], Uc[2]);
        double nn = sqrt(Uc[2][0] * Uc[2][0] + Uc[2][1] * Uc[2][1] + Uc[2][2] * Uc[2][2]);
        if (nn > 0) for (int i = 0; i < 3; i++) Uc[2][i] /= nn;
    }
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++) { U[i * 3 + j] = Uc[j][i]; Vt[j * 3 + i] = Vc[j][i]; }
}

struct K4 { double fx, fy, cx, cy; };

// n_dev (may be NULL): the number of correspondences when only the device knows it (vo_mono_pair: the ratio test's
// survivor count); fewer than 8 correspondences give all-zero hypotheses that score no inlier
// F = K^-T E K^-1, scaled to max |entry| = 1, rounded to float32
__device__ void rs_emit(const double* E, const K4& K, int h, double* __restrict__ E_out, float* __restrict__ F_out)
{
    const double ifx = 1.0 / K.fx, ify = 1.0 / K.fy;
    const double Ki[9] = { ifx, 0, -K.cx * ifx, 0, ify, -K.cy * ify, 0, 0, 1 };
    double T[9], Fd[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += E[r * 3 + k] * Ki[k * 3 + c]; T[r * 3 + c] = s; }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += Ki[k * 3 + r] * T[k * 3 + c]; Fd[r * 3 + c] = s; }
    double mx = 0;
    for (int k = 0; k < 9; k++) if (fabs(Fd[k]) > mx) mx = fabs(Fd[k]);
    const double sc = mx > 0 ? 1.0 / mx : 1.0;
    for (int k = 0; k < 9; k++) { E_out[(size_t)h * 9 + k] = E[k]; F_out[(size_t)h * 9 + k] = (float)(Fd[k] * sc); }
}

__global__ void __launch_bounds__(64) k_ransac_hyp(const float* __restrict__ p1, const float* __restrict__ p2, int n, K4 K,
                                                   int iters, uint32_t seed, double* __restrict__ E_out, float* __restrict__ F_out,
                                                   const int* __restrict__ n_dev)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= iters) return;
    if (n_dev) n = *n_dev;
    if (n < 8) {
        for (int k = 0; k < 9; k++) { E_out[(size_t)h * 9 + k] = 0.0; F_out[(size_t)h * 9 + k] = 0.f; }
        return;
    }
    int idx[8];
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
    double a[9][9];
    for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) a[i][j] = 0.0;
    for (int s = 0; s < 8; s++) {
        const int i = idx[s];
        const double x1 = ((double)p1[2 * i] - K.cx) / K.fx, y1 = ((double)p1[2 * i + 1] - K.cy) / K.fy;
        const double x2 = ((double)p2[2 * i] - K.cx) / K.fx, y2 = ((double)p2[2 * i + 1] - K.cy) / K.fy;
        const double r[9] = { x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0 };
        for (int u = 0; u < 9; u++) for (int v = 0; v < 9; v++) a[u][v] += r[u] * r[v];
    }
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
    double e0[9], U[9], w[3], Vt[9], E[9];
    for (int k = 0; k < 9; k++) e0[k] = v[k][m];
    rs_svd3(e0, U, w, Vt);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) E[r * 3 + c] = U[r * 3 + 0] * Vt[0 * 3 + c] + U[r * 3 + 1] * Vt[1 * 3 + c];
    rs_emit(E, K, h, E_out, F_out);
}

#include "fivept.inc"

__device__ __forceinline__ bool sampson_inlier(const float* F, float u1, float v1, float u2, float v2, float thr2)
{
    const float fx0 = (F[0] * u1 + F[1] * v1) + F[2];
    const float fx1 = (F[3] * u1 + F[4] * v1) + F[5];
    const float fx2 = (F[6] * u1 + F[7] * v1) + F[8];
    const float ft0 = (F[0] * u2 + F[3] * v2) + F[6];
    const float ft1 = (F[1] * u2 + F[4] * v2) + F[7];
    const float num = (u2 * fx0 + v2 * fx1) + fx2;
    const float den = ((fx0 * fx0 + fx1 * fx1) + ft0 * ft0) + ft1 * ft1;
    return (num * num) / den < thr2;
}

// one wave per hypothesis
__global__ void __launch_bounds__(256) k_ransac_score(const float* __restrict__ p1, const float* __restrict__ p2, int n,
                                                     const float* __restrict__ F_all, int iters, float thr2, int32_t* __restrict__ counts,
                                                     const int* __restrict__ n_dev, int min_n)
{
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (h >= iters) return;
    if (n_dev) n = *n_dev < min_n ? 0 : *n_dev;
    float F[9];
#pragma unroll
    for (int k = 0; k < 9; k++) F[k] = F_all[(size_t)h * 9 + k];
    int cnt = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        bool in = false;
        if (i < n) {
            const float2 a = ((const float2*)p1)[i], b = ((const float2*)p2)[i];
            in = sampson_inlier(F, a.x, a.y, b.x, b.y, thr2);
        }
        cnt += __popcll(__ballot(in));
    }
    if (lane == 0) counts[h] = cnt;
}

__global__ void __launch_bounds__(1024) k_ransac_best(const int32_t* __restrict__ counts, int iters, int32_t* __restrict__ best /*[2]: index, count*/)
{
    __shared__ long long s_key[16];
    long long key = -1;
    for (int h = threadIdx.x; h < iters; h += blockDim.x) {
        const long long k = ((long long)counts[h] << 32) | (long long)(0x7fffffff - h);   // most inliers, then lowest index
        key = k > key ? k : key;
    }
    for (int o = 32; o > 0; o >>= 1) { const long long other = __shfl_xor(key, o, 64); key = other > key ? other : key; }
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < (int)(blockDim.x >> 6); q++) key = s_key[q] > key ? s_key[q] : key;
        best[0] = 0x7fffffff - (int)(key & 0x7fffffffLL);
        best[1] = (int)(key >> 32);
    }
}

__global__ void k_ransac_mask(const float* __restrict__ p1, const float* __restrict__ p2, int n, const float* __restrict__ F_all,
                              const int32_t* __restrict__ best, float thr2, uint8_t* __restrict__ mask, const int* __restrict__ n_dev, int min_n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) { const int m = *n_dev; if (i < n && (i >= m || m < min_n)) mask[i] = 0; n = m < min_n ? 0 : m; }
    if (i >= n) return;
    const float* F = F_all + (size_t)best[0] * 9;
    float Fl[9];
    for (int k = 0; k < 9; k++) Fl[k] = F[k];
    const float2 a = ((const float2*)p1)[i], b = ((const float2*)p2)[i];
    mask[i] = sampson_inlier(Fl, a.x, a.y, b.x, b.y, thr2) ? 1 : 0;
}

// the five-point kernel keeps its matrices in 100 KB of LDS per wave: above the 64 KB a launch may ask for by default
static int hyp5_prepare(vo_ctx* ctx)
{
    static unsigned long long done = 0;         // per device (the attribute belongs to the device's copy of the code object)
    if ((done >> (ctx->device & 63)) & 1ull) return VO_OK;
    VO_HIP(ctx, hipFuncSetAttribute((const void*)k_ransac_hyp5, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FP_LDS_DOUBLES * sizeof(double))));
    done |= 1ull << (ctx->device & 63);
    return VO_OK;
}

static int ransac_essential(vo_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K4v, int iters, float thr,
                            uint32_t seed, double* E9_out, uint8_t* mask_out, int32_t* counts_out, int32_t* best2_out, int solver)
{
    const int min_n = solver == 5 ? 6 : 8;
    if (!ctx || !pts1 || !pts2 || !K4v || !E9_out || !best2_out) return vo_fail(ctx, VO_E_ARG, "vo_ransac_essential: bad argument");
    if (n < min_n || iters <= 0 || iters > (1 << 22) || n > (1 << 24))
        return vo_fail(ctx, VO_E_ARG, "vo_ransac_essential: need n >= %d and 0 < iters <= 4194304", min_n);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    // workspace: points (2 x n x 8 B), E (iters x 72 B), F (iters x 36 B), counts, mask, best
    const size_t need = (size_t)n * 16 + (size_t)iters * (72 + 36 + 4 + FP_REC_DOUBLES * 8) + (size_t)n + 4096;
    if (ctx->mw->ransac_ws_bytes < need) {
        if (ctx->mw->ransac_ws) (void)hipFree(ctx->mw->ransac_ws);
        ctx->mw->ransac_ws = nullptr; ctx->mw->ransac_ws_bytes = 0;
        VO_HIP(ctx, hipMalloc((void**)&ctx->mw->ransac_ws, need));
        ctx->mw->ransac_ws_bytes = need;
    }
    uint8_t* w = ctx->mw->ransac_ws;
    double* d_E = (double*)w; w += (size_t)iters * 72;
    double* d_rec = (double*)w; w += (size_t)iters * FP_REC_DOUBLES * 8;       // five-point solver: k_ransac_hyp5 -> k_ransac_roots5
    float* d_p1 = (float*)w; w += (size_t)n * 8;
    float* d_p2 = (float*)w; w += (size_t)n * 8;
    float* d_F = (float*)w; w += (size_t)iters * 36;
    int32_t* d_counts = (int32_t*)w; w += (size_t)iters * 4;
    int32_t* d_best = (int32_t*)w; w += 256;
    uint8_t* d_mask = w;
    StageTimer t(ctx, VO_T_POSE);
    int rc = xfer_h2d(ctx, d_p1, pts1, (size_t)n * 8);
    if (!rc) rc = xfer_h2d(ctx, d_p2, pts2, (size_t)n * 8);
    if (rc) return rc;
    const K4 K{ K4v[0], K4v[1], K4v[2], K4v[3] };
    const float thr2 = thr * thr;
    if (solver == 5) {
        if (int rc5 = hyp5_prepare(ctx)) return rc5;
        hipLaunchKernelGGL(k_ransac_hyp5, dim3(div_up(iters, 64)), dim3(64), FP_LDS_DOUBLES * sizeof(double), ctx->stream, d_p1, d_p2, n, K, iters, seed, d_rec, nullptr);
        hipLaunchKernelGGL(k_ransac_roots5, dim3(div_up(iters, 8)), dim3(256), 0, ctx->stream, d_rec, K, iters, d_E, d_F);
    }
    else
        hipLaunchKernelGGL(k_ransac_hyp, dim3(div_up(iters, 64)), dim3(64), 0, ctx->stream, d_p1, d_p2, n, K, iters, seed, d_E, d_F, nullptr);
    hipLaunchKernelGGL(k_ransac_score, dim3(div_up(iters, 4)), dim3(256), 0, ctx->stream, d_p1, d_p2, n, d_F, iters, thr2, d_counts, nullptr, min_n);
    hipLaunchKernelGGL(k_ransac_best, dim3(1), dim3(1024), 0, ctx->stream, d_counts, iters, d_best);
    hipLaunchKernelGGL(k_ransac_mask, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_p1, d_p2, n, d_F, d_best, thr2, d_mask, nullptr, min_n);
    VO_CHECK_LAUNCH(ctx);
    VO_HIP(ctx, hipMemcpyAsync(ctx->pinned, d_best, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (mask_out && (rc = xfer_d2h(ctx, mask_out, d_mask, (size_t)n))) return rc;
    if (counts_out && (rc = xfer_d2h(ctx, counts_out, d_counts, (size_t)iters * 4))) return rc;
    if ((rc = xfer_flush(ctx))) return rc;
    best2_out[0] = ((int32_t*)ctx->pinned)[0];
    best2_out[1] = ((int32_t*)ctx->pinned)[1];
    VO_HIP(ctx, hipMemcpy(E9_out, d_E + (size_t)best2_out[0] * 9, 72, hipMemcpyDeviceToHost));
    return VO_OK;
}

extern "C" int vo_ransac_essential(vo_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K4v, int iters, float thr,
                                   uint32_t seed, double* E9_out, uint8_t* mask_out, int32_t* counts_out, int32_t* best2_out)
{
    return ransac_essential(ctx, pts1, pts2, n, K4v, iters, thr, seed, E9_out, mask_out, counts_out, best2_out, 8);
}

extern "C" int vo_ransac_essential5(vo_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K4v, int iters, float thr,
                                    uint32_t seed, double* E9_out, uint8_t* mask_out, int32_t* counts_out, int32_t* best2_out)
{
    return ransac_essential(ctx, pts1, pts2, n, K4v, iters, thr, seed, E9_out, mask_out, counts_out, best2_out, 5);
}

// winner's E (9 doubles) gathered on the device so that one flush brings everything home
__global__ void k_ransac_pick(const double* __restrict__ E_all, const int32_t* __restrict__ best, double* __restrict__ E9)
{
    if (threadIdx.x < 9) E9[threadIdx.x] = E_all[(size_t)best[0] * 9 + threadIdx.x];
}

// Monocular pair step (BASELINE config 5; no reference counterpart): the keypoints and descriptors two slots
// already hold -> brute-force Hamming kNN-2 -> ratio test + ordered compaction -> essential-matrix RANSAC on the
// surviving correspondences, every stage on the device.  mono_enqueue puts the whole chain on ctx->stream with the
// context's CURRENT match scratch (m_idx .. xy_b, m_count) and RANSAC workspace -- the main ones (vo_mono_pair: one
// host synchronisation at the end) or those of an asynchronous alternate (vo_mono_pair_begin / _end).
struct MonoDev { int32_t* d_best; double* d_E9; uint8_t* d_mask; };

// The end of an asynchronous pair step as ONE launch (was three kernels and seven copy commands: at 8000 keypoints the
// monocular loop is bound by the host thread's queue entries): argmax of the hypotheses' inlier counts (most inliers, then the
// lowest index) -> the winner's E -> its inlier mask over the M correspondences -> the whole record -- M, winner, count, E,
// mask, the surviving index pairs and the second frame's keypoint positions -- written straight into the alternate's PINNED
// host record (one block: the record is 140 KB at 8000 keypoints).  Element for element the arithmetic of k_ransac_best /
// k_ransac_mask / k_ransac_pick.
__global__ void __launch_bounds__(1024) k_mono_finish(const int32_t* __restrict__ counts, int iters, const double* __restrict__ E_all,
                                                      const float* __restrict__ F_all, const float* __restrict__ p1, const float* __restrict__ p2,
                                                      int nq, const int* __restrict__ n_dev, int min_n, float thr2, const int32_t* __restrict__ mq,
                                                      const int32_t* __restrict__ mt, const float* __restrict__ xy_b, int nb, int want, size_t cap,
                                                      uint8_t* __restrict__ rec, size_t hdr)
{
    __shared__ long long s_key[16];
    __shared__ int s_best[2];
    long long key = -1;
    for (int h = threadIdx.x; h < iters; h += blockDim.x) {
        const long long k = ((long long)counts[h] << 32) | (long long)(0x7fffffff - h);   // most inliers, then lowest index
        key = k > key ? k : key;
    }
    for (int o = 32; o > 0; o >>= 1) { const long long other = __shfl_xor(key, o, 64); key = other > key ? other : key; }
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < (int)(blockDim.x >> 6); q++) key = s_key[q] > key ? s_key[q] : key;
        s_best[0] = 0x7fffffff - (int)(key & 0x7fffffffLL);
        s_best[1] = (int)(key >> 32);
    }
    __syncthreads();
    const int best = s_best[0], m = *n_dev;
    int32_t* const h32 = (int32_t*)rec;
    if (threadIdx.x == 0) { h32[0] = m; h32[1] = best; h32[2] = s_best[1]; }
    if (threadIdx.x < 9) ((double*)(rec + 64))[threadIdx.x] = E_all[(size_t)best * 9 + threadIdx.x];
    if (!want) return;
    uint8_t* const q = rec + hdr;
    float Fl[9];
    for (int k = 0; k < 9; k++) Fl[k] = F_all[(size_t)best * 9 + k];
    const int live = m < min_n ? 0 : m;
    for (int i = threadIdx.x; i < nq; i += blockDim.x) {
        uint8_t in = 0;
        if (i < live) {
            const float2 a = ((const float2*)p1)[i], b = ((const float2*)p2)[i];
            in = sampson_inlier(Fl, a.x, a.y, b.x, b.y, thr2) ? 1 : 0;
        }
        q[i] = in;
        ((int32_t*)(q + cap))[i] = mq[i];
        ((int32_t*)(q + cap * 5))[i] = mt[i];
    }
    for (int i = threadIdx.x; i < nb; i += blockDim.x) ((float2*)(q + cap * 9))[i] = ((const float2*)xy_b)[i];
}

struct MonoTail { const int32_t* counts; const double* E; const float* F; float thr2; int min_n; };
static int mono_enqueue(vo_ctx* ctx, FrameSlot& a, FrameSlot& b, double ratio, const double* K4v, int iters, float thr, uint32_t seed,
                        int solver, MonoDev& o, MonoTail* tail = nullptr)
{
    const int min_n = solver == 5 ? 6 : 8;
    const int nq = a.n_kp;
    const size_t need = (size_t)iters * (72 + 36 + 4 + FP_REC_DOUBLES * 8) + (size_t)nq + 4096;
    if (ctx->mw->ransac_ws_bytes < need) {
        VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->mw->ransac_ws) (void)hipFree(ctx->mw->ransac_ws);
        ctx->mw->ransac_ws = nullptr; ctx->mw->ransac_ws_bytes = 0;
        VO_HIP(ctx, hipMalloc((void**)&ctx->mw->ransac_ws, need));
        ctx->mw->ransac_ws_bytes = need;
    }
    uint8_t* w = ctx->mw->ransac_ws;
    double* d_E = (double*)w; w += (size_t)iters * 72;
    double* d_rec = (double*)w; w += (size_t)iters * FP_REC_DOUBLES * 8;       // five-point solver: k_ransac_hyp5 -> k_ransac_roots5
    float* d_F = (float*)w; w += (size_t)iters * 36;
    int32_t* d_counts = (int32_t*)w; w += (size_t)iters * 4;
    o.d_best = (int32_t*)w; w += 256;
    o.d_E9 = (double*)w; w += 256;
    o.d_mask = w;
    int rc;
    {
        StageTimer t(ctx, VO_T_MATCH);
        if ((rc = match_knn2(ctx, a.desc, a.n_kp, b.desc, b.n_kp, ctx->mw->m_idx, ctx->mw->m_dist))) return rc;
        hipLaunchKernelGGL(k_ratio_compact, dim3(1), dim3(nq > 512 ? 1024 : 256), 0, ctx->stream, ctx->mw->m_idx, ctx->mw->m_dist, nq, ratio, a.kp_xy, b.kp_xy,
                           ctx->mw->mq_idx, ctx->mw->mt_idx, ctx->mw->xy_a, ctx->mw->xy_b, ctx->mw->m_count);
    }
    {
        StageTimer t(ctx, VO_T_POSE);
        const K4 K{ K4v[0], K4v[1], K4v[2], K4v[3] };
        const float thr2 = thr * thr;
        if (solver == 5) {
            if (int rc5 = hyp5_prepare(ctx)) return rc5;
            hipLaunchKernelGGL(k_ransac_hyp5, dim3(div_up(iters, 64)), dim3(64), FP_LDS_DOUBLES * sizeof(double), ctx->stream, ctx->mw->xy_a, ctx->mw->xy_b, nq, K, iters, seed, d_rec, ctx->mw->m_count);
            hipLaunchKernelGGL(k_ransac_roots5, dim3(div_up(iters, 8)), dim3(256), 0, ctx->stream, d_rec, K, iters, d_E, d_F);
        }
        else
            hipLaunchKernelGGL(k_ransac_hyp, dim3(div_up(iters, 64)), dim3(64), 0, ctx->stream, ctx->mw->xy_a, ctx->mw->xy_b, nq, K, iters, seed, d_E, d_F, ctx->mw->m_count);
        hipLaunchKernelGGL(k_ransac_score, dim3(div_up(iters, 4)), dim3(256), 0, ctx->stream, ctx->mw->xy_a, ctx->mw->xy_b, nq, d_F, iters, thr2, d_counts, ctx->mw->m_count, min_n);
        if (tail) {                                  // the asynchronous step ends with k_mono_finish (one launch, record written in place)
            tail->counts = d_counts; tail->E = d_E; tail->F = d_F; tail->thr2 = thr2; tail->min_n = min_n;
        } else {
            hipLaunchKernelGGL(k_ransac_best, dim3(1), dim3(1024), 0, ctx->stream, d_counts, iters, o.d_best);
            hipLaunchKernelGGL(k_ransac_mask, dim3(div_up(nq, 256)), dim3(256), 0, ctx->stream, ctx->mw->xy_a, ctx->mw->xy_b, nq, d_F, o.d_best, thr2, o.d_mask, ctx->mw->m_count, min_n);
            hipLaunchKernelGGL(k_ransac_pick, dim3(1), dim3(64), 0, ctx->stream, d_E, o.d_best, o.d_E9);
        }
        VO_CHECK_LAUNCH(ctx);
    }
    return VO_OK;
}

static int mono_check(vo_ctx* ctx, int slot_a, int slot_b, const double* K4v, int iters, int solver, const char* who)
{
    if (solver != 5 && solver != 8) return vo_fail(ctx, VO_E_ARG, "%s: solver is 5 (five-point) or 8 (eight-point)", who);
    if (!ctx || slot_a < 0 || slot_a >= VO_NUM_SLOTS || slot_b < 0 || slot_b >= VO_NUM_SLOTS || !K4v)
        return vo_fail(ctx, VO_E_ARG, "%s: bad argument", who);
    if (iters <= 0 || iters > (1 << 22)) return vo_fail(ctx, VO_E_ARG, "%s: need 0 < iters <= 4194304", who);
    FrameSlot& a = ctx->slots[slot_a];
    FrameSlot& b = ctx->slots[slot_b];
    if (!a.has_kp || !b.has_kp) return vo_fail(ctx, VO_E_STATE, "%s: both slots need keypoints (vo_orb_detect_and_compute)", who);
    if (a.n_kp > 0 && b.n_kp < 2) return vo_fail(ctx, VO_E_ARG, "train set has fewer than 2 descriptors");
    return VO_OK;
}

extern "C" int vo_mono_pair(vo_ctx* ctx, int slot_a, int slot_b, double ratio, const double* K4v, int iters, float thr, uint32_t seed,
                            int solver, double* E9_out, int32_t* counts3, uint8_t* mask_out, int32_t* q_idx, int32_t* t_idx, int cap)
{
    if (ctx && (!E9_out || !counts3)) return vo_fail(ctx, VO_E_ARG, "vo_mono_pair: bad argument");
    int rc = mono_check(ctx, slot_a, slot_b, K4v, iters, solver, "vo_mono_pair");
    if (rc) return rc;
    const int min_n = solver == 5 ? 6 : 8;
    FrameSlot& a = ctx->slots[slot_a];
    FrameSlot& b = ctx->slots[slot_b];
    counts3[0] = counts3[1] = counts3[2] = 0;
    for (int k = 0; k < 9; k++) E9_out[k] = 0.0;
    if (a.n_kp == 0) return VO_OK;
    if ((mask_out || q_idx || t_idx) && cap < a.n_kp) return vo_fail(ctx, VO_E_CAP, "vo_mono_pair: outputs hold %d entries, %d keypoints", cap, a.n_kp);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    { int rcw = slot_wait(ctx, a); if (!rcw) rcw = slot_wait(ctx, b); if (rcw) return rcw; }
    const int nq = a.n_kp;
    MonoDev o;
    if ((rc = mono_enqueue(ctx, a, b, ratio, K4v, iters, thr, seed, solver, o))) return rc;
    int32_t* h = (int32_t*)ctx->pinned;         // [0] M, [1..2] best, then E9 at byte 64
    VO_HIP(ctx, hipMemcpyAsync(h, ctx->mw->m_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(ctx, hipMemcpyAsync(h + 1, o.d_best, 8, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(ctx, hipMemcpyAsync((uint8_t*)ctx->pinned + 64, o.d_E9, 72, hipMemcpyDeviceToHost, ctx->stream));
    if (mask_out && (rc = xfer_d2h(ctx, mask_out, o.d_mask, (size_t)nq))) return rc;
    if (q_idx && (rc = xfer_d2h(ctx, q_idx, ctx->mw->mq_idx, (size_t)nq * 4))) return rc;
    if (t_idx && (rc = xfer_d2h(ctx, t_idx, ctx->mw->mt_idx, (size_t)nq * 4))) return rc;
    if ((rc = xfer_flush(ctx))) return rc;       // the one synchronisation
    counts3[0] = h[0]; counts3[1] = h[1]; counts3[2] = h[0] >= min_n ? h[2] : 0;
    memcpy(E9_out, (uint8_t*)ctx->pinned + 64, 72);
    return VO_OK;
}

// ---- the same step, asynchronous: several pairs' chains in flight (each on an alternate's own stream and scratch) ----------
// A monocular stream is latency-bound when every pair is enqueued, waited for and only then followed by the next one (the
// five-point kernel alone runs 0.26 ms on 79 of the 1024 SIMDs).  Consecutive pairs do not depend on each other's result --
// only on the caller's decision which frame is the reference -- so a caller that knows the next frame may begin its pair
// before it collects this one's.  Results land in the alternate's pinned record; vo_mono_pair_end waits for its event only.
static const size_t MONO_HDR = 4096;             // record: [0] M, [1..2] best, E9 at byte 64; arrays from MONO_HDR on

// the context works on alternate k's stream and in its match / RANSAC scratch for the lifetime of the object
struct MonoScope {
    vo_ctx* c;
    int k;
    MonoScope(vo_ctx* c_, int k_) : c(c_), k(k_)
    {
        std::swap(c->stream, c->mono_alt[k].stream);
        c->mw = &c->mono_alt[k].mw;
    }
    ~MonoScope()
    {
        c->mw = &c->main_mw;
        std::swap(c->stream, c->mono_alt[k].stream);
    }
    MonoScope(const MonoScope&) = delete;
    MonoScope& operator=(const MonoScope&) = delete;
};

static void mono_alt_release(vo_ctx::MonoAlt& p)
{
    if (p.stream) (void)hipStreamSynchronize(p.stream);
    void* ps[] = { p.mw.m_idx, p.mw.m_dist, p.mw.m_count, p.mw.mq_idx, p.mw.mt_idx, p.mw.xy_a, p.mw.xy_b, p.mw.ransac_ws };
    for (void* q : ps) if (q) (void)hipFree(q);
    if (p.result) (void)hipHostFree(p.result);
    if (p.done) (void)hipEventDestroy(p.done);
    if (p.stream) (void)hipStreamDestroy(p.stream);
    p = vo_ctx::MonoAlt();
}

static int mono_alt_prepare(vo_ctx* ctx, int k)
{
    vo_ctx::MonoAlt& p = ctx->mono_alt[k];
    if (p.ready) return VO_OK;
    const size_t cap = (size_t)ctx->kp_cap;
    hipError_t e = hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p.done, hipEventDisableTiming);
    p.result_bytes = MONO_HDR + cap * (1 + 4 + 4 + 8) + 64;
    if (e == hipSuccess) e = hipHostMalloc((void**)&p.result, p.result_bytes, hipHostMallocDefault);
    void** ps[] = { (void**)&p.mw.m_idx, (void**)&p.mw.m_count, (void**)&p.mw.mq_idx, (void**)&p.mw.mt_idx, (void**)&p.mw.xy_a, (void**)&p.mw.xy_b };
    const size_t sz[] = { cap * 8, 256, cap * 4, cap * 4, cap * 8, cap * 8 };
    for (size_t i = 0; i < sizeof(ps) / sizeof(ps[0]) && e == hipSuccess; i++) e = hipMalloc(ps[i], sz[i] + 256);
    if (e == hipSuccess && match_dist_alloc(ctx, &p.mw.m_dist)) e = hipErrorOutOfMemory;
    if (e != hipSuccess) {
        mono_alt_release(p);                      // a partly built alternate is given back whole
        return vo_fail(ctx, VO_E_HIP, "asynchronous monocular step: allocation failed: %s", hipGetErrorString(e));
    }
    p.ready = true;
    return VO_OK;
}

void mono_alt_free(vo_ctx* ctx)
{
    for (int k = 0; k < vo_ctx::N_MONO_ALT; k++) mono_alt_release(ctx->mono_alt[k]);
}

extern "C" int vo_mono_pair_begin(vo_ctx* ctx, int slot_a, int slot_b, double ratio, const double* K4v, int iters, float thr, uint32_t seed,
                                  int solver, int want_matches, int* ticket_out)
{
    if (ctx && !ticket_out) return vo_fail(ctx, VO_E_ARG, "vo_mono_pair_begin: bad argument");
    int rc = mono_check(ctx, slot_a, slot_b, K4v, iters, solver, "vo_mono_pair_begin");
    if (rc) return rc;
    VO_HIP(ctx, hipSetDevice(ctx->device));
    int k = -1;                                   // the first free alternate from the round-robin position on (tickets need not end in order)
    for (int i = 0; i < vo_ctx::N_MONO_ALT && k < 0; i++)
        if (!ctx->mono_alt[(ctx->mono_next + i) % vo_ctx::N_MONO_ALT].busy) k = (ctx->mono_next + i) % vo_ctx::N_MONO_ALT;
    if (k < 0) return vo_fail(ctx, VO_E_STATE, "vo_mono_pair_begin: every asynchronous step is still open (end one first)");
    vo_ctx::MonoAlt& p = ctx->mono_alt[k];
    for (int i = 0; i < vo_ctx::N_MONO_ALT; i++)            // (every alternate with the first step: see vo_pose_pair_begin)
        if ((rc = mono_alt_prepare(ctx, (k + i) % vo_ctx::N_MONO_ALT))) return rc;
    FrameSlot& a = ctx->slots[slot_a];
    FrameSlot& b = ctx->slots[slot_b];
    memset(p.result, 0, MONO_HDR);
    p.nq = a.n_kp; p.nb = b.n_kp; p.min_n = solver == 5 ? 6 : 8; p.want = want_matches != 0;
    // the step runs on the alternate's own stream: behind whatever still produces the two slots (look-ahead engines) and
    // behind the main stream's work on them
    VO_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    {
        MonoScope on_alt(ctx, k);
        hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->ev0, 0);
        if (e == hipSuccess && a.pending) e = hipStreamWaitEvent(ctx->stream, a.ready, 0);
        if (e == hipSuccess && b.pending) e = hipStreamWaitEvent(ctx->stream, b.ready, 0);
        rc = e == hipSuccess ? VO_OK : vo_fail(ctx, VO_E_HIP, "hipStreamWaitEvent failed: %s", hipGetErrorString(e));
        if (!rc && a.n_kp > 0) {
            MonoDev o;
            MonoTail tl;
            rc = mono_enqueue(ctx, a, b, ratio, K4v, iters, thr, seed, solver, o, &tl);
            if (!rc) {
                // (p.result is pinned host memory: the kernel writes the record across the link itself -- no copy command)
                hipLaunchKernelGGL(k_mono_finish, dim3(1), dim3(1024), 0, ctx->stream, tl.counts, iters, tl.E, tl.F, ctx->mw->xy_a, ctx->mw->xy_b, a.n_kp,
                                   ctx->mw->m_count, tl.min_n, tl.thr2, ctx->mw->mq_idx, ctx->mw->mt_idx, b.kp_xy, b.n_kp, p.want ? 1 : 0, (size_t)ctx->kp_cap,
                                   p.result, MONO_HDR);
                if (hipGetLastError() != hipSuccess) rc = vo_fail(ctx, VO_E_HIP, "k_mono_finish: launch failed");
            }
        }
        if (!rc && hipEventRecord(p.done, ctx->stream) != hipSuccess) rc = vo_fail(ctx, VO_E_HIP, "hipEventRecord failed");
    }
    if (rc) return rc;
    // both slots are read on this alternate's stream until p.done: whoever refills one of them waits for it first
    for (FrameSlot* f : { &a, &b }) {
        slot_add_reader(*f, p.done);
    }
    p.busy = true;
    ctx->mono_next = (k + 1) % vo_ctx::N_MONO_ALT;
    *ticket_out = k;
    return VO_OK;
}

extern "C" int vo_mono_pair_end(vo_ctx* ctx, int ticket, double* E9_out, int32_t* counts3, uint8_t* mask_out, int32_t* q_idx, int32_t* t_idx,
                                float* xy_b_out, int cap)
{
    if (!ctx || ticket < 0 || ticket >= vo_ctx::N_MONO_ALT || !E9_out || !counts3) return vo_fail(ctx, VO_E_ARG, "vo_mono_pair_end: bad argument");
    vo_ctx::MonoAlt& p = ctx->mono_alt[ticket];
    if (!p.busy) return vo_fail(ctx, VO_E_STATE, "vo_mono_pair_end: ticket %d is not open", ticket);
    if ((mask_out || q_idx || t_idx) && (!p.want || cap < p.nq)) return vo_fail(ctx, VO_E_CAP, "vo_mono_pair_end: outputs hold %d entries, %d keypoints (or the step was begun without want_matches)", cap, p.nq);
    if (xy_b_out && (!p.want || cap < p.nb)) return vo_fail(ctx, VO_E_CAP, "vo_mono_pair_end: xy_b_out holds %d entries, %d keypoints", cap, p.nb);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    p.busy = false;
    VO_HIP(ctx, hipEventSynchronize(p.done));
    const int32_t* h = (const int32_t*)p.result;
    counts3[0] = h[0]; counts3[1] = h[1]; counts3[2] = h[0] >= p.min_n ? h[2] : 0;
    memcpy(E9_out, p.result + 64, 72);
    const uint8_t* q = p.result + MONO_HDR;
    const size_t kc = (size_t)ctx->kp_cap;
    if (mask_out) memcpy(mask_out, q, (size_t)p.nq);
    if (q_idx) memcpy(q_idx, q + kc, (size_t)p.nq * 4);
    if (t_idx) memcpy(t_idx, q + kc * 5, (size_t)p.nq * 4);
    if (xy_b_out) memcpy(xy_b_out, q + kc * 9, (size_t)p.nb * 8);
    return VO_OK;
}

// =========================================================================================
// RANSAC solvePnP hypothesis generation + reprojection scoring (the north star's "solvePnP
// hypothesis-scoring loop").  No openVO counterpart either; same structure as above:
//   k_pnp_hyp     one lane per hypothesis: 4 hash-sampled correspondences, P3P on three of them in
//                 float64 (depths along the bearings; singular member of the two-conic pencil by a
//                 bisected cubic root -- only + - * / sqrt; <= 4 poses, Gauss-Newton polish), the fourth
//                 picks the pose; P = K [R|t] rounded to float32
//   k_pnp_score   one wave per hypothesis; division-free reprojection test in float32
//                 ((xc - u zc)^2 + (yc - v zc)^2 < thr^2 zc^2, zc > 0), ballot + popcount
// =========================================================================================
__device__ __forceinline__ double pn_det3(const double m[3][3])
{
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

__device__ double pn_det3_col(const double A[3][3], const double B[3][3], int c)
{
    double m[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = j == c ? B[i][j] : A[i][j];
    return pn_det3(m);
}

__device__ __forceinline__ double pn_cubic(const double* c, double g) { return ((c[3] * g + c[2]) * g + c[1]) * g + c[0]; }

__device__ double pn_cubic_root(const double* c)
{
    double m = fabs(c[2]);
    if (fabs(c[1]) > m) m = fabs(c[1]);
    if (fabs(c[0]) > m) m = fabs(c[0]);
    double hi = 1.0 + m / fabs(c[3]), lo = -hi;
    double flo = pn_cubic(c, lo);
    for (int it = 0; it < 80; it++) {
        const double mid = 0.5 * (lo + hi), fm = pn_cubic(c, mid);
        if ((fm < 0.0) == (flo < 0.0)) { lo = mid; flo = fm; } else hi = mid;
    }
    double g = 0.5 * (lo + hi);
    for (int it = 0; it < 2; it++) {
        const double d = (3.0 * c[3] * g + 2.0 * c[2]) * g + c[1];
        if (d != 0.0) g -= pn_cubic(c, g) / d;
    }
    return g;
}

__device__ bool pn_null_vec(const double M[3][3], double s, double* e)
{
    double r[3][3], c[3][3], best = -1.0;
    int bi = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i][j] = M[i][j] - (i == j ? s : 0.0);
    rs_cross3(r[0], r[1], c[0]);
    rs_cross3(r[0], r[2], c[1]);
    rs_cross3(r[1], r[2], c[2]);
    for (int k = 0; k < 3; k++) {
        const double n2 = c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2];
        if (n2 > best) { best = n2; bi = k; }
    }
    if (!(best > 0.0)) return false;
    const double inv = 1.0 / sqrt(best);
    for (int k = 0; k < 3; k++) e[k] = c[bi][k] * inv;
    return true;
}

// y: 3 unit bearings, x: 3 points (rows); Rs/ts receive up to 4 poses
__device__ int pn_p3p(const double* y, const double* x, double* Rs, double* ts)
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
    rs_cross3(d12, d13, dx);
    const double area2 = dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2];
    if (!(a12 > 0.0) || !(a13 > 0.0) || !(a23 > 0.0) || !(area2 > 1e-24 * a12 * a13)) return 0;

    const double D1[3][3] = { { a23, -a23 * b12, 0.0 }, { -a23 * b12, a23 - a12, a12 * b23 }, { 0.0, a12 * b23, -a12 } };
    const double D2[3][3] = { { a23, 0.0, -a23 * b13 }, { 0.0, -a13, a13 * b23 }, { -a23 * b13, a13 * b23, a23 - a13 } };
    double c[4];
    c[0] = pn_det3(D1);
    c[3] = pn_det3(D2);
    c[1] = (pn_det3_col(D1, D2, 0) + pn_det3_col(D1, D2, 1)) + pn_det3_col(D1, D2, 2);
    c[2] = (pn_det3_col(D2, D1, 0) + pn_det3_col(D2, D1, 1)) + pn_det3_col(D2, D1, 2);
    double D0[3][3];
    if (fabs(c[3]) >= fabs(c[0])) {
        if (c[3] == 0.0) return 0;
        const double g = pn_cubic_root(c);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D0[i][j] = D1[i][j] + g * D2[i][j];
    } else {
        const double cr[4] = { c[3], c[2], c[1], c[0] };
        const double g = pn_cubic_root(cr);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D0[i][j] = g * D1[i][j] + D2[i][j];
    }
    const double tr = (D0[0][0] + D0[1][1]) + D0[2][2];
    const double m2 = ((D0[0][0] * D0[1][1] - D0[0][1] * D0[0][1]) + (D0[0][0] * D0[2][2] - D0[0][2] * D0[0][2])) +
                      (D0[1][1] * D0[2][2] - D0[1][2] * D0[1][2]);
    if (!(m2 < 0.0)) return 0;
    const double disc = tr * tr - 4.0 * m2;
    const double s1 = 0.5 * (tr + (tr >= 0.0 ? 1.0 : -1.0) * sqrt(disc));
    const double s2 = m2 / s1;
    double e1[3], e2[3];
    if (!pn_null_vec(D0, s1, e1) || !pn_null_vec(D0, s2, e2)) return 0;
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
            for (int it = 0; it < 3; it++) {
                const double r0 = ((l1 * l1 + l2 * l2) - 2.0 * b12 * l1 * l2) - a12;
                const double r1 = ((l1 * l1 + l3 * l3) - 2.0 * b13 * l1 * l3) - a13;
                const double r2 = ((l2 * l2 + l3 * l3) - 2.0 * b23 * l2 * l3) - a23;
                const double J[3][3] = { { 2.0 * l1 - 2.0 * b12 * l2, 2.0 * l2 - 2.0 * b12 * l1, 0.0 },
                                         { 2.0 * l1 - 2.0 * b13 * l3, 0.0, 2.0 * l3 - 2.0 * b13 * l1 },
                                         { 0.0, 2.0 * l2 - 2.0 * b23 * l3, 2.0 * l3 - 2.0 * b23 * l2 } };
                const double dj = pn_det3(J);
                if (dj == 0.0) break;
                const double rr[3][3] = { { r0, r0, r0 }, { r1, r1, r1 }, { r2, r2, r2 } };
                const double n0 = pn_det3_col(J, rr, 0), n1 = pn_det3_col(J, rr, 1), n2 = pn_det3_col(J, rr, 2);
                l1 -= n0 / dj; l2 -= n1 / dj; l3 -= n2 / dj;
            }
            if (!(l1 > 0.0) || !(l2 > 0.0) || !(l3 > 0.0)) continue;
            double ya[3], yb[3], yc[3];
            for (int k = 0; k < 3; k++) { ya[k] = l1 * y1[k] - l2 * y2[k]; yb[k] = l1 * y1[k] - l3 * y3[k]; }
            rs_cross3(ya, yb, yc);
            const double X[3][3] = { { d12[0], d13[0], dx[0] }, { d12[1], d13[1], dx[1] }, { d12[2], d13[2], dx[2] } };
            const double dX = pn_det3(X);
            if (dX == 0.0) continue;
            double Xi[3][3];
            Xi[0][0] = (X[1][1] * X[2][2] - X[1][2] * X[2][1]) / dX;
            Xi[0][1] = (X[0][2] * X[2][1] - X[0][1] * X[2][2]) / dX;
            Xi[0][2] = (X[0][1] * X[1][2] - X[0][2] * X[1][1]) / dX;
            Xi[1][0] = (X[1][2] * X[2][0] - X[1][0] * X[2][2]) / dX;
            Xi[1][1] = (X[0][0] * X[2][2] - X[0][2] * X[2][0]) / dX;
            Xi[1][2] = (X[0][2] * X[1][0] - X[0][0] * X[1][2]) / dX;
            Xi[2][0] = (X[1][0] * X[2][1] - X[1][1] * X[2][0]) / dX;
            Xi[2][1] = (X[0][1] * X[2][0] - X[0][0] * X[2][1]) / dX;
            Xi[2][2] = (X[0][0] * X[1][1] - X[0][1] * X[1][0]) / dX;
            double* R = Rs + 9 * ns;
            double* tt = ts + 3 * ns;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) R[i * 3 + j] = (ya[i] * Xi[0][j] + yb[i] * Xi[1][j]) + yc[i] * Xi[2][j];
            for (int i = 0; i < 3; i++) tt[i] = l1 * y1[i] - ((R[i * 3] * x1[0] + R[i * 3 + 1] * x1[1]) + R[i * 3 + 2] * x1[2]);
            ns++;
        }
    }
    return ns;
}

__global__ void __launch_bounds__(64) k_pnp_hyp(const float* __restrict__ X, const float* __restrict__ uv, int n, K4 K, int iters,
                                                uint32_t seed, double* __restrict__ Rt_out, float* __restrict__ P_out)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= iters) return;
    int idx[4];
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
    double y[9], x[9], Rs[36], ts[12];
    for (int s = 0; s < 3; s++) {
        const int i = idx[s];
        const double a = ((double)uv[2 * i] - K.cx) / K.fx, b = ((double)uv[2 * i + 1] - K.cy) / K.fy;
        const double inv = 1.0 / sqrt((a * a + b * b) + 1.0);
        y[3 * s] = a * inv; y[3 * s + 1] = b * inv; y[3 * s + 2] = inv;
        for (int k = 0; k < 3; k++) x[3 * s + k] = (double)X[3 * i + k];
    }
    const int ns = pn_p3p(y, x, Rs, ts);
    const int i4 = idx[3];
    const double u4 = ((double)uv[2 * i4] - K.cx) / K.fx, v4 = ((double)uv[2 * i4 + 1] - K.cy) / K.fy;
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
    double Rt[12];
    float P[12];
    if (best < 0) {
        for (int k = 0; k < 12; k++) { Rt[k] = 0.0; P[k] = 0.0f; }
    } else {
        const double* R = Rs + 9 * best;
        const double* t = ts + 3 * best;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rt[r * 4 + c] = R[r * 3 + c]; Rt[r * 4 + 3] = t[r]; }
        for (int c = 0; c < 4; c++) {
            P[c] = (float)(K.fx * Rt[c] + K.cx * Rt[8 + c]);
            P[4 + c] = (float)(K.fy * Rt[4 + c] + K.cy * Rt[8 + c]);
            P[8 + c] = (float)Rt[8 + c];
        }
    }
    for (int k = 0; k < 12; k++) { Rt_out[(size_t)h * 12 + k] = Rt[k]; P_out[(size_t)h * 12 + k] = P[k]; }
}

__device__ __forceinline__ bool reproj_inlier(const float* P, float X, float Y, float Z, float u, float v, float thr2)
{
    const float xc = ((P[0] * X + P[1] * Y) + P[2] * Z) + P[3];
    const float yc = ((P[4] * X + P[5] * Y) + P[6] * Z) + P[7];
    const float zc = ((P[8] * X + P[9] * Y) + P[10] * Z) + P[11];
    const float du = xc - u * zc, dv = yc - v * zc;
    const float e = du * du + dv * dv;
    const float lim = thr2 * (zc * zc);
    return zc > 0.0f && e < lim;
}

// one wave per hypothesis
__global__ void __launch_bounds__(256) k_pnp_score(const float* __restrict__ X, const float* __restrict__ uv, int n,
                                                  const float* __restrict__ P_all, int iters, float thr2, int32_t* __restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (h >= iters) return;
    float P[12];
#pragma unroll
    for (int k = 0; k < 12; k++) P[k] = P_all[(size_t)h * 12 + k];
    int cnt = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        bool in = false;
        if (i < n) {
            const float2 b = ((const float2*)uv)[i];
            in = reproj_inlier(P, X[3 * i], X[3 * i + 1], X[3 * i + 2], b.x, b.y, thr2);
        }
        cnt += __popcll(__ballot(in));
    }
    if (lane == 0) counts[h] = cnt;
}

__global__ void k_pnp_mask(const float* __restrict__ X, const float* __restrict__ uv, int n, const float* __restrict__ P_all,
                           const int32_t* __restrict__ best, float thr2, uint8_t* __restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float P[12];
    for (int k = 0; k < 12; k++) P[k] = P_all[(size_t)best[0] * 12 + k];
    const float2 b = ((const float2*)uv)[i];
    mask[i] = reproj_inlier(P, X[3 * i], X[3 * i + 1], X[3 * i + 2], b.x, b.y, thr2) ? 1 : 0;
}

extern "C" int vo_ransac_pnp(vo_ctx* ctx, const float* pts3d, const float* pts2d, int n, const double* K4v, int iters, float thr,
                             uint32_t seed, double* Rt12_out, uint8_t* mask_out, int32_t* counts_out, int32_t* best2_out)
{
    if (!ctx || !pts3d || !pts2d || !K4v || !Rt12_out || !best2_out) return vo_fail(ctx, VO_E_ARG, "vo_ransac_pnp: bad argument");
    if (n < 4 || iters <= 0 || iters > (1 << 22) || n > (1 << 24)) return vo_fail(ctx, VO_E_ARG, "vo_ransac_pnp: need n >= 4 and 0 < iters <= 4194304");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    // workspace: Rt (iters x 96 B), points (n x 12 B + n x 8 B), P (iters x 48 B), counts, best, mask
    const size_t need = (size_t)n * 24 + (size_t)iters * (96 + 48 + 4) + (size_t)n + 4096;
    if (ctx->mw->ransac_ws_bytes < need) {
        if (ctx->mw->ransac_ws) (void)hipFree(ctx->mw->ransac_ws);
        ctx->mw->ransac_ws = nullptr; ctx->mw->ransac_ws_bytes = 0;
        VO_HIP(ctx, hipMalloc((void**)&ctx->mw->ransac_ws, need));
        ctx->mw->ransac_ws_bytes = need;
    }
    uint8_t* w = ctx->mw->ransac_ws;
    double* d_Rt = (double*)w; w += (size_t)iters * 96;
    float* d_uv = (float*)w; w += (size_t)n * 8;
    float* d_X = (float*)w; w += (size_t)n * 12 + 8;
    w = (uint8_t*)(((uintptr_t)w + 15) & ~(uintptr_t)15);
    float* d_P = (float*)w; w += (size_t)iters * 48;
    int32_t* d_counts = (int32_t*)w; w += (size_t)iters * 4;
    int32_t* d_best = (int32_t*)w; w += 256;
    uint8_t* d_mask = w;
    StageTimer t(ctx, VO_T_POSE);
    int rc = xfer_h2d(ctx, d_X, pts3d, (size_t)n * 12);
    if (!rc) rc = xfer_h2d(ctx, d_uv, pts2d, (size_t)n * 8);
    if (rc) return rc;
    const K4 K{ K4v[0], K4v[1], K4v[2], K4v[3] };
    const float thr2 = thr * thr;
    hipLaunchKernelGGL(k_pnp_hyp, dim3(div_up(iters, 64)), dim3(64), 0, ctx->stream, d_X, d_uv, n, K, iters, seed, d_Rt, d_P);
    hipLaunchKernelGGL(k_pnp_score, dim3(div_up(iters, 4)), dim3(256), 0, ctx->stream, d_X, d_uv, n, d_P, iters, thr2, d_counts);
    hipLaunchKernelGGL(k_ransac_best, dim3(1), dim3(1024), 0, ctx->stream, d_counts, iters, d_best);
    hipLaunchKernelGGL(k_pnp_mask, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_X, d_uv, n, d_P, d_best, thr2, d_mask);
    VO_CHECK_LAUNCH(ctx);
    VO_HIP(ctx, hipMemcpyAsync(ctx->pinned, d_best, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (mask_out && (rc = xfer_d2h(ctx, mask_out, d_mask, (size_t)n))) return rc;
    if (counts_out && (rc = xfer_d2h(ctx, counts_out, d_counts, (size_t)iters * 4))) return rc;
    if ((rc = xfer_flush(ctx))) return rc;
    best2_out[0] = ((int32_t*)ctx->pinned)[0];
    best2_out[1] = ((int32_t*)ctx->pinned)[1];
    VO_HIP(ctx, hipMemcpy(Rt12_out, d_Rt + (size_t)best2_out[0] * 12, 96, hipMemcpyDeviceToHost));
    return VO_OK;
}
