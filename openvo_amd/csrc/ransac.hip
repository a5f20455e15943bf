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

__global__ void __launch_bounds__(64) k_ransac_hyp(const float* __restrict__ p1, const float* __restrict__ p2, int n, K4 K,
                                                   int iters, uint32_t seed, double* __restrict__ E_out, float* __restrict__ F_out)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= iters) return;
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
    // F = K^-T E K^-1, scaled to max |entry| = 1, rounded to float32
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
                                                     const float* __restrict__ F_all, int iters, float thr2, int32_t* __restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (h >= iters) return;
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
                              const int32_t* __restrict__ best, float thr2, uint8_t* __restrict__ mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* F = F_all + (size_t)best[0] * 9;
    float Fl[9];
    for (int k = 0; k < 9; k++) Fl[k] = F[k];
    const float2 a = ((const float2*)p1)[i], b = ((const float2*)p2)[i];
    mask[i] = sampson_inlier(Fl, a.x, a.y, b.x, b.y, thr2) ? 1 : 0;
}

extern "C" int vo_ransac_essential(vo_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K4v, int iters, float thr,
                                   uint32_t seed, double* E9_out, uint8_t* mask_out, int32_t* counts_out, int32_t* best2_out)
{
    if (!ctx || !pts1 || !pts2 || !K4v || !E9_out || !best2_out) return vo_fail(ctx, VO_E_ARG, "vo_ransac_essential: bad argument");
    if (n < 8 || iters <= 0 || iters > (1 << 22) || n > (1 << 24)) return vo_fail(ctx, VO_E_ARG, "vo_ransac_essential: need n >= 8 and 0 < iters <= 4194304");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    // workspace: points (2 x n x 8 B), E (iters x 72 B), F (iters x 36 B), counts, mask, best
    const size_t need = (size_t)n * 16 + (size_t)iters * (72 + 36 + 4) + (size_t)n + 4096;
    if (ctx->ransac_ws_bytes < need) {
        if (ctx->ransac_ws) (void)hipFree(ctx->ransac_ws);
        ctx->ransac_ws = nullptr; ctx->ransac_ws_bytes = 0;
        VO_HIP(ctx, hipMalloc((void**)&ctx->ransac_ws, need));
        ctx->ransac_ws_bytes = need;
    }
    uint8_t* w = ctx->ransac_ws;
    double* d_E = (double*)w; w += (size_t)iters * 72;
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
    hipLaunchKernelGGL(k_ransac_hyp, dim3(div_up(iters, 64)), dim3(64), 0, ctx->stream, d_p1, d_p2, n, K, iters, seed, d_E, d_F);
    hipLaunchKernelGGL(k_ransac_score, dim3(div_up(iters, 4)), dim3(256), 0, ctx->stream, d_p1, d_p2, n, d_F, iters, thr2, d_counts);
    hipLaunchKernelGGL(k_ransac_best, dim3(1), dim3(1024), 0, ctx->stream, d_counts, iters, d_best);
    hipLaunchKernelGGL(k_ransac_mask, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_p1, d_p2, n, d_F, d_best, thr2, d_mask);
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
