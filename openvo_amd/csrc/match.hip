// Brute-force Hamming kNN (k = 2) on gfx950: replaces
// self.matcher.knnMatch(desc1, desc2, k=2) with cv2.NORM_HAMMING [reference stereo_odometer.py:22,163].
// Ordering rule of OpenCV's batch_distance.cpp: ascending distance, ties -> lower train index,
// which is exactly the lexicographic minimum of (distance, trainIdx).
//
// One wave per query descriptor.  The 64 lanes each take one train descriptor per round
// (32-byte rows, two coalesced 16-byte loads), XOR + v_bcnt popcount, and keep a private
// best/second key = distance << 16 | trainIdx; a DPP wave-min then gives the global best, the
// winner lane is retired to its second key and a second wave-min gives the runner-up.
#include "vo_internal.h"

__device__ __forceinline__ uint32_t wave_min_u32_m(uint32_t v)
{
#define DPP_MIN(ctrl, rmask) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false))
    DPP_MIN(0x111, 0xf);
    DPP_MIN(0x112, 0xf);
    DPP_MIN(0x114, 0xf);
    DPP_MIN(0x118, 0xf);
    DPP_MIN(0x142, 0xa);
    DPP_MIN(0x143, 0xc);
#undef DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ void __launch_bounds__(256) k_bf_knn2(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                                int32_t* __restrict__ idx, int32_t* __restrict__ dist)
{
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const uint4* qp = (const uint4*)(q + (size_t)qi * 32);
    const uint4 qa = qp[0], qb = qp[1];
    uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;  // best, second (private to the lane)
    for (int j = lane; j < nt; j += 64) {
        const uint4* tp = (const uint4*)(t + (size_t)j * 32);
        const uint4 ta = tp[0], tb = tp[1];
        uint32_t d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                     __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
        uint32_t key = (d << 16) | (uint32_t)j;
        if (key < k0) { k1 = k0; k0 = key; }
        else if (key < k1) k1 = key;
    }
    const uint32_t b0 = wave_min_u32_m(k0);
    // retire the winner: its lane now offers its private runner-up
    const uint32_t mine = (k0 == b0) ? k1 : k0;
    const uint32_t b1 = wave_min_u32_m(mine);
    if (lane == 0) {
        idx[2 * qi] = b0 == 0xFFFFFFFFu ? -1 : (int32_t)(b0 & 0xFFFFu);
        dist[2 * qi] = b0 == 0xFFFFFFFFu ? 0x7FFFFFFF : (int32_t)(b0 >> 16);
        idx[2 * qi + 1] = b1 == 0xFFFFFFFFu ? -1 : (int32_t)(b1 & 0xFFFFu);
        dist[2 * qi + 1] = b1 == 0xFFFFFFFFu ? 0x7FFFFFFF : (int32_t)(b1 >> 16);
    }
}

int match_knn2(vo_ctx* ctx, const uint8_t* dq, int nq, const uint8_t* dt, int nt, int32_t* d_idx, int32_t* d_dist)
{
    if (nt > 65535) return vo_fail(ctx, VO_E_CAP, "train set of %d descriptors exceeds 65535", nt);
    if (nq <= 0) return VO_OK;
    hipLaunchKernelGGL(k_bf_knn2, dim3(div_up(nq, 4)), dim3(256), 0, ctx->stream, dq, nq, dt, nt, d_idx, d_dist);
    VO_CHECK_LAUNCH(ctx);
    return VO_OK;
}

extern "C" int vo_bf_knn2_hamming(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx,
                                  int32_t* dist)
{
    if (!ctx || nq < 0 || nt < 0 || (nq && (!q || !idx || !dist)) || (nt && !t)) return vo_fail(ctx, VO_E_ARG, "vo_bf_knn2_hamming: bad argument");
    if (nq > ctx->kp_cap || nt > ctx->kp_cap) return vo_fail(ctx, VO_E_CAP, "descriptor count exceeds capacity %d", ctx->kp_cap);
    if (nq == 0) return VO_OK;
    VO_HIP(ctx, hipSetDevice(ctx->device));
    StageTimer tm(ctx, VO_T_MATCH);
    int rc = xfer_h2d(ctx, ctx->mq, q, (size_t)nq * 32);
    if (!rc && nt) rc = xfer_h2d(ctx, ctx->mt, t, (size_t)nt * 32);
    if (rc) return rc;
    rc = match_knn2(ctx, ctx->mq, nq, ctx->mt, nt, ctx->mw->m_idx, ctx->mw->m_dist);
    if (rc) return rc;
    rc = xfer_d2h(ctx, idx, ctx->mw->m_idx, (size_t)nq * 8);
    if (!rc) rc = xfer_d2h(ctx, dist, ctx->mw->m_dist, (size_t)nq * 8);
    if (rc) return rc;
    return xfer_flush(ctx);
}
