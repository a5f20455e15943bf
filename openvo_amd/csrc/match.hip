// Brute-force Hamming kNN (k = 2) on gfx950: replaces
// self.matcher.knnMatch(desc1, desc2, k=2) with cv2.NORM_HAMMING [reference stereo_odometer.py:22,163].
// Ordering rule of OpenCV's batch_distance.cpp: ascending distance, ties -> lower train index,
// which is exactly the lexicographic minimum of (distance, trainIdx).
//
//
// Round 5: the distance table as an integer-exact dense contraction on the matrix cores.  popcount(q ^ t) = |q| + |t| - 2 q.t
// over {0,1} vectors, and q.t over 256 bits is four v_mfma_i32_16x16x64_i8 on bits expanded to bytes -- exact in int32 (the
// one stage of this path that IS a dense contraction: 8000 x 8000 x 256 bit = 1.6e10 multiply-adds at config 5).
//   * a wave owns 64 queries (four 16-row A operands, expanded once into 64 registers) and walks a slice of the train set in
//     tiles of 16 descriptors.  Bits become bytes by v_bfe, v_mul_u32_u24 by 0x204081, v_and (a nibble -> four bytes); the
//     queries' bytes are {0, -1} so that the accumulators hold -(q.t).  The eight waves of a workgroup share ONE expanded copy
//     of their slice in LDS (32 tiles = 128 KB at a time): per tile a wave issues four ds_read_b128, sixteen MFMAs, and per
//     accumulator register key = ((|t| + 256) << 16 | j) + (-(q.t) << 17) (one v_lshl_add_u32) and a
//     private best / second (v_med3_u32, v_min_u32).  |q| is the same for every candidate of a query, so it is added when the
//     result is written.  The k order inside the MFMA is irrelevant: both operands cut a descriptor the same way (lane group
//     g = lane >> 4 takes bits 64 g .. 64 g + 63, MFMA m its m-th 16), and a dot product does not care in which order it sums.
//   * key order = (distance, train index) lexicographic = batch_distance.cpp's "ascending, ties to the lower index".
//   * a query's candidates sit in the 16 lanes of a DPP row (the C layout puts the train on lane & 15): four row_ror butterfly
//     steps merge the private pairs once per wave.
//   * the train set is cut into up to 16 slices (about 2000 waves at 8000 x 8000: two per SIMD); a slice's pairs go to scratch
//     behind the distances, the wave that draws a group's last ticket merges them (agent-scope release / acquire around the
//     ticket) and writes idx / dist.  One launch; 4 MB of L2 traffic at config 5 (every workgroup reads its slice once).
#include "vo_internal.h"

typedef int v4i __attribute__((ext_vector_type(4)));

// bits 4 i .. 4 i + 3 of x -> four bytes of {0, 1} (bit k of the nibble in byte k)
#define NIB(x, i) ((__umul24(__builtin_amdgcn_ubfe((x), 4 * (i), 4), 0x00204081u)) & 0x01010101u)
__device__ __forceinline__ void expand64(uint32_t lo, uint32_t hi, v4i (&o)[4])
{
    o[0] = (v4i){ (int)NIB(lo, 0), (int)NIB(lo, 1), (int)NIB(lo, 2), (int)NIB(lo, 3) };
    o[1] = (v4i){ (int)NIB(lo, 4), (int)NIB(lo, 5), (int)NIB(lo, 6), (int)NIB(lo, 7) };
    o[2] = (v4i){ (int)NIB(hi, 0), (int)NIB(hi, 1), (int)NIB(hi, 2), (int)NIB(hi, 3) };
    o[3] = (v4i){ (int)NIB(hi, 4), (int)NIB(hi, 5), (int)NIB(hi, 6), (int)NIB(hi, 7) };
}
#undef NIB

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// (a0 <= a1) and (b0 <= b1) -> the two smallest of the four
__device__ __forceinline__ void merge2(uint32_t& a0, uint32_t& a1, uint32_t b0, uint32_t b1)
{
    const uint32_t hi = max(a0, b0);
    a0 = min(a0, b0);
    a1 = min(hi, min(a1, b1));
}

#define KNN_NONE 0x70000000u     // keys at or above: no such neighbour (valid keys stay below 0x03010000)

#define KNN_WAVES 8              // waves (= groups of 64 queries) per workgroup: they share one slice of the train set in LDS
#define KNN_CHUNK 32            // tiles of 16 train descriptors expanded into LDS at a time: 32 x 4 KB + their bit counts = 130 KB
__global__ void __launch_bounds__(KNN_WAVES * 64) k_bf_knn2(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt, int splits,
                                               int tiles_per_split, int ngroups, unsigned long long* __restrict__ part, int* __restrict__ tickets,
                                               int32_t* __restrict__ idx, int32_t* __restrict__ dist)
{
    const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
    const int gblock = blockIdx.x / splits, split = blockIdx.x - gblock * splits;
    const int group = gblock * KNN_WAVES + (threadIdx.x >> 6);
    const int q0 = group * 64, nqp = ngroups * 64;
    const int ntiles = (nt + 15) >> 4;
    const int t0 = split * tiles_per_split, t1 = min(ntiles, t0 + tiles_per_split);
    const bool active = group < ngroups;                  // (a wave past the last group of queries only helps to fill the LDS)

    // The slice of the train set this workgroup walks is EXPANDED into LDS, a chunk of KNN_CHUNK tiles at a time, by all waves
    // together: the B operand of MFMA m of tile T for lane (col, g) = the 16 bytes made of bits 64 g + 16 m .. + 15 of train
    // 16 T + col, at ((T * 4 + m) * 64 + lane) * 16 -- a wave's ds_read_b128 covers 1 KB contiguously.  The expansion (three
    // vector instructions per four bits) is thereby paid once per workgroup instead of once per wave, and the tile loop never
    // waits for global memory.
    extern __shared__ uint4 s_lds4[];
    const int chunk_tiles = min(KNN_CHUNK, max(1, tiles_per_split));
    uint4* const s_B = s_lds4;                                           // [chunk_tiles][4][64]
    int* const s_tn = (int*)(s_lds4 + (size_t)chunk_tiles * 256);        // [chunk_tiles * 16]: |t| + 256, or 0x7FFF past the end

    v4i A[4][4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int qi = max(0, min(q0 + 16 * s + col, nq - 1));
        const uint2 w = *(const uint2*)(q + (size_t)qi * 32 + 8 * g);
        expand64(w.x, w.y, A[s]);
#pragma unroll
        for (int m = 0; m < 4; m++) A[s][m] *= 0xFF;      // bytes of {0, -1}: the accumulators hold -(q.t), and the key below is ONE v_lshl_add_u32
    }
    uint32_t k0[4][4], k1[4][4];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int r = 0; r < 4; r++) k0[s][r] = k1[s][r] = 0xFFFFFFFFu;

    for (int c0 = t0; c0 < t1; c0 += KNN_CHUNK) {
        const int c1 = min(t1, c0 + KNN_CHUNK);
        if (c0 != t0) __syncthreads();                    // (every wave is done with the previous chunk)
        for (int i = threadIdx.x; i < (c1 - c0) * 16; i += KNN_WAVES * 64) {     // one train descriptor per thread
            const int j = c0 * 16 + i;
            uint4 a = make_uint4(0, 0, 0, 0), b = a;
            if (j < nt) {
                const uint4* tp = (const uint4*)(t + (size_t)j * 32);
                a = tp[0]; b = tp[1];
            }
            uint4* const dst = s_B + (size_t)(i >> 4) * 256 + (i & 15);          // + m * 64 + g * 16
            const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
            for (int gg = 0; gg < 4; gg++) {
                v4i e[4];
                expand64(w[2 * gg], w[2 * gg + 1], e);
#pragma unroll
                for (int m = 0; m < 4; m++) dst[m * 64 + gg * 16] = make_uint4((uint32_t)e[m].x, (uint32_t)e[m].y, (uint32_t)e[m].z, (uint32_t)e[m].w);
            }
            s_tn[i] = j < nt ? __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w) + 256 : 0x7FFF;
        }
        __syncthreads();
        if (!active) continue;
        const uint4* const s_w = s_B + lane;
        const int* const s_n = s_tn + col;
        v4i Bn[4];
#pragma unroll
        for (int m = 0; m < 4; m++) { const uint4 u = s_w[m * 64]; Bn[m] = (v4i){ (int)u.x, (int)u.y, (int)u.z, (int)u.w }; }
        int tn_nxt = s_n[0];
        for (int tile = c0; tile < c1; tile++) {
            v4i B[4];
#pragma unroll
            for (int m = 0; m < 4; m++) B[m] = Bn[m];
            const int tn = tn_nxt;                       // |t| + 256, or 0x7FFF past the end of the train set: never a winner
            if (tile + 1 < c1) {
#pragma unroll
                for (int m = 0; m < 4; m++) { const uint4 u = s_w[(tile + 1 - c0) * 256 + m * 64]; Bn[m] = (v4i){ (int)u.x, (int)u.y, (int)u.z, (int)u.w }; }
                tn_nxt = s_n[(tile + 1 - c0) * 16];
            }
            const uint32_t base = ((uint32_t)tn << 16) | (uint32_t)((tile * 16 + col) & 0xFFFF);
#pragma unroll
            for (int s = 0; s < 4; s++) {
                v4i acc = { 0, 0, 0, 0 };
#pragma unroll
                for (int m = 0; m < 4; m++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[s][m], B[m], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    // base - (q.t << 17): row 4 g + r of sub-tile s against train j.  Plain C on purpose: an inline-asm instruction
                    // that reads an MFMA result gets none of the wait states the compiler pads its own instructions with
                    // (measured: short sums)
                    const uint32_t key = base + ((uint32_t)acc[r] << 17);
                    k1[s][r] = med3_u32(k0[s][r], k1[s][r], key);
                    k0[s][r] = min(k0[s][r], key);
                }
            }
        }
    }
    if (!active) return;
    // the 16 lanes of a DPP row hold disjoint candidate sets of the same four queries: butterfly (row_ror 8, 4, 2, 1)
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t a0 = k0[s][r], a1 = k1[s][r];
#define ROR(v, c) (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0x120 + (c), 0xf, 0xf, false)
            merge2(a0, a1, ROR(a0, 8), ROR(a1, 8));
            merge2(a0, a1, ROR(a0, 4), ROR(a1, 4));
            merge2(a0, a1, ROR(a0, 2), ROR(a1, 2));
            merge2(a0, a1, ROR(a0, 1), ROR(a1, 1));
#undef ROR
            if (col == 4 * s + r)                       // one lane per (group of rows, state): query 16 s + 4 g + r
                __hip_atomic_store(part + (size_t)split * nqp + q0 + 16 * s + 4 * g + r, (unsigned long long)a0 | ((unsigned long long)a1 << 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // write-through (sc1): see the ticket below
        }
    // ticket: the wave that finishes a group's last slice merges the slices (its own included) and writes the result
    // Hand-off without fences (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores -> the storing wave's vmcnt(0) -> ONE
    // agent-scope add by one lane -> the wave whose add came last reads with sc1 loads after its add has returned).  An
    // agent-scope release fence instead writes back the XCD's whole L2 (1.7 - 6.5 us each, and 2000 waves would each pay it).
    int last = 1;
    if (splits > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(tickets + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = __builtin_amdgcn_readfirstlane(tk);
        last = tk == splits - 1;
    }
    if (!last) return;
    if (splits > 1 && lane == 0) __hip_atomic_store(tickets + group, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch finds it cleared
    const int qi = q0 + lane;
    if (qi >= nq) return;
    uint32_t a0 = 0xFFFFFFFFu, a1 = 0xFFFFFFFFu;
    unsigned long long p[VO_KNN_SPLITS];             // every slice's pair in flight at once (one after the other: 16 x the latency)
#pragma unroll
    for (int s = 0; s < VO_KNN_SPLITS; s++)
        p[s] = __hip_atomic_load(part + (size_t)(s < splits ? s : 0) * nqp + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (no branch: slice 0 again)
#pragma unroll
    for (int s = 0; s < VO_KNN_SPLITS; s++)
        if (s < splits) merge2(a0, a1, (uint32_t)p[s], (uint32_t)(p[s] >> 32));
    const uint4* qp = (const uint4*)(q + (size_t)qi * 32);
    const uint4 qa = qp[0], qb = qp[1];
    const int qn = __popc(qa.x) + __popc(qa.y) + __popc(qa.z) + __popc(qa.w) + __popc(qb.x) + __popc(qb.y) + __popc(qb.z) + __popc(qb.w);
    const bool h0 = a0 < KNN_NONE, h1 = a1 < KNN_NONE;
    idx[2 * qi] = h0 ? (int32_t)(a0 & 0xFFFFu) : -1;
    dist[2 * qi] = h0 ? (int32_t)(a0 >> 16) - 256 + qn : 0x7FFFFFFF;
    idx[2 * qi + 1] = h1 ? (int32_t)(a1 & 0xFFFFu) : -1;
    dist[2 * qi + 1] = h1 ? (int32_t)(a1 >> 16) - 256 + qn : 0x7FFFFFFF;
}

int match_dist_alloc(vo_ctx* ctx, int32_t** p)
{
    const size_t n = match_dist_bytes(ctx->kp_cap);
    if (hipMalloc((void**)p, n) != hipSuccess) { *p = nullptr; return VO_E_HIP; }
    // (the tickets must read zero before the first launch on WHATEVER stream uses this scratch: the alternates' streams do not
    // synchronise with the null stream the memset runs on, and a memset landing in the middle of a launch clears tickets that
    // have been drawn -- the group's result is then never written)
    if (hipMemsetAsync(*p, 0, n, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return VO_E_HIP;
    return VO_OK;
}

// d_dist must be an allocation of match_dist_bytes(kp_cap): the distances, then the slices' partial pairs, then the tickets
// (zeroed once at allocation; every launch leaves them zero)
int match_knn2(vo_ctx* ctx, const uint8_t* dq, int nq, const uint8_t* dt, int nt, int32_t* d_idx, int32_t* d_dist)
{
    if (nt > 65535) return vo_fail(ctx, VO_E_CAP, "train set of %d descriptors exceeds 65535", nt);
    if (nq <= 0) return VO_OK;
    if (nq > ctx->kp_cap) return vo_fail(ctx, VO_E_CAP, "query set of %d descriptors exceeds capacity %d", nq, ctx->kp_cap);
    const int groups = div_up(nq, 64), gblocks = div_up(groups, KNN_WAVES), ntiles = div_up(nt, 16);
    // about 2000 waves (two per SIMD), a slice of at least 2 tiles, at most VO_KNN_SPLITS slices
    int splits = std::min(std::min(VO_KNN_SPLITS, std::max(1, ntiles / 2)), div_up(2048, gblocks * KNN_WAVES));
    const int per = std::max(1, div_up(ntiles, splits));
    splits = std::max(1, div_up(ntiles, per));
    const size_t capq = ((size_t)ctx->kp_cap + 63) & ~(size_t)63;
    unsigned long long* part = (unsigned long long*)(d_dist + 2 * capq);
    int* tickets = (int*)(part + (size_t)VO_KNN_SPLITS * capq);
    const size_t lds = (size_t)std::min(per, KNN_CHUNK) * 16 * (256 + 4);
    if (lds > 64 * 1024) {                                 // (allow more than 64 KB of dynamic LDS)
        static unsigned long long attr_set = 0;
        if (!((attr_set >> (ctx->device & 63)) & 1ull)) {
            VO_HIP(ctx, hipFuncSetAttribute((const void*)k_bf_knn2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set |= 1ull << (ctx->device & 63);
        }
    }
    StageTimer tk(ctx, VO_T_KNN);
    hipLaunchKernelGGL(k_bf_knn2, dim3(gblocks * splits), dim3(KNN_WAVES * 64), lds, ctx->stream, dq, nq, dt, nt, splits, per, groups, part,
                       tickets, d_idx, d_dist);
    VO_CHECK_LAUNCH(ctx);
    return VO_OK;
}

extern "C" int vo_measure_knn(vo_ctx* ctx, int slot_a, int slot_b, int reps, double* us_per_launch)
{
    if (!ctx || !us_per_launch || reps <= 0 || reps > 10000 || slot_a < 0 || slot_a >= VO_NUM_SLOTS || slot_b < 0 || slot_b >= VO_NUM_SLOTS)
        return vo_fail(ctx, VO_E_ARG, "vo_measure_knn: bad argument");
    FrameSlot& a = ctx->slots[slot_a];
    FrameSlot& b = ctx->slots[slot_b];
    if (!a.has_kp || !b.has_kp || a.n_kp <= 0) return vo_fail(ctx, VO_E_STATE, "vo_measure_knn: both slots need keypoints");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    { int rcw = slot_wait(ctx, a); if (!rcw) rcw = slot_wait(ctx, b); if (rcw) return rcw; }
    hipEvent_t e0, e1;
    VO_HIP(ctx, hipEventCreate(&e0));
    VO_HIP(ctx, hipEventCreate(&e1));
    int rc = match_knn2(ctx, a.desc, a.n_kp, b.desc, b.n_kp, ctx->mw->m_idx, ctx->mw->m_dist);      // warm-up (LDS attribute, caches)
    if (!rc && hipEventRecord(e0, ctx->stream) != hipSuccess) rc = VO_E_HIP;
    for (int r = 0; r < reps && !rc; r++) rc = match_knn2(ctx, a.desc, a.n_kp, b.desc, b.n_kp, ctx->mw->m_idx, ctx->mw->m_dist);
    if (!rc && hipEventRecord(e1, ctx->stream) != hipSuccess) rc = VO_E_HIP;
    float ms = 0.f;
    if (!rc && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)) rc = VO_E_HIP;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc == VO_E_HIP ? vo_fail(ctx, VO_E_HIP, "vo_measure_knn: HIP error") : rc;
    *us_per_launch = 1e3 * (double)ms / reps;
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
