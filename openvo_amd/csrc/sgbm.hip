// StereoSGBM on gfx950: replaces self.stereoSGBM.compute(L, R)
// [reference stereo_camera.py:23-27,51].  Arithmetic definition: OpenCV 4.x
// calib3d/src/stereosgbm.cpp (MODE_SGBM 5 paths / MODE_HH 8 paths), all-integer, so the
// output is bit-exact against the oracle.
//
// Data layout in HBM
//   planesL[y][x]        2 x u32  : (u,u0,u1) bytes of the x-Sobel channel, then of the raw channel
//   planesR[k][y][p]     u32      : plane k of (v,v0,v1)x2, packed as value(p) | value(p-1) << 16
//   C[y][x1][d], S[...]  int16    : cost volume / path sum, d fastest (x1 = x - minX1)
// Kernels
//   k_sgbm_planes  prefilter + Birchfield-Tomasi half-pixel bounds (elementwise)
//   k_sgbm_cost    BT pixel cost + (2*SW2+1)^2 box sum, one wave = D/2 disparity PAIRS in
//                  packed int16x2 lanes, sliding sums in registers; writes C once
//   k_sgbm_path    one aggregation direction: one wave per scan line, lanes over d (packed
//                  pairs), DPP neighbour exchange + DPP wave-min; reads C, accumulates S
//   k_sgbm_wta     last path (x+1,y) fused with WTA / uniqueness / sub-pixel / disp2 / LR check
//   k_median3, k_ccl_* : medianBlur(3) and filterSpeckles (union-find labelling)
#include "vo_internal.h"

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define MAXC 0x7FFF
#define MAXC2 0x7FFF7FFFu

__device__ __forceinline__ s16x2 as_s(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_min(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_max(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u(as_s(a) + as_s(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u(as_s(a) - as_s(b)); }
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_add_sat(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_rep(int v) { return (uint32_t)(v & 0xFFFF) * 0x00010001u; }

// lane i <- lane i-1 (lane 0 keeps `fill`)
__device__ __forceinline__ uint32_t lane_from_prev(uint32_t v, uint32_t fill)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);
}
// lane i <- lane i+1 (lane 63 keeps `fill`)
__device__ __forceinline__ uint32_t lane_from_next(uint32_t v, uint32_t fill)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false);
}

// full-wave unsigned min, result uniform in every lane
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#define DPP_MIN(ctrl, rmask) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false))
    DPP_MIN(0x111, 0xf);  // row_shr:1
    DPP_MIN(0x112, 0xf);  // row_shr:2
    DPP_MIN(0x114, 0xf);  // row_shr:4
    DPP_MIN(0x118, 0xf);  // row_shr:8
    DPP_MIN(0x142, 0xa);  // row_bcast:15
    DPP_MIN(0x143, 0xc);  // row_bcast:31
#undef DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct SgbmGeom {
    int W, H, W1, D, minD, minX1, P1, P2, ur, d12, ftzero, invalid16, SW2;
};

// ---------------------------------------------------------------------------------------
// planes: per pixel the two pseudo-channels of calcPixelCostBT and their half-pixel bounds
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int chan_val(const uint8_t* img, int W, int H, int x, int y, int c, int ft)
{
    if (x <= 0 || x >= W - 1) return ft;  // both channels take tab[0] at the two border columns
    const uint8_t* r0 = img + (size_t)y * W;
    if (c == 1) return r0[x];
    const uint8_t* rn = y > 0 ? r0 - W : r0;
    const uint8_t* rs = y < H - 1 ? r0 + W : r0;
    int g = (r0[x + 1] - r0[x - 1]) * 2 + rn[x + 1] - rn[x - 1] + rs[x + 1] - rs[x - 1];
    return min(max(g, -ft), ft) + ft;
}

__device__ __forceinline__ void chan_bounds(const uint8_t* img, int W, int H, int x, int y, int c, int ft,
                                            int& v, int& lo, int& hi)
{
    v = chan_val(img, W, H, x, y, c, ft);
    int vl = x > 0 ? (v + chan_val(img, W, H, x - 1, y, c, ft)) / 2 : v;
    int vr = x < W - 1 ? (v + chan_val(img, W, H, x + 1, y, c, ft)) / 2 : v;
    lo = min(min(vl, vr), v);
    hi = max(max(vl, vr), v);
}

__global__ void k_sgbm_planes(const uint8_t* __restrict__ L, const uint8_t* __restrict__ R, int W, int H,
                              int ft, uint32_t* __restrict__ PL, uint32_t* __restrict__ PR)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    size_t i = (size_t)y * W + x, plane = (size_t)W * H;
    for (int c = 0; c < 2; c++) {
        int v, lo, hi;
        chan_bounds(L, W, H, x, y, c, ft, v, lo, hi);
        PL[i * 2 + c] = (uint32_t)v | ((uint32_t)lo << 8) | ((uint32_t)hi << 16);
        int v1 = 0, lo1 = 0, hi1 = 0;
        chan_bounds(R, W, H, x, y, c, ft, v, lo, hi);
        if (x > 0) chan_bounds(R, W, H, x - 1, y, c, ft, v1, lo1, hi1);
        PR[(size_t)(c * 3 + 0) * plane + i] = (uint32_t)v | ((uint32_t)v1 << 16);
        PR[(size_t)(c * 3 + 1) * plane + i] = (uint32_t)lo | ((uint32_t)lo1 << 16);
        PR[(size_t)(c * 3 + 2) * plane + i] = (uint32_t)hi | ((uint32_t)hi1 << 16);
    }
}

// ---------------------------------------------------------------------------------------
// cost volume
// ---------------------------------------------------------------------------------------
// BT cost of one (row, column) for this lane's disparity pair (d, d+1), both channels
__device__ __forceinline__ uint32_t bt_pair(const uint32_t* __restrict__ PL, const uint32_t* __restrict__ PR,
                                            size_t plane, int W, int r, int ximg, int p)
{
    const size_t li = ((size_t)r * W + ximg) * 2;
    const size_t ri = (size_t)r * W + p;
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        uint32_t lw = PL[li + c];
        uint32_t U = pk_rep(lw & 255), U0 = pk_rep((lw >> 8) & 255), U1 = pk_rep((lw >> 16) & 255);
        uint32_t V = PR[(size_t)(c * 3 + 0) * plane + ri];
        uint32_t V0 = PR[(size_t)(c * 3 + 1) * plane + ri];
        uint32_t V1 = PR[(size_t)(c * 3 + 2) * plane + ri];
        uint32_t c0 = pk_max(pk_max(pk_sub(U, V1), pk_sub(V0, U)), 0u);
        uint32_t c1 = pk_max(pk_max(pk_sub(V, U1), pk_sub(U0, V)), 0u);
        uint32_t m = pk_min(c0, c1);
        if (c == 1) m = (m >> 2) & 0x3FFF3FFFu;
        acc = pk_add(acc, m);
    }
    return acc;
}

// block = one wave per 64 disparity pairs; tile TX columns x TY rows.  Horizontal window via a
// register sliding sum over the TX+2*SW2 evaluated columns, vertical window via a register ring.
// Specialised for SW2 == 2 (blockSize 5, the reference configuration) with a generic fallback.
template <int TX, int SW2>
__global__ void __launch_bounds__(256) k_sgbm_cost(const uint32_t* __restrict__ PL, const uint32_t* __restrict__ PR,
                                                  SgbmGeom g, int TY, int16_t* __restrict__ C)
{
    constexpr int WIN = 2 * SW2 + 1;
    const int dp = threadIdx.x;  // disparity pair index
    if (2 * dp >= g.D) return;
    const int xa = blockIdx.x * TX, ya = blockIdx.y * TY;
    const size_t plane = (size_t)g.W * g.H;
    const int pbase = g.minX1 - g.minD - 2 * dp;  // right position = x1 + pbase

    uint32_t ring[TX][WIN];
    uint32_t acc[TX];
#pragma unroll
    for (int j = 0; j < TX; j++) acc[j] = pk_rep(g.P2);

    // horizontal box sums of one image row r into hs[TX]
    auto row_hsum = [&](int r, uint32_t* hs) {
        uint32_t pc[TX + 2 * SW2];
#pragma unroll
        for (int k = 0; k < TX + 2 * SW2; k++) {
            int xe = min(max(xa - SW2 + k, 0), g.W1 - 1);
            pc[k] = bt_pair(PL, PR, plane, g.W, r, xe + g.minX1, xe + pbase);
        }
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < WIN; k++) s = pk_add(s, pc[k]);
        hs[0] = s;
#pragma unroll
        for (int j = 1; j < TX; j++) {
            s = pk_sub(pk_add(s, pc[j + WIN - 1]), pc[j - 1]);
            hs[j] = s;
        }
    };

    // prime the ring with rows clamp(ya-SW2 .. ya+SW2)
#pragma unroll
    for (int k = 0; k < WIN; k++) {
        uint32_t hs[TX];
        row_hsum(min(max(ya - SW2 + k, 0), g.H - 1), hs);
#pragma unroll
        for (int j = 0; j < TX; j++) { ring[j][k] = hs[j]; acc[j] = pk_add(acc[j], hs[j]); }
    }
    const int yend = min(ya + TY, g.H);
    for (int y0 = ya; y0 < yend; y0 += WIN) {
#pragma unroll
        for (int k = 0; k < WIN; k++) {
            const int y = y0 + k;
            if (y < yend) {
                // store row y
#pragma unroll
                for (int j = 0; j < TX; j++) {
                    int x1 = xa + j;
                    if (x1 < g.W1)
                        *(uint32_t*)(C + ((size_t)y * g.W1 + x1) * g.D + 2 * dp) = acc[j];
                }
                // slide to row y+1: add hsum(clamp(y+1+SW2)), drop hsum(clamp(y-SW2)) == ring slot k
                if (y + 1 < yend) {
                    uint32_t hs[TX];
                    row_hsum(min(y + 1 + SW2, g.H - 1), hs);
#pragma unroll
                    for (int j = 0; j < TX; j++) {
                        acc[j] = pk_sub(pk_add(acc[j], hs[j]), ring[j][k]);
                        ring[j][k] = hs[j];
                    }
                }
            }
        }
    }
}

// generic (any block size) fallback: direct box sum, one thread per (x1, d pair), slow but exact
__global__ void k_sgbm_cost_generic(const uint32_t* __restrict__ PL, const uint32_t* __restrict__ PR,
                                    SgbmGeom g, int16_t* __restrict__ C)
{
    const int dp = threadIdx.x;
    if (2 * dp >= g.D) return;
    const int x1 = blockIdx.x, y = blockIdx.y;
    const size_t plane = (size_t)g.W * g.H;
    const int pbase = g.minX1 - g.minD - 2 * dp;
    uint32_t acc = pk_rep(g.P2);
    for (int j = -g.SW2; j <= g.SW2; j++) {
        int r = min(max(y + j, 0), g.H - 1);
        for (int i = -g.SW2; i <= g.SW2; i++) {
            int xe = min(max(x1 + i, 0), g.W1 - 1);
            acc = pk_add(acc, bt_pair(PL, PR, plane, g.W, r, xe + g.minX1, xe + pbase));
        }
    }
    *(uint32_t*)(C + ((size_t)y * g.W1 + x1) * g.D + 2 * dp) = acc;
}

// ---------------------------------------------------------------------------------------
// path aggregation
// ---------------------------------------------------------------------------------------
// One lane holds NP packed registers = 2*NP consecutive disparities.  `active` lanes cover
// [0, D); the others carry MAX_COST so they act as the d = D sentinel.
template <int NP>
struct LaneVec { uint32_t r[NP]; };

template <int NP>
__device__ __forceinline__ uint32_t lane_min(const LaneVec<NP>& v)
{
    uint32_t m = v.r[0];
#pragma unroll
    for (int k = 1; k < NP; k++) m = pk_min(m, v.r[k]);
    return min(m & 0xFFFFu, m >> 16);
}

// L(d) = C(d) + min(Lp(d), Lp(d-1)+P1, Lp(d+1)+P1, delta) - delta
template <int NP>
__device__ __forceinline__ LaneVec<NP> path_step(const LaneVec<NP>& Cp, const LaneVec<NP>& Lp, uint32_t delta2,
                                                 uint32_t P1_2, bool active)
{
    LaneVec<NP> out;
    uint32_t prev_hi = lane_from_prev(Lp.r[NP - 1], MAXC2);  // lane-1's last register
    uint32_t next_lo = lane_from_next(Lp.r[0], MAXC2);       // lane+1's first register
#pragma unroll
    for (int k = 0; k < NP; k++) {
        uint32_t left = k == 0 ? prev_hi : Lp.r[k - 1];
        uint32_t right = k == NP - 1 ? next_lo : Lp.r[k + 1];
        uint32_t dm1 = __builtin_amdgcn_alignbit(Lp.r[k], left, 16);   // (L[d-1], L[d])
        uint32_t dp1 = __builtin_amdgcn_alignbit(right, Lp.r[k], 16);  // (L[d+1], L[d+2])
        uint32_t m = pk_min(pk_min(Lp.r[k], delta2), pk_min(pk_add_sat(dm1, P1_2), pk_add_sat(dp1, P1_2)));
        uint32_t L = pk_sub(pk_add(Cp.r[k], m), delta2);
        out.r[k] = active ? L : MAXC2;
    }
    return out;
}

template <int NP>
__device__ __forceinline__ LaneVec<NP> load_vec(const int16_t* p)
{
    LaneVec<NP> v;
    if (NP == 1) v.r[0] = *(const uint32_t*)p;
    else {
        uint2 t = *(const uint2*)p;
        v.r[0] = t.x;
        if (NP > 1) v.r[NP - 1] = t.y;
    }
    return v;
}
template <int NP>
__device__ __forceinline__ void store_vec(int16_t* p, const LaneVec<NP>& v)
{
    if (NP == 1) *(uint32_t*)p = v.r[0];
    else *(uint2*)p = make_uint2(v.r[0], v.r[NP - 1]);
}

// One wave per scan line.  dir: step (sx, sy).  Lines are enumerated so that neighbouring
// waves touch neighbouring memory.  FIRST: S = L (no read), else S = sat(S + L).
template <int NP, bool FIRST>
__global__ void __launch_bounds__(256) k_sgbm_path(const int16_t* __restrict__ C, int16_t* __restrict__ S, SgbmGeom g,
                                                  int sx, int sy, int nlines)
{
    const int lane = threadIdx.x & 63;
    const int line = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (line >= nlines) return;
    const int W1 = g.W1, H = g.H;
    // start cell and length of this line
    int x0, y0;
    if (sy == 0) { y0 = line; x0 = sx > 0 ? 0 : W1 - 1; }
    else if (sx == 0) { x0 = line; y0 = sy > 0 ? 0 : H - 1; }
    else {
        // diagonals: first W1 lines start on the first row in sweep order, the rest on the side
        int ytop = sy > 0 ? 0 : H - 1;
        if (line < W1) { x0 = line; y0 = ytop; }
        else { x0 = sx > 0 ? 0 : W1 - 1; y0 = ytop + sy * (line - W1 + 1); }
    }
    int nx = sx > 0 ? W1 - x0 : (sx < 0 ? x0 + 1 : 1 << 30);
    int ny = sy > 0 ? H - y0 : (sy < 0 ? y0 + 1 : 1 << 30);
    const int n = min(nx, ny);

    const int nact = g.D / (2 * NP);
    const bool active = lane < nact;
    const int lofs = active ? lane * 2 * NP : 0;
    const uint32_t P1_2 = pk_rep(g.P1);
    const ptrdiff_t stride = ((ptrdiff_t)sy * W1 + sx) * g.D;
    const int16_t* cp = C + ((size_t)y0 * W1 + x0) * g.D + lofs;
    int16_t* sp = S + ((size_t)y0 * W1 + x0) * g.D + lofs;

    LaneVec<NP> Lp;
#pragma unroll
    for (int k = 0; k < NP; k++) Lp.r[k] = active ? 0u : MAXC2;  // predecessor outside: zeros, min 0
    uint32_t delta2 = pk_rep(g.P2);

    constexpr int PF = 8;  // software prefetch depth (steps)
    LaneVec<NP> cbuf[PF], sbuf[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        if (k < n) {
            cbuf[k] = load_vec<NP>(cp + (ptrdiff_t)k * stride);
            if (!FIRST) sbuf[k] = load_vec<NP>(sp + (ptrdiff_t)k * stride);
        }
    }
    for (int i0 = 0; i0 < n; i0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int i = i0 + k;
            if (i < n) {
                LaneVec<NP> Cv = cbuf[k], Sv;
                if (!FIRST) Sv = sbuf[k];
                if (i + PF < n) {
                    cbuf[k] = load_vec<NP>(cp + (ptrdiff_t)(i + PF) * stride);
                    if (!FIRST) sbuf[k] = load_vec<NP>(sp + (ptrdiff_t)(i + PF) * stride);
                }
                LaneVec<NP> L = path_step<NP>(Cv, Lp, delta2, P1_2, active);
                uint32_t mn = wave_min_u32(lane_min<NP>(L));
                delta2 = pk_rep((int)mn + g.P2);
                Lp = L;
                LaneVec<NP> So;
#pragma unroll
                for (int q = 0; q < NP; q++) So.r[q] = FIRST ? L.r[q] : pk_add_sat(Sv.r[q], L.r[q]);
                if (active) store_vec<NP>(sp + (ptrdiff_t)i * stride, So);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// WTA row kernel: (MODE_SGBM) fifth path (x+1,y) swept right-to-left fused with winner-take-all,
// uniqueness, sub-pixel, disp2 and the LR check.  One wave per image row.
// ---------------------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ int pick_elem(const LaneVec<NP>& v, int d)
{
    // value of disparity d (wave-uniform) out of the distributed vector
    const int per = 2 * NP;
    const int tl = d / per, te = d - tl * per;
    uint32_t reg = v.r[0];
#pragma unroll
    for (int k = 1; k < NP; k++) if ((te >> 1) == k) reg = v.r[k];
    uint32_t got = (uint32_t)__builtin_amdgcn_readlane((int)reg, tl);
    return (te & 1) ? (int)(got >> 16) : (int)(got & 0xFFFFu);
}

template <int NP, bool LAST_PATH>
__global__ void __launch_bounds__(64) k_sgbm_wta(const int16_t* __restrict__ C, const int16_t* __restrict__ S, SgbmGeom g,
                                                 int16_t* __restrict__ disp)
{
    extern __shared__ int16_t smem[];
    int16_t* d1row = smem;             // W
    int16_t* d2row = smem + g.W;       // W
    int16_t* d2cost = smem + 2 * g.W;  // W
    const int lane = threadIdx.x, y = blockIdx.x;
    const int W1 = g.W1, D = g.D;
    for (int x = lane; x < g.W; x += 64) { d1row[x] = (int16_t)g.invalid16; d2row[x] = (int16_t)g.invalid16; d2cost[x] = MAXC; }
    __syncthreads();

    const int nact = D / (2 * NP);
    const bool active = lane < nact;
    const int lofs = active ? lane * 2 * NP : 0;
    const uint32_t P1_2 = pk_rep(g.P1);
    const int16_t* crow = C + (size_t)y * W1 * D + lofs;
    const int16_t* srow = S + (size_t)y * W1 * D + lofs;
    LaneVec<NP> Lp;
#pragma unroll
    for (int k = 0; k < NP; k++) Lp.r[k] = active ? 0u : MAXC2;
    uint32_t delta2 = pk_rep(g.P2);
    const int ur100 = 100 - g.ur;

    constexpr int PF = 4;
    LaneVec<NP> cbuf[PF], sbuf[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        int x = W1 - 1 - k;
        if (x >= 0) {
            if (LAST_PATH) cbuf[k] = load_vec<NP>(crow + (size_t)x * D);
            sbuf[k] = load_vec<NP>(srow + (size_t)x * D);
        }
    }
    for (int xb = W1 - 1; xb >= 0; xb -= PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int x = xb - k;
            if (x >= 0) {
                LaneVec<NP> Cv, Sv = sbuf[k];
                if (LAST_PATH) Cv = cbuf[k];
                if (x - PF >= 0) {
                    if (LAST_PATH) cbuf[k] = load_vec<NP>(crow + (size_t)(x - PF) * D);
                    sbuf[k] = load_vec<NP>(srow + (size_t)(x - PF) * D);
                }
                LaneVec<NP> Sf;
                if (LAST_PATH) {
                    LaneVec<NP> L = path_step<NP>(Cv, Lp, delta2, P1_2, active);
                    uint32_t mn = wave_min_u32(lane_min<NP>(L));
                    delta2 = pk_rep((int)mn + g.P2);
                    Lp = L;
#pragma unroll
                    for (int q = 0; q < NP; q++) Sf.r[q] = active ? pk_add_sat(Sv.r[q], L.r[q]) : MAXC2;
                } else {
#pragma unroll
                    for (int q = 0; q < NP; q++) Sf.r[q] = active ? Sv.r[q] : MAXC2;
                }
                // first minimum over d: key = S << 8 | d
                uint32_t key = 0xFFFFFFFFu;
#pragma unroll
                for (int q = 0; q < NP; q++) {
                    int d = lofs + 2 * q;
                    uint32_t k0 = ((Sf.r[q] & 0xFFFFu) << 8) | (uint32_t)d;
                    uint32_t k1 = ((Sf.r[q] >> 16) << 8) | (uint32_t)(d + 1);
                    key = min(key, min(k0, k1));
                }
                if (!active) key = 0xFFFFFFFFu;
                key = wave_min_u32(key);
                const int minS = (int)(key >> 8), best = (int)(key & 255u);
                // uniqueness
                bool viol = false;
#pragma unroll
                for (int q = 0; q < NP; q++) {
                    int d = lofs + 2 * q;
                    int s0 = (int)(Sf.r[q] & 0xFFFFu), s1 = (int)(Sf.r[q] >> 16);
                    viol |= (s0 * ur100 < minS * 100) && (abs(best - d) > 1);
                    viol |= (s1 * ur100 < minS * 100) && (abs(best - d - 1) > 1);
                }
                viol = viol && active;
                if (__ballot(viol) == 0ull) {
                    int dd = best * 16;
                    if (best > 0 && best < D - 1) {
                        int sm = pick_elem<NP>(Sf, best - 1), spl = pick_elem<NP>(Sf, best + 1);
                        int denom2 = max(sm + spl - 2 * minS, 1);
                        dd = best * 16 + ((sm - spl) * 16 + denom2) / (denom2 * 2);
                    }
                    if (lane == 0) {
                        int x2 = x + g.minX1 - best - g.minD;
                        if (d2cost[x2] > minS) { d2cost[x2] = (int16_t)minS; d2row[x2] = (int16_t)(best + g.minD); }
                        d1row[x + g.minX1] = (int16_t)(dd + g.minD * 16);
                    }
                }
            }
        }
    }
    __syncthreads();
    // LR consistency check, then write the row
    for (int x = lane; x < g.W; x += 64) {
        int d1 = d1row[x];
        if (x >= g.minX1 && x < g.minX1 + W1 && d1 != g.invalid16) {
            int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
            int _x = x - _d, x_ = x - d_;
            if (0 <= _x && _x < g.W && d2row[_x] >= g.minD && abs(d2row[_x] - _d) > g.d12 &&
                0 <= x_ && x_ < g.W && d2row[x_] >= g.minD && abs(d2row[x_] - d_) > g.d12)
                d1 = g.invalid16;
        }
        disp[(size_t)y * g.W + x] = (int16_t)d1;
    }
}

// ---------------------------------------------------------------------------------------
// post filters
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void cswap(int& a, int& b) { int t = min(a, b); b = max(a, b); a = t; }

__global__ void k_median3(const int16_t* __restrict__ src, int W, int H, int16_t* __restrict__ dst)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int16_t* r0 = src + (size_t)max(y - 1, 0) * W;
    const int16_t* r1 = src + (size_t)y * W;
    const int16_t* r2 = src + (size_t)min(y + 1, H - 1) * W;
    int xl = max(x - 1, 0), xr = min(x + 1, W - 1);
    int p0 = r0[xl], p1 = r0[x], p2 = r0[xr], p3 = r1[xl], p4 = r1[x], p5 = r1[xr], p6 = r2[xl], p7 = r2[x], p8 = r2[xr];
    cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p1);
    cswap(p3, p4); cswap(p6, p7); cswap(p1, p2); cswap(p4, p5);
    cswap(p7, p8); cswap(p0, p3); cswap(p5, p8); cswap(p4, p7);
    cswap(p3, p6); cswap(p1, p4); cswap(p2, p5); cswap(p4, p7);
    cswap(p4, p2); cswap(p6, p4); cswap(p4, p2);
    dst[(size_t)y * W + x] = (int16_t)p4;
}

// filterSpeckles as connected-component labelling: union-find over the 4-neighbour graph whose
// edges join pixels that are both != newVal and differ by <= maxDiff.
__device__ __forceinline__ int uf_find(int* L, int i)
{
    int p = L[i];
    while (p != i) { i = p; p = L[i]; }
    return i;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b)
{
    for (int it = 0; it < (1 << 24); it++) {
        a = uf_find(L, a);
        b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}
__global__ void k_ccl_init(const int16_t* __restrict__ img, int n, int newVal, int* __restrict__ L, int* __restrict__ size)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    L[i] = img[i] != newVal ? i : -1;
    size[i] = 0;
}
__global__ void k_ccl_merge(const int16_t* __restrict__ img, int W, int H, int newVal, int maxDiff, int* __restrict__ L)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    int i = y * W + x;
    int v = img[i];
    if (v == newVal) return;
    if (x + 1 < W) { int u = img[i + 1]; if (u != newVal && abs(v - u) <= maxDiff) uf_union(L, i, i + 1); }
    if (y + 1 < H) { int u = img[i + W]; if (u != newVal && abs(v - u) <= maxDiff) uf_union(L, i, i + W); }
}
__global__ void k_ccl_count(int n, int* __restrict__ L, int* __restrict__ size)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (L[i] < 0) return;
    int r = uf_find(L, i);
    L[i] = r;
    atomicAdd(&size[r], 1);
}
__global__ void k_ccl_apply(int16_t* __restrict__ img, int n, int newVal, int maxSize, const int* __restrict__ L,
                            const int* __restrict__ size)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = L[i];
    if (r >= 0 && size[uf_find((int*)L, r)] <= maxSize) img[i] = (int16_t)newVal;
}

// ---------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------
template <int NP>
static int launch_paths(vo_ctx* ctx, const SgbmGeom& g, int mode, int16_t* d_disp)
{
    struct Dir { int sx, sy; };
    // MODE_SGBM: (x-1,y) (x-1,y-1) (x,y-1) (x+1,y-1) predecessors, then (x+1,y) fused into WTA.
    // Steps are the negated predecessor offsets.
    static const Dir d5[] = { {0, 1}, {1, 1}, {-1, 1}, {1, 0} };
    static const Dir d8[] = { {0, 1}, {1, 1}, {-1, 1}, {1, 0}, {-1, 0}, {1, -1}, {0, -1}, {-1, -1} };
    const Dir* dirs = mode == 1 ? d8 : d5;
    const int nd = mode == 1 ? 8 : 4;
    {
        StageTimer t(ctx, VO_T_SGBM_AGG);
        for (int k = 0; k < nd; k++) {
            int nlines = dirs[k].sy == 0 ? g.H : (dirs[k].sx == 0 ? g.W1 : g.W1 + g.H - 1);
            dim3 grid(div_up(nlines, 4)), block(256);
            if (k == 0)
                hipLaunchKernelGGL((k_sgbm_path<NP, true>), grid, block, 0, ctx->stream, ctx->C, ctx->S, g, dirs[k].sx, dirs[k].sy, nlines);
            else
                hipLaunchKernelGGL((k_sgbm_path<NP, false>), grid, block, 0, ctx->stream, ctx->C, ctx->S, g, dirs[k].sx, dirs[k].sy, nlines);
        }
        VO_CHECK_LAUNCH(ctx);
    }
    ctx->last_paths = mode == 1 ? 8 : 5;
    {
        StageTimer t(ctx, VO_T_SGBM_WTA);
        size_t sh = (size_t)3 * g.W * sizeof(int16_t);
        if (mode == 1)
            hipLaunchKernelGGL((k_sgbm_wta<NP, false>), dim3(g.H), dim3(64), sh, ctx->stream, ctx->C, ctx->S, g, d_disp);
        else
            hipLaunchKernelGGL((k_sgbm_wta<NP, true>), dim3(g.H), dim3(64), sh, ctx->stream, ctx->C, ctx->S, g, d_disp);
        VO_CHECK_LAUNCH(ctx);
    }
    return VO_OK;
}

int sgbm_run(vo_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int w, int h, int16_t* d_disp)
{
    const SgbmEff& e = ctx->sg;
    if (!e.set) return vo_fail(ctx, VO_E_STATE, "vo_set_sgbm has not been called");
    SgbmGeom g;
    g.W = w; g.H = h; g.D = e.D; g.minD = e.minD; g.P1 = e.P1; g.P2 = e.P2; g.ur = e.ur; g.d12 = e.d12;
    g.ftzero = e.ftzero; g.SW2 = e.SW2;
    g.minX1 = e.maxD > 0 ? e.maxD : 0;
    int maxX1 = w + (e.minD < 0 ? e.minD : 0);
    g.W1 = maxX1 - g.minX1;
    g.invalid16 = (e.minD - 1) * 16;
    const int n = w * h;
    if (g.W1 <= 0) {
        // every pixel invalid
        std::vector<int16_t> inv((size_t)n, (int16_t)g.invalid16);
        VO_HIP(ctx, hipMemcpyAsync(d_disp, inv.data(), (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream));
        VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->last_cells = 0;
        return VO_OK;
    }
    const size_t cells = (size_t)g.W1 * h * g.D;
    if (cells > ctx->vol_cells || e.D > 256)
        return vo_fail(ctx, VO_E_CAP, "cost volume %zu cells exceeds the capacity given to vo_create (or D > 256)", cells);
    ctx->last_cells = (int64_t)cells;
    {
        StageTimer t(ctx, VO_T_SGBM_COST);
        hipLaunchKernelGGL(k_sgbm_planes, dim3(div_up(w, 256), h), dim3(256), 0, ctx->stream, dL, dR, w, h, g.ftzero,
                           ctx->planesL, ctx->planesR);
        const int bx = ((g.D / 2 + 63) / 64) * 64;
        if (g.SW2 == 2) {
            constexpr int TX = 8;
            const int TY = 60;
            hipLaunchKernelGGL((k_sgbm_cost<TX, 2>), dim3(div_up(g.W1, TX), div_up(h, TY)), dim3(bx), 0, ctx->stream,
                               ctx->planesL, ctx->planesR, g, TY, ctx->C);
        } else {
            hipLaunchKernelGGL(k_sgbm_cost_generic, dim3(g.W1, h), dim3(bx), 0, ctx->stream, ctx->planesL, ctx->planesR, g, ctx->C);
        }
        VO_CHECK_LAUNCH(ctx);
    }
    int rc = g.D > 128 ? launch_paths<2>(ctx, g, e.mode, ctx->disp_raw) : launch_paths<1>(ctx, g, e.mode, ctx->disp_raw);
    if (rc) return rc;
    {
        StageTimer t(ctx, VO_T_SGBM_POST);
        hipLaunchKernelGGL(k_median3, dim3(div_up(w, 256), h), dim3(256), 0, ctx->stream, ctx->disp_raw, w, h, d_disp);
        if (e.speckleWindow > 0) {
            const int newVal = g.invalid16, maxDiff = 16 * e.speckleRange;
            hipLaunchKernelGGL(k_ccl_init, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_disp, n, newVal, ctx->ccl_label, ctx->ccl_size);
            hipLaunchKernelGGL(k_ccl_merge, dim3(div_up(w, 256), h), dim3(256), 0, ctx->stream, d_disp, w, h, newVal, maxDiff, ctx->ccl_label);
            hipLaunchKernelGGL(k_ccl_count, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, n, ctx->ccl_label, ctx->ccl_size);
            hipLaunchKernelGGL(k_ccl_apply, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_disp, n, newVal, e.speckleWindow, ctx->ccl_label, ctx->ccl_size);
        }
        VO_CHECK_LAUNCH(ctx);
    }
    return VO_OK;
}
