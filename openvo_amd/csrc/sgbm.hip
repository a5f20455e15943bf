// StereoSGBM on gfx950: replaces self.stereoSGBM.compute(L, R)
// [reference stereo_camera.py:23-27,51].  Arithmetic definition: OpenCV 4.x
// calib3d/src/stereosgbm.cpp (MODE_SGBM 5 paths / MODE_HH 8 paths), all-integer, so the
// output is bit-exact against the CPU restatement.
//
// Data layout in HBM
//   planesL[y][x]        2 x u32  : (u,u0,u1) bytes of the x-Sobel channel, then of the raw channel
//   planesR[k][y][p]     u32      : plane k of (v,v0,v1)x2, packed as value(p) | value(p-1) << 16
//   C[y][x1][Dp]         int16    : block cost (+P2), d fastest, Dp = D rounded up to 32, x1 = x - minX1
//   L[dir][y][x1][Dp]    int16    : one aggregated volume per STORED path direction (all but the top-down
//                                   vertical one, which only ever lives in registers)
// Kernels
//   k_sgbm_planes  prefilter + Birchfield-Tomasi half-pixel bounds (elementwise)
//   k_sgbm_cost_sweep  BT pixel cost + (2*SW2+1)^2 box sum; one lane = one disparity PAIR in packed
//                  int16x2 math; right-image planes travel through DPP lane-shift chains, the horizontal
//                  window slides in registers, the vertical one through an LDS ring; writes C once
//   k_sgbm_paths   every stored path direction in one launch.  A scan line lives in one 16-lane DPP row
//                  (each lane holds Dp/16 consecutive disparities in packed registers), so a wave
//                  advances 4 independent lines; neighbours d-1/d+1 come from row_shr/row_shl
//                  (row ends are the MAX_COST sentinels for free), the min over d is a 4-step
//                  row_ror all-reduce.  Reads C once per direction, writes that direction's L with
//                  non-temporal stores (single use; C must stay cached for the other directions).
//   k_sgbm_vwta    the top-down vertical direction fused with the WTA: a wave walks 4 columns, per row it
//                  reads C and the stored L volumes, advances its own path in registers, sums, picks the
//                  first minimum and tests uniqueness; leaves a 2-word record per pixel
//   k_sgbm_fin     records -> sub-pixel disp1 + disp2 candidates via atomicMin on (cost, scan order) keys
//   k_sgbm_wta     unfused fallback (VO_FUSE_WTA=0, uniquenessRatio >= 100): per pixel (16 lanes) S = sat-sum
//                  of all L volumes, same winner logic inline
//   k_lr_median3   left-right consistency check evaluated inside medianBlur(3)
//   k_ccl_*        filterSpeckles (run-based union-find labelling)
#include "vo_internal.h"
#include <stdlib.h>
#include <type_traits>

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define MAXC 0x7FFF
#define MAXC2 0x7FFF7FFFu
#define D2_EMPTY 0x7FFFFFFF

__device__ __forceinline__ s16x2 as_s(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_min(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_max(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u(as_s(a) + as_s(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u(as_s(a) - as_s(b)); }
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_add_sat(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t pk_sub_sat_u(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_rep(int v) { return (uint32_t)(v & 0xFFFF) * 0x00010001u; }

#define DPP(old, src, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(src), ctrl, 0xf, 0xf, false))
#define ROW_SHL1 0x101
#define ROW_SHR1 0x111
#define ROW_ROR(n) (0x120 + (n))

// all-reduce (min, unsigned) inside each 16-lane row: every lane ends with the row minimum
// (old = the identity of min lets the compiler fold each rotate into one v_min_u32_dpp)
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v)
{
    v = min(v, DPP(0xFFFFFFFFu, v, ROW_ROR(8)));
    v = min(v, DPP(0xFFFFFFFFu, v, ROW_ROR(4)));
    v = min(v, DPP(0xFFFFFFFFu, v, ROW_ROR(2)));
    v = min(v, DPP(0xFFFFFFFFu, v, ROW_ROR(1)));
    return v;
}

struct SgbmGeom {
    int W, H, W1, D, Dp, minD, minX1, P1, P2, ur, d12, ftzero, invalid16, SW2;
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
                              int ft, uint32_t* __restrict__ PL, uint32_t* __restrict__ PR, int* __restrict__ d2key,
                              int* __restrict__ rs_ctl, int rs_ctl_words)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    // the raster sweeps' band tickets and progress counters of this run start at zero (word 1 of each of the
    // two control blocks is its sticky error flag and is left alone)
    if (y == 0 && blockIdx.x == 0)
        for (int i = threadIdx.x; i < rs_ctl_words; i += blockDim.x)
            if ((i % (rs_ctl_words / 2)) != 1) rs_ctl[i] = 0;
    if (x >= W) return;
    size_t i = (size_t)y * W + x, plane = (size_t)W * H;
    d2key[i] = D2_EMPTY;   // the disp2 candidates of this run start empty (saves a fill launch before the WTA)
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

// Sweep formulation of the same cost volume.  A wave (64 disparity pairs) walks a strip of XT
// columns of one image row left to right.  At column x lane l needs the right-image planes at
// position x - minD - 2l, i.e. what lane l-1 held two columns earlier: the six plane registers
// live in two lane-shift chains (even / odd columns) advanced by one DPP wave_shr, lane 0 loading
// the entering element (one broadcast load instead of a 64-lane strided gather).  The horizontal
// window slides in registers, the vertical window through a 5-row LDS ring per (column, lane) and
// a register accumulator per column, so C is the only volume written and nothing is re-read.
__device__ __forceinline__ uint32_t bt_regs(uint32_t lw0, uint32_t lw1, const uint32_t* V)
{
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const uint32_t lw = c ? lw1 : lw0;
        const uint32_t U = pk_rep(lw & 255), U0 = pk_rep((lw >> 8) & 255), U1 = pk_rep((lw >> 16) & 255);
        const uint32_t c0 = pk_max(pk_max(pk_sub(U, V[c * 3 + 2]), pk_sub(V[c * 3 + 1], U)), 0u);
        const uint32_t c1 = pk_max(pk_max(pk_sub(V[c * 3 + 0], U1), pk_sub(U0, V[c * 3 + 0])), 0u);
        uint32_t m = pk_min(c0, c1);
        if (c == 1) m = (m >> 2) & 0x3FFF3FFFu;
        acc = pk_add(acc, m);
    }
    return acc;
}

template <int XT, int SW2>
__global__ void __launch_bounds__(128) k_sgbm_cost_sweep(const uint32_t* __restrict__ PL, const uint32_t* __restrict__ PR,
                                                        SgbmGeom g, int TY, int16_t* __restrict__ C)
{
    constexpr int WIN = 2 * SW2 + 1, NC = XT + 2 * SW2;
    extern __shared__ uint32_t s_ring[];  // [waves][WIN][XT][64]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int dpl = threadIdx.x;                  // disparity pair index (padded layout)
    const bool pad = 2 * dpl >= g.D;
    const int xa = blockIdx.x * XT, ya = blockIdx.y * TY;
    const size_t plane = (size_t)g.W * g.H;
    uint32_t* ring = s_ring + (size_t)wv * WIN * XT * 64 + lane;
    const int yend = min(ya + TY, g.H);
    const int nrows = (yend - ya) + 2 * SW2;       // rows ya-2 .. yend+1 (clamped)
    const int lane0_d = 2 * 64 * wv;               // disparity of this wave's lane 0 (relative to minD)
    // lane i (< NC) fetches, once per row, what column k = i needs: the left-image words and the
    // element entering the shift chain at lane 0; the columns then pick them up with v_readlane
    const int krun = min(lane, NC - 1);
    const int xrun = min(max(xa - SW2 + krun, 0), g.W1 - 1) + g.minX1;
    const int prun = max(xrun - g.minD - lane0_d, 0);
    const int pown = max(-g.minD - 2 * (pad ? 0 : dpl), -(1 << 28));   // + ximg = this lane's own position

    uint32_t acc[XT];
#pragma unroll
    for (int j = 0; j < XT; j++) acc[j] = pk_rep(g.P2);

    for (int rr = 0; rr < nrows; rr++) {
        const int r = min(max(ya - SW2 + rr, 0), g.H - 1);
        const int slot = rr % WIN;
        const size_t rowi = (size_t)r * g.W;
        const uint32_t plr0 = PL[(rowi + xrun) * 2], plr1 = PL[(rowi + xrun) * 2 + 1];
        uint32_t run[6];
#pragma unroll
        for (int q = 0; q < 6; q++) run[q] = PR[(size_t)q * plane + rowi + prun];
        uint32_t A[6], B[6];                      // shift chains for even / odd k
        bool a_init = false, b_init = false;
        uint32_t pc[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const int x = xa - SW2 + k;           // block-uniform
            const int xe = min(max(x, 0), g.W1 - 1);
            const int ximg = xe + g.minX1;
            uint32_t* S = (k & 1) ? B : A;
            bool& inited = (k & 1) ? b_init : a_init;
            const uint32_t lw0 = (uint32_t)__builtin_amdgcn_readlane((int)plr0, k);
            const uint32_t lw1 = (uint32_t)__builtin_amdgcn_readlane((int)plr1, k);
            if (x >= 0 && x < g.W1) {
                if (!inited) {
                    // first in-range column of this chain: every lane gathers its own position
                    const int p = max(ximg + pown, 0);
#pragma unroll
                    for (int q = 0; q < 6; q++) S[q] = PR[(size_t)q * plane + rowi + p];
                    inited = true;
                } else {
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        const uint32_t nv = (uint32_t)__builtin_amdgcn_readlane((int)run[q], k);   // lane 0's new element
                        S[q] = (uint32_t)__builtin_amdgcn_update_dpp((int)nv, (int)S[q], 0x138, 0xf, 0xf, false);  // wave_shr:1
                    }
                }
                pc[k] = bt_regs(lw0, lw1, S);
            } else {
                // replicated border column: same value as the clamped column (direct gather)
                const int p = max(ximg + pown, 0);
                uint32_t T[6];
#pragma unroll
                for (int q = 0; q < 6; q++) T[q] = PR[(size_t)q * plane + rowi + p];
                pc[k] = bt_regs(lw0, lw1, T);
            }
        }
        // horizontal sliding sums -> vertical ring / accumulators
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < WIN; k++) s = pk_add(s, pc[k]);
#pragma unroll
        for (int j = 0; j < XT; j++) {
            if (j > 0) s = pk_sub(pk_add(s, pc[j + WIN - 1]), pc[j - 1]);
            uint32_t* cell = ring + (size_t)(slot * XT + j) * 64;
            if (rr >= WIN) acc[j] = pk_sub(acc[j], *cell);   // the row leaving the window
            acc[j] = pk_add(acc[j], s);
            *cell = s;
        }
        if (rr >= WIN - 1) {
            const int y = ya + rr - (WIN - 1);
#pragma unroll
            for (int j = 0; j < XT; j++) {
                const int x1 = xa + j;
                if (x1 < g.W1 && 2 * dpl < g.Dp)
                    *(uint32_t*)(C + ((size_t)y * g.W1 + x1) * g.Dp + 2 * dpl) = pad ? MAXC2 : acc[j];
            }
        }
    }
}

// any block size: direct box sum, one thread per (x1, d pair); slow but exact
__global__ void k_sgbm_cost_generic(const uint32_t* __restrict__ PL, const uint32_t* __restrict__ PR,
                                    SgbmGeom g, int16_t* __restrict__ C)
{
    const int dpl = threadIdx.x;
    if (2 * dpl >= g.Dp) return;
    const bool pad = 2 * dpl >= g.D;
    const int dp = pad ? 0 : dpl;
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
    *(uint32_t*)(C + ((size_t)y * g.W1 + x1) * g.Dp + 2 * dpl) = pad ? MAXC2 : acc;
}

// ---------------------------------------------------------------------------------------
// path aggregation: 16 lanes per scan line, NP packed registers (2*NP disparities) per lane
// ---------------------------------------------------------------------------------------
template <int NP>
struct LV { uint32_t r[NP]; };

template <int NP>
__device__ __forceinline__ LV<NP> lv_load(const int16_t* p)
{
    LV<NP> v;
    if constexpr (NP == 1) v.r[0] = *(const uint32_t*)p;
    else if constexpr (NP == 2) { uint2 t = *(const uint2*)p; v.r[0] = t.x; v.r[1] = t.y; }
    else if constexpr (NP % 4 == 0) {
#pragma unroll
        for (int k = 0; k < NP / 4; k++) {
            uint4 t = ((const uint4*)p)[k];
            v.r[4 * k] = t.x; v.r[4 * k + 1] = t.y; v.r[4 * k + 2] = t.z; v.r[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NP; k++) v.r[k] = ((const uint32_t*)p)[k];
    }
    return v;
}
template <int NP>
__device__ __forceinline__ void lv_store(int16_t* p, const LV<NP>& v)
{
    if constexpr (NP == 1) *(uint32_t*)p = v.r[0];
    else if constexpr (NP == 2) *(uint2*)p = make_uint2(v.r[0], v.r[1]);
    else if constexpr (NP % 4 == 0) {
#pragma unroll
        for (int k = 0; k < NP / 4; k++) ((uint4*)p)[k] = make_uint4(v.r[4 * k], v.r[4 * k + 1], v.r[4 * k + 2], v.r[4 * k + 3]);
    } else {
#pragma unroll
        for (int k = 0; k < NP; k++) ((uint32_t*)p)[k] = v.r[k];
    }
}
// Non-temporal variants for the streams that are touched exactly once more (every L volume: written by
// k_sgbm_paths, read by the fused sweep) or never again (C in the fused sweep): they should not displace
// the cost volume, which is re-read by every direction, from L2 / the Infinity Cache.  +6 % pairs/s.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NP>
__device__ __forceinline__ void lv_store_nt(int16_t* p, const LV<NP>& v)
{
    if constexpr (NP % 4 == 0) {
#pragma unroll
        for (int k = 0; k < NP / 4; k++) {
            u32x4 t = { v.r[4 * k], v.r[4 * k + 1], v.r[4 * k + 2], v.r[4 * k + 3] };
            __builtin_nontemporal_store(t, (u32x4*)p + k);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NP; k++) __builtin_nontemporal_store(v.r[k], (uint32_t*)p + k);
    }
}
template <int NP>
__device__ __forceinline__ LV<NP> lv_load_nt(const int16_t* p)
{
    LV<NP> v;
    if constexpr (NP % 4 == 0) {
#pragma unroll
        for (int k = 0; k < NP / 4; k++) {
            u32x4 t = __builtin_nontemporal_load((const u32x4*)p + k);
            v.r[4 * k] = t.x; v.r[4 * k + 1] = t.y; v.r[4 * k + 2] = t.z; v.r[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NP; k++) v.r[k] = __builtin_nontemporal_load((const uint32_t*)p + k);
    }
    return v;
}
template <int NP>
__device__ __forceinline__ LV<NP> lv_fill(uint32_t x)
{
    LV<NP> v;
#pragma unroll
    for (int k = 0; k < NP; k++) v.r[k] = x;
    return v;
}

// L(d) = C(d) + min(Lp(d), Lp(d-1)+P1, Lp(d+1)+P1, delta) - delta  (C carries +P2, delta = minLp + P2)
// padreg bit k set: register k lies beyond D and must stay at MAX_COST
template <int NP>
__device__ __forceinline__ LV<NP> path_step(const LV<NP>& Cp, const LV<NP>& Lp, uint32_t delta2, uint32_t P1_2, unsigned padreg)
{
    LV<NP> out;
    const uint32_t prev_hi = DPP(MAXC2, Lp.r[NP - 1], ROW_SHR1);  // lane-1's last register; row start keeps the sentinel
    const uint32_t next_lo = DPP(MAXC2, Lp.r[0], ROW_SHL1);       // lane+1's first register; row end keeps the sentinel
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const uint32_t left = k == 0 ? prev_hi : Lp.r[k - 1];
        const uint32_t right = k == NP - 1 ? next_lo : Lp.r[k + 1];
        const uint32_t dm1 = __builtin_amdgcn_alignbit(Lp.r[k], left, 16);   // (L[d-1], L[d])
        const uint32_t dp1 = __builtin_amdgcn_alignbit(right, Lp.r[k], 16);  // (L[d+1], L[d+2])
        const uint32_t m = pk_min(pk_min(Lp.r[k], delta2), pk_min(pk_add_sat(dm1, P1_2), pk_add_sat(dp1, P1_2)));
        const uint32_t L = pk_sub(pk_add(Cp.r[k], m), delta2);
        out.r[k] = ((padreg >> k) & 1u) ? MAXC2 : L;
    }
    return out;
}

// the same step with the d-1 / d+1 windows shared between neighbouring registers: window j = (L[2j-1], L[2j]);
// LPL = lanes per scan line (16: a DPP row; 8: half a row, the group's first / last lane take the sentinels explicitly)
template <int NP, bool PAD, int LPL = 16>
__device__ __forceinline__ LV<NP> path_step2(const LV<NP>& Cp, const LV<NP>& Lp, uint32_t delta2, uint32_t P1_2, unsigned padreg, int l = 0)
{
    LV<NP> out;
    uint32_t prev_hi = DPP(MAXC2, Lp.r[NP - 1], ROW_SHR1);
    uint32_t next_lo = DPP(MAXC2, Lp.r[0], ROW_SHL1);
    if constexpr (LPL == 8) {
        prev_hi = l == 0 ? MAXC2 : prev_hi;
        next_lo = l == 7 ? MAXC2 : next_lo;
    }
    uint32_t w[NP + 1];
    w[0] = __builtin_amdgcn_alignbit(Lp.r[0], prev_hi, 16);
#pragma unroll
    for (int k = 1; k < NP; k++) w[k] = __builtin_amdgcn_alignbit(Lp.r[k], Lp.r[k - 1], 16);
    w[NP] = __builtin_amdgcn_alignbit(next_lo, Lp.r[NP - 1], 16);
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const uint32_t nb = pk_add_sat(pk_min(w[k], w[k + 1]), P1_2);     // min(a,b)+P1 = min(a+P1, b+P1) under saturation
        const uint32_t m = pk_min(pk_min(Lp.r[k], delta2), nb);
        const uint32_t L = pk_sub(pk_add(Cp.r[k], m), delta2);
        out.r[k] = (PAD && ((padreg >> k) & 1u)) ? MAXC2 : L;
    }
    return out;
}

// all-reduce (min, unsigned) inside each group of 8 lanes
__device__ __forceinline__ uint32_t half_min_u32(uint32_t v)
{
    v = min(v, DPP(0xFFFFFFFFu, v, 0xB1));    // quad_perm [1,0,3,2]
    v = min(v, DPP(0xFFFFFFFFu, v, 0x4E));    // quad_perm [2,3,0,1]
    v = min(v, DPP(0xFFFFFFFFu, v, 0x141));   // row_half_mirror
    return v;
}

template <int NP>
__device__ __forceinline__ uint32_t lane_min16(const LV<NP>& v)
{
    uint32_t m = v.r[0];
#pragma unroll
    for (int k = 1; k < NP; k++) m = pk_min(m, v.r[k]);
    return min(m & 0xFFFFu, m >> 16);
}

#define VO_MAX_DIRS 8
struct PathPlan {
    int n_dirs;
    int sx[VO_MAX_DIRS], sy[VO_MAX_DIRS];
    int nlines[VO_MAX_DIRS];
    int first_wave[VO_MAX_DIRS + 1];  // prefix sum of ceil(nlines / lines per wave)
    int lpw;                          // lines per wave (host side: 4 = 16 lanes per line, 8 = 8 lanes per line)
};

// NP = registers per lane (2*NP disparities), LPL = lanes per scan line (LPL * 2 * NP = Dp), 64 / LPL lines per wave.
// Fewer lanes per line = more disparities per lane: the per-step fixed work (addresses, the min all-reduce, delta) is
// spread over more cells and the launch has half the waves.
// (TAG only makes the launch of the fused W+E schedule -- NW and NE alone, 1 -- a kernel name of its own in profiles)
template <int NP, int PF, int LPL = 16, bool PAD = true, int TAG = 0>
__global__ void __launch_bounds__(256) k_sgbm_paths(const int16_t* __restrict__ C, int16_t* __restrict__ Lbase, size_t vol,
                                                   SgbmGeom g, PathPlan plan, int16_t* __restrict__ dump)
{
    constexpr int LPW = 64 / LPL;
    const int lane = threadIdx.x & 63, row = lane / LPL, l16 = lane % LPL;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= plan.first_wave[plan.n_dirs]) return;
    int dir = 0;
    while (dir + 1 < plan.n_dirs && wave >= plan.first_wave[dir + 1]) dir++;
    const int sx = plan.sx[dir], sy = plan.sy[dir];
    const int line = (wave - plan.first_wave[dir]) * LPW + row;
    const int W1 = g.W1, H = g.H;
    int x0 = 0, y0 = 0, n = 0;
    if (line < plan.nlines[dir]) {
        if (sy == 0) { y0 = line; x0 = sx > 0 ? 0 : W1 - 1; }
        else if (sx == 0) { x0 = line; y0 = sy > 0 ? 0 : H - 1; }
        else {
            const int ytop = sy > 0 ? 0 : H - 1;
            if (line < W1) { x0 = line; y0 = ytop; }
            else { x0 = sx > 0 ? 0 : W1 - 1; y0 = ytop + sy * (line - W1 + 1); }
        }
        const int nx = sx > 0 ? W1 - x0 : (sx < 0 ? x0 + 1 : 1 << 30);
        const int ny = sy > 0 ? H - y0 : (sy < 0 ? y0 + 1 : 1 << 30);
        n = min(nx, ny);
    }
    int nmax = 0;
#pragma unroll
    for (int r = 0; r < LPW; r++) nmax = max(nmax, __builtin_amdgcn_readlane(n, r * LPL));

    const int d0 = l16 * 2 * NP;
    unsigned padreg = 0;
    if constexpr (PAD) {
#pragma unroll
        for (int k = 0; k < NP; k++) padreg |= (unsigned)(d0 + 2 * k >= g.D) << k;
    }
    const uint32_t P1_2 = pk_rep(g.P1);
    const uint32_t P2_2 = pk_rep(g.P2);
    const ptrdiff_t stride = ((ptrdiff_t)sy * W1 + sx) * g.Dp;
    const size_t start = ((size_t)y0 * W1 + x0) * g.Dp + d0;
    const int16_t* cp = C + start;
    int16_t* lp = Lbase + (size_t)dir * vol + start;

    LV<NP> Lp;
#pragma unroll
    for (int k = 0; k < NP; k++) Lp.r[k] = ((padreg >> k) & 1u) ? MAXC2 : 0u;  // predecessor outside: zeros, min 0
    uint32_t delta2 = P2_2;

    // No lane-dependent branch may guard a load or a store in the loop: behind one the compiler has to
    // drain the whole memory queue (s_waitcnt vmcnt(0)) at the join and the prefetch depth is lost.
    // So every lane always loads (index clamped to its line) and always stores (lanes past the end of
    // their line write their 16..32 bytes into a dump area).
    const int last = max(n, 1) - 1;
    int16_t* const sink = dump + lane * 2 * NP;
    LV<NP> cbuf[PF];  // PF = software prefetch depth (steps)
#pragma unroll
    for (int k = 0; k < PF; k++) cbuf[k] = lv_load<NP>(cp + (ptrdiff_t)min(k, last) * stride);
    const int16_t* pld = cp + (ptrdiff_t)min(PF, last) * stride;   // next cell to prefetch
    int16_t* pst = lp;                                              // cell of the current step
    // (the trip count is rounded up to whole groups of PF steps -- the surplus steps only feed the sink --
    // so that the unrolled body is straight-line code and the waits stay partial)
    // (TAG 2) every line of this wave starts on the image's top row
    const bool alltop = __builtin_amdgcn_readfirstlane((int)(__ballot(n > 0 && y0 != 0) == 0ull)) != 0;
    for (int i0 = 0; i0 < nmax; i0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int i = i0 + k;
            const LV<NP> Cv = cbuf[k];
            cbuf[k] = lv_load<NP>(pld);
            pld += (i + PF < last) ? stride : 0;
            const LV<NP> L = path_step2<NP, PAD, LPL>(Cv, Lp, delta2, P1_2, padreg, l16);
            const uint32_t lm = lane_min16<NP>(L);
            const uint32_t mn = LPL == 8 ? half_min_u32(lm) : row_min_u32(lm);
            delta2 = pk_add(pk_rep((int)mn), P2_2);
            Lp = L;
            if constexpr (TAG == 2) {
                // band schedule: only the last row of every 8-row band is kept, as rk[dir][y >> 3][x][d] (`vol` carries the
                // number of bands; every direction of such a launch steps one row down per step).  Lines that start on
                // the top row (all of N, most of NW / NE) are in row i at step i: for a wave made of such lines the
                // store is the unrolled body's 8th step and nothing else; the other waves store every step, mostly
                // into the dump area.
                const int yy = y0 + i, xx = x0 + i * sx;
                const bool keep = i < n && (yy & 7) == 7;
                if (!alltop || (PF == 8 && k == 7))
                    lv_store<NP>(keep ? Lbase + (((size_t)dir * vol + (size_t)(yy >> 3)) * W1 + xx) * g.Dp + d0 : sink, L);
            } else {
                lv_store_nt<NP>(i < n ? pst : sink, L);
                pst += stride;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// the two horizontal directions as ONE stored volume (MODE_SGBM, W1 a multiple of 8)
// ---------------------------------------------------------------------------------------
// W and E run along the same image row in opposite directions, so their sum L_W + L_E could be written as one volume
// if both were known at a pixel at the same time.  The E sweep cannot be held on chip (a row of L is 295 KB at C2), but
// it can be RECOMPUTED in pieces: pass 1 runs E right-to-left over the row and keeps only every 8th column's L (a
// checkpoint, 1/8 of a volume); pass 2 walks left-to-right in segments of 8 columns, re-runs E inside the segment from
// the checkpoint at its right edge (8 columns of L_E in registers), then advances W through the same 8 columns and
// stores sat(L_W + L_E).  All operands are non-negative, so the saturating sum is order-independent and the fused
// vertical sweep adds this one volume where it used to add two.  Traffic of the two horizontal directions: C read
// twice, 1/8 volume written and read, one volume written = 3.25 passes instead of 4, and one volume less for the final
// sweep to read -- 12.25 passes per pair instead of 14; the price is a third path step per pixel and rows that are
// three sweeps long.  16 lanes per row, 4 rows per wave, as k_sgbm_paths.
template <int NP, bool PAD>
__global__ void __launch_bounds__(256) k_sgbm_we(const int16_t* __restrict__ C, int16_t* __restrict__ Swe, int16_t* __restrict__ ckpt,
                                                SgbmGeom g, int16_t* __restrict__ dump)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int y = wave * 4 + row;
    if (wave * 4 >= g.H) return;
    const bool live = y < g.H;
    const int yc = live ? y : g.H - 1;
    const int W1 = g.W1, nseg = W1 >> 3;
    const int d0 = l16 * 2 * NP;
    unsigned padreg = 0;
    if constexpr (PAD) {
#pragma unroll
        for (int k = 0; k < NP; k++) padreg |= (unsigned)(d0 + 2 * k >= g.D) << k;
    }
    const uint32_t P1_2 = pk_rep(g.P1), P2_2 = pk_rep(g.P2);
    const ptrdiff_t Dp = g.Dp;
    const int16_t* crow = C + (size_t)yc * W1 * Dp + d0;              // this row of C, this lane's disparities
    int16_t* const sink = dump + lane * 2 * NP;
    int16_t* const krow = live ? ckpt + (size_t)yc * nseg * Dp + d0 : sink;   // checkpoints of this row: L_E at x = 8 j
    const ptrdiff_t kstep = live ? Dp : 0;
    LV<NP> border;
#pragma unroll
    for (int k = 0; k < NP; k++) border.r[k] = ((padreg >> k) & 1u) ? MAXC2 : 0u;

    // ---- pass 1: E, right to left, keeping every 8th column ------------------------------------------------------
    {
        LV<NP> Lp = border;
        uint32_t delta2 = P2_2;
        LV<NP> cbuf[8];
#pragma unroll
        for (int k = 0; k < 8; k++) cbuf[k] = lv_load<NP>(crow + (ptrdiff_t)(W1 - 1 - k) * Dp);
        const int16_t* pld = crow + (ptrdiff_t)(W1 - 9) * Dp;       // (W1 >= 16 is checked by the launcher)
        for (int s = nseg - 1; s >= 0; s--) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const LV<NP> Cv = cbuf[k];
                cbuf[k] = lv_load<NP>(pld);
                pld -= (8 * s - k - 1 > 0) ? Dp : 0;               // next column to fetch is 8 s - k - 1; stop at 0
                const LV<NP> L = path_step2<NP, PAD>(Cv, Lp, delta2, P1_2, padreg);
                delta2 = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(L))), P2_2);
                Lp = L;
                if (k == 7) lv_store<NP>(krow + (ptrdiff_t)s * kstep, L);   // column 8 s
            }
        }
    }
    // (the checkpoints are read back by the lanes that wrote them: let the stores land before the first load is issued)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- pass 2: per segment E again (from the checkpoint on its right), then W, store the sum -------------------
    {
        LV<NP> LpW = border;
        uint32_t dW = P2_2;
        LV<NP> cseg[8], cnext[8], ck, cknext;
#pragma unroll
        for (int k = 0; k < 8; k++) cseg[k] = lv_load<NP>(crow + (ptrdiff_t)k * Dp);
        ck = lv_load<NP>(krow + (ptrdiff_t)(nseg > 1 ? 1 : 0) * kstep);
        int16_t* pst = live ? Swe + (size_t)yc * W1 * Dp + d0 : sink;
        const ptrdiff_t sstep = live ? Dp : 0;
        for (int s = 0; s < nseg; s++) {
            // prefetch the next segment's costs and checkpoint (clamped at the row's end)
            const int sn = min(s + 1, nseg - 1);
#pragma unroll
            for (int k = 0; k < 8; k++) cnext[k] = lv_load<NP>(crow + (ptrdiff_t)(8 * sn + k) * Dp);
            cknext = lv_load<NP>(krow + (ptrdiff_t)min(sn + 1, nseg - 1) * kstep);
            const bool last = s == nseg - 1;                          // the row's last segment starts E from the border
            LV<NP> LpE;
#pragma unroll
            for (int q = 0; q < NP; q++) LpE.r[q] = last ? border.r[q] : ck.r[q];
            uint32_t dE = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(ck))), P2_2);
            dE = last ? P2_2 : dE;
            LV<NP> Le[8];
#pragma unroll
            for (int k = 7; k >= 0; k--) {
                Le[k] = path_step2<NP, PAD>(cseg[k], LpE, dE, P1_2, padreg);
                dE = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(Le[k]))), P2_2);
                LpE = Le[k];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const LV<NP> Lw = path_step2<NP, PAD>(cseg[k], LpW, dW, P1_2, padreg);
                dW = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(Lw))), P2_2);
                LpW = Lw;
                LV<NP> S;
#pragma unroll
                for (int q = 0; q < NP; q++) S.r[q] = pk_add_sat(Lw.r[q], Le[k].r[q]);
                lv_store_nt<NP>(pst, S);
                pst += sstep;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) cseg[k] = cnext[k];
            ck = cknext;
        }
    }
}

// The same pairing for ANY two opposite directions (MODE_HH: W/E, NW/SE, NE/SW): a scan line of direction (sx, sy) is
// also a scan line of (-sx, -sy) walked from its other end.  Pass 1 runs the backward direction from the line's far end
// and keeps L at every 8th step (step index i = 8 j, stored AT the pixel it belongs to, in a scratch volume that is
// otherwise untouched: 1/8 of its lines are written); pass 2 walks forward in 8-step segments, re-runs the backward
// direction inside the segment from the checkpoint at step 8 (j + 1), advances the forward direction and stores
// sat(L_fwd + L_bwd) at the pixel.  Lines of a wave differ in length (diagonals): steps past a line's end are computed
// and discarded by selects, their loads clamped to the line and their stores sent to the dump area, so the body stays
// straight-line code.  One launch covers all pairs of the plan (wave ranges per pair, as k_sgbm_paths).
template <int NP, bool PAD>
__global__ void __launch_bounds__(256) k_sgbm_pair(const int16_t* __restrict__ C, int16_t* __restrict__ Sbase, int16_t* __restrict__ Kbase,
                                                  size_t vol, SgbmGeom g, PathPlan plan, int16_t* __restrict__ dump)
{
    const int lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= plan.first_wave[plan.n_dirs]) return;
    int dir = 0;
    while (dir + 1 < plan.n_dirs && wave >= plan.first_wave[dir + 1]) dir++;
    const int sx = plan.sx[dir], sy = plan.sy[dir];
    const int line = (wave - plan.first_wave[dir]) * 4 + row;
    const int W1 = g.W1, H = g.H;
    int x0 = 0, y0 = 0, n = 0;
    if (line < plan.nlines[dir]) {
        if (sy == 0) { y0 = line; x0 = sx > 0 ? 0 : W1 - 1; }
        else if (sx == 0) { x0 = line; y0 = sy > 0 ? 0 : H - 1; }
        else {
            const int ytop = sy > 0 ? 0 : H - 1;
            if (line < W1) { x0 = line; y0 = ytop; }
            else { x0 = sx > 0 ? 0 : W1 - 1; y0 = ytop + sy * (line - W1 + 1); }
        }
        const int nx = sx > 0 ? W1 - x0 : (sx < 0 ? x0 + 1 : 1 << 30);
        const int ny = sy > 0 ? H - y0 : (sy < 0 ? y0 + 1 : 1 << 30);
        n = min(nx, ny);
    }
    int nmax = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) nmax = max(nmax, __builtin_amdgcn_readlane(n, r * 16));
    const int nsegs = (nmax + 7) >> 3;

    const int d0 = l16 * 2 * NP;
    unsigned padreg = 0;
    if constexpr (PAD) {
#pragma unroll
        for (int k = 0; k < NP; k++) padreg |= (unsigned)(d0 + 2 * k >= g.D) << k;
    }
    const uint32_t P1_2 = pk_rep(g.P1), P2_2 = pk_rep(g.P2);
    const ptrdiff_t stride = ((ptrdiff_t)sy * W1 + sx) * g.Dp;      // cells per forward step
    const size_t start = ((size_t)y0 * W1 + x0) * g.Dp + d0;
    const int16_t* cp = C + start;
    int16_t* const sp = Sbase + (size_t)dir * vol + start;
    int16_t* const kp = Kbase + (size_t)dir * vol + start;
    int16_t* const sink = dump + lane * 2 * NP;
    const int last = max(n, 1) - 1;
    LV<NP> border;
#pragma unroll
    for (int k = 0; k < NP; k++) border.r[k] = ((padreg >> k) & 1u) ? MAXC2 : 0u;

    // ---- pass 1: the backward direction from step n - 1 down to 0, checkpoints at steps 8 j ----------------------
    {
        LV<NP> Lp = border;
        uint32_t delta2 = P2_2;
        for (int j = nsegs - 1; j >= 0; j--) {
            LV<NP> cseg[8];
#pragma unroll
            for (int k = 0; k < 8; k++) cseg[k] = lv_load<NP>(cp + (ptrdiff_t)min(8 * j + k, last) * stride);
#pragma unroll
            for (int k = 7; k >= 0; k--) {
                const int i = 8 * j + k;
                const bool on = i < n;                               // steps past this line's end leave the state alone
                const LV<NP> L = path_step2<NP, PAD>(cseg[k], Lp, delta2, P1_2, padreg);
                const uint32_t dn = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(L))), P2_2);
#pragma unroll
                for (int q = 0; q < NP; q++) Lp.r[q] = on ? L.r[q] : Lp.r[q];
                delta2 = on ? dn : delta2;
                if (k == 0) lv_store<NP>(on ? kp + (ptrdiff_t)i * stride : sink, L);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // checkpoint stores land before their lanes read them back
    // ---- pass 2: per segment the backward direction again (from the checkpoint at step 8 j + 8), then forward ------
    {
        LV<NP> LpF = border;
        uint32_t dF = P2_2;
        for (int j = 0; j < nsegs; j++) {
            LV<NP> cseg[8];
#pragma unroll
            for (int k = 0; k < 8; k++) cseg[k] = lv_load<NP>(cp + (ptrdiff_t)min(8 * j + k, last) * stride);
            const bool has_ck = 8 * j + 8 < n;                       // otherwise the backward direction starts inside this segment
            const LV<NP> ck = lv_load<NP>(has_ck ? kp + (ptrdiff_t)(8 * j + 8) * stride : sink);
            LV<NP> LpB;
#pragma unroll
            for (int q = 0; q < NP; q++) LpB.r[q] = has_ck ? ck.r[q] : border.r[q];
            uint32_t dB = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(ck))), P2_2);
            dB = has_ck ? dB : P2_2;
            LV<NP> Lb[8];
#pragma unroll
            for (int k = 7; k >= 0; k--) {
                const bool on = 8 * j + k < n;
                const LV<NP> L = path_step2<NP, PAD>(cseg[k], LpB, dB, P1_2, padreg);
                const uint32_t dn = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(L))), P2_2);
                Lb[k] = L;
#pragma unroll
                for (int q = 0; q < NP; q++) LpB.r[q] = on ? L.r[q] : LpB.r[q];
                dB = on ? dn : dB;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = 8 * j + k;
                const bool on = i < n;
                const LV<NP> Lf = path_step2<NP, PAD>(cseg[k], LpF, dF, P1_2, padreg);
                const uint32_t dn = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(Lf))), P2_2);
#pragma unroll
                for (int q = 0; q < NP; q++) LpF.r[q] = on ? Lf.r[q] : LpF.r[q];
                dF = on ? dn : dF;
                LV<NP> S;
#pragma unroll
                for (int q = 0; q < NP; q++) S.r[q] = pk_add_sat(Lf.r[q], Lb[k].r[q]);
                lv_store_nt<NP>(on ? sp + (ptrdiff_t)i * stride : sink, S);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// winner-take-all: 16 lanes per pixel
// ---------------------------------------------------------------------------------------
// winner of one pixel's summed costs: lowest S (first d on ties) and the uniqueness verdict, per 16-lane row.
// THR (needs uniquenessRatio < 100): the uniqueness test S*(100-ur) < minS*100 is rewritten as
// S < T = ceil(minS*100 / (100-ur)) (all operands non-negative integers) and evaluated on packed
// halves with no branches; the generic form keeps OpenCV's products.
template <int NP, bool PAD, bool THR>
__device__ __forceinline__ void wta_core(const LV<NP>& S, const SgbmGeom& g, int lane, int& minS, int& best, bool& row_viol)
{
    const int l16 = lane & 15, d0 = l16 * 2 * NP;
    uint32_t key = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const int d = d0 + 2 * k;
        if (!PAD || d < g.D) {
            const uint32_t k0 = ((S.r[k] & 0xFFFFu) << 8) | (uint32_t)d;
            const uint32_t k1 = ((S.r[k] >> 16) << 8) | (uint32_t)(d + 1);
            key = min(key, min(k0, k1));
        }
    }
    key = row_min_u32(key);
    minS = (int)(key >> 8);
    best = (int)(key & 255u);
    bool viol = false;
    const int ur100 = 100 - g.ur;
    if (THR) {
        // exact ceil(minS*100 / ur100): float estimate (operands < 2^24) + integer fix-up
        const int a = minS * 100;
        int T = (int)((float)a * __builtin_amdgcn_rcpf((float)ur100));
        T += (T * ur100 < a);
        T += (T * ur100 < a);
        T -= ((T - 1) * ur100 >= a);
        T -= ((T - 1) * ur100 >= a);
        const uint32_t T2 = pk_rep(min(T, 32768));   // S <= 32767: a larger T changes nothing
        // bit e of m8: this lane's element e (d = d0 + e) lies within one of the winner -- never a violation
        const int t = min(max(best + 4 - d0, 0), 31);
        uint32_t m8 = ((7u << t) >> 5) & ((1u << (2 * NP)) - 1u);   // 2NP <= 16 elements per lane
        if (PAD) {
            const int npad = min(max(d0 + 2 * NP - g.D, 0), 2 * NP);   // trailing elements beyond D
            m8 |= (0xFFFFu << (2 * NP - npad)) & ((1u << (2 * NP)) - 1u);
        }
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)m8, 2 * k, 1);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)m8, 2 * k + 1, 1);
            const uint32_t excl = (lo & 0xFFFFu) | (hi & 0xFFFF0000u);
            any |= pk_sub_sat_u(T2, S.r[k]) & ~excl;   // a non-zero half: S < T
        }
        viol = any != 0;
    } else {
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int d = d0 + 2 * k;
            if (!PAD || d < g.D) {
                const int s0 = (int)(S.r[k] & 0xFFFFu), s1 = (int)(S.r[k] >> 16);
                viol |= (s0 * ur100 < minS * 100) && (abs(best - d) > 1);
                viol |= (s1 * ur100 < minS * 100) && (abs(best - d - 1) > 1);
            }
        }
    }
    const unsigned long long bal = __ballot(viol);
    row_viol = ((bal >> (lane & 48)) & 0xFFFFull) != 0ull;
}

// sub-pixel disparity (x16) of a winner from its two neighbours' costs; C's truncating division
__device__ __forceinline__ int wta_subpixel(int best, int minS, int sm, int sp)
{
    const int denom2 = max(sm + sp - 2 * minS, 1);
    return best * 16 + ((sm - sp) * 16 + denom2) / (denom2 * 2);
}

// WTA of one pixel held by a 16-lane row: S = aggregated costs (pads irrelevant), myS = this row's LDS
// scratch (Dp int16).  Writes disp1 and the disp2 key.  All 64 lanes must call it together.
// S = sum over directions for the Dp disparities of one pixel, spread over 16 lanes
// (lane l16 holds d = l16*2NP .. +2NP-1).  Winner = lowest S, first d on ties (stereosgbm.cpp
// computeDisparitySGBM: "if (Sval < minS) { minS = Sval; bestDisp = d; }").
template <int NP, bool PAD, bool THR>
__device__ __forceinline__ void wta_pixel(const LV<NP>& S, const SgbmGeom& g, int lane, bool live, int x1, int y,
                                          int16_t* myS, int16_t* __restrict__ disp1, int* __restrict__ d2key)
{
    const int l16 = lane & 15, d0 = l16 * 2 * NP;
    int minS, best;
    bool row_viol;
    wta_core<NP, PAD, THR>(S, g, lane, minS, best, row_viol);
    lv_store<NP>(myS + d0, S);
    __builtin_amdgcn_wave_barrier();   // LDS accesses of one wave are served in order
    if (live && l16 == 0) {
        const int ximg = x1 + g.minX1;
        int out = g.invalid16;
        if (!row_viol) {
            int dd = best * 16;
            if (best > 0 && best < g.D - 1) dd = wta_subpixel(best, minS, myS[best - 1], myS[best + 1]);
            out = dd + g.minD * 16;
            // disp2: lowest cost wins, ties go to the pixel OpenCV scans first (largest x)
            const int x2 = ximg - best - g.minD;
            atomicMin(&d2key[(size_t)y * g.W + x2], (minS << 16) | (0xFFFF - ximg));
        }
        disp1[(size_t)y * g.W + ximg] = (int16_t)out;
    }
    __builtin_amdgcn_wave_barrier();
}

// Second half of the fused sweep's WTA, one thread per pixel: the sweep leaves a two-word record
// (aux0 = minS << 8 | best or -1 when the uniqueness test failed, aux1 = S[best-1] << 16 | S[best+1]);
// this pass turns it into the sub-pixel disp1 and the disp2 candidates (atomicMin).
__global__ void k_sgbm_fin(const int* __restrict__ aux0, const int* __restrict__ aux1, SgbmGeom g,
                           int16_t* __restrict__ disp1, int* __restrict__ d2key)
{
    const int x1 = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x1 >= g.W1) return;
    const int ximg = x1 + g.minX1;
    const size_t o = (size_t)y * g.W + ximg;
    const int a = aux0[o];
    int out = g.invalid16;
    if (a >= 0) {
        const int minS = a >> 8, best = a & 255;
        int dd = best * 16;
        if (best > 0 && best < g.D - 1) {
            const uint32_t n = (uint32_t)aux1[o];
            dd = wta_subpixel(best, minS, (int)(n >> 16), (int)(n & 0xFFFFu));
        }
        out = dd + g.minD * 16;
        atomicMin(&d2key[(size_t)y * g.W + ximg - best - g.minD], (minS << 16) | (0xFFFF - ximg));
    }
    disp1[o] = (int16_t)out;
}

template <int NP>
__global__ void __launch_bounds__(256) k_sgbm_wta(const int16_t* __restrict__ Lbase, size_t vol, int nvol, SgbmGeom g,
                                                 int16_t* __restrict__ disp1, int* __restrict__ d2key)
{
    extern __shared__ int16_t s_S[];  // [blockDim/16][Dp]
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long long pid = (long long)blockIdx.x * (blockDim.x >> 4) + grp;
    const long long npix = (long long)g.W1 * g.H;
    const bool live = pid < npix;
    const int y = live ? (int)(pid / g.W1) : 0, x1 = live ? (int)(pid - (long long)y * g.W1) : 0;
    const int d0 = l16 * 2 * NP;
    const size_t cell = ((size_t)y * g.W1 + x1) * g.Dp + d0;
    LV<NP> S = lv_load<NP>(Lbase + cell);
    for (int v = 1; v < nvol; v++) {
        const LV<NP> t = lv_load<NP>(Lbase + (size_t)v * vol + cell);
#pragma unroll
        for (int k = 0; k < NP; k++) S.r[k] = pk_add_sat(S.r[k], t.r[k]);
    }
    int16_t* myS = s_S + (size_t)grp * g.Dp;
    const int lane = threadIdx.x & 63;
    if (g.ur < 100) {
        if (g.D == g.Dp) wta_pixel<NP, false, true>(S, g, lane, live, x1, y, myS, disp1, d2key);
        else wta_pixel<NP, true, true>(S, g, lane, live, x1, y, myS, disp1, d2key);
    } else {
        wta_pixel<NP, true, false>(S, g, lane, live, x1, y, myS, disp1, d2key);
    }
}

// ---------------------------------------------------------------------------------------
// band schedule (MODE_SGBM): the three top-down directions are never stored as volumes
// ---------------------------------------------------------------------------------------
// N, NW and NE of a row depend on the row above only.  A pre-pass (k_sgbm_paths<..., 2>) walks their lines as usual
// but keeps only the last row of every 8-row band (3/8 of a volume).  This kernel then finishes the job band by band,
// in parallel: a workgroup owns TW columns of one band, loads the checkpoint row above it for TW + 16 columns, and
// recomputes the three directions row by row in LDS -- the columns it needs shrink by one on each side per row, so the
// 8-column halos are all it ever reads from its neighbours' territory -- adding the stored W+E volume and running the
// winner-take-all for its own columns.  Nothing here waits for another workgroup, and the final stage has no 720-step
// serial sweep any more.  16 lanes per pixel, 16 pixels per 256-thread workgroup per step; two state buffers (row above /
// this row) of 3 directions x (TW + 16) columns x Dp cells in LDS; columns outside the image hold the border state
// (zeros), whose minimum is 0, so "predecessor outside" needs no special case.
template <int NP, bool PAD, int TW>
__global__ void __launch_bounds__(256) k_sgbm_band(const int16_t* __restrict__ C, const int16_t* __restrict__ Swe, const int16_t* __restrict__ rk,
                                                  int nbands, SgbmGeom g, int16_t* __restrict__ disp1, int* __restrict__ d2key)
{
    constexpr int CW = TW + 16;
    extern __shared__ __attribute__((aligned(16))) int16_t s_band[];   // [2][3][CW][Dp] state, then [16][Dp] WTA scratch
    const int lane = threadIdx.x & 63, l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;   // 16 pixel groups
    const int Dp = g.Dp, W1 = g.W1, H = g.H;
    const int band = blockIdx.y, x0 = blockIdx.x * TW, x1 = min(x0 + TW, W1);
    const int base = x0 - 8;                                             // image column of state column 0
    const int d0 = l16 * 2 * NP;
    unsigned padreg = 0;
    if constexpr (PAD) {
#pragma unroll
        for (int k = 0; k < NP; k++) padreg |= (unsigned)(d0 + 2 * k >= g.D) << k;
    }
    const uint32_t P1_2 = pk_rep(g.P1), P2_2 = pk_rep(g.P2);
    LV<NP> border;
#pragma unroll
    for (int k = 0; k < NP; k++) border.r[k] = ((padreg >> k) & 1u) ? MAXC2 : 0u;
    const size_t bufsz = (size_t)3 * CW * Dp;
    int16_t* myS = s_band + 2 * bufsz + (size_t)grp * Dp;
    // state of the row above the band: the checkpoints of band - 1 (border for the top band and outside the image);
    // the other buffer starts as border everywhere so that out-of-image columns read as border on every row
    for (int it = grp; it < 3 * CW; it += 16) {
        const int dir = it / CW, col = it % CW, x = base + col;
        LV<NP> v = border;
        if (band > 0 && x >= 0 && x < W1) v = lv_load<NP>(rk + (((size_t)dir * nbands + (band - 1)) * W1 + x) * Dp + d0);
        lv_store<NP>(s_band + ((size_t)dir * CW + col) * Dp + d0, v);
        lv_store<NP>(s_band + bufsz + ((size_t)dir * CW + col) * Dp + d0, border);
    }
    __syncthreads();
    int cur = 1;
    for (int r = 0; r < 8; r++) {
        const int y = band * 8 + r;
        if (y >= H) break;                                               // (uniform)
        const int lo = max(0, x0 - (7 - r)), hi = min(W1, x1 + (7 - r));
        const int16_t* prev = s_band + (size_t)(cur ^ 1) * bufsz;
        int16_t* now = s_band + (size_t)cur * bufsz;
        const size_t rowoff = (size_t)y * W1;
        // this group's columns of the row: lo + grp, + 16, + 32 ... (at most (TW + 14 + 15) / 16 of them)
        constexpr int MAXI = (TW + 14 + 15) / 16;
        LV<NP> cv[MAXI], sw[MAXI];
#pragma unroll
        for (int q = 0; q < MAXI; q++) {
            const int c = min(lo + grp + 16 * q, W1 - 1);
            cv[q] = lv_load<NP>(C + (rowoff + c) * Dp + d0);
            const int cs = (c >= x0 && c < x1) ? c : x0;               // halo columns have no pixel to decide: re-read a cached cell
            sw[q] = lv_load_nt<NP>(Swe + (rowoff + cs) * Dp + d0);
        }
#pragma unroll
        for (int q = 0; q < MAXI; q++) {
            const int c = lo + grp + 16 * q;
            const bool act = c < hi;                                     // lane-group uniform
            const int cc = act ? c - base : 8;                           // (inactive groups replay a harmless column into scratch-free registers)
            LV<NP> S = sw[q];
#pragma unroll
            for (int dir = 0; dir < 3; dir++) {
                const int pc = cc + (dir == 1 ? -1 : (dir == 2 ? 1 : 0));   // predecessor column: N same, NW left, NE right
                const LV<NP> Lp = lv_load<NP>(prev + ((size_t)dir * CW + pc) * Dp + d0);
                const uint32_t delta2 = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(Lp))), P2_2);
                const LV<NP> L = path_step2<NP, PAD>(cv[q], Lp, delta2, P1_2, padreg);
                if (act) lv_store<NP>(now + ((size_t)dir * CW + cc) * Dp + d0, L);
#pragma unroll
                for (int k = 0; k < NP; k++) S.r[k] = pk_add_sat(S.r[k], L.r[k]);
            }
            const bool live = act && c >= x0 && c < x1;
            if (__ballot(live) != 0ull)                                  // (wave-uniform: halo-only waves skip the winner search)
                wta_pixel<NP, PAD, true>(S, g, lane, live, c, y, myS, disp1, d2key);
        }
        __syncthreads();
        cur ^= 1;
    }
}

// Last aggregation direction (predecessor (x, y-1), swept top to bottom) fused with the WTA: the wave
// walks 4 adjacent image columns; per row it reads C and the other directions' L volumes, advances
// its own path in registers, sums and picks the winner -- this direction's L is never written and
// the WTA needs no pass of its own.
template <int NP, int NV, bool PAD>
__global__ void __launch_bounds__(256) k_sgbm_vwta(const int16_t* __restrict__ C, const int16_t* __restrict__ Lbase, size_t vol,
                                                  SgbmGeom g, int* __restrict__ aux0, int* __restrict__ aux1)
{
    extern __shared__ int16_t s_S[];  // [blockDim/16][2][Dp]
    const int lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int x1 = wave * 4 + row;
    if (wave * 4 >= g.W1) return;
    const bool live = x1 < g.W1;
    const int xc = live ? x1 : g.W1 - 1;          // dead rows shadow a valid column, never store
    const int d0 = l16 * 2 * NP;
    unsigned padreg = 0;
    if (PAD) {
#pragma unroll
        for (int k = 0; k < NP; k++) padreg |= (unsigned)(d0 + 2 * k >= g.D) << k;
    }
    const uint32_t P1_2 = pk_rep(g.P1), P2_2 = pk_rep(g.P2);
    const size_t stride = (size_t)g.W1 * g.Dp;
    const uint32_t start = (uint32_t)(xc * g.Dp + d0);   // lane offset inside one image row of the volume
    int16_t* myS = s_S + (size_t)(threadIdx.x >> 4) * 2 * g.Dp;   // two rows of S: this step's and the previous one's
    LV<NP> Lp;
#pragma unroll
    for (int k = 0; k < NP; k++) Lp.r[k] = ((padreg >> k) & 1u) ? MAXC2 : 0u;
    uint32_t delta2 = P2_2;
    // The winner's record is written one step late: its two neighbour costs come back from LDS while
    // the next row is being computed, so the sweep never waits on the LDS round trip.
    const bool writer = live && l16 == 0;
    uint32_t aoff = (uint32_t)(x1 + g.minX1);   // record index of the PREVIOUS row's pixel
    int prec = -1, pbest = 0, par = 0;
    // rows in flight: one lone wave per SIMD must cover the ~2 us HBM latency by itself, within ~128 VGPRs
    constexpr int RPS = NP * (NV + 1);   // registers per prefetched row
    constexpr int PF = RPS <= 20 ? 6 : RPS <= 32 ? 4 : RPS <= 48 ? 3 : 2;
    LV<NP> cbuf[PF], lbuf[PF][NV];
    // wave-uniform row pointers advance by one row per step; the lane part stays a 32-bit offset
    const int16_t* rowC = C;
    const int16_t* rowL = Lbase;
#pragma unroll
    for (int k = 0; k < PF; k++) {
        if (k < g.H) {
            cbuf[k] = lv_load_nt<NP>(rowC + start);
#pragma unroll
            for (int v = 0; v < NV; v++) lbuf[k][v] = lv_load_nt<NP>(rowL + (size_t)v * vol + start);
            rowC += stride;
            rowL += stride;
        }
    }
    for (int y0 = 0; y0 < g.H; y0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int y = y0 + k;
            if (y < g.H) {
                const LV<NP> Cv = cbuf[k];
                LV<NP> S = lbuf[k][0];
#pragma unroll
                for (int v = 1; v < NV; v++)
#pragma unroll
                    for (int q = 0; q < NP; q++) S.r[q] = pk_add_sat(S.r[q], lbuf[k][v].r[q]);
                if (y + PF < g.H) {
                    cbuf[k] = lv_load_nt<NP>(rowC + start);
#pragma unroll
                    for (int v = 0; v < NV; v++) lbuf[k][v] = lv_load_nt<NP>(rowL + (size_t)v * vol + start);
                    rowC += stride;
                    rowL += stride;
                }
                const int i0 = max(pbest - 1, 0), i1 = min(pbest + 1, g.Dp - 1);
                const int16_t* prevS = myS + (par ^ 1) * g.Dp;
                const uint32_t nb = ((uint32_t)(uint16_t)prevS[i0] << 16) | (uint32_t)(uint16_t)prevS[i1];
                const LV<NP> L = path_step<NP>(Cv, Lp, delta2, P1_2, padreg);
                const uint32_t mn = row_min_u32(lane_min16<NP>(L));
                delta2 = pk_add(pk_rep((int)mn), P2_2);
                Lp = L;
#pragma unroll
                for (int q = 0; q < NP; q++) S.r[q] = pk_add_sat(S.r[q], L.r[q]);
                int minS, best;
                bool row_viol;
                wta_core<NP, PAD, true>(S, g, lane, minS, best, row_viol);
                lv_store<NP>(myS + par * g.Dp + d0, S);
                if (writer && y > 0) {
                    aux0[aoff] = prec;
                    aux1[aoff] = (int)nb;
                }
                if (y > 0) aoff += (uint32_t)g.W;
                prec = row_viol ? -1 : ((minS << 8) | best);
                pbest = best;
                par ^= 1;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    {   // the last row's record
        const int i0 = max(pbest - 1, 0), i1 = min(pbest + 1, g.Dp - 1);
        const int16_t* prevS = myS + (par ^ 1) * g.Dp;
        if (writer) {
            aux0[aoff] = prec;
            aux1[aoff] = (int)(((uint32_t)(uint16_t)prevS[i0] << 16) | (uint32_t)(uint16_t)prevS[i1]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// the same fused sweep with 32 lanes per column (two columns per wave): twice the waves -- 1152 columns make 576
// waves instead of 288 on 1024 SIMDs -- and half the packed registers per lane, i.e. half the instructions on each
// wave's serial chain (the counters say this sweep is issue-bound on its lone wave per SIMD: VALU busy 61 %, parked
// 24 %; profiles/r02_occupancy_c2.csv).  The d-1 / d+1 neighbours cross the two 16-lane DPP rows of a column with
// wave_shr / wave_shl (column edges patched to MAX_COST), the 32-lane minimum closes with one v_permlane16_swap.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t grp32_min_u32(uint32_t v)
{
    v = row_min_u32(v);
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // rows (0,1) and (2,3) exchange
    return min(r[0], r[1]);
}

template <int NP>
__device__ __forceinline__ LV<NP> path_step32(const LV<NP>& Cp, const LV<NP>& Lp, uint32_t delta2, uint32_t P1_2, unsigned padreg, int l32)
{
    LV<NP> out;
    uint32_t prev_hi = DPP(MAXC2, Lp.r[NP - 1], 0x138);   // wave_shr:1 -- lane-1's last register
    uint32_t next_lo = DPP(MAXC2, Lp.r[0], 0x130);        // wave_shl:1 -- lane+1's first register
    prev_hi = l32 == 0 ? MAXC2 : prev_hi;                 // the column's first / last lane see the MAX_COST sentinels
    next_lo = l32 == 31 ? MAXC2 : next_lo;
    uint32_t w[NP + 1];
    w[0] = __builtin_amdgcn_alignbit(Lp.r[0], prev_hi, 16);
#pragma unroll
    for (int k = 1; k < NP; k++) w[k] = __builtin_amdgcn_alignbit(Lp.r[k], Lp.r[k - 1], 16);
    w[NP] = __builtin_amdgcn_alignbit(next_lo, Lp.r[NP - 1], 16);
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const uint32_t nb = pk_add_sat(pk_min(w[k], w[k + 1]), P1_2);
        const uint32_t m = pk_min(pk_min(Lp.r[k], delta2), nb);
        const uint32_t L = pk_sub(pk_add(Cp.r[k], m), delta2);
        out.r[k] = ((padreg >> k) & 1u) ? MAXC2 : L;
    }
    return out;
}

// wta_core for a 32-lane column group (threshold form of the uniqueness test only)
template <int NP, bool PAD>
__device__ __forceinline__ void wta_core32(const LV<NP>& S, const SgbmGeom& g, int lane, int& minS, int& best, bool& grp_viol)
{
    const int l32 = lane & 31, d0 = l32 * 2 * NP;
    uint32_t key = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const int d = d0 + 2 * k;
        if (!PAD || d < g.D) {
            const uint32_t k0 = ((S.r[k] & 0xFFFFu) << 8) | (uint32_t)d;
            const uint32_t k1 = ((S.r[k] >> 16) << 8) | (uint32_t)(d + 1);
            key = min(key, min(k0, k1));
        }
    }
    key = grp32_min_u32(key);
    minS = (int)(key >> 8);
    best = (int)(key & 255u);
    const int ur100 = 100 - g.ur;
    const int a = minS * 100;
    int T = (int)((float)a * __builtin_amdgcn_rcpf((float)ur100));
    T += (T * ur100 < a);
    T += (T * ur100 < a);
    T -= ((T - 1) * ur100 >= a);
    T -= ((T - 1) * ur100 >= a);
    const uint32_t T2 = pk_rep(min(T, 32768));
    const int t = min(max(best + 4 - d0, 0), 31);
    uint32_t m8 = ((7u << t) >> 5) & ((1u << (2 * NP)) - 1u);
    if (PAD) {
        const int npad = min(max(d0 + 2 * NP - g.D, 0), 2 * NP);
        m8 |= (0xFFFFu << (2 * NP - npad)) & ((1u << (2 * NP)) - 1u);
    }
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)m8, 2 * k, 1);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)m8, 2 * k + 1, 1);
        const uint32_t excl = (lo & 0xFFFFu) | (hi & 0xFFFF0000u);
        any |= pk_sub_sat_u(T2, S.r[k]) & ~excl;
    }
    const unsigned long long bal = __ballot(any != 0);
    grp_viol = ((bal >> (lane & 32)) & 0xFFFFFFFFull) != 0ull;
}

template <int NP, int NV, bool PAD>
__global__ void __launch_bounds__(256) k_sgbm_vwta32(const int16_t* __restrict__ C, const int16_t* __restrict__ Lbase, size_t vol,
                                                    SgbmGeom g, int* __restrict__ aux0, int* __restrict__ aux1)
{
    extern __shared__ int16_t s_S[];  // [blockDim/32][2][Dp]
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int x1 = wave * 2 + half;
    if (wave * 2 >= g.W1) return;
    const bool live = x1 < g.W1;
    const int xc = live ? x1 : g.W1 - 1;
    const int d0 = l32 * 2 * NP;
    unsigned padreg = 0;
    if (PAD) {
#pragma unroll
        for (int k = 0; k < NP; k++) padreg |= (unsigned)(d0 + 2 * k >= g.D) << k;
    }
    const uint32_t P1_2 = pk_rep(g.P1), P2_2 = pk_rep(g.P2);
    const size_t stride = (size_t)g.W1 * g.Dp;
    const uint32_t start = (uint32_t)(xc * g.Dp + d0);
    int16_t* myS = s_S + (size_t)(threadIdx.x >> 5) * 2 * g.Dp;
    LV<NP> Lp;
#pragma unroll
    for (int k = 0; k < NP; k++) Lp.r[k] = ((padreg >> k) & 1u) ? MAXC2 : 0u;
    uint32_t delta2 = P2_2;
    const bool writer = live && l32 == 0;
    uint32_t aoff = (uint32_t)(x1 + g.minX1);
    int prec = -1, pbest = 0, par = 0;
    constexpr int RPS = NP * (NV + 1);
    constexpr int PF = RPS <= 10 ? 8 : RPS <= 20 ? 6 : RPS <= 32 ? 4 : 3;
    LV<NP> cbuf[PF], lbuf[PF][NV];
    const int16_t* rowC = C;
    const int16_t* rowL = Lbase;
#pragma unroll
    for (int k = 0; k < PF; k++) {
        if (k < g.H) {
            cbuf[k] = lv_load_nt<NP>(rowC + start);
#pragma unroll
            for (int v = 0; v < NV; v++) lbuf[k][v] = lv_load_nt<NP>(rowL + (size_t)v * vol + start);
            rowC += stride;
            rowL += stride;
        }
    }
    for (int y0 = 0; y0 < g.H; y0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int y = y0 + k;
            if (y < g.H) {
                const LV<NP> Cv = cbuf[k];
                LV<NP> S = lbuf[k][0];
#pragma unroll
                for (int v = 1; v < NV; v++)
#pragma unroll
                    for (int q = 0; q < NP; q++) S.r[q] = pk_add_sat(S.r[q], lbuf[k][v].r[q]);
                if (y + PF < g.H) {
                    cbuf[k] = lv_load_nt<NP>(rowC + start);
#pragma unroll
                    for (int v = 0; v < NV; v++) lbuf[k][v] = lv_load_nt<NP>(rowL + (size_t)v * vol + start);
                    rowC += stride;
                    rowL += stride;
                }
                const int i0 = max(pbest - 1, 0), i1 = min(pbest + 1, g.Dp - 1);
                const int16_t* prevS = myS + (par ^ 1) * g.Dp;
                const uint32_t nb = ((uint32_t)(uint16_t)prevS[i0] << 16) | (uint32_t)(uint16_t)prevS[i1];
                const LV<NP> L = path_step32<NP>(Cv, Lp, delta2, P1_2, padreg, l32);
                const uint32_t mn = grp32_min_u32(lane_min16<NP>(L));
                delta2 = pk_add(pk_rep((int)mn), P2_2);
                Lp = L;
#pragma unroll
                for (int q = 0; q < NP; q++) S.r[q] = pk_add_sat(S.r[q], L.r[q]);
                int minS, best;
                bool viol;
                wta_core32<NP, PAD>(S, g, lane, minS, best, viol);
                lv_store<NP>(myS + par * g.Dp + d0, S);
                if (writer && y > 0) {
                    aux0[aoff] = prec;
                    aux1[aoff] = (int)nb;
                }
                if (y > 0) aoff += (uint32_t)g.W;
                prec = viol ? -1 : ((minS << 8) | best);
                pbest = best;
                par ^= 1;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    {
        const int i0 = max(pbest - 1, 0), i1 = min(pbest + 1, g.Dp - 1);
        const int16_t* prevS = myS + (par ^ 1) * g.Dp;
        if (writer) {
            aux0[aoff] = prec;
            aux1[aoff] = (int)(((uint32_t)(uint16_t)prevS[i0] << 16) | (uint32_t)(uint16_t)prevS[i1]);
        }
    }
}

// ---- the same fused sweep with 64 lanes per column (one column per wave): Dp / 128 registers per lane -------------
// 1152 waves at C2 instead of 576 (more than one per SIMD), each step shorter; only for Dp = 128 or 256.
__device__ __forceinline__ uint32_t wave64_min_u32(uint32_t v)
{
    v = row_min_u32(v);
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // rows (0,1) and (2,3) exchange
    v = min(r[0], r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // halves exchange
    return min(q[0], q[1]);
}

template <int NP>
__device__ __forceinline__ LV<NP> path_step64(const LV<NP>& Cp, const LV<NP>& Lp, uint32_t delta2, uint32_t P1_2, int lane)
{
    LV<NP> out;
    uint32_t prev_hi = DPP(MAXC2, Lp.r[NP - 1], 0x138);   // wave_shr:1
    uint32_t next_lo = DPP(MAXC2, Lp.r[0], 0x130);        // wave_shl:1
    prev_hi = lane == 0 ? MAXC2 : prev_hi;
    next_lo = lane == 63 ? MAXC2 : next_lo;
    uint32_t w[NP + 1];
    w[0] = __builtin_amdgcn_alignbit(Lp.r[0], prev_hi, 16);
#pragma unroll
    for (int k = 1; k < NP; k++) w[k] = __builtin_amdgcn_alignbit(Lp.r[k], Lp.r[k - 1], 16);
    w[NP] = __builtin_amdgcn_alignbit(next_lo, Lp.r[NP - 1], 16);
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const uint32_t nb = pk_add_sat(pk_min(w[k], w[k + 1]), P1_2);
        const uint32_t m = pk_min(pk_min(Lp.r[k], delta2), nb);
        out.r[k] = pk_sub(pk_add(Cp.r[k], m), delta2);
    }
    return out;
}

// winner + uniqueness verdict of one pixel held by the whole wave (threshold form; no padded disparities: D = Dp)
template <int NP>
__device__ __forceinline__ void wta_core64(const LV<NP>& S, const SgbmGeom& g, int lane, int& minS, int& best, bool& viol)
{
    const int d0 = lane * 2 * NP;
    uint32_t key = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const int d = d0 + 2 * k;
        const uint32_t k0 = ((S.r[k] & 0xFFFFu) << 8) | (uint32_t)d;
        const uint32_t k1 = ((S.r[k] >> 16) << 8) | (uint32_t)(d + 1);
        key = min(key, min(k0, k1));
    }
    key = wave64_min_u32(key);
    minS = (int)(key >> 8);
    best = (int)(key & 255u);
    const int ur100 = 100 - g.ur;
    const int a = minS * 100;
    int T = (int)((float)a * __builtin_amdgcn_rcpf((float)ur100));
    T += (T * ur100 < a);
    T += (T * ur100 < a);
    T -= ((T - 1) * ur100 >= a);
    T -= ((T - 1) * ur100 >= a);
    const uint32_t T2 = pk_rep(min(T, 32768));
    const int t = min(max(best + 4 - d0, 0), 31);
    const uint32_t m8 = ((7u << t) >> 5) & ((1u << (2 * NP)) - 1u);       // bits of this lane's disparities within best +- 1
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe((int)m8, 2 * k, 1);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)m8, 2 * k + 1, 1);
        const uint32_t excl = (lo & 0xFFFFu) | (hi & 0xFFFF0000u);
        any |= pk_sub_sat_u(T2, S.r[k]) & ~excl;
    }
    viol = __ballot(any != 0) != 0ull;
}

template <int NP, int NV>
__global__ void __launch_bounds__(256) k_sgbm_vwta64(const int16_t* __restrict__ C, const int16_t* __restrict__ Lbase, size_t vol,
                                                    SgbmGeom g, int* __restrict__ aux0, int* __restrict__ aux1)
{
    extern __shared__ int16_t s_S[];  // [blockDim/64][2][Dp]
    const int lane = threadIdx.x & 63;
    const int x1 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (x1 >= g.W1) return;
    const int d0 = lane * 2 * NP;
    const uint32_t P1_2 = pk_rep(g.P1), P2_2 = pk_rep(g.P2);
    const size_t stride = (size_t)g.W1 * g.Dp;
    const uint32_t start = (uint32_t)(x1 * g.Dp + d0);
    int16_t* myS = s_S + (size_t)(threadIdx.x >> 6) * 2 * g.Dp;
    LV<NP> Lp;
#pragma unroll
    for (int k = 0; k < NP; k++) Lp.r[k] = 0u;
    uint32_t delta2 = P2_2;
    const bool writer = lane == 0;
    uint32_t aoff = (uint32_t)(x1 + g.minX1);
    int prec = -1, pbest = 0, par = 0;
    constexpr int PF = 8;
    LV<NP> cbuf[PF], lbuf[PF][NV];
    const int16_t* rowC = C;
    const int16_t* rowL = Lbase;
#pragma unroll
    for (int k = 0; k < PF; k++) {
        if (k < g.H) {
            cbuf[k] = lv_load_nt<NP>(rowC + start);
#pragma unroll
            for (int v = 0; v < NV; v++) lbuf[k][v] = lv_load_nt<NP>(rowL + (size_t)v * vol + start);
            rowC += stride;
            rowL += stride;
        }
    }
    for (int y0 = 0; y0 < g.H; y0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int y = y0 + k;
            if (y < g.H) {
                const LV<NP> Cv = cbuf[k];
                LV<NP> S = lbuf[k][0];
#pragma unroll
                for (int v = 1; v < NV; v++)
#pragma unroll
                    for (int q = 0; q < NP; q++) S.r[q] = pk_add_sat(S.r[q], lbuf[k][v].r[q]);
                if (y + PF < g.H) {
                    cbuf[k] = lv_load_nt<NP>(rowC + start);
#pragma unroll
                    for (int v = 0; v < NV; v++) lbuf[k][v] = lv_load_nt<NP>(rowL + (size_t)v * vol + start);
                    rowC += stride;
                    rowL += stride;
                }
                const int i0 = max(pbest - 1, 0), i1 = min(pbest + 1, g.Dp - 1);
                const int16_t* prevS = myS + (par ^ 1) * g.Dp;
                const uint32_t nb = ((uint32_t)(uint16_t)prevS[i0] << 16) | (uint32_t)(uint16_t)prevS[i1];
                const LV<NP> L = path_step64<NP>(Cv, Lp, delta2, P1_2, lane);
                const uint32_t mn = wave64_min_u32(lane_min16<NP>(L));
                delta2 = pk_add(pk_rep((int)mn), P2_2);
                Lp = L;
#pragma unroll
                for (int q = 0; q < NP; q++) S.r[q] = pk_add_sat(S.r[q], L.r[q]);
                int minS, best;
                bool viol;
                wta_core64<NP>(S, g, lane, minS, best, viol);
                lv_store<NP>(myS + par * g.Dp + d0, S);
                if (writer && y > 0) {
                    aux0[aoff] = prec;
                    aux1[aoff] = (int)nb;
                }
                if (y > 0) aoff += (uint32_t)g.W;
                prec = viol ? -1 : ((minS << 8) | best);
                pbest = best;
                par ^= 1;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    {
        const int i0 = max(pbest - 1, 0), i1 = min(pbest + 1, g.Dp - 1);
        const int16_t* prevS = myS + (par ^ 1) * g.Dp;
        if (writer) {
            aux0[aoff] = prec;
            aux1[aoff] = (int)(((uint32_t)(uint16_t)prevS[i0] << 16) | (uint32_t)(uint16_t)prevS[i1]);
        }
    }
}

#include "sgbm_raster.inc"
#include "sgbm_diag.inc"

// left-right check of one pixel on the WTA results: disp1 or INVALID
__device__ __forceinline__ int lr_value(const int16_t* __restrict__ disp1, const int* __restrict__ d2key, const SgbmGeom& g, int x, int y)
{
    const size_t rowo = (size_t)y * g.W;
    int d1 = g.invalid16;
    if (x >= g.minX1 && x < g.minX1 + g.W1) {
        d1 = disp1[rowo + x];
        if (d1 != g.invalid16) {
            auto disp2 = [&](int xx) -> int {
                const int k = d2key[rowo + xx];
                return k == D2_EMPTY ? g.invalid16 : (0xFFFF - (k & 0xFFFF)) - xx;  // ximg - x2 = d + minD
            };
            const int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
            const int _x = x - _d, x_ = x - d_;
            bool bad = false;
            if (0 <= _x && _x < g.W && 0 <= x_ && x_ < g.W) {
                const int a = disp2(_x), b = disp2(x_);
                bad = a >= g.minD && abs(a - _d) > g.d12 && b >= g.minD && abs(b - d_) > g.d12;
            }
            if (bad) d1 = g.invalid16;
        }
    }
    return d1;
}

// ---------------------------------------------------------------------------------------
// post filters
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void cswap(int& a, int& b) { int t = min(a, b); b = max(a, b); a = t; }

// left-right check + medianBlur(3) in one pass: the 9 checked values are evaluated in place (a few cached
// loads each) instead of being written out by a launch of their own
__global__ void k_lr_median3(const int16_t* __restrict__ disp1, const int* __restrict__ d2key, SgbmGeom g, int16_t* __restrict__ dst)
{
    const int W = g.W, H = g.H;
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int y0 = max(y - 1, 0), y2 = min(y + 1, H - 1), xl = max(x - 1, 0), xr = min(x + 1, W - 1);
    int p0 = lr_value(disp1, d2key, g, xl, y0), p1 = lr_value(disp1, d2key, g, x, y0), p2 = lr_value(disp1, d2key, g, xr, y0);
    int p3 = lr_value(disp1, d2key, g, xl, y), p4 = lr_value(disp1, d2key, g, x, y), p5 = lr_value(disp1, d2key, g, xr, y);
    int p6 = lr_value(disp1, d2key, g, xl, y2), p7 = lr_value(disp1, d2key, g, x, y2), p8 = lr_value(disp1, d2key, g, xr, y2);
    cswap(p1, p2); cswap(p4, p5); cswap(p7, p8); cswap(p0, p1);
    cswap(p3, p4); cswap(p6, p7); cswap(p1, p2); cswap(p4, p5);
    cswap(p7, p8); cswap(p0, p3); cswap(p5, p8); cswap(p4, p7);
    cswap(p3, p6); cswap(p1, p4); cswap(p2, p5); cswap(p4, p7);
    cswap(p4, p2); cswap(p6, p4); cswap(p4, p2);
    dst[(size_t)y * W + x] = (int16_t)p4;
}

// filterSpeckles as connected-component labelling.  Edges join 4-neighbours that are both
// != newVal and differ by <= maxDiff.  Horizontal runs are labelled by their first pixel in one
// pass per row; only run heads take part in the union-find, sizes are added once per run.
__device__ __forceinline__ int uf_find(const int* L, int i)
{
    int p = L[i];
    while (p != i) { i = p; p = L[i]; }
    return i;
}
// find with path halving: every write stores an ancestor (a smaller index), so concurrent
// halving and atomicMin unions cannot create cycles or disconnect a set
__device__ __forceinline__ int uf_find_halve(int* L, int i)
{
    for (;;) {
        const int p = ((volatile int*)L)[i];
        if (p == i || p < 0) return i;      // p < 0 cannot happen for a labelled pixel; never index with it
        const int gp = ((volatile int*)L)[p];
        if (gp == p || gp < 0) return p;
        L[i] = gp;
        i = gp;
    }
}
__device__ __forceinline__ void uf_union(int* L, int a, int b)
{
    for (int it = 0; it < (1 << 24); it++) {
        a = uf_find_halve(L, a);
        b = uf_find_halve(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

// one block per image row: label[i] = index of the head of i's horizontal run (or -1), runlen[head],
// size[head] = 0
__global__ void __launch_bounds__(256) k_ccl_rows(const int16_t* __restrict__ img, int W, int newVal, int maxDiff,
                                                 int* __restrict__ L, int* __restrict__ runlen, int* __restrict__ size)
{
    __shared__ int s_tot[256];
    const int y = blockIdx.x, tid = threadIdx.x;
    const int per = (W + 255) / 256;
    const int xa = tid * per, xb = min(xa + per, W);
    const int16_t* r = img + (size_t)y * W;
    // last run start at or before each pixel inside this thread's chunk (-1: none yet)
    int last = -1;
    for (int x = xa; x < xb; x++) {
        const int v = r[x];
        const bool valid = v != newVal;
        const bool start = valid && (x == 0 || r[x - 1] == newVal || abs(v - r[x - 1]) > maxDiff);
        if (start) last = x;
    }
    s_tot[tid] = last;
    __syncthreads();
    // inclusive prefix max over the threads' chunk results
    for (int o = 1; o < 256; o <<= 1) {
        int v = tid >= o ? s_tot[tid - o] : -1;
        __syncthreads();
        s_tot[tid] = max(s_tot[tid], v);
        __syncthreads();
    }
    int cur = tid > 0 ? s_tot[tid - 1] : -1;
    for (int x = xa; x < xb; x++) {
        const int v = r[x];
        const size_t i = (size_t)y * W + x;
        if (v == newVal) { L[i] = -1; continue; }
        const bool start = x == 0 || r[x - 1] == newVal || abs(v - r[x - 1]) > maxDiff;
        if (start) { cur = x; size[i] = 0; }
        L[i] = y * W + cur;
        const bool end = x == W - 1 || r[x + 1] == newVal || abs(v - r[x + 1]) > maxDiff;
        if (end) runlen[(size_t)y * W + cur] = x - cur + 1;
    }
}

// Only "is the component larger than maxSize" is ever asked, so runs that are longer than maxSize by
// themselves ("big") never enter the union-find: big-big contacts are ignored (both survive anyway) and a
// small run touching a big one is only flagged (RUN_TOUCH in its runlen word; k_ccl_sizes then credits its
// component with maxSize + 1).  The wide regions of a disparity map -- whose unions all fought over the same
// few roots -- drop out; the union-find is left with the small runs.
#define RUN_TOUCH (1 << 30)
__global__ void k_ccl_vmerge(const int16_t* __restrict__ img, int W, int H, int newVal, int maxDiff, int maxSize, int* __restrict__ L,
                             int* __restrict__ runlen)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y + 1 >= H) return;
    const int i = y * W + x;
    const int v = img[i], u = img[i + W];
    if (v == newVal || u == newVal || abs(v - u) > maxDiff) return;
    const int a = L[i], b = L[i + W];
    bool head_a = x == 0, head_b = x == 0, joined_left = false;
    if (x > 0) {
        const int v1 = img[i - 1], u1 = img[i + W - 1];
        head_a = v1 == newVal || abs(v - v1) > maxDiff;
        head_b = u1 == newVal || abs(u - u1) > maxDiff;
        joined_left = v1 != newVal && u1 != newVal && abs(v1 - u1) <= maxDiff && !head_a && !head_b;   // same two runs, one column earlier
    }
    if (joined_left) return;
    // a head pixel's label may already point at an ancestor: its own index is the run head
    const int ha = head_a ? i : a, hb = head_b ? i + W : b;
    const int la = ((volatile int*)runlen)[ha], lb = ((volatile int*)runlen)[hb];
    const bool big_a = (la & ~RUN_TOUCH) > maxSize, big_b = (lb & ~RUN_TOUCH) > maxSize;
    if (big_a && big_b) return;
    if (big_a != big_b) {
        const int hs = big_a ? hb : ha, ls = big_a ? lb : la;
        if (!(ls & RUN_TOUCH)) atomicOr(&runlen[hs], RUN_TOUCH);
        return;
    }
    uf_union(L, a, b);
}

// component sizes: every run head (recognised geometrically -- after unions a head's label may
// point elsewhere) adds its run length to the component root
__global__ void k_ccl_sizes(const int16_t* __restrict__ img, int W, int H, int newVal, int maxDiff, int maxSize,
                            int* __restrict__ L, const int* __restrict__ runlen, int* __restrict__ size)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int i = y * W + x;
    const int v = img[i];
    if (v == newVal) return;
    const bool start = x == 0 || img[i - 1] == newVal || abs(v - img[i - 1]) > maxDiff;
    if (!start) return;
    const int root = uf_find_halve(L, i);
    L[i] = root;   // flatten: k_ccl_apply then needs two hops (pixel -> run head -> root)
    // only "size <= maxSize" is ever asked, and the counter only grows: once it is past the limit
    // further adds are pointless (this removes the contention on the few huge components)
    if (((volatile int*)size)[root] > maxSize) return;
    const int len = runlen[i];
    atomicAdd(&size[root], (len & ~RUN_TOUCH) + ((len & RUN_TOUCH) ? maxSize + 1 : 0));
}

__global__ void k_ccl_apply(int16_t* __restrict__ img, int n, int newVal, int maxSize, const int* __restrict__ L,
                            const int* __restrict__ size)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = L[i];
    if (r >= 0 && size[uf_find(L, r)] <= maxSize) img[i] = (int16_t)newVal;
}

// ---------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------
static PathPlan make_plan(const SgbmGeom& g, int mode, int lpw)
{
    // steps = negated predecessor offsets.  MODE_SGBM predecessors: (x-1,y) (x+1,y) (x-1,y-1) (x,y-1)
    // (x+1,y-1); MODE_HH adds (x-1,y+1) (x,y+1) (x+1,y+1).  Longest lines first.
    static const int sx5[] = { 1, -1, 1, -1, 0 }, sy5[] = { 0, 0, 1, 1, 1 };
    static const int sx8[] = { 1, -1, 1, -1, 1, -1, 0, 0 }, sy8[] = { 0, 0, 1, 1, -1, -1, -1, 1 };   // (0,+1) last: fused with the WTA
    PathPlan p;
    p.lpw = lpw;
    p.n_dirs = mode == 1 ? 8 : 5;
    p.first_wave[0] = 0;
    for (int k = 0; k < p.n_dirs; k++) {
        p.sx[k] = mode == 1 ? sx8[k] : sx5[k];
        p.sy[k] = mode == 1 ? sy8[k] : sy5[k];
        p.nlines[k] = p.sy[k] == 0 ? g.H : (p.sx[k] == 0 ? g.W1 : g.W1 + g.H - 1);
        p.first_wave[k + 1] = p.first_wave[k] + div_up(p.nlines[k], lpw);
    }
    for (int k = p.n_dirs; k < VO_MAX_DIRS; k++) { p.sx[k] = p.sy[k] = p.nlines[k] = 0; p.first_wave[k + 1] = p.first_wave[p.n_dirs]; }
    return p;
}

// rows per band (= compute waves per workgroup) of the raster sweep: two waves per SIMD while the ring fits in LDS
static inline int raster_rows(int NP)
{
    if (const char* e = getenv("VO_RASTER_ROWS")) { int v = atoi(e); if (v >= 1 && v <= 8) return v; }
    return NP <= 4 ? 8 : 4;
}

template <int NP, bool PAD, bool REV, bool HASIN, bool WTA>
static int launch_raster(vo_ctx* ctx, const SgbmGeom& g, const int16_t* Lin, int16_t* Sout, int* ctl)
{
    const int R = raster_rows(NP), nbands = div_up(g.H, R);
    const size_t lds = (size_t)(R + 1) * RS_RING * NP * 256 + (size_t)(R + 1) * RS_RING * 16 + (size_t)32 * 4 +
                       (size_t)R * 2 * 4 * g.Dp * 2;
    if (!ctx->rs_bnd) VO_HIP(ctx, hipMalloc((void**)&ctx->rs_bnd, (ctx->vol_cells * 3 / 8 + 4096) * 8 + 256));   // first use in this workspace
    auto kern = k_sgbm_raster<NP, PAD, REV, HASIN, WTA>;
    static unsigned long long attr_set = 0;     // per instantiation and device: allow more than 64 KB of dynamic LDS
    if (!((attr_set >> (ctx->device & 63)) & 1ull)) {
        VO_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set |= 1ull << (ctx->device & 63);
    }
    hipLaunchKernelGGL(kern, dim3(nbands < ctx->tune_raster_wgs ? nbands : ctx->tune_raster_wgs), dim3((R + 2) * 64), lds, ctx->stream, ctx->C, Lin, Sout, ctx->rs_bnd, ctl, g, R, nbands,
                       ctx->ccl_label, ctx->ccl_runlen, ctx->dump);
    VO_CHECK_LAUNCH(ctx);
    return VO_OK;
}


// ---- diagonal sweep (sgbm_diag.inc) ------------------------------------------------------------------------------------
template <int NP, bool PAD, int NWC, bool REV, bool WTA>
static int launch_diag_k(vo_ctx* ctx, const SgbmGeom& g, const int16_t* in1, int16_t* sout, int* ctl)
{
    constexpr int UW = 4 * NWC, CW = UW + 2;
    DiagJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    jobs.n = 1;
    jobs.nstrips = div_up(g.W1 + g.H - 1, UW);
    const size_t need = (size_t)jobs.nstrips * (g.H + 1) * 64 * NP * sizeof(uint64_t);
    const size_t have = (ctx->vol_cells / 4 + 4096) * sizeof(uint64_t);
    if (!ctx->rs_bnd || need > have) return vo_fail(ctx, VO_E_CAP, "diagonal sweep: boundary buffer too small (%zu > %zu bytes)", need, have);
    if (ctx->sw_tag == 0) VO_HIP(ctx, hipMemsetAsync(ctx->rs_bnd, 0, have, ctx->stream));   // tags start at 1: no stale granule may match
    DiagJob& j = jobs.j[0];
    j.C = ctx->C; j.in1 = in1; j.sout = sout; j.bnd = ctx->rs_bnd; j.aux0 = ctx->ccl_label; j.aux1 = ctx->ccl_runlen;
    j.tag = ++ctx->sw_tag;
    j.dbg = ctx->tune_diag_dbg;
    if (j.tag == 0) j.tag = ++ctx->sw_tag;
    const size_t lds = (size_t)2 * 2 * CW * g.Dp * 2 + (WTA ? (size_t)NWC * 4 * 2 * g.Dp * 2 : 0) + 64;
    auto kern = k_sgbm_diag<NP, PAD, NWC, REV, WTA>;
    static unsigned long long attr_set = 0;     // per instantiation and device: allow more than 64 KB of dynamic LDS
    if (!((attr_set >> (ctx->device & 63)) & 1ull)) {
        VO_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set |= 1ull << (ctx->device & 63);
    }
    // strips that can be active at the same time: one image row's worth (+ slack for the hand-over between strips)
    const int wgs = min(jobs.nstrips, div_up(g.W1, UW) + 2);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3((NWC + 1) * 64), lds, ctx->stream, jobs, g, ctl, ctx->dump, ctx->rs_ctl_words / 2);
    VO_CHECK_LAUNCH(ctx);
    return VO_OK;
}

template <int NP, bool REV, bool WTA>
static int launch_diag(vo_ctx* ctx, const SgbmGeom& g, const int16_t* in1, int16_t* sout, int* ctl)
{
    const bool pad = g.D != g.Dp;
    if constexpr (NP <= 4) {
        if (ctx->tune_diag_nwc == 15)
            return pad ? launch_diag_k<NP, true, 15, REV, WTA>(ctx, g, in1, sout, ctl) : launch_diag_k<NP, false, 15, REV, WTA>(ctx, g, in1, sout, ctl);
    }
    return pad ? launch_diag_k<NP, true, 7, REV, WTA>(ctx, g, in1, sout, ctl) : launch_diag_k<NP, false, 7, REV, WTA>(ctx, g, in1, sout, ctl);
}

template <int NP>
static int launch_agg(vo_ctx* ctx, const SgbmGeom& g, const PathPlan& plan_all, size_t vol)
{
    // the last direction of the plan is the top-down vertical one: it runs fused with the WTA
    PathPlan plan = plan_all;
    const bool fuse = ctx->tune_fuse_wta != 0 && g.ur < 100;   // the fused sweep only carries the threshold form of the uniqueness test
    // Diagonal-sweep schedule: W + E as one stored volume (k_sgbm_we), then NW / N / NE + WTA in ONE pass over C and that
    // volume (k_sgbm_diag); MODE_HH: a reverse pass first adds SW / S / SE to the W + E volume.  Layout of S: [0] = L_W + L_E,
    // [1] = MODE_HH: [0] + the three bottom-up directions, last = the E checkpoints of k_sgbm_we (1/8 of a volume).
    if (fuse && ctx->tune_diag && plan.lpw == 4 && g.W1 % 8 == 0 && g.W1 >= 16) {
        const bool pad = g.D != g.Dp;
        const bool hh = plan_all.n_dirs == 8;
        int rc;
        {
            StageTimer t(ctx, VO_T_SGBM_AGG);
            const int nw = div_up(g.H, 4);
            int16_t* ck = ctx->S + (size_t)(hh ? 2 : 1) * vol;
            if (ctx->tune_diag_dbg & 8) {
            } else if (pad) hipLaunchKernelGGL((k_sgbm_we<NP, true>), dim3(div_up(nw, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, ck, g, ctx->dump);
            else hipLaunchKernelGGL((k_sgbm_we<NP, false>), dim3(div_up(nw, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, ck, g, ctx->dump);
            VO_CHECK_LAUNCH(ctx);
            if (hh && (rc = launch_diag<NP, true, false>(ctx, g, ctx->S, ctx->S + vol, ctx->rs_ctl))) return rc;
        }
        {
            StageTimer t(ctx, VO_T_SGBM_WTA);
            if (!(ctx->tune_diag_dbg & 16) && (rc = launch_diag<NP, false, true>(ctx, g, hh ? ctx->S + vol : ctx->S, nullptr, ctx->rs_ctl + (hh ? ctx->rs_ctl_words / 2 : 0)))) return rc;
            hipLaunchKernelGGL(k_sgbm_fin, dim3(div_up(g.W1, 256), g.H), dim3(256), 0, ctx->stream, ctx->ccl_label, ctx->ccl_runlen, g, ctx->disp_tmp, ctx->ccl_size);
            VO_CHECK_LAUNCH(ctx);
        }
        ctx->last_paths = 1;
        return VO_OK;
    }
    if (fuse && ctx->tune_raster) {
        // raster scheme: MODE_SGBM = E by the line kernel, then W/NW/N/NE + WTA in one raster pass;
        // MODE_HH = E/SE/S/SW in a reverse raster pass (sum stored), then W/NW/N/NE + WTA in the forward pass
        int* ctlA = ctx->rs_ctl;
        int* ctlB = ctx->rs_ctl + ctx->rs_ctl_words / 2;
        int rc;
        {
            StageTimer t(ctx, VO_T_SGBM_AGG);
            if (plan_all.n_dirs == 5) {
                PathPlan pe = plan_all;
                pe.n_dirs = 1;
                pe.sx[0] = -1; pe.sy[0] = 0; pe.nlines[0] = g.H;
                pe.first_wave[0] = 0;
                for (int k = 0; k < VO_MAX_DIRS; k++) pe.first_wave[k + 1] = div_up(g.H, 4);
                const int nwaves = pe.first_wave[1];
                if (NP <= 4)
                    hipLaunchKernelGGL((k_sgbm_paths<NP, 8>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, vol, g, pe, ctx->dump);
                else
                    hipLaunchKernelGGL((k_sgbm_paths<NP, 4>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, vol, g, pe, ctx->dump);
                VO_CHECK_LAUNCH(ctx);
            } else {
                if (g.D == g.Dp) rc = launch_raster<NP, false, true, false, false>(ctx, g, nullptr, ctx->S, ctlA);
                else rc = launch_raster<NP, true, true, false, false>(ctx, g, nullptr, ctx->S, ctlA);
                if (rc) return rc;
            }
        }
        {
            StageTimer t(ctx, VO_T_SGBM_WTA);
            if (g.D == g.Dp) rc = launch_raster<NP, false, false, true, true>(ctx, g, ctx->S, nullptr, ctlB);
            else rc = launch_raster<NP, true, false, true, true>(ctx, g, ctx->S, nullptr, ctlB);
            if (rc) return rc;
            hipLaunchKernelGGL(k_sgbm_fin, dim3(div_up(g.W1, 256), g.H), dim3(256), 0, ctx->stream, ctx->ccl_label, ctx->ccl_runlen, g, ctx->disp_tmp, ctx->ccl_size);
            VO_CHECK_LAUNCH(ctx);
        }
        ctx->last_paths = plan_all.n_dirs == 5 ? 1 : 0;
        return VO_OK;
    }
    if (plan_all.n_dirs > ctx->S_vols) {
        // the line-per-direction scheme stores one volume per direction; grow once
        if (ctx->S) (void)hipFree(ctx->S);
        ctx->S = nullptr; ctx->S_vols = 0;
        VO_HIP(ctx, hipMalloc((void**)&ctx->S, ctx->vol_cells * sizeof(int16_t) * plan_all.n_dirs + 256));
        ctx->S_vols = plan_all.n_dirs;
    }
    if (fuse) {
        plan.n_dirs = plan_all.n_dirs - 1;
        for (int k = plan.n_dirs; k < VO_MAX_DIRS; k++) plan.first_wave[k + 1] = plan.first_wave[plan.n_dirs];
    }
    // VO_WE_FUSE: W and E as one stored volume (k_sgbm_we), then the other stored directions by the line kernel (NW, NE;
    // MODE_HH: also the three bottom-up ones), then the fused sweep over ND + 1 volumes.
    // Layout of S: [0] = L_W + L_E, [1 .. ND] = the other directions, [ND + 1] = the E checkpoints (1/8 of a volume).
    // Band schedule (MODE_SGBM, VO_BAND): W + E as one volume, row checkpoints of N / NW / NE, then the band kernel.
    // Layout of S: [0] = L_W + L_E, [1] = row checkpoints (3/8 of a volume), [3] = the E checkpoints of k_sgbm_we.
    if constexpr (NP <= 4) {
        if (fuse && ctx->band_now && plan.lpw == 4 && plan_all.n_dirs == 5 && g.W1 % 8 == 0 && g.W1 >= 16 && g.H >= 8) {
            const bool pad = g.D != g.Dp;
            const int nb = div_up(g.H, 8);
            constexpr int TW = 32;
            const size_t lds = ((size_t)2 * 3 * (TW + 16) + 16) * g.Dp * sizeof(int16_t);
            static unsigned long long attr_done = 0;   // per device
            if (!((attr_done >> (ctx->device & 63)) & 1ull)) {
                VO_HIP(ctx, hipFuncSetAttribute((const void*)k_sgbm_band<NP, true, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                VO_HIP(ctx, hipFuncSetAttribute((const void*)k_sgbm_band<NP, false, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_done |= 1ull << (ctx->device & 63);
            }
            {
                StageTimer t(ctx, VO_T_SGBM_AGG);
                const int nw = div_up(g.H, 4);
                if (pad) hipLaunchKernelGGL((k_sgbm_we<NP, true>), dim3(div_up(nw, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, ctx->S + 3 * vol, g, ctx->dump);
                else hipLaunchKernelGGL((k_sgbm_we<NP, false>), dim3(div_up(nw, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, ctx->S + 3 * vol, g, ctx->dump);
                PathPlan pd = plan_all;
                const int src[3] = { 4, 2, 3 };                       // N, NW, NE: the order k_sgbm_band indexes the checkpoints by
                pd.n_dirs = 3;
                pd.first_wave[0] = 0;
                for (int k = 0; k < 3; k++) {
                    pd.sx[k] = plan_all.sx[src[k]]; pd.sy[k] = plan_all.sy[src[k]]; pd.nlines[k] = plan_all.nlines[src[k]];
                    pd.first_wave[k + 1] = pd.first_wave[k] + div_up(pd.nlines[k], 4);
                }
                for (int k = 3; k < VO_MAX_DIRS; k++) { pd.sx[k] = pd.sy[k] = pd.nlines[k] = 0; pd.first_wave[k + 1] = pd.first_wave[3]; }
                const int nwaves = pd.first_wave[3];
                if (pad) hipLaunchKernelGGL((k_sgbm_paths<NP, 8, 16, true, 2>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S + vol, (size_t)nb, g, pd, ctx->dump);
                else hipLaunchKernelGGL((k_sgbm_paths<NP, 8, 16, false, 2>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S + vol, (size_t)nb, g, pd, ctx->dump);
                VO_CHECK_LAUNCH(ctx);
            }
            {
                StageTimer t(ctx, VO_T_SGBM_WTA);
                if (pad) hipLaunchKernelGGL((k_sgbm_band<NP, true, TW>), dim3(div_up(g.W1, TW), nb), dim3(256), lds, ctx->stream, ctx->C, ctx->S, ctx->S + vol, nb, g, ctx->disp_tmp, ctx->ccl_size);
                else hipLaunchKernelGGL((k_sgbm_band<NP, false, TW>), dim3(div_up(g.W1, TW), nb), dim3(256), lds, ctx->stream, ctx->C, ctx->S, ctx->S + vol, nb, g, ctx->disp_tmp, ctx->ccl_size);
                VO_CHECK_LAUNCH(ctx);
            }
            ctx->last_paths = 4;
            return VO_OK;
        }
    }
    // MODE_HH with the pair schedule: all three opposite pairs that do not involve the final sweep's own direction
    // (W/E, NW/SE, NE/SW) as one stored volume each (k_sgbm_pair), the bottom-up vertical direction by the line kernel,
    // the fused sweep over four volumes.  Layout of S: [0..2] = the pair sums, [3] = S direction, [4..6] = checkpoints.
    if constexpr (NP % 2 == 0 && NP <= 8) {
        if (fuse && ctx->we_now && ctx->tune_pair_hh && ctx->tune_vwta32 && plan.lpw == 4 && plan_all.n_dirs == 8) {
            const bool pad = g.D != g.Dp;
            {
                StageTimer t(ctx, VO_T_SGBM_AGG);
                PathPlan pp = plan_all;
                const int src[3] = { 0, 2, 3 };                       // (1,0), (1,1), (-1,1): the forward member of each pair
                pp.n_dirs = 3;
                pp.first_wave[0] = 0;
                for (int k = 0; k < 3; k++) {
                    pp.sx[k] = plan_all.sx[src[k]]; pp.sy[k] = plan_all.sy[src[k]]; pp.nlines[k] = plan_all.nlines[src[k]];
                    pp.first_wave[k + 1] = pp.first_wave[k] + div_up(pp.nlines[k], 4);
                }
                for (int k = 3; k < VO_MAX_DIRS; k++) { pp.sx[k] = pp.sy[k] = pp.nlines[k] = 0; pp.first_wave[k + 1] = pp.first_wave[3]; }
                const int nwp = pp.first_wave[3];
                if (pad) hipLaunchKernelGGL((k_sgbm_pair<NP, true>), dim3(div_up(nwp, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, ctx->S + 4 * vol, vol, g, pp, ctx->dump);
                else hipLaunchKernelGGL((k_sgbm_pair<NP, false>), dim3(div_up(nwp, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, ctx->S + 4 * vol, vol, g, pp, ctx->dump);
                PathPlan pd = plan_all;
                pd.n_dirs = 1;
                pd.sx[0] = plan_all.sx[6]; pd.sy[0] = plan_all.sy[6]; pd.nlines[0] = plan_all.nlines[6];
                pd.first_wave[0] = 0;
                for (int k = 0; k < VO_MAX_DIRS; k++) pd.first_wave[k + 1] = div_up(pd.nlines[0], 4);
                for (int k = 1; k < VO_MAX_DIRS; k++) pd.sx[k] = pd.sy[k] = pd.nlines[k] = 0;
                const int nwaves = pd.first_wave[1];
                constexpr int PFD = NP <= 4 ? 8 : 4;
                if (pad) hipLaunchKernelGGL((k_sgbm_paths<NP, PFD, 16, true, 1>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S + 3 * vol, vol, g, pd, ctx->dump);
                else hipLaunchKernelGGL((k_sgbm_paths<NP, PFD, 16, false, 1>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S + 3 * vol, vol, g, pd, ctx->dump);
                VO_CHECK_LAUNCH(ctx);
            }
            {
                StageTimer t(ctx, VO_T_SGBM_WTA);
                constexpr int NP2 = NP / 2;
                const int nw2 = div_up(g.W1, 2);
                const size_t sh32 = (size_t)8 * 2 * g.Dp * sizeof(int16_t);
                if (ctx->tune_vwta_queued == 16 && ctx->tune_we_fuse == 2 && NP <= 4) {   // (8 registers per lane and volume: the 32-lane form wins, measured on config 4)
                    const int nw16 = div_up(g.W1, 4);
                    const size_t sh16 = (size_t)2 * 16 * g.Dp * sizeof(int16_t);
                    if (pad) hipLaunchKernelGGL((k_sgbm_vwta<NP, 4, true>), dim3(div_up(nw16, 4)), dim3(256), sh16, ctx->stream, ctx->C, ctx->S, vol, g, ctx->ccl_label, ctx->ccl_runlen);
                    else hipLaunchKernelGGL((k_sgbm_vwta<NP, 4, false>), dim3(div_up(nw16, 4)), dim3(256), sh16, ctx->stream, ctx->C, ctx->S, vol, g, ctx->ccl_label, ctx->ccl_runlen);
                } else if (pad) hipLaunchKernelGGL((k_sgbm_vwta32<NP2, 4, true>), dim3(div_up(nw2, 4)), dim3(256), sh32, ctx->stream, ctx->C, ctx->S, vol, g, ctx->ccl_label, ctx->ccl_runlen);
                else hipLaunchKernelGGL((k_sgbm_vwta32<NP2, 4, false>), dim3(div_up(nw2, 4)), dim3(256), sh32, ctx->stream, ctx->C, ctx->S, vol, g, ctx->ccl_label, ctx->ccl_runlen);
                hipLaunchKernelGGL(k_sgbm_fin, dim3(div_up(g.W1, 256), g.H), dim3(256), 0, ctx->stream, ctx->ccl_label, ctx->ccl_runlen, g, ctx->disp_tmp, ctx->ccl_size);
                VO_CHECK_LAUNCH(ctx);
            }
            ctx->last_paths = plan_all.n_dirs - 1;
            return VO_OK;
        }
    }
    if constexpr (NP % 2 == 0 && NP <= 8) {
        if (fuse && ctx->we_now && ctx->tune_vwta32 && plan.lpw == 4 && g.W1 % 8 == 0 && g.W1 >= 16) {
            const bool pad = g.D != g.Dp;
            const int ND = plan_all.n_dirs - 3;               // stored directions besides W and E: 2 (MODE_SGBM) or 5 (MODE_HH)
            {
                StageTimer t(ctx, VO_T_SGBM_AGG);
                const int nw = div_up(g.H, 4);
                int16_t* ck = ctx->S + (size_t)(ND + 1) * vol;
                if (pad) hipLaunchKernelGGL((k_sgbm_we<NP, true>), dim3(div_up(nw, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, ck, g, ctx->dump);
                else hipLaunchKernelGGL((k_sgbm_we<NP, false>), dim3(div_up(nw, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, ck, g, ctx->dump);
                PathPlan pd = plan_all;
                pd.n_dirs = ND;
                pd.first_wave[0] = 0;
                for (int k = 0; k < ND; k++) {
                    pd.sx[k] = plan_all.sx[2 + k]; pd.sy[k] = plan_all.sy[2 + k]; pd.nlines[k] = plan_all.nlines[2 + k];
                    pd.first_wave[k + 1] = pd.first_wave[k] + div_up(pd.nlines[k], 4);
                }
                for (int k = ND; k < VO_MAX_DIRS; k++) { pd.sx[k] = pd.sy[k] = pd.nlines[k] = 0; pd.first_wave[k + 1] = pd.first_wave[ND]; }
                const int nwaves = pd.first_wave[ND];
                constexpr int PFD = NP <= 4 ? 8 : 4;
                if (pad) hipLaunchKernelGGL((k_sgbm_paths<NP, PFD, 16, true, 1>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S + vol, vol, g, pd, ctx->dump);
                else hipLaunchKernelGGL((k_sgbm_paths<NP, PFD, 16, false, 1>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S + vol, vol, g, pd, ctx->dump);
                VO_CHECK_LAUNCH(ctx);
            }
            {
                StageTimer t(ctx, VO_T_SGBM_WTA);
                constexpr int NP2 = NP / 2;
                const int nw2 = div_up(g.W1, 2);
                const size_t sh32 = (size_t)8 * 2 * g.Dp * sizeof(int16_t);
#define LAUNCH_VWTA32_WE(NV, PAD) hipLaunchKernelGGL((k_sgbm_vwta32<NP2, NV, PAD>), dim3(div_up(nw2, 4)), dim3(256), sh32, ctx->stream, ctx->C, ctx->S, vol, g, ctx->ccl_label, ctx->ccl_runlen)
                bool done64 = false;
                if constexpr (NP % 4 == 0) {
                    if (ctx->tune_vwta64 && !pad && ND == 2) {
                        hipLaunchKernelGGL((k_sgbm_vwta64<NP / 4, 3>), dim3(div_up(g.W1, 4)), dim3(256), (size_t)4 * 2 * g.Dp * sizeof(int16_t), ctx->stream, ctx->C, ctx->S, vol, g, ctx->ccl_label, ctx->ccl_runlen);
                        done64 = true;
                    }
                }
                // a pair on this schedule is not waited for soon: the 16-lane form of the sweep (half the waves, a third
                // fewer instructions per pixel) serves throughput, the 32-lane form latency (VO_VWTA_QUEUED=32 to force it)
                const int nw16 = div_up(g.W1, 4);
                const size_t sh16 = (size_t)2 * 16 * g.Dp * sizeof(int16_t);
#define LAUNCH_VWTA16_WE(NV, PAD) hipLaunchKernelGGL((k_sgbm_vwta<NP, NV, PAD>), dim3(div_up(nw16, 4)), dim3(256), sh16, ctx->stream, ctx->C, ctx->S, vol, g, ctx->ccl_label, ctx->ccl_runlen)
                if (done64) {
                } else if (ctx->tune_vwta_queued == 16 && ctx->tune_we_fuse == 2 && NP <= 4) {
                    if (ND == 2) { if (pad) LAUNCH_VWTA16_WE(3, true); else LAUNCH_VWTA16_WE(3, false); }
                    else { if (pad) LAUNCH_VWTA16_WE(6, true); else LAUNCH_VWTA16_WE(6, false); }
                    done64 = true;
                }
#undef LAUNCH_VWTA16_WE
                if (done64) {
                } else if (ND == 2) { if (pad) LAUNCH_VWTA32_WE(3, true); else LAUNCH_VWTA32_WE(3, false); }
                else { if (pad) LAUNCH_VWTA32_WE(6, true); else LAUNCH_VWTA32_WE(6, false); }
#undef LAUNCH_VWTA32_WE
                hipLaunchKernelGGL(k_sgbm_fin, dim3(div_up(g.W1, 256), g.H), dim3(256), 0, ctx->stream, ctx->ccl_label, ctx->ccl_runlen, g, ctx->disp_tmp, ctx->ccl_size);
                VO_CHECK_LAUNCH(ctx);
            }
            ctx->last_paths = plan_all.n_dirs - 1;
            return VO_OK;
        }
    }
    {
        StageTimer t(ctx, VO_T_SGBM_AGG);
        const int nwaves = plan.first_wave[plan.n_dirs];
        const bool pad = g.D != g.Dp;
#define LAUNCH_PATHS(NPL, PF, LPL, PAD) hipLaunchKernelGGL((k_sgbm_paths<NPL, PF, LPL, PAD>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->C, ctx->S, vol, g, plan, ctx->dump)
        bool done = false;
        if constexpr (NP <= 4) {
            if (plan.lpw == 8) {
                done = true;
                if (ctx->tune_path_pf == 8) { if (pad) LAUNCH_PATHS(2 * NP, 8, 8, true); else LAUNCH_PATHS(2 * NP, 8, 8, false); }
                else { if (pad) LAUNCH_PATHS(2 * NP, 4, 8, true); else LAUNCH_PATHS(2 * NP, 4, 8, false); }
            }
        }
        if (done) {
        } else if (ctx->tune_path_pf == 8 && NP <= 4) { if (pad) LAUNCH_PATHS(NP, 8, 16, true); else LAUNCH_PATHS(NP, 8, 16, false); }
        else if (ctx->tune_path_pf == 2) { if (pad) LAUNCH_PATHS(NP, 2, 16, true); else LAUNCH_PATHS(NP, 2, 16, false); }
        else { if (pad) LAUNCH_PATHS(NP, 4, 16, true); else LAUNCH_PATHS(NP, 4, 16, false); }
#undef LAUNCH_PATHS
        VO_CHECK_LAUNCH(ctx);
    }
    {
        StageTimer t(ctx, VO_T_SGBM_WTA);
        const size_t sh = (size_t)16 * g.Dp * sizeof(int16_t);   // one row of S per 16-lane group (the fused sweep keeps two)
        if (fuse) {
            if (ctx->tune_vwta32 && NP % 2 == 0) {
                // 32 lanes per column: Dp / 64 registers per lane, two columns per wave
                constexpr int NP2 = NP % 2 == 0 ? NP / 2 : 1;
                const int nw2 = div_up(g.W1, 2);
                const size_t sh32 = (size_t)8 * 2 * g.Dp * sizeof(int16_t);   // 8 column groups per 256-thread block, two rows of S each
#define LAUNCH_VWTA32(NV, PAD) hipLaunchKernelGGL((k_sgbm_vwta32<NP2, NV, PAD>), dim3(div_up(nw2, 4)), dim3(256), sh32, ctx->stream, ctx->C, ctx->S, vol, g, ctx->ccl_label, ctx->ccl_runlen)
                bool done64 = false;
                if constexpr (NP % 4 == 0) {
                    if (ctx->tune_vwta64 && g.D == g.Dp && plan.n_dirs == 4) {
                        hipLaunchKernelGGL((k_sgbm_vwta64<NP / 4, 4>), dim3(div_up(g.W1, 4)), dim3(256), (size_t)4 * 2 * g.Dp * sizeof(int16_t), ctx->stream, ctx->C, ctx->S, vol, g, ctx->ccl_label, ctx->ccl_runlen);
                        done64 = true;
                    }
                }
                if (done64) {
                } else if (plan.n_dirs == 4) { if (g.D == g.Dp) LAUNCH_VWTA32(4, false); else LAUNCH_VWTA32(4, true); }
                else { if (g.D == g.Dp) LAUNCH_VWTA32(7, false); else LAUNCH_VWTA32(7, true); }
#undef LAUNCH_VWTA32
                hipLaunchKernelGGL(k_sgbm_fin, dim3(div_up(g.W1, 256), g.H), dim3(256), 0, ctx->stream, ctx->ccl_label, ctx->ccl_runlen, g, ctx->disp_tmp, ctx->ccl_size);
                VO_CHECK_LAUNCH(ctx);
                return VO_OK;
            }
            const int nw = div_up(g.W1, 4);
            // the sweep leaves a two-word record per pixel in ccl_label / ccl_runlen (free until the speckle
            // filter); k_sgbm_fin turns the records into disp1 + the disp2 atomicMin for all pixels in parallel
            // (an atomic or an LDS round trip inside the sequential sweep costs it a third of its time)
#define LAUNCH_VWTA(NV, PAD) hipLaunchKernelGGL((k_sgbm_vwta<NP, NV, PAD>), dim3(div_up(nw, 4)), dim3(256), 2 * sh, ctx->stream, ctx->C, ctx->S, vol, g, ctx->ccl_label, ctx->ccl_runlen)
            if (plan.n_dirs == 4) { if (g.D == g.Dp) LAUNCH_VWTA(4, false); else LAUNCH_VWTA(4, true); }
            else { if (g.D == g.Dp) LAUNCH_VWTA(7, false); else LAUNCH_VWTA(7, true); }
#undef LAUNCH_VWTA
            hipLaunchKernelGGL(k_sgbm_fin, dim3(div_up(g.W1, 256), g.H), dim3(256), 0, ctx->stream, ctx->ccl_label, ctx->ccl_runlen, g, ctx->disp_tmp, ctx->ccl_size);
        } else {
            const long long npix = (long long)g.W1 * g.H;
            hipLaunchKernelGGL((k_sgbm_wta<NP>), dim3((unsigned)((npix + 15) / 16)), dim3(256), sh, ctx->stream, ctx->S, vol, plan.n_dirs,
                               g, ctx->disp_tmp, ctx->ccl_size);
        }
        VO_CHECK_LAUNCH(ctx);
    }
    return VO_OK;
}

static int sgbm_run_impl(vo_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int w, int h, int16_t* d_disp);

// The SGBM workspaces (planes, C, L volumes, CCL arrays) are shared by the main and the look-ahead
// stream: a run on one stream must not start before the previous run -- possibly on the other
// stream -- has finished.  An event chain orders them on the device without blocking the host.
int sgbm_run(vo_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int w, int h, int16_t* d_disp)
{
    if (ctx->sgbm_done_valid) VO_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->sgbm_done, 0));
    int rc = sgbm_run_impl(ctx, dL, dR, w, h, d_disp);
    if (ctx->sgbm_done) {
        VO_HIP(ctx, hipEventRecord(ctx->sgbm_done, ctx->stream));
        ctx->sgbm_done_valid = true;
    }
    return rc;
}

static int sgbm_run_impl(vo_ctx* ctx, const uint8_t* dL, const uint8_t* dR, int w, int h, int16_t* d_disp)
{
    const SgbmEff& e = ctx->sg;
    if (!e.set) return vo_fail(ctx, VO_E_STATE, "vo_set_sgbm has not been called");
    SgbmGeom g;
    g.W = w; g.H = h; g.D = e.D; g.Dp = (e.D + 31) & ~31; g.minD = e.minD; g.P1 = e.P1; g.P2 = e.P2; g.ur = e.ur; g.d12 = e.d12;
    g.ftzero = e.ftzero; g.SW2 = e.SW2;
    g.minX1 = e.maxD > 0 ? e.maxD : 0;
    int maxX1 = w + (e.minD < 0 ? e.minD : 0);
    g.W1 = maxX1 - g.minX1;
    g.invalid16 = (e.minD - 1) * 16;
    const int n = w * h;
    if (g.W1 <= 0) {
        std::vector<int16_t> inv((size_t)n, (int16_t)g.invalid16);  // every pixel invalid
        VO_HIP(ctx, hipMemcpyAsync(d_disp, inv.data(), (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream));
        VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->last_cells = 0;
        return VO_OK;
    }
    const size_t vol = (size_t)g.W1 * h * g.Dp;
    if (vol > ctx->vol_cells || e.D > 256)
        return vo_fail(ctx, VO_E_CAP, "cost volume %zu cells exceeds the capacity given to vo_create (or D > 256)", vol);
    // 8 lanes per scan line (16 disparities per lane) up to D = 128 in the line scheme; 16 lanes per line otherwise
    const bool lanes8 = ctx->tune_path_lanes == 8 && g.Dp <= 128 && !(ctx->tune_raster && ctx->tune_fuse_wta && g.ur < 100);
    const PathPlan plan = make_plan(g, e.mode, lanes8 ? 8 : 4);
    ctx->last_cells = (int64_t)g.W1 * h * g.D;
    ctx->last_paths = (ctx->tune_fuse_wta && g.ur < 100) ? plan.n_dirs - 1 : plan.n_dirs;   // directions inside the k_sgbm_paths launch
    {
        StageTimer t(ctx, VO_T_SGBM_COST);
        const int dbg = ctx->tune_diag_dbg;          // development only (VO_DIAG_DEBUG): 4 / 8 / 16 / 32 skip the cost / W+E / diagonal / post stage
        hipLaunchKernelGGL(k_sgbm_planes, dim3(div_up(w, 256), h), dim3(256), 0, ctx->stream, dL, dR, w, h, g.ftzero,
                           ctx->planesL, ctx->planesR, ctx->ccl_size, ctx->rs_ctl, ctx->rs_ctl_words);
        const int bx = ((g.Dp / 2 + 63) / 64) * 64;
        const int TY = ctx->tune_sweep_ty;
        const int nw = bx / 64;
#define LAUNCH_SWEEP(XT, SW)                                                                                                   \
    hipLaunchKernelGGL((k_sgbm_cost_sweep<XT, SW>), dim3(div_up(g.W1, XT), div_up(h, TY)), dim3(bx),                            \
                       (size_t)nw * (2 * SW + 1) * XT * 64 * 4, ctx->stream, ctx->planesL, ctx->planesR, g, TY, ctx->C)
        if (dbg & 4) {
        } else
        switch (g.SW2) {
            case 0: LAUNCH_SWEEP(8, 0); break;
            case 1: LAUNCH_SWEEP(8, 1); break;
            case 2: if (ctx->tune_sweep_xt == 16) LAUNCH_SWEEP(16, 2); else LAUNCH_SWEEP(8, 2); break;
            case 3: LAUNCH_SWEEP(8, 3); break;
            case 4: LAUNCH_SWEEP(8, 4); break;
            case 5: LAUNCH_SWEEP(8, 5); break;
            default:
                hipLaunchKernelGGL(k_sgbm_cost_generic, dim3(g.W1, h), dim3(bx), 0, ctx->stream, ctx->planesL, ctx->planesR, g, ctx->C);
        }
#undef LAUNCH_SWEEP
        VO_CHECK_LAUNCH(ctx);
    }
    int rc;
    switch (g.Dp / 32) {
        case 1: rc = launch_agg<1>(ctx, g, plan, vol); break;
        case 2: rc = launch_agg<2>(ctx, g, plan, vol); break;
        case 3: rc = launch_agg<3>(ctx, g, plan, vol); break;
        case 4: rc = launch_agg<4>(ctx, g, plan, vol); break;
        case 5: rc = launch_agg<5>(ctx, g, plan, vol); break;
        case 6: rc = launch_agg<6>(ctx, g, plan, vol); break;
        case 7: rc = launch_agg<7>(ctx, g, plan, vol); break;
        default: rc = launch_agg<8>(ctx, g, plan, vol); break;
    }
    if (rc) return rc;
    if (ctx->stream_hi && !ctx->on_hi && ctx->cur_engine >= 0) {
        // the rest of this pair (post filters, then the ORB chain) is a chain of short latency-bound launches: it
        // continues on the engine's high-priority stream, behind everything queued so far
        hipEvent_t hop = ctx->la_hop[ctx->cur_engine];
        VO_HIP(ctx, hipEventRecord(hop, ctx->stream));
        VO_HIP(ctx, hipStreamWaitEvent(ctx->stream_hi, hop, 0));
        std::swap(ctx->stream, ctx->stream_hi);
        ctx->on_hi = true;
    }
    if (!(ctx->tune_diag_dbg & 32)) {
        StageTimer t(ctx, VO_T_SGBM_POST);
        hipLaunchKernelGGL(k_lr_median3, dim3(div_up(w, 256), h), dim3(256), 0, ctx->stream, ctx->disp_tmp, ctx->ccl_size, g, d_disp);
        if (e.speckleWindow > 0) {
            const int newVal = g.invalid16, maxDiff = 16 * e.speckleRange;
            hipLaunchKernelGGL(k_ccl_rows, dim3(h), dim3(256), 0, ctx->stream, d_disp, w, newVal, maxDiff, ctx->ccl_label, ctx->ccl_runlen, ctx->ccl_size);
            hipLaunchKernelGGL(k_ccl_vmerge, dim3(div_up(w, 256), h), dim3(256), 0, ctx->stream, d_disp, w, h, newVal, maxDiff, e.speckleWindow, ctx->ccl_label, ctx->ccl_runlen);
            hipLaunchKernelGGL(k_ccl_sizes, dim3(div_up(w, 256), h), dim3(256), 0, ctx->stream, d_disp, w, h, newVal, maxDiff, e.speckleWindow, ctx->ccl_label, ctx->ccl_runlen, ctx->ccl_size);
            hipLaunchKernelGGL(k_ccl_apply, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, d_disp, n, newVal, e.speckleWindow, ctx->ccl_label, ctx->ccl_size);
        }
        VO_CHECK_LAUNCH(ctx);
    }
    return VO_OK;
}
