// StereoSGBM on gfx950: replaces self.stereoSGBM.compute(L, R)
// [reference stereo_camera.py:23-27,51].  Arithmetic definition: OpenCV 4.x
// calib3d/src/stereosgbm.cpp (MODE_SGBM 5 paths / MODE_HH 8 paths), all-integer, so the
// output is bit-exact against the CPU restatement.
//
// Data layout in HBM
//   planesL[y][x]        2 x u32  : (u,u0,u1) bytes of the x-Sobel channel, then of the raw channel
//   planesR[k][y][p]     u32      : plane k of (v,v0,v1)x2, packed as value(p) | value(p-1) << 16
//   C[y][x1][Dp]         int16    : block cost (+P2), d fastest, Dp = D rounded up to 32, x1 = x - minX1
//   S[0][y][x1][Dp]      int16    : L_W + L_E, the only aggregated volume MODE_SGBM ever stores
//   S[1]                          : MODE_HH: S[0] + the three bottom-up directions;  S[2]: checkpoints of the W + E kernel
//   sw_bnd               u64      : boundary granules of the diagonal sweep [strip][H + 1][NP][64]
// Kernels (the schedule: launch_agg below)
//   k_sgbm_planes  prefilter + Birchfield-Tomasi half-pixel bounds (elementwise)
//   k_sgbm_cost_sweep  BT pixel cost + (2*SW2+1)^2 box sum; one lane = one disparity PAIR in packed
//                  int16x2 math; right-image planes travel through DPP lane-shift chains, the horizontal
//                  window slides in registers, the vertical one through an LDS ring; writes C once
//   k_sgbm_we      W and E as ONE stored volume: a scan line lives in one 16-lane DPP row (each lane holds
//                  Dp/16 consecutive disparities in packed registers), so a wave advances 4 image rows;
//                  neighbours d-1/d+1 come from row_shr/row_shl (row ends are the MAX_COST sentinels for
//                  free), the min over d is a 4-step row_ror all-reduce; E is kept as checkpoints every 8th
//                  column and recomputed per segment while W passes (k_sgbm_pair: the same for any line length)
//   k_sgbm_diag    (sgbm_diag.inc) NW / N / NE computed together in skewed columns, added to the W + E volume,
//                  winner-take-all on the total: leaves a 2-word record per pixel.  MODE_HH: a reverse pass
//                  first adds SW / S / SE to the volume.  No direction of these three is ever stored
//   k_sgbm_fin     records -> sub-pixel disp1 + disp2 candidates via atomicMin on (cost, scan order) keys
//   k_sgbm_paths + k_sgbm_wta   uniquenessRatio >= 100 only (no threshold form of the uniqueness test): one
//                  stored volume per direction, per pixel (16 lanes) S = sat-sum of all of them, winner logic inline
//   k_lr_median3   left-right consistency check evaluated inside medianBlur(3)
//   k_ccl_*        filterSpeckles (run-based union-find labelling)
#include "vo_internal.h"
#include <stdlib.h>
#include <algorithm>
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
__device__ __forceinline__ uint32_t pk_shr2(uint32_t a)       // both halves >> 2 (v_pk_lshrrev_b16)
{
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) >> (u16x2){2, 2});
}
__device__ __forceinline__ uint32_t pk_rep(int v) { return (uint32_t)(v & 0xFFFF) * 0x00010001u; }
// both halves = v + (the halves of add): one v_mad_u32_u24 when v < 2^16 and no half overflows (v a 15-bit cost, add = P2 twice)
__device__ __forceinline__ uint32_t pk_rep_add(uint32_t v, uint32_t add) { return __umul24(v, 0x00010001u) + add; }

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
    uint32_t urM, urAdd;   // ceil(a / (100 - ur)) = mulhi(a + urAdd, urM) for a < 2^22: urM = ceil(2^32 / (100 - ur)), urAdd = 99 - ur
                           // (100 - ur == 1: urM = 2^32 - 1, urAdd = 1)
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
                              int* __restrict__ sw_ctl, int sw_ctl_words, uint32_t* __restrict__ c_dummy, int dummy_words, uint32_t P2_2,
                              uint8_t* __restrict__ keepL, uint8_t* __restrict__ keepR)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    // (a few hundred workgroups walking the rows instead of one per 256 pixels: a short kernel's workgroups each have to win a
    // slot from the dispatcher against the other pairs' sweeps)
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
    // one row of cells holding P2 (= cost 0) right behind the used part of the cost volume: what the diagonal sweep reads for a
    // pixel outside the image (such a pixel carries the border state exactly when its cost is zero)
    if (y == 1 && blockIdx.x == 0 && (int)threadIdx.x < dummy_words) c_dummy[threadIdx.x] = P2_2;
    // the diagonal sweeps' work-item tickets, "a wait gave up" words and timelines of this run start at zero (k_sgbm_fin of
    // the previous run in this workspace has already reported its word to the slot it filled)
    if (y == 0 && blockIdx.x == 0)
        for (int i = threadIdx.x; i < sw_ctl_words; i += blockDim.x) sw_ctl[i] = 0;
    if (x >= W) return;
    size_t i = (size_t)y * W + x, plane = (size_t)W * H;
    // (inputs resident in HBM and already rectified: the pair is read where it lies and its copy into the frame slot -- which
    // ORB and the downloads read later -- is written here instead of by two copy commands in front of this kernel)
    if (keepL) { keepL[i] = L[i]; keepR[i] = R[i]; }
    if (d2key) d2key[i] = D2_EMPTY;   // uniquenessRatio >= 100 only (k_sgbm_wta keeps disp2 in HBM): its candidates start empty
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
        // distance of a value to an interval [lo, hi], lo <= hi, all in 0..255: at most one of (x - hi), (lo - x) is positive,
        // so max(0, x - hi, lo - x) = usat(x - hi) | usat(lo - x)  (3 packed operations instead of 4)
        const uint32_t c0 = pk_sub_sat_u(U, V[c * 3 + 2]) | pk_sub_sat_u(V[c * 3 + 1], U);
        const uint32_t c1 = pk_sub_sat_u(V[c * 3 + 0], U1) | pk_sub_sat_u(U0, V[c * 3 + 0]);
        uint32_t m = pk_min(c0, c1);
        if (c == 1) m = pk_shr2(m);
        acc = pk_add(acc, m);
    }
    return acc;
}

template <int XT, int SW2>
__global__ void __launch_bounds__(128) k_sgbm_cost_sweep(const uint32_t* __restrict__ PL, const uint32_t* __restrict__ PR,
                                                        SgbmGeom g, int TY, int16_t* __restrict__ C)
{
    constexpr int WIN = 2 * SW2 + 1, NC = XT + 2 * SW2;
    // The vertical window's ring -- the horizontal sums of the last WIN rows per column and lane.  Blocks up to 5 x 5: 40 registers
    // (the row loop is unrolled by WIN, so every ring index is a constant); the LDS the sweep then asks for is 3.3 KB per wave and
    // the mix is sensitive to exactly that.  Larger blocks (56 .. 88 words per lane) keep the ring in LDS.
    constexpr bool RING_REGS = WIN <= 5;
    extern __shared__ uint32_t s_ring[];  // [waves][WIN][XT][64] when the ring lives here, then the interior strips' staging [waves][12][NJ]
    uint32_t ringv[RING_REGS ? WIN : 1][XT];
    uint32_t* const ring = s_ring + (size_t)(threadIdx.x >> 6) * WIN * XT * 64 + (threadIdx.x & 63);   // (LDS variant only)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int dpl = threadIdx.x;                  // disparity pair index (padded layout)
    const bool pad = 2 * dpl >= g.D;
    // XCD-aware tile order: consecutive workgroups go to the 8 XCDs round-robin, each with an L2 of its own -- give every XCD a
    // contiguous band of tile rows, so that a row of the image planes is fetched into one L2 (plus the bands' halos), not eight
    // (the launch is one-dimensional: 8 x ceil(tiles / 8) workgroups)
    const int gx = (g.W1 + XT - 1) / XT, total = gx * ((g.H + TY - 1) / TY);
    const int chunk = (total + 7) >> 3, tile = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (tile >= total) return;
    const int tbx = tile % gx, tby = tile / gx;
    const int xa = tbx * XT, ya = tby * TY;
    const size_t plane = (size_t)g.W * g.H;
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

    // Interior strips (every column of the strip + halo inside the image, every lane's right-image positions inside the row):
    // the six right-image planes of the row segment this wave needs -- positions Pmin .. Pmin + 126 + NC - 1 -- are staged in
    // LDS once per row, split by parity (lane l reads position x - minD - 2 l: one parity per column, consecutive words for
    // consecutive lanes, so the reads are conflict-free and their addresses are immediates on one register).  A column then
    // costs its BT arithmetic only; the shift chains of the general path below (a v_readlane + v_mov + DPP shift per plane and
    // column) remain for the strips at the image border.
    {
        constexpr int NJ = 64 + (NC - 1) / 2;                          // words per (plane, parity)
        const int x0img = xa - SW2 + g.minX1;                          // image column of halo column 0
        const int pmin = x0img - g.minD - lane0_d - 126;               // lane 63's position at halo column 0
        const bool interior = xa >= SW2 && xa - SW2 + NC <= g.W1 && pmin >= 0 && pmin + 2 * NJ <= g.W;
        if (interior) {
            typedef __attribute__((address_space(3))) uint32_t lds_w;
            typedef uint32_t w2 __attribute__((ext_vector_type(2)));
            typedef __attribute__((address_space(3))) w2 lds_w2;
            // staging layout: [parity][j][the six planes] -- a lane fetches the six planes of its position with three 8-byte reads
            // at immediate offsets; consecutive lanes are 6 words apart, which spreads every 32-lane pass over all 64 banks
            lds_w* const stage = (lds_w*)(s_ring + (RING_REGS ? 0 : (size_t)(blockDim.x >> 6) * WIN * XT * 64)) + (size_t)wv * 12 * NJ;
            lds_w* const st_wr = stage + lane * 6;                     // + parity * NJ * 6  (+ 64 * 6 for the tail lanes)
            const lds_w* const st_rd = stage + (63 - lane) * 6;        // + ((k & 1) * NJ + (k >> 1)) * 6
            const bool tail = lane < NJ - 64;
            const int lrun = min(lane, NC - 1);
            auto row_of = [&](int rr) { return (size_t)min(max(ya - SW2 + rr, 0), g.H - 1) * g.W; };
            w2 pfm[6], pft[6];                                         // the next row's words, in flight while this row is computed
            uint32_t pl0, pl1;
            // buffer addressing: a resource descriptor (scalar registers) + a scalar byte offset that carries everything uniform
            // (plane, row, strip) + one constant per-lane offset -- the loads and stores of the loop need no vector address
            // arithmetic at all (global_load / global_store would spend ~50 64-bit vector adds per row on it)
            const __amdgpu_buffer_rsrc_t rPR = __builtin_amdgcn_make_buffer_rsrc((void*)PR, 0, (int)(6 * plane * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t rPL = __builtin_amdgcn_make_buffer_rsrc((void*)PL, 0, (int)(2 * plane * 4), 0x00020000);
            const int voff = 8 * lane, voff_t = voff + (tail ? 512 : 0), vpl = 8 * lrun;
            auto fetch = [&](int rr) {
                const size_t rowi = row_of(rr);
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const int so = (int)(((size_t)q * plane + rowi + pmin) * 4);           // uniform
                    pfm[q] = __builtin_amdgcn_raw_buffer_load_b64(rPR, voff, so, 0);
                    pft[q] = __builtin_amdgcn_raw_buffer_load_b64(rPR, voff_t, so, 0);
                }
                const w2 lw = __builtin_amdgcn_raw_buffer_load_b64(rPL, vpl, (int)((rowi + x0img) * 8), 0);
                pl0 = lw.x;
                pl1 = lw.y;
            };
            fetch(0);
            for (int rr0 = 0; rr0 < nrows; rr0 += WIN) {
#pragma unroll RING_REGS ? WIN : 1
            for (int slot = 0; slot < WIN; slot++) {
                const int rr = rr0 + slot;
                if (rr >= nrows) break;                                // (uniform)
#pragma unroll
                for (int h = 0; h < 3; h++) {
                    *(lds_w2*)(st_wr + 2 * h) = (w2){ pfm[2 * h].x, pfm[2 * h + 1].x };
                    *(lds_w2*)(st_wr + NJ * 6 + 2 * h) = (w2){ pfm[2 * h].y, pfm[2 * h + 1].y };
                }
                if (tail) {
#pragma unroll
                    for (int h = 0; h < 3; h++) {
                        *(lds_w2*)(st_wr + 64 * 6 + 2 * h) = (w2){ pft[2 * h].x, pft[2 * h + 1].x };
                        *(lds_w2*)(st_wr + (NJ + 64) * 6 + 2 * h) = (w2){ pft[2 * h].y, pft[2 * h + 1].y };
                    }
                }
                const uint32_t plr0 = pl0, plr1 = pl1;
                fetch(min(rr + 1, nrows - 1));
                uint32_t pc[NC];
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    const lds_w* const sp = st_rd + ((k & 1) * NJ + (k >> 1)) * 6;
                    const w2 va = *(const lds_w2*)sp, vb = *(const lds_w2*)(sp + 2), vc = *(const lds_w2*)(sp + 4);
                    const uint32_t V[6] = { va.x, va.y, vb.x, vb.y, vc.x, vc.y };
                    const uint32_t lw0 = (uint32_t)__builtin_amdgcn_readlane((int)plr0, k);
                    const uint32_t lw1 = (uint32_t)__builtin_amdgcn_readlane((int)plr1, k);
                    pc[k] = bt_regs(lw0, lw1, V);
                }
                uint32_t s = 0;
#pragma unroll
                for (int k = 0; k < WIN; k++) s = pk_add(s, pc[k]);
#pragma unroll
                for (int j = 0; j < XT; j++) {
                    if (j > 0) s = pk_sub(pk_add(s, pc[j + WIN - 1]), pc[j - 1]);
                    if constexpr (RING_REGS) {
                        if (rr >= WIN) acc[j] = pk_sub(acc[j], ringv[slot][j]);
                        acc[j] = pk_add(acc[j], s);
                        ringv[slot][j] = s;
                    } else {
                        uint32_t* cell = ring + (size_t)(slot * XT + j) * 64;
                        if (rr >= WIN) acc[j] = pk_sub(acc[j], *cell);
                        acc[j] = pk_add(acc[j], s);
                        *cell = s;
                    }
                }
                if (rr >= WIN - 1) {
                    const int y = ya + rr - (WIN - 1);
                    // (a descriptor per row segment: a 4K volume exceeds the 4 GB one descriptor can span; lanes past Dp are cut
                    // off by its bounds check)
                    int16_t* const crow = C + ((size_t)y * g.W1 + xa) * g.Dp;              // uniform
                    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)crow, 0, XT * g.Dp * 2, 0x00020000);
#pragma unroll
                    for (int j = 0; j < XT; j++)
                        __builtin_amdgcn_raw_buffer_store_b32(pad ? MAXC2 : acc[j], rC, 2 * dpl < g.Dp ? 4 * dpl : 0x7FFFFFF0, j * g.Dp * 2, 0);
                }
            }
            }
            return;
        }
    }

    for (int rr0 = 0; rr0 < nrows; rr0 += WIN) {
#pragma unroll RING_REGS ? WIN : 1
    for (int slot = 0; slot < WIN; slot++) {
        const int rr = rr0 + slot;
        if (rr >= nrows) break;                                        // (uniform)
        const int r = min(max(ya - SW2 + rr, 0), g.H - 1);
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
            if constexpr (RING_REGS) {
                if (rr >= WIN) acc[j] = pk_sub(acc[j], ringv[slot][j]);   // the row leaving the window
                acc[j] = pk_add(acc[j], s);
                ringv[slot][j] = s;
            } else {
                uint32_t* cell = ring + (size_t)(slot * XT + j) * 64;
                if (rr >= WIN) acc[j] = pk_sub(acc[j], *cell);
                acc[j] = pk_add(acc[j], s);
                *cell = s;
            }
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
// the d-1 / d+1 windows are shared between neighbouring registers: window j = (L[2j-1], L[2j]); a scan line (or a pixel) is
// one 16-lane DPP row, so the row ends take the MAX_COST sentinels for free
template <int NP, bool PAD>
__device__ __forceinline__ LV<NP> path_step2(const LV<NP>& Cp, const LV<NP>& Lp, uint32_t delta2, uint32_t P1_2, unsigned padreg)
{
    LV<NP> out;
    const uint32_t prev_hi = DPP(MAXC2, Lp.r[NP - 1], ROW_SHR1);  // lane-1's last register; row start keeps the sentinel
    const uint32_t next_lo = DPP(MAXC2, Lp.r[0], ROW_SHL1);       // lane+1's first register; row end keeps the sentinel
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
    int first_wave[VO_MAX_DIRS + 1];  // prefix sum of ceil(nlines / 4): four scan lines per wave
};

// One stored volume per direction: only the schedule without the threshold form of the uniqueness test (uniquenessRatio >= 100)
// still runs it.  NP = registers per lane (2 NP disparities), 16 lanes per scan line, four lines per wave.
template <int NP, int PF, bool PAD>
__global__ void __launch_bounds__(256) k_sgbm_paths(const int16_t* __restrict__ C, int16_t* __restrict__ Lbase, size_t vol,
                                                   SgbmGeom g, PathPlan plan, int16_t* __restrict__ dump)
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
    for (int i0 = 0; i0 < nmax; i0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int i = i0 + k;
            const LV<NP> Cv = cbuf[k];
            cbuf[k] = lv_load<NP>(pld);
            pld += (i + PF < last) ? stride : 0;
            const LV<NP> L = path_step2<NP, PAD>(Cv, Lp, delta2, P1_2, padreg);
            delta2 = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(L))), P2_2);
            Lp = L;
            lv_store_nt<NP>(i < n ? pst : sink, L);
            pst += stride;
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

// The same two passes with a row CUT IN TWO at its middle column m = W1 / 2 (round 5): two waves per group of four rows, each
// owning a half, both running the same code with opposite signs.  "Forward" is the direction that leaves the middle (W for the
// right half [m, W1), E for the left half [0, m)), "backward" the one that arrives there.  Phase 1: each wave runs its backward
// direction from the image border to the middle, keeping every 8th column (checkpoints) -- and what it holds at the end is
// exactly the state the OTHER half's forward direction starts from (L_W(m - 1) for the right half, L_E(m) for the left one): one
// exchange through LDS behind the kernel's only barrier.  Phase 2: k_sgbm_we's second pass, outward from the middle.  The same
// 3 path steps per cell and the same bytes; a wave's chain is 1.5 W1 dependent steps instead of 3 W1 and twice as many waves
// share the work (360 at config 2).  Needs W1 % 16 == 0 (otherwise k_sgbm_we).
template <int NP, bool PAD>
__global__ void __launch_bounds__(256) k_sgbm_we2(const int16_t* __restrict__ C, int16_t* __restrict__ Swe, int16_t* __restrict__ ckpt,
                                                 SgbmGeom g, int16_t* __restrict__ dump)
{
    __shared__ uint32_t s_x[4][NP][64];
    const int lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15, wv = threadIdx.x >> 6;
    const int group = blockIdx.x * 2 + (wv >> 1);          // four rows
    const bool right = (wv & 1) != 0;                      // this wave's half
    const int y = group * 4 + row;
    const bool live = y < g.H;                             // (a group past the image still runs: the barrier below wants every wave)
    const int yc = live ? y : g.H - 1;
    const int W1 = g.W1, nh = W1 >> 1, nseg = nh >> 3, nseg_row = W1 >> 3;
    const int d0 = l16 * 2 * NP;
    unsigned padreg = 0;
    if constexpr (PAD) {
#pragma unroll
        for (int k = 0; k < NP; k++) padreg |= (unsigned)(d0 + 2 * k >= g.D) << k;
    }
    const uint32_t P1_2 = pk_rep(g.P1), P2_2 = pk_rep(g.P2);
    const ptrdiff_t Dp = g.Dp;
    const ptrdiff_t fs = right ? Dp : -Dp;                 // one column in the forward direction
    const int x0 = right ? nh : nh - 1;                    // forward index f <-> column x0 + (right ? f : -f)
    const int16_t* const cf = C + ((size_t)yc * W1 + x0) * Dp + d0;               // cost cells of forward index 0
    int16_t* const sink = dump + lane * 2 * NP;
    int16_t* const krow = live ? ckpt + ((size_t)yc * nseg_row + (right ? nseg : 0)) * Dp + d0 : sink;   // this half's checkpoints: L_back at f = 8 s
    const ptrdiff_t kstep = live ? Dp : 0;
    LV<NP> border;
#pragma unroll
    for (int k = 0; k < NP; k++) border.r[k] = ((padreg >> k) & 1u) ? MAXC2 : 0u;

    // ---- phase 1: the backward direction, from the border (f = nh - 1) to the middle (f = 0), keeping every 8th column ----
    LV<NP> Lp = border;
    {
        uint32_t delta2 = P2_2;
        LV<NP> cbuf[8];
#pragma unroll
        for (int k = 0; k < 8; k++) cbuf[k] = lv_load<NP>(cf + (ptrdiff_t)(nh - 1 - k) * fs);
        const int16_t* pld = cf + (ptrdiff_t)(nh - 9) * fs;        // (nh >= 16 is checked by the launcher)
        for (int s = nseg - 1; s >= 0; s--) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const LV<NP> Cv = cbuf[k];
                cbuf[k] = lv_load<NP>(pld);
                pld -= (8 * s - k - 1 > 0) ? fs : 0;               // next forward index to fetch is 8 s - k - 1; stop at 0
                const LV<NP> L = path_step2<NP, PAD>(Cv, Lp, delta2, P1_2, padreg);
                delta2 = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(L))), P2_2);
                Lp = L;
                if (k == 7) lv_store<NP>(krow + (ptrdiff_t)s * kstep, L);   // forward index 8 s
            }
        }
    }
    // what this wave holds now (its backward direction at f = 0) is the other half's forward predecessor
#pragma unroll
    for (int k = 0; k < NP; k++) s_x[wv][k][lane] = Lp.r[k];
    // (the checkpoints are read back by the lanes that wrote them: let the stores land before the first load is issued)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- phase 2: per segment the backward direction again (from the checkpoint on its far side), then forward, store the sum ----
    {
        LV<NP> LpF;
#pragma unroll
        for (int k = 0; k < NP; k++) LpF.r[k] = s_x[wv ^ 1][k][lane];
        uint32_t dF = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(LpF))), P2_2);
        LV<NP> cseg[8], cnext[8], ck, cknext;
#pragma unroll
        for (int k = 0; k < 8; k++) cseg[k] = lv_load<NP>(cf + (ptrdiff_t)k * fs);
        ck = lv_load<NP>(krow + (ptrdiff_t)(nseg > 1 ? 1 : 0) * kstep);
        int16_t* pst = live ? Swe + ((size_t)yc * W1 + x0) * Dp + d0 : sink;
        const ptrdiff_t sstep = live ? fs : 0;
        for (int s = 0; s < nseg; s++) {
            const int sn = min(s + 1, nseg - 1);
#pragma unroll
            for (int k = 0; k < 8; k++) cnext[k] = lv_load<NP>(cf + (ptrdiff_t)(8 * sn + k) * fs);
            cknext = lv_load<NP>(krow + (ptrdiff_t)min(sn + 1, nseg - 1) * kstep);
            const bool last = s == nseg - 1;                          // the half's last segment starts the backward direction from the border
            LV<NP> LpB;
#pragma unroll
            for (int q = 0; q < NP; q++) LpB.r[q] = last ? border.r[q] : ck.r[q];
            uint32_t dB = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(ck))), P2_2);
            dB = last ? P2_2 : dB;
            LV<NP> Lb[8];
#pragma unroll
            for (int k = 7; k >= 0; k--) {
                Lb[k] = path_step2<NP, PAD>(cseg[k], LpB, dB, P1_2, padreg);
                dB = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(Lb[k]))), P2_2);
                LpB = Lb[k];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const LV<NP> Lf = path_step2<NP, PAD>(cseg[k], LpF, dF, P1_2, padreg);
                dF = pk_add(pk_rep((int)row_min_u32(lane_min16<NP>(Lf))), P2_2);
                LpF = Lf;
                LV<NP> S;
#pragma unroll
                for (int q = 0; q < NP; q++) S.r[q] = pk_add_sat(Lf.r[q], Lb[k].r[q]);
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
        // exact ceil(minS*100 / ur100) by a multiply-high with the reciprocal the host prepared (a < 2^22, ur100 <= 100)
        const int a = minS * 100;
        const int T = (int)__umulhi((unsigned)a + g.urAdd, g.urM);
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

// The diagonal sweep's form of the same test.  The uniqueness rule "no d outside best-1 .. best+1 with S(d) * (100 - ur) <
// minS * 100" is S(d) < T with T = ceil(minS * 100 / (100 - ur)) (non-negative integers), i.e. it fails exactly when MORE
// costs lie below T than the (up to three) excluded ones do.  So this half only COUNTS the costs below T -- four saturating
// subtractions, four clamps to {0, 1}, a packed sum and one 16-lane row sum -- and returns T; the caller knows minS and, one
// row later, the winner's two neighbour costs (they come back from LDS for the sub-pixel fit anyway) and compares the count
// with the number of excluded costs below T.  No per-element exclusion masks (they were ~45 instructions per pixel row).
template <int NP, bool PAD>
__device__ __forceinline__ void wta_count(const LV<NP>& S, const SgbmGeom& g, int lane, int& minS, int& best, int& cnt, int& Tout)
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
    // exact ceil(minS*100 / (100 - ur)) by a multiply-high with the reciprocal the host prepared (a < 2^22, 100 - ur <= 100)
    const int T = min((int)__umulhi((unsigned)(minS * 100) + g.urAdd, g.urM), 32768);   // S <= 32767: a larger T changes nothing
    const uint32_t T2 = pk_rep(T);
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        typedef uint16_t u16x2v __attribute__((ext_vector_type(2)));
        const u16x2v below = __builtin_elementwise_min(__builtin_bit_cast(u16x2v, pk_sub_sat_u(T2, S.r[k])), (u16x2v){1, 1});   // 1 where S < T
        c = pk_add(c, __builtin_bit_cast(uint32_t, below));
    }
    uint32_t n = (c & 0xFFFFu) + (c >> 16);
    n += DPP(0u, n, ROW_ROR(8));
    n += DPP(0u, n, ROW_ROR(4));
    n += DPP(0u, n, ROW_ROR(2));
    n += DPP(0u, n, ROW_ROR(1));
    // (padded disparities hold the saturated sum 32767: they count only when T is 32768, and there are Dp - D of them)
    cnt = (int)n - ((PAD && T > 32767) ? g.Dp - g.D : 0);
    Tout = T;
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
        // (minS == 32767: every sum saturated -- OpenCV's search starts from minS = SHRT_MAX with `<`, finds nothing, and the pixel
        //  comes out as (minD - 1) * 16 = INVALID)
        if (!row_viol && minS < MAXC) {
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

// Second half of the fused sweep's WTA as a pass of its own (images too wide for k_sgbm_post_rows' LDS rows), one thread per pixel: the sweep leaves a two-word record
// (aux0 = minS << 8 | best or -1 when the uniqueness test failed, aux1 = S[best-1] << 16 | S[best+1]);
// this pass turns it into the sub-pixel disp1 and the disp2 candidates (atomicMin).
// It also reports the run's health: when a wait inside one of its sweeps gave up (word 1 of a control block), the run's
// generation goes into the pinned word of the slot it fills (FrameSlot::sweep_word) and the context's counter is bumped.
__global__ void k_sgbm_fin(const int* __restrict__ aux0, const int* __restrict__ aux1, SgbmGeom g,
                           int16_t* __restrict__ disp1, int* __restrict__ d2key, const int* __restrict__ ctlA,
                           const int* __restrict__ ctlB, int* sweep_word, int gen, int* __restrict__ sweep_errs)
{
    const int x1 = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x1 == 0 && y == 0 && (ctlA[1] | ctlB[1])) {
        __hip_atomic_store(sweep_word, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        atomicAdd(sweep_errs, 1);
    }
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

// ---- the diagonal sweep's records -> final disparity rows, in ONE launch (was: k_sgbm_fin, k_lr_median3, k_ccl_rows) ----------
// Everything between the winner records and the labelled runs of filterSpeckles is local to a few image rows: the disp2
// candidates of a row come from that row's pixels only (computeDisparitySGBM fills disp2 per row), the left-right check reads
// its own row, medianBlur(3) reads rows y - 1 .. y + 1, and the run labelling is per row.  A block therefore takes RB rows:
// it turns the records of rows y0 - 1 .. y0 + RB into sub-pixel disp1 and the disp2 keys (atomicMin in LDS instead of HBM),
// applies the left-right check in place, writes the 3x3 medians of its RB rows to the disparity image and labels their runs
// (one wave per row, wave-level prefix maximum instead of the block-wide scan).  The two halo rows are recomputed by the
// neighbouring blocks (25 % at RB = 8); disp1 / disp2 never exist in HBM.  Also reports the run's health (see k_sgbm_planes).
__global__ void __launch_bounds__(512) k_sgbm_post_rows(const int* __restrict__ aux0, const int* __restrict__ aux1, SgbmGeom g, int RB,
                                                       int16_t* __restrict__ dst, int do_ccl, int maxDiff,
                                                       int* __restrict__ L, int* __restrict__ runlen, int* __restrict__ size,
                                                       const int* __restrict__ ctlA, const int* __restrict__ ctlB, int* sweep_word, int gen,
                                                       int* __restrict__ sweep_errs)
{
    extern __shared__ __attribute__((aligned(16))) int s_post[];
    const int W = g.W, H = g.H, R = RB + 2;
    int* const d2 = s_post;                                   // [R][W] disp2 keys; later [RB][W] int16 medians
    int16_t* const d1 = (int16_t*)(s_post + (size_t)R * W);   // [R][W] disp1, then the left-right-checked values
    const int y0 = blockIdx.x * RB, tid = threadIdx.x, nt = blockDim.x;
    if (blockIdx.x == 0 && tid == 0 && (ctlA[1] | ctlB[1])) {
        __hip_atomic_store(sweep_word, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        atomicAdd(sweep_errs, 1);
    }
    // LDS row r holds image row y0 - 1 + r (rows outside the image are never read: the median clamps its row indices)
    for (int i = tid; i < R * W; i += nt) { d2[i] = D2_EMPTY; d1[i] = (int16_t)g.invalid16; }
    __syncthreads();
    // (four records per thread in flight: the two loads of a pixel are independent of each other and of the other pixels')
    const int items = R * g.W1;
    for (int i0 = tid; i0 < items; i0 += 4 * nt) {
        int a[4], rr[4], xi[4];
        uint32_t nb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + k * nt;
            rr[k] = i / g.W1;
            xi[k] = i - rr[k] * g.W1 + g.minX1;
            const int y = y0 - 1 + rr[k];
            const bool ok = i < items && y >= 0 && y < H;
            const size_t o = ok ? (size_t)y * W + xi[k] : 0;
            a[k] = aux0[o];
            nb[k] = (uint32_t)aux1[o];
            if (!ok) rr[k] = -1;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (rr[k] < 0) continue;
            const int ximg = xi[k], r = rr[k];
            int out = g.invalid16;
            if (a[k] >= 0) {
                const int minS = a[k] >> 8, best = a[k] & 255;
                int dd = best * 16;
                if (best > 0 && best < g.D - 1) dd = wta_subpixel(best, minS, (int)(nb[k] >> 16), (int)(nb[k] & 0xFFFFu));
                out = dd + g.minD * 16;
                // disp2: lowest cost wins, ties go to the pixel OpenCV scans first (largest x)
                atomicMin(&d2[r * W + ximg - best - g.minD], (minS << 16) | (0xFFFF - ximg));
            }
            d1[r * W + ximg] = (int16_t)out;
        }
    }
    __syncthreads();
    // left-right check, in place (a pixel reads its own disp1 and two disp2 entries of its row)
    for (int r = 0; r < R; r++) {
        const int y = y0 - 1 + r;
        if (y < 0 || y >= H) continue;
        for (int x = g.minX1 + tid; x < g.minX1 + g.W1; x += nt) {
            int v = d1[r * W + x];
            if (v != g.invalid16) {
                auto disp2 = [&](int xx) -> int {
                    const int k = d2[r * W + xx];
                    return k == D2_EMPTY ? g.invalid16 : (0xFFFF - (k & 0xFFFF)) - xx;  // ximg - x2 = d + minD
                };
                const int _d = v >> 4, d_ = (v + 15) >> 4;
                const int _x = x - _d, x_ = x - d_;
                if (0 <= _x && _x < W && 0 <= x_ && x_ < W) {
                    const int a = disp2(_x), b = disp2(x_);
                    if (a >= g.minD && abs(a - _d) > g.d12 && b >= g.minD && abs(b - d_) > g.d12) d1[r * W + x] = (int16_t)g.invalid16;
                }
            }
        }
    }
    __syncthreads();
    // 3x3 median of the checked values (border rows / columns replicated); a copy stays in LDS for the run labelling
    int16_t* const med = (int16_t*)d2;                        // (the disp2 keys are dead since the barrier above)
    int p[9];
    for (int i = tid; i < RB * W; i += nt) {
        const int rr = i / W, x = i - rr * W, y = y0 + rr;
        if (y >= H) break;
        const int ra = (max(y - 1, 0) - (y0 - 1)) * W, rb = (rr + 1) * W, rc = (min(y + 1, H - 1) - (y0 - 1)) * W;
        const int xl = max(x - 1, 0), xr = min(x + 1, W - 1);
        p[0] = d1[ra + xl]; p[1] = d1[ra + x]; p[2] = d1[ra + xr];
        p[3] = d1[rb + xl]; p[4] = d1[rb + x]; p[5] = d1[rb + xr];
        p[6] = d1[rc + xl]; p[7] = d1[rc + x]; p[8] = d1[rc + xr];
        cswap(p[1], p[2]); cswap(p[4], p[5]); cswap(p[7], p[8]); cswap(p[0], p[1]);
        cswap(p[3], p[4]); cswap(p[6], p[7]); cswap(p[1], p[2]); cswap(p[4], p[5]);
        cswap(p[7], p[8]); cswap(p[0], p[3]); cswap(p[5], p[8]); cswap(p[4], p[7]);
        cswap(p[3], p[6]); cswap(p[1], p[4]); cswap(p[2], p[5]); cswap(p[4], p[7]);
        cswap(p[4], p[2]); cswap(p[6], p[4]); cswap(p[4], p[2]);
        dst[(size_t)y * W + x] = (int16_t)p[4];
        med[i] = (int16_t)p[4];
    }
    if (!do_ccl) return;
    __syncthreads();
    // filterSpeckles, first step: label[i] = index of the head of i's horizontal run (or -1), runlen[head], size[head] = 0 --
    // one wave per row: every lane scans a chunk of consecutive pixels, the last run start before a chunk comes from an
    // inclusive prefix maximum over the lanes
    const int lane = tid & 63, wv = tid >> 6, nwv = nt >> 6, newVal = g.invalid16;
    const int per = (W + 63) / 64;
    for (int rr = wv; rr < RB; rr += nwv) {
        const int y = y0 + rr;
        if (y >= H) break;
        const int16_t* const row = med + rr * W;
        const int xa = lane * per, xb = min(xa + per, W);
        int last = -1;
        for (int x = xa; x < xb; x++) {
            const int v = row[x];
            if (v != newVal && (x == 0 || row[x - 1] == newVal || abs(v - row[x - 1]) > maxDiff)) last = x;
        }
        int run = last;                                        // inclusive prefix maximum over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(run, o, 64);
            if (lane >= o) run = max(run, t);
        }
        int cur = __shfl_up(run, 1, 64);
        if (lane == 0) cur = -1;
        for (int x = xa; x < xb; x++) {
            const int v = row[x];
            const size_t i = (size_t)y * W + x;
            if (v == newVal) { L[i] = -1; continue; }
            const bool start = x == 0 || row[x - 1] == newVal || abs(v - row[x - 1]) > maxDiff;
            if (start) { cur = x; size[i] = 0; }
            L[i] = y * W + cur;
            const bool end = x == W - 1 || row[x + 1] == newVal || abs(v - row[x + 1]) > maxDiff;
            if (end) runlen[(size_t)y * W + cur] = x - cur + 1;
        }
    }
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
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    for (int y = blockIdx.y; y + 1 < H; y += gridDim.y) {
    const int i = y * W + x;
    const int v = img[i], u = img[i + W];
    if (v == newVal || u == newVal || abs(v - u) > maxDiff) continue;
    const int a = L[i], b = L[i + W];
    bool head_a = x == 0, head_b = x == 0, joined_left = false;
    if (x > 0) {
        const int v1 = img[i - 1], u1 = img[i + W - 1];
        head_a = v1 == newVal || abs(v - v1) > maxDiff;
        head_b = u1 == newVal || abs(u - u1) > maxDiff;
        joined_left = v1 != newVal && u1 != newVal && abs(v1 - u1) <= maxDiff && !head_a && !head_b;   // same two runs, one column earlier
    }
    if (joined_left) continue;
    // a head pixel's label may already point at an ancestor: its own index is the run head
    const int ha = head_a ? i : a, hb = head_b ? i + W : b;
    const int la = ((volatile int*)runlen)[ha], lb = ((volatile int*)runlen)[hb];
    const bool big_a = (la & ~RUN_TOUCH) > maxSize, big_b = (lb & ~RUN_TOUCH) > maxSize;
    if (big_a && big_b) continue;
    if (big_a != big_b) {
        const int hs = big_a ? hb : ha, ls = big_a ? lb : la;
        if (!(ls & RUN_TOUCH)) atomicOr(&runlen[hs], RUN_TOUCH);
        continue;
    }
    uf_union(L, a, b);
    }
}

// component sizes: every run head (recognised geometrically -- after unions a head's label may
// point elsewhere) adds its run length to the component root
__global__ void k_ccl_sizes(const int16_t* __restrict__ img, int W, int H, int newVal, int maxDiff, int maxSize,
                            int* __restrict__ L, const int* __restrict__ runlen, int* __restrict__ size)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
    const int i = y * W + x;
    const int v = img[i];
    if (v == newVal) continue;
    const bool start = x == 0 || img[i - 1] == newVal || abs(v - img[i - 1]) > maxDiff;
    if (!start) continue;
    const int root = uf_find_halve(L, i);
    L[i] = root;   // flatten: k_ccl_apply then needs two hops (pixel -> run head -> root)
    // only "size <= maxSize" is ever asked, and the counter only grows: once it is past the limit
    // further adds are pointless (this removes the contention on the few huge components)
    if (((volatile int*)size)[root] > maxSize) continue;
    const int len = runlen[i];
    atomicAdd(&size[root], (len & ~RUN_TOUCH) + ((len & RUN_TOUCH) ? maxSize + 1 : 0));
    }
}

__global__ void k_ccl_apply(int16_t* __restrict__ img, int n, int newVal, int maxSize, const int* __restrict__ L,
                            const int* __restrict__ size)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int r = L[i];
        if (r >= 0 && size[uf_find(L, r)] <= maxSize) img[i] = (int16_t)newVal;
    }
}

// ---------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------
static PathPlan make_plan(const SgbmGeom& g, int mode)
{
    // steps = negated predecessor offsets.  MODE_SGBM predecessors: (x-1,y) (x+1,y) (x-1,y-1) (x,y-1)
    // (x+1,y-1); MODE_HH adds (x-1,y+1) (x,y+1) (x+1,y+1).  Longest lines first.
    static const int sx5[] = { 1, -1, 1, -1, 0 }, sy5[] = { 0, 0, 1, 1, 1 };
    static const int sx8[] = { 1, -1, 1, -1, 1, -1, 0, 0 }, sy8[] = { 0, 0, 1, 1, -1, -1, -1, 1 };
    PathPlan p;
    p.n_dirs = mode == 1 ? 8 : 5;
    p.first_wave[0] = 0;
    for (int k = 0; k < p.n_dirs; k++) {
        p.sx[k] = mode == 1 ? sx8[k] : sx5[k];
        p.sy[k] = mode == 1 ? sy8[k] : sy5[k];
        p.nlines[k] = p.sy[k] == 0 ? g.H : (p.sx[k] == 0 ? g.W1 : g.W1 + g.H - 1);
        p.first_wave[k + 1] = p.first_wave[k] + div_up(p.nlines[k], 4);
    }
    for (int k = p.n_dirs; k < VO_MAX_DIRS; k++) { p.sx[k] = p.sy[k] = p.nlines[k] = 0; p.first_wave[k + 1] = p.first_wave[p.n_dirs]; }
    return p;
}

// the current workspace's S holds at least `vols` volumes (grown once; the default covers the fused schedule)
static int ensure_S(vo_ctx* ctx, int vols)
{
    if (ctx->ws->S_vols >= vols) return VO_OK;
    if (ctx->ws->S) (void)hipFree(ctx->ws->S);
    ctx->ws->S = nullptr; ctx->ws->S_vols = 0;
    VO_HIP(ctx, hipMalloc((void**)&ctx->ws->S, ctx->vol_cells * sizeof(int16_t) * vols + 256));
    ctx->ws->S_vols = vols;
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
    // boundary granules of this workspace: [strip][H + 1 rows][NP][48 lanes] x 8 bytes, allocated (and cleared: tags start at
    // 1, no stale granule may match) when first needed or outgrown
    const size_t need = (size_t)jobs.nstrips * (g.H + 1) * DG_GLANES * NP * sizeof(uint64_t);
    if (need > ctx->ws->sw_bnd_bytes) {
        if (ctx->ws->sw_bnd) (void)hipFree(ctx->ws->sw_bnd);
        ctx->ws->sw_bnd = nullptr; ctx->ws->sw_bnd_bytes = 0;
        VO_HIP(ctx, hipMalloc((void**)&ctx->ws->sw_bnd, need + 256));
        ctx->ws->sw_bnd_bytes = need;
        VO_HIP(ctx, hipMemsetAsync(ctx->ws->sw_bnd, 0, need, ctx->stream));
    }
    DiagJob& j = jobs.j[0];
    j.C = ctx->ws->C; j.in1 = in1; j.sout = sout; j.bnd = ctx->ws->sw_bnd; j.aux0 = ctx->ws->rec; j.aux1 = ctx->ws->rec + (size_t)ctx->max_w * ctx->max_h + 64;
    j.tag = ++ctx->ws->sw_tag;
    if (j.tag == 0) j.tag = ++ctx->ws->sw_tag;
    j.sink = ctx->max_w * ctx->max_h;          // 64 spare words behind each of the two record arrays
    j.dbg = ctx->tune_diag_dbg;
    j.spin_limit = ctx->tune_spin_limit;
    j.vol_bytes = (uint32_t)(((size_t)g.W1 * g.H + 1) * g.Dp * 2);            // one volume + the dummy row behind C
    j.rec_bytes = (uint32_t)(((size_t)ctx->max_w * ctx->max_h + 64) * 4);
    if (ctx->fault_sweep > 0 && --ctx->fault_sweep == 0) {   // (only the test-hooks build ever sets it: this launch's strips export
        j.dbg |= 2;                                          //  nothing, so every import misses and gives up after a few polls)
        j.spin_limit = 64;
    }
    const size_t lds = (size_t)DG_RING * 2 * CW * g.Dp * 2 + (WTA ? (size_t)NWC * 4 * 2 * g.Dp * 2 : 0) + 64 * 4;
    auto kern = k_sgbm_diag<NP, PAD, NWC, REV, WTA>;
    static unsigned long long attr_set = 0;     // per instantiation and device: allow more than 64 KB of dynamic LDS
    if (!((attr_set >> (ctx->device & 63)) & 1ull)) {
        VO_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set |= 1ull << (ctx->device & 63);
    }
    // strips that can be active at the same time: one image row's worth (+ slack for the hand-over between strips)
    const int wgs = ctx->tune_diag_wgs > 0 ? min(jobs.nstrips, ctx->tune_diag_wgs) : min(jobs.nstrips, div_up(g.W1, UW) + 2);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3((NWC + 1) * 64), lds, ctx->stream, jobs, g, ctl, ctx->dump, ctx->sw_ctl_words / 2);
    VO_CHECK_LAUNCH(ctx);
    return VO_OK;
}

template <int NP, bool REV, bool WTA>
static int launch_diag(vo_ctx* ctx, const SgbmGeom& g, const int16_t* in1, int16_t* sout, int* ctl)
{
    const bool pad = g.D != g.Dp;
    if constexpr (NP <= 4) {
        // Strip width for up to 128 disparities.  7 compute waves (28 columns, 43 workgroups of 8 waves at config 2) is the shortest
        // chain: 0.76 ms when the launch has the GPU to itself.  Wider strips take longer alone (11 waves: 1.02 ms, 15: 1.33 ms) but
        // pack the sweep: a workgroup of 12 or 16 waves with 94 .. 157 KB of LDS has its CU to itself and fills its SIMDs with sweep
        // waves, so the other pairs' W + E and cost kernels get CUs of their own instead of sharing SIMDs with half-idle sweep
        // waves.  Steady rate with 7 / 9 / 10 / 11 / 12 / 13 / 15 waves: 2292 / 2381 / 2466 / 2479 / 2436 / 2345 / 2366 pairs/s
        // (DESIGN 4b).  So: the narrow strips for a synchronous call on the main stream (one pair, latency), 11-wave strips for
        // pairs streamed through the look-ahead engines (throughput).  A workspace thereby keeps one strip width -- a change of
        // width re-clears its boundary granules -- except the main one, which engine 0 shares with the synchronous calls.
        const int nwc = ctx->tune_diag_nwc ? ctx->tune_diag_nwc : (ctx->cur_engine >= 0 ? 11 : 7);
        if (nwc == 15)
            return pad ? launch_diag_k<NP, true, 15, REV, WTA>(ctx, g, in1, sout, ctl) : launch_diag_k<NP, false, 15, REV, WTA>(ctx, g, in1, sout, ctl);
        if (nwc == 11)
            return pad ? launch_diag_k<NP, true, 11, REV, WTA>(ctx, g, in1, sout, ctl) : launch_diag_k<NP, false, 11, REV, WTA>(ctx, g, in1, sout, ctl);
    }

    return pad ? launch_diag_k<NP, true, 7, REV, WTA>(ctx, g, in1, sout, ctl) : launch_diag_k<NP, false, 7, REV, WTA>(ctx, g, in1, sout, ctl);
}

// Aggregation + winner-take-all of one pair.  The schedule follows from the parameters alone (nothing per pair, no
// context member is written):
//   uniquenessRatio < 100 (the reference's own setting is 10):
//     1. W + E as ONE stored volume: k_sgbm_we (rows in 8-column segments; width a multiple of 8), otherwise k_sgbm_pair
//        (any width: steps past a line's end are computed and discarded)
//     2. MODE_HH: a reverse diagonal sweep adds SW / S / SE to that volume
//     3. the forward diagonal sweep: NW / N / NE + the volume -> winner-take-all records; k_sgbm_fin turns them into
//        disp1 + the disp2 candidates
//     volume passes over HBM: C written 1, W+E 3.25, (MODE_HH reverse 3,) forward 2 = 6.25 / 9.25
//   uniquenessRatio >= 100 (the threshold form of the uniqueness test does not exist): every direction stored by
//     k_sgbm_paths, per-pixel winner search over all of them (k_sgbm_wta)
// Layout of S (fused): [0] = L_W + L_E, [1] = MODE_HH: [0] + the three bottom-up directions, [2] = checkpoints of step 1.
template <int NP>
static int launch_agg(vo_ctx* ctx, const SgbmGeom& g, const PathPlan& plan, size_t vol)
{
    const bool pad = g.D != g.Dp;
    const bool hh = plan.n_dirs == 8;
    int rc;
    if (g.ur < 100) {
        if ((rc = ensure_S(ctx, 3))) return rc;
        int16_t* const Swe = ctx->ws->S;
        int16_t* const Srev = ctx->ws->S + vol;
        int16_t* const ck = ctx->ws->S + 2 * vol;
        int* const ctlA = ctx->ws->sw_ctl;
        int* const ctlB = ctx->ws->sw_ctl + ctx->sw_ctl_words / 2;
        {
            StageTimer t(ctx, VO_T_SGBM_AGG);
            if (ctx->tune_diag_dbg & 8) {
            } else if (g.W1 % 8 == 0 && g.W1 >= 16) {
                const int nw = div_up(g.H, 4);
                if (g.W1 % 16 == 0 && g.W1 >= 32) {               // a row cut at its middle: two waves per four rows, half the chain each
                    if (pad) hipLaunchKernelGGL((k_sgbm_we2<NP, true>), dim3(div_up(nw, 2)), dim3(256), 0, ctx->stream, ctx->ws->C, Swe, ck, g, ctx->dump);
                    else hipLaunchKernelGGL((k_sgbm_we2<NP, false>), dim3(div_up(nw, 2)), dim3(256), 0, ctx->stream, ctx->ws->C, Swe, ck, g, ctx->dump);
                } else if (pad) hipLaunchKernelGGL((k_sgbm_we<NP, true>), dim3(div_up(nw, 4)), dim3(256), 0, ctx->stream, ctx->ws->C, Swe, ck, g, ctx->dump);
                else hipLaunchKernelGGL((k_sgbm_we<NP, false>), dim3(div_up(nw, 4)), dim3(256), 0, ctx->stream, ctx->ws->C, Swe, ck, g, ctx->dump);
            } else {
                PathPlan pp = plan;                                  // the W / E pair alone: one line per image row
                pp.n_dirs = 1;
                pp.sx[0] = 1; pp.sy[0] = 0; pp.nlines[0] = g.H;
                pp.first_wave[0] = 0;
                for (int k = 0; k < VO_MAX_DIRS; k++) pp.first_wave[k + 1] = div_up(g.H, 4);
                for (int k = 1; k < VO_MAX_DIRS; k++) pp.sx[k] = pp.sy[k] = pp.nlines[k] = 0;
                const int nwp = pp.first_wave[1];
                if (pad) hipLaunchKernelGGL((k_sgbm_pair<NP, true>), dim3(div_up(nwp, 4)), dim3(256), 0, ctx->stream, ctx->ws->C, Swe, ck, vol, g, pp, ctx->dump);
                else hipLaunchKernelGGL((k_sgbm_pair<NP, false>), dim3(div_up(nwp, 4)), dim3(256), 0, ctx->stream, ctx->ws->C, Swe, ck, vol, g, pp, ctx->dump);
            }
            VO_CHECK_LAUNCH(ctx);
            if (ctx->cur_engine >= 0 && ctx->ws_alt[ctx->cur_engine].mid) {
                vo_ctx::SgbmWs& a = ctx->ws_alt[ctx->cur_engine];
                if (hipEventRecord(a.mid, ctx->stream) == hipSuccess) a.mid_valid = true;
            }
            if (hh && (rc = launch_diag<NP, true, false>(ctx, g, Swe, Srev, ctlA))) return rc;
        }
        {
            StageTimer t(ctx, VO_T_SGBM_WTA);
            if (!(ctx->tune_diag_dbg & 16) && (rc = launch_diag<NP, false, true>(ctx, g, hh ? Srev : Swe, nullptr, hh ? ctlB : ctlA))) return rc;
        }
        ctx->last_paths = 3;
        ctx->last_schedule = (g.W1 % 8 == 0 && g.W1 >= 16) ? VO_SCHED_DIAG : VO_SCHED_DIAG_RAGGED;
        return VO_OK;
    }
    if ((rc = ensure_S(ctx, plan.n_dirs))) return rc;
    {
        StageTimer t(ctx, VO_T_SGBM_AGG);
        const int nwaves = plan.first_wave[plan.n_dirs];
        constexpr int PF = NP <= 4 ? 8 : 4;
        if (pad) hipLaunchKernelGGL((k_sgbm_paths<NP, PF, true>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->ws->C, ctx->ws->S, vol, g, plan, ctx->dump);
        else hipLaunchKernelGGL((k_sgbm_paths<NP, PF, false>), dim3(div_up(nwaves, 4)), dim3(256), 0, ctx->stream, ctx->ws->C, ctx->ws->S, vol, g, plan, ctx->dump);
        VO_CHECK_LAUNCH(ctx);
    }
    {
        StageTimer t(ctx, VO_T_SGBM_WTA);
        const size_t sh = (size_t)16 * g.Dp * sizeof(int16_t);   // one row of S per 16-lane group
        const long long npix = (long long)g.W1 * g.H;
        hipLaunchKernelGGL((k_sgbm_wta<NP>), dim3((unsigned)((npix + 15) / 16)), dim3(256), sh, ctx->stream, ctx->ws->S, vol, plan.n_dirs,
                           g, ctx->ws->disp_tmp, ctx->ws->ccl_size);
        VO_CHECK_LAUNCH(ctx);
    }
    ctx->last_paths = plan.n_dirs;
    ctx->last_schedule = VO_SCHED_UNFUSED;
    return VO_OK;
}

static int sgbm_run_impl(vo_ctx* ctx, FrameSlot& f, int w, int h, const uint8_t* srcL, const uint8_t* srcR);
// rows per block of k_sgbm_post_rows: (RB + 2) rows of disp1 (2 bytes per pixel) and disp2 keys (4 bytes) must fit in LDS
static uint32_t pk_rep_host(int v) { return (uint32_t)(v & 0xFFFF) * 0x00010001u; }
static int post_rows_per_block(int w) { return std::min(6, (int)(150 * 1024 / ((size_t)w * 6)) - 2); }

// The MAIN SGBM workspace (planes, C, S volumes, CCL arrays) is shared by the main stream and look-ahead engine 0's: a run on
// one of them must not start before the previous run -- possibly on the other -- has finished.  An event chain orders them on
// the device without blocking the host.  An engine's own workspace is only ever used on that engine's stream, which is ordered
// already: no wait, no record (two packets less per pair on the engine's queue).
int sgbm_run(vo_ctx* ctx, FrameSlot& f, int w, int h, const uint8_t* srcL, const uint8_t* srcR)
{
    const bool shared = ctx->ws == &ctx->main_ws;
    if (shared && ctx->ws->done_valid) VO_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ws->done, 0));
    if (++ctx->sweep_gen_next <= 0) ctx->sweep_gen_next = 1;     // this run's generation: never 0 (FrameSlot::sweep_word)
    f.disp_gen = ctx->sweep_gen_next;
    int rc = sgbm_run_impl(ctx, f, w, h, srcL, srcR);
    if (shared && ctx->ws->done) {
        VO_HIP(ctx, hipEventRecord(ctx->ws->done, ctx->stream));
        ctx->ws->done_valid = true;
    }
    return rc;
}

static int sgbm_run_impl(vo_ctx* ctx, FrameSlot& f, int w, int h, const uint8_t* srcL, const uint8_t* srcR)
{
    const uint8_t* const dL = srcL ? srcL : f.left;
    const uint8_t* const dR = srcL ? srcR : f.right;
    int16_t* const d_disp = f.disp16;
    const SgbmEff& e = ctx->sg;
    if (!e.set) return vo_fail(ctx, VO_E_STATE, "vo_set_sgbm has not been called");
    SgbmGeom g;
    g.W = w; g.H = h; g.D = e.D; g.Dp = (e.D + 31) & ~31; g.minD = e.minD; g.P1 = e.P1; g.P2 = e.P2; g.ur = e.ur; g.d12 = e.d12;
    g.ftzero = e.ftzero; g.SW2 = e.SW2;
    {
        const int u = 100 - e.ur;                // (only used when 1 <= u <= 100: the threshold form of the uniqueness test)
        g.urM = u > 1 ? (uint32_t)((0x100000000ull + (uint64_t)u - 1) / (uint64_t)u) : 0xFFFFFFFFu;
        g.urAdd = u > 1 ? (uint32_t)(u - 1) : 1u;
    }
    g.minX1 = e.maxD > 0 ? e.maxD : 0;
    int maxX1 = w + (e.minD < 0 ? e.minD : 0);
    g.W1 = maxX1 - g.minX1;
    g.invalid16 = (e.minD - 1) * 16;
    const int n = w * h;
    if (g.W1 <= 0) {
        // (in-place ingest: the kernel that would have left the slot's copy of the pair behind does not run on this path)
        if (srcL) {
            VO_HIP(ctx, hipMemcpyAsync(f.left, srcL, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
            VO_HIP(ctx, hipMemcpyAsync(f.right, srcR, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        }
        std::vector<int16_t> inv((size_t)n, (int16_t)g.invalid16);  // every pixel invalid
        VO_HIP(ctx, hipMemcpyAsync(d_disp, inv.data(), (size_t)n * 2, hipMemcpyHostToDevice, ctx->stream));
        VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->last_cells = 0;
        return VO_OK;
    }
    const size_t vol = (size_t)g.W1 * h * g.Dp;
    if (vol > ctx->vol_cells || e.D > 256)
        return vo_fail(ctx, VO_E_CAP, "cost volume %zu cells exceeds the capacity given to vo_create (or D > 256)", vol);
    // the diagonal sweep addresses a volume through one buffer descriptor: 32-bit byte offsets, the top 256 reserved as the
    // out-of-range value (3840 x 2160 with D = 256 is 3.96 GB and fits; 4096 x 2304 with D = 256 does not)
    if (((size_t)g.W1 * h + 1) * g.Dp * 2 >= 0xFFFFFF00ull)
        return vo_fail(ctx, VO_E_CAP, "cost volume of %zu bytes exceeds the 4 GiB a volume kernel addresses", ((size_t)g.W1 * h + 1) * g.Dp * 2);
    const PathPlan plan = make_plan(g, e.mode);
    ctx->last_cells = (int64_t)g.W1 * h * g.D;
    {
        StageTimer t(ctx, VO_T_SGBM_COST);
        const int dbg = ctx->tune_diag_dbg;          // development only (VO_DIAG_DEBUG): 4 / 8 / 16 / 32 skip the cost / W+E / diagonal / post stage
        hipLaunchKernelGGL(k_sgbm_planes, dim3(div_up(w, 256), std::min(h, 128)), dim3(256), 0, ctx->stream, dL, dR, w, h, g.ftzero,
                           ctx->ws->planesL, ctx->ws->planesR, (g.ur >= 100 || post_rows_per_block(w) < 1) ? ctx->ws->ccl_size : nullptr, ctx->ws->sw_ctl, ctx->sw_ctl_words,
                           (uint32_t*)(ctx->ws->C + vol), g.Dp / 2, pk_rep_host(g.P2), srcL ? f.left : nullptr, srcL ? f.right : nullptr);
        const int bx = ((g.Dp / 2 + 63) / 64) * 64;
        const int TY = ctx->tune_sweep_ty;
        const int nw = bx / 64;
#define LAUNCH_SWEEP(XT, SW)                                                                                                   \
    hipLaunchKernelGGL((k_sgbm_cost_sweep<XT, SW>), dim3(8 * div_up(div_up(g.W1, XT) * div_up(h, TY), 8)), dim3(bx),            \
                       (size_t)nw * (((2 * SW + 1) <= 5 ? 0 : (2 * SW + 1) * XT * 64) + 12 * (64 + (XT + 2 * SW - 1) / 2)) * 4, ctx->stream, ctx->ws->planesL, ctx->ws->planesR, g, TY, ctx->ws->C)
        if (dbg & 4) {
        } else
        switch (g.SW2) {
            case 0: LAUNCH_SWEEP(8, 0); break;
            case 1: LAUNCH_SWEEP(8, 1); break;
            case 2: LAUNCH_SWEEP(8, 2); break;
            case 3: LAUNCH_SWEEP(8, 3); break;
            case 4: LAUNCH_SWEEP(8, 4); break;
            case 5: LAUNCH_SWEEP(8, 5); break;
            default:
                hipLaunchKernelGGL(k_sgbm_cost_generic, dim3(g.W1, h), dim3(bx), 0, ctx->stream, ctx->ws->planesL, ctx->ws->planesR, g, ctx->ws->C);
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
    if (!(ctx->tune_diag_dbg & 32)) {
        StageTimer t(ctx, VO_T_SGBM_POST);
        const int newVal = g.invalid16, maxDiff = 16 * e.speckleRange;
        const bool speckle = e.speckleWindow > 0;
        // rows per block of the fused kernel: (RB + 2) rows of disp1 (2 bytes) and disp2 keys (4 bytes) must fit in LDS
        const int rb = post_rows_per_block(w);
        if (ctx->last_schedule != VO_SCHED_UNFUSED && rb >= 1) {
            // records of the diagonal sweep -> sub-pixel disp1 + disp2 -> left-right check -> medianBlur(3) -> labelled runs: one launch
            static unsigned long long attr_set = 0;
            if (!((attr_set >> (ctx->device & 63)) & 1ull)) {
                VO_HIP(ctx, hipFuncSetAttribute((const void*)k_sgbm_post_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_set |= 1ull << (ctx->device & 63);
            }
            const int* const ctlA = ctx->ws->sw_ctl;
            const int* const ctlB = ctx->ws->sw_ctl + ctx->sw_ctl_words / 2;
            hipLaunchKernelGGL(k_sgbm_post_rows, dim3(div_up(h, rb)), dim3(512), (size_t)(rb + 2) * w * 6, ctx->stream, ctx->ws->rec,
                               ctx->ws->rec + (size_t)ctx->max_w * ctx->max_h + 64, g, rb, d_disp, speckle ? 1 : 0, maxDiff, ctx->ws->ccl_label,
                               ctx->ws->ccl_runlen, ctx->ws->ccl_size, ctlA, ctlB, f.sweep_word, f.disp_gen, ctx->d_sweep_errs);
        } else {
            if (ctx->last_schedule != VO_SCHED_UNFUSED) {
                // (an image too wide for the fused kernel's LDS rows: the same steps as separate passes over HBM)
                hipLaunchKernelGGL(k_sgbm_fin, dim3(div_up(g.W1, 256), g.H), dim3(256), 0, ctx->stream, ctx->ws->rec, ctx->ws->rec + (size_t)ctx->max_w * ctx->max_h + 64,
                                   g, ctx->ws->disp_tmp, ctx->ws->ccl_size, ctx->ws->sw_ctl, ctx->ws->sw_ctl + ctx->sw_ctl_words / 2, f.sweep_word, f.disp_gen, ctx->d_sweep_errs);
            }
            hipLaunchKernelGGL(k_lr_median3, dim3(div_up(w, 256), h), dim3(256), 0, ctx->stream, ctx->ws->disp_tmp, ctx->ws->ccl_size, g, d_disp);
            if (speckle)
                hipLaunchKernelGGL(k_ccl_rows, dim3(h), dim3(256), 0, ctx->stream, d_disp, w, newVal, maxDiff, ctx->ws->ccl_label, ctx->ws->ccl_runlen, ctx->ws->ccl_size);
        }
        if (speckle) {
            hipLaunchKernelGGL(k_ccl_vmerge, dim3(div_up(w, 256), std::min(h, 128)), dim3(256), 0, ctx->stream, d_disp, w, h, newVal, maxDiff, e.speckleWindow, ctx->ws->ccl_label, ctx->ws->ccl_runlen);
            hipLaunchKernelGGL(k_ccl_sizes, dim3(div_up(w, 256), std::min(h, 128)), dim3(256), 0, ctx->stream, d_disp, w, h, newVal, maxDiff, e.speckleWindow, ctx->ws->ccl_label, ctx->ws->ccl_runlen, ctx->ws->ccl_size);
            hipLaunchKernelGGL(k_ccl_apply, dim3(std::min(div_up(n, 256), 1024)), dim3(256), 0, ctx->stream, d_disp, n, newVal, e.speckleWindow, ctx->ws->ccl_label, ctx->ws->ccl_size);
        }
        VO_CHECK_LAUNCH(ctx);
    }
    return VO_OK;
}
