// Internal declarations of libvo355 (gfx950 only).  See include/vo355.h for the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include <deque>
#include <thread>
#include <mutex>
#include <condition_variable>
#include "../../include/vo355.h"

#define VO_ORB_LEVELS 8
#define VO_ORB_EDGE 31
#define VO_ORB_HALF_PATCH 15

struct SgbmEff {
    int minD, maxD, D, ur, d12, P1, P2, SW2, SH2, ftzero, minX1, maxX1, W1, invalid16;
    int speckleWindow, speckleRange, mode;
    bool set;
};

struct OrbLevel {
    int w, h;            // level size
    size_t off;          // byte offset of the level inside the pyramid buffers
    float scale;         // 1.2^l as float
    int quota;           // filled per call (depends on nfeatures)
};

struct FrameSlot {
    uint8_t* left = nullptr;    // rectified gray, max_w*max_h
    uint8_t* right = nullptr;
    int16_t* disp16 = nullptr;  // max_w*max_h
    // keypoints (device), capacity kp_cap
    float* kp_xy = nullptr;
    float* kp_size = nullptr;
    float* kp_angle = nullptr;
    float* kp_resp = nullptr;
    int32_t* kp_oct = nullptr;
    uint8_t* desc = nullptr;
    int n_kp = 0;
    int w = 0, h = 0;
    bool has_pair = false, has_disp = false, has_kp = false;
    // look-ahead: the pair was ingested + SGBM'd on the second stream; `ready` orders consumers
    hipEvent_t ready = nullptr;
    bool pending = false;
    bool counted = false;        // part of vo_lookahead_depth (submitted, neither waited for nor dropped)
    // asynchronous pose steps still reading this slot on their own streams (vo_pose_pair_begin): whoever overwrites the slot
    // orders itself behind them (borrowed events: they belong to the pose alternates and live as long as the context)
    hipEvent_t readers[2] = {nullptr, nullptr};
    // look-ahead ORB: keypoints were extracted behind the SGBM on the engine's stream; the count lands
    // in the slot's pinned word once `ready` has fired
    int32_t* n_kp_host = nullptr;
    bool kp_pending = false;
    // health of the disparity in this slot: every SGBM run gets a generation number (disp_gen, never 0); a run whose diagonal
    // sweep gave up a strip hand-off (its disparity is then undefined) writes ITS generation into the slot's pinned word from
    // the device (k_sgbm_fin).  The slot is bad exactly while *sweep_word == disp_gen: a refill gets a new generation, a late
    // write of an older run can never match it.  Checked wherever the host picks up results that depend on the disparity.
    int32_t* sweep_word = nullptr;
    int32_t disp_gen = 0;
    int kp_params[4] = {0, 0, 0, 0};   // nfeatures, mask_mode, min_disp16, max_disp16 of the pending run
};

// scratch of one ORB run (pyramids, candidate lists, counters); one per look-ahead engine
struct OrbWs {
    uint8_t *pyr_img = nullptr, *pyr_mask = nullptr;
    int32_t *cand_pos = nullptr, *candA_pos = nullptr, *candB_pos = nullptr, *counters = nullptr;
    float *cand_resp = nullptr, *candA_resp = nullptr, *candB_resp = nullptr;
    hipEvent_t done = nullptr;   // end of the latest run in this workspace (any stream)
    bool done_valid = false;
};

struct vo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // look-ahead engines (vo_prefetch_*): each has its own stream and staging; engine 0 shares the main
    // SGBM workspace, engines 1.. own an alternate one, so several pairs' SGBM can be in flight (one
    // pair's latency-bound kernels overlap another's bandwidth-bound ones)
    static const int MAX_ENGINES = 24;
    hipStream_t la_stream[MAX_ENGINES] = {};
    int cur_engine = -1;
    // VO_STAGGER = K: a pair's early stages (cost volume, W + E) start only after those of the pair K places before it have finished
    // (an event wait on the engine's stream; default: 7/16 of the engines, 0 = off).  Keeps the pairs in flight spread over the stages
    // -- at most K of them in the early ones -- instead of all in the same one after a cold start: +2.5 % on a 20-pair burst,
    // +1.5 % in the steady state with 16 engines (K = 6..8: 6 and 7 are 1 % ahead of 8 on the burst, equal in the steady state;
    // K <= 4 costs throughput, K >= 12 changes nothing).
    int tune_stagger = -1;
    uint8_t* la_stage[MAX_ENGINES] = {};
    // One SGBM workspace: everything a disparity run writes.  The context owns one (`main_ws`: the main stream and look-ahead
    // engine 0 use it) and one per further engine; `ws` / `orbws` name the workspace and the ORB scratch the code in sgbm.hip /
    // orb.hip works in right now -- the main ones, or an engine's while that engine is being fed (EngineScope).
    struct SgbmWs {
        uint32_t *planesL = nullptr, *planesR = nullptr;
        int16_t *C = nullptr, *S = nullptr, *disp_tmp = nullptr;
        int32_t *ccl_runlen = nullptr, *ccl_label = nullptr, *ccl_size = nullptr;
        int32_t* rec = nullptr;          // the diagonal sweep's winner records: two arrays of max_w * max_h + 64 words
        int S_vols = 0;
        uint64_t* sw_bnd = nullptr;
        size_t sw_bnd_bytes = 0;
        int* sw_ctl = nullptr;
        uint32_t sw_tag = 0;
        hipEvent_t done = nullptr;       // end of the latest SGBM run in this workspace (any stream)
        bool done_valid = false;
        bool ready = false;
        OrbWs orb;
        uint8_t* pinned = nullptr;       // host staging of vo_prefetch_pair (two raw images)
        hipEvent_t h2d_done = nullptr;   // the copies out of `pinned` have finished
        bool h2d_valid = false;
        hipEvent_t mid = nullptr;        // the early stages (cost volume, W + E) of the engine's latest pair have finished
        bool mid_valid = false;
    } ws_alt[MAX_ENGINES];           // [0]: only its ORB scratch / staging / events are used (engine 0 works in main_ws)
    SgbmWs main_ws;
    SgbmWs* ws = &main_ws;
    OrbWs* orbws = &main_ws.orb;
    // ORB behind the look-ahead SGBM (vo_set_lookahead_orb): nfeatures, mask_mode, min/max disp16
    bool la_orb = false;
    int la_orb_params[4] = {0, 0, 0, 0};
    int32_t* slot_words = nullptr;   // pinned: one word per slot (keypoint counts of pending runs), then one per slot for FrameSlot::sweep_word
    int32_t sweep_gen_next = 0;      // generations handed to SGBM runs so far
    int* d_sweep_errs = nullptr;     // device counter: SGBM runs of this context whose sweep gave up a hand-off (vo_sgbm_sweep_status)
    int tune_spin_limit = 1 << 22;   // polls before a wait inside the diagonal sweep is declared dead
    int fault_sweep = 0;             // VO_FAULT_SWEEP=n (VO_TEST_HOOKS builds only): the n-th diagonal sweep exports nothing and gives up after a few polls
    int engines_fit = 24;            // what 40 % of the device's free memory held at vo_create (vo_set_engines clamps to it)
    int n_engines = 16;              // VO_ENGINES (needs GPU_MAX_HW_QUEUES >= engines + 4: streams sharing a hardware queue serialise)
    int next_engine = 0;
    int max_w = 0, max_h = 0, max_disp = 0, max_kp = 0, kp_cap = 0;
    std::string err;
    char devname[256] = {0};

    SgbmEff sg{};
    double Q[16];
    bool has_Q = false;
    int roi[4] = {0, 0, 0, 0};
    bool has_roi = false;

    // rectification maps
    int16_t* map1[2] = {nullptr, nullptr};
    uint16_t* map2[2] = {nullptr, nullptr};
    int map_w = 0, map_h = 0;
    bool has_map[2] = {false, false};

    FrameSlot slots[VO_NUM_SLOTS + 1];  // last slot = scratch for the *_host seams

    // staging
    uint8_t* stage_in = nullptr;   // raw upload (max_w*max_h*3)
    size_t stage_bytes = 0;

    // SGBM workspace: see SgbmWs (ws->planesL: per pixel 2 x u32 (u,u0,u1 for both channels); ws->planesR: per pixel 6 x u32
    // pair-packed (v,v0,v1 x 2 channels); ws->C: cost volume; ws->S: aggregated volumes (S_vols of them); ws->disp_tmp: WTA output
    // before the LR check; ws->sw_bnd: boundary granules of the diagonal sweep (allocated on first use); ws->sw_ctl: its two control
    // blocks {work items taken, sticky error, ..., per-strip timeline}; ws->sw_tag: launches so far in that workspace)
    size_t vol_cells = 0;
    int16_t* dump = nullptr;       // sink for the stores of lanes past the end of their scan line
    int sw_ctl_words = 0;
    int tune_diag_wgs = 0;         // VO_DIAG_WGS (development): workgroups of one diagonal sweep (0 = one image row's worth of strips + 2)
    int tune_diag_dbg = 0;         // VO_DIAG_DEBUG (development): bit 0 / 1 = strips import / export nothing, 4 / 8 / 16 / 32 = skip the cost / W+E / diagonal / post stage
    int tune_diag_nwc = 0;         // VO_DIAG_WAVES: compute waves per strip workgroup (7, 11 or 15) for Dp <= 128; 0 = 7 for synchronous calls, 11 on the look-ahead engines (launch_diag)
    int64_t last_cells = 0;
    int last_paths = 0;
    int last_schedule = 0;         // VO_SCHED_* of the latest run (vo_sgbm_last_schedule)

    // ORB workspace
    OrbLevel lv[VO_ORB_LEVELS];
    int orb_w = 0, orb_h = 0;      // geometry the pyramid tables were built for
    size_t pyr_bytes = 0;
    int32_t* rs_ofs = nullptr;     // resize tables (all levels): x then y offsets
    uint16_t* rs_coef = nullptr;
    // k_orb_pyramid: the pyramid is cut into pyr_nbx x pyr_nby cones (one workgroup each); pyr_rects = per level the column
    // interval every cone column needs (own part + what the next level's interval reads), then the row intervals:
    // [l][bx][2] (inclusive lo, hi) for x, followed by [l][by][2] for y.  pyr_buf[2] = bytes of the largest even- / odd-level
    // rectangle (the kernel's two ping-pong LDS buffers per image kind).
    int32_t* pyr_rects = nullptr;
    int pyr_nbx = 0, pyr_nby = 0, pyr_buf[2] = {0, 0}, pyr_tab = 0;
    char rs_meta_host[1024];       // host copy of the level descriptors (LevelsDev)
    int orb_quota_nfeatures = -1;  // nfeatures the device quotas were uploaded for
    int cand_cap = 0;
    uint8_t* host_mask_dev = nullptr;  // explicit mask upload (scratch)

    // match / pose workspace.  MatchWs = what one matching + pose (or essential-matrix) step writes; the context owns one
    // (`main_mw`) and one per asynchronous alternate, `mw` names the one the code in match.hip / geom.hip / ransac.hip works in
    // right now (PoseScope / MonoScope retarget it together with the stream; nothing is swapped member by member).
    struct MatchWs {
        int32_t *m_idx = nullptr, *m_count = nullptr, *m_dist = nullptr, *mq_idx = nullptr, *mt_idx = nullptr;   // m_count: match counter of the ratio filter
        float *pts_a = nullptr, *pts_b = nullptr, *xy_a = nullptr, *xy_b = nullptr;
        uint8_t *st_a = nullptr, *st_b = nullptr, *clique_ws = nullptr;
        size_t clique_ws_bytes = 0;
        uint8_t* ransac_ws = nullptr;
        size_t ransac_ws_bytes = 0;
    };
    MatchWs main_mw;
    MatchWs* mw = &main_mw;
    uint8_t* mq = nullptr;
    uint8_t* mt = nullptr;
    double* red = nullptr;         // reduction scratch
    // asynchronous pose steps (vo_pose_pair_begin / _end): two alternates of the match / pose scratch above,
    // each with its own stream, a pinned result record and a completion event
    static const int N_POSE_ALT = VO_NUM_POSE_ASYNC;
    // the alternates' steps run on THREE streams (alternate k on stream k % 3): a context must stay at about twenty HIP streams
    // (16 engines + main + these) -- beyond that the hardware queues are time-sliced in ~10 ms quanta and a pair's diagonal
    // sweep stalls behind a descheduled neighbour (measured: 23 streams -> 200-1000 pairs/s).  Steps that share a stream run
    // back to back without the host in between; their scratch and records are separate.
    static const int N_POSE_STREAMS = 3;
    int n_pose_streams = N_POSE_STREAMS;     // VO_POSE_STREAMS (1..3): fewer when the GPU's hardware queues are shared with other processes
    hipStream_t pose_streams[N_POSE_STREAMS] = {};
    struct PoseAlt {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        void* result = nullptr;        // pinned PoseOut
        bool ready = false, busy = false;
        int slot_a = -1, slot_b = -1;
        int32_t gen_a = 0, gen_b = 0;      // disparity generations of the two slots when the step was begun (slot health at _end)
        double params[4] = {0, 0, 0, 0};   // ratio, min_matches, rigidity_thr, outlier_thr
        MatchWs mw;
    } pose_alt[N_POSE_ALT];
    int pose_next = 0;
    // asynchronous monocular pair steps (vo_mono_pair_begin / _end): match scratch + RANSAC workspace + stream + pinned record each
    static const int N_MONO_ALT = VO_NUM_MONO_ASYNC;
    struct MonoAlt {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        uint8_t* result = nullptr;     // pinned: header (M, best, E) + mask / q / t / xy of the second frame
        size_t result_bytes = 0;
        bool ready = false, busy = false, want = false;
        int nq = 0, nb = 0, min_n = 0;
        MatchWs mw;
    } mono_alt[N_MONO_ALT];
    int mono_next = 0;
    float* img3_ws = nullptr;
    size_t img3_ws_bytes = 0;
    void* pinned = nullptr;        // pinned host buffer: first 4 KB scalar readbacks, rest = transfer arena
    size_t pinned_bytes = 0;
    size_t arena_off = 0;          // bump pointer into the arena (reset by xfer_flush)
    struct PendingCopy { void* dst; const void* src; size_t bytes; };
    std::vector<PendingCopy> pending;  // device->host copies staged in the arena, scattered at flush

    // pinned staging buffers for host images filled AHEAD by a helper thread of the caller (vo_host_stage_pair) and consumed by
    // vo_prefetch_host_staged on the thread that drives the context: the launching thread does no memcpy
    static const int N_HOST_STAGE = VO_NUM_HOST_STAGE;
    // `state` (under stage_mu): 0 = idle or filled, 1 = a copy into the buffer is queued or running on the library's staging
    // thread, < 0 = that copy failed (VO_E_*).  `valid` / `h2d_done` are written by the driving thread (the upload out of
    // the buffer has been enqueued) and read by the staging thread before it overwrites the buffer: under stage_mu too.
    struct HostStage { uint8_t* pinned = nullptr; hipEvent_t h2d_done = nullptr; bool valid = false; int state = 0; } host_stage[VO_NUM_HOST_STAGE];
    // vo_host_stage_begin: the copy of a host pair into pinned memory runs on ONE thread owned by the library (started on first
    // use, joined by vo_destroy), so that a driving thread written in an interpreted language never shares its interpreter
    // lock with a copying thread
    struct StageJob { int buf; const uint8_t *left, *right; size_t per; };
    std::thread stage_thread;
    std::mutex stage_mu;
    std::condition_variable stage_cv;
    std::deque<StageJob> stage_jobs;
    bool stage_stop = false;

    // inputs staged in HBM
    uint8_t* staged = nullptr;
    int staged_n = 0, staged_w = 0, staged_h = 0, staged_ch = 1;

    // tuning knobs (environment), read once in vo_create and never written afterwards
    int tune_sweep_ty = 30;         // VO_SWEEP_TY: rows per tile of the cost sweep
    int mono_engine = 0;            // round robin of vo_prefetch_staged_mono
    int inflight = 0;               // look-ahead pairs submitted and not yet waited for (vo_lookahead_depth)
    int fault_prefetch = 0;         // VO_FAULT_PREFETCH=n (VO_TEST_HOOKS builds only): the n-th look-ahead submission fails inside its engine scope

    // timing
    bool timing = false;
    unsigned timing_mask = ~0u;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_stage;
    size_t ev_used = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double t_ms[VO_T_NSTAGES] = {0};
    int64_t t_n[VO_T_NSTAGES] = {0};
};

int vo_fail(vo_ctx* ctx, int code, const char* fmt, ...);
#define VO_HIP(ctx, call)                                                               \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess)                                                          \
            return vo_fail(ctx, VO_E_HIP, "%s failed: %s (%s:%d)", #call,              \
                           hipGetErrorString(e__), __FILE__, __LINE__);                 \
    } while (0)

#define VO_CHECK_LAUNCH(ctx) VO_HIP(ctx, hipGetLastError())

// hipEvent pair around a stage, recorded on the context stream WITHOUT blocking it; the pairs are
// resolved in vo_get_timings.
struct StageTimer {
    vo_ctx* c;
    int stage;
    long idx;
    StageTimer(vo_ctx* ctx, int s);
    ~StageTimer();
};

static inline int div_up(int a, int b) { return (a + b - 1) / b; }

// Small host<->device transfers go through the pinned arena: asynchronous copies from / to
// pageable memory make the runtime pin and unpin the pages on every call (hundreds of
// microseconds each).  xfer_d2h defers the final host memcpy to xfer_flush (which synchronises).
int xfer_h2d(vo_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int xfer_d2h(vo_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
int xfer_flush(vo_ctx* ctx);

// make the main stream wait for a slot whose look-ahead work may still be running
int slot_wait(vo_ctx* ctx, FrameSlot& f);
// the stream the context currently works on is about to overwrite the slot: order it behind work that still writes or reads it
int slot_before_overwrite(vo_ctx* ctx, FrameSlot& f);
// `done` marks the end of an asynchronous step that reads the slot: whoever overwrites the slot waits for it first
void slot_add_reader(FrameSlot& f, hipEvent_t done);
int orb_slot_enqueue(vo_ctx* ctx, FrameSlot& f, int nfeatures, int mask_mode, int min_disp16, int max_disp16);
void pose_alt_free(vo_ctx* ctx);
void mono_alt_free(vo_ctx* ctx);
size_t pose_ws_bytes(int nq);
__global__ void k_ratio_compact(const int32_t* idx, const int32_t* dist, int nq, double ratio, const float* xy_q, const float* xy_t,
                                int32_t* q_out, int32_t* t_out, float* xyq_out, float* xyt_out, int32_t* m_out);   // pose / clique scratch for nq query keypoints

// implemented in the per-stage files
// f.left, f.right (w x h) -> f.disp16; gives the run its generation.  srcL / srcR (both or neither): the rectified gray pair lies
// THERE in device memory and f.left / f.right still have to receive their copy (done by the run's first kernel)
int sgbm_run(vo_ctx* ctx, FrameSlot& f, int w, int h, const uint8_t* srcL = nullptr, const uint8_t* srcR = nullptr);
// VO_E_SWEEP when the disparity the slot holds comes from a run whose sweep gave up a hand-off.  Only meaningful once the host
// has waited for work that depends on that run (a stream or event synchronisation).
int slot_health(vo_ctx* ctx, const FrameSlot& f, int slot);
int orb_prepare_tables(vo_ctx* ctx, int w, int h);
int orb_run(vo_ctx* ctx, FrameSlot* fs, const uint8_t* d_img, int img_stride, int w, int h,
            int nfeatures, int mask_mode, const int16_t* d_disp16, int disp_stride, int min_d16,
            int max_d16, const uint8_t* d_mask, int mask_stride);
// the kNN kernel's scratch lives behind the distances in ONE allocation (MatchWs::m_dist): 2 distances per query, then up to
// VO_KNN_SPLITS partial (best, second) pairs per query, then one ticket word per 64 queries (zero between launches)
#define VO_KNN_SPLITS 16
static inline size_t match_dist_bytes(int kp_cap)
{
    const size_t capq = ((size_t)kp_cap + 63) & ~(size_t)63;
    return capq * 8 + (size_t)VO_KNN_SPLITS * capq * 8 + (capq / 64 + 1) * 4 + 256;
}
int match_dist_alloc(vo_ctx* ctx, int32_t** p);      // hipMalloc + the tickets cleared
int match_knn2(vo_ctx* ctx, const uint8_t* dq, int nq, const uint8_t* dt, int nt, int32_t* d_idx,
               int32_t* d_dist);
int points3d_launch(vo_ctx* ctx, const int16_t* d_disp16, int w, int h, const float* d_xy, int n,
                    float* d_xyz, uint8_t* d_status);
void host_svd3(const double* A, double* U, double* w, double* Vt);
