// Context, configuration, image front-end (cvtColor / remap), lazy downloads.
// Replaces the cv2 calls of StereoCamera.compute_3d [reference stereo_camera.py:43-55].
#include <stdarg.h>
#include <stdlib.h>
#include "vo_internal.h"

int vo_fail(vo_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

static thread_local std::string g_create_err;

// ---- pinned transfer arena ---------------------------------------------------------------
static const size_t ARENA_BASE = 4096;

static void* arena_take(vo_ctx* ctx, size_t bytes)
{
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (ARENA_BASE + ctx->arena_off + need > ctx->pinned_bytes) return nullptr;
    void* p = (char*)ctx->pinned + ARENA_BASE + ctx->arena_off;
    ctx->arena_off += need;
    return p;
}

int xfer_flush(vo_ctx* ctx)
{
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (const auto& c : ctx->pending) memcpy(c.dst, c.src, c.bytes);
    ctx->pending.clear();
    ctx->arena_off = 0;
    return VO_OK;
}

int xfer_h2d(vo_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes)
{
    if (bytes == 0) return VO_OK;
    void* st = arena_take(ctx, bytes);
    if (!st && bytes + 64 <= ctx->pinned_bytes - ARENA_BASE) {
        int rc = xfer_flush(ctx);  // arena full: drain what is in flight, then reuse it
        if (rc) return rc;
        st = arena_take(ctx, bytes);
    }
    if (!st) {  // larger than the arena: plain copy
        VO_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return VO_OK;
    }
    memcpy(st, host_src, bytes);
    VO_HIP(ctx, hipMemcpyAsync(dev_dst, st, bytes, hipMemcpyHostToDevice, ctx->stream));
    return VO_OK;
}

int xfer_d2h(vo_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes)
{
    if (bytes == 0) return VO_OK;
    void* st = arena_take(ctx, bytes);
    if (!st && bytes + 64 <= ctx->pinned_bytes - ARENA_BASE) {
        int rc = xfer_flush(ctx);
        if (rc) return rc;
        st = arena_take(ctx, bytes);
    }
    if (!st) {
        VO_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return VO_OK;
    }
    VO_HIP(ctx, hipMemcpyAsync(st, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ctx->pending.push_back({host_dst, st, bytes});
    return VO_OK;
}

template <typename T>
static int dalloc(vo_ctx* ctx, T** p, size_t count)
{
    VO_HIP(ctx, hipMalloc((void**)p, count * sizeof(T) + 256));
    return VO_OK;
}
#define DALLOC(p, n)                         \
    do {                                     \
        int rc__ = dalloc(ctx, &(p), (n));   \
        if (rc__) { g_create_err = ctx->err; vo_destroy(ctx); return rc__; } \
    } while (0)

extern "C" const char* vo_last_error(const vo_ctx* ctx)
{
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

static int orb_ws_alloc(vo_ctx* ctx, OrbWs& o)
{
    const size_t cand = (size_t)ctx->cand_cap * 4 * sizeof(int32_t);   // 8 levels together: < 3.2x level 0
    void** ps[] = { (void**)&o.pyr_img, (void**)&o.pyr_mask,
                    (void**)&o.cand_pos, (void**)&o.cand_resp, (void**)&o.candA_pos, (void**)&o.candA_resp, (void**)&o.candB_pos,
                    (void**)&o.candB_resp, (void**)&o.counters };
    const size_t sz[] = { ctx->pyr_bytes, ctx->pyr_bytes, cand, cand, cand, cand, cand,
                          cand, 8192 * 4 };
    for (size_t k = 0; k < sizeof(ps) / sizeof(ps[0]); k++)
        if (hipMalloc(ps[k], sz[k] + 256) != hipSuccess) return VO_E_HIP;
    return VO_OK;
}

static void orb_ws_free(OrbWs& o)
{
    void* ps[] = { o.pyr_img, o.pyr_mask, o.cand_pos, o.cand_resp, o.candA_pos, o.candA_resp,
                   o.candB_pos, o.candB_resp, o.counters };
    for (void* p : ps) if (p) (void)hipFree(p);
    o = OrbWs();
}

extern "C" int vo_create(int device_id, int max_w, int max_h, int max_disp, int max_kp, vo_ctx** out)
{
    if (!out) return VO_E_ARG;
    *out = nullptr;
    if (max_w < 64 || max_h < 64 || max_disp < 16 || max_disp > 256 || max_disp % 16 || max_kp < 16) {
        g_create_err = "vo_create: need max_w,max_h >= 64, 16 <= max_disp <= 256 (multiple of 16), max_kp >= 16";
        return VO_E_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_create_err = std::string("vo_create: no HIP device available (") + hipGetErrorString(e) +
                       "); libvo355 has no CPU fallback";
        return VO_E_HIP;
    }
    if (device_id < 0 || device_id >= ndev) { g_create_err = "vo_create: bad device id"; return VO_E_ARG; }
    vo_ctx* ctx = new vo_ctx();
    ctx->device = device_id;
    ctx->max_w = max_w; ctx->max_h = max_h; ctx->max_disp = max_disp; ctx->max_kp = max_kp;
    if ((e = hipSetDevice(device_id)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete ctx; return VO_E_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        // (some driver stacks report no marketing name: say what the device is by its architecture and size then)
        char generic[96];
        snprintf(generic, sizeof(generic), "AMD GPU, %d CUs, %.0f GB", prop.multiProcessorCount, (double)prop.totalGlobalMem / 1e9);
        snprintf(ctx->devname, sizeof(ctx->devname), "%s (%s)", prop.name[0] ? prop.name : generic, prop.gcnArchName);
    }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) { g_create_err = hipGetErrorString(e); delete ctx; return VO_E_HIP; }
    (void)hipEventCreateWithFlags(&ctx->ws->done, hipEventDisableTiming);
    (void)hipEventCreate(&ctx->ev0);
    (void)hipEventCreate(&ctx->ev1);

    const size_t npx = (size_t)max_w * max_h;
    // keypoint capacity: OpenCV keeps response ties, so allow head-room over nfeatures
    ctx->kp_cap = max_kp * 2 + 1024;
    for (int s = 0; s <= VO_NUM_SLOTS; s++) {
        FrameSlot& f = ctx->slots[s];
        DALLOC(f.left, npx); DALLOC(f.right, npx); DALLOC(f.disp16, npx);
        (void)hipEventCreateWithFlags(&f.ready, hipEventDisableTiming);
        DALLOC(f.kp_xy, (size_t)ctx->kp_cap * 2); DALLOC(f.kp_size, ctx->kp_cap); DALLOC(f.kp_angle, ctx->kp_cap);
        DALLOC(f.kp_resp, ctx->kp_cap); DALLOC(f.kp_oct, ctx->kp_cap); DALLOC(f.desc, (size_t)ctx->kp_cap * 32);
    }
    ctx->stage_bytes = npx * 3;
    DALLOC(ctx->stage_in, ctx->stage_bytes * 2);
    for (int c = 0; c < 2; c++) { DALLOC(ctx->map1[c], npx * 2); DALLOC(ctx->map2[c], npx); }
    DALLOC(ctx->ws->planesL, npx * 2); DALLOC(ctx->ws->planesR, npx * 6);
    ctx->vol_cells = npx * (size_t)((max_disp + 31) & ~31);
    DALLOC(ctx->ws->C, ctx->vol_cells);
    // aggregated volumes: L_W + L_E, MODE_HH's reverse-pass sum, checkpoints (grown on demand for uniquenessRatio >= 100)
    ctx->ws->S_vols = 3;
    DALLOC(ctx->ws->S, ctx->vol_cells * ctx->ws->S_vols);
    ctx->sw_ctl_words = 2 * 2048;                        // two control blocks: {work items taken, sticky error, ...} + per-strip timeline
    DALLOC(ctx->ws->sw_ctl, ctx->sw_ctl_words);
    VO_HIP(ctx, hipMemset(ctx->ws->sw_ctl, 0, ctx->sw_ctl_words * sizeof(int)));
    DALLOC(ctx->ws->disp_tmp, npx); DALLOC(ctx->dump, 4096);
    DALLOC(ctx->ws->ccl_label, npx); DALLOC(ctx->ws->ccl_size, npx); DALLOC(ctx->ws->ccl_runlen, npx);
    DALLOC(ctx->ws->rec, 2 * (npx + 64));
    // ORB: 8-level pyramid is < 3.2x the base image
    ctx->pyr_bytes = npx * 4;
    DALLOC(ctx->rs_ofs, (size_t)(max_w + max_h) * 2 * VO_ORB_LEVELS);
    DALLOC(ctx->rs_coef, (size_t)(max_w + max_h) * 4 * VO_ORB_LEVELS);
    DALLOC(ctx->pyr_rects, 4096);
    // FAST candidates after NMS are never 8-adjacent: at most ceil(w/2)*ceil(h/2) per level
    ctx->cand_cap = (int)((size_t)((max_w + 1) / 2) * ((max_h + 1) / 2));
    // (summed over the 8 levels: < 3.2x that)
    if (orb_ws_alloc(ctx, (*ctx->orbws))) { g_create_err = "hipMalloc failed (ORB workspace)"; vo_destroy(ctx); return VO_E_HIP; }
    DALLOC(ctx->mw->m_count, 64);
    DALLOC(ctx->host_mask_dev, npx);
    DALLOC(ctx->mq, (size_t)ctx->kp_cap * 32); DALLOC(ctx->mt, (size_t)ctx->kp_cap * 32);
    DALLOC(ctx->mw->m_idx, (size_t)ctx->kp_cap * 2);
    if (match_dist_alloc(ctx, &ctx->mw->m_dist)) { g_create_err = "hipMalloc failed (match scratch)"; vo_destroy(ctx); return VO_E_HIP; }
    DALLOC(ctx->mw->pts_a, (size_t)ctx->kp_cap * 3); DALLOC(ctx->mw->pts_b, (size_t)ctx->kp_cap * 3);
    DALLOC(ctx->mw->st_a, ctx->kp_cap); DALLOC(ctx->mw->st_b, ctx->kp_cap);
    DALLOC(ctx->mw->xy_a, (size_t)ctx->kp_cap * 2); DALLOC(ctx->mw->xy_b, (size_t)ctx->kp_cap * 2);
    DALLOC(ctx->mw->mq_idx, ctx->kp_cap); DALLOC(ctx->mw->mt_idx, ctx->kp_cap);
    DALLOC(ctx->red, 4096);
    ctx->mw->clique_ws_bytes = pose_ws_bytes(ctx->kp_cap);     // sized once: the pose step never reallocates mid-stream
    DALLOC(ctx->mw->clique_ws, ctx->mw->clique_ws_bytes);
    ctx->pinned_bytes = 8 << 20;
    if (hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&ctx->slot_words, 128 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
        g_create_err = "hipHostMalloc failed"; vo_destroy(ctx); return VO_E_HIP;
    }
    memset(ctx->slot_words, 0, 128 * sizeof(int32_t));
    for (int s = 0; s <= VO_NUM_SLOTS; s++) { ctx->slots[s].n_kp_host = ctx->slot_words + s; ctx->slots[s].sweep_word = ctx->slot_words + 64 + s; }
    DALLOC(ctx->d_sweep_errs, 64);
    VO_HIP(ctx, hipMemset(ctx->d_sweep_errs, 0, 64 * sizeof(int)));
    if (const char* e8 = getenv("VO_ENGINES")) { int v = atoi(e8); if (v >= 1 && v <= vo_ctx::MAX_ENGINES) ctx->n_engines = v; }
    {
        // every engine owns a full SGBM workspace: cost volume + 3 aggregated volumes + boundary granules (< 1 volume) + planes.
        // Keep all of them within 40 % of what the device has free right now (frame slots, ORB scratch and the caller's staged
        // inputs need room too).
        const double per_engine = (double)ctx->vol_cells * 2.0 * 5.0 + (double)npx * 64.0;
        size_t free_b = 0, total_b = 0;
        double budget = 96e9;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = 0.4 * (double)free_b;
        const int fit = (int)(budget / per_engine);
        ctx->engines_fit = fit < 2 ? 2 : fit;
        if (ctx->n_engines > ctx->engines_fit) ctx->n_engines = ctx->engines_fit;
    }
    if (const char* e9 = getenv("VO_POSE_STREAMS")) { int v = atoi(e9); if (v >= 1 && v <= vo_ctx::N_POSE_STREAMS) ctx->n_pose_streams = v; }
    if (const char* e27 = getenv("VO_STAGGER")) ctx->tune_stagger = atoi(e27);
    if (const char* e25 = getenv("VO_DIAG_WAVES")) { const int v = atoi(e25); ctx->tune_diag_nwc = (v == 7 || v == 11 || v == 15) ? v : 0; }   // the three strip widths that are built
#ifdef VO_TEST_HOOKS
    // development switches live in the test-only build alone (libvo355_hooks.so; tools/stage_ablation.py loads it): some of them
    // switch stages or the strip hand-off OFF, and nothing in a user's environment may make the product compute garbage
    if (const char* e27 = getenv("VO_DIAG_WGS")) ctx->tune_diag_wgs = atoi(e27);
    if (const char* e26 = getenv("VO_DIAG_DEBUG")) ctx->tune_diag_dbg = atoi(e26);
    if (const char* e10 = getenv("VO_FAULT_PREFETCH")) ctx->fault_prefetch = atoi(e10);   // test-only build (libvo355_hooks.so)
    if (const char* e11 = getenv("VO_FAULT_SWEEP")) ctx->fault_sweep = atoi(e11);
#endif
    if (const char* e6 = getenv("VO_SWEEP_TY")) { int v = atoi(e6); if (v >= 4 && v <= 4096) ctx->tune_sweep_ty = v; }
    *out = ctx;
    return VO_OK;
}

extern "C" void vo_destroy(vo_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stage_thread.joinable()) {
        { std::lock_guard<std::mutex> lk(ctx->stage_mu); ctx->stage_stop = true; }
        ctx->stage_cv.notify_all();
        ctx->stage_thread.join();
    }
    for (int k = 0; k < vo_ctx::MAX_ENGINES; k++) if (ctx->la_stream[k]) (void)hipStreamSynchronize(ctx->la_stream[k]);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int s = 0; s <= VO_NUM_SLOTS; s++) {
        FrameSlot& f = ctx->slots[s];
        void* ps[] = { f.left, f.right, f.disp16, f.kp_xy, f.kp_size, f.kp_angle, f.kp_resp, f.kp_oct, f.desc };
        for (void* p : ps) if (p) (void)hipFree(p);
        if (f.ready) (void)hipEventDestroy(f.ready);
    }
    void* ps[] = { ctx->stage_in, ctx->map1[0], ctx->map1[1], ctx->map2[0], ctx->map2[1], ctx->ws->planesL, ctx->ws->planesR,
                   ctx->ws->C, ctx->ws->S, ctx->ws->sw_bnd, ctx->ws->sw_ctl, ctx->ws->disp_tmp, ctx->dump, ctx->ws->ccl_runlen, ctx->ws->ccl_label, ctx->ws->ccl_size, ctx->ws->rec, ctx->rs_ofs, ctx->rs_coef, ctx->pyr_rects, ctx->mw->m_count, ctx->host_mask_dev, ctx->mq, ctx->mt,
                   ctx->mw->m_idx, ctx->mw->m_dist, ctx->mw->pts_a, ctx->mw->pts_b, ctx->mw->st_a, ctx->mw->st_b, ctx->mw->xy_a, ctx->mw->xy_b,
                   ctx->mw->mq_idx, ctx->mw->mt_idx, ctx->red, ctx->mw->clique_ws, ctx->img3_ws, ctx->mw->ransac_ws, ctx->d_sweep_errs };
    for (void* p : ps) if (p) (void)hipFree(p);
    orb_ws_free(*ctx->orbws);
    pose_alt_free(ctx);
    mono_alt_free(ctx);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->slot_words) (void)hipHostFree(ctx->slot_words);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->staged) (void)hipFree(ctx->staged);
    if (ctx->ws->done) (void)hipEventDestroy(ctx->ws->done);
    for (int k = 0; k < vo_ctx::MAX_ENGINES; k++) {
        vo_ctx::SgbmWs& a = ctx->ws_alt[k];
        void* pa[] = { a.planesL, a.planesR, a.C, a.S, a.sw_bnd, a.sw_ctl, a.disp_tmp, a.ccl_runlen, a.ccl_label, a.ccl_size, a.rec, ctx->la_stage[k] };
        for (void* q : pa) if (q) (void)hipFree(q);
        if (a.done) (void)hipEventDestroy(a.done);
        orb_ws_free(a.orb);
        if (a.pinned) (void)hipHostFree(a.pinned);
        if (a.h2d_done) (void)hipEventDestroy(a.h2d_done);
        if (a.mid) (void)hipEventDestroy(a.mid);
        if (ctx->la_stream[k]) (void)hipStreamDestroy(ctx->la_stream[k]);
    }
    for (vo_ctx::HostStage& hs : ctx->host_stage) {
        if (hs.pinned) (void)hipHostFree(hs.pinned);
        if (hs.h2d_done) (void)hipEventDestroy(hs.h2d_done);
    }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int vo_set_engines(vo_ctx* ctx, int n)
{
    if (!ctx) return VO_E_ARG;
    if (n > 0) {
        if (n > vo_ctx::MAX_ENGINES) n = vo_ctx::MAX_ENGINES;
        if (n > ctx->engines_fit) n = ctx->engines_fit;
        ctx->n_engines = n;
        ctx->next_engine %= n;          // (the stagger wait and the monocular span index modulo n_engines: nothing else remembers one)
    }
    return ctx->n_engines;
}

extern "C" int vo_device_name(const vo_ctx* ctx, char* buf, int buflen)
{
    if (!ctx || !buf || buflen <= 0) return VO_E_ARG;
    snprintf(buf, buflen, "%s", ctx->devname);
    return VO_OK;
}

extern "C" int vo_synchronize(vo_ctx* ctx)
{
    if (!ctx) return VO_E_ARG;
    for (int k = 0; k < vo_ctx::MAX_ENGINES; k++)
        if (ctx->la_stream[k]) VO_HIP(ctx, hipStreamSynchronize(ctx->la_stream[k]));
    for (int k = 0; k < vo_ctx::N_POSE_ALT; k++)
        if (ctx->pose_alt[k].stream) VO_HIP(ctx, hipStreamSynchronize(ctx->pose_alt[k].stream));
    for (int k = 0; k < vo_ctx::N_MONO_ALT; k++)
        if (ctx->mono_alt[k].stream) VO_HIP(ctx, hipStreamSynchronize(ctx->mono_alt[k].stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_set_rectify_maps(vo_ctx* ctx, int cam, const int16_t* map1, const uint16_t* map2, int w, int h)
{
    if (!ctx || cam < 0 || cam > 1 || !map1 || !map2) return vo_fail(ctx, VO_E_ARG, "vo_set_rectify_maps: bad argument");
    if (w > ctx->max_w || h > ctx->max_h || w <= 0 || h <= 0) return vo_fail(ctx, VO_E_CAP, "maps %dx%d exceed context %dx%d", w, h, ctx->max_w, ctx->max_h);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    VO_HIP(ctx, hipMemcpyAsync(ctx->map1[cam], map1, (size_t)w * h * 4, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(ctx, hipMemcpyAsync(ctx->map2[cam], map2, (size_t)w * h * 2, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->map_w = w; ctx->map_h = h; ctx->has_map[cam] = true;
    return VO_OK;
}

extern "C" int vo_set_sgbm(vo_ctx* ctx, int minDisparity, int numDisparities, int blockSize, int P1, int P2,
                           int disp12MaxDiff, int preFilterCap, int uniquenessRatio, int speckleWindowSize,
                           int speckleRange, int mode)
{
    if (!ctx) return VO_E_ARG;
    if (numDisparities <= 0 || numDisparities % 16) return vo_fail(ctx, VO_E_ARG, "numDisparities must be a positive multiple of 16");
    if (numDisparities > ctx->max_disp) return vo_fail(ctx, VO_E_CAP, "numDisparities %d > max_disp %d", numDisparities, ctx->max_disp);
    if (mode != 0 && mode != 1) return vo_fail(ctx, VO_E_ARG, "mode must be 0 (MODE_SGBM) or 1 (MODE_HH)");
    // effective parameters exactly as computeDisparitySGBM derives them
    SgbmEff& e = ctx->sg;
    e.minD = minDisparity; e.D = numDisparities; e.maxD = minDisparity + numDisparities;
    e.ur = uniquenessRatio >= 0 ? uniquenessRatio : 10;
    e.d12 = disp12MaxDiff > 0 ? disp12MaxDiff : 1;
    e.P1 = P1 > 0 ? P1 : 2;
    int p2 = P2 > 0 ? P2 : 5;
    e.P2 = p2 > e.P1 + 1 ? p2 : e.P1 + 1;
    int bs = blockSize > 0 ? blockSize : 5;
    e.SW2 = e.SH2 = bs / 2;
    e.ftzero = (preFilterCap > 15 ? preFilterCap : 15) | 1;
    e.speckleWindow = speckleWindowSize; e.speckleRange = speckleRange; e.mode = mode;
    if (e.SW2 > 5) return vo_fail(ctx, VO_E_ARG, "blockSize > 11 is not supported");
    if (e.P2 > 8000 || e.ftzero > 127) return vo_fail(ctx, VO_E_ARG, "P2 > 8000 or preFilterCap > 127 would overflow int16 costs");
    // a path cost is at most the block's largest matching cost + P2; OpenCV keeps it in a short and wraps beyond 32767 (its
    // results are then garbage): refused here rather than reproduced
    {
        const int side = 2 * e.SW2 + 1, cmax = side * side * (2 * e.ftzero + 63);
        if (cmax + e.P2 > 32767)
            return vo_fail(ctx, VO_E_ARG, "blockSize %d with preFilterCap %d and P2 %d lets a path cost reach %d: beyond the int16 OpenCV computes in",
                           side, e.ftzero, e.P2, cmax + e.P2);
    }
    e.set = true;
    return VO_OK;
}

extern "C" int vo_set_Q(vo_ctx* ctx, const double* Q16)
{
    if (!ctx || !Q16) return VO_E_ARG;
    memcpy(ctx->Q, Q16, sizeof(ctx->Q));
    ctx->has_Q = true;
    return VO_OK;
}

extern "C" int vo_set_roi(vo_ctx* ctx, int x0, int y0, int x1, int y1)
{
    if (!ctx) return VO_E_ARG;
    if (x0 < 0 || y0 < 0) return vo_fail(ctx, VO_E_ARG, "negative ROI origin is not supported");
    ctx->roi[0] = x0; ctx->roi[1] = y0; ctx->roi[2] = x1; ctx->roi[3] = y1;
    ctx->has_roi = true;
    return VO_OK;
}

// ---- cvtColor BGR2GRAY (OpenCV 4.x RGB2Gray<uchar>, 15-bit coefficients) -------------
__global__ void k_bgr2gray(const uint8_t* __restrict__ bgr, int n, uint8_t* __restrict__ gray)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = bgr[3 * (size_t)i], g = bgr[3 * (size_t)i + 1], r = bgr[3 * (size_t)i + 2];
    gray[i] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
}

// ---- remap INTER_LINEAR with CV_16SC2 + fractional maps, border constant 0 ------------
__global__ void k_remap(const uint8_t* __restrict__ src, int sw, int sh, const int16_t* __restrict__ map1,
                        const uint16_t* __restrict__ map2, int w, int h, uint8_t* __restrict__ dst)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    size_t i = (size_t)y * w + x;
    int sx = map1[2 * i], sy = map1[2 * i + 1];
    int f = map2[i] & 1023, fx = f & 31, fy = f >> 5;
    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    int p00 = 0, p01 = 0, p10 = 0, p11 = 0;
    bool x0ok = (unsigned)sx < (unsigned)sw, x1ok = (unsigned)(sx + 1) < (unsigned)sw;
    if ((unsigned)sy < (unsigned)sh) {
        if (x0ok) p00 = src[(size_t)sy * sw + sx];
        if (x1ok) p01 = src[(size_t)sy * sw + sx + 1];
    }
    if ((unsigned)(sy + 1) < (unsigned)sh) {
        if (x0ok) p10 = src[(size_t)(sy + 1) * sw + sx];
        if (x1ok) p11 = src[(size_t)(sy + 1) * sw + sx + 1];
    }
    int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
    dst[i] = (uint8_t)min(v, 255);
}

static int check_slot(vo_ctx* ctx, int slot)
{
    if (!ctx) return VO_E_ARG;
    if (slot < 0 || slot >= VO_NUM_SLOTS) return vo_fail(ctx, VO_E_ARG, "slot %d out of range [0,%d)", slot, VO_NUM_SLOTS);
    return VO_OK;
}

// upload one camera image into dst (gray, rectified)
// Image bytes out of pinned host memory by a kernel (the GPU reads them over the host link itself).  An asynchronous
// hipMemcpy from pinned memory goes to a DMA engine instead, and every hand-over between that engine's queue and the compute
// queue of the look-ahead engine costs the stream tens of microseconds -- with a dozen engines uploading, the from-host rate
// fell to 0.6-0.8 of the resident rate.
typedef uint32_t ingest_u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_ingest_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t bytes)
{
    const size_t n16 = bytes / 16, stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = i0; i < n16; i += stride)
        ((ingest_u32x4*)dst)[i] = __builtin_nontemporal_load((const ingest_u32x4*)src + i);
    for (size_t i = n16 * 16 + i0; i < bytes; i += stride) dst[i] = src[i];
}

static int ingest_bytes(vo_ctx* ctx, uint8_t* dst, const uint8_t* src, size_t bytes, hipMemcpyKind kind, bool by_kernel)
{
    if (by_kernel && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0)) {
        const int blocks = (int)((bytes / 16 + 255) / 256 < 2048 ? (bytes / 16 + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_ingest_copy, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, ctx->stream, src, dst, bytes);
        return VO_OK;
    }
    VO_HIP(ctx, hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
    return VO_OK;
}

static int ingest(vo_ctx* ctx, int cam, const uint8_t* host, int w, int h, int channels, int preprocessed, uint8_t* dst,
                  uint8_t* stage, hipMemcpyKind kind = hipMemcpyHostToDevice, bool by_kernel = false)
{
    const size_t n = (size_t)w * h;
    const bool need_remap = !preprocessed;
    if (need_remap && (!ctx->has_map[cam] || ctx->map_w != w || ctx->map_h != h))
        return vo_fail(ctx, VO_E_STATE, "rectification maps for camera %d not set for %dx%d", cam, w, h);
    uint8_t* gray = need_remap ? stage + n * 3 : dst;  // gray staging behind the colour staging
    if (channels == 3) {
        if (int rcb = ingest_bytes(ctx, stage, host, n * 3, kind, by_kernel)) return rcb;
        hipLaunchKernelGGL(k_bgr2gray, dim3(div_up((int)n, 256)), dim3(256), 0, ctx->stream, stage, (int)n, gray);
    } else {
        if (int rcb = ingest_bytes(ctx, gray, host, n, kind, by_kernel)) return rcb;
    }
    if (need_remap)
        hipLaunchKernelGGL(k_remap, dim3(div_up(w, 256), h), dim3(256), 0, ctx->stream, gray, w, h, ctx->map1[cam],
                           ctx->map2[cam], w, h, dst);
    VO_CHECK_LAUNCH(ctx);
    return VO_OK;
}

extern "C" int vo_upload_pair(vo_ctx* ctx, int slot, const uint8_t* left, const uint8_t* right, int w, int h,
                              int channels, int preprocessed)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!left || !right || (channels != 1 && channels != 3)) return vo_fail(ctx, VO_E_ARG, "vo_upload_pair: bad argument");
    if (w > ctx->max_w || h > ctx->max_h || w < 16 || h < 16) return vo_fail(ctx, VO_E_CAP, "image %dx%d exceeds context %dx%d", w, h, ctx->max_w, ctx->max_h);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    FrameSlot& f = ctx->slots[slot];
    if ((rc = slot_wait(ctx, f)) || (rc = slot_before_overwrite(ctx, f))) return rc;
    StageTimer t(ctx, VO_T_UPLOAD);
    // the two cameras use disjoint halves of the staging buffer (4 bytes/pixel each would be
    // needed for colour + gray; stage_in holds 6 bytes/pixel)
    rc = ingest(ctx, 0, left, w, h, channels, preprocessed, f.left, ctx->stage_in);
    if (rc) return rc;
    // the second ingest reuses the staging area: order on the stream makes that safe
    rc = ingest(ctx, 1, right, w, h, channels, preprocessed, f.right, ctx->stage_in);
    if (rc) return rc;
    f.w = w; f.h = h; f.has_pair = true; f.has_disp = false; f.has_kp = false; f.n_kp = 0; f.kp_pending = false;
    return VO_OK;
}

// one image into a slot (monocular front end, BASELINE config 5): same ingest as the left image of a pair
extern "C" int vo_upload_mono(vo_ctx* ctx, int slot, const uint8_t* img, int w, int h, int channels)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!img || (channels != 1 && channels != 3)) return vo_fail(ctx, VO_E_ARG, "vo_upload_mono: bad argument");
    if (w > ctx->max_w || h > ctx->max_h || w < 16 || h < 16) return vo_fail(ctx, VO_E_CAP, "image %dx%d exceeds context %dx%d", w, h, ctx->max_w, ctx->max_h);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    FrameSlot& f = ctx->slots[slot];
    if ((rc = slot_wait(ctx, f)) || (rc = slot_before_overwrite(ctx, f))) return rc;
    StageTimer t(ctx, VO_T_UPLOAD);
    if ((rc = ingest(ctx, 0, img, w, h, channels, 1, f.left, ctx->stage_in))) return rc;
    f.w = w; f.h = h; f.has_pair = true; f.has_disp = false; f.has_kp = false; f.n_kp = 0; f.kp_pending = false;
    return VO_OK;
}

// ---- inputs kept resident in HBM (streaming ingest / benchmarking) -----------------------
extern "C" int vo_stage_pairs_alloc(vo_ctx* ctx, int n, int w, int h, int channels)
{
    if (!ctx || n <= 0 || (channels != 1 && channels != 3)) return vo_fail(ctx, VO_E_ARG, "vo_stage_pairs_alloc: bad argument");
    if (w > ctx->max_w || h > ctx->max_h || w < 16 || h < 16) return vo_fail(ctx, VO_E_CAP, "image %dx%d exceeds context", w, h);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->staged) { (void)hipFree(ctx->staged); ctx->staged = nullptr; }
    ctx->staged_n = 0;
    const size_t per = (size_t)w * h * channels;
    VO_HIP(ctx, hipMalloc((void**)&ctx->staged, per * 2 * n));
    ctx->staged_n = n; ctx->staged_w = w; ctx->staged_h = h; ctx->staged_ch = channels;
    return VO_OK;
}

extern "C" int vo_stage_pair(vo_ctx* ctx, int index, const uint8_t* left, const uint8_t* right)
{
    if (!ctx || !left || !right || index < 0 || index >= ctx->staged_n) return vo_fail(ctx, VO_E_ARG, "vo_stage_pair: bad index");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t per = (size_t)ctx->staged_w * ctx->staged_h * ctx->staged_ch;
    VO_HIP(ctx, hipMemcpyAsync(ctx->staged + per * 2 * index, left, per, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(ctx, hipMemcpyAsync(ctx->staged + per * (2 * index + 1), right, per, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_load_staged_pair(vo_ctx* ctx, int slot, int index, int preprocessed)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (index < 0 || index >= ctx->staged_n) return vo_fail(ctx, VO_E_ARG, "vo_load_staged_pair: bad index");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    FrameSlot& f = ctx->slots[slot];
    const int w = ctx->staged_w, h = ctx->staged_h;
    const size_t per = (size_t)w * h * ctx->staged_ch;
    StageTimer t(ctx, VO_T_UPLOAD);
    if ((rc = slot_wait(ctx, f)) || (rc = slot_before_overwrite(ctx, f))) return rc;
    rc = ingest(ctx, 0, ctx->staged + per * 2 * index, w, h, ctx->staged_ch, preprocessed, f.left, ctx->stage_in, hipMemcpyDeviceToDevice);
    if (rc) return rc;
    rc = ingest(ctx, 1, ctx->staged + per * (2 * index + 1), w, h, ctx->staged_ch, preprocessed, f.right, ctx->stage_in, hipMemcpyDeviceToDevice);
    if (rc) return rc;
    f.w = w; f.h = h; f.has_pair = true; f.has_disp = false; f.has_kp = false; f.n_kp = 0; f.kp_pending = false;
    return VO_OK;
}

// Retargets the context at a look-ahead engine for the lifetime of the object: its stream and staging buffer change places
// with the main ones, `ws` / `orbws` point at the engine's SGBM workspace (engine 0: the main one) and ORB scratch.  No member
// of any workspace is copied or swapped.  Whatever path leaves the scope -- a status funnelled through rc or an early return of
// a VO_HIP check added later -- the context comes back pointing at its own.
struct EngineScope {
    vo_ctx* ctx;
    int engine;
    EngineScope(vo_ctx* c, int e) : ctx(c), engine(e)
    {
        std::swap(ctx->stream, ctx->la_stream[engine]);
        std::swap(ctx->stage_in, ctx->la_stage[engine]);
        ctx->ws = engine == 0 ? &ctx->main_ws : &ctx->ws_alt[engine];
        ctx->orbws = &ctx->ws_alt[engine].orb;
        ctx->cur_engine = engine;
    }
    ~EngineScope()
    {
        ctx->cur_engine = -1;
        ctx->ws = &ctx->main_ws;
        ctx->orbws = &ctx->main_ws.orb;
        std::swap(ctx->stage_in, ctx->la_stage[engine]);
        std::swap(ctx->stream, ctx->la_stream[engine]);
    }
    EngineScope(const EngineScope&) = delete;
    EngineScope& operator=(const EngineScope&) = delete;
};

static int engine_prepare(vo_ctx* ctx, int engine)
{
    if (!ctx->la_stream[engine]) {
        VO_HIP(ctx, hipStreamCreateWithFlags(&ctx->la_stream[engine], hipStreamNonBlocking));
        VO_HIP(ctx, hipEventCreateWithFlags(&ctx->ws_alt[engine].mid, hipEventDisableTiming));
        VO_HIP(ctx, hipMalloc((void**)&ctx->la_stage[engine], ctx->stage_bytes * 2 + 256));
        if (orb_ws_alloc(ctx, ctx->ws_alt[engine].orb)) return vo_fail(ctx, VO_E_HIP, "hipMalloc failed (look-ahead ORB workspace)");
    }
    if (engine == 0 || ctx->ws_alt[engine].ready) return VO_OK;
    vo_ctx::SgbmWs& a = ctx->ws_alt[engine];
    const size_t npx = (size_t)ctx->max_w * ctx->max_h;
    const int vols = 3;                          // L_W + L_E, MODE_HH's reverse-pass sum, checkpoints (ensure_S grows it if ever needed)
    void** ps[] = { (void**)&a.planesL, (void**)&a.planesR, (void**)&a.C, (void**)&a.S, (void**)&a.sw_ctl, (void**)&a.disp_tmp,
                    (void**)&a.ccl_runlen, (void**)&a.ccl_label, (void**)&a.ccl_size, (void**)&a.rec };
    const size_t sz[] = { npx * 2 * 4, npx * 6 * 4, ctx->vol_cells * 2, ctx->vol_cells * 2 * vols, ctx->sw_ctl_words * sizeof(int), npx * 2,
                          npx * 4, npx * 4, npx * 4, (npx + 64) * 8 };
    hipError_t e = hipSuccess;
    for (size_t k = 0; k < sizeof(ps) / sizeof(ps[0]) && e == hipSuccess; k++) e = hipMalloc(ps[k], sz[k] + 256);
    if (e == hipSuccess) e = hipMemset(a.sw_ctl, 0, ctx->sw_ctl_words * sizeof(int));
    if (e == hipSuccess && !a.done) e = hipEventCreateWithFlags(&a.done, hipEventDisableTiming);
    if (e != hipSuccess) {
        // a partly built workspace is given back whole: the next call starts from nothing instead of leaking these
        for (void** q : ps) { if (*q) (void)hipFree(*q); *q = nullptr; }
        return vo_fail(ctx, VO_E_HIP, "look-ahead engine %d: workspace allocation failed: %s", engine, hipGetErrorString(e));
    }
    a.S_vols = vols;
    a.ready = true;
    return VO_OK;
}

int slot_wait(vo_ctx* ctx, FrameSlot& f)
{
    if (f.pending) {
        VO_HIP(ctx, hipStreamWaitEvent(ctx->stream, f.ready, 0));
        f.pending = false;
        if (f.counted && ctx->inflight > 0) ctx->inflight--;
        f.counted = false;
    }
    return VO_OK;
}

// the caller gives up a look-ahead slot without consuming it (a prediction that did not come true): its work may still be
// running -- the slot stays `pending` for ordering -- but it no longer counts as in flight
extern "C" int vo_lookahead_drop(vo_ctx* ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= VO_NUM_SLOTS) return vo_fail(ctx, VO_E_ARG, "vo_lookahead_drop: bad slot");
    FrameSlot& f = ctx->slots[slot];
    if (f.pending && f.counted && ctx->inflight > 0) ctx->inflight--;
    f.counted = false;
    return VO_OK;
}

int slot_before_overwrite(vo_ctx* ctx, FrameSlot& f)
{
    // an earlier look-ahead run into this slot that nobody waited for (a voided prediction) may still be writing it on
    // another engine's stream, and pose steps started ahead may still be reading it on theirs
    if (f.pending) VO_HIP(ctx, hipStreamWaitEvent(ctx->stream, f.ready, 0));
    for (hipEvent_t& r : f.readers)
        if (r) { VO_HIP(ctx, hipStreamWaitEvent(ctx->stream, r, 0)); r = nullptr; }
    return VO_OK;
}

// Registers the completion event of an asynchronous step that reads the slot.  The slot keeps two entries: an entry is free
// when it is empty or its event has completed; with both still open (a step begun on a slot while two earlier readers of
// it are unfinished) the older one is waited for here, on the host, rather than dropped from the set a refill orders
// itself behind.
void slot_add_reader(FrameSlot& f, hipEvent_t done)
{
    for (hipEvent_t r : f.readers) if (r == done) return;
    for (hipEvent_t& r : f.readers)
        if (!r || hipEventQuery(r) == hipSuccess) { r = done; return; }
    (void)hipEventSynchronize(f.readers[0]);
    f.readers[0] = f.readers[1];
    f.readers[1] = done;
}

// common tail of the look-ahead entry points: ingest (device or pinned-host source) + SGBM (+ ORB) of one
// pair on the next engine's stream, `ready` recorded at the end
static int prefetch_pair(vo_ctx* ctx, int slot, const uint8_t* srcL, const uint8_t* srcR, bool from_host, int w, int h,
                         int channels, int preprocessed, vo_ctx::HostStage* hs = nullptr)
{
    int rc;
    FrameSlot& f = ctx->slots[slot];
    const int engine = ctx->next_engine;
    if ((rc = engine_prepare(ctx, engine))) return rc;
    const size_t per = (size_t)w * h * channels;
    hipMemcpyKind kind = hipMemcpyDeviceToDevice;
    if (hs) {
        kind = hipMemcpyHostToDevice;                 // pinned staging filled ahead by vo_host_stage_pair
    } else if (from_host) {
        // user memory -> this engine's pinned buffer (plain memcpy) -> async H2D on the engine's stream;
        // the buffer is reused only after the previous copy out of it has finished
        vo_ctx::SgbmWs& a = ctx->ws_alt[engine];
        if (!a.pinned) {
            VO_HIP(ctx, hipHostMalloc((void**)&a.pinned, ctx->stage_bytes * 2, hipHostMallocDefault));
            VO_HIP(ctx, hipEventCreateWithFlags(&a.h2d_done, hipEventDisableTiming));
        }
        if (a.h2d_valid) VO_HIP(ctx, hipEventSynchronize(a.h2d_done));
        memcpy(a.pinned, srcL, per);
        memcpy(a.pinned + per, srcR, per);
        srcL = a.pinned; srcR = a.pinned + per;
        kind = hipMemcpyHostToDevice;
    }
    ctx->next_engine = (engine + 1) % ctx->n_engines;
    f.w = w; f.h = h; f.has_kp = false; f.n_kp = 0; f.kp_pending = false;
    {
        EngineScope on_engine(ctx, engine);          // ctx->stream / staging / SGBM + ORB workspaces are the engine's in here
        const int stagger = ctx->tune_stagger >= 0 ? ctx->tune_stagger : (7 * ctx->n_engines + 8) / 16;
        if (stagger > 0 && stagger < ctx->n_engines) {
            vo_ctx::SgbmWs& p = ctx->ws_alt[(engine - stagger + ctx->n_engines) % ctx->n_engines];
            if (p.mid_valid) (void)hipStreamWaitEvent(ctx->stream, p.mid, 0);
        }
        rc = slot_before_overwrite(ctx, f);
        // a rectified gray pair that already lies in HBM needs no ingest step of its own: the SGBM run's first kernel reads it
        // where it lies and leaves the slot's copy behind (two copy commands less per pair on the engine's queue)
        const bool in_place = kind == hipMemcpyDeviceToDevice && preprocessed && channels == 1;
        if (!rc && !in_place) {
            StageTimer t(ctx, VO_T_UPLOAD);
            const bool pinned_src = hs != nullptr || from_host;   // (the library's own pinned staging either way)
            rc = ingest(ctx, 0, srcL, w, h, channels, preprocessed, f.left, ctx->stage_in, kind, pinned_src);
            if (!rc) rc = ingest(ctx, 1, srcR, w, h, channels, preprocessed, f.right, ctx->stage_in, kind, pinned_src);
        }
        if (!rc && hs) {
            std::lock_guard<std::mutex> lk(ctx->stage_mu);
            if (hipEventRecord(hs->h2d_done, ctx->stream) == hipSuccess) hs->valid = true;
            else rc = vo_fail(ctx, VO_E_HIP, "hipEventRecord failed");
        } else if (!rc && from_host) {
            vo_ctx::SgbmWs& a = ctx->ws_alt[engine];
            if (hipEventRecord(a.h2d_done, ctx->stream) == hipSuccess) a.h2d_valid = true;
            else rc = vo_fail(ctx, VO_E_HIP, "hipEventRecord failed");
        }
#ifdef VO_TEST_HOOKS
        if (!rc && ctx->fault_prefetch > 0 && --ctx->fault_prefetch == 0)
            rc = vo_fail(ctx, VO_E_STATE, "injected failure (VO_FAULT_PREFETCH) inside the engine scope");
#endif
        if (!rc) rc = in_place ? sgbm_run(ctx, f, w, h, srcL, srcR) : sgbm_run(ctx, f, w, h);
        if (!rc && ctx->la_orb) {
            const int* q = ctx->la_orb_params;
            rc = orb_slot_enqueue(ctx, f, q[0], q[1], q[2], q[3]);
            if (!rc) { memcpy(f.kp_params, q, sizeof(f.kp_params)); f.kp_pending = true; }
        }
        if (!rc && hipEventRecord(f.ready, ctx->stream) != hipSuccess) rc = vo_fail(ctx, VO_E_HIP, "hipEventRecord failed");
    }
    if (rc) {
        // the slot holds a half-processed pair: nothing in it may be handed out
        f.kp_pending = false; f.has_pair = false; f.has_disp = false;
        if (f.counted && ctx->inflight > 0) ctx->inflight--;
        f.counted = false;                               // (`pending` stays: an earlier run into this slot may still be in flight)
        return rc;
    }
    f.has_pair = true; f.has_disp = true;
    if (!f.counted) ctx->inflight++;
    f.pending = true; f.counted = true;
    return VO_OK;
}

// look-ahead: ingest + SGBM of a staged pair on an engine's stream.  The main stream keeps running
// the current pair's ORB / matching / pose meanwhile; consumers of the slot wait for `ready`.
extern "C" int vo_prefetch_staged_pair(vo_ctx* ctx, int slot, int index, int preprocessed)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (index < 0 || index >= ctx->staged_n) return vo_fail(ctx, VO_E_ARG, "vo_prefetch_staged_pair: bad index");
    if (!ctx->sg.set) return vo_fail(ctx, VO_E_STATE, "vo_set_sgbm has not been called");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t per = (size_t)ctx->staged_w * ctx->staged_h * ctx->staged_ch;
    return prefetch_pair(ctx, slot, ctx->staged + per * 2 * index, ctx->staged + per * (2 * index + 1), false, ctx->staged_w,
                         ctx->staged_h, ctx->staged_ch, preprocessed);
}

// monocular look-ahead (config 5): the left image of a staged pair into a slot and its ORB extraction (no disparity
// mask), all on a look-ahead engine's stream; vo_orb_detect_and_compute with the same nfeatures / mask_mode 0 then only
// waits.  The main stream keeps matching and scoring the previous pair meanwhile.
extern "C" int vo_prefetch_staged_mono(vo_ctx* ctx, int slot, int index, int nfeatures)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (index < 0 || index >= ctx->staged_n) return vo_fail(ctx, VO_E_ARG, "vo_prefetch_staged_mono: bad index");
    if (nfeatures < 0 || nfeatures > ctx->max_kp) return vo_fail(ctx, VO_E_CAP, "nfeatures %d exceeds max_kp %d", nfeatures, ctx->max_kp);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    FrameSlot& f = ctx->slots[slot];
    // a monocular extraction is short: four engines cover any useful look-ahead (an engine's first use allocates its workspaces)
    const int span = ctx->n_engines < 4 ? ctx->n_engines : 4;
    const int engine = ctx->mono_engine % span;
    if ((rc = engine_prepare(ctx, engine))) return rc;
    ctx->mono_engine = (engine + 1) % span;
    const size_t per = (size_t)ctx->staged_w * ctx->staged_h * ctx->staged_ch;
    const int w = ctx->staged_w, h = ctx->staged_h;
    f.w = w; f.h = h; f.has_kp = false; f.n_kp = 0; f.kp_pending = false; f.has_disp = false;
    {
        EngineScope on_engine(ctx, engine);
        rc = slot_before_overwrite(ctx, f);
        if (!rc) {
            StageTimer t(ctx, VO_T_UPLOAD);
            rc = ingest(ctx, 0, ctx->staged + per * 2 * index, w, h, ctx->staged_ch, 1, f.left, ctx->stage_in, hipMemcpyDeviceToDevice);
        }
        const int params[4] = { nfeatures, 0, 0, 0 };
        if (!rc) rc = orb_slot_enqueue(ctx, f, nfeatures, 0, 0, 0);
        if (!rc) { memcpy(f.kp_params, params, sizeof(f.kp_params)); f.kp_pending = true; }
        if (!rc && hipEventRecord(f.ready, ctx->stream) != hipSuccess) rc = vo_fail(ctx, VO_E_HIP, "hipEventRecord failed");
    }
    if (rc) {
        f.kp_pending = false; f.has_pair = false;
        if (f.counted && ctx->inflight > 0) ctx->inflight--;
        f.counted = false;
        return rc;
    }
    f.has_pair = true;
    if (!f.counted) ctx->inflight++;
    f.pending = true; f.counted = true;
    return VO_OK;
}

// the same from host images (the user's decode/ingest step): the upload of pair i+k overlaps the work
// on pair i; the host buffers are free again when the call returns
extern "C" int vo_prefetch_pair(vo_ctx* ctx, int slot, const uint8_t* left, const uint8_t* right, int w, int h, int channels,
                                int preprocessed)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!left || !right || (channels != 1 && channels != 3)) return vo_fail(ctx, VO_E_ARG, "vo_prefetch_pair: bad argument");
    if (w > ctx->max_w || h > ctx->max_h || w < 16 || h < 16) return vo_fail(ctx, VO_E_CAP, "image %dx%d exceeds context %dx%d", w, h, ctx->max_w, ctx->max_h);
    if (!ctx->sg.set) return vo_fail(ctx, VO_E_STATE, "vo_set_sgbm has not been called");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    return prefetch_pair(ctx, slot, left, right, true, w, h, channels, preprocessed);
}

static int host_stage_alloc(vo_ctx* ctx)
{
    // first use of the staging path: every buffer at once (a pinned allocation takes about a millisecond -- not something to
    // pay inside a stream, buffer by buffer).  Under the staging lock: vo_host_stage_pair may run on a helper thread of the
    // caller while the driving thread calls vo_host_stage_begin.
    std::lock_guard<std::mutex> lk(ctx->stage_mu);
    for (vo_ctx::HostStage& q : ctx->host_stage) {
        if (q.pinned) continue;
        if (hipHostMalloc((void**)&q.pinned, ctx->stage_bytes * 2, hipHostMallocDefault) != hipSuccess) return VO_E_HIP;
        if (hipEventCreateWithFlags(&q.h2d_done, hipEventDisableTiming) != hipSuccess) return VO_E_HIP;
    }
    return VO_OK;
}

// body of the staging thread: one copy at a time, in the order asked for
static void host_stage_loop(vo_ctx* ctx)
{
    (void)hipSetDevice(ctx->device);
    for (;;) {
        vo_ctx::StageJob job;
        hipEvent_t wait_for = nullptr;
        {
            std::unique_lock<std::mutex> lk(ctx->stage_mu);
            ctx->stage_cv.wait(lk, [&] { return ctx->stage_stop || !ctx->stage_jobs.empty(); });
            if (ctx->stage_stop) return;
            job = ctx->stage_jobs.front();
            ctx->stage_jobs.pop_front();
            vo_ctx::HostStage& hs = ctx->host_stage[job.buf];
            if (hs.valid) wait_for = hs.h2d_done;     // the previous upload out of this buffer
        }
        int rc = VO_OK;
        if (wait_for && hipEventSynchronize(wait_for) != hipSuccess) rc = VO_E_HIP;
        if (!rc) {
            vo_ctx::HostStage& hs = ctx->host_stage[job.buf];
            memcpy(hs.pinned, job.left, job.per);
            memcpy(hs.pinned + job.per, job.right, job.per);
        }
        {
            std::lock_guard<std::mutex> lk(ctx->stage_mu);
            ctx->host_stage[job.buf].state = rc;
        }
        ctx->stage_cv.notify_all();
    }
}

// until the copy begun into `buf` has finished (at once when none is pending); its status
static int host_stage_wait(vo_ctx* ctx, int buf)
{
    std::unique_lock<std::mutex> lk(ctx->stage_mu);
    vo_ctx::HostStage& hs = ctx->host_stage[buf];
    ctx->stage_cv.wait(lk, [&] { return hs.state != 1; });
    const int rc = hs.state;
    hs.state = 0;
    return rc;
}

// The copy of a host pair into pinned staging buffer `buf`, handed to the library's staging thread: returns at once.  The two
// images must stay untouched until vo_host_stage_wait / vo_prefetch_host_staged / vo_host_stage_fetch on that buffer returns.
extern "C" int vo_host_stage_begin(vo_ctx* ctx, int buf, const uint8_t* left, const uint8_t* right, int w, int h, int channels)
{
    if (!ctx || buf < 0 || buf >= vo_ctx::N_HOST_STAGE || !left || !right || (channels != 1 && channels != 3))
        return vo_fail(ctx, VO_E_ARG, "vo_host_stage_begin: bad argument");
    if (w > ctx->max_w || h > ctx->max_h || w < 16 || h < 16) return vo_fail(ctx, VO_E_CAP, "image %dx%d exceeds context %dx%d", w, h, ctx->max_w, ctx->max_h);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    if (host_stage_alloc(ctx)) return vo_fail(ctx, VO_E_HIP, "pinned staging memory: allocation failed");
    {
        std::lock_guard<std::mutex> lk(ctx->stage_mu);
        vo_ctx::HostStage& hs = ctx->host_stage[buf];
        if (hs.state == 1) return vo_fail(ctx, VO_E_STATE, "vo_host_stage_begin: a copy into buffer %d is still pending", buf);
        hs.state = 1;
        ctx->stage_jobs.push_back({ buf, left, right, (size_t)w * h * channels });
        if (!ctx->stage_thread.joinable()) ctx->stage_thread = std::thread(host_stage_loop, ctx);
    }
    ctx->stage_cv.notify_all();
    return VO_OK;
}

extern "C" int vo_host_stage_wait(vo_ctx* ctx, int buf)
{
    if (!ctx || buf < 0 || buf >= vo_ctx::N_HOST_STAGE) return vo_fail(ctx, VO_E_ARG, "vo_host_stage_wait: bad argument");
    const int rc = host_stage_wait(ctx, buf);
    return rc ? vo_fail(ctx, rc, "the copy into staging buffer %d failed", buf) : VO_OK;
}

// The same copy done by the calling thread.  May run on a helper thread of the caller while another thread drives the
// context: it touches nothing but staging buffer `buf` (and reports failures by code only: the context's error string
// belongs to the driving thread).
extern "C" int vo_host_stage_pair(vo_ctx* ctx, int buf, const uint8_t* left, const uint8_t* right, int w, int h, int channels)
{
    if (!ctx || buf < 0 || buf >= vo_ctx::N_HOST_STAGE || !left || !right || (channels != 1 && channels != 3)) return VO_E_ARG;
    if (w > ctx->max_w || h > ctx->max_h || w < 16 || h < 16) return VO_E_CAP;
    if (hipSetDevice(ctx->device) != hipSuccess) return VO_E_HIP;
    if (host_stage_alloc(ctx)) return VO_E_HIP;
    if (int rcw = host_stage_wait(ctx, buf)) return rcw;
    vo_ctx::HostStage& hs = ctx->host_stage[buf];
    hipEvent_t wait_for = nullptr;
    { std::lock_guard<std::mutex> lk(ctx->stage_mu); if (hs.valid) wait_for = hs.h2d_done; }
    if (wait_for && hipEventSynchronize(wait_for) != hipSuccess) return VO_E_HIP;   // the previous upload out of this buffer
    const size_t per = (size_t)w * h * channels;
    memcpy(hs.pinned, left, per);
    memcpy(hs.pinned + per, right, per);
    return VO_OK;
}

// the pair a staging buffer holds, copied back out (the caller found no free slot for it and keeps it on the host)
extern "C" int vo_host_stage_fetch(vo_ctx* ctx, int buf, uint8_t* left, uint8_t* right, int w, int h, int channels)
{
    if (!ctx || buf < 0 || buf >= vo_ctx::N_HOST_STAGE || !left || !right || (channels != 1 && channels != 3)) return VO_E_ARG;
    vo_ctx::HostStage& hs = ctx->host_stage[buf];
    if (!hs.pinned || (size_t)w * h * channels > ctx->stage_bytes) return VO_E_STATE;
    if (int rcw = host_stage_wait(ctx, buf)) return rcw;
    const size_t per = (size_t)w * h * channels;
    memcpy(left, hs.pinned, per);
    memcpy(right, hs.pinned + per, per);
    return VO_OK;
}

extern "C" int vo_prefetch_host_staged(vo_ctx* ctx, int slot, int buf, int w, int h, int channels, int preprocessed)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (buf < 0 || buf >= vo_ctx::N_HOST_STAGE || (channels != 1 && channels != 3)) return vo_fail(ctx, VO_E_ARG, "vo_prefetch_host_staged: bad argument");
    if (w > ctx->max_w || h > ctx->max_h || w < 16 || h < 16) return vo_fail(ctx, VO_E_CAP, "image %dx%d exceeds context %dx%d", w, h, ctx->max_w, ctx->max_h);
    if (!ctx->sg.set) return vo_fail(ctx, VO_E_STATE, "vo_set_sgbm has not been called");
    vo_ctx::HostStage& hs = ctx->host_stage[buf];
    if (!hs.pinned) return vo_fail(ctx, VO_E_STATE, "staging buffer %d has not been filled (vo_host_stage_pair / vo_host_stage_begin)", buf);
    if ((rc = host_stage_wait(ctx, buf))) return vo_fail(ctx, rc, "the copy into staging buffer %d failed", buf);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t per = (size_t)w * h * channels;
    return prefetch_pair(ctx, slot, hs.pinned, hs.pinned + per, true, w, h, channels, preprocessed, &hs);
}

extern "C" int vo_set_lookahead_orb(vo_ctx* ctx, int enable, int nfeatures, int mask_mode, int min_disp16, int max_disp16)
{
    if (!ctx) return VO_E_ARG;
    if (enable) {
        if (nfeatures < 0 || nfeatures > ctx->max_kp) return vo_fail(ctx, VO_E_CAP, "nfeatures %d exceeds max_kp %d", nfeatures, ctx->max_kp);
        if (mask_mode != 0 && mask_mode != 1) return vo_fail(ctx, VO_E_ARG, "mask_mode must be 0 or 1");
    }
    ctx->la_orb = enable != 0;
    ctx->la_orb_params[0] = nfeatures; ctx->la_orb_params[1] = mask_mode;
    ctx->la_orb_params[2] = min_disp16; ctx->la_orb_params[3] = max_disp16;
    return VO_OK;
}

extern "C" int vo_sgbm_compute(vo_ctx* ctx, int slot, int16_t* disp16_out)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    FrameSlot& f = ctx->slots[slot];
    if (!f.has_pair) return vo_fail(ctx, VO_E_STATE, "slot %d holds no image pair", slot);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = slot_wait(ctx, f))) return rc;
    rc = sgbm_run(ctx, f, f.w, f.h);
    if (rc) return rc;
    f.has_disp = true; f.kp_pending = false; f.has_kp = false;
    if (disp16_out) {
        VO_HIP(ctx, hipMemcpyAsync(disp16_out, f.disp16, (size_t)f.w * f.h * 2, hipMemcpyDeviceToHost, ctx->stream));
        VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return slot_health(ctx, f, slot);
    }
    return VO_OK;
}

extern "C" int vo_sgbm_compute_host(vo_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h, int16_t* disp16_out)
{
    if (!ctx || !left || !right || !disp16_out) return vo_fail(ctx, VO_E_ARG, "vo_sgbm_compute_host: bad argument");
    if (w > ctx->max_w || h > ctx->max_h || w < 16 || h < 16) return vo_fail(ctx, VO_E_CAP, "image %dx%d exceeds context", w, h);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    FrameSlot& f = ctx->slots[VO_NUM_SLOTS];
    VO_HIP(ctx, hipMemcpyAsync(f.left, left, (size_t)w * h, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(ctx, hipMemcpyAsync(f.right, right, (size_t)w * h, hipMemcpyHostToDevice, ctx->stream));
    int rc = sgbm_run(ctx, f, w, h);
    if (rc) return rc;
    VO_HIP(ctx, hipMemcpyAsync(disp16_out, f.disp16, (size_t)w * h * 2, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return slot_health(ctx, f, VO_NUM_SLOTS);
}

extern "C" int vo_cvt_bgr2gray(vo_ctx* ctx, const uint8_t* bgr, int w, int h, uint8_t* gray)
{
    if (!ctx || !bgr || !gray) return vo_fail(ctx, VO_E_ARG, "vo_cvt_bgr2gray: bad argument");
    if ((size_t)w * h > (size_t)ctx->max_w * ctx->max_h) return vo_fail(ctx, VO_E_CAP, "image too large");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)w * h;
    VO_HIP(ctx, hipMemcpyAsync(ctx->stage_in, bgr, n * 3, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_bgr2gray, dim3(div_up((int)n, 256)), dim3(256), 0, ctx->stream, ctx->stage_in, (int)n, ctx->stage_in + n * 3);
    VO_CHECK_LAUNCH(ctx);
    VO_HIP(ctx, hipMemcpyAsync(gray, ctx->stage_in + n * 3, n, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_remap(vo_ctx* ctx, int cam, const uint8_t* src, int w, int h, uint8_t* dst)
{
    if (!ctx || !src || !dst || cam < 0 || cam > 1) return vo_fail(ctx, VO_E_ARG, "vo_remap: bad argument");
    if (!ctx->has_map[cam]) return vo_fail(ctx, VO_E_STATE, "maps of camera %d not set", cam);
    if ((size_t)w * h > (size_t)ctx->max_w * ctx->max_h) return vo_fail(ctx, VO_E_CAP, "image too large");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)w * h, nm = (size_t)ctx->map_w * ctx->map_h;
    VO_HIP(ctx, hipMemcpyAsync(ctx->stage_in, src, n, hipMemcpyHostToDevice, ctx->stream));
    uint8_t* out = ctx->stage_in + ctx->stage_bytes;
    hipLaunchKernelGGL(k_remap, dim3(div_up(ctx->map_w, 256), ctx->map_h), dim3(256), 0, ctx->stream, ctx->stage_in, w, h,
                       ctx->map1[cam], ctx->map2[cam], ctx->map_w, ctx->map_h, out);
    VO_CHECK_LAUNCH(ctx);
    VO_HIP(ctx, hipMemcpyAsync(dst, out, nm, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

// ---- lazy downloads ---------------------------------------------------------------------
__global__ void k_disp_to_f32(const int16_t* __restrict__ d, int n, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)d[i] / 16.0f;
}

static int ensure_img3(vo_ctx* ctx, size_t bytes)
{
    if (ctx->img3_ws_bytes >= bytes) return VO_OK;
    if (ctx->img3_ws) (void)hipFree(ctx->img3_ws);
    ctx->img3_ws = nullptr; ctx->img3_ws_bytes = 0;
    VO_HIP(ctx, hipMalloc((void**)&ctx->img3_ws, bytes));
    ctx->img3_ws_bytes = bytes;
    return VO_OK;
}

extern "C" int vo_download_disparity_f32(vo_ctx* ctx, int slot, float* out)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    FrameSlot& f = ctx->slots[slot];
    if (!f.has_disp || !out) return vo_fail(ctx, VO_E_STATE, "slot %d holds no disparity", slot);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = slot_wait(ctx, f))) return rc;
    const int n = f.w * f.h;
    if ((rc = ensure_img3(ctx, (size_t)n * 12))) return rc;
    hipLaunchKernelGGL(k_disp_to_f32, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, f.disp16, n, ctx->img3_ws);
    VO_CHECK_LAUNCH(ctx);
    VO_HIP(ctx, hipMemcpyAsync(out, ctx->img3_ws, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return slot_health(ctx, f, slot);
}

extern "C" int vo_download_left(vo_ctx* ctx, int slot, uint8_t* out)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    FrameSlot& f = ctx->slots[slot];
    if (!f.has_pair || !out) return vo_fail(ctx, VO_E_STATE, "slot %d holds no image", slot);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = slot_wait(ctx, f))) return rc;
    VO_HIP(ctx, hipMemcpyAsync(out, f.left, (size_t)f.w * f.h, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_download_right(vo_ctx* ctx, int slot, uint8_t* out)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    FrameSlot& f = ctx->slots[slot];
    if (!f.has_pair || !out) return vo_fail(ctx, VO_E_STATE, "slot %d holds no image", slot);
    VO_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = slot_wait(ctx, f))) return rc;
    VO_HIP(ctx, hipMemcpyAsync(out, f.right, (size_t)f.w * f.h, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_slot_num_keypoints(vo_ctx* ctx, int slot, int* n_out)
{
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!n_out) return VO_E_ARG;
    *n_out = ctx->slots[slot].has_kp ? ctx->slots[slot].n_kp : 0;
    return VO_OK;
}

extern "C" int vo_enable_timing(vo_ctx* ctx, int on)
{
    if (!ctx) return VO_E_ARG;
    // on = 1: every stage; on > 1: bit mask (1 << stage) << 1 of the stages to time; 0: off
    ctx->timing = on != 0;
    ctx->timing_mask = on == 1 ? ~0u : (unsigned)on >> 1;
    return VO_OK;
}

// resolve the event pairs recorded so far (events are recorded without blocking the stream)
static void resolve_events(vo_ctx* ctx)
{
    if (ctx->ev_used == 0) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (size_t i = 0; i < ctx->ev_used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->ev_pool[2 * i], ctx->ev_pool[2 * i + 1]) == hipSuccess) {
            ctx->t_ms[ctx->ev_stage[i]] += ms;
            ctx->t_n[ctx->ev_stage[i]] += 1;
        }
    }
    ctx->ev_used = 0;
}

StageTimer::StageTimer(vo_ctx* ctx, int s) : c(ctx), stage(s), idx(-1)
{
    if (!c->timing || !((c->timing_mask >> s) & 1u)) return;
    if (c->ev_used * 2 + 2 > c->ev_pool.size()) {
        if (c->ev_pool.size() >= 32768) resolve_events(c);
        else
            for (int k = 0; k < 512; k++) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) c->ev_pool.push_back(e); }
    }
    if (c->ev_used * 2 + 2 > c->ev_pool.size()) return;
    idx = (long)c->ev_used++;
    if (c->ev_stage.size() <= (size_t)idx) c->ev_stage.resize(idx + 1);
    c->ev_stage[idx] = s;
    (void)hipEventRecord(c->ev_pool[2 * idx], c->stream);
}
StageTimer::~StageTimer()
{
    if (idx >= 0) (void)hipEventRecord(c->ev_pool[2 * idx + 1], c->stream);
}

extern "C" int vo_get_timings(vo_ctx* ctx, double* ms_out, int64_t* launches_out, int reset)
{
    if (!ctx) return VO_E_ARG;
    resolve_events(ctx);
    for (int i = 0; i < VO_T_NSTAGES; i++) {
        if (ms_out) ms_out[i] = ctx->t_ms[i];
        if (launches_out) launches_out[i] = ctx->t_n[i];
        if (reset) { ctx->t_ms[i] = 0; ctx->t_n[i] = 0; }
    }
    return VO_OK;
}

// the disparity a slot holds comes from a run whose diagonal sweep gave up a strip hand-off (see FrameSlot::sweep_word)
int slot_health(vo_ctx* ctx, const FrameSlot& f, int slot)
{
    if (f.disp_gen != 0 && f.sweep_word && *(volatile int32_t*)f.sweep_word == f.disp_gen)
        return vo_fail(ctx, VO_E_SWEEP, "slot %d: a strip hand-off of this pair's aggregation sweep gave up waiting (GPU oversubscribed?): "
                                        "its disparity is undefined and nothing computed from it is handed out; submit the pair again", slot);
    return VO_OK;
}

extern "C" int vo_sgbm_sweep_status(vo_ctx* ctx, int* error_out)
{
    if (!ctx || !error_out) return VO_E_ARG;
    VO_HIP(ctx, hipSetDevice(ctx->device));
    int rc = vo_synchronize(ctx);
    if (rc) return rc;
    // SGBM runs of this context in which a wait inside a diagonal sweep exceeded its poll limit (counted by k_sgbm_fin)
    int n = 0;
    VO_HIP(ctx, hipMemcpy(&n, ctx->d_sweep_errs, sizeof(int), hipMemcpyDeviceToHost));
    *error_out = n;
    return VO_OK;
}

// ---- the box's streaming-copy ceiling (the HBM roofline is reported against the 8 TB/s peak and against
// what a plain device copy reaches on this very GPU) -------------------------------------------------------------
typedef uint32_t copy_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) k_copy_stream(const copy_u32x4* __restrict__ src, copy_u32x4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        else dst[i] = src[i];
    }
}

extern "C" int vo_measure_copy(vo_ctx* ctx, int64_t bytes, int reps, int nontemporal, double* gb_per_s)
{
    if (!ctx || !gb_per_s || reps <= 0) return vo_fail(ctx, VO_E_ARG, "vo_measure_copy: bad argument");
    // source = the first path volume, destination = the cost volume: both exist, neither holds anything a later call needs
    const int64_t cap = (int64_t)ctx->vol_cells * 2;
    if (bytes <= 0 || bytes > cap) bytes = cap;
    bytes &= ~(int64_t)4095;
    if (bytes <= 0 || !ctx->ws->C || !ctx->ws->S) return vo_fail(ctx, VO_E_STATE, "vo_measure_copy: no volumes to copy between");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    // the volumes belong to the main workspace (engine 0 uses it too): nothing of the pipeline may still be running
    for (int k = 0; k < vo_ctx::MAX_ENGINES; k++)
        if (ctx->la_stream[k]) VO_HIP(ctx, hipStreamSynchronize(ctx->la_stream[k]));
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipEvent_t e0, e1;
    VO_HIP(ctx, hipEventCreate(&e0));
    VO_HIP(ctx, hipEventCreate(&e1));
    const size_t n16 = (size_t)bytes / 16;
    const int blocks = 256 * 16;                           // 16 workgroups per CU, grid-stride
    auto launch = [&]() {
        if (nontemporal) hipLaunchKernelGGL(k_copy_stream<true>, dim3(blocks), dim3(256), 0, ctx->stream, (const copy_u32x4*)ctx->ws->S, (copy_u32x4*)ctx->ws->C, n16);
        else hipLaunchKernelGGL(k_copy_stream<false>, dim3(blocks), dim3(256), 0, ctx->stream, (const copy_u32x4*)ctx->ws->S, (copy_u32x4*)ctx->ws->C, n16);
    };
    launch();                                               // warm-up (page tables, clocks)
    VO_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; r++) launch();
    VO_HIP(ctx, hipEventRecord(e1, ctx->stream));
    VO_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    VO_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *gb_per_s = ms > 0.f ? 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9 : 0.0;   // bytes read + bytes written
    return VO_OK;
}

__global__ void k_clock_probe(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long w = w0;
    unsigned v = threadIdx.x;
    while (w - w0 < ticks) {
#pragma unroll
        for (int i = 0; i < 64; i++) asm volatile("v_pk_add_i16 %0, %0, %0" : "+v"(v));
        w = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w - w0; out[2] = v; }
}

extern "C" int vo_shader_clock(vo_ctx* ctx, int micros, double* mhz)
{
    if (!ctx || !mhz || micros <= 0 || micros > 1000000) return vo_fail(ctx, VO_E_ARG, "vo_shader_clock: bad argument");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long* d = (unsigned long long*)ctx->red;              // (reduction scratch: nothing of the pipeline keeps state in it)
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, ctx->stream, d, (unsigned long long)micros * 100ull);
    VO_CHECK_LAUNCH(ctx);
    unsigned long long h[2] = { 0, 0 };
    VO_HIP(ctx, hipStreamSynchronize(ctx->stream));
    VO_HIP(ctx, hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    *mhz = h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0;           // cycles per microsecond
    return VO_OK;
}

// timeline of the latest diagonal sweep in the main workspace (development aid): words [8 + 8 s ..] of control block `block`
// = {start, end, failed polls, ticks waiting, misses} of strip s in 100 MHz ticks
extern "C" int vo_sgbm_sweep_stats(vo_ctx* ctx, int block, int32_t* out, int n_words)
{
    if (!ctx || !out || block < 0 || block > 1 || n_words <= 0) return vo_fail(ctx, VO_E_ARG, "vo_sgbm_sweep_stats: bad argument");
    VO_HIP(ctx, hipSetDevice(ctx->device));
    int rc = vo_synchronize(ctx);
    if (rc) return rc;
    const int half = ctx->sw_ctl_words / 2;
    if (n_words > half) n_words = half;
    VO_HIP(ctx, hipMemcpy(out, ctx->ws->sw_ctl + block * half, (size_t)n_words * sizeof(int32_t), hipMemcpyDeviceToHost));
    return VO_OK;
}

// look-ahead pairs submitted and not yet waited for or dropped (what a per-pair scheduling policy would look at)
extern "C" int vo_lookahead_depth(vo_ctx* ctx, int* depth_out)
{
    if (!ctx || !depth_out) return VO_E_ARG;
    *depth_out = ctx->inflight;
    return VO_OK;
}

// has the look-ahead work into this slot finished?  (never blocks: a caller that runs ahead speculatively only picks up what is there)
extern "C" int vo_slot_ready(vo_ctx* ctx, int slot, int* ready_out)
{
    if (!ctx || slot < 0 || slot >= VO_NUM_SLOTS || !ready_out) return vo_fail(ctx, VO_E_ARG, "vo_slot_ready: bad argument");
    FrameSlot& f = ctx->slots[slot];
    *ready_out = 1;
    if (f.pending) {
        const hipError_t e = hipEventQuery(f.ready);
        if (e == hipErrorNotReady) *ready_out = 0;
        else if (e != hipSuccess) return vo_fail(ctx, VO_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
    }
    return VO_OK;
}

// which schedule the latest SGBM run of this context took (VO_SCHED_*)
extern "C" int vo_sgbm_last_schedule(vo_ctx* ctx, int* schedule_out)
{
    if (!ctx || !schedule_out) return VO_E_ARG;
    *schedule_out = ctx->last_schedule;
    return VO_OK;
}

extern "C" int vo_sgbm_last_geometry(vo_ctx* ctx, int64_t* cells, int* n_paths)
{
    if (!ctx) return VO_E_ARG;
    if (cells) *cells = ctx->last_cells;
    if (n_paths) *n_paths = ctx->last_paths;
    return VO_OK;
}
