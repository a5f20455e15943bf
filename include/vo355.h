/*
 * vo355.h -- C ABI of libvo355.so: the MI355X (gfx950) implementation of openVO's
 * stereo-odometry hot path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * The reference (KevinSpevak/openVO) has no FFI of its own: its boundary is the set of
 * cv2 object call sites inside StereoCamera.compute_3d and StereoOdometer.update.  Each
 * entry point below names the reference line(s) it replaces (paths relative to
 * /root/reference/src/openVO/).  The Python classes in openvo_amd/ bind these with ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success, a negative VO_E_* code otherwise;
 * vo_last_error(ctx) returns a message owned by the context.  Host pointers are
 * caller-owned, C-contiguous, and never retained past the call.  Device memory belongs to
 * the opaque context.  A context is bound to one device and one HIP stream and is NOT
 * thread-safe; use one context per thread / per GPU.  No C++ exception crosses the ABI.
 * There is no CPU fallback: if no gfx950 device is usable, vo_create fails.
 */
#ifndef VO355_H
#define VO355_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vo_ctx vo_ctx;

enum {
    VO_OK = 0,
    VO_E_ARG = -1,     /* bad argument / shape */
    VO_E_HIP = -2,     /* HIP runtime error (message has the hipError string) */
    VO_E_STATE = -3,   /* call order violated (e.g. no disparity in that slot yet) */
    VO_E_CAP = -4,     /* capacity given at vo_create exceeded */
    VO_E_NUMERIC = -5, /* degenerate input (Umeyama: <3 points / colinear) */
    VO_E_SWEEP = -6    /* the disparity a result depends on is undefined: a strip hand-off inside the aggregation sweep of that
                          pair gave up waiting (an oversubscribed GPU).  Nothing computed from it is handed out; the pair can be
                          submitted again.  cv2's StereoSGBM is one sequential pass and cannot fail this way (stereo_camera.py:51) */
};

#define VO_NUM_SLOTS 28 /* frame slots per context: the odometer keeps prev and current, up to 25 more hold
                           look-ahead pairs (vo_prefetch_*), one is spare */

/* lifetime ------------------------------------------------------------------------- */
int vo_create(int device_id, int max_w, int max_h, int max_disp, int max_kp, vo_ctx** out);
void vo_destroy(vo_ctx* ctx);
const char* vo_last_error(const vo_ctx* ctx);
int vo_device_name(const vo_ctx* ctx, char* buf, int buflen);
int vo_synchronize(vo_ctx* ctx);
/* How many look-ahead engines the context may use (each owns a HIP stream and a full SGBM + ORB workspace, allocated on its
 * first use: ~0.95 GB at 1280x720 / D = 128).  n <= 0 only asks.  Returns the count in effect (clamped to 1 .. 24 and to what
 * fits 40 % of the device's free memory), or a negative status.  May be called at any time; engines already created above a
 * lowered count keep their memory until vo_destroy but receive no further pairs.  The reference has no counterpart (cv2 keeps
 * one StereoSGBM object per StereoCamera, stereo_camera.py:23): this bounds the footprint of the replacement. */
int vo_set_engines(vo_ctx* ctx, int n);

/* one-off configuration (stereo_camera.py:16-27) ------------------------------------ */
/* map_left_1/2, map_right_1/2 of cv2.initUndistortRectifyMap(..., CV_16SC2)  [stereo_camera.py:19-22] */
int vo_set_rectify_maps(vo_ctx* ctx, int cam /*0=L,1=R*/, const int16_t* map1 /*h*w*2*/,
                        const uint16_t* map2 /*h*w*/, int w, int h);
/* cv2.StereoSGBM_create positional parameters [stereo_camera.py:23-27]; mode 0 = MODE_SGBM
 * (5 paths, the reference's default), 1 = MODE_HH (8 paths) */
int vo_set_sgbm(vo_ctx* ctx, int minDisparity, int numDisparities, int blockSize, int P1, int P2,
                int disp12MaxDiff, int preFilterCap, int uniquenessRatio, int speckleWindowSize,
                int speckleRange, int mode);
/* Q of cv2.stereoRectify [stereo_camera.py:17-18], row-major 4x4 */
int vo_set_Q(vo_ctx* ctx, const double* Q16);
/* slice bounds used by crop_to_valid_region_left [stereo_camera.py:35-37]:
 * rows y0:y1, cols x0:x1 where (x0,y0,x1,y1) = valid_region_left (quirk kept) */
int vo_set_roi(vo_ctx* ctx, int x0, int y0, int x1, int y1);

/* per frame pair (stereo_camera.py:43-55) -------------------------------------------- */
/* cvtColor(BGR2GRAY) if channels==3 [:44-47]; remap x2 unless preprocessed [:48-50].
 * Leaves rectified gray left/right on the device in `slot`. */
int vo_upload_pair(vo_ctx* ctx, int slot, const uint8_t* left, const uint8_t* right, int w, int h,
                   int channels, int preprocessed);
/* streaming ingest: keep n input pairs resident in HBM (vo_stage_pairs_alloc + vo_stage_pair),
 * then feed a slot from pair `index` without touching the host (same processing as vo_upload_pair) */
int vo_stage_pairs_alloc(vo_ctx* ctx, int n, int w, int h, int channels);
int vo_stage_pair(vo_ctx* ctx, int index, const uint8_t* left, const uint8_t* right);
int vo_load_staged_pair(vo_ctx* ctx, int slot, int index, int preprocessed);
/* look-ahead: ingest + StereoSGBM of staged pair `index` into `slot` on the context's second stream,
 * asynchronously; later calls that use the slot wait for it on the device.  Lets the next pair's
 * disparity overlap the current pair's ORB / matching / pose kernels. */
int vo_prefetch_staged_pair(vo_ctx* ctx, int slot, int index, int preprocessed);
/* look-ahead pairs submitted and not yet waited for or dropped (instrumentation; the reference is synchronous:
 * stereo_odometer.py:115-117 computes one pair per update() call) */
int vo_lookahead_depth(vo_ctx* ctx, int* depth_out);
/* the caller gives up a look-ahead slot without consuming it (a prediction of the next pair that did not come true) */
int vo_lookahead_drop(vo_ctx* ctx, int slot);
/* the same from host images -- the caller's decode/ingest step in front of update():
 * copied to pinned staging, uploaded asynchronously on the engine's stream, then as above.  The host
 * buffers are free again when the call returns. */
int vo_prefetch_pair(vo_ctx* ctx, int slot, const uint8_t* left, const uint8_t* right, int w, int h,
                     int channels, int preprocessed);
/* the same in two halves, so that the thread that launches kernels does no memcpy (the reference hands host arrays to
 * update(), stereo_odometer.py:115-116): vo_host_stage_pair copies the two images into pinned staging buffer `buf`
 * (0 .. VO_NUM_HOST_STAGE-1) -- it first waits until the previous upload out of that buffer has finished, and it is the ONE
 * entry point that may run on another thread while the context is in use (it touches nothing but that buffer; two calls
 * must not name the same buffer concurrently); vo_prefetch_host_staged then starts upload + SGBM (+ ORB) of that buffer's
 * pair on a look-ahead engine like vo_prefetch_pair.
 * vo_host_stage_begin hands the same copy to ONE staging thread owned by the library and returns at once (a host written in
 * an interpreted language then needs no thread of its own, and no interpreter lock changes hands per pair); the two images
 * must stay untouched until vo_host_stage_wait, vo_prefetch_host_staged or vo_host_stage_fetch on that buffer has returned
 * (each waits for the copy).  openvo_amd.StereoOdometer.run() keeps a few copies ahead of the pair it submits. */
#define VO_NUM_HOST_STAGE 20
int vo_host_stage_pair(vo_ctx* ctx, int buf, const uint8_t* left, const uint8_t* right, int w, int h, int channels);
int vo_host_stage_begin(vo_ctx* ctx, int buf, const uint8_t* left, const uint8_t* right, int w, int h, int channels);
int vo_host_stage_wait(vo_ctx* ctx, int buf);
int vo_prefetch_host_staged(vo_ctx* ctx, int slot, int buf, int w, int h, int channels, int preprocessed);
/* the pair staging buffer `buf` holds, copied back out (a caller that found no free slot keeps the pair on the host) */
int vo_host_stage_fetch(vo_ctx* ctx, int buf, uint8_t* left, uint8_t* right, int w, int h, int channels);
/* look-ahead keypoints: when enabled, every vo_prefetch_staged_pair also runs the ORB extraction
 * (same arguments as vo_orb_detect_and_compute) behind the SGBM on the engine's stream; a later
 * vo_orb_detect_and_compute on that slot with the SAME arguments only waits and downloads, any
 * other arguments recompute.  Results are identical either way. */
int vo_set_lookahead_orb(vo_ctx* ctx, int enable, int nfeatures, int mask_mode, int min_disp16, int max_disp16);
/* self.stereoSGBM.compute(L, R) [:51]: int16 disparity x16 of the slot's pair; kept on the
 * device; disp16_out (h*w) may be NULL */
int vo_sgbm_compute(vo_ctx* ctx, int slot, int16_t* disp16_out);
/* stand-alone stereoSGBM.compute on host images (the cv2 object seam) */
int vo_sgbm_compute_host(vo_ctx* ctx, const uint8_t* left, const uint8_t* right, int w, int h,
                         int16_t* disp16_out);
/* lazy materialisation of compute_3d's return values [:51-55], FULL (uncropped) images */
int vo_download_disparity_f32(vo_ctx* ctx, int slot, float* out /*h*w*/);
int vo_download_xyz(vo_ctx* ctx, int slot, float* out /*h*w*3*/); /* cv2.reprojectImageTo3D [:52] */
int vo_download_left(vo_ctx* ctx, int slot, uint8_t* out /*h*w*/);
int vo_download_right(vo_ctx* ctx, int slot, uint8_t* out /*h*w*/);
/* stand-alone helpers at the cv2 seams */
int vo_cvt_bgr2gray(vo_ctx* ctx, const uint8_t* bgr, int w, int h, uint8_t* gray);      /* [:45,47] */
int vo_remap(vo_ctx* ctx, int cam, const uint8_t* src, int w, int h, uint8_t* dst);     /* [:30,33] */
int vo_reproject_to_3d(vo_ctx* ctx, const float* disp, int w, int h, const double* Q16,
                       float* xyz /*h*w*3*/);                                           /* [:52] */

/* features (stereo_odometer.py:22,38-41,117) ------------------------------------------ */
/* orb.detectAndCompute(next_img, feature_mask(next_disp)) on the slot's cropped left image.
 * mask_mode 0: no mask; 1: feature_mask fused -- pixel allowed iff
 * min_disp16 <= disp16 <= max_disp16 (MIN/MAX_VALID_DISPARITY*16, [stereo_odometer.py:6-7,38-41]).
 * Results stay on the device in the slot; host outputs may each be NULL.  Keypoints come in
 * canonical order (octave, y, x).  *n_out may exceed nfeatures (OpenCV keeps response ties). */
int vo_orb_detect_and_compute(vo_ctx* ctx, int slot, int nfeatures, int mask_mode, int min_disp16,
                              int max_disp16, float* kp_xy /*cap*2*/, float* kp_size,
                              float* kp_angle, float* kp_response, int32_t* kp_octave,
                              uint8_t* desc /*cap*32*/, int cap, int* n_out);
/* the cv2 object seam on host arrays: img (h rows, stride bytes), mask NULL or same geometry */
int vo_orb_detect_and_compute_host(vo_ctx* ctx, const uint8_t* img, int w, int h, int stride,
                                   const uint8_t* mask, int mask_stride, int nfeatures,
                                   float* kp_xy, float* kp_size, float* kp_angle,
                                   float* kp_response, int32_t* kp_octave, uint8_t* desc, int cap,
                                   int* n_out);
int vo_slot_num_keypoints(vo_ctx* ctx, int slot, int* n_out);
/* download what vo_orb_detect_and_compute left in the slot (when that call passed NULL outputs to keep
 * everything on the device); any output may be NULL */
int vo_download_keypoints(vo_ctx* ctx, int slot, float* kp_xy, float* kp_size, float* kp_angle, float* kp_response,
                          int32_t* kp_octave, uint8_t* desc, int cap, int* n_out);

/* matching (stereo_odometer.py:163-164) ------------------------------------------------ */
/* matcher.knnMatch(q, t, k=2) with NORM_HAMMING: idx/dist nq*2, ascending distance, ties ->
 * lower train index; -1 / INT32_MAX where the train set has fewer than 2 rows */
int vo_bf_knn2_hamming(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                       int32_t* idx, int32_t* dist);
/* m[0].distance < ratio * m[1].distance in double on float32 distances [:164]; host only */
int vo_ratio_filter(const int32_t* idx, const int32_t* dist, int nq, double ratio, int32_t* q_out,
                    int32_t* t_out, int* m_out);

/* 3-D lookup (stereo_camera.py:52 fused with stereo_odometer.py:50-79) ------------------ */
/* bilinear_interpolate_pixels of the slot's (cropped) reprojected image at float keypoint
 * coords; status 0 ok, 1 NaN, 2 no usable tap (reference raises ZeroDivisionError) */
int vo_points3d_at(vo_ctx* ctx, int slot, const float* xy, int n, float* xyz_out /*n*3*/,
                   uint8_t* status_out);
/* same on an explicit host H*W*3 float image (the reference method's own signature) */
int vo_bilinear_at(vo_ctx* ctx, const float* img3d, int w, int h, const float* xy, int n,
                   float* out, uint8_t* status_out);

/* fused pair step: point_clouds(kps_a, kps_b, desc_a, desc_b, 3d_a, 3d_b) [:162-175] entirely on
 * the device for two slots: kNN-2 + ratio + both 3-D lookups.  Returns M matches in match order
 * (ascending query index); outputs may be NULL except m_out. */
int vo_point_clouds(vo_ctx* ctx, int slot_a, int slot_b, double ratio, int32_t* q_idx,
                    int32_t* t_idx, float* pts_a /*cap*3*/, float* pts_b, uint8_t* status_a,
                    uint8_t* status_b, int cap, int* m_out);

/* fused pair step with ONE host synchronisation: point_clouds [:162-175] followed by the filtering
 * and fitting part of point_cloud_transform [:177-205] for two device-resident slots.
 * counts4 = {M matches after the ratio test, n1 after the rigid-body filter (= M when
 * rigidity_thr <= 0), n2 after the outlier pass (= n1 when it did not run), flags}; flags bit0: a
 * 3-D lookup had no usable tap (reference raises ZeroDivisionError), bit1: NaN residual.
 * rc2 = Umeyama status of {first fit, final fit}: 0 ok, 1 not attempted, -1 fewer than 3 points,
 * -2 "Points cannot be colinear".  T2_12 = final 3x4 transform (valid when rc2[1] == 0); T1_12 may be
 * NULL.  The caller applies the motion gates [:207-221]. */
int vo_pose_pair(vo_ctx* ctx, int slot_a, int slot_b, double ratio, int min_matches, double rigidity_thr,
                 double outlier_thr, int32_t* counts4, int32_t* rc2, double* T1_12, double* T2_12);
/* the same step split in two so that the host does not wait for it: _begin enqueues it on a stream of its
 * own (ordered behind the producers of both slots) and returns a ticket, _end waits for that ticket and
 * delivers what vo_pose_pair would have.  At most VO_NUM_POSE_ASYNC tickets may be open (each has scratch and a pinned record
 * of its own; they share three streams; tickets may end in any order).  Lets the pose steps of the pairs whose frames are already on the
 * device run while the caller still handles an earlier pair. */
#define VO_NUM_POSE_ASYNC 8
int vo_pose_pair_begin(vo_ctx* ctx, int slot_a, int slot_b, double ratio, int min_matches, double rigidity_thr,
                       double outlier_thr, int* ticket_out);
int vo_pose_pair_end(vo_ctx* ctx, int ticket, int32_t* counts4, int32_t* rc2, double* T1_12, double* T2_12);

/* pose (stereo_odometer.py:82-105,177-223) ---------------------------------------------- */
/* cv2.estimateAffine3D(src, dst, force_rotation) Umeyama [:190,204]: T 3x4 row-major, scale */
int vo_umeyama(vo_ctx* ctx, const float* src /*m*3*/, const float* dst, int m, int force_rotation,
               double* T12, double* scale_out);
/* rigid_body_filter [:82-105] */
int vo_rigid_clique(vo_ctx* ctx, const float* prev, const float* cur, int m, double thr,
                    int64_t* mask_out /*m*/);
/* cv2.Rodrigues(R)[0] [:212]; host only */
int vo_rodrigues(const double* R9, double* r3);

/* RANSAC essential-matrix hypothesis scoring (BASELINE config 5) -------------------------------
 * NOT part of the reference (openVO has no RANSAC): defined by this build.  iters
 * hypotheses from 8-point minimal sets drawn by a counter-based hash RNG (seed), Sampson distance
 * (pixels, float32) against thr, inliers counted with wavefront ballot + popcount.  pts: n x 2
 * float32 pixel coordinates; K4 = fx, fy, cx, cy.  best2 = {winning hypothesis, its inlier count}
 * (ties -> lowest hypothesis index); mask_out (n) and counts_out (iters) may be NULL. */
int vo_ransac_essential(vo_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K4, int iters,
                        float thr, uint32_t seed, double* E9_out, uint8_t* mask_out, int32_t* counts_out,
                        int32_t* best2_out);
/* Same contract with the five-point minimal solver (Nister 2004; the estimator cv2.findEssentialMat uses):
 * each hypothesis draws 6 correspondences, 5 give up to 10 essential matrices (float64, 10th-degree polynomial, real
 * roots isolated between derivative roots and bisected), the 6th picks the one with the smallest Sampson error; a
 * hypothesis without a real solution scores 0.  n >= 6. */
int vo_ransac_essential5(vo_ctx* ctx, const float* pts1, const float* pts2, int n, const double* K4, int iters,
                         float thr, uint32_t seed, double* E9_out, uint8_t* mask_out, int32_t* counts_out,
                         int32_t* best2_out);

/* Monocular front end of BASELINE config 5 (NOT part of the reference): vo_upload_mono puts one image into a slot
 * (vo_orb_detect_and_compute with mask_mode 0 then extracts its keypoints); vo_mono_pair chains, entirely on the
 * device and with ONE host synchronisation, Hamming kNN-2 between the two slots' descriptors -> ratio test ->
 * essential-matrix RANSAC (solver 8: as vo_ransac_essential, 5: as vo_ransac_essential5) on the survivors.  counts3 = {matches after the
 * ratio test, winning hypothesis, its inlier count}; E9_out = the winner; mask_out / q_idx / t_idx (each `cap`
 * entries, may be NULL): inlier flag, query and train keypoint index of the first counts3[0] entries. */
int vo_upload_mono(vo_ctx* ctx, int slot, const uint8_t* img, int w, int h, int channels);
/* look-ahead for a monocular stream: the left image of staged pair `index` into `slot` and its ORB extraction (mask_mode
 * 0) on a look-ahead engine's stream; a later vo_orb_detect_and_compute(slot, nfeatures, 0, ...) only waits for it */
int vo_prefetch_staged_mono(vo_ctx* ctx, int slot, int index, int nfeatures);
int vo_mono_pair(vo_ctx* ctx, int slot_a, int slot_b, double ratio, const double* K4, int iters, float thr, uint32_t seed,
                 int solver, double* E9_out, int32_t* counts3, uint8_t* mask_out, int32_t* q_idx, int32_t* t_idx, int cap);
/* The same step in two halves, so that several pairs can be in flight (a monocular stream is latency-bound otherwise: each
 * pair's chain is short and narrow).  _begin enqueues the chain on one of VO_NUM_MONO_ASYNC alternates (own stream, scratch
 * and pinned result record), ordered behind whatever still produces the two slots, and returns a ticket; _end waits for that
 * ticket's completion event only and copies the record out.  want_matches != 0: mask / q / t and the second slot's keypoint
 * positions (xy_b_out: `cap` x 2 floats, may be NULL) travel with the record.  Results are those of vo_mono_pair, bit for bit.
 * A slot read by an open ticket may be refilled at any time: the refill is ordered behind the ticket's work on the device.
 * VO_E_STATE when every alternate is open. */
#define VO_NUM_MONO_ASYNC 5
/* 1 when the look-ahead work into `slot` (vo_prefetch_*) has finished or none is pending, 0 while it still runs; never blocks */
int vo_slot_ready(vo_ctx* ctx, int slot, int* ready_out);
int vo_mono_pair_begin(vo_ctx* ctx, int slot_a, int slot_b, double ratio, const double* K4, int iters, float thr, uint32_t seed,
                       int solver, int want_matches, int* ticket_out);
int vo_mono_pair_end(vo_ctx* ctx, int ticket, double* E9_out, int32_t* counts3, uint8_t* mask_out, int32_t* q_idx, int32_t* t_idx,
                     float* xy_b_out, int cap);

/* RANSAC solvePnP hypothesis scoring (north star; BASELINE config 2 names "ORB+SGBM+PnP") ----------
 * NOT part of the reference either (openVO fits 3-D/3-D, stereo_odometer.py:187-205): defined by this
 * build.  iters hypotheses; each draws 4 correspondences (same hash RNG), solves P3P on three of them in
 * float64 and lets the fourth pick among the <= 4 poses; P = K [R|t] in float32; a point is an inlier iff
 * it is in front of the camera and its reprojection error is below thr pixels (division-free float32
 * test), counted with wavefront ballot + popcount.  pts3d: n x 3 float32 (frame of the first view),
 * pts2d: n x 2 float32 pixels in the second view.  Rt12_out: row-major 3x4 world-to-camera pose of the
 * winner (ties -> lowest hypothesis index; all zeros if no hypothesis produced a pose). */
int vo_ransac_pnp(vo_ctx* ctx, const float* pts3d, const float* pts2d, int n, const double* K4, int iters,
                  float thr, uint32_t seed, double* Rt12_out, uint8_t* mask_out, int32_t* counts_out,
                  int32_t* best2_out);

/* instrumentation ------------------------------------------------------------------------ */
/* hipEvent timing of the kernels launched on the context stream (events are recorded without
 * blocking and resolved by vo_get_timings).  Stage ids: */
enum { VO_T_UPLOAD = 0, VO_T_SGBM_COST, VO_T_SGBM_AGG, VO_T_SGBM_WTA, VO_T_SGBM_POST, VO_T_ORB,
       VO_T_MATCH, VO_T_POSE, VO_T_KNN /* the Hamming kNN kernel alone (inside VO_T_MATCH or VO_T_POSE) */, VO_T_NSTAGES };
/* on = 0 off, 1 every stage, otherwise (stage bit mask << 1), e.g. (1 << VO_T_SGBM_AGG) << 1 */
int vo_enable_timing(vo_ctx* ctx, int on);
/* accumulated milliseconds and launch counts per stage since the last reset */
int vo_get_timings(vo_ctx* ctx, double* ms_out /*VO_T_NSTAGES*/, int64_t* launches_out, int reset);
/* algorithmic cost-volume cells (width1*H*D) of the last vo_sgbm_compute, and the number of path directions its dominant
 * aggregation kernel covers (3: NW / N / NE inside k_sgbm_diag; all of them for uniquenessRatio >= 100) */
int vo_sgbm_last_geometry(vo_ctx* ctx, int64_t* cells, int* n_paths);
/* which aggregation schedule the latest SGBM run of this context took -- it follows from the parameters alone (every
 * schedule gives the same bits: stereo_camera.py:51 only sees the disparity) */
enum { VO_SCHED_DIAG = 1,          /* W + E as one volume (k_sgbm_we), NW / N / NE + WTA in the diagonal sweep */
       VO_SCHED_DIAG_RAGGED = 2,   /* the same with k_sgbm_pair for W + E (width - numDisparities not a multiple of 8) */
       VO_SCHED_UNFUSED = 3 };     /* uniquenessRatio >= 100: one volume per direction + per-pixel winner search */
int vo_sgbm_last_schedule(vo_ctx* ctx, int* schedule_out);
/* measurement aid (SURVEY 8(d) "device-copy ceiling"): `reps` streaming copies of `bytes` (<= one cost volume; 0 = a
 * whole one) between two of the context's volumes, timed with HIP events; *gb_per_s counts bytes read + bytes written.
 * Overwrites the cost volume: call it between, not inside, vo_sgbm_compute / vo_prefetch_pair sequences. */
int vo_measure_copy(vo_ctx* ctx, int64_t bytes, int reps, int nontemporal, double* gb_per_s);
/* Measurement aid: `reps` launches of the Hamming kNN-2 kernel (slot_a's descriptors against slot_b's) back to back on the main
 * stream between two HIP events -> microseconds per launch (the event pair's own cost is spread over the launches).  The result
 * arrays are the context's match scratch; nothing the caller holds changes.  Semantics of the kernel: stereo_odometer.py:163. */
int vo_measure_knn(vo_ctx* ctx, int slot_a, int slot_b, int reps, double* us_per_launch);
/* the shader clock the GPU holds right now (MHz): one wave counts its cycles (s_memtime) against the 100 MHz wall counter
 * (s_memrealtime) for `micros` microseconds on the context's main stream; synchronous.  Measurement aid (bench.py records it
 * after every timed window: a GPU that has been idle ramps its clock up over the first tens of milliseconds of work). */
int vo_shader_clock(vo_ctx* ctx, int micros, double* mhz);
/* health of the diagonal aggregation sweeps (synchronises): *error_out = number of SGBM runs of this context in which a wait
 * between strips exceeded its poll limit (the affected pair's results are refused with VO_E_SWEEP where they are picked up:
 * vo_sgbm_compute with an output pointer, vo_download_disparity_f32 / _xyz, vo_orb_detect_and_compute with the fused mask,
 * vo_points3d_at, vo_point_clouds, vo_pose_pair, vo_pose_pair_end); the count is sticky until vo_destroy, a later pair in
 * the same workspace is unaffected */
int vo_sgbm_sweep_status(vo_ctx* ctx, int* error_out);
/* development aid: control block `block` (0 | 1) of the latest aggregation sweep in the main workspace -- word 0 = work items
   taken, word 1 = a wait gave up in this launch (cleared by the next run), words 8 + 8 s .. = {start, end, failed polls, ticks waiting, misses} of strip s (100 MHz ticks).
   No reference counterpart (stereosgbm.cpp is one sequential pass). */
int vo_sgbm_sweep_stats(vo_ctx* ctx, int block, int32_t* out, int n_words);

/* multi-GPU ----------------------------------------------------------------------------
 * NOT part of the reference (openVO is one process on one thread): frame pairs shard across the GPUs of a
 * node, one process per GPU, each running its own context on a contiguous chunk of the stream; the path's
 * only exchange is the gather of the relative poses -- 17 float64 per frame: the row-major 4x4 transform
 * one accepted update() multiplied into c_T_w [stereo_odometer.py:137-138,146-149] and the accept flag --
 * done with RCCL (ncclAllGather over xGMI), bound directly: librccl.so.1 is loaded on first use.  Rank 0
 * obtains the 128-byte id with vo_mgpu_unique_id and hands it to the other ranks by any means (the Python
 * host uses a TCP socket on the node); every rank then calls vo_mgpu_create with the same id. */
typedef struct vo_mgpu vo_mgpu;
int vo_device_count(int* n_out);                       /* HIP devices visible to this process (0 if none) */
int vo_mgpu_unique_id(uint8_t* id128 /*128 bytes*/);   /* ncclGetUniqueId */
int vo_mgpu_create(int device, int rank, int world, const uint8_t* id128, vo_mgpu** out);   /* ncclCommInitRank */
void vo_mgpu_destroy(vo_mgpu* g);
/* what RCCL itself reports for the communicator: ncclCommCount, ncclCommUserRank, and the device it was created on */
int vo_mgpu_info(vo_mgpu* g, int* n_ranks, int* user_rank, int* device);
const char* vo_mgpu_last_error(const vo_mgpu* g);      /* g may be NULL: error of the last failed create / id call */
/* every rank passes n_frames x 17 float64 (same n_frames on every rank); all_n17 receives world x n_frames x 17
 * in rank order */
int vo_mgpu_gather_poses(vo_mgpu* g, const double* local_n17, int n_frames, double* all_n17);
int vo_mgpu_all_gather_f64(vo_mgpu* g, const double* local, int n, double* all /*world*n*/);
/* element-wise max over the ranks, in place (the slowest rank's time of a benchmark; doubles as a barrier) */
int vo_mgpu_all_reduce_max_f64(vo_mgpu* g, double* v, int n);

#ifdef __cplusplus
}
#endif
#endif
