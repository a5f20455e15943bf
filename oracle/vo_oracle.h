/*
 * vo_oracle.h -- CPU ORACLE for the openVO stereo-odometry hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under openvo_amd/ may include, link,
 * import or execute anything from oracle/.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker / reported baseline.
 *
 * What it restates: the arithmetic behind the cv2.* call sites of
 *   /root/reference/src/openVO/stereo_camera.py:43-55   (compute_3d)
 *   /root/reference/src/openVO/stereo_odometer.py:115-223 (update, point_clouds,
 *                                                     point_cloud_transform)
 * i.e. OpenCV 4.x (>= 4.5.5, the first release with the Umeyama overload of
 * estimateAffine3D the reference calls).  OpenCV is a third-party dependency
 * that is neither vendored nor pinned by the reference (setup.cfg:17-24 has no
 * install_requires) and is absent from this image, so each function below
 * restates OpenCV's published algorithm from its upstream source file (named
 * per function).
 *
 * PARITY STATUS: "parity unpinned" at every cv2 boundary -- the reference has no
 * tests, golden vectors or fixtures, and cv2 cannot be run
 * here.  The numpy-only parts of the reference (feature_mask,
 * bilinear_interpolate_pixels, rigid_body_filter, outlier formula, update state
 * machine, rot2RPY) ARE pinned by tests/golden/ fixtures generated from the
 * reference's own code (tests/golden/make_golden.py).
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- image front-end (reference stereo_camera.py:44-50) ---- */
/* cv2.cvtColor(BGR2GRAY), 8-bit: imgproc/src/color_rgb.simd.hpp RGB2Gray<uchar> */
void vo_ref_bgr2gray(const uint8_t* bgr, int w, int h, uint8_t* gray);
/* cv2.remap(img, map1(CV_16SC2), map2(CV_16UC1), INTER_LINEAR), border constant 0:
 * imgproc/src/imgwarp.cpp remapBilinear */
void vo_ref_remap_bilinear(const uint8_t* src, int sw, int sh, const int16_t* map1,
                           const uint16_t* map2, int w, int h, uint8_t* dst);

/* ---- StereoSGBM (reference stereo_camera.py:23-27,51): calib3d/src/stereosgbm.cpp ---- */
typedef struct {
    int minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap,
        uniquenessRatio, speckleWindowSize, speckleRange;
    int mode; /* 0 = MODE_SGBM (5 paths, reference default), 1 = MODE_HH (8 paths) */
} vo_ref_sgbm_params;

/* full StereoSGBM::compute: raw disparity -> medianBlur 3x3 -> filterSpeckles.
 * disp_raw / disp_median may be NULL; disp_final H*W int16 (x16 fixed point). */
int vo_ref_sgbm_compute(const uint8_t* L, const uint8_t* R, int w, int h,
                        const vo_ref_sgbm_params* p, int16_t* disp_raw, int16_t* disp_median,
                        int16_t* disp_final);
/* block cost volume C[y][x-minX1][d] (int16, P2 pre-added as OpenCV does); for kernel tests */
int vo_ref_sgbm_cost_volume(const uint8_t* L, const uint8_t* R, int w, int h,
                            const vo_ref_sgbm_params* p, int16_t* C);
void vo_ref_median3x3_s16(const int16_t* src, int w, int h, int16_t* dst);
void vo_ref_filter_speckles(int16_t* img, int w, int h, int newVal, int maxSpeckleSize, int maxDiff);

/* ---- ORB (reference stereo_odometer.py:22,117): features2d/src/orb.cpp, fast.cpp ---- */
/* img: h rows of w bytes at stride `stride`; mask same geometry or NULL.
 * Keypoints are returned in CANONICAL order: by octave, then row-major position at
 * that octave (OpenCV's own order is std::nth_element-dependent).
 * blur_mode 0 = sepFilter2D 8-bit-coefficient path (what ORB's in-place blur of a
 * pyramid sub-matrix takes), 1 = ufixedpoint16 bit-exact GaussianBlur path. */
int vo_ref_orb_detect_and_compute(const uint8_t* img, int w, int h, int stride,
                                  const uint8_t* mask, int mask_stride, int nfeatures,
                                  int blur_mode, float* kp_xy, float* kp_size, float* kp_angle,
                                  float* kp_response, int32_t* kp_octave, uint8_t* desc, int cap,
                                  int* n_out);
/* FAST-9/16 with NMS on one image (score map out, 0 where not a kept corner) */
void vo_ref_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold,
                           uint8_t* score_nms);
/* one pyramid step: resize INTER_LINEAR_EXACT (imgproc/src/resize.cpp bit-exact path) */
void vo_ref_resize_linear_exact(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst,
                                int dw, int dh, int dstride);
int vo_ref_orb_level_size(int w, int h, int level, int* lw, int* lh);
/* the two recalled-not-read details of orb.cpp as switches (0 = what the oracle and the HIP path use):
 * bit 0 level size cvRound(cols / scale); bit 1 cosf / sinf for the descriptor rotation */
void vo_ref_orb_set_variant(int flags);
int vo_ref_orb_get_variant(void);

/* ---- matcher (reference stereo_odometer.py:163-164): core/src/batch_distance.cpp ---- */
void vo_ref_bf_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx,
                            int32_t* dist);
int vo_ref_ratio_filter(const int32_t* idx, const int32_t* dist, int nq, double ratio,
                        int32_t* q_out, int32_t* t_out);

/* ---- 3-D (reference stereo_camera.py:52, stereo_odometer.py:50-79) ---- */
/* cv2.reprojectImageTo3D(disp f32, Q): calib3d/src/calibration.cpp */
void vo_ref_reproject_to_3d(const float* disp, int w, int h, const double* Q, float* xyz);
/* fused: disparity(int16 x16)->f32/16 -> reproject -> crop -> openVO bilinear lookup.
 * roi = (x0,y0,x1,y1) slice bounds of crop_to_valid_region_left. status: 0 ok, 1 NaN
 * result, 2 all four taps excluded (reference raises ZeroDivisionError). */
void vo_ref_points3d_at(const int16_t* disp16, int w, int h, const double* Q, int x0, int y0,
                        int x1, int y1, const float* xy, int n, float* xyz, uint8_t* status);

/* openVO bilinear_interpolate_pixels on an explicit H*W*3 float image (pinned by golden g2) */
void vo_ref_bilinear_at(const float* img3d, int w, int h, const float* xy, int n, float* out,
                        uint8_t* status);

/* ---- pose (reference stereo_odometer.py:82-105,177-223) ---- */
/* cv2.estimateAffine3D(src,dst,force_rotation) Umeyama: calib3d/src/ptsetreg.cpp.
 * returns 0 ok, -1 n<3, -2 "Points cannot be colinear" */
int vo_ref_umeyama(const float* src, const float* dst, int m, int force_rotation, double* T12,
                   double* scale);
void vo_ref_rodrigues(const double* R9, double* r3);
void vo_ref_rigid_clique(const float* prev, const float* cur, int m, double thr, int64_t* mask);
void vo_ref_svd3(const double* A9, double* U9, double* w3, double* Vt9);

/* ---- RANSAC essential-matrix hypothesis scoring (BASELINE config 5; no openVO counterpart) ---- */
void vo_ref_ransac_sample8(uint32_t seed, int h, int n, int* idx8);
void vo_ref_essential_8pt(const float* p1, const float* p2, const int* idx8, const double* K4 /*fx fy cx cy*/, double* E9);
void vo_ref_fundamental_f32(const double* E9, const double* K4, float* F9);
int vo_ref_sampson_count(const float* F9, const float* p1, const float* p2, int n, float thr, uint8_t* mask);
int vo_ref_ransac_essential(const float* p1, const float* p2, int n, const double* K4, int iters, float thr,
                            uint32_t seed, double* E_best9, uint8_t* mask, int32_t* counts, int* best_iter);

/* five-point minimal solver (Nister 2004) and the RANSAC loop around it: x1, x2 = 5 normalised points each; E_out up to
 * 10 x 9 (unit Frobenius norm); returns the number of real solutions.  The hypothesis takes 6 samples: 5 solve, the 6th picks. */
int vo_ref_poly10_roots_unit(const double* c11, double* out10);   /* real roots of a degree-10 polynomial in [-1, 1] */
int vo_ref_essential_5pt(const double* x1_10, const double* x2_10, double* E_out90);
void vo_ref_ransac_sample6(uint32_t seed, int h, int n, int* idx6);
void vo_ref_essential_5pt_hyp(const float* p1, const float* p2, const int* idx6, const double* K4, double* E9);
int vo_ref_ransac_essential5(const float* p1, const float* p2, int n, const double* K4, int iters, float thr,
                             uint32_t seed, double* E_best9, uint8_t* mask, int32_t* counts, int* best_iter);

/* ---- RANSAC solvePnP hypothesis scoring (north star; no openVO counterpart) -------------------- */
void vo_ref_ransac_sample4(uint32_t seed, int h, int n, int* idx4);
/* y: 3 unit bearings, x: 3 points (rows) -> number of poses (<= 4), R (row-major 9 each), t (3 each) */
int vo_ref_p3p(const double* y9, const double* x9, double* R36, double* t12);
int vo_ref_pnp_hypothesis(const float* X, const float* uv, const int* idx4, const double* K4, double* Rt12, float* P12);
int vo_ref_reproj_count(const float* P12, const float* X, const float* uv, int n, float thr, uint8_t* mask);
int vo_ref_ransac_pnp(const float* X, const float* uv, int n, const double* K4, int iters, float thr, uint32_t seed,
                      double* Rt_best12, uint8_t* mask, int32_t* counts, int* best_iter);

/* ---- one-off rectification setup (reference stereo_camera.py:17-22): cv2.stereoRectify with its Python defaults and
 * cv2.initUndistortRectifyMap(..., CV_16SC2); the checker of openvo_amd/calib.py (src/calib.c) ---------------------------- */
void vo_ref_stereo_rectify(const double* K1, const double* d1, int n1, const double* K2, const double* d2, int n2, int w, int h,
                           const double* R9, const double* T3, double* R1_9, double* R2_9, double* P1_12, double* P2_12, double* Q16,
                           int* roi1_xywh, int* roi2_xywh);
void vo_ref_init_undistort_rectify_map(const double* K9, const double* d, int nd, const double* R9, const double* P12, int w, int h,
                                       int16_t* map1 /*h*w*2*/, uint16_t* map2 /*h*w*/);

#ifdef __cplusplus
}
#endif
#endif
