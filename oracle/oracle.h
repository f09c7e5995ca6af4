/*
 * oracle.h - CPU restatement of the reference's stitching hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle_rtvm.so.  The product library (librtvm.so) never links,
 * loads or calls it.
 *
 * PARITY UNPINNED.  The arithmetic of the reference's hot path lives in
 * OpenCV (opencv-python, version not pinned by the reference's
 * requirements.txt:1-2), which is neither vendored in the reference nor
 * installed in the build image, and the reference ships no tests, fixtures or
 * golden vectors.  Each function below restates the published OpenCV 4.x
 * algorithm behind one call site of the reference's main.py (cited per
 * function) as plain sequential C++; the numpy statements of main.py itself
 * (masks, weights, blend, validate, smooth) are restated literally.
 *
 * Every function mirrors the signature of the librtvm.so entry point it
 * checks (include/rtvm.h) minus the context argument.
 */
#ifndef RTVM_ORACLE_H
#define RTVM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Threads the per-pixel and per-row loops may use (OpenMP; default 1).  Only loops whose
 * iterations are independent are split, so the results do not depend on the count. */
extern int rtvm_oracle_nthreads;
int rtvm_oracle_set_threads(int n);

/* main.py:111,717  cv2.cvtColor(BGR2GRAY); rtvm_oracle_set_gray_legacy(1) switches every gray conversion of the
 * detectors to the 14-bit coefficients of OpenCV <= 3.x (returns the previous setting) */
void rtvm_oracle_gray(const uint8_t* bgr, int h, int w, uint8_t* gray);
int rtvm_oracle_set_gray_legacy(int on);

/* ORB parameters that are not fixed by the reference call site are OpenCV's defaults. */
int rtvm_oracle_max_keypoints(int nfeatures);
/* main.py:36,112,718  cv2.ORB_create(n).detectAndCompute on a BGR frame */
int rtvm_oracle_orb(const uint8_t* bgr, int h, int w, int nfeatures, int* n_out, float* xy,
                    float* size, float* angle_deg, float* response, int32_t* octave,
                    uint8_t* desc);
/* intermediate products of rtvm_oracle_orb for stage-level checks: level geometry and
 * the pyramid / blurred pyramid pixels, levels concatenated without borders */
int rtvm_oracle_orb_pyramid(const uint8_t* gray, int h, int w, int* level_h, int* level_w,
                            uint8_t* pyr, uint8_t* pyr_blur);

/* main.py:33,112,718  cv2.SIFT_create(n).detectAndCompute on a BGR frame; desc: n x 128 f32 */
int rtvm_oracle_sift_max_keypoints(int nfeatures);
int rtvm_oracle_sift(const uint8_t* bgr, int h, int w, int nfeatures, int* n_out, float* xy,
                     float* size, float* angle_deg, float* response, int32_t* octave, float* desc);

/* main.py:37,695 (+698 when sorted != 0) */
int rtvm_oracle_match_hamming_xcheck(const uint8_t* q, int nq, const uint8_t* t, int nt,
                                     int32_t* qidx, int32_t* tidx, int32_t* dist, int* m_out,
                                     int sorted);
/* the same under the one-sided cross-check of BFMatcher before OpenCV 4.2 */
int rtvm_oracle_match_hamming_legacy(const uint8_t* q, int nq, const uint8_t* t, int nt,
                                     int32_t* qidx, int32_t* tidx, int32_t* dist, int* m_out,
                                     int sorted);
/* main.py:34,688 */
int rtvm_oracle_match_l2_knn2(const float* q, int nq, const float* t, int nt, int dim,
                              int32_t* idx, float* dist);

/* main.py:856-857 */
int rtvm_oracle_find_homography_ransac(const float* src_xy, const float* dst_xy, int m,
                                       double thresh, int max_iters, double confidence,
                                       double H[9], uint8_t* inlier_mask, int* found);

/* main.py:745, np.matmul(H_old, H_smoothed) for 3x3 float64: numpy hands the product to its BLAS, and the dgemm
 * kernels of a CPU with fused multiply-add (OpenBLAS from Haswell / Zen on, which the numpy wheels carry) sum over
 * k = 0, 1, 2 into a zeroed accumulator with one fused multiply-add per term:
 * C[r][c] = fma(A[r][2], B[2][c], fma(A[r][1], B[1][c], A[r][0] * B[0][c])). */
void rtvm_oracle_chain3x3(const double A[9], const double B[9], double C[9]);

/* main.py:871; fixed != 0 selects the 1/32-pixel fixed-point arithmetic */
int rtvm_oracle_warp_perspective(const uint8_t* bgr, int fh, int fw, const double H[9], int ch,
                                 int cw, int fixed, uint8_t* out);
/* main.py:888-889 */
void rtvm_oracle_distance_transform(const uint8_t* mask, int h, int w, float* dist);
/* main.py:897-898  cv2.GaussianBlur(f32, (31,31), 0) */
void rtvm_oracle_gaussian_blur31(const float* src, int h, int w, float* dst);
/* main.py:871-927 on a u8 canvas (the reference's float64 canvas holds only these integers) */
int rtvm_oracle_warp_blend(uint8_t* canvas, int ch, int cw, const uint8_t* bgr, int fh, int fw,
                           const double H[9], int fixed);

/* main.py:991-997: bounding rectangle {x, y, w, h} of the pixels whose gray value exceeds
 * `threshold` (crop_black_areas); returns 0 when there is none, else 1 */
int rtvm_oracle_crop_rect(const uint8_t* bgr, int h, int w, int threshold, int rect[4]);
/* main.py:1037: cv2.resize(image, (nw, nh), INTER_LINEAR) on 8UC3 (scale_to_screen) */
int rtvm_oracle_resize_linear(const uint8_t* bgr, int h, int w, uint8_t* out, int nh, int nw);

/* main.py:471,1065: cv2.cvtColor(BGR2HSV), 8 bit */
int rtvm_oracle_bgr2hsv(const uint8_t* bgr, int h, int w, uint8_t* hsv);
/* cv2.erode/dilate/morphologyEx on a 0/255 mask; op 0 erode 1 dilate 2 open 3 close,
 * shape 0 rect 1 cross 2 ellipse */
int rtvm_oracle_morphology(const uint8_t* mask, int h, int w, int op, int shape, int ksize, int iterations,
                           uint8_t* out);
/* main.py:1065-1232: the masks of analyze_for_navigation; any output may be NULL */
int rtvm_oracle_nav_obstacles(const uint8_t* bgr, int h, int w, const int32_t* boxes, int n_boxes,
                              uint8_t* mask_fire, uint8_t* mask_smoke, uint8_t* texture_mask,
                              uint8_t* obstacles, uint8_t* obstacles_nav);
/* main.py:1201: cv2.GaussianBlur(gray, (11,11), 0) on 8-bit pixels */
int rtvm_oracle_gauss11_u8(const uint8_t* src, int h, int w, uint8_t* dst);
/* cv2.Canny(gray, low, high), aperture 3, L1 gradient (main.py:489,1141); edges 0 / 255 */
int rtvm_oracle_canny(const uint8_t* gray, int h, int w, double low_thresh, double high_thresh, uint8_t* edges);
/* main.py:1346-1359: occupancy matrix of find_path_astar */
int rtvm_oracle_occupancy_grid(const uint8_t* mask, int h, int w, int scale, uint8_t* grid);

#ifdef __cplusplus
}
#endif
#endif
