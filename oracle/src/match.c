/*
 * ORACLE (test infrastructure only -- see ../vo_oracle.h).
 *
 * cv2.BFMatcher(NORM_HAMMING).knnMatch(q, t, k=2) and the reference's ratio test
 * (stereo_odometer.py:163-164).  Follows OpenCV core/src/batch_distance.cpp
 * (BatchDistInvoker: strict-< insertion, ties keep the lower train index first).
 * Parity unpinned at the cv2 boundary.
 */
#include <limits.h>
#include <stddef.h>
#include "../vo_oracle.h"

void vo_ref_bf_knn2_hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx,
                            int32_t* dist)
{
    for (int i = 0; i < nq; i++) {
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
        const uint8_t* a = q + (size_t)i * 32;
        for (int j = 0; j < nt; j++) {
            const uint8_t* b = t + (size_t)j * 32;
            int d = 0;
            for (int k = 0; k < 32; k++) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
            if (d < d1) {
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
                else { d1 = d; i1 = j; }
            }
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        dist[2 * i] = d0; dist[2 * i + 1] = d1;
    }
}

/* keep m[0] when (float)d0 < ratio * (float)d1, evaluated in double as Python does.
 * A query with fewer than 2 neighbours would raise IndexError in the reference; here it
 * is reported by returning -1. */
int vo_ref_ratio_filter(const int32_t* idx, const int32_t* dist, int nq, double ratio,
                        int32_t* q_out, int32_t* t_out)
{
    int m = 0;
    for (int i = 0; i < nq; i++) {
        if (idx[2 * i + 1] < 0) return -1;
        double a = (double)(float)dist[2 * i], b = (double)(float)dist[2 * i + 1];
        if (a < ratio * b) { q_out[m] = i; t_out[m] = idx[2 * i]; m++; }
    }
    return m;
}
