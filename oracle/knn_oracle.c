/*
 * oracle/knn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of simple_knn.distCUDA2 (submodules/simple-knn, abbreviated KNN/):
 * for every point the mean of the squared distances to its 3 nearest OTHER points
 * (KNN/simple_knn.cu:131-183, "self" excluded by index :157,175, result (b0+b1+b2)/3 :182).
 * The Morton ordering / 1024-point boxes of the reference (KNN/simple_knn.cu:63-117,185-221) only
 * prune the search; the value they produce is the exact 3-NN mean, so an O(P^2) scan restates it.
 * With fewer than 4 points the missing neighbours stay at FLT_MAX as in the reference (:153).
 *
 * PARITY STATUS: unpinned by reference outputs (CUDA-only, no tests in the reference); pinned by a
 * scipy cKDTree cross-check in tests/test_oracle_knn.py.
 */
#include <float.h>
#include <stdint.h>

/* KNN/simple_knn.cu:131-145 */
static void updateKBest3(const float* ref, const float* point, float* knn)
{
    float dx = point[0] - ref[0], dy = point[1] - ref[1], dz = point[2] - ref[2];
    float dist = dx * dx + dy * dy + dz * dz;
    for (int j = 0; j < 3; j++) {
        if (knn[j] > dist) { float t = knn[j]; knn[j] = dist; dist = t; }
    }
}

void gso_knn_dist2(int P, const float* points, float* mean_dists)
{
    for (int i = 0; i < P; i++) {
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            updateKBest3(points + 3 * i, points + 3 * j, best);
        }
        mean_dists[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}
