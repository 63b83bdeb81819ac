/*
 * simple_knn.h -- C ABI of the MI355X-native replacement for simple_knn._C.distCUDA2
 * (submodules/simple-knn/spatial.cu:15-26 -> SimpleKNN::knn, submodules/simple-knn/simple_knn.cu:185-221),
 * exported by the same shared library as gs_rasterizer.h (libgs_rasterizer_hip.so).
 *
 * Semantics (simple_knn.cu:131-183): for every point i, the mean of the squared Euclidean distances to its three
 * nearest OTHER points (self excluded by index, so exact duplicates count with distance 0); a slot for which no
 * neighbour exists keeps FLT_MAX, i.e. P < 4 yields ~FLT_MAX/3 (P = 3) or +inf (P <= 2) exactly as the reference does.
 *
 * The reference orders points along a Morton curve and prunes 1024-point boxes; this implementation bins the points
 * into a uniform spatial-hash grid (about 2 points per cell) and searches growing cube shells until the third-best
 * distance is provably final. Both are exact, so results agree up to fp32 rounding of dx*dx+dy*dy+dz*dz.
 * No host synchronisation, no Thrust allocations (simple_knn.cu:187-214 has two D2H copies and five device_vectors).
 */
#ifndef GS_SIMPLE_KNN_H_INCLUDED
#define GS_SIMPLE_KNN_H_INCLUDED

#include <stddef.h>
#include "gs_rasterizer.h" /* gsr_alloc_fn, error codes, gsr_last_error */

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of device scratch gsr_knn_mean_dist2 needs for P points. */
size_t gsr_knn_workspace_size(int P);

/* points: device float[P,3] contiguous (spatial.cu:24 casts the same layout to float3*); mean_dists: device float[P].
 * workspace: device scratch of gsr_knn_workspace_size(P) bytes. Enqueues on `stream` and returns immediately.
 * Returns 0 or a negative GSR_ERR_* code. */
int gsr_knn_mean_dist2(int P, const float* points, float* mean_dists, char* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif
