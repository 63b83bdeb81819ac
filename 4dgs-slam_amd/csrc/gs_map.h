// gs_map.h -- per-Gaussian / per-camera kernels of the SLAM loop around the rasterizer (include/slam_map.h):
// RGB-D seeding, densify / clone / split / prune, camera step. gfx950 / wave64. GM = gaussian_splatting/scene/gaussian_model.py.
#pragma once
#include "gs_device.h"

namespace gsr {

// ------------------------------------------------------------------------------------------------------------------
// Seeding, GM:185-255. One thread per selected pixel: back-projection through the keyframe's pose (Open3D's
// PointCloud::CreateFromRGBDImage with extrinsic = W2C: p_world = W2C^-1 p_cam; rigid, so R^T (p - T)) and the colour of the byte
// image the reference builds first (:186-188).
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) seed_backproject_kernel(int n, const int* __restrict__ pix, int W, int H, const float* __restrict__ depth,
                                                               const float* __restrict__ image, const float* exposure_a, const float* exposure_b,
                                                               float fx, float fy, float cx, float cy, const float* __restrict__ R,
                                                               const float* __restrict__ T, float* __restrict__ xyz, float* __restrict__ f_dc,
                                                               float* __restrict__ rot, float* __restrict__ logit_opacity)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = pix[i];
    const int u = p % W, v = p / W;
    const float z = depth[p];
    const float xc = ((float)u - cx) * z / fx, yc = ((float)v - cy) * z / fy;
    const float dx = xc - T[0], dy = yc - T[1], dz = z - T[2];
    xyz[3 * (size_t)i] = R[0] * dx + R[3] * dy + R[6] * dz;          // R^T (p_cam - T)
    xyz[3 * (size_t)i + 1] = R[1] * dx + R[4] * dy + R[7] * dz;
    xyz[3 * (size_t)i + 2] = R[2] * dx + R[5] * dy + R[8] * dz;
    const float ea = exposure_a ? expf(exposure_a[0]) : 1.0f, eb = exposure_b ? exposure_b[0] : 0.0f;
    const size_t N = (size_t)W * H;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float ab = fminf(fmaxf(ea * image[c * N + p] + eb, 0.0f), 1.0f);          // :186-187
        const float byte = floorf(ab * 255.0f);                                         // .byte() truncates, :188
        f_dc[3 * (size_t)i + c] = (byte / 255.0f - 0.5f) / SH_C0;                       // Open3D colour in [0,1] -> RGB2SH
    }
    rot[4 * (size_t)i] = 1.f; rot[4 * (size_t)i + 1] = 0.f; rot[4 * (size_t)i + 2] = 0.f; rot[4 * (size_t)i + 3] = 0.f;   // :244-245
    logit_opacity[i] = 0.0f;                                                             // inverse_sigmoid(0.5), :246-253
}

// GM:235-242: scales = log(sqrt(clamp_min(distCUDA2, 1e-7) * point_size)), one value (isotropic) or the value three times.
__global__ void __launch_bounds__(256) seed_scales_kernel(int n, const float* __restrict__ mean_dist2, float point_size, int scale_dim,
                                                          float* __restrict__ log_scales)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float s = logf(sqrtf(fmaxf(mean_dist2[i], 0.0000001f) * point_size));
    for (int k = 0; k < scale_dim; k++) log_scales[(size_t)i * scale_dim + k] = s;
}

// ------------------------------------------------------------------------------------------------------------------
// Densification decisions, GM:866-971 (see include/slam_map.h for the exact predicate). flags[4][P].
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) densify_select_kernel(int P, const float* __restrict__ accum, const float* __restrict__ denom,
                                                             const float* __restrict__ log_scales, int scale_dim,
                                                             const float* __restrict__ logit_opacity, float grad_threshold, float dense_scale,
                                                             float min_opacity, float big_scale, int* __restrict__ flags)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float g = accum[i] / denom[i];                                    // :954
    if (g != g) g = 0.0f;                                             // grads[grads.isnan()] = 0, :955
    float s = expf(log_scales[(size_t)i * scale_dim]);
    for (int k = 1; k < scale_dim; k++) s = fmaxf(s, expf(log_scales[(size_t)i * scale_dim + k]));
    const float o = 1.0f / (1.0f + expf(-logit_opacity[i]));
    const bool hot = g >= grad_threshold;
    const bool clone = hot && s <= dense_scale;                       // :922-929
    const bool split = hot && s > dense_scale;                        // :871-877
    const bool faint = o < min_opacity;                               // :960
    const bool pruned = faint || (big_scale > 0.0f && s > big_scale);                         // :963-969 (the screen-size term is dead: :857)
    // the children's scale is exp(log(scale / 1.6)): evaluate the test on what get_scaling will return for them
    const bool pruned_child = faint || (big_scale > 0.0f && expf(logf(s / 1.6f)) > big_scale);
    flags[i] = (!split && !pruned) ? 1 : 0;
    flags[(size_t)P + i] = (clone && !pruned) ? 1 : 0;
    flags[2 * (size_t)P + i] = split ? 1 : 0;
    flags[3 * (size_t)P + i] = (split && !pruned_child) ? 1 : 0;
}

constexpr int DENSIFY_MAX_TENSORS = 32;
struct DensifyTensor { const float* src; float* dst; int width; int kind; };
struct DensifyArgs {
    int P, n_keep, n_clone, n_split, n_child, ntensors, scale_dim;
    const int* flags; const int* offsets;
    const float* xyz; const float* log_scales; const float* raw_rot; const float* noise;
    DensifyTensor t[DENSIFY_MAX_TENSORS];
};

// One thread per SOURCE Gaussian: writes its surviving copy, its clone and its two children (whichever exist) into every
// destination tensor. Source-parallel, so no index map is needed: the exclusive prefix sums of the flag rows are the destinations.
__global__ void __launch_bounds__(256) densify_apply_kernel(DensifyArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    const size_t P = (size_t)a.P;
    const bool keep = a.flags[i] != 0, clone = a.flags[P + i] != 0, split = a.flags[2 * P + i] != 0, child = a.flags[3 * P + i] != 0;
    if (!keep && !clone && !child) return;
    const long long r_keep = keep ? a.offsets[i] : -1;
    const long long r_clone = clone ? (long long)a.n_keep + a.offsets[P + i] : -1;
    const long long r_c0 = child ? (long long)a.n_keep + a.n_clone + a.offsets[3 * P + i] : -1;
    const long long r_c1 = child ? r_c0 + a.n_child : -1;
    float cx[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};   // children's positions
    if (child && split) {
        // GM:877-883: samples ~ N(0, diag(scale^2)) rotated by the (normalised) quaternion, added to the parent's position
        float st[3];
        for (int k = 0; k < 3; k++) st[k] = expf(a.log_scales[(size_t)i * a.scale_dim + (a.scale_dim == 1 ? 0 : k)]);
        const float qr = a.raw_rot[4 * (size_t)i], qx = a.raw_rot[4 * (size_t)i + 1], qy = a.raw_rot[4 * (size_t)i + 2], qz = a.raw_rot[4 * (size_t)i + 3];
        const float nrm = sqrtf(qr * qr + qx * qx + qy * qy + qz * qz);           // build_rotation, general_utils.py:118-141
        const float r = qr / nrm, x = qx / nrm, y = qy / nrm, z = qz / nrm;
        const float Rm[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                                {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                                {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
        const size_t s0 = (size_t)a.offsets[2 * P + i];
        for (int c = 0; c < 2; c++) {
            const float* nz = a.noise + 3 * (s0 + (size_t)c * a.n_split);
            const float v0 = nz[0] * st[0], v1 = nz[1] * st[1], v2 = nz[2] * st[2];
            for (int k = 0; k < 3; k++) cx[c][k] = Rm[k][0] * v0 + Rm[k][1] * v1 + Rm[k][2] * v2 + a.xyz[3 * (size_t)i + k];
        }
    }
    for (int ti = 0; ti < a.ntensors; ti++) {
        const DensifyTensor& t = a.t[ti];
        const float* src = t.src + (size_t)i * t.width;
        for (int k = 0; k < t.width; k++) {
            const float v = src[k];
            if (r_keep >= 0) t.dst[(size_t)r_keep * t.width + k] = v;
            if (t.kind == GSR_DENSIFY_STATE) {                     // fresh rows start with zero moments, GM:812-830
                if (r_clone >= 0) t.dst[(size_t)r_clone * t.width + k] = 0.0f;
                if (r_c0 >= 0) { t.dst[(size_t)r_c0 * t.width + k] = 0.0f; t.dst[(size_t)r_c1 * t.width + k] = 0.0f; }
            } else {
                if (r_clone >= 0) t.dst[(size_t)r_clone * t.width + k] = v;
                if (r_c0 >= 0) {
                    float c0 = v, c1 = v;
                    if (t.kind == GSR_DENSIFY_XYZ) { c0 = cx[0][k]; c1 = cx[1][k]; }
                    else if (t.kind == GSR_DENSIFY_SCALE) { c0 = c1 = logf(expf(v) / (0.8f * 2.0f)); }      // GM:885-887
                    t.dst[(size_t)r_c0 * t.width + k] = c0; t.dst[(size_t)r_c1 * t.width + k] = c1;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Camera step (include/slam_map.h): Adam on the camera's four small tensors, update_pose (utils/pose_utils.py:80-97),
// and the matrices of utils/camera_utils.py:124-148. One wave; lane 0 does the 3x3 algebra.
// ------------------------------------------------------------------------------------------------------------------
struct CameraStepArgs {
    float* p[4]; const float* g[4]; int n[4]; float lr[4];
    float* exp_avg; float* exp_avg_sq; float* step; float beta1, beta2, eps;
    float* R; float* T; const float* proj; float* view; float* full; float* campos; int* converged; float thr; int do_pose; int latch;
};

__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C)
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// Everything the step reads, requested in one go (round 6): the step count, the eight Adam elements (lane k < 8 owns flat element k of the state:
// rot 3, trans 3, a, b -- the n[] of camera_step_args), the pose. Taken where they are used -- four tensors one after the other, each behind the
// previous one's stores, then the pose behind a block barrier, then the stepped deltas back from memory -- they were seven dependent memory
// round trips in a one-block kernel that runs once per tracking iteration. track_tail_kernel issues them before its gradient sums.
struct CameraStepLoads { bool skip, any_grad, mine; int e; float* ps; float lrs, step0, gr, m, v, pv; float Rm[9], Tv[3], proj[16]; };
__device__ __forceinline__ CameraStepLoads camera_step_loads(const CameraStepArgs& a, bool grads_come_later)
{
    CameraStepLoads L;
    L.skip = a.latch && a.converged && a.converged[0];       // converged earlier in this frame's loop: the reference has left the loop by now
    const int lane = threadIdx.x;
    L.any_grad = false;
    for (int s = 0; s < 4; s++) L.any_grad |= a.g[s] != nullptr;
    const int s = lane < 3 ? 0 : lane < 6 ? 1 : lane == 6 ? 2 : 3;
    L.e = lane - (s == 0 ? 0 : s == 1 ? 3 : s == 2 ? 6 : 7);
    // (selects between values pinned in scalar registers: left to itself the compiler turns the selects back into ONE load with a run-time index,
    // and for that it copies the whole argument block to scratch memory -- 216 bytes per lane written and read back at the head of a kernel
    // whose whole point is latency)
    const float* g4[4]; float* p4[4]; float lr4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        g4[k] = a.g[k]; p4[k] = a.p[k]; lr4[k] = a.lr[k];
        asm volatile("" : "+s"(g4[k]), "+s"(p4[k]), "+s"(lr4[k]));
    }
    const float* const gs = s == 0 ? g4[0] : s == 1 ? g4[1] : s == 2 ? g4[2] : g4[3];
    L.ps = s == 0 ? p4[0] : s == 1 ? p4[1] : s == 2 ? p4[2] : p4[3];
    L.lrs = s == 0 ? lr4[0] : s == 1 ? lr4[1] : s == 2 ? lr4[2] : lr4[3];
    L.mine = lane < 8 && gs != nullptr;
    L.step0 = L.any_grad ? a.step[0] : 0.f;
    L.gr = 0.f; L.m = 0.f; L.v = 0.f; L.pv = 0.f;
    if (L.mine) { if (!grads_come_later) L.gr = gs[L.e]; L.m = a.exp_avg[lane]; L.v = a.exp_avg_sq[lane]; }
    if (lane < 8 && L.ps != nullptr) L.pv = L.ps[L.e];
    for (int k = 0; k < 9; k++) L.Rm[k] = a.R[k];
    for (int k = 0; k < 3; k++) L.Tv[k] = a.T[k];
    // (the projection too: read where it is used -- between the stores of view[] and full[], which it may alias for all the compiler knows --
    // every row of the product waited for its own four loads behind the previous row's store: sixteen dependent round trips in lane 0)
    for (int k = 0; k < 16; k++) L.proj[k] = a.full ? a.proj[k] : 0.f;
    return L;
}

// lg (optional, LDS): the eight gradients in the Adam state's order (rot 3, trans 3, a, b) -- track_tail_kernel formed them a moment ago in this
// very block; a.g[s] then only says WHICH tensors are stepped.
__device__ __forceinline__ void camera_step_apply(const CameraStepArgs& a, const CameraStepLoads& L, const float* lg = nullptr)
{
    if (L.skip) return;
    const int lane = threadIdx.x;
    const bool any_grad = L.any_grad, mine = L.mine;
    const int e = L.e;
    float* const ps = L.ps;
    const float lrs = L.lrs, step0 = L.step0;
    float gr = lg && mine ? lg[lane] : L.gr, m = L.m, v = L.v, pv = L.pv;
    float Rm[9], Tv[3];
    for (int k = 0; k < 9; k++) Rm[k] = L.Rm[k];
    for (int k = 0; k < 3; k++) Tv[k] = L.Tv[k];
    // (1) Adam, torch.optim.Adam single-tensor arithmetic (bias corrections in double like torch's _single_tensor_adam)
    if (any_grad) {
        const float stepf = step0 + 1.0f;
        const double bc1 = 1.0 - pow((double)a.beta1, (double)stepf), bc2 = 1.0 - pow((double)a.beta2, (double)stepf);
        if (mine) {
            m = m + (gr - m) * (float)(1.0 - (double)a.beta1);
            v = v * a.beta2 + (float)(1.0 - (double)a.beta2) * gr * gr;
            a.exp_avg[lane] = m; a.exp_avg_sq[lane] = v;
            const float step_size = (float)((double)lrs / bc1), inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
            pv = pv - step_size * (m / (sqrtf(v) * inv_bc2_sqrt + a.eps));
            ps[e] = pv;
        }
        if (lane == 0) a.step[0] = stepf;
    }
    if (threadIdx.x >= 64) return;                             // (track_tail_kernel: the other waves are done)
    // the stepped pose deltas, from the lanes that hold them (the whole first wave takes part in the shuffles)
    const float th0 = __shfl(pv, 0, 64), th1 = __shfl(pv, 1, 64), th2 = __shfl(pv, 2, 64);
    const float rh0 = __shfl(pv, 3, 64), rh1 = __shfl(pv, 4, 64), rh2 = __shfl(pv, 5, 64);
    if (lane != 0) return;
    if (a.do_pose) {
        // (2) SE3_exp(tau) @ [R | T], pose_utils.py:27-97
        const float rho[3] = {rh0, rh1, rh2}, th[3] = {th0, th1, th2};
        const float Wm[9] = {0.f, -th[2], th[1], th[2], 0.f, -th[0], -th[1], th[0], 0.f};          // skew_sym_mat
        float W2[9];
        mat3_mul(Wm, Wm, W2);
        const float angle = sqrtf(th[0] * th[0] + th[1] * th[1] + th[2] * th[2]);
        float ca, cb, va, vb;   // R = I + ca W + cb W2;  V = I + va W + vb W2
        if (angle < 1e-5f) { ca = 1.0f; cb = 0.5f; va = 0.5f; vb = 1.0f / 6.0f; }
        else {
            ca = sinf(angle) / angle; cb = (1.0f - cosf(angle)) / (angle * angle);
            va = (1.0f - cosf(angle)) / (angle * angle); vb = (angle - sinf(angle)) / (angle * angle * angle);
        }
        float dR[9], Vm[9];
        for (int k = 0; k < 9; k++) {
            const float I = (k == 0 || k == 4 || k == 8) ? 1.0f : 0.0f;
            dR[k] = I + ca * Wm[k] + cb * W2[k];
            Vm[k] = I + Wm[k] * va + W2[k] * vb;
        }
        float dt[3];
        for (int i = 0; i < 3; i++) dt[i] = Vm[3 * i] * rho[0] + Vm[3 * i + 1] * rho[1] + Vm[3 * i + 2] * rho[2];
        float Rn[9], Tn[3];
        mat3_mul(dR, Rm, Rn);
        for (int i = 0; i < 3; i++) Tn[i] = dR[3 * i] * Tv[0] + dR[3 * i + 1] * Tv[1] + dR[3 * i + 2] * Tv[2] + dt[i];
        const float tn = sqrtf(rho[0] * rho[0] + rho[1] * rho[1] + rho[2] * rho[2] + th[0] * th[0] + th[1] * th[1] + th[2] * th[2]);
        if (a.converged) a.converged[0] = tn < a.thr ? 1 : 0;
        for (int k = 0; k < 9; k++) { Rm[k] = Rn[k]; a.R[k] = Rn[k]; }
        for (int k = 0; k < 3; k++) { Tv[k] = Tn[k]; a.T[k] = Tn[k]; a.p[0][k] = 0.f; a.p[1][k] = 0.f; }
    }
    // (3) viewmatrix = W2C^T (row-major memory of the transposed matrix): view[4*c + r] = W2C[r][c]
    float view[16];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) view[4 * c + r] = Rm[3 * r + c];
        view[12 + r] = Tv[r];
        view[4 * r + 3] = 0.f;
    }
    view[15] = 1.f;
    float full[16], campos[3];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float s = 0.f;
            for (int k = 0; k < 4; k++) s += view[4 * i + k] * L.proj[4 * k + j];
            full[4 * i + j] = s;
        }
    for (int c = 0; c < 3; c++) campos[c] = -(Rm[c] * Tv[0] + Rm[3 + c] * Tv[1] + Rm[6 + c] * Tv[2]);   // camera centre = -R^T T  (= inverse(view)[3, :3])
    // all stores at the end, nothing read in between
    for (int k = 0; k < 16; k++) a.view[k] = view[k];
    if (a.full) for (int k = 0; k < 16; k++) a.full[k] = full[k];
    if (a.campos) for (int c = 0; c < 3; c++) a.campos[c] = campos[c];
}

__device__ __forceinline__ void camera_step_body(const CameraStepArgs& a) { camera_step_apply(a, camera_step_loads(a, false)); }

__global__ void __launch_bounds__(64) camera_step_kernel(CameraStepArgs a) { camera_step_body(a); }

// ---- the tail of a tracking iteration in ONE launch (include/slam_map.h: gsr_track_step) ---------------------------------------------
// What tau_sum_kernel, loss_finalize_kernel and camera_step_kernel did as three dependent launches of one block each (4.3 + 4.3 + 5.7 us of a
// ~80 us iteration at SLAM sizes, every one of them a kernel boundary plus two dependent memory round trips): the last level of the pose-gradient
// sum (six waves, one per component, fixed order: tau_sum_body), the last level of the two exposure-gradient sums (the tiles' partial sums that
// render_fwd's tracking epilogue left, waves 0 and 1, fixed order), then the camera step with the gradients handed over through LDS.
__global__ void __launch_bounds__(384) track_tail_kernel(int nblocks, const float* __restrict__ tau_partials, float* __restrict__ tau6, int ntiles,
                                                         const float* __restrict__ exposure_partials, float* __restrict__ dL_dexposure, CameraStepArgs a)
{
    __shared__ float s_g[8];
    const CameraStepLoads pre = camera_step_loads(a, true);        // in flight while the sums are formed
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (tau_sum_body's order, a lane's terms fetched eight at a time: one load per trip of a loop of unknown length waits for each of them in turn --
    // 19 dependent round trips for the 1 200 tiles of a 640 x 480 frame, 10 of this kernel's 14.5 us)
    float v = 0.f, e = 0.f;
    for (int b0 = lane; b0 < nblocks; b0 += 64 * 8) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = b0 + 64 * j < nblocks ? tau_partials[(size_t)(b0 + 64 * j) * 6 + k] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) if (b0 + 64 * j < nblocks) v += t[j];
    }
    if (k < 2)
        for (int b0 = lane; b0 < ntiles; b0 += 64 * 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; j++) t[j] = b0 + 64 * j < ntiles ? exposure_partials[2 * (size_t)(b0 + 64 * j) + k] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) if (b0 + 64 * j < ntiles) e += t[j];
        }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { v += __shfl_xor(v, d, 64); e += __shfl_xor(e, d, 64); }
    if (lane == 0) {
        tau6[k] = v;                              // tau = [rho | theta] (DGR/diff_gaussian_rasterization/__init__.py:152-154)
        s_g[k < 3 ? 3 + k : k - 3] = v;           // Adam state order: rot (theta) first, then trans (rho)
        if (k < 2) { dL_dexposure[k] = e; s_g[6 + k] = e; }
    }
    __syncthreads();
    camera_step_apply(a, pre, s_g);
}

// the window keyframes of a mapping iteration in one launch: block k steps camera k
constexpr int CAMERA_STEPS_MAX = 12;
struct CameraStepsArgs { CameraStepArgs c[CAMERA_STEPS_MAX]; };
__global__ void __launch_bounds__(64) camera_steps_kernel(CameraStepsArgs a) { camera_step_body(a.c[blockIdx.x]); }

// ---- device-side schedule + keyframe slots of a graph-captured mapping iteration (include/slam_map.h) ---------------------------
__global__ void __launch_bounds__(64) schedule_advance_kernel(int* __restrict__ counter, const uint32_t* __restrict__ table, int row_words, int rows,
                                                              uint32_t* __restrict__ current)
{
    const int it = counter[0];
    const int row = it < 0 ? 0 : (it < rows ? it : rows - 1);
    for (int k = threadIdx.x; k < row_words; k += 64) current[k] = table[(size_t)row * row_words + k];
    __syncthreads();
    if (threadIdx.x == 0) counter[0] = it + 1;
}

struct KeyframeEntry { float* view; float* full; float* campos; float* exposure_a; float* exposure_b; float* gt_image; float* gt_depth; float* w_rgb; float* w_depth; };
constexpr int SLOTS_MAX = 4, SLOT_GATHER_BLOCKS = 120;
struct SlotGatherArgs { int n_slots; int pixels; const KeyframeEntry* table; const int* index; KeyframeEntry dst[SLOTS_MAX]; };

__device__ __forceinline__ void slot_copy(float* __restrict__ dst, const float* __restrict__ src, size_t n, size_t tid, size_t nthreads)
{
    if (!dst || !src) return;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {
        const size_t n4 = n / 4;
        const float4* s4 = reinterpret_cast<const float4*>(src); float4* d4 = reinterpret_cast<float4*>(dst);
        for (size_t i = tid; i < n4; i += nthreads) d4[i] = s4[i];
        for (size_t i = n4 * 4 + tid; i < n; i += nthreads) dst[i] = src[i];
    } else {
        for (size_t i = tid; i < n; i += nthreads) dst[i] = src[i];
    }
}

__global__ void __launch_bounds__(256) slot_gather_kernel(SlotGatherArgs a)
{
    const int s = blockIdx.y;
    const KeyframeEntry e = a.table[a.index[s]];
    const KeyframeEntry& d = a.dst[s];
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    if (blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t < 16) { d.view[t] = e.view[t]; d.full[t] = e.full[t]; }
        if (t < 3) d.campos[t] = e.campos[t];
        if (t == 0) { if (d.exposure_a && e.exposure_a) d.exposure_a[0] = e.exposure_a[0]; if (d.exposure_b && e.exposure_b) d.exposure_b[0] = e.exposure_b[0]; }
    }
    const size_t N = (size_t)a.pixels;
    slot_copy(d.gt_image, e.gt_image, 3 * N, tid, nt);
    slot_copy(d.gt_depth, e.gt_depth, N, tid, nt);
    slot_copy(d.w_rgb, e.w_rgb, N, tid, nt);
    slot_copy(d.w_depth, e.w_depth, N, tid, nt);
}

// ---- edge mask of a frame: Camera.compute_grad_mask, utils/camera_utils.py:205-233 (+ image_gradient / image_gradient_mask,
// utils/slam_utils.py:5-39), the non-replica branch -------------------------------------------------------------------------------
// The reference does this with ~25 small torch launches (mean, two grouped conv2d on a reflect-padded image, two more on the validity
// mask, products, sqrt, median, compare); here: (1) one kernel per pixel -> gradient magnitude, (2) the exact lower median by a 4-pass
// radix select in one block (what torch.median returns), (3) the compare.
__device__ __forceinline__ int reflect_index(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ void __launch_bounds__(256) edge_intensity_kernel(const float* __restrict__ image, int H, int W, float eps, float* __restrict__ intensity)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= W || y >= H) return;
    const size_t N = (size_t)H * W;
    float p[3][3];
    bool valid = true;
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const size_t at = (size_t)reflect_index(y + dy - 1, H) * W + reflect_index(x + dx - 1, W);   // F.pad(mode="reflect")
            const float g = (image[at] + image[N + at] + image[2 * N + at]) / 3.0f;                       // original_image.mean(dim=0)
            p[dy][dx] = g;
            valid = valid && fabsf(g) > eps;                                                              // image_gradient_mask: all nine
        }
    }
    // Scharr, normalised by 16 (slam_utils.py:5-22): conv_x = [[3,10,3],[0,0,0],[-3,-10,-3]] -> "v", conv_y = its transpose -> "h"
    float gv = (1.0f / 16.0f) * ((3.f * p[0][0] + 10.f * p[0][1] + 3.f * p[0][2]) - (3.f * p[2][0] + 10.f * p[2][1] + 3.f * p[2][2]));
    float gh = (1.0f / 16.0f) * ((3.f * p[0][0] + 10.f * p[1][0] + 3.f * p[2][0]) - (3.f * p[0][2] + 10.f * p[1][2] + 3.f * p[2][2]));
    if (!valid) { gv = 0.f; gh = 0.f; }
    intensity[(size_t)y * W + x] = sqrtf(gv * gv + gh * gh);
}

// k-th smallest (0-based) of n non-negative floats: their bit patterns order like unsigned integers. Radix select, one launch per 8-bit
// digit (most significant first): every block histograms the digit of its slice's candidates in LDS (the top digit of an image's
// gradient magnitudes falls into a handful of bins: there one atomic per (wave, distinct digit) instead of 64 serialised ones on the
// same word), adds its bins to the global histogram, and the block that arrives last (ticket) picks the bin that holds the k-th
// element and narrows (prefix, mask, k) for the next launch. state: {prefix, mask, k, ticket, hist[256]}, zero before the first launch.
// (Round 2 ran ONE block over the whole image four times: 1.9 ms per 640x480 frame on textured input.)
constexpr int SELECT_BLOCKS = 64;
__global__ void __launch_bounds__(1024) radix_select_pass_kernel(const float* __restrict__ x, int n, int k0, int shift, uint32_t* __restrict__ state,
                                                                 float* __restrict__ out)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_last;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t prefix = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t mask = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i0 = blockIdx.x * 1024; i0 < n; i0 += SELECT_BLOCKS * 1024) {
        const int i = i0 + (int)threadIdx.x;
        const uint32_t u = i < n ? __float_as_uint(x[i]) : 0u;
        bool todo = i < n && (u & mask) == prefix;
        const uint32_t d = (u >> shift) & 255u;
        if (shift >= 24) {
            unsigned long long left = __ballot(todo);
            while (left) {
                const uint32_t d0 = (uint32_t)__shfl((int)d, (int)__builtin_ctzll(left), 64);
                const unsigned long long same = __ballot(todo && d == d0);
                if (todo && d == d0 && (int)__builtin_ctzll(same) == (int)(threadIdx.x & 63)) atomicAdd(&hist[d0], (uint32_t)__popcll(same));
                left &= ~same;
                todo = todo && d != d0;
            }
        } else if (todo) {
            atomicAdd(&hist[d], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256 && hist[threadIdx.x]) __hip_atomic_fetch_add(&state[4 + threadIdx.x], hist[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();                                           // (every wave waits for its own atomics' returns before the barrier releases it)
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(&state[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 256) hist[threadIdx.x] = __hip_atomic_load(&state[4 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t kk = shift >= 24 ? (uint32_t)k0 : __hip_atomic_load(&state[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = 0;
        for (; b < 255; b++) { if (kk < hist[b]) break; kk -= hist[b]; }
        const uint32_t np = prefix | (b << shift);
        __hip_atomic_store(&state[0], np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&state[1], mask | (255u << shift), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&state[2], kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&state[3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (shift == 0) out[0] = __uint_as_float(np);
    }
    if (threadIdx.x < 256) __hip_atomic_store(&state[4 + threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next digit
}

__global__ void __launch_bounds__(256) edge_compare_kernel(const float* __restrict__ intensity, int n, const float* __restrict__ median, float edge_threshold,
                                                           unsigned char* __restrict__ mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) mask[i] = intensity[i] > median[0] * edge_threshold ? 1 : 0;
}

// ---- best-fit rotations of the node graph: estimate_rotation, utils/deform_utils.py:130-166 ----------------------------------------
// For every 3x3 cross-covariance S (= sum_k w_k e0_k et_k^T over a node's edges) the rotation R = V U^T of its singular value
// decomposition S = U Sigma V^T, with the column of the SMALLEST singular value flipped when det(V U^T) <= 0 (:157-162): the proper
// rotation closest to S^T. The reference calls torch.svd on [Nv,3,3] per time sample; here one thread per matrix: Jacobi eigenvectors of
// S^T S (double precision), u_i = S v_i / sigma_i for the two largest singular values, third axes by cross products -- which IS the
// reflection rule. S = 0 (the reference zeroes S for unchanged vertices, :148-149) gives the identity.
__device__ inline void kabsch_rotation(const double S[3][3], float* __restrict__ R)
{
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) A[r][c] = S[0][r] * S[0][c] + S[1][r] * S[1][c] + S[2][r] * S[2][c];   // S^T S
    for (int sweep = 0; sweep < 12; sweep++) {                   // cyclic Jacobi: A <- J^T A J, V <- V J
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; p++) {
            for (int q = p + 1; q < 3; q++) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; k++) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
                for (int k = 0; k < 3; k++) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
                for (int k = 0; k < 3; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
            }
        }
    }
    int o[3] = {0, 1, 2};                                        // eigenvalues (sigma^2) in descending order
    for (int a = 0; a < 2; a++) for (int b = a + 1; b < 3; b++) if (A[o[b]][o[b]] > A[o[a]][o[a]]) { const int t = o[a]; o[a] = o[b]; o[b] = t; }
    double v1[3], v2[3], u1[3], u2[3];
    for (int k = 0; k < 3; k++) { v1[k] = V[k][o[0]]; v2[k] = V[k][o[1]]; }
    const double s1 = A[o[0]][o[0]];
    if (!(s1 > 1e-60)) { for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.f : 0.f; return; }
    for (int r = 0; r < 3; r++) { u1[r] = S[r][0] * v1[0] + S[r][1] * v1[1] + S[r][2] * v1[2]; u2[r] = S[r][0] * v2[0] + S[r][1] * v2[1] + S[r][2] * v2[2]; }
    const double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    for (int r = 0; r < 3; r++) u1[r] /= n1;
    const double d12 = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
    for (int r = 0; r < 3; r++) u2[r] -= d12 * u1[r];
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    if (!(n2 > 1e-12 * n1)) {                                    // rank one: any unit vector orthogonal to u1 (the reference's choice is the SVD routine's)
        const int m = fabs(u1[0]) <= fabs(u1[1]) ? (fabs(u1[0]) <= fabs(u1[2]) ? 0 : 2) : (fabs(u1[1]) <= fabs(u1[2]) ? 1 : 2);
        double e[3] = {0, 0, 0}; e[m] = 1.0;
        const double d = u1[m];
        for (int r = 0; r < 3; r++) u2[r] = e[r] - d * u1[r];
        n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    }
    for (int r = 0; r < 3; r++) u2[r] /= n2;
    const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
    const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = (float)(v1[r] * u1[c] + v2[r] * u2[c] + v3[r] * u3[c]);   // V U^T
}

__global__ void __launch_bounds__(64) kabsch_rotation_kernel(int n, const float* __restrict__ S_in, float* __restrict__ R_out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double S[3][3];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) S[r][c] = (double)S_in[9 * (size_t)i + 3 * r + c];
    kabsch_rotation(S, R_out + 9 * (size_t)i);
}

// ---- the two regularisers of the node graph as one kernel each way (round 4) -------------------------------------------------------------
// ARAP (cal_arap_error, utils/deform_utils.py:177-205, without edge weights): for V views x T time samples x M nodes with K neighbours each,
//   E_t[k] = (p_t - nb_t[k]) keep[k];  S = sum_k E_0[k] E_t[k]^T (zero when, in some coordinate, no edge changed, :147-149);  R = kabsch(S);
//   partial[v][t-1][m] = sum_k keep[k] |E_t[k] - R E_0[k]|^2            (R is a constant of the backward pass, :190-204 detach it)
// as the op-by-op tensor program: gather, subtract, mask, einsum, compare, where, a Kabsch launch, einsum, subtract, square, two sums -- and
// twice as many kernels on the way back, each on ~10^5 values: launch latency, not work. One thread per (view, sample, node) here; the
// backward kernel takes one thread per (view, node) and walks the samples, so the first sample's gradient (every E_0) needs no atomics.
// p [V][T][M][3], nb [V][T][M][K][3] (the neighbours' positions, gathered by the caller: control_nodes.gather_rows, whose backward is the
// ordered scatter), keep [V][M][K] (0 / 1), R [V][T-1][M][9], partial [V][T-1][M].
__global__ void __launch_bounds__(64) arap_forward_kernel(int V, int T, int M, int K, const float* __restrict__ p, const float* __restrict__ nb,
                                                          const float* __restrict__ keep, float* __restrict__ R_out, float* __restrict__ partial)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= V * (T - 1) * M) return;
    const int m = i % M, t = 1 + (i / M) % (T - 1), v = i / (M * (T - 1));
    const float* p0 = p + ((size_t)(v * T) * M + m) * 3, * pt = p + ((size_t)(v * T + t) * M + m) * 3;
    const float* n0 = nb + ((size_t)(v * T) * M + m) * K * 3, * nt = nb + ((size_t)(v * T + t) * M + m) * K * 3;
    const float* kp = keep + ((size_t)v * M + m) * K;
    double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    bool same[3] = {true, true, true};
    for (int k = 0; k < K; k++) {
        const float w = kp[k];
        float e0[3], et[3];
        for (int c = 0; c < 3; c++) { e0[c] = (p0[c] - n0[3 * k + c]) * w; et[c] = (pt[c] - nt[3 * k + c]) * w; same[c] = same[c] && e0[c] == et[c]; }
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) S[a][b] += (double)(e0[a] * w) * (double)et[b];
    }
    if (same[0] || same[1] || same[2]) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) S[a][b] = 0.0;
    float R[9];
    {   // (the op-by-op version hands Kabsch an fp32 S: round here too, so that both see the same matrix)
        double Sf[3][3];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Sf[a][b] = (double)(float)S[a][b];
        kabsch_rotation(Sf, R);
    }
    float acc = 0.f;
    for (int k = 0; k < K; k++) {
        const float w = kp[k];
        float e0[3], et[3];
        for (int c = 0; c < 3; c++) { e0[c] = (p0[c] - n0[3 * k + c]) * w; et[c] = (pt[c] - nt[3 * k + c]) * w; }
        float q = 0.f;
        for (int a = 0; a < 3; a++) { const float r = et[a] - (R[3 * a] * e0[0] + R[3 * a + 1] * e0[1] + R[3 * a + 2] * e0[2]); q += r * r; }
        acc += w * q;
    }
    for (int c = 0; c < 9; c++) R_out[9 * (size_t)i + c] = R[c];
    partial[i] = acc;
}

// g [V][T-1][M] (cotangent of partial) -> dp [V][T][M][3], dnb [V][T][M][K][3] (both fully written)
__global__ void __launch_bounds__(64) arap_backward_kernel(int V, int T, int M, int K, const float* __restrict__ p, const float* __restrict__ nb,
                                                           const float* __restrict__ keep, const float* __restrict__ R_in, const float* __restrict__ g,
                                                           float* __restrict__ dp, float* __restrict__ dnb)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= V * M) return;
    const int m = i % M, v = i / M;
    const float* kp = keep + ((size_t)v * M + m) * K;
    const float* p0 = p + ((size_t)(v * T) * M + m) * 3;
    const float* n0 = nb + ((size_t)(v * T) * M + m) * K * 3;
    float* dn0 = dnb + ((size_t)(v * T) * M + m) * K * 3;
    float dp0[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < K; k++) for (int c = 0; c < 3; c++) dn0[3 * k + c] = 0.f;
    for (int t = 1; t < T; t++) {
        const size_t j = ((size_t)v * (T - 1) + (t - 1)) * M + m;
        const float* R = R_in + 9 * j;
        const float gj = g[j];
        const float* pt = p + ((size_t)(v * T + t) * M + m) * 3;
        const float* nt = nb + ((size_t)(v * T + t) * M + m) * K * 3;
        float* dnt = dnb + ((size_t)(v * T + t) * M + m) * K * 3;
        float dpt[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < K; k++) {
            const float w = kp[k];
            float e0[3], et[3], r[3];
            for (int c = 0; c < 3; c++) { e0[c] = (p0[c] - n0[3 * k + c]) * w; et[c] = (pt[c] - nt[3 * k + c]) * w; }
            for (int a = 0; a < 3; a++) r[a] = et[a] - (R[3 * a] * e0[0] + R[3 * a + 1] * e0[1] + R[3 * a + 2] * e0[2]);
            const float cf = 2.0f * w * gj;
            for (int c = 0; c < 3; c++) {
                const float det = cf * r[c] * w;                                                   // d / d E_t, through the mask
                const float de0 = -cf * (R[c] * r[0] + R[3 + c] * r[1] + R[6 + c] * r[2]) * w;    // d / d E_0 = -R^T r
                dpt[c] += det; dnt[3 * k + c] = -det;
                dp0[c] += de0; dn0[3 * k + c] -= de0;
            }
        }
        float* o = dp + ((size_t)(v * T + t) * M + m) * 3;
        o[0] = dpt[0]; o[1] = dpt[1]; o[2] = dpt[2];
    }
    float* o = dp + ((size_t)(v * T) * M + m) * 3;
    o[0] = dp0[0]; o[1] = dp0[1]; o[2] = dp0[2];
}

// Elastic term (ControlNodeWarp.elastic_loss, utils/time_utils.py:1160-1165): per (view, node, neighbour) the UNBIASED variance over the T time
// samples of the edge length |nb_t - x_t|, divided by its own detached value + 1e-5. x [V][M][T][3], nb [V][M][K][T][3] -> ratio [V][M][K]
// (the caller weighs it with the RBF weights, sums the neighbours and averages the nodes: small tensor ops that carry the weights' gradient).
constexpr int ELASTIC_MAX_T = 16;
__global__ void __launch_bounds__(64) elastic_forward_kernel(int n, int M, int K, int T, const float* __restrict__ x, const float* __restrict__ nb,
                                                             float* __restrict__ ratio)
{
    const int i = blockIdx.x * 64 + threadIdx.x;           // (v, m, k)
    if (i >= n) return;
    const float* xs = x + (size_t)(i / K) * T * 3;
    const float* ns = nb + (size_t)i * T * 3;
    float e[ELASTIC_MAX_T], mean = 0.f;
    for (int t = 0; t < T; t++) {
        const float a = ns[3 * t] - xs[3 * t], b = ns[3 * t + 1] - xs[3 * t + 1], c = ns[3 * t + 2] - xs[3 * t + 2];
        e[t] = sqrtf(a * a + b * b + c * c);
        mean += e[t];
    }
    mean /= (float)T;
    float var = 0.f;
    for (int t = 0; t < T; t++) var += (e[t] - mean) * (e[t] - mean);
    var /= (float)(T - 1);
    ratio[i] = var / (var + 1e-5f);
}

// g [V][M][K] -> dnb [V][M][K][T][3] (fully written) and dx_parts [V][M][K][T][3]'s negative: dx[v][m][t] = -sum_k dnb[v][m][k][t] is formed by
// the second kernel in neighbour order
__global__ void __launch_bounds__(64) elastic_backward_kernel(int n, int M, int K, int T, const float* __restrict__ x, const float* __restrict__ nb,
                                                              const float* __restrict__ g, float* __restrict__ dnb)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float* xs = x + (size_t)(i / K) * T * 3;
    const float* ns = nb + (size_t)i * T * 3;
    float* o = dnb + (size_t)i * T * 3;
    float e[ELASTIC_MAX_T], mean = 0.f;
    for (int t = 0; t < T; t++) {
        const float a = ns[3 * t] - xs[3 * t], b = ns[3 * t + 1] - xs[3 * t + 1], c = ns[3 * t + 2] - xs[3 * t + 2];
        e[t] = sqrtf(a * a + b * b + c * c);
        mean += e[t];
    }
    mean /= (float)T;
    float var = 0.f;
    for (int t = 0; t < T; t++) var += (e[t] - mean) * (e[t] - mean);
    var /= (float)(T - 1);
    const float dvar = g[i] / (var + 1e-5f);                                   // the denominator is detached
    for (int t = 0; t < T; t++) {
        const float de = dvar * 2.0f * (e[t] - mean) / (float)(T - 1);
        const float s = e[t] > 0.f ? de / e[t] : 0.f;                            // d |d| / d d = d / |d| (0 at d = 0, like torch's norm)
        for (int c = 0; c < 3; c++) o[3 * t + c] = s * (ns[3 * t + c] - xs[3 * t + c]);
    }
}
__global__ void __launch_bounds__(64) elastic_backward_self_kernel(int n, int K, int T3, const float* __restrict__ dnb, float* __restrict__ dx)
{
    const int i = blockIdx.x * 64 + threadIdx.x;           // (v, m, t, c) flattened: n = V M T 3
    if (i >= n) return;
    const int vm = i / T3, tc = i % T3;
    float acc = 0.f;
    for (int k = 0; k < K; k++) acc -= dnb[((size_t)vm * K + k) * T3 + tc];
    dx[i] = acc;
}

// ---- the isotropic regulariser of the mapping loops (utils/slam_backend.py:653-655, :1189-1191): 10 * mean |s - mean_k s| over the [P, 3]
// activated scales s = exp(raw). As tensor ops: exp, mean, sub, abs, mean, mul and their backward -- a dozen launches per mapping iteration on
// a tensor of a few hundred KB. Forward: per-block sums in a fixed order, then ONE block adds them (in order) and scales; backward: one launch.
__global__ void __launch_bounds__(256) isotropic_partial_kernel(int P, const float* __restrict__ raw, float* __restrict__ partial)
{
    __shared__ float s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (i < P) {
        const float a = expf(raw[3 * (size_t)i]), b = expf(raw[3 * (size_t)i + 1]), c = expf(raw[3 * (size_t)i + 2]);
        const float m = (a + b + c) / 3.0f;
        v = fabsf(a - m) + fabsf(b - m) + fabsf(c - m);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
__global__ void __launch_bounds__(256) isotropic_finalize_kernel(int nblocks, int P, const float* __restrict__ partial, float* __restrict__ loss)
{
    __shared__ float s_p[256];
    float acc = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += partial[b];
    s_p[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 256; k++) t += s_p[k];
        loss[0] = P > 0 ? 10.0f * t / (3.0f * (float)P) : 0.f;
    }
}
__global__ void __launch_bounds__(256) isotropic_backward_kernel(int P, const float* __restrict__ raw, const float* __restrict__ g_loss, float* __restrict__ d_raw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float a = expf(raw[3 * (size_t)i]), b = expf(raw[3 * (size_t)i + 1]), c = expf(raw[3 * (size_t)i + 2]);
    const float m = (a + b + c) / 3.0f;
    auto sgn = [](float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); };
    const float sa = sgn(a - m), sb = sgn(b - m), sc = sgn(c - m), sm = (sa + sb + sc) / 3.0f;
    const float k = g_loss[0] * 10.0f / (3.0f * (float)P);
    d_raw[3 * (size_t)i] = k * (sa - sm) * a; d_raw[3 * (size_t)i + 1] = k * (sb - sm) * b; d_raw[3 * (size_t)i + 2] = k * (sc - sm) * c;
}

}  // namespace gsr
