// gs_forward.h -- forward kernels: per-Gaussian preprocess, per-tile binning + depth sort, tile compositing.
// gfx950 / wave64. Reference behaviour being reproduced is cited per function (DGR = submodules/diff-gaussian-rasterization).
#pragma once
#include "gs_device.h"

#ifndef GSR_FUSED_SORT_M
#define GSR_FUSED_SORT_M 2   // stages per LDS round trip (2^M keys per thread) of the fused sort networks for 256 / 512 keys
#endif

namespace gsr {

// Gaussians are processed in blocks of GB = 1024 threads. When the tile grid fits in LDS (T <= HIST_LDS_TILES) each block
// bins its instances into a private LDS histogram and publishes it as one row of a [blocks][tiles] matrix; a column scan
// (tile_offsets_kernel) turns the rows into each block's contiguous sub-range inside every tile's segment; the scatter pass
// re-derives the same instances and ranks them with LDS atomics only. No global atomics at all on this path; the fallback
// for huge tile grids uses one global atomic per instance in both passes.
constexpr int GB = 1024;
constexpr int HIST_LDS_TILES = 12288;   // 48 KiB of dynamic LDS

// Per-tile counters live one per 128-byte L2 line: on the atomic fallback path ~500 atomics hit each counter, and atomics to
// one line serialise in the L2 atomic unit -- packed (32 counters per line) the histogram cost 85 us at 200k Gaussians.
constexpr int CTR_STRIDE = 32;

// ---- the geometry and image scratch buffers (rasterizer_impl.h:22-27 `obtain`, 256-byte aligned carves) -------------------------
// __host__ __device__: the host lays a view's buffers out when it launches the single-view kernels; the multi-view kernels
// (gs_views.h) derive the same pointers on the device from the buffers' base addresses.
template <typename T>
__host__ __device__ inline void carve(char*& p, T*& ptr, size_t count)
{
    uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255);
    ptr = reinterpret_cast<T*>(a);
    p = reinterpret_cast<char*>(ptr + count);
}

struct GeomState {
    uint32_t* header;   // HDR_* words of gs_device.h
    TileRec* rec; float* cov3D; uint8_t* clamped;
    int* internal_radii; uint32_t* tiles_touched; uint32_t* point_offsets; uint32_t* block_sums; uint32_t* block_base;
    float* tau_partials;   // [ceil(P/256)][6] per-block sums of dL_dtau (backward)
    __host__ __device__ static GeomState from(char*& p, size_t P)
    {
        GeomState g;
        const size_t nb = (P + GB - 1) / GB + 1;
        carve(p, g.header, HDR_WORDS);
        carve(p, g.rec, P);
        carve(p, g.cov3D, 6 * P); carve(p, g.clamped, P); carve(p, g.internal_radii, P);
        carve(p, g.tiles_touched, P); carve(p, g.point_offsets, P); carve(p, g.block_sums, nb); carve(p, g.block_base, nb);
        carve(p, g.tau_partials, ((P + 255) / 256 + 1) * 6);
        return g;
    }
};
__host__ __device__ inline bool use_lds_hist(size_t T) { return T <= (size_t)HIST_LDS_TILES; }
struct ImageState {
    float* final_T; uint32_t* n_contrib; uint2* ranges; uint32_t* tile_count; uint32_t* tile_cursor;
    float4* final_C;        // per pixel: colour / depth sums without the background term (render_bwd's chunk start-up needs them)
    uint32_t* chunk_base;   // [T + 1] exclusive scan of ceil(tile list length / CHUNK)
    uint32_t* block_tile_base;   // forward-only binning scratch [ceil(P/GB)][T]; last, so backward (which passes P = 0) never needs it
    __host__ __device__ static ImageState from(char*& p, size_t N, size_t T, size_t P)
    {
        ImageState s;
        carve(p, s.final_T, N); carve(p, s.n_contrib, N); carve(p, s.ranges, T); carve(p, s.final_C, N); carve(p, s.chunk_base, T + 1);
        // the flag word sits behind the per-tile counters (index T*CTR_STRIDE)
        carve(p, s.tile_count, T * CTR_STRIDE + 64);
        carve(p, s.tile_cursor, T * CTR_STRIDE);
        carve(p, s.block_tile_base, use_lds_hist(T) ? ((P + GB - 1) / GB) * T : 0);
        return s;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// F1: per-Gaussian preprocess (DGR/cuda_rasterizer/forward.cu:157-258) fused with
//     (a) the per-tile instance histogram (replaces the global 64-bit radix sort's first pass) and
//     (b) the per-block partial sum of tiles_touched (first level of the scan, rasterizer_impl.cu:280).
// One thread per Gaussian, GB threads per block.
// ------------------------------------------------------------------------------------------------------------------
struct PreprocessArgs {
    int P, D, M, W, H, gx, gy;
    const float* means3D; const float* scales; float scale_modifier; const float* rotations; const float* opacities;
    const float* shs; const float* cov3D_precomp; const float* colors_precomp;
    const float* viewmatrix; const float* projmatrix; const float* cam_pos;
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int prefiltered;
    int eager;                   // request every per-Gaussian row up front (small launches: latency-bound); 0 = rows of culled Gaussians are never read
    int* radii; int* n_touched;
    TileRec* rec; float* cov3D; uint8_t* clamped;
    uint32_t* tiles_touched; uint32_t* block_sums; uint32_t* tile_count; uint32_t* flags;   // flags: zero-filled together with tile_count
    uint32_t* block_tile_base;   // [nblocks][T] when the LDS histogram path is taken, else nullptr
    int sh_win_offset;           // > 0: byte offset, in the dynamic LDS, of GB / 64 SH row windows (gs_device.h: stage_rows) -- the launch reserved them
    RawInputs raw;               // raw.xyz != nullptr: read the model's raw parameters instead (fused prologue, gs_device.h)
};

// forward.cu:120-154 -- Sigma = Rq diag(s*mod)^2 Rq^T, quaternion deliberately NOT normalised (:129).
__device__ __forceinline__ void cov3d_from_scale_rot(const float* s3, float mod, const float* q4, float* cov6)
{
    const float sx = mod * s3[0], sy = mod * s3[1], sz = mod * s3[2];
    const float r = q4[0], x = q4[1], y = q4[2], z = q4[3];
    const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
    const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
    const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
    // M[k][i] = s_k * R[i][k];  Sigma[i][j] = sum_k M[k][i] M[k][j]
    const float m00 = sx * R00, m01 = sx * R10, m02 = sx * R20;
    const float m10 = sy * R01, m11 = sy * R11, m12 = sy * R21;
    const float m20 = sz * R02, m21 = sz * R12, m22 = sz * R22;
    cov6[0] = m00 * m00 + m10 * m10 + m20 * m20;
    cov6[1] = m00 * m01 + m10 * m11 + m20 * m21;
    cov6[2] = m00 * m02 + m10 * m12 + m20 * m22;
    cov6[3] = m01 * m01 + m11 * m11 + m21 * m21;
    cov6[4] = m01 * m02 + m11 * m12 + m21 * m22;
    cov6[5] = m02 * m02 + m12 * m12 + m22 * m22;
}

// The 2x3 matrix A = J * R_cw and cov2D = A Sigma A^T + 0.3 I (forward.cu:76-115; re-used by backward.cu:171-206).
struct Cov2D {
    f3 t;                 // view-space mean with the fov clamp applied to x,y (forward.cu:84-89)
    float txtz, tytz;     // unclamped ratios (backward.cu:177-183 needs them for the gradient masks)
    float J00, J02, J11, J12;
    float A[2][3];
    float a, b, c;
};
__device__ __forceinline__ Cov2D cov2d_eval(f3 mean, float fx, float fy, float tan_fovx, float tan_fovy,
                                            const float* cov6, const float* __restrict__ vm)
{
    Cov2D o;
    f3 t = xform_point_4x3(mean, vm);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    o.txtz = t.x / t.z;
    o.tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, o.txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, o.tytz)) * t.z;
    o.t = t;
    o.J00 = fx / t.z; o.J02 = -(fx * t.x) / (t.z * t.z);
    o.J11 = fy / t.z; o.J12 = -(fy * t.y) / (t.z * t.z);
#pragma unroll
    for (int k = 0; k < 3; k++) {  // R_cw[i][k] = vm[i + 4k]
        o.A[0][k] = o.J00 * vm[0 + 4 * k] + o.J02 * vm[2 + 4 * k];
        o.A[1][k] = o.J11 * vm[1 + 4 * k] + o.J12 * vm[2 + 4 * k];
    }
    const float V00 = cov6[0], V01 = cov6[1], V02 = cov6[2], V11 = cov6[3], V12 = cov6[4], V22 = cov6[5];
    const float av00 = o.A[0][0] * V00 + o.A[0][1] * V01 + o.A[0][2] * V02;
    const float av01 = o.A[0][0] * V01 + o.A[0][1] * V11 + o.A[0][2] * V12;
    const float av02 = o.A[0][0] * V02 + o.A[0][1] * V12 + o.A[0][2] * V22;
    const float av10 = o.A[1][0] * V00 + o.A[1][1] * V01 + o.A[1][2] * V02;
    const float av11 = o.A[1][0] * V01 + o.A[1][1] * V11 + o.A[1][2] * V12;
    const float av12 = o.A[1][0] * V02 + o.A[1][1] * V12 + o.A[1][2] * V22;
    o.a = av00 * o.A[0][0] + av01 * o.A[0][1] + av02 * o.A[0][2] + 0.3f;  // low-pass, forward.cu:112-113
    o.b = av00 * o.A[1][0] + av01 * o.A[1][1] + av02 * o.A[1][2];
    o.c = av10 * o.A[1][0] + av11 * o.A[1][1] + av12 * o.A[1][2] + 0.3f;
    return o;
}

// forward.cu:22-73. sh points at this Gaussian's [M,3] coefficients. Returns rgb (>=0) and the 3 clamp flags as bits.
// SH coefficients with the DC band already in registers (preprocess_fwd loads it together with the other per-Gaussian rows)
struct ShRegs {
    float d0, d1, d2; const float* rest;
    __device__ __forceinline__ float operator[](int k) const { return k == 0 ? d0 : k == 1 ? d1 : k == 2 ? d2 : rest[k - 3]; }
};
template <typename SH>
__device__ __forceinline__ f3 sh_to_rgb(int deg, const SH sh, f3 pos, f3 campos, uint32_t& clamp_bits)
{
    f3 dir = mk3(pos.x - campos.x, pos.y - campos.y, pos.z - campos.z);
    const float inv = 1.0f / sqrtf(dot3(dir, dir));
    const float x = dir.x * inv, y = dir.y * inv, z = dir.z * inv;
    float res[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float v = SH_C0 * sh[k];
        if (deg > 0) {
            v = v - SH_C1 * y * sh[3 + k] + SH_C1 * z * sh[6 + k] - SH_C1 * x * sh[9 + k];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                v = v + SH_C2[0] * xy * sh[12 + k] + SH_C2[1] * yz * sh[15 + k] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + k] +
                    SH_C2[3] * xz * sh[21 + k] + SH_C2[4] * (xx - yy) * sh[24 + k];
                if (deg > 2) {
                    v = v + SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + k] + SH_C3[1] * xy * z * sh[30 + k] +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + k] + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + k] +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + k] + SH_C3[5] * z * (xx - yy) * sh[42 + k] +
                        SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + k];
                }
            }
        }
        res[k] = v + 0.5f;
    }
    clamp_bits = (res[0] < 0 ? 1u : 0u) | (res[1] < 0 ? 2u : 0u) | (res[2] < 0 ? 4u : 0u);  // forward.cu:69-71
    return mk3(fmaxf(res[0], 0.f), fmaxf(res[1], 0.f), fmaxf(res[2], 0.f));
}

#ifndef GSR_FWD_TIMING
#define GSR_FWD_TIMING 0
#endif
#if GSR_FWD_TIMING
__device__ uint32_t g_pre_timing[8 * 16 * 2048];
__device__ uint32_t g_sca_timing[8 * 16 * 2048];
#define PRE_TICK(arr, k) do { if (blockIdx.x < 2048 && (threadIdx.x & 63) == 0) arr[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (k)] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PRE_TICK(arr, k)
#endif
// RAW: the fused-prologue mode (raw.xyz != nullptr) as a template parameter, so that the plain instantiation is straight-line code.
// Loads: the mean first; then -- `eager` launches (up to ~0.5 M Gaussians: a handful of waves per SIMD, nothing to hide a latency
// behind) -- scale, rotation, opacity and the DC colour of EVERY Gaussian in one go, so that a visible Gaussian pays two global
// latencies instead of four (mean -> scale / rotation -> colour -> opacity); large launches keep the lazy order, which spares the
// culled Gaussians' rows (71 % of 2 M at BASELINE config #5) and has enough waves in flight.
template <bool RAW, bool PRE = false>     // PRE: deformation-network deltas in front of the activations (RawInputs::delta_mode = 1)
__device__ __forceinline__ void preprocess_fwd_body(PreprocessArgs a)
{
    RawInputs R = a.raw;
    if constexpr (!RAW) R = RawInputs{};
    const int idx = blockIdx.x * GB + threadIdx.x;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    __shared__ uint32_t s_wave_sum[GB / 64];
    extern __shared__ uint32_t s_hist[];   // [T] when a.block_tile_base != nullptr
    const int T = a.gx * a.gy;
    const bool lds_hist = a.block_tile_base != nullptr;
    PRE_TICK(g_pre_timing, 0);
    if (lds_hist) {
        for (int t = threadIdx.x; t < T; t += GB) s_hist[t] = 0;
        __syncthreads();
    }
    PRE_TICK(g_pre_timing, 1);

    uint32_t touched = 0;
    int rx0 = 0, ry0 = 0, rw = 0;
    // SH rows through LDS (gs_device.h: stage_rows): when the launch reserved the windows and the colours come from SH coefficients above the DC
    // band, a lane only NOTES that it needs a colour; the wave evaluates them together behind the per-Gaussian part (same arithmetic)
    const bool sh_rows = a.sh_win_offset > 0 && a.D > 0 && a.colors_precomp == nullptr && !(RAW && R.flow_proj1 != nullptr) && (a.M == 9 || a.M == 16);
    bool need_sh = false;
    f3 sh_pos = mk3(0.f, 0.f, 0.f);
    float sh_dc[3] = {0.f, 0.f, 0.f};
    const float* sh_rest = nullptr;
    // eager launches (small P: latency-bound, every row is requested up front anyway): the first band's rows of ALL the wave's Gaussians are
    // requested here, before anything is known about their visibility, and land while the geometry below is computed
    const bool sh_early = sh_rows && a.eager != 0;
    float* const sh_win = sh_rows ? reinterpret_cast<float*>(reinterpret_cast<char*>(s_hist) + a.sh_win_offset) + (size_t)wave * SH_WIN_FLOATS : nullptr;
    if (sh_early) {
        const float* my_rest = idx < a.P ? sh_view(a.shs, R, (size_t)idx, a.M).rest : nullptr;
        stage_rows_issue(sh_win, __ballot(idx < a.P), my_rest, 0, a.D == 1 ? 9 : 24);
    }
    if (idx < a.P) {
        int my_radius = 0;
        if (a.n_touched) a.n_touched[idx] = 0;
        const f3 p = load_mean<PRE>(a.means3D, R, (size_t)idx);
        const bool flow = RAW && R.flow_proj1 != nullptr;
        const bool eager = a.eager != 0;
        float s3[3] = {0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f}, cov6[6], opac = 0.f, dc[3] = {0.f, 0.f, 0.f};
        auto load_shape = [&]() {
            if (a.cov3D_precomp) {
#pragma unroll
                for (int k = 0; k < 6; k++) cov6[k] = a.cov3D_precomp[6 * (size_t)idx + k];
            } else {
                load_scale<PRE>(a.scales, R, (size_t)idx, s3);
                load_rot<PRE>(a.rotations, R, (size_t)idx, q4);
            }
        };
        const ShView shv = sh_view(a.shs, R, (size_t)idx, a.M);
        auto load_look = [&]() {
            opac = load_opacity(a.opacities, R, (size_t)idx);
            if (!flow) {
                if (a.colors_precomp != nullptr) {
#pragma unroll
                    for (int k = 0; k < 3; k++) dc[k] = a.colors_precomp[3 * (size_t)idx + k];   // rasterizer_impl.cu:324
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) dc[k] = shv.dc[k];
                }
            }
        };
        if (eager) { load_shape(); load_look(); }
        const f3 p_view = xform_point_4x3(p, a.viewmatrix);
        // near cull only (auxiliary.h:139-164); a culled point under `prefiltered` is an error (:156-160)
        if (p_view.z <= 0.2f) {
            if (a.prefiltered) atomicOr(a.flags, (uint32_t)FLAG_PREFILTERED);
        } else {
            const float* pm = a.projmatrix;
            const float hx = pm[0] * p.x + pm[4] * p.y + pm[8] * p.z + pm[12];
            const float hy = pm[1] * p.x + pm[5] * p.y + pm[9] * p.z + pm[13];
            const float hw = pm[3] * p.x + pm[7] * p.y + pm[11] * p.z + pm[15];
            const float p_w = 1.0f / (hw + 0.0000001f);  // forward.cu:201
            if (!eager) load_shape();
            if (!a.cov3D_precomp) {
                cov3d_from_scale_rot(s3, a.scale_modifier, q4, cov6);
#pragma unroll
                for (int k = 0; k < 6; k++) a.cov3D[6 * (size_t)idx + k] = cov6[k];
            }
            const Cov2D cv = cov2d_eval(p, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov6, a.viewmatrix);
            const float det = cv.a * cv.c - cv.b * cv.b;  // forward.cu:221-225
            if (det != 0.0f) {
                const float det_inv = 1.f / det;
                const float mid = 0.5f * (cv.a + cv.c);  // forward.cu:231-234
                const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float rad_f = ceilf(3.f * sqrtf(fmaxf(mid + sq, mid - sq)));
                const float px = ndc2pix(hx * p_w, a.W), py = ndc2pix(hy * p_w, a.H);
                int x0, y0, x1, y1;
                tile_rect(px, py, (int)rad_f, a.gx, a.gy, x0, y0, x1, y1);
                int area = (x1 - x0) * (y1 - y0);
                if constexpr (RAW) {
                    // a flow view whose caller only reads the tiles of a rectangle (the keyframe's moving pixels: the flow loss is masked to
                    // them): the Gaussian's tile rectangle is clipped to it -- no instance in a tile that is not read, none at all (culled like
                    // a Gaussian off screen) when nothing is left; scatter_instances_body clips the same way
                    if (flow && R.flow_clip != nullptr) {
                        const int* c = R.flow_clip;
                        x0 = max(x0, c[0]); y0 = max(y0, c[1]); x1 = min(x1, c[2]); y1 = min(y1, c[3]);
                        area = (x1 > x0 && y1 > y0) ? (x1 - x0) * (y1 - y0) : 0;
                    }
                }
                if (area != 0) {
                    if (!eager) load_look();
                    f3 col;
                    if (flow) {
                        // render_flow's colour (gaussian_renderer/__init__.py:262-284): NDC displacement between the two projections
                        // of the (detached) position moved by dx / dx2, and the dynamic-mask channel
                        const size_t row = raw_row(R, (size_t)idx);
                        const int sl = raw_slot(R, row);
                        const f3 base = mk3(R.xyz[3 * row], R.xyz[3 * row + 1], R.xyz[3 * row + 2]);
                        f3 t1 = base, t2 = base;
                        if (sl >= 0) {
                            if (R.dx) { t1.x += R.dx[3 * sl]; t1.y += R.dx[3 * sl + 1]; t1.z += R.dx[3 * sl + 2]; }
                            if (R.flow_dx2) { t2.x += R.flow_dx2[3 * sl]; t2.y += R.flow_dx2[3 * sl + 1]; t2.z += R.flow_dx2[3 * sl + 2]; }
                        }
                        float u1, v1, u2, v2;
                        flow_ndc(R.flow_proj1, t1, u1, v1);
                        flow_ndc(R.flow_proj2, t2, u2, v2);
                        col = mk3(u2 - u1, v2 - v1, sl >= 0 ? 1.0f : 0.0f);
                        a.clamped[idx] = 0;
                    } else if (a.colors_precomp == nullptr && sh_rows) {
                        need_sh = true; sh_pos = p; sh_dc[0] = dc[0]; sh_dc[1] = dc[1]; sh_dc[2] = dc[2]; sh_rest = shv.rest;
                        col = mk3(0.f, 0.f, 0.f);                    // (q2 is written by the wave-cooperative part below)
                    } else if (a.colors_precomp == nullptr) {
                        uint32_t cb;
                        col = sh_to_rgb(a.D, ShRegs{dc[0], dc[1], dc[2], shv.rest}, p, mk3(a.cam_pos[0], a.cam_pos[1], a.cam_pos[2]), cb);
                        a.clamped[idx] = (uint8_t)cb;
                    } else {
                        col = mk3(dc[0], dc[1], dc[2]);
                    }
                    TileRec* const rec = a.rec + idx;
                    rec->q0 = make_float4(px, py, p_view.z, opac);
                    rec->q1 = make_float4(cv.c * det_inv, -cv.b * det_inv, cv.a * det_inv, 0.f);
                    if (!need_sh) rec->q2 = make_float4(col.x, col.y, col.z, 0.f);
                    my_radius = (int)rad_f;
                    touched = (uint32_t)area;
                    rx0 = x0; ry0 = y0; rw = x1 - x0;
                }
            }
        }
        a.radii[idx] = my_radius;
        a.tiles_touched[idx] = touched;
    }
    if (sh_rows) {
        // forward.cu:22-73 with the coefficient rows moved by the wave: columns 3 .. 26 (degrees 1, 2), then 27 .. 47 (degree 3); per channel the
        // same chain of operations as sh_to_rgb
        const unsigned long long rows = __ballot(need_sh);
        if (sh_early && !rows) stage_rows_wait();          // (nothing visible in this wave: the early rows must still have landed before the window is reused)
        if (rows) {
            float* const win = sh_win;
            const int deg = a.D;
            const f3 dir = mk3(sh_pos.x - a.cam_pos[0], sh_pos.y - a.cam_pos[1], sh_pos.z - a.cam_pos[2]);
            const float inv = 1.0f / sqrtf(dot3(dir, dir));
            const float x = dir.x * inv, y = dir.y * inv, z = dir.z * inv;
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            float res[3] = {0.f, 0.f, 0.f};
            if (sh_early) stage_rows_wait(); else stage_rows(win, rows, sh_rest, 0, deg == 1 ? 9 : 24);
            if (need_sh) {
                const float* sh = win + lane * SH_WIN_STRIDE - 3;          // sh[k] = coefficient k of this Gaussian, k >= 3
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float v = SH_C0 * sh_dc[k];
                    v = v - SH_C1 * y * sh[3 + k] + SH_C1 * z * sh[6 + k] - SH_C1 * x * sh[9 + k];
                    if (deg > 1)
                        v = v + SH_C2[0] * xy * sh[12 + k] + SH_C2[1] * yz * sh[15 + k] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + k] +
                            SH_C2[3] * xz * sh[21 + k] + SH_C2[4] * (xx - yy) * sh[24 + k];
                    res[k] = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (deg > 2) {
                stage_rows(win, rows, sh_rest, 24, 21);
                if (need_sh) {
                    const float* sh = win + lane * SH_WIN_STRIDE - 27;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        float v = res[k];
                        v = v + SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + k] + SH_C3[1] * xy * z * sh[30 + k] +
                            SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + k] + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + k] +
                            SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + k] + SH_C3[5] * z * (xx - yy) * sh[42 + k] +
                            SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + k];
                        res[k] = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (need_sh) {
#pragma unroll
                for (int k = 0; k < 3; k++) res[k] = res[k] + 0.5f;
                a.clamped[idx] = (uint8_t)((res[0] < 0 ? 1u : 0u) | (res[1] < 0 ? 2u : 0u) | (res[2] < 0 ? 4u : 0u));  // forward.cu:69-71
                a.rec[idx].q2 = make_float4(fmaxf(res[0], 0.f), fmaxf(res[1], 0.f), fmaxf(res[2], 0.f), 0.f);
            }
        }
    }
    PRE_TICK(g_pre_timing, 2);
    // (a) per-tile histogram: every (Gaussian, tile) instance adds one to its tile's counter.
    auto count_tile = [&](auto, int, uint32_t, int tx, int ty) {
        if (lds_hist) atomicAdd(&s_hist[ty * a.gx + tx], 1u);
        else atomicAdd(&a.tile_count[(size_t)(ty * a.gx + tx) * CTR_STRIDE], 1u);
    };
    wave_visit_small(touched, rx0, ry0, max(1, rw), count_tile);
    wave_visit_large(touched, rx0, ry0, max(1, rw), count_tile);
    PRE_TICK(g_pre_timing, 3);
    // (b) block partial sum
    uint32_t s = touched;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) s_wave_sum[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < GB / 64; w++) tot += s_wave_sum[w];
        a.block_sums[blockIdx.x] = tot;
    }
    // (c) publish this block's histogram row; tile_offsets_kernel turns every column into an exclusive prefix over the
    //     blocks (= the block's sub-range inside the tile's segment) and a column total. Plain coalesced stores: an earlier
    //     version reserved the sub-ranges with one returning atomic per (block, tile), and those ~215 k agent-scope atomics
    //     (executed at the memory side on this multi-XCD part) cost more than the rest of the kernel.
    PRE_TICK(g_pre_timing, 4);
    if (lds_hist) {
        uint32_t* row = a.block_tile_base + (size_t)blockIdx.x * T;
        for (int t = threadIdx.x; t < T; t += GB) row[t] = s_hist[t];
    }
    PRE_TICK(g_pre_timing, 5);
}

// F1b: column scan of the [nblocks][T] histogram: hist[b][t] <- sum_{b' < b} hist[b'][t], tile_count[t] <- column total.
// 16 columns x SEGS row segments per block; a wave reads 4 x 64 contiguous bytes per row. SEGS = 16 (256 threads) up to 256 Gaussian
// blocks; SEGS = 64 (1024 threads) beyond, so that a thread's rows still fit in registers (32 x 64 = 2048 blocks = 2.1 M Gaussians:
// BASELINE config #5 has 1954) -- the serial fall-back loop took 40 us there.
constexpr int TO_COLS = 16, TO_SEGS = 16, TO_SEGS_BIG = 64;
template <int SEGS, int RMAX>
__device__ __forceinline__ void tile_offsets_body(int nblocks, int T, uint32_t* __restrict__ hist,
                                                                      uint32_t* __restrict__ tile_count, uint32_t* __restrict__ dense_total = nullptr)
{
    __shared__ uint32_t s_seg[SEGS][TO_COLS];
    const int c = threadIdx.x & (TO_COLS - 1), seg = threadIdx.x / TO_COLS;
    const int col = blockIdx.x * TO_COLS + c;
    const int rows_per_seg = (nblocks + SEGS - 1) / SEGS;
    const int r0 = seg * rows_per_seg, r1 = min(nblocks, r0 + rows_per_seg);
    const bool in_regs = rows_per_seg <= RMAX;       // rows a thread keeps in registers: one pass over the matrix
    uint32_t v[RMAX];
    uint32_t sum = 0;
    if (col < T) {
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < RMAX; k++) v[k] = r0 + k < r1 ? hist[(size_t)(r0 + k) * T + col] : 0u;   // independent loads, all in flight
#pragma unroll
            for (int k = 0; k < RMAX; k++) sum += v[k];
        } else {
            for (int r = r0; r < r1; r++) sum += hist[(size_t)r * T + col];
        }
    }
    s_seg[seg][c] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (int k = 0; k < SEGS; k++) { const uint32_t x = s_seg[k][c]; if (k < seg) run += x; total += x; }
    if (col < T) {
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < RMAX; k++) {
                if (r0 + k < r1) hist[(size_t)(r0 + k) * T + col] = run;
                run += v[k];
            }
        } else {
            for (int r = r0; r < r1; r++) {
                uint32_t* p = &hist[(size_t)r * T + col];
                const uint32_t x = *p;
                *p = run;
                run += x;
            }
        }
        if (seg == 0) {
            tile_count[(size_t)col * CTR_STRIDE] = total;
            if (dense_total != nullptr) dense_total[col] = total;      // (contiguous copy for the scatter launch's own scan: ScanInScatter)
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// F2: one block. (1) exclusive scan of the per-block sums -> block_base, R; (2) exclusive scan of the per-tile instance
// counts -> tile ranges. A tile whose list exceeds the LDS sort capacity gets a power-of-two sized segment so that the
// in-place global-memory bitonic sort can run on it (padding is pre-filled with ~0 keys); header[2] = allocated length.
// ------------------------------------------------------------------------------------------------------------------
constexpr int SORT_LDS_CAP = 4096;  // keys per tile sorted in LDS (32 KiB)

__device__ __forceinline__ uint32_t next_pow2(uint32_t v)
{
    return v <= 1 ? 1u : 1u << (32 - __clz((int)(v - 1)));
}

template <typename LOAD, typename STORE>
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(int n, LOAD load, STORE store, uint32_t* s_tmp /*[17]*/)
{
    // sequential chunks of 1024 with a running carry; returns the grand total (valid in every thread). Used by the k-NN grid.
    uint32_t carry = 0;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < n ? load(i) : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane == 63) s_tmp[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (int w = 0; w < 16; w++) { const uint32_t t = s_tmp[w]; s_tmp[w] = run; run += t; }
            s_tmp[16] = run;
        }
        __syncthreads();
        if (i < n) store(i, carry + s_tmp[wave] + incl - v, v);
        carry += s_tmp[16];
        __syncthreads();
    }
    return carry;
}

// n <= 2048, 1024 threads, ONE block barrier: thread t takes elements 2t and 2t + 1; the sixteen wave totals go through LDS and every thread adds up
// the ones in front of its wave itself (sixteen broadcast reads) instead of waiting for a serial pass of thread 0 behind a second barrier.
// extra[0 .. 15], extra2[0 .. 15] (optional, LDS): per-wave values the caller wants every thread to see behind the same barrier. Returns the total.
template <typename LOAD, typename STORE>
__device__ __forceinline__ uint32_t block_exclusive_scan_2048_once(int n, LOAD load, STORE store, uint32_t* s_tot /*[16]*/)
{
    const int t = threadIdx.x, lane = lane_id(), wave = t >> 6;
    const int i0 = 2 * t, i1 = i0 + 1;
    const uint32_t v0 = i0 < n ? load(i0) : 0u, v1 = i1 < n ? load(i1) : 0u;
    const uint32_t incl = wave_inclusive_scan(v0 + v1);
    if (lane == 63) s_tot[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const uint32_t x = s_tot[w]; base += w < wave ? x : 0u; total += x; }
    const uint32_t e0 = base + incl - (v0 + v1);
    if (i0 < n) store(i0, e0, v0);
    if (i1 < n) store(i1, e0 + v0, v1);
    return total;
}

__device__ __forceinline__ unsigned long long wave_inclusive_scan64(unsigned long long v)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)v, d, 64), hi = __shfl_up((uint32_t)(v >> 32), d, 64);
        if (lane >= d) v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// One pass over max(nblocks, ntiles) elements, 1024 at a time, carries three exclusive scans at once: the per-block instance
// sums (32 bit) and, packed into one 64-bit value, the (padded) tile counts (low word) and the tile chunk counts (high word).
// Three separate block scans cost 15 block barriers and three rounds of dependent loads on the critical path of every frame.
__device__ __forceinline__ void scan_body(int nblocks, const uint32_t* block_sums, uint32_t* block_base,
                                                    int ntiles, const uint32_t* tile_count, uint2* ranges, uint32_t* tile_cursor,
                                                    const uint32_t* flags, uint32_t cap_R, uint32_t cap_tile_list,
                                                    uint32_t* chunk_base, uint32_t* header, uint32_t* host_mailbox, uint32_t seq)
{
    __shared__ uint32_t s_a[17];
    __shared__ unsigned long long s_b[17];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t carry_a = 0, mx = 0;
    unsigned long long carry_b = 0;
    const int n = max(nblocks, ntiles);
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t va = i < nblocks ? block_sums[i] : 0u;
        const uint32_t cnt = i < ntiles ? tile_count[(size_t)i * CTR_STRIDE] : 0u;
        const uint32_t padded = cnt;   // segments are exact: lists beyond the LDS sort are sorted chunk-wise + ranked (sort_long_* kernels), no padding
        const unsigned long long vb = ((unsigned long long)((cnt + (uint32_t)CHUNK - 1) / (uint32_t)CHUNK) << 32) | padded;
        mx = max(mx, cnt);
        const uint32_t incl_a = wave_inclusive_scan(va);
        const unsigned long long incl_b = wave_inclusive_scan64(vb);
        if (lane == 63) { s_a[wave] = incl_a; s_b[wave] = incl_b; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (int w = 0; w < 16; w++) { const uint32_t t = s_a[w]; s_a[w] = run; run += t; }
            s_a[16] = run;
        } else if (threadIdx.x == 64) {
            unsigned long long run = 0;
            for (int w = 0; w < 16; w++) { const unsigned long long t = s_b[w]; s_b[w] = run; run += t; }
            s_b[16] = run;
        }
        __syncthreads();
        if (i < nblocks) block_base[i] = carry_a + s_a[wave] + incl_a - va;
        if (i < ntiles) {
            const unsigned long long ex = carry_b + s_b[wave] + incl_b - vb;
            const uint32_t first = (uint32_t)ex;
            ranges[i] = make_uint2(first, first + cnt);
            tile_cursor[(size_t)i * CTR_STRIDE] = first;
            chunk_base[i] = (uint32_t)(ex >> 32);          // number of CHUNK-entry pieces in front of tile i (render_bwd's work items)
        }
        carry_a += s_a[16];
        carry_b += s_b[16];
        __syncthreads();
    }
    const uint32_t R = carry_a, R_alloc = (uint32_t)carry_b, nchunks = (uint32_t)(carry_b >> 32);
    if (threadIdx.x == 0) chunk_base[ntiles] = nchunks;
    // largest tile list: lets the host pick the sort kernel variant (LDS footprint decides how many tiles sort concurrently)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    __shared__ uint32_t s_mx[16];
    if (lane_id() == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) mx = max(mx, s_mx[w]);
        // cap_R != 0: the binning kernels were enqueued behind this one on a buffer sized for cap_R instances and tile lists of
        // at most cap_tile_list entries; if this frame needs more they must not run (they test FLAG_OVERFLOW) and the host redoes them.
        uint32_t err = flags ? __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;   // flags == nullptr: nothing can have raised one
        if (cap_R && (R > cap_R || R_alloc > cap_R || mx > cap_tile_list)) err |= (uint32_t)FLAG_OVERFLOW;
        header[HDR_R] = R; header[HDR_FLAGS] = err; header[HDR_R_ALLOC] = R_alloc; header[HDR_MAX_TILE] = mx; header[HDR_CARVE_R] = cap_R;
        header[HDR_CAP_SORTED] = cap_R; header[HDR_CHUNKS] = nchunks;
        // Host mailbox (pinned, host-coherent): the host spins on word 4 instead of paying a hipMemcpyAsync + a blocking
        // hipStreamSynchronize (whose wake-up alone left the GPU idle for ~60 us per forward pass).
        if (host_mailbox) {
            __hip_atomic_store(&host_mailbox[0], R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_mailbox[1], err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_mailbox[2], R_alloc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_mailbox[3], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (err & (uint32_t)FLAG_OVERFLOW)   // sticky count of frames that outgrew their speculative buffer (lazy mode polls it)
                __hip_atomic_fetch_add(&host_mailbox[5], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(&host_mailbox[4], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// F2b (round 6): the scan INSIDE the scatter launch. On a speculative frame (the steady state: the binning buffer was laid out from the
// previous frame, nobody waits for the instance count before the binning is enqueued) the one-block scan_kernel above was a launch of its
// own between tile_offsets and scatter: 7.7 us of a 210 us step for a 1200-element prefix sum, plus a kernel boundary. Every scatter block
// needs the tiles' first positions and the instances in front of its Gaussians; both are prefix sums of data that tile_offsets and
// preprocess_fwd have already written, so every scatter block now forms them ITSELF (a 1200-element block scan in LDS and a 196-element
// masked sum: ~1.5 us, in parallel on all blocks, no communication between blocks), and the launch's extra block -- the one that ranks
// pieces and tiles (F3b, F3c) -- writes what the rest of the frame reads from memory: ranges, chunk_base, block_base, the header and the
// host mailbox, exactly as scan_body does. The non-speculative path (first frame, or the redo of a frame that outgrew its buffer), where the
// host needs the instance count BEFORE it can lay the buffer out, keeps scan_kernel.
// ------------------------------------------------------------------------------------------------------------------
struct ScanInScatter {
    const uint32_t* dense_total;     // [T] tile list lengths, contiguous (tile_offsets_body); nullptr: the scan ran as its own launch
    const uint32_t* block_sums; int nblocks;
    uint2* ranges; uint32_t* block_base; uint32_t* chunk_base;
    const uint32_t* flags; uint32_t cap_R, cap_tile_list; uint32_t* host_mailbox; uint32_t seq;
};
// the extra block: everything scan_body writes, from the dense totals. s_n [T] (LDS) is left holding the list lengths; returns the longest
// list in `longest` and whether the frame fits its speculative buffer.
__device__ __forceinline__ bool scan_in_scatter_bookkeeping(int T, const ScanInScatter& m, uint32_t* s_n, uint32_t* __restrict__ header, uint32_t& longest)
{
    __shared__ uint32_t s_tmp[17], s_mx, s_t1[16], s_t2[16], s_t3[16];
    if (threadIdx.x == 0) s_mx = 0;
    uint32_t mx = 0, R, nchunks;
    auto load_n = [&](int i) { const uint32_t v = m.dense_total[i]; s_n[i] = v; mx = max(mx, v); return v; };
    auto store_range = [&](int i, uint32_t excl, uint32_t v) { m.ranges[i] = make_uint2(excl, excl + v); };
    auto store_chunks = [&](int i, uint32_t excl, uint32_t) { m.chunk_base[i] = excl; };
    auto load_bs = [&](int i) { return m.block_sums[i]; };
    auto store_bb = [&](int i, uint32_t excl, uint32_t) { m.block_base[i] = excl; };
    if (T <= 2048 && m.nblocks <= 2048) {            // three one-barrier scans (their wave totals in three arrays: no barrier between them either)
        const int i0 = 2 * (int)threadIdx.x;
        const uint32_t v0 = i0 < T ? load_n(i0) : 0u, v1 = i0 + 1 < T ? load_n(i0 + 1) : 0u;
        auto pick = [&](int i) { return (i & 1) ? v1 : v0; };
        R = block_exclusive_scan_2048_once(T, pick, store_range, s_t1);
        nchunks = block_exclusive_scan_2048_once(T, [&](int i) { return (pick(i) + (uint32_t)CHUNK - 1) / (uint32_t)CHUNK; }, store_chunks, s_t2);
        block_exclusive_scan_2048_once(m.nblocks, load_bs, store_bb, s_t3);
    } else {
        R = block_exclusive_scan_1024(T, load_n, store_range, s_tmp);
        nchunks = block_exclusive_scan_1024(T, [&](int i) { return (s_n[i] + (uint32_t)CHUNK - 1) / (uint32_t)CHUNK; }, store_chunks, s_tmp);
        block_exclusive_scan_1024(m.nblocks, load_bs, store_bb, s_tmp);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if (lane_id() == 0) atomicMax(&s_mx, mx);
    __syncthreads();
    mx = s_mx;
    longest = mx;
    uint32_t err = m.flags ? __hip_atomic_load(m.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (m.cap_R && (R > m.cap_R || mx > m.cap_tile_list)) err |= (uint32_t)FLAG_OVERFLOW;
    if (threadIdx.x == 0) {
        m.chunk_base[T] = nchunks;
        header[HDR_R] = R; header[HDR_FLAGS] = err; header[HDR_R_ALLOC] = R; header[HDR_MAX_TILE] = mx; header[HDR_CARVE_R] = m.cap_R;
        header[HDR_CAP_SORTED] = m.cap_R; header[HDR_CHUNKS] = nchunks;
        if (m.host_mailbox) {        // (as scan_body: pinned, host-coherent)
            __hip_atomic_store(&m.host_mailbox[0], R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&m.host_mailbox[1], err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&m.host_mailbox[2], R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&m.host_mailbox[3], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (err & (uint32_t)FLAG_OVERFLOW) __hip_atomic_fetch_add(&m.host_mailbox[5], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(&m.host_mailbox[4], m.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    return !(err & (uint32_t)FLAG_OVERFLOW);
}

// ------------------------------------------------------------------------------------------------------------------
// F3: scatter every (Gaussian, tile) instance into its tile's segment (replaces duplicateWithKeys,
// rasterizer_impl.cu:70-111). Instance id u = point_offsets_exclusive[g] + k is the reference's position in the
// unsorted duplicate list; since u grows with g, sorting by (depth bits, u) reproduces the reference's stable
// (tile | depth) sort order (rasterizer_impl.cu:98-108,306-311), ties included.
// Also materialises the global inclusive scan point_offsets (rasterizer_impl.cu:280).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void scatter_instances_body(int P, int gx, int gy, const int* radii, const TileRec* rec,
                                                               const uint32_t* tiles_touched,
                                                               const uint32_t* block_base, uint32_t* point_offsets,
                                                               uint32_t* tile_cursor, const uint2* ranges, const uint32_t* block_tile_base,
                                                               uint64_t* keys, uint32_t* inst_gauss, uint32_t* header, int speculative,
                                                               uint32_t carve_R, uint32_t cap_sorted, int eager, const int* clip = nullptr,
                                                               const ScanInScatter* scan = nullptr)
{
    const bool merged = scan != nullptr && scan->dense_total != nullptr;     // F2b: this launch does the scan's work (always a speculative frame)
    if (merged) {
        // (the header of this frame is being written by the launch's extra block: the overflow test is made from the block's own sums below)
    } else if (speculative) {
        if (header[HDR_FLAGS] & FLAG_OVERFLOW) return;          // uniform: the buffer behind keys/inst_gauss is too small for this frame
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        header[HDR_CARVE_R] = carve_R;                          // the host waited for R and laid the buffer out for exactly R
        header[HDR_CAP_SORTED] = cap_sorted;
        header[HDR_FLAGS] &= ~(uint32_t)FLAG_OVERFLOW;          // this IS the redo of a frame that outgrew its speculative buffer: the
                                                                // backward kernels must not mistake it for a lazy frame without lists
    }
    PRE_TICK(g_sca_timing, 0);
    const int idx = blockIdx.x * GB + threadIdx.x;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    __shared__ uint32_t s_wave_sum[GB / 64];
    extern __shared__ uint32_t s_pos[];   // [T] next free position of this block inside each tile's segment (LDS path)
    const int T = gx * gy;
    const bool lds_path = block_tile_base != nullptr;
    const uint32_t cnt = idx < P ? tiles_touched[idx] : 0u;
    // eager (small launches, latency-bound): the Gaussian's record and radius are requested together with its instance count instead of
    // behind it (rows of culled Gaussians hold no data: loaded, never used)
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f);
    int my_radius = 0;
    if (eager && idx < P) { q0 = rec[idx].q0; my_radius = radii[idx]; }
    const uint32_t incl = wave_inclusive_scan(cnt);
    if (lane == 63) s_wave_sum[wave] = incl;
    __shared__ uint32_t s_red[2], s_scan_tmp[17], s_wmx[16], s_wbefore[16];
    uint32_t merged_before = 0;          // F2b: instances of the Gaussian blocks in front of this one
    if (merged) {
        // F2b: first position of every tile = exclusive scan of the list lengths (every block forms it for itself), + this block's offset inside the
        // tile's segment; the instances in front of this block's Gaussians = the sum of the earlier blocks' counts; the overflow test from the
        // same sums the extra block writes into the header. Up to 2048 tiles all of it sits behind ONE block barrier (the one this kernel had).
        const uint32_t* row = block_tile_base + (size_t)blockIdx.x * T;
        uint32_t mx = 0, before = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += GB) before += scan->block_sums[b];
        auto load_total = [&](int i) { const uint32_t v = scan->dense_total[i]; mx = max(mx, v); return v; };
        auto store_pos = [&](int i, uint32_t excl, uint32_t) { s_pos[i] = excl + row[i]; };
        uint32_t R;
        if (T <= 2048) {
            // (the per-wave maxima and sums ride behind the scan's barrier: the loads above are issued first, their values reduced here)
            const uint32_t v0 = 2 * (int)threadIdx.x < T ? load_total(2 * (int)threadIdx.x) : 0u, v1 = 2 * (int)threadIdx.x + 1 < T ? load_total(2 * (int)threadIdx.x + 1) : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64)); before += __shfl_xor(before, d, 64); }
            if (lane == 0) { s_wmx[wave] = mx; s_wbefore[wave] = before; }
            R = block_exclusive_scan_2048_once(T, [&](int i) { return (i & 1) ? v1 : v0; }, store_pos, s_scan_tmp);
            mx = 0; before = 0;
#pragma unroll
            for (int w = 0; w < 16; w++) { mx = max(mx, s_wmx[w]); before += s_wbefore[w]; }
        } else {
            if (threadIdx.x < 2) s_red[threadIdx.x] = 0;
            R = block_exclusive_scan_1024(T, load_total, store_pos, s_scan_tmp);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64)); before += __shfl_xor(before, d, 64); }
            if (lane == 0) { atomicMax(&s_red[0], mx); if (before) atomicAdd(&s_red[1], before); }
            __syncthreads();
            mx = s_red[0]; before = s_red[1];
        }
        merged_before = before;
        if (scan->cap_R && (R > scan->cap_R || mx > scan->cap_tile_list)) return;      // uniform: the frame does not fit its speculative buffer
        __syncthreads();                                                                // s_pos and s_wave_sum are complete
    } else {
        if (lds_path) {
            const uint32_t* row = block_tile_base + (size_t)blockIdx.x * T;
            for (int t = threadIdx.x; t < T; t += GB) s_pos[t] = ranges[t].x + row[t];
        }
        __syncthreads();
    }
    PRE_TICK(g_sca_timing, 1);
    uint32_t wbase = merged ? merged_before : block_base[blockIdx.x];
    for (int w = 0; w < wave; w++) wbase += s_wave_sum[w];
    const uint32_t off_incl = wbase + incl;
    if (idx < P) point_offsets[idx] = off_incl;
    int rx0 = 0, ry0 = 0, rw = 1;
    uint32_t dbits = 0;
    if (cnt) {
        if (!eager) { q0 = rec[idx].q0; my_radius = radii[idx]; }
        int x1, y1;
        tile_rect(q0.x, q0.y, my_radius, gx, gy, rx0, ry0, x1, y1);
        if (clip) { rx0 = max(rx0, clip[0]); ry0 = max(ry0, clip[1]); x1 = min(x1, clip[2]); }      // (a flow view's rectangle: as in preprocess_fwd_body)
        rw = x1 - rx0;
        dbits = __float_as_uint(q0.z);
    }
    const uint32_t off_excl = off_incl - cnt;
    PRE_TICK(g_sca_timing, 2);
    auto emit = [&](int src, uint32_t sd, uint32_t u, int tx, int ty) {
        const int g = (blockIdx.x * GB + (wave << 6)) + src;
        const uint32_t pos = lds_path ? atomicAdd(&s_pos[ty * gx + tx], 1u)
                                      : atomicAdd(&tile_cursor[(size_t)(ty * gx + tx) * CTR_STRIDE], 1u);
        keys[pos] = ((uint64_t)sd << 32) | (uint64_t)u;
        inst_gauss[u] = (uint32_t)g;
    };
    // Gaussians of few tiles through the lane-parallel walk (consecutive lanes write consecutive inst_gauss entries), the large ones
    // one at a time with their fields in scalar registers (gs_device.h)
    wave_expand(cnt <= INSTANCES_SMALL ? cnt : 0u, [&](int src, uint32_t k, bool active) {
        const int sx0 = __shfl(rx0, src, 64), sy0 = __shfl(ry0, src, 64), sw = max(1, __shfl(rw, src, 64));
        const uint32_t sd = __shfl(dbits, src, 64), so = __shfl(off_excl, src, 64);
        if (active) emit(src, sd, so + k, sx0 + (int)(k % (uint32_t)sw), sy0 + (int)(k / (uint32_t)sw));
    });
    wave_visit_large(cnt, rx0, ry0, max(1, rw), [&](auto tag, int src, uint32_t k, int tx, int ty) {
        emit(src, of_source(tag, dbits, src), of_source(tag, off_excl, src) + k, tx, ty);
    });
    PRE_TICK(g_sca_timing, 3);
}

// ------------------------------------------------------------------------------------------------------------------
// F3b + F3c: ONE extra block of the scatter launch (it only needs the tiles' list lengths, which the scan wrote; it runs beside the scatter
// blocks and is done before them: no launch of its own, nothing on the critical path) decides
//  (b) where render_bwd's work items go (gs_device.h: item_block_*) -- per tile: the number of FULL pieces (CHUNK entries) of the tiles in
//      front of it, and the rank of its PARTIAL last piece among all partial pieces of the frame, longest first;
//  (c) which tile a render_fwd block takes. Every tile of a 640x480 frame is resident at once (1200 blocks, 4.7 per CU) and the launch lasts
//      as long as its slowest block; a block is slow when its CU is crowded -- measured (tools/tile_timeline.py --raw): the dispatcher deals
//      block b of XCD b % 8 to that XCD's CU (b / 8) % 32, so the first T / 8 - 128 CUs of an XCD host five blocks and the rest four; blocks
//      last 48.7 us on average on the former, 45.4 on the latter, and the longest block of the launch always sat on a five-block CU. The tiles
//      of an XCD's band (the band itself stays: neighbouring tiles share Gaussians, same L2) are dealt to the CUs by list length: longest
//      first, in a serpentine over the 32 CUs whose first pass runs from the LAST CU down (the longest lists land on the CUs that get one
//      block less) and whose last, partial pass hands the shortest lists to the CUs that get one more.
// Both are counting sorts in LDS (lengths 1 .. CHUNK - 1; list length >> shift in 256 buckets per band); inside a bucket the order comes from an
// LDS atomic and is not fixed -- it only decides which block takes which piece / tile, observed placement is used for speed only, and every
// piece / tile writes its own outputs: the results do not depend on it.
// ------------------------------------------------------------------------------------------------------------------
constexpr int XCD_CUS = 32, ORDER_BUCKETS = 256;
constexpr int ORDER_FWD_MIN_BAND = 2 * XCD_CUS;      // below: at most two blocks per CU, nothing to balance
__device__ __forceinline__ void order_tiles_body(int T, const uint2* __restrict__ ranges, uint32_t* s_n /* [T], LDS */, uint32_t* __restrict__ tile_pos,
                                                 const bool order_fwd, const uint32_t longest_list, const uint32_t heavy)
{
    static_assert(CHUNK == 128 && GB == 1024, "128 lengths, 1024 threads");
    // s_len[x][m]: partial pieces of length m whose tile lies in XCD band x -- a tile's partial piece runs on the XCD of its band of tiles (where
    // most of the tile's full pieces ran: dealt round robin over all XCDs, the partial pieces cost render_bwd 7 MB more fabric traffic per frame;
    // placing them by the XCD of the tile's own full pieces instead of by its band was measured too: the same traffic, and the extra table
    // reads made this block the last one of the scatter launch). Row 8: all bands together, for the round-robin fallback.
    __shared__ uint32_t s_len[9][CHUNK], s_tmp[17], s_band[8][ORDER_BUCKETS], s_fs[10];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const int q = T >> 3, r = T & 7, head = r * (q + 1);
    auto band_of = [&](int i) { return i < head ? i / (q + 1) : r + (i - head) / q; };
    const int shift = max(0, 32 - __clz((int)max(longest_list, 1u)) - 8);                  // longest list >> shift < 256
    for (uint32_t k = t; k < 9u * CHUNK; k += GB) (&s_len[0][0])[k] = 0;
    for (uint32_t k = t; k < 8u * ORDER_BUCKETS; k += GB) (&s_band[0][0])[k] = 0;
    if (ranges != nullptr) for (int i = (int)t; i < T; i += GB) { const uint2 g = ranges[i]; s_n[i] = g.y - g.x; }      // (nullptr: s_n holds the lengths already)
    __syncthreads();
    // the number of FULL pieces (CHUNK entries) in front of every tile, and their total
    const uint32_t total_full = block_exclusive_scan_1024(T, [&](int i) { return s_n[i] / (uint32_t)CHUNK; },
                                                          [&](int i, uint32_t excl, uint32_t) { tile_pos[(size_t)i * CTR_STRIDE + POS_FULL_BASE] = excl; }, s_tmp);
    for (int i = (int)t; i < T; i += GB) {
        const uint32_t n = s_n[i], m = n & (uint32_t)(CHUNK - 1);
        if (m) { atomicAdd(&s_len[band_of(i)][m], 1u); atomicAdd(&s_len[8][m], 1u); }
        if (order_fwd) atomicAdd(&s_band[band_of(i)][ORDER_BUCKETS - 1 - min((uint32_t)ORDER_BUCKETS - 1u, n >> shift)], 1u);
    }
    __syncthreads();
    // exclusive scans, descending length. Waves 0-8: one row of s_len each (lane l holds lengths 127 - 2 l and 126 - 2 l); waves 2-9 ALSO one
    // band's list-length buckets each (bucket index = 255 - key: ascending index is descending length; four consecutive buckets per lane)
    if (wave < 9) {
        uint32_t* const row = s_len[wave];
        const uint32_t c0 = row[CHUNK - 1 - 2 * lane], c1 = row[CHUNK - 2 - 2 * lane];
        const uint32_t incl = wave_inclusive_scan(c0 + c1);
        row[CHUNK - 1 - 2 * lane] = incl - c0 - c1; row[CHUNK - 2 - 2 * lane] = incl - c1;          // first rank of every length inside the band
        if (lane == 63) s_tmp[wave] = incl;                                                         // partial pieces of the band (row 8: of the frame)
    }
    if (wave >= 2 && wave < 10 && order_fwd) {
        uint32_t* const b = &s_band[wave - 2][4 * lane];
        const uint32_t c0 = b[0], c1 = b[1], c2 = b[2], c3 = b[3];
        const uint32_t base = wave_inclusive_scan(c0 + c1 + c2 + c3) - (c0 + c1 + c2 + c3);
        b[0] = base; b[1] = base + c0; b[2] = base + c0 + c1; b[3] = base + c0 + c1 + c2;
    }
    __syncthreads();
    // full-piece ranges of the XCDs: XCD x runs #{b < N : b % 8 == x} blocks, its band's partial pieces among them, full pieces for the rest
    if (t == 0) {
        const uint32_t N = total_full + s_tmp[8];
        uint32_t fs = 0; bool home = true;
        for (uint32_t x = 0; x < 8; x++) {
            const uint32_t cnt = items_below(N, x + 1) - items_below(N, x);
            s_fs[x] = fs;
            if (s_tmp[x] > cnt) home = false; else fs += cnt - s_tmp[x];
        }
        s_fs[8] = fs; s_fs[9] = home ? 1u : 0u;                // (a band with more partial pieces than its XCD runs blocks: round robin over all XCDs)
        for (int x = 0; x < 9; x++) tile_pos[(size_t)T * CTR_STRIDE + POS_FULL_START + x] = s_fs[x];
        tile_pos[(size_t)T * CTR_STRIDE + POS_TOTAL_FULL] = total_full;
        tile_pos[(size_t)T * CTR_STRIDE + POS_PART_HOME] = s_fs[9];
    }
    __syncthreads();
    const bool home = s_fs[9] != 0;
    for (int i = (int)t; i < T; i += GB) {
        const uint32_t n = s_n[i], m = n & (uint32_t)(CHUNK - 1);
        if (m) tile_pos[(size_t)i * CTR_STRIDE + POS_PART_RANK] = atomicAdd(&s_len[home ? band_of(i) : 8][m], 1u);
        if (order_fwd) {
            const int x = band_of(i), start = x < r ? x * (q + 1) : head + (x - r) * q, size = x < r ? q + 1 : q;
            const uint32_t rank = atomicAdd(&s_band[x][ORDER_BUCKETS - 1 - min((uint32_t)ORDER_BUCKETS - 1u, n >> shift)], 1u);
            // deal: the band's CUs get `base` blocks each, the first `more` of them one more (dispatch order, observed). The `heavy` longest
            // lists per CU go to the CUs WITHOUT the extra block first, then whole passes over all CUs, and the shortest lists -- one pass more
            // than was held back -- to the CUs with the extra block; every pass a serpentine (odd passes run backwards)
            const uint32_t base = (uint32_t)size / XCD_CUS, more = (uint32_t)size % XCD_CUS, lean = (uint32_t)XCD_CUS - more, H = min(heavy, base);
            auto snake = [](uint32_t pass, uint32_t p, uint32_t n) { return (pass & 1u) ? n - 1u - p : p; };
            uint32_t c, j;
            if (rank < lean * H) { j = rank / lean; c = more + snake(j, rank % lean, lean); }
            else if (rank - lean * H < (uint32_t)XCD_CUS * (base - H)) {
                const uint32_t r2 = rank - lean * H, pass = r2 / XCD_CUS;
                c = snake(pass + H, r2 % XCD_CUS, XCD_CUS); j = c >= more ? H + pass : pass;
            } else {
                const uint32_t r3 = rank - lean * H - (uint32_t)XCD_CUS * (base - H), pass = r3 / more;
                c = snake(pass + base, r3 % more, more); j = base - H + pass;
            }
            tile_pos[(size_t)(start + (int)(j * XCD_CUS + c)) * CTR_STRIDE + POS_FWD_TILE] = (uint32_t)i;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// F4: per-tile depth sort. One 256-thread block per tile; bitonic network on 64-bit keys (depth bits << 32 | u), in
// LDS when the (power-of-two padded) list fits SORT_LDS_CAP keys, otherwise in place in global memory on the tile's
// power-of-two sized segment. Emits the sorted (gaussian id, instance id) pairs the render kernels walk.
// ------------------------------------------------------------------------------------------------------------------
// WAVE_LOCAL: with 256 threads, thread t of pass p owns the pair (i, i|j) inside the 128-key block 128*((t>>6) + 4p) whenever
// j <= 64, and that ownership is the same for every such stage; so between two consecutive stages with j <= 64 only the
// wave's own LDS traffic has to be ordered (DS operations of one wave execute in order) and the block barrier can go.
// For 1024 keys that leaves 9 block barriers instead of 55. The global-memory variant keeps a barrier per stage.
template <bool WAVE_LOCAL, typename KEYS>
__device__ __forceinline__ void bitonic_sort_block(KEYS keys, uint32_t npad)
{
    for (uint32_t k = 2; k <= npad; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < (npad >> 1); t += blockDim.x) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t p = i | j;
                const uint64_t a = keys[i], b = keys[p];
                const bool asc = (i & k) == 0;
                if ((a > b) == asc) { keys[i] = b; keys[p] = a; }
            }
            const uint32_t next_j = j > 1 ? (j >> 1) : k;   // first stage of the next merge level has j = k
            if (!WAVE_LOCAL || j > 64 || next_j > 64) {
                __syncthreads();
            } else {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (WAVE_LOCAL) __syncthreads();
}

// The same network with the size as a template parameter: every (k, j) stage is unrolled with its constants folded, which
// removes the scalar loop control that made up two thirds of the generic version's instructions (sort_tiles was issue-bound:
// 1205 SALU + 524 branch of 2706 instructions per wave). Used for the LDS sizes 64 .. 1024; 256 threads.
// LDS key array with one unused slot per 8 keys: element i lives at i + (i >> 3). Threads that each hold 8 keys a fixed stride apart
// (bitonic_sort_lds_reg below) then spread over all banks for every stride; a plain layout puts 16 lanes on one bank pair at stride 1.
struct PaddedKeys {
    uint64_t* p;
    __device__ __forceinline__ uint64_t& operator[](uint32_t i) const { return p[i + (i >> 3)]; }
    __device__ __forceinline__ PaddedKeys operator+(uint32_t off) const { return PaddedKeys{p + off + (off >> 3)}; }   // off: a multiple of 8
};
__host__ __device__ constexpr uint32_t padded_keys_size(uint32_t n) { return n + (n >> 3); }

template <uint32_t N, typename KEYS>
__device__ __forceinline__ void bitonic_sort_lds(KEYS keys)
{
#pragma unroll
    for (uint32_t k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (uint32_t t0 = 0; t0 < (N >> 1); t0 += 256) {
                const uint32_t t = t0 + threadIdx.x;
                if ((N >> 1) >= 256 || t < (N >> 1)) {
                    const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const uint32_t p = i | j;
                    const uint64_t a = keys[i], b = keys[p];
                    const bool asc = (i & k) == 0;
                    const bool swap = (a > b) == asc;
                    keys[i] = swap ? b : a;                      // unconditional stores: no divergent branch per stage
                    keys[p] = swap ? a : b;
                }
            }
            const uint32_t next_j = j > 1 ? (j >> 1) : k;
            if (j > 64 || next_j > 64) {
                __syncthreads();
            } else {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
}
// The network again, 2^MS stages per LDS round trip: a thread loads the 2^MS keys that the next MS stages of one merge level exchange
// among themselves (strides S << (MS-1) .. S), runs those stages in registers and writes the keys back. 2048 keys: 24 passes and
// block barriers instead of 66, a third of the LDS traffic, ~half the instructions. sort_tiles at 3.5 M instances: 153 -> see
// profiles/r02_long_lists.json. KEYS must be PaddedKeys (bank spread, above).
__device__ __forceinline__ void compare_exchange(uint64_t& a, uint64_t& b, bool asc)
{
    const bool swap = (a > b) == asc;
    const uint64_t lo = swap ? b : a, hi = swap ? a : b;
    a = lo; b = hi;
}
__host__ __device__ constexpr int ilog2_c(uint32_t v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }

// stages with strides S << (MS-1), ..., S of merge level K
template <uint32_t N, uint32_t K, uint32_t S, int MS>
__device__ __forceinline__ void bitonic_reg_pass(PaddedKeys keys)
{
    constexpr uint32_t G = 1u << MS, NG = N / G;
    constexpr int b = ilog2_c(S);
#pragma unroll 1
    for (uint32_t g = threadIdx.x; g < NG; g += 256) {
        const uint32_t base = ((g >> b) << (b + MS)) | (g & (S - 1));
        uint64_t v[G];
#pragma unroll
        for (uint32_t a = 0; a < G; a++) v[a] = keys[base + a * S];
        const bool asc = K >= N || (base & K) == 0;
#pragma unroll
        for (int q = MS - 1; q >= 0; q--) {
#pragma unroll
            for (uint32_t a = 0; a < G; a++)
                if (!(a & (1u << q))) compare_exchange(v[a], v[a | (1u << q)], asc);
        }
#pragma unroll
        for (uint32_t a = 0; a < G; a++) keys[base + a * S] = v[a];
    }
    __syncthreads();
}
template <uint32_t N, int M, uint32_t K, int REM>   // REM: stages of level K still to run (strides 2^(REM-1) .. 1)
__device__ __forceinline__ void bitonic_reg_level(PaddedKeys keys)
{
    if constexpr (REM > 0) {
        constexpr int ms = (REM % M) ? (REM % M) : M;
        bitonic_reg_pass<N, K, (1u << (REM - ms)), ms>(keys);
        bitonic_reg_level<N, M, K, REM - ms>(keys);
    }
}
template <uint32_t N, int M, uint32_t K>
__device__ __forceinline__ void bitonic_reg_levels(PaddedKeys keys)
{
    if constexpr (K <= N) {
        bitonic_reg_level<N, M, K, ilog2_c(K)>(keys);
        bitonic_reg_levels<N, M, K * 2>(keys);
    }
}
template <uint32_t N, int M>
__device__ __forceinline__ void bitonic_sort_lds_reg(PaddedKeys keys)
{
    constexpr uint32_t G = 1u << M, NG = N / G;
    // merge levels 2 .. G: each thread sorts G consecutive keys (ascending or descending as level G wants it) in registers
#pragma unroll 1
    for (uint32_t g = threadIdx.x; g < NG; g += 256) {
        const uint32_t base = g * G;
        uint64_t v[G];
#pragma unroll
        for (uint32_t a = 0; a < G; a++) v[a] = keys[base + a];
#pragma unroll
        for (uint32_t k = 2; k <= G; k <<= 1) {
#pragma unroll
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
                for (uint32_t a = 0; a < G; a++)
                    if (!(a & j)) compare_exchange(v[a], v[a | j], k < G ? (a & k) == 0 : (G >= N || (base & G) == 0));
            }
        }
#pragma unroll
        for (uint32_t a = 0; a < G; a++) keys[base + a] = v[a];
    }
    __syncthreads();
    bitonic_reg_levels<N, M, 2 * G>(keys);
}

// MAXN: largest size this call site can see. The unrolled 2048 / 4096 networks are ~20 KB of code each: instantiated inside
// render_fwd_kernel (whose fused sort never exceeds 1024 keys) they cost that kernel 25 % through instruction-cache misses.
template <uint32_t MAXN = 4096>
__device__ __forceinline__ void bitonic_sort_lds_pow2(uint64_t* keys, uint32_t npad)
{
    switch (npad) {
        case 1024: bitonic_sort_lds<1024>(keys); break;
        case 512: bitonic_sort_lds<512>(keys); break;
        case 256: bitonic_sort_lds<256>(keys); break;
        case 128: bitonic_sort_lds<128>(keys); break;
        case 64: bitonic_sort_lds<64>(keys); break;
        default: bitonic_sort_block<true>(keys, npad); break;   // < 64
    }
}
// the padded-layout variant of the kernels that see lists beyond 1024 keys: register-blocked networks from 1024 keys on
template <uint32_t MAXN = 4096>
__device__ __forceinline__ void bitonic_sort_lds_pow2(PaddedKeys keys, uint32_t npad)
{
    switch (npad) {
        case 4096: bitonic_sort_lds_reg<4096, 3>(keys); break;
        case 2048: bitonic_sort_lds_reg<2048, 3>(keys); break;
        case 1024: bitonic_sort_lds_reg<1024, 2>(keys); break;
        case 512: bitonic_sort_lds<512>(keys); break;
        case 256: bitonic_sort_lds<256>(keys); break;
        case 128: bitonic_sort_lds<128>(keys); break;
        case 64: bitonic_sort_lds<64>(keys); break;
        default: bitonic_sort_block<true>(keys, npad); break;   // < 64
    }
}

// ---- round 4: the network in REGISTERS, partners fetched across lanes ---------------------------------------------------------------
// Every stage of the LDS networks above is a round trip through LDS (write, wait, read, wait) and most of them a block barrier; with every
// tile of the frame sorting at the same moment nothing hides that latency, and the phase is a third of render_fwd's wave lifetime
// (profiles/r03_phase_cycles.json: 19.5 k of 104 k cycles for ~520 keys). Here a wave OWNS a contiguous block of 64 R keys (R = N / 256 keys
// per lane, key index e = 64 R wave + 64 r + lane) and keeps them in registers for the whole network:
//   stride J < 64      the partner sits in another lane of the same wave: its key comes through the LDS crossbar (ds_swizzle xor-J inside
//                      32 lanes, ds_bpermute for J = 32) -- no LDS memory, no barrier -- and BOTH lanes of a pair compare; one keeps the
//                      smaller key, the other the larger (mask = compare result XOR a per-stage constant: literal lane masks);
//   stride 64 .. 32 R  both keys in this thread's registers: compare-exchange;
//   stride >= 64 R     another wave: one LDS round trip with a block barrier (3 of the 45 stages at 512 keys, 24 barriers before).
// Same comparator network, same result; keys are unique (the instance id is part of the key), padding keys (~0) compare equal among themselves
// and stay at the end.
struct Key64 { uint32_t lo, hi; };
// m ? a : b per lane with the lane mask in an SGPR pair, as the ONE v_cndmask it is (written as `(m >> lane) & 1 ? a : b` hipcc rebuilds a
// per-lane bool with five VALU instructions)
__device__ __forceinline__ uint32_t mask_select(unsigned long long m, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ unsigned long long opaque_mask(unsigned long long m)      // a constant the optimiser must keep in SGPRs
{
    asm volatile("" : "+s"(m));
    return m;
}
__device__ __forceinline__ bool key_less(const Key64 a, const Key64 b)
{
    return (((uint64_t)a.hi << 32) | a.lo) < (((uint64_t)b.hi << 32) | b.lo);
}
// value of lane (lane ^ J), VALU only (no LDS-pipe latency): DPP for the strides inside a row of 16 lanes -- quad_perm for 1 and 2, row_ror:8
// for 8, and 4 = row_half_mirror (lane ^ 7) followed by a reversed quad (lane ^ 3) --, gfx950's half-wave / row swaps for 16 and 32
template <uint32_t J>
__device__ __forceinline__ uint32_t lane_xor_value(uint32_t v)
{
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "lane stride");
    if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);          // quad_perm:[1,0,3,2]
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);     // quad_perm:[2,3,0,1]
    else if constexpr (J == 4) {
        const int m = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);                              // row_half_mirror
        return (uint32_t)__builtin_amdgcn_update_dpp(0, m, 0x1B, 0xf, 0xf, false);                                 // quad_perm:[3,2,1,0]
    } else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);  // row_ror:8
    else if constexpr (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);      // r[0] = rows {0, 0, 2, 2}, r[1] = rows {1, 1, 3, 3}
        return mask_select(opaque_mask(0xFFFF0000FFFF0000ull), r[0], r[1]);         // odd rows read r[0]
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);      // r[0] = halves {0, 0}, r[1] = halves {1, 1}
        return mask_select(opaque_mask(0xFFFFFFFF00000000ull), r[0], r[1]);         // the upper half reads r[0]
    }
}
__host__ __device__ constexpr unsigned long long lane_bit_mask(uint32_t bit)      // lanes whose index has `bit` set (bit < 64)
{
    unsigned long long m = 0;
    for (uint32_t l = 0; l < 64; l++) if (l & bit) m |= 1ull << l;
    return m;
}

template <uint32_t N, uint32_t R, uint32_t K, uint32_t J, typename KEYS>
__device__ __forceinline__ void wave_bitonic_stage(Key64 (&v)[R], const uint32_t ebase /* 64 R wave */, KEYS s_keys, const bool active)
{
    const uint32_t lane = threadIdx.x & 63u;
    if constexpr (J < 64) {
#pragma unroll
        for (uint32_t r = 0; r < R; r++) {
            Key64 p;
            p.lo = lane_xor_value<J>(v[r].lo); p.hi = lane_xor_value<J>(v[r].hi);
            // keep the larger key iff (e & J != 0) != (e & K != 0): K < 64 -> a literal lane mask, else uniform for this (wave, r)
            unsigned long long keep_max = lane_bit_mask(J);
            if constexpr (K < 64) keep_max ^= lane_bit_mask(K);
            else if (K < N && ((ebase + 64u * r) & K)) keep_max = ~keep_max;
            const unsigned long long less = __builtin_amdgcn_ballot_w64(key_less(p, v[r]));
            const unsigned long long take = less ^ keep_max;             // keep-min lanes take a smaller partner, keep-max lanes a not-smaller one
            // (the lane mask goes straight into v_cndmask: written as `(take >> lane) & 1 ? .. : ..` hipcc rebuilds a per-lane bool with five VALU ops)
            v[r].lo = mask_select(take, p.lo, v[r].lo); v[r].hi = mask_select(take, p.hi, v[r].hi);
        }
    } else if constexpr (J < 64 * R) {
        constexpr uint32_t m = J / 64;
#pragma unroll
        for (uint32_t r = 0; r < R; r++) {
            if (!(r & m)) {
                const bool asc = K >= N || ((ebase + 64u * r) & K) == 0;
                const bool swap = key_less(v[r | m], v[r]) == asc;
                const Key64 a = v[r], b = v[r | m];
                v[r] = swap ? b : a; v[r | m] = swap ? a : b;
            }
        }
    } else {
        // another wave holds the partner: through LDS (every thread writes its keys, reads the partner's, keeps one of the two)
        if (active) {
#pragma unroll
            for (uint32_t r = 0; r < R; r++) s_keys[ebase + 64u * r + lane] = ((uint64_t)v[r].hi << 32) | v[r].lo;
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (uint32_t r = 0; r < R; r++) {
                const uint32_t e = ebase + 64u * r + lane;
                const uint64_t pk = s_keys[e ^ J];
                const Key64 p{(uint32_t)pk, (uint32_t)(pk >> 32)};
                const bool keep_max = ((e & J) != 0) != (K < N && (e & K) != 0);
                const bool t = key_less(p, v[r]) != keep_max;
                v[r].lo = t ? p.lo : v[r].lo; v[r].hi = t ? p.hi : v[r].hi;
            }
        }
        __syncthreads();
    }
}
template <uint32_t N, uint32_t R, uint32_t K, uint32_t J, typename KEYS>
__device__ __forceinline__ void wave_bitonic_level(Key64 (&v)[R], const uint32_t ebase, KEYS s_keys, const bool active)
{
    if constexpr (J >= 1) {
        wave_bitonic_stage<N, R, K, J>(v, ebase, s_keys, active);
        wave_bitonic_level<N, R, K, J / 2>(v, ebase, s_keys, active);
    }
}
template <uint32_t N, uint32_t R, uint32_t K, typename KEYS>
__device__ __forceinline__ void wave_bitonic_levels(Key64 (&v)[R], const uint32_t ebase, KEYS s_keys, const bool active)
{
    if constexpr (K <= N) {
        wave_bitonic_level<N, R, K, K / 2>(v, ebase, s_keys, active);
        wave_bitonic_levels<N, R, K * 2>(v, ebase, s_keys, active);
    }
}
// sorts s_keys[0 .. N) (N = 64 .. 1024, a power of two; filled and barriered by the caller) ascending, in place; NT threads, all must call
template <uint32_t N, typename KEYS, uint32_t NT = 256>
__device__ __forceinline__ void wave_bitonic_sort(KEYS s_keys)
{
    constexpr uint32_t R = N >= NT ? N / NT : 1;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;   // (uniform: the stage masks live in SGPRs)
    const uint32_t ebase = 64u * R * wave;
    const bool active = ebase < N;                       // N < 256: only the first N / 64 waves hold keys (the others still take part in barriers)
    Key64 v[R];
#pragma unroll
    for (uint32_t r = 0; r < R; r++) {
        const uint64_t k = active ? (uint64_t)s_keys[ebase + 64u * r + lane] : ~0ull;
        v[r] = Key64{(uint32_t)k, (uint32_t)(k >> 32)};
    }
    if constexpr (N > 64 * R) __syncthreads();           // (the cross-wave stages overwrite s_keys)
    wave_bitonic_levels<N, R, 2>(v, ebase, s_keys, active);
    if (active) {
#pragma unroll
        for (uint32_t r = 0; r < R; r++) s_keys[ebase + 64u * r + lane] = ((uint64_t)v[r].hi << 32) | v[r].lo;
    }
    __syncthreads();
}

// the fused sort of render_fwd_kernel (lists of up to 1024 keys, one block per tile, every tile of the frame sorting at the same time):
// the phase is bound by its chain of dependent LDS round trips, not by bandwidth or issue, so the register-blocked networks (two
// stages per round trip) are taken from 256 keys on
#ifndef GSR_WAVE_SORT
#define GSR_WAVE_SORT 1     // 0: round 3's LDS networks in the fused sort (A/B runs)
#endif
__device__ __forceinline__ void bitonic_sort_lds_pow2_fused(PaddedKeys keys, uint32_t npad)
{
    switch (npad) {
#if GSR_WAVE_SORT
        case 1024: bitonic_sort_lds_reg<1024, 2>(keys); break;
        case 512: wave_bitonic_sort<512>(keys); break;
        case 256: wave_bitonic_sort<256>(keys); break;
        case 128: wave_bitonic_sort<128>(keys); break;
        case 64: wave_bitonic_sort<64>(keys); break;
#else
        case 1024: bitonic_sort_lds_reg<1024, 2>(keys); break;
        case 512: bitonic_sort_lds_reg<512, GSR_FUSED_SORT_M>(keys); break;
        case 256: bitonic_sort_lds_reg<256, GSR_FUSED_SORT_M>(keys); break;
        case 128: bitonic_sort_lds<128>(keys); break;
        case 64: bitonic_sort_lds<64>(keys); break;
#endif
        default: bitonic_sort_block<true>(keys, npad); break;   // < 64
    }
}
// Two instantiations: CAP = SORT_SMALL_CAP (8 KiB of LDS: every tile of a 1200-tile frame sorts concurrently) handles the
// lists of up to 1024 keys; CAP = SORT_LDS_CAP (32 KiB, only 4 blocks per CU) is launched only when some list is longer and
// handles those (in LDS up to 4096 keys, in global memory beyond). With the 32 KiB variant alone the 1200 blocks of a
// 640x480 frame did not fit in one residency round and the kernel took two (40 us -> 20 us).
constexpr int SORT_SMALL_CAP = 1024;
constexpr int SORT_MID_CAP = 2048;     // lists of 1025 .. 2048 keys: a block of their own with half the LDS of the SORT_LDS_CAP one
// Sort of one tile list of n <= CAP keys in LDS (256 threads, s_keys holds CAP keys). A bitonic network wants a power of two;
// padding 520 keys to 1024 would more than double the work. Instead the list is split into A = the largest power of two <= n
// and the rest (padded to its own power of two), both halves are sorted, and every key finds its final rank with one binary
// search in the other half (keys are unique). Ends with the sorted (gaussian, instance) pairs in global memory.
template <uint32_t MAXN = 4096, typename KEYS = uint64_t*>
__device__ __forceinline__ void sort_tile_in_lds(const uint2 r, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ inst_gauss,
                                                 uint2* __restrict__ sorted, KEYS s_keys, uint32_t* dbg_ticks = nullptr)
{
    const uint32_t n = r.y - r.x;
    const uint64_t* seg = keys + r.x;
    // lists of up to 64 keys are padded to one power of two (a padded key array wants the second half to start at a multiple of 8)
    const uint32_t A = n <= 64 ? next_pow2(n) : (1u << (31 - __clz((int)n)));
    const uint32_t B = n > A ? n - A : 0u, Bpad = B ? max(8u, next_pow2(B)) : 0u;
    for (uint32_t i = threadIdx.x; i < A + Bpad; i += 256) s_keys[i] = i < n ? seg[i] : ~0ull;
    __syncthreads();
    if (dbg_ticks) dbg_ticks[0] = (uint32_t)__builtin_amdgcn_s_memtime();
    if constexpr (MAXN == 0) {      // render_fwd_kernel's fused sort
        bitonic_sort_lds_pow2_fused(s_keys, A);
        if (B) bitonic_sort_lds_pow2_fused(s_keys + A, Bpad);
    } else {
        bitonic_sort_lds_pow2<MAXN>(s_keys, A);
        if (B) bitonic_sort_lds_pow2<MAXN>(s_keys + A, Bpad);
    }
    if (dbg_ticks) dbg_ticks[1] = (uint32_t)__builtin_amdgcn_s_memtime();
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint64_t key = s_keys[i];
        const bool in_a = i < A;
        const uint32_t other = in_a ? A : 0u;
        uint32_t lo = 0, hi = in_a ? B : A;              // number of keys of the other half that are smaller
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_keys[other + mid] < key) lo = mid + 1; else hi = mid;
        }
        const uint32_t rank = (in_a ? i : i - A) + lo;
        const uint32_t u = (uint32_t)key;
        sorted[r.x + rank] = make_uint2(inst_gauss[u], u);
    }
}

template <int CAP, int LOWER>
__device__ __forceinline__ void sort_tiles_body(int ntiles, const uint2* ranges, uint64_t* keys, const uint32_t* inst_gauss,
                                                         uint2* sorted, const uint32_t* spec_header)
{
    constexpr bool PAD = CAP > SORT_SMALL_CAP;              // the variant for lists beyond 1024 keys: padded layout, register-blocked networks
    __shared__ uint64_t s_keys[PAD ? padded_keys_size(CAP) : CAP];
    if (spec_header && (spec_header[HDR_FLAGS] & FLAG_OVERFLOW)) return;   // speculative launch on a buffer that turned out too small
    const int tile = xcd_tile_of_block(blockIdx.x, ntiles);
    const uint2 r = ranges[tile];
    const uint32_t n = r.y - r.x;
    if (n <= (uint32_t)LOWER) return;                       // empty, or the other instantiation's tile
    if (CAP < SORT_LDS_CAP && n > (uint32_t)CAP) return;
    if (n <= (uint32_t)CAP) {
        if constexpr (PAD) sort_tile_in_lds<(uint32_t)CAP>(r, keys, inst_gauss, sorted, PaddedKeys{s_keys});
        else sort_tile_in_lds<(uint32_t)CAP>(r, keys, inst_gauss, sorted, s_keys);
    }
    // n > SORT_LDS_CAP: sort_long_chunks_kernel + rank_long_chunks_kernel
}

// F4b: tile lists longer than SORT_LDS_CAP keys (SLAM-shaped maps: large Gaussians, thousands of entries per tile). The list is cut into
// SORT_LDS_CAP-key chunks; (1) every chunk is sorted in LDS, in place in the key segment; (2) every key's final position is its index
// in its own chunk plus, for every OTHER chunk, the number of keys there that are smaller (binary search; keys are unique because the
// instance id is part of the key). Work grows with (number of chunks)^2 per tile, which stays small next to what the compositing
// kernels do with such a list; the global-memory bitonic network this replaces took 2.0 ms for 6.5 M instances (now 0.15 ms).
// grid = (tiles, chunks of the longest list the launch is sized for).
template <int CHUNKK>
__device__ __forceinline__ void sort_long_chunks_body(const uint2* __restrict__ ranges, uint64_t* __restrict__ keys, const uint32_t* spec_header,
                                                               uint32_t lower)
{
    __shared__ uint64_t s_raw[padded_keys_size(CHUNKK)];
    const PaddedKeys s_keys{s_raw};
    if (spec_header && (spec_header[HDR_FLAGS] & FLAG_OVERFLOW)) return;
    const uint2 r = ranges[blockIdx.x];
    const uint32_t n = r.y - r.x, c0 = blockIdx.y * (uint32_t)CHUNKK;
    if (n <= lower || c0 >= n) return;
    const uint32_t m = min((uint32_t)CHUNKK, n - c0), mpad = next_pow2(m);
    uint64_t* seg = keys + r.x + c0;
    for (uint32_t i = threadIdx.x; i < mpad; i += 256) s_keys[i] = i < m ? seg[i] : ~0ull;
    __syncthreads();
    bitonic_sort_lds_pow2(s_keys, mpad);
    for (uint32_t i = threadIdx.x; i < m; i += 256) seg[i] = s_keys[i];
}

template <int CHUNKK>
__device__ __forceinline__ void rank_long_chunks_body(const uint2* __restrict__ ranges, const uint64_t* __restrict__ keys,
                                                               const uint32_t* __restrict__ inst_gauss, uint2* __restrict__ sorted,
                                                               const uint32_t* spec_header, uint32_t lower)
{
    if (spec_header && (spec_header[HDR_FLAGS] & FLAG_OVERFLOW)) return;
    const uint2 r = ranges[blockIdx.x];
    const uint32_t n = r.y - r.x, c0 = blockIdx.y * (uint32_t)CHUNKK;
    if (n <= lower || c0 >= n) return;
    const uint32_t m = min((uint32_t)CHUNKK, n - c0);
    const uint64_t* seg = keys + r.x;
    const uint32_t nchunks = (n + (uint32_t)CHUNKK - 1) / (uint32_t)CHUNKK;
    for (uint32_t i = threadIdx.x; i < m; i += 256) {
        const uint64_t key = seg[c0 + i];
        uint32_t rank = i;
        for (uint32_t c = 0; c < nchunks; c++) {
            if (c == blockIdx.y) continue;
            const uint64_t* other = seg + c * (uint32_t)CHUNKK;
            uint32_t lo = 0, hi = min((uint32_t)CHUNKK, n - c * (uint32_t)CHUNKK);
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (other[mid] < key) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        const uint32_t u = (uint32_t)key;
        sorted[r.x + rank] = make_uint2(inst_gauss[u], u);
    }
}

// rasterizer_impl.cu:54-66
__global__ void mark_visible_kernel(int P, const float* means3D, const float* viewmatrix, uint8_t* present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const f3 pv = xform_point_4x3(mk3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]), viewmatrix);
    present[idx] = pv.z > 0.2f ? 1 : 0;
}

// ---- the single-view kernels: the bodies above with their arguments passed by value (gs_views.h launches the same bodies once for
// several views) ----------------------------------------------------------------------------------------------------------------------
template <bool RAW, bool PRE = false>
__global__ void __launch_bounds__(GB) preprocess_fwd_kernel(PreprocessArgs a)
{
    preprocess_fwd_body<RAW, PRE>(a);
}

template <int SEGS, int RMAX>
__global__ void __launch_bounds__(TO_COLS * SEGS) tile_offsets_kernel(int nblocks, int T, uint32_t* __restrict__ hist,
                                                                      uint32_t* __restrict__ tile_count, uint32_t* __restrict__ dense_total)
{
    tile_offsets_body<SEGS, RMAX>(nblocks, T, hist, tile_count, dense_total);
}

__global__ void __launch_bounds__(1024) scan_kernel(int nblocks, const uint32_t* block_sums, uint32_t* block_base,
                                                    int ntiles, const uint32_t* tile_count, uint2* ranges, uint32_t* tile_cursor,
                                                    const uint32_t* flags, uint32_t cap_R, uint32_t cap_tile_list,
                                                    uint32_t* chunk_base, uint32_t* header, uint32_t* host_mailbox, uint32_t seq)
{
    scan_body(nblocks, block_sums, block_base, ntiles, tile_count, ranges, tile_cursor, flags, cap_R, cap_tile_list, chunk_base, header, host_mailbox, seq);
}

__global__ void __launch_bounds__(GB) scatter_instances_kernel(int P, int gx, int gy, const int* radii, const TileRec* rec,
                                                               const uint32_t* tiles_touched,
                                                               const uint32_t* block_base, uint32_t* point_offsets,
                                                               uint32_t* tile_cursor, const uint2* ranges, const uint32_t* block_tile_base,
                                                               uint64_t* keys, uint32_t* inst_gauss, uint32_t* header, int speculative,
                                                               uint32_t carve_R, uint32_t cap_sorted, int eager, const int* clip, uint32_t* tile_pos, int order_fwd,
                                                               ScanInScatter scan)
{
    const bool merged = scan.dense_total != nullptr;
    if ((tile_pos != nullptr || merged) && blockIdx.x == gridDim.x - 1) {        // the launch's extra block (F2b, F3b, F3c)
        extern __shared__ uint32_t s_dyn[];
        if (merged) {
            uint32_t longest = 0;
            const bool fits = scan_in_scatter_bookkeeping(gx * gy, scan, s_dyn, header, longest);
            if (fits && tile_pos != nullptr) { __syncthreads(); order_tiles_body(gx * gy, nullptr, s_dyn, tile_pos, order_fwd != 0, longest, (uint32_t)max(0, order_fwd - 1)); }
        } else if (!(speculative && (header[HDR_FLAGS] & FLAG_OVERFLOW))) {
            order_tiles_body(gx * gy, ranges, s_dyn, tile_pos, order_fwd != 0, header[HDR_MAX_TILE], (uint32_t)max(0, order_fwd - 1));
        }
        return;
    }
    scatter_instances_body(P, gx, gy, radii, rec, tiles_touched, block_base, point_offsets, tile_cursor, ranges, block_tile_base, keys, inst_gauss, header, speculative, carve_R, cap_sorted, eager, clip, &scan);
}

template <int CAP, int LOWER>
__global__ void __launch_bounds__(256) sort_tiles_kernel(int ntiles, const uint2* ranges, uint64_t* keys, const uint32_t* inst_gauss,
                                                         uint2* sorted, const uint32_t* spec_header)
{
    sort_tiles_body<CAP, LOWER>(ntiles, ranges, keys, inst_gauss, sorted, spec_header);
}

template <int CHUNKK>
__global__ void __launch_bounds__(256) sort_long_chunks_kernel(const uint2* __restrict__ ranges, uint64_t* __restrict__ keys, const uint32_t* spec_header,
                                                               uint32_t lower)
{
    sort_long_chunks_body<CHUNKK>(ranges, keys, spec_header, lower);
}

template <int CHUNKK>
__global__ void __launch_bounds__(256) rank_long_chunks_kernel(const uint2* __restrict__ ranges, const uint64_t* __restrict__ keys,
                                                               const uint32_t* __restrict__ inst_gauss, uint2* __restrict__ sorted,
                                                               const uint32_t* spec_header, uint32_t lower)
{
    rank_long_chunks_body<CHUNKK>(ranges, keys, inst_gauss, sorted, spec_header, lower);
}

}  // namespace gsr
