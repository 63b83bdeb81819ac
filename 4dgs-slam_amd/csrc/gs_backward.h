// gs_backward.h -- backward kernels: per-tile back-to-front gradient pass and the per-Gaussian geometry backward.
// gfx950 / wave64. DGR = submodules/diff-gaussian-rasterization.
#pragma once
#include "gs_forward.h"
#include "gs_render.h"

namespace gsr {

// ------------------------------------------------------------------------------------------------------------------
// B2: per-Gaussian backward. Sums the Gaussian's instance slots (fixed order => deterministic), then runs the
// reference's computeCov2DCUDA (backward.cu:150-346) and preprocessCUDA backward (:418-539) back to back in
// registers, so dL_dconic / dL_dmean2D never round-trip through memory between two kernels.
// Writes EVERY output element (zeros for culled Gaussians), so outputs need no pre-zeroing.
// ------------------------------------------------------------------------------------------------------------------
struct GeomBwdArgs {
    int P, D, M, W, H;
    const float* means3D; const int* radii; const float* shs; const uint8_t* clamped; const float* scales; const float* rotations;
    float scale_modifier; const float* cov3Ds; const float* viewmatrix; const float* projmatrix; const float* projmatrix_raw;
    const float* campos; float focal_x, focal_y, tan_fovx, tan_fovy;
    const uint32_t* tiles_touched; const uint32_t* point_offsets;
    const char* bin_base; const uint32_t* header;   // binning buffer + geometry header: the per-instance slots are located on the device
    float* dL_dmean2D; float* dL_dconic; float* dL_dopacity; float* dL_dcolor; float* dL_ddepth;
    float* dL_dmean3D; float* dL_dcov3D; float* dL_dsh; float* dL_dscale; float* dL_drot; float* dL_dtau;
    float* tau_partials;   // optional [nblocks][6]: per-block sums of dL_dtau, so the caller does not have to reduce [P,6]
    // GSR_BACKWARD_ACCUMULATE: the PARAMETER gradients (dL_dmean3D, dL_dsh, dL_dopacity, dL_dscale, dL_drot; raw mode: their raw
    // counterparts) are added to what the buffers hold, for visible Gaussians only -- rows of invisible Gaussians are not touched.
    // A caller that sums several views into one .grad buffer (mapping: 8-64 keyframes per optimizer step) thereby skips both the
    // zero rows this kernel would write (71 % of 2 M Gaussians at BASELINE config #5) and autograd's read-modify-write of five tensors.
    int accumulate;
    // GSR_BACKWARD_POSE_ONLY: nobody wants the Gaussians' parameter gradients (camera tracking: only the pose and the screen-space
    // gradients are read). Their stores, the covariance -> scale / rotation chain and the SH coefficient gradients are skipped; the
    // output pointers may be NULL.
    int pose_only;
    int sh_rows;   // SH coefficient rows through LDS (below); 0: per lane, as rounds 1-5
    // Raw mode (raw.xyz != nullptr, see gs_device.h): the inputs are the model's raw parameters and the outputs their gradients:
    // dL_dmean3D -> d/d_xyz, dL_dscale -> d/d_scaling [P,scale_dim], dL_drot -> d/d_rotation, dL_dopacity -> d/d_opacity (logit),
    // rawg.f_dc / f_rest -> d/d_features_*, rawg.ddx / dds / ddr [K,*] -> gradients of the control-node deltas.
    RawInputs raw; RawGrads rawg;
};

#if GSR_FWD_TIMING
__device__ uint32_t g_geo_timing[8 * 4 * 8192];
#define GEO_TICK(k) do { if (blockIdx.x < 8192 && (threadIdx.x & 63) == 0) g_geo_timing[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define GEO_TICK(k)
#endif
template <bool RAW, bool PRE = false>     // PRE: deformation-network deltas in front of the activations (RawInputs::delta_mode = 1)
__device__ __forceinline__ void geometry_bwd_body(GeomBwdArgs a)
{
    // RAW: the fused-prologue mode (raw.xyz != nullptr). As a template parameter the plain instantiation is straight-line code: with the
    // runtime test every parameter load sat in its own uniform branch with its own s_waitcnt behind it.
    RawInputs R = a.raw; RawGrads RG = a.rawg;
    if constexpr (!RAW) { R = RawInputs{}; RG = RawGrads{}; }
    GEO_TICK(0);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = lane_id();
    const bool in_range = idx < a.P;
    const size_t i = (size_t)(in_range ? idx : 0);
    // ---- level 1 of the loads: what decides everything else ------------------------------------------------------------------
    const int radius = in_range ? a.radii[idx] : 0;
    const uint32_t touched = in_range ? a.tiles_touched[idx] : 0u;
    const uint32_t incl = in_range ? a.point_offsets[idx] : 0u;
    const float* __restrict__ partials = reinterpret_cast<const float*>(carve_binning(const_cast<char*>(a.bin_base), a.header[HDR_CARVE_R], a.header[HDR_CAP_SORTED],
                                                        (size_t)(((a.W + TILE_X - 1) / TILE_X) * ((a.H + TILE_Y - 1) / TILE_Y))).partials);
    if (a.header[HDR_FLAGS] & FLAG_OVERFLOW) return;     // lazy forward pass that outgrew its buffer: there are no instance slots to sum (tested
                                                         // HERE, behind the level-1 loads: in front of them it was a global latency of its own)
    const bool visible = radius > 0;                     // backward.cu:163,443
    const size_t o = in_range && RAW ? raw_row(R, i) : i;   // row of the parameter-gradient outputs (raw mode with a mask: the selected row)
    const uint32_t cnt = visible ? touched : 0u;
    const uint32_t u0 = incl - touched;                  // instance ids are a global running count over Gaussians: this one owns [u0, u0 + cnt)
    // ---- level 2: EVERYTHING else this thread will read is requested here, in one go: the first chunk of the wave's instance slots,
    // the Gaussian's parameters and (accumulate mode) what the caller's gradient buffers hold. Round 2 fetched the parameters where the
    // chain rules first touched them and staged the slots per BLOCK (global -> registers -> LDS -> block barrier -> per-thread ds_read
    // loops, 3.3 chunks per block at config #2): eight exposed global latencies and four block barriers in a kernel whose 782 blocks
    // (3 waves per SIMD at 200k Gaussians) have nothing else to cover them with -- 38 000 cycles per wave, 14 500 of them in the chunk loop.
    // Now the staging is per WAVE: the 64 Gaussians of a wave own one contiguous slot range [W0, W1) (instance ids are a running count),
    // which the wave copies through a private 6 KiB LDS window in fully coalesced 16-byte loads (128 slots per step, the next step's
    // loads in flight while the lanes sum the current one) -- no block barrier anywhere in the kernel.
    constexpr uint32_t COOP = 16;                        // Gaussians with more instances are summed by the whole wave (below)
    constexpr uint32_t WCH = 128;                        // slots per step of a wave's window
    static_assert(WCH * SLOT_FLOATS <= SH_WIN_FLOATS && (WCH * SLOT_FLOATS) % 4 == 0, "the SH row window reuses the slot window");
    constexpr int WIN4 = WCH * SLOT_FLOATS / 4;       // float4 per full window (320: five per lane)
    __shared__ float4 s_slot[4][SH_WIN_FLOATS / 4];   // per wave: 128 instance slots (1280 floats), later the SH row window (64 x 25 floats)
    const bool coop = cnt > COOP;
    const uint32_t ser = coop ? 0u : cnt;
    const int wv = threadIdx.x >> 6;
    const uint32_t W0 = __builtin_amdgcn_readfirstlane(u0);                                   // lane 0 of a wave that has any lane in range is in range
    const uint32_t W1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, min(63, a.P - 1 - (blockIdx.x * 256 + wv * 64)));
    const bool any_ser = __any(ser > 0);
    float g_m2x = 0.f, g_m2y = 0.f, g_cx = 0.f, g_cy = 0.f, g_cw = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f, g_d = 0.f;
    float4 nx[WIN4 / 64];
    auto fetch_window = [&](uint32_t c0) {      // this lane's five float4 of the window that starts at slot c0 (8-byte aligned: 40-byte slots)
        const uint32_t n4 = (min(WCH, W1 - c0) * (uint32_t)SLOT_FLOATS + 3u) / 4u;      // (an odd slot count ends in half a float4: its other half is the next slot's, or padding)
        const float4_a8* src = reinterpret_cast<const float4_a8*>(partials + (size_t)c0 * SLOT_FLOATS);
#pragma unroll
        for (int j = 0; j < WIN4 / 64; j++) nx[j] = (uint32_t)lane + 64u * j < n4 ? (float4)src[lane + 64 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const bool wave_has_slots = any_ser && W0 < W1;                 // uniform per wave (lanes beyond P take part in the window copy)
    // Only windows that hold a slot of a serially summed Gaussian are copied: on a SLAM-sized map most of a wave's range belongs to
    // Gaussians of > COOP instances, which the cooperative path below reads straight from global memory (85 % of 3 000 slots per wave
    // at 30 k Gaussians x 48 tiles went through LDS for nothing: 36 k cycles per wave).
    auto next_needed_window = [&](uint32_t c0) {                    // first window start >= c0 some lane sums from, or >= W1 (wave-uniform)
        while (c0 < W1 && !__any(max(u0, c0) < min(u0 + ser, min(c0 + WCH, W1)))) c0 += WCH;
        return c0;
    };
    const uint32_t first_window = wave_has_slots ? next_needed_window(W0) : W1;
    if (first_window < W1) fetch_window(first_window);
    const bool scale1 = RAW && R.scale_dim == 1;
    const bool want_cov_chain = !a.pose_only && (a.scales != nullptr || RAW);
    f3 mean = mk3(0.f, 0.f, 0.f);
    float cov6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f}, s3[3] = {0.f, 0.f, 0.f};
    uint32_t clamp_bits = 0;
    // Accumulate mode (GeomBwdArgs::accumulate): the old values ride with the other loads; add_separately: the addition must not be
    // contracted into the expression that produced the gradient (autograd's accumulation rounds it first). old_*: zero when the mode is off.
    float old_m[3] = {0.f, 0.f, 0.f}, old_s[3] = {0.f, 0.f, 0.f}, old_r[4] = {0.f, 0.f, 0.f, 0.f}, old_o = 0.f, old_c[3] = {0.f, 0.f, 0.f};
    float* const dc_out = RAW ? (RG.f_dc ? RG.f_dc + 3 * o : nullptr) : (a.dL_dsh ? a.dL_dsh + i * a.M * 3 : nullptr);
    if (visible) {
        mean = load_mean<PRE>(a.means3D, R, i);
#pragma unroll
        for (int k = 0; k < 6; k++) cov6[k] = a.cov3Ds[6 * i + k];
        if (want_cov_chain) { load_rot<PRE>(a.rotations, R, i, q4); load_scale<PRE>(a.scales, R, i, s3); }
        clamp_bits = a.clamped[idx];
        if (a.accumulate && !a.pose_only) {
#pragma unroll
            for (int k = 0; k < 3; k++) old_m[k] = a.dL_dmean3D[3 * o + k];
            old_o = a.dL_dopacity[o];
            if (a.dL_dscale) {
                if (scale1) old_s[0] = a.dL_dscale[o];
                else {
#pragma unroll
                    for (int k = 0; k < 3; k++) old_s[k] = a.dL_dscale[3 * o + k];
                }
            }
            if (a.dL_drot) {
#pragma unroll
                for (int k = 0; k < 4; k++) old_r[k] = a.dL_drot[4 * o + k];
            }
            if (dc_out) {
#pragma unroll
                for (int k = 0; k < 3; k++) old_c[k] = dc_out[k];
            }
        }
    }
    GEO_TICK(1);
    if (first_window < W1) {
        float4* const win = s_slot[wv];
        for (uint32_t c0 = first_window, cn; c0 < W1; c0 = cn) {
#pragma unroll
            for (int j = 0; j < WIN4 / 64; j++) win[lane + 64 * j] = nx[j];
            cn = next_needed_window(c0 + WCH);
            if (cn < W1) fetch_window(cn);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the window is wave-private: DS operations of one wave execute in order
            __builtin_amdgcn_wave_barrier();
            const uint32_t lo = max(u0, c0), hi = min(u0 + ser, min(c0 + WCH, W1));
            for (uint32_t u = lo; u < hi; u++) {                       // ascending instance order: the order every earlier revision summed in
                const float2* sl = reinterpret_cast<const float2*>(reinterpret_cast<const float*>(win) + (u - c0) * SLOT_FLOATS);   // ten floats, 8-byte aligned
                const float2 v0 = sl[0], v1 = sl[1], v2 = sl[2], v3 = sl[3], v4 = sl[4];
                g_m2x += v0.x; g_m2y += v0.y; g_cx += v1.x; g_cy += v1.y;
                g_cw += v2.x; g_op += v2.y; g_r += v3.x; g_g += v3.y;
                g_b += v4.x; g_d += v4.y;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    GEO_TICK(2);
    // Gaussians that own many instances (SLAM-shaped maps: 20-70 tiles per Gaussian) are summed by the whole wave, one after the other:
    // lane l takes slots l, l + 64, ... (coalesced 48-byte rows), the ten values are reduced with the transposed butterfly of
    // gs_device.h and the totals handed to the owning lane. The rule depends on the Gaussian alone => still bit-reproducible.
    {
        unsigned long long cm = __ballot(coop);
        if (cm) {
            const WaveSelectMasks wsm = wave_select_masks();
            // the first 64 slots of the NEXT Gaussian are requested before the current one is reduced: one at a time, every Gaussian
            // paid a full global latency in front of its butterfly (~1 100 cycles each, 51 k cycles per wave on a SLAM-sized map)
            int h = pop_lowest_bit(cm);
            uint32_t hc = (uint32_t)__builtin_amdgcn_readlane((int)cnt, h), hu = (uint32_t)__builtin_amdgcn_readlane((int)u0, h);
            const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float2 zero2 = make_float2(0.f, 0.f);
            float4 p0 = zero4, p1 = zero4; float2 p2 = zero2;
            bool pv = (uint32_t)lane < hc;
            if (pv) slot_load(partials, (size_t)(hu + (uint32_t)lane), p0, p1, p2);
            for (;;) {
                const bool more = cm != 0ull;
                int h2 = 0;
                uint32_t hc2 = 0, hu2 = 0;
                float4 q0 = zero4, q1 = zero4; float2 q2 = zero2;
                bool qv = false;
                if (more) {
                    h2 = pop_lowest_bit(cm);
                    hc2 = (uint32_t)__builtin_amdgcn_readlane((int)cnt, h2); hu2 = (uint32_t)__builtin_amdgcn_readlane((int)u0, h2);
                    qv = (uint32_t)lane < hc2;
                    if (qv) slot_load(partials, (size_t)(hu2 + (uint32_t)lane), q0, q1, q2);
                }
                float s0 = 0.f, s5 = 0.f;
                f2v a12 = {0.f, 0.f}, a34 = {0.f, 0.f}, a67 = {0.f, 0.f}, a89 = {0.f, 0.f};
                if (pv) { s0 += p0.x; a12 += f2v{p0.y, p0.z}; a34 += f2v{p0.w, p1.x}; s5 += p1.y; a67 += f2v{p1.z, p1.w}; a89 += f2v{p2.x, p2.y}; }
                for (uint32_t k = (uint32_t)lane + 64u; k < hc; k += 128) {       // two trips' loads in flight (same summation order)
                    const bool two = k + 64u < hc;
                    float4 v0, v1, w0 = zero4, w1 = zero4; float2 v2, w2 = zero2;
                    slot_load(partials, (size_t)(hu + k), v0, v1, v2);
                    if (two) slot_load(partials, (size_t)(hu + k + 64u), w0, w1, w2);
                    s0 += v0.x; a12 += f2v{v0.y, v0.z}; a34 += f2v{v0.w, v1.x}; s5 += v1.y; a67 += f2v{v1.z, v1.w}; a89 += f2v{v2.x, v2.y};
                    if (two) { s0 += w0.x; a12 += f2v{w0.y, w0.z}; a34 += f2v{w0.w, w1.x}; s5 += w1.y; a67 += f2v{w1.z, w1.w}; a89 += f2v{w2.x, w2.y}; }
                }
                unsigned long long dummy_proc = 0; uint32_t dummy_addr;
                const float tot = wave_sum10_transposed(wsm, s0, a12, a34, s5, a67, a89, dummy_proc, 0, 0u, 0, dummy_addr);
                // lanes holding total k (wave_sum10_slot_of_lane): 0 -> 2, 1 -> 0, 2 -> 1, 3 -> 32, 4 -> 33, 5 -> 34, 6 -> 16, 7 -> 17, 8 -> 48, 9 -> 49
                const int ti = __float_as_int(tot);
                const float t0 = __int_as_float(__builtin_amdgcn_readlane(ti, 2)), t1 = __int_as_float(__builtin_amdgcn_readlane(ti, 0));
                const float t2 = __int_as_float(__builtin_amdgcn_readlane(ti, 1)), t3 = __int_as_float(__builtin_amdgcn_readlane(ti, 32));
                const float t4 = __int_as_float(__builtin_amdgcn_readlane(ti, 33)), t5 = __int_as_float(__builtin_amdgcn_readlane(ti, 34));
                const float t6 = __int_as_float(__builtin_amdgcn_readlane(ti, 16)), t7 = __int_as_float(__builtin_amdgcn_readlane(ti, 17));
                const float t8 = __int_as_float(__builtin_amdgcn_readlane(ti, 48)), t9 = __int_as_float(__builtin_amdgcn_readlane(ti, 49));
                if (lane == h) { g_m2x = t0; g_m2y = t1; g_cx = t2; g_cy = t3; g_cw = t4; g_op = t5; g_r = t6; g_g = t7; g_b = t8; g_d = t9; }
                if (!more) break;
                h = h2; hc = hc2; hu = hu2; p0 = q0; p1 = q1; p2 = q2; pv = qv;
            }
        }
    }
    GEO_TICK(3);

    if (in_range) {
        a.dL_dmean2D[3 * o] = g_m2x; a.dL_dmean2D[3 * o + 1] = g_m2y; a.dL_dmean2D[3 * o + 2] = 0.f;   // z never written, Q14
        if (a.dL_dconic) { a.dL_dconic[4 * i] = g_cx; a.dL_dconic[4 * i + 1] = g_cy; a.dL_dconic[4 * i + 2] = 0.f; a.dL_dconic[4 * i + 3] = g_cw; }
        // dL_dopacity is stored with the other parameter gradients at the end (its old value is still in flight in accumulate mode)
        if (a.dL_dcolor) { a.dL_dcolor[3 * i] = g_r; a.dL_dcolor[3 * i + 1] = g_g; a.dL_dcolor[3 * i + 2] = g_b; }
        if (a.dL_ddepth) a.dL_ddepth[i] = g_d;
    }

    float dmean[3] = {0.f, 0.f, 0.f}, dtau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
    const bool flow = RAW && R.flow_proj1;
    const bool has_sh = flow ? false : (RAW ? (RG.f_dc != nullptr || a.pose_only != 0) : (a.shs != nullptr && (a.dL_dsh != nullptr || a.pose_only != 0)));   // pose-only: the view-direction term of dL_dtau still needs the SH pass
    const ShOut dsh = RAW ? ShOut{RG.f_dc + 3 * o, RG.f_rest ? RG.f_rest + o * (size_t)(a.M - 1) * 3 : nullptr, a.accumulate != 0, {old_c[0], old_c[1], old_c[2]}, a.pose_only != 0}
                                : ShOut{a.dL_dsh ? a.dL_dsh + i * a.M * 3 : nullptr, a.dL_dsh ? a.dL_dsh + i * a.M * 3 + 3 : nullptr, a.accumulate != 0, {old_c[0], old_c[1], old_c[2]}, a.pose_only != 0};
    // SH rows through LDS (top of the file) for the coefficient counts the models use; a.sh_rows = 0 (gsr_set_option "sh_rows") keeps the per-lane path
    const bool sh_staged = a.sh_rows && has_sh && (a.M == 9 || a.M == 16);
    if (!visible) {
        // (plain stores, not dsh[k] with a run-time k: that indexed dsh.old_dc dynamically and sent the whole struct to scratch memory -- the
        // 48 bytes per lane this kernel carried for three rounds)
        if (has_sh && !sh_staged && in_range && !a.accumulate && !a.pose_only) {
            dsh.dc[0] = 0.f; dsh.dc[1] = 0.f; dsh.dc[2] = 0.f;
            for (int k = 3; k < a.M * 3; k++) dsh.rest[k - 3] = 0.f;
        }
    } else {
        const float* vm = a.viewmatrix;

        // ---- backward.cu:171-346: conic -> cov2D -> (cov3D, t) ----
        const Cov2D cv = cov2d_eval(mean, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov6, vm);
        const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
        const float x_grad_mul = (cv.txtz < -limx || cv.txtz > limx) ? 0.f : 1.f;   // :182-183
        const float y_grad_mul = (cv.tytz < -limy || cv.tytz > limy) ? 0.f : 1.f;
        const float ca = cv.a, cb = cv.b, cc = cv.c;
        const float denom = ca * cc - cb * cb;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);               // :210
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float (&A)[2][3] = cv.A;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-cc * cc * g_cx + 2 * cb * cc * g_cy + (denom - ca * cc) * g_cw);     // :217-219
            dL_dc = denom2inv * (-ca * ca * g_cw + 2 * ca * cb * g_cy + (denom - ca * cc) * g_cx);
            dL_db = denom2inv * 2 * (cb * cc * g_cx - (denom + 2 * cb * cb) * g_cy + ca * cb * g_cw);
            dcov[0] = A[0][0] * A[0][0] * dL_da + A[0][0] * A[1][0] * dL_db + A[1][0] * A[1][0] * dL_dc;   // :224-234
            dcov[3] = A[0][1] * A[0][1] * dL_da + A[0][1] * A[1][1] * dL_db + A[1][1] * A[1][1] * dL_dc;
            dcov[5] = A[0][2] * A[0][2] * dL_da + A[0][2] * A[1][2] * dL_db + A[1][2] * A[1][2] * dL_dc;
            dcov[1] = 2 * A[0][0] * A[0][1] * dL_da + (A[0][0] * A[1][1] + A[0][1] * A[1][0]) * dL_db + 2 * A[1][0] * A[1][1] * dL_dc;
            dcov[2] = 2 * A[0][0] * A[0][2] * dL_da + (A[0][0] * A[1][2] + A[0][2] * A[1][0]) * dL_db + 2 * A[1][0] * A[1][2] * dL_dc;
            dcov[4] = 2 * A[0][2] * A[0][1] * dL_da + (A[0][1] * A[1][2] + A[0][2] * A[1][1]) * dL_db + 2 * A[1][1] * A[1][2] * dL_dc;
        }
        const float V[3][3] = {{cov6[0], cov6[1], cov6[2]}, {cov6[1], cov6[3], cov6[4]}, {cov6[2], cov6[4], cov6[5]}};
        float dT0[3], dT1[3];   // dL/dA rows (:244-255)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float a0v = A[0][0] * V[k][0] + A[0][1] * V[k][1] + A[0][2] * V[k][2];
            const float a1v = A[1][0] * V[k][0] + A[1][1] * V[k][1] + A[1][2] * V[k][2];
            dT0[k] = 2 * a0v * dL_da + a1v * dL_db;
            dT1[k] = 2 * a1v * dL_dc + a0v * dL_db;
        }
        const float dL_dJ00 = vm[0] * dT0[0] + vm[4] * dT0[1] + vm[8] * dT0[2];      // :259-262
        const float dL_dJ02 = vm[2] * dT0[0] + vm[6] * dT0[1] + vm[10] * dT0[2];
        const float dL_dJ11 = vm[1] * dT1[0] + vm[5] * dT1[1] + vm[9] * dT1[2];
        const float dL_dJ12 = vm[2] * dT1[0] + vm[6] * dT1[1] + vm[10] * dT1[2];
        const f3 t = cv.t;
        const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        const float fx = a.focal_x, fy = a.focal_y;
        const float dL_dtx = x_grad_mul * -fx * tz2 * dL_dJ02;                        // :269-271
        const float dL_dty = y_grad_mul * -fy * tz2 * dL_dJ12;
        const float dL_dtz = -fx * tz2 * dL_dJ00 - fy * tz2 * dL_dJ11 + (2 * fx * t.x) * tz3 * dL_dJ02 + (2 * fy * t.y) * tz3 * dL_dJ12;
        // pose, :273-288: d p_C / d rho = I, d p_C / d theta = -[t]x with the CLAMPED t (Q17 iii)
        dtau[0] += dL_dtx; dtau[1] += dL_dty; dtau[2] += dL_dtz;
        dtau[3] += -dL_dty * t.z + dL_dtz * t.y;      // column 0 of -[t]x = (0, -t.z, t.y)
        dtau[4] += dL_dtx * t.z - dL_dtz * t.x;       // column 1        = (t.z, 0, -t.x)
        dtau[5] += -dL_dtx * t.y + dL_dty * t.x;      // column 2        = (-t.y, t.x, 0)
        // :292-297 (assignment)
        dmean[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
        dmean[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
        dmean[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;
        // rotation part through W = R_cw, :299-343
        {
            const float dW[3][3] = {   // dW[col c of dL_dW][component]: cols[c] = (dW0c, dW1c, dW2c)
                {cv.J00 * dT0[0], cv.J11 * dT1[0], cv.J02 * dT0[0] + cv.J12 * dT1[0]},
                {cv.J00 * dT0[1], cv.J11 * dT1[1], cv.J02 * dT0[1] + cv.J12 * dT1[1]},
                {cv.J00 * dT0[2], cv.J11 * dT1[2], cv.J02 * dT0[2] + cv.J12 * dT1[2]}};
            float th0 = 0.f, th1 = 0.f, th2 = 0.f;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float cx_ = vm[4 * c], cy_ = vm[4 * c + 1], cz_ = vm[4 * c + 2];   // column c of R_cw
                // columns of -[c]x: (0,-cz,cy), (cz,0,-cx), (-cy,cx,0)
                th0 += -dW[c][1] * cz_ + dW[c][2] * cy_;
                th1 += dW[c][0] * cz_ - dW[c][2] * cx_;
                th2 += -dW[c][0] * cy_ + dW[c][1] * cx_;
            }
            dtau[3] += th0; dtau[4] += th1; dtau[5] += th2;
        }

        // ---- backward.cu:446-528: mean2D / depth -> mean3D and tau ----
        const float* pr = a.projmatrix;
        const float mhx = pr[0] * mean.x + pr[4] * mean.y + pr[8] * mean.z + pr[12];
        const float mhy = pr[1] * mean.x + pr[5] * mean.y + pr[9] * mean.z + pr[13];
        const float mhw = pr[3] * mean.x + pr[7] * mean.y + pr[11] * mean.z + pr[15];
        const float m_w = 1.0f / (mhw + 0.0000001f);
        const float mul1 = mhx * m_w * m_w, mul2 = mhy * m_w * m_w;
        dmean[0] += (pr[0] * m_w - pr[3] * mul1) * g_m2x + (pr[1] * m_w - pr[3] * mul2) * g_m2y;   // :457-463
        dmean[1] += (pr[4] * m_w - pr[7] * mul1) * g_m2x + (pr[5] * m_w - pr[7] * mul2) * g_m2y;
        dmean[2] += (pr[8] * m_w - pr[11] * mul1) * g_m2x + (pr[9] * m_w - pr[11] * mul2) * g_m2y;
        {   // approximate projection Jacobian w.r.t. the pose, :465-512 (only proj_raw[0], [5], [11]; Q17 i)
            const float al = m_w, be = -mhx * m_w * m_w, ga = -mhy * m_w * m_w;
            const float pa = a.projmatrix_raw[0], pb = a.projmatrix_raw[5], pe = a.projmatrix_raw[11];
            const f3 pC = xform_point_4x3(mean, vm);   // unclamped (Q17 iii)
            const f3 d1 = mk3(al * pa, 0.f, be * pe), d2 = mk3(0.f, al * pb, ga * pe);
            // (-[pC]x)^T d = [pC]x d = pC x d
            const f3 d1t = mk3(pC.y * d1.z - pC.z * d1.y, pC.z * d1.x - pC.x * d1.z, pC.x * d1.y - pC.y * d1.x);
            const f3 d2t = mk3(pC.y * d2.z - pC.z * d2.y, pC.z * d2.x - pC.x * d2.z, pC.x * d2.y - pC.y * d2.x);
            dtau[0] += g_m2x * d1.x + g_m2y * d2.x;
            dtau[1] += g_m2x * d1.y + g_m2y * d2.y;
            dtau[2] += g_m2x * d1.z + g_m2y * d2.z;
            dtau[3] += g_m2x * d1t.x + g_m2y * d2t.x;
            dtau[4] += g_m2x * d1t.y + g_m2y * d2t.y;
            dtau[5] += g_m2x * d1t.z + g_m2y * d2t.z;
            // depth, :518-528: z-row of I and of -[pC]x = (pC.y, -pC.x, 0)
            dmean[0] += g_d * vm[2]; dmean[1] += g_d * vm[6]; dmean[2] += g_d * vm[10];
            dtau[2] += g_d;
            dtau[3] += g_d * pC.y;
            dtau[4] += g_d * -pC.x;
        }

        // ---- backward.cu:21-145: colour -> SH coefficients and (through the view direction) the mean ----
        if (has_sh && !sh_staged) {
            const ShView sh = sh_view(a.shs, R, i, a.M);
            const uint32_t cb = clamp_bits;
            const float dRGB[3] = {(cb & 1u) ? 0.f : g_r, (cb & 2u) ? 0.f : g_g, (cb & 4u) ? 0.f : g_b};   // :32-35
            const f3 dir_orig = mk3(mean.x - a.campos[0], mean.y - a.campos[1], mean.z - a.campos[2]);
            const float inv = 1.0f / sqrtf(dot3(dir_orig, dir_orig));
            const float x = dir_orig.x * inv, y = dir_orig.y * inv, z = dir_orig.z * inv;
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;   // dL/ddir
            const int deg = a.D;
            const int used = (deg + 1) * (deg + 1);
            if (!a.accumulate && !a.pose_only) for (int k = used * 3; k < a.M * 3; k++) dsh.rest[k - 3] = 0.f;   // bands above the active degree (k >= 3: plain stores, see above)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float g = dRGB[k];
                float rx = 0.f, ry = 0.f, rz = 0.f;   // dRGB/d{x,y,z} for this channel
                dsh[k] = SH_C0 * g;
                if (deg > 0) {
                    dsh[3 + k] = -SH_C1 * y * g; dsh[6 + k] = SH_C1 * z * g; dsh[9 + k] = -SH_C1 * x * g;
                    rx = -SH_C1 * sh[9 + k]; ry = -SH_C1 * sh[3 + k]; rz = SH_C1 * sh[6 + k];
                    if (deg > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        dsh[12 + k] = SH_C2[0] * xy * g; dsh[15 + k] = SH_C2[1] * yz * g; dsh[18 + k] = SH_C2[2] * (2.f * zz - xx - yy) * g;
                        dsh[21 + k] = SH_C2[3] * xz * g; dsh[24 + k] = SH_C2[4] * (xx - yy) * g;
                        rx += SH_C2[0] * y * sh[12 + k] + SH_C2[2] * 2.f * -x * sh[18 + k] + SH_C2[3] * z * sh[21 + k] + SH_C2[4] * 2.f * x * sh[24 + k];
                        ry += SH_C2[0] * x * sh[12 + k] + SH_C2[1] * z * sh[15 + k] + SH_C2[2] * 2.f * -y * sh[18 + k] + SH_C2[4] * 2.f * -y * sh[24 + k];
                        rz += SH_C2[1] * y * sh[15 + k] + SH_C2[2] * 2.f * 2.f * z * sh[18 + k] + SH_C2[3] * x * sh[21 + k];
                        if (deg > 2) {
                            dsh[27 + k] = SH_C3[0] * y * (3.f * xx - yy) * g; dsh[30 + k] = SH_C3[1] * xy * z * g;
                            dsh[33 + k] = SH_C3[2] * y * (4.f * zz - xx - yy) * g; dsh[36 + k] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                            dsh[39 + k] = SH_C3[4] * x * (4.f * zz - xx - yy) * g; dsh[42 + k] = SH_C3[5] * z * (xx - yy) * g;
                            dsh[45 + k] = SH_C3[6] * x * (xx - 3.f * yy) * g;
                            rx += SH_C3[0] * sh[27 + k] * 3.f * 2.f * xy + SH_C3[1] * sh[30 + k] * yz + SH_C3[2] * sh[33 + k] * -2.f * xy +
                                  SH_C3[3] * sh[36 + k] * -3.f * 2.f * xz + SH_C3[4] * sh[39 + k] * (-3.f * xx + 4.f * zz - yy) +
                                  SH_C3[5] * sh[42 + k] * 2.f * xz + SH_C3[6] * sh[45 + k] * 3.f * (xx - yy);
                            ry += SH_C3[0] * sh[27 + k] * 3.f * (xx - yy) + SH_C3[1] * sh[30 + k] * xz + SH_C3[2] * sh[33 + k] * (-3.f * yy + 4.f * zz - xx) +
                                  SH_C3[3] * sh[36 + k] * -3.f * 2.f * yz + SH_C3[4] * sh[39 + k] * -2.f * xy + SH_C3[5] * sh[42 + k] * -2.f * yz +
                                  SH_C3[6] * sh[45 + k] * -3.f * 2.f * xy;
                            rz += SH_C3[1] * sh[30 + k] * xy + SH_C3[2] * sh[33 + k] * 4.f * 2.f * yz + SH_C3[3] * sh[36 + k] * 3.f * (2.f * zz - xx - yy) +
                                  SH_C3[4] * sh[39 + k] * 4.f * 2.f * xz + SH_C3[5] * sh[42 + k] * (xx - yy);
                        }
                    }
                }
                ddx += rx * g; ddy += ry * g; ddz += rz * g;   // :131
            }
            // dnormvdv, auxiliary.h:107-117
            const f3 v = dir_orig;
            const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            const float mx = ((+sum2 - v.x * v.x) * ddx - v.y * v.x * ddy - v.z * v.x * ddz) * invsum32;
            const float my = (-v.x * v.y * ddx + (sum2 - v.y * v.y) * ddy - v.z * v.y * ddz) * invsum32;
            const float mz = (-v.x * v.z * ddx - v.y * v.z * ddy + (sum2 - v.z * v.z) * ddz) * invsum32;
            dmean[0] += mx; dmean[1] += my; dmean[2] += mz;       // :139
            dtau[0] -= mx; dtau[1] -= my; dtau[2] -= mz;          // :141-143 (Q17 ii)
        }

        // ---- backward.cu:350-413: cov3D -> scale, quaternion (no normalisation backward, Q1) ----
        if (want_cov_chain) {
            const float r = q4[0], x = q4[1], y = q4[2], z = q4[3];
            const float Rq[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                                    {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                                    {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
            const float s[3] = {a.scale_modifier * s3[0], a.scale_modifier * s3[1], a.scale_modifier * s3[2]};
            const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]}, {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]}, {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            float Dm[3][3];   // Dm[k][j] = (2 M dSigma)[k][j] * s_k with M[k][j] = s_k Rq[j][k]
#pragma unroll
            for (int k = 0; k < 3; k++) {
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float dM = 2.0f * (s[k] * Rq[0][k] * dS[0][j] + s[k] * Rq[1][k] * dS[1][j] + s[k] * Rq[2][k] * dS[2][j]);
                    Dm[k][j] = dM;
                }
                dscale[k] = Rq[0][k] * Dm[k][0] + Rq[1][k] * Dm[k][1] + Rq[2][k] * Dm[k][2];   // :394-397 (not multiplied by scale_modifier)
#pragma unroll
                for (int j = 0; j < 3; j++) Dm[k][j] *= s[k];                                     // :399-401
            }
            drot[0] = 2 * z * (Dm[0][1] - Dm[1][0]) + 2 * y * (Dm[2][0] - Dm[0][2]) + 2 * x * (Dm[1][2] - Dm[2][1]);   // :405-408
            drot[1] = 2 * y * (Dm[1][0] + Dm[0][1]) + 2 * z * (Dm[2][0] + Dm[0][2]) + 2 * r * (Dm[1][2] - Dm[2][1]) - 4 * x * (Dm[2][2] + Dm[1][1]);
            drot[2] = 2 * x * (Dm[1][0] + Dm[0][1]) + 2 * r * (Dm[2][0] - Dm[0][2]) + 2 * z * (Dm[1][2] + Dm[2][1]) - 4 * y * (Dm[2][2] + Dm[0][0]);
            drot[3] = 2 * r * (Dm[0][1] - Dm[1][0]) + 2 * x * (Dm[2][0] + Dm[0][2]) + 2 * y * (Dm[1][2] + Dm[2][1]) - 4 * z * (Dm[1][1] + Dm[0][0]);
        }
    }
    if (sh_staged) {
        // ---- backward.cu:21-145 with the coefficient rows moved by the wave (helpers at the top of the file). Control flow is wave-uniform;
        // `visible` lanes evaluate, the row copies cover the lanes their masks name. Order of the floating-point operations: as in the
        // per-lane path above (per channel: degree 1, 2, 3 terms of dRGB/ddir in that order; the channels' contributions added in channel order).
        float* const win = reinterpret_cast<float*>(s_slot[wv]);
        const int deg = a.D, ncoef = a.M * 3;
        const ShView sh = sh_view(a.shs, R, i, a.M);
        const uint32_t cb = clamp_bits;
        const float dRGB[3] = {(cb & 1u) || !visible ? 0.f : g_r, (cb & 2u) || !visible ? 0.f : g_g, (cb & 4u) || !visible ? 0.f : g_b};   // :32-35
        const f3 dir_orig = mk3(mean.x - a.campos[0], mean.y - a.campos[1], mean.z - a.campos[2]);
        const float inv = 1.0f / sqrtf(dot3(dir_orig, dir_orig));
        const float x = dir_orig.x * inv, y = dir_orig.y * inv, z = dir_orig.z * inv;
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        float rx[3] = {0.f, 0.f, 0.f}, ry[3] = {0.f, 0.f, 0.f}, rz[3] = {0.f, 0.f, 0.f};      // dRGB/d{x,y,z} per channel
        const unsigned long long rows_rd = __ballot(visible);
        if (deg > 0 && rows_rd) {
            stage_rows(win, rows_rd, sh.rest, 0, deg == 1 ? 9 : 24);                          // coefficients 3 .. 11 (degree 1) or 3 .. 26 (degrees 1 + 2)
            if (visible) {
                const float* c = win + lane * SH_WIN_STRIDE - 3;                                // c[k] = coefficient k of this Gaussian
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    rx[k] = -SH_C1 * c[9 + k]; ry[k] = -SH_C1 * c[3 + k]; rz[k] = SH_C1 * c[6 + k];
                    if (deg > 1) {
                        rx[k] += SH_C2[0] * y * c[12 + k] + SH_C2[2] * 2.f * -x * c[18 + k] + SH_C2[3] * z * c[21 + k] + SH_C2[4] * 2.f * x * c[24 + k];
                        ry[k] += SH_C2[0] * x * c[12 + k] + SH_C2[1] * z * c[15 + k] + SH_C2[2] * 2.f * -y * c[18 + k] + SH_C2[4] * 2.f * -y * c[24 + k];
                        rz[k] += SH_C2[1] * y * c[15 + k] + SH_C2[2] * 2.f * 2.f * z * c[18 + k] + SH_C2[3] * x * c[21 + k];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (deg > 2) {
                stage_rows(win, rows_rd, sh.rest, 24, 21);                                      // coefficients 27 .. 47
                if (visible) {
                    const float* c = win + lane * SH_WIN_STRIDE - 27;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        rx[k] += SH_C3[0] * c[27 + k] * 3.f * 2.f * xy + SH_C3[1] * c[30 + k] * yz + SH_C3[2] * c[33 + k] * -2.f * xy +
                                 SH_C3[3] * c[36 + k] * -3.f * 2.f * xz + SH_C3[4] * c[39 + k] * (-3.f * xx + 4.f * zz - yy) +
                                 SH_C3[5] * c[42 + k] * 2.f * xz + SH_C3[6] * c[45 + k] * 3.f * (xx - yy);
                        ry[k] += SH_C3[0] * c[27 + k] * 3.f * (xx - yy) + SH_C3[1] * c[30 + k] * xz + SH_C3[2] * c[33 + k] * (-3.f * yy + 4.f * zz - xx) +
                                 SH_C3[3] * c[36 + k] * -3.f * 2.f * yz + SH_C3[4] * c[39 + k] * -2.f * xy + SH_C3[5] * c[42 + k] * -2.f * yz +
                                 SH_C3[6] * c[45 + k] * -3.f * 2.f * xy;
                        rz[k] += SH_C3[1] * c[30 + k] * xy + SH_C3[2] * c[33 + k] * 4.f * 2.f * yz + SH_C3[3] * c[36 + k] * 3.f * (2.f * zz - xx - yy) +
                                 SH_C3[4] * c[39 + k] * 4.f * 2.f * xz + SH_C3[5] * c[42 + k] * (xx - yy);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (visible) {
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;   // dL/ddir
#pragma unroll
            for (int k = 0; k < 3; k++) { ddx += rx[k] * dRGB[k]; ddy += ry[k] * dRGB[k]; ddz += rz[k] * dRGB[k]; }   // :131
            // dnormvdv, auxiliary.h:107-117
            const f3 v = dir_orig;
            const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            const float mx = ((+sum2 - v.x * v.x) * ddx - v.y * v.x * ddy - v.z * v.x * ddz) * invsum32;
            const float my = (-v.x * v.y * ddx + (sum2 - v.y * v.y) * ddy - v.z * v.y * ddz) * invsum32;
            const float mz = (-v.x * v.z * ddx - v.y * v.z * ddy + (sum2 - v.z * v.z) * ddz) * invsum32;
            dmean[0] += mx; dmean[1] += my; dmean[2] += mz;       // :139
            dtau[0] -= mx; dtau[1] -= my; dtau[2] -= mz;          // :141-143 (Q17 ii)
        }
        // ---- the coefficient gradients: DC per lane (three floats), the other bands as rows; rows of invisible Gaussians are zero (left alone in
        // accumulate mode), coefficients above the active degree are zero
        if (!a.pose_only) {
            const bool wr = in_range && (visible || !a.accumulate);
            if (wr) {
#pragma unroll
                for (int k = 0; k < 3; k++) dsh[k] = SH_C0 * dRGB[k];                         // (an invisible Gaussian's dRGB is zero)
            }
            const bool have_rest = RAW ? RG.f_rest != nullptr : a.dL_dsh != nullptr;         // (uniform: the row masks live in scalar registers)
            const unsigned long long rows_wr = have_rest ? __ballot(wr) : 0ull;
            // (accumulate mode adds nothing above the active degree, as the per-lane path: the rows end at the degree's last coefficient)
            const int ncols = a.accumulate ? (deg + 1) * (deg + 1) * 3 : ncoef;
            if (rows_wr && ncols > 3) {
                float* const w = win + lane * SH_WIN_STRIDE;
                // (an invisible Gaussian's row is zero by SELECTION, not by multiplication: its direction is not a number when the camera sits at the origin)
                const bool d1 = visible && deg > 0, d2 = visible && deg > 1, d3 = visible && deg > 2;
                const int w1 = min(24, ncols - 3);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float g = dRGB[k];
                    w[0 + k] = d1 ? -SH_C1 * y * g : 0.f; w[3 + k] = d1 ? SH_C1 * z * g : 0.f; w[6 + k] = d1 ? -SH_C1 * x * g : 0.f;
                    if (w1 > 9) {
                        w[9 + k] = d2 ? SH_C2[0] * xy * g : 0.f; w[12 + k] = d2 ? SH_C2[1] * yz * g : 0.f; w[15 + k] = d2 ? SH_C2[2] * (2.f * zz - xx - yy) * g : 0.f;
                        w[18 + k] = d2 ? SH_C2[3] * xz * g : 0.f; w[21 + k] = d2 ? SH_C2[4] * (xx - yy) * g : 0.f;
                    }
                }
                flush_rows(win, rows_wr, dsh.rest, 0, w1, a.accumulate != 0);
                if (ncols > 27) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const float g = dRGB[k];
                        w[0 + k] = d3 ? SH_C3[0] * y * (3.f * xx - yy) * g : 0.f; w[3 + k] = d3 ? SH_C3[1] * xy * z * g : 0.f;
                        w[6 + k] = d3 ? SH_C3[2] * y * (4.f * zz - xx - yy) * g : 0.f; w[9 + k] = d3 ? SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g : 0.f;
                        w[12 + k] = d3 ? SH_C3[4] * x * (4.f * zz - xx - yy) * g : 0.f; w[15 + k] = d3 ? SH_C3[5] * z * (xx - yy) * g : 0.f;
                        w[18 + k] = d3 ? SH_C3[6] * x * (xx - 3.f * yy) * g : 0.f;
                    }
                    flush_rows(win, rows_wr, dsh.rest, 24, 21, a.accumulate != 0);
                }
            }
        }
    }
    GEO_TICK(4);
    // The pose gradient is the sum of dL_dtau over all Gaussians (DGR/diff_gaussian_rasterization/__init__.py:152-154 does it with
    // torch.sum on [P,6]); the first two levels are folded into this kernel in a fixed order: the wave (transposed butterfly), the
    // block (four waves through LDS); one row per block, summed by tau_sum_kernel. Folding that last level in as well -- the block
    // that finishes last, found with a ticket -- was built twice: with an agent-scope release per block (round 2: an L2 write-back on
    // this multi-XCD part, 100 us) and with write-through row stores + one atomic per block and no fence (round 3: correct, but the
    // store -> ticket -> row reads chain of the last block is three memory round trips, as long as the second launch it replaces).
    if (a.tau_partials) {
        __shared__ float s_tau[4][6];
        const WaveSelectMasks wsm = wave_select_masks();
        unsigned long long dummy_proc = 0; uint32_t dummy_addr;
        const float z = 0.f;
        const float tot = wave_sum10_transposed(wsm, in_range ? dtau[0] : z, f2v{in_range ? dtau[1] : z, in_range ? dtau[2] : z},
                                                f2v{in_range ? dtau[3] : z, in_range ? dtau[4] : z}, in_range ? dtau[5] : z, f2v{z, z}, f2v{z, z},
                                                dummy_proc, 0, 0u, 0, dummy_addr);
        // lanes holding total k (wave_sum10_slot_of_lane): 0 -> 2, 1 -> 0, 2 -> 1, 3 -> 32, 4 -> 33, 5 -> 34
        const int wv = threadIdx.x >> 6;
        if (lane == 2) s_tau[wv][0] = tot;
        if (lane == 0) s_tau[wv][1] = tot;
        if (lane == 1) s_tau[wv][2] = tot;
        if (lane == 32) s_tau[wv][3] = tot;
        if (lane == 33) s_tau[wv][4] = tot;
        if (lane == 34) s_tau[wv][5] = tot;
        __syncthreads();
        if (threadIdx.x < 6) a.tau_partials[(size_t)blockIdx.x * 6 + threadIdx.x] = (s_tau[0][threadIdx.x] + s_tau[1][threadIdx.x]) + (s_tau[2][threadIdx.x] + s_tau[3][threadIdx.x]);
    }
    GEO_TICK(5);
    if (in_range) {
        const bool wr = !a.pose_only && (visible || !a.accumulate);     // accumulate mode leaves the rows of invisible Gaussians alone
        if (wr)
            a.dL_dopacity[o] = add_separately(old_o, RAW ? [&] { const float sg = load_opacity(nullptr, R, i); return g_op * sg * (1.0f - sg); }() : g_op);   // raw: through the sigmoid
        if (wr) {
#pragma unroll
            for (int k = 0; k < 3; k++) a.dL_dmean3D[3 * o + k] = add_separately(old_m[k], dmean[k]);
        }
#pragma unroll
        for (int k = 0; k < 6; k++) if (a.dL_dcov3D) a.dL_dcov3D[6 * i + k] = dcov[k];
#pragma unroll
        for (int k = 0; k < 6; k++) if (a.dL_dtau) a.dL_dtau[6 * i + k] = dtau[k];
        if (!RAW) {
            if (a.dL_dscale && wr) {
#pragma unroll
                for (int k = 0; k < 3; k++) a.dL_dscale[3 * i + k] = add_separately(old_s[k], dscale[k]);
            }
            if (a.dL_drot && wr) {
#pragma unroll
                for (int k = 0; k < 4; k++) a.dL_drot[4 * i + k] = add_separately(old_r[k], drot[k]);
            }
        } else {
            // chain rules of the fused prologue (gaussian_model.py:60-68): exp, normalize; the deltas' gradients are the
            // effective parameters' gradients of their Gaussian (one Gaussian per slot: plain stores)
            if constexpr (PRE) {
                // deformation-network deltas (added in front of the activations): the effective raw parameters are _scaling + ds and
                // _rotation + dr; a delta's gradient is its parameter's gradient of this view (before any accumulation). Every row of
                // ddx / dds / ddr is written (zeros for the Gaussians this view does not see).
                const int sl = pre_slot(R, o);
                const float* pds = sl >= 0 && R.ds ? R.ds + (size_t)pre_stride(R, 3) * sl : nullptr;
                const float* pdr = sl >= 0 && R.dr ? R.dr + (size_t)pre_stride(R, 4) * sl : nullptr;
                float gls[3];
#pragma unroll
                for (int k = 0; k < 3; k++) gls[k] = visible ? dscale[k] * expf(R.log_scales[R.scale_dim == 1 ? o : 3 * o + k] + (pds ? pds[k] : 0.f)) : 0.f;
                float ra = R.raw_rot[4 * o], rb = R.raw_rot[4 * o + 1], rc = R.raw_rot[4 * o + 2], rd = R.raw_rot[4 * o + 3];
                if (pdr) { ra += pdr[0]; rb += pdr[1]; rc += pdr[2]; rd += pdr[3]; }
                const float inv = 1.0f / fmaxf(sqrtf(ra * ra + rb * rb + rc * rc + rd * rd), 1e-12f);
                const float qa = ra * inv, qb = rb * inv, qc = rc * inv, qd = rd * inv;
                const float dotg = qa * drot[0] + qb * drot[1] + qc * drot[2] + qd * drot[3];
                const float gr[4] = {(drot[0] - qa * dotg) * inv, (drot[1] - qb * dotg) * inv, (drot[2] - qc * dotg) * inv, (drot[3] - qd * dotg) * inv};
                if (wr) {
                    if (R.scale_dim == 1) a.dL_dscale[o] = add_separately(old_s[0], gls[0] + gls[1] + gls[2]);
                    else {
#pragma unroll
                        for (int k = 0; k < 3; k++) a.dL_dscale[3 * o + k] = add_separately(old_s[k], gls[k]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) a.dL_drot[4 * o + k] = add_separately(old_r[k], gr[k]);
                }
                if (sl >= 0) {
                    const int w3 = pre_stride(R, 3), w4 = pre_stride(R, 4);
                    if (RG.ddx) { float* d = RG.ddx + (size_t)w3 * sl; d[0] = dmean[0]; d[1] = dmean[1]; d[2] = dmean[2]; }
                    if (RG.dds) { float* d = RG.dds + (size_t)w3 * sl; d[0] = gls[0]; d[1] = gls[1]; d[2] = gls[2]; }
                    if (RG.ddr) { float* d = RG.ddr + (size_t)w4 * sl; d[0] = gr[0]; d[1] = gr[1]; d[2] = gr[2]; d[3] = gr[3]; }
                }
                return;
            }
            const int sl = raw_slot(R, o);
            if (sl >= 0 && flow) {
                // render_flow (gaussian_renderer/__init__.py:262-284): the colour (g_r, g_g) reaches dx through -ndc(.; proj1) and dx2
                // through +ndc(.; proj2); the position itself is detached on that path (:262), so dL_dmean3D keeps the geometric part only
                const f3 base = mk3(R.xyz[3 * o], R.xyz[3 * o + 1], R.xyz[3 * o + 2]);
                f3 t1 = base, t2 = base;
                if (R.dx) { t1.x += R.dx[3 * sl]; t1.y += R.dx[3 * sl + 1]; t1.z += R.dx[3 * sl + 2]; }
                if (R.flow_dx2) { t2.x += R.flow_dx2[3 * sl]; t2.y += R.flow_dx2[3 * sl + 1]; t2.z += R.flow_dx2[3 * sl + 2]; }
                const f3 j1 = visible ? flow_ndc_vjp(R.flow_proj1, t1, g_r, g_g) : mk3(0.f, 0.f, 0.f);
                const f3 j2 = visible ? flow_ndc_vjp(R.flow_proj2, t2, g_r, g_g) : mk3(0.f, 0.f, 0.f);
                if (RG.ddx2) { RG.ddx2[3 * sl] = j2.x; RG.ddx2[3 * sl + 1] = j2.y; RG.ddx2[3 * sl + 2] = j2.z; }
                if (RG.ddx) { RG.ddx[3 * sl] = dmean[0] - j1.x; RG.ddx[3 * sl + 1] = dmean[1] - j1.y; RG.ddx[3 * sl + 2] = dmean[2] - j1.z; }
                if (RG.dds) { RG.dds[3 * sl] = dscale[0]; RG.dds[3 * sl + 1] = dscale[1]; RG.dds[3 * sl + 2] = dscale[2]; }
                if (RG.ddr) { RG.ddr[4 * sl] = drot[0]; RG.ddr[4 * sl + 1] = drot[1]; RG.ddr[4 * sl + 2] = drot[2]; RG.ddr[4 * sl + 3] = drot[3]; }
            } else if (sl >= 0) {
                if (RG.ddx) { RG.ddx[3 * sl] = dmean[0]; RG.ddx[3 * sl + 1] = dmean[1]; RG.ddx[3 * sl + 2] = dmean[2]; }
                if (RG.dds) { RG.dds[3 * sl] = dscale[0]; RG.dds[3 * sl + 1] = dscale[1]; RG.dds[3 * sl + 2] = dscale[2]; }
                if (RG.ddr) { RG.ddr[4 * sl] = drot[0]; RG.ddr[4 * sl + 1] = drot[1]; RG.ddr[4 * sl + 2] = drot[2]; RG.ddr[4 * sl + 3] = drot[3]; }
            }
            if (!wr) {
                // accumulate mode, invisible Gaussian: its parameter-gradient rows stay as they are
            } else if (R.scale_dim == 1) {
                a.dL_dscale[o] = add_separately(old_s[0], (dscale[0] + dscale[1] + dscale[2]) * expf(R.log_scales[o]));
            } else {
#pragma unroll
                for (int k = 0; k < 3; k++) a.dL_dscale[3 * o + k] = add_separately(old_s[k], dscale[k] * expf(R.log_scales[3 * o + k]));
            }
            const float ra = R.raw_rot[4 * o], rb = R.raw_rot[4 * o + 1], rc = R.raw_rot[4 * o + 2], rd = R.raw_rot[4 * o + 3];
            const float inv = 1.0f / fmaxf(sqrtf(ra * ra + rb * rb + rc * rc + rd * rd), 1e-12f);
            const float qa = ra * inv, qb = rb * inv, qc = rc * inv, qd = rd * inv;
            const float dotg = qa * drot[0] + qb * drot[1] + qc * drot[2] + qd * drot[3];
            if (wr) {
                a.dL_drot[4 * o] = add_separately(old_r[0], (drot[0] - qa * dotg) * inv); a.dL_drot[4 * o + 1] = add_separately(old_r[1], (drot[1] - qb * dotg) * inv);
                a.dL_drot[4 * o + 2] = add_separately(old_r[2], (drot[2] - qc * dotg) * inv); a.dL_drot[4 * o + 3] = add_separately(old_r[3], (drot[3] - qd * dotg) * inv);
            }
        }
    }
    GEO_TICK(6);
}

// Last level of the pose-gradient sum: one block, 6 x 64 threads, fixed order.
__device__ __forceinline__ void tau_sum_body(int nblocks, const float* __restrict__ partials, float* __restrict__ out6)
{
    const int k = threadIdx.x >> 6, lane = lane_id();
    float v = 0.f;
    for (int b0 = lane; b0 < nblocks; b0 += 64 * 8) {          // eight of a lane's terms in flight per trip; added in the order of b (782 blocks at config #2:
        float t[8];                                             // thirteen dependent round trips otherwise)
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = b0 + 64 * j < nblocks ? partials[(size_t)(b0 + 64 * j) * 6 + k] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) if (b0 + 64 * j < nblocks) v += t[j];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) out6[k] = v;
}

// ---- the single-view kernels: the bodies above with their arguments passed by value (gs_views.h launches the same bodies once for
// several views) ----------------------------------------------------------------------------------------------------------------------
template <bool RAW, bool PRE = false>
__global__ void __launch_bounds__(256) geometry_bwd_kernel(GeomBwdArgs a)
{
    geometry_bwd_body<RAW, PRE>(a);
}

__global__ void __launch_bounds__(384) tau_sum_kernel(int nblocks, const float* __restrict__ partials, float* __restrict__ out6)
{
    tau_sum_body(nblocks, partials, out6);
}

}  // namespace gsr
