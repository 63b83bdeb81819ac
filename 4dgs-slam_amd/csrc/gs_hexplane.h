// gs_hexplane.h -- the HexPlane feature field of the deformation network (include/deformation_field.h), forward and backward.
//
// Reference: gaussian_splatting/utils/hexplane.py:19-22 (normalize_aabb), :23-50 (grid_sample_wrapper -> F.grid_sample bilinear /
// border / align_corners=True), :81-112 (interpolate_ms_features), :162-188 (HexPlaneField).  The reference runs 24 grid_sample
// launches, each of which -- in its [C][H][W] layout -- reads C scattered dwords per corner, then 20 elementwise products and a
// concat; autograd replays all of it with one atomic per (corner, channel).
//
// Here one launch does the whole field.  A point is owned by LPP = C/4 adjacent lanes, lane `sub` holding channels 4 sub .. 4 sub+3:
// with the channels-last plane layout a corner is ONE C*4-byte run that the LPP lanes fetch as float4 each (C = 32: one 128-byte
// line per corner, 8 points per wave instruction), the six samples of a level are multiplied in registers and the product is
// stored as a float4 into the point's feature row (again one contiguous run per point).  Per point and level that is 24 independent
// 16-byte loads in flight per lane, which is what a gather that lives in L2/MALL wants.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/deformation_field.h"

namespace gsr {

constexpr int HEX_BLOCK = 256;

struct HexAxis {        // one coordinate of one plane: F.grid_sample's unnormalise + border clip (GridSampler.h)
    int i0;             // floor of the clipped source index
    float w0, w1;       // weights of i0 and i0 + 1:  (i0 + 1) - u,  u - i0
    bool has1;          // i0 + 1 is inside the plane
    float dmult;        // d u / d coord: (size - 1) / 2 strictly inside, 0 on and outside the border (clip_coordinates_set_grad)
};

__device__ __forceinline__ HexAxis hex_axis(float coord, int size)
{
    HexAxis a;
    const float hi = (float)(size - 1);
    float u = ((coord + 1.f) * 0.5f) * hi;                       // grid_sampler_unnormalize, align_corners = true
    const bool inside = u > 0.f && u < hi;
    a.dmult = inside ? 0.5f * hi : 0.f;
    u = fminf(fmaxf(u, 0.f), hi);                                // clip_coordinates (a NaN coordinate ends at 0, like ATen's min/max)
    const float f = floorf(u);
    a.i0 = (int)f;
    a.w0 = (f + 1.f) - u;                                        // ATen: (ix_se - ix)
    a.w1 = u - f;                                                //       (ix - ix_nw)
    a.has1 = a.i0 + 1 < size;
    return a;
}

// plane index -> the two coordinates it spans, itertools.combinations(range(4), 2) order (hexplane.py:86-88)
__device__ __forceinline__ constexpr int hex_c0(int p) { return p < 3 ? 0 : (p < 5 ? 1 : 2); }
__device__ __forceinline__ constexpr int hex_c1(int p) { return p == 0 ? 1 : (p == 1 || p == 3) ? 2 : 3; }

// How the C channels of a texel are spread over the LPP = C/4 lanes of a point, four per lane:
//   PLANAR   reference memory [C][H][W]:      lane `sub` holds channels 4 sub + k, k = 0..3, each H*W floats apart
//   VEC      channels-last memory [H][W][C]:  lane `sub` holds channels 4 sub + k as one float4 (16-byte loads)
enum HexLayout { HEX_PLANAR = 0, HEX_VEC = 1 };

template <int MODE>
struct HexCorner {      // element offset of this lane's first channel of texel (y, x), and the stride to its next channel
    __device__ __forceinline__ static size_t offset(int y, int x, int W, int H, int C, int sub)
    {
        return MODE == HEX_PLANAR ? ((size_t)(4 * sub) * H + y) * W + x : ((size_t)y * W + x) * C + 4 * sub;
    }
    __device__ __forceinline__ static size_t cstride(int W, int H, int C) { return MODE == HEX_PLANAR ? (size_t)W * H : 1; }
};

template <int MODE>
__device__ __forceinline__ float4 hex_load4(const float* __restrict__ plane, size_t off, size_t cstride)
{
    if (MODE == HEX_VEC) return *reinterpret_cast<const float4*>(plane + off);
    return make_float4(plane[off], plane[off + cstride], plane[off + 2 * cstride], plane[off + 3 * cstride]);
}

__device__ __forceinline__ float4 fma4(float4 a, float s, float4 c) { return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w)); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// bilinear sample of one plane at (ax, ay) for this lane's four channels: ((nw + ne) + sw) + se, ATen's accumulation order
template <int CL>
__device__ __forceinline__ float4 hex_sample(const float* __restrict__ plane, const HexAxis& ax, const HexAxis& ay, int W, int H, int C, int sub)
{
    const size_t cs = HexCorner<CL>::cstride(W, H, C);
    const int x1 = ax.has1 ? ax.i0 + 1 : ax.i0, y1 = ay.has1 ? ay.i0 + 1 : ay.i0;   // an outside corner has weight 0: read a valid texel
    const float wx1 = ax.has1 ? ax.w1 : 0.f, wy1 = ay.has1 ? ay.w1 : 0.f;
    const float4 nw = hex_load4<CL>(plane, HexCorner<CL>::offset(ay.i0, ax.i0, W, H, C, sub), cs);
    const float4 ne = hex_load4<CL>(plane, HexCorner<CL>::offset(ay.i0, x1, W, H, C, sub), cs);
    const float4 sw = hex_load4<CL>(plane, HexCorner<CL>::offset(y1, ax.i0, W, H, C, sub), cs);
    const float4 se = hex_load4<CL>(plane, HexCorner<CL>::offset(y1, x1, W, H, C, sub), cs);
    float4 v = nw * (ax.w0 * ay.w0);
    v = fma4(ne, wx1 * ay.w0, v);
    v = fma4(sw, ax.w0 * wy1, v);
    v = fma4(se, wx1 * wy1, v);
    return v;
}

struct HexPoint {
    float c[4];         // normalised (x, y, z) and the raw time
    float dscale[3];    // d c[k] / d xyz[k]: 2 / (aabb1 - aabb0) where the clamp passes the gradient, else 0
};

__device__ __forceinline__ HexPoint hex_point(const float* __restrict__ aabb, const float* __restrict__ xyz, const float* __restrict__ time)
{
    HexPoint p;
    p.c[3] = time[0];                                            // concatenated after the normalisation (:166-167): not clamped
    if (!aabb) {                                                 // interpolate_ms_features on caller-normalised coordinates
#pragma unroll
        for (int k = 0; k < 3; k++) { p.c[k] = xyz[k]; p.dscale[k] = 1.f; }
        return p;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {                                // hexplane.py:19-22
        const float a0 = aabb[k], a1 = aabb[3 + k];
        const float s = 2.0f / (a1 - a0);
        const float v = (xyz[k] - a0) * s - 1.0f;
        p.dscale[k] = (v >= -1.f && v <= 1.f) ? s : 0.f;         // torch.clamp passes the gradient on the closed interval
        p.c[k] = fminf(fmaxf(v, -1.f), 1.f);
    }
    return p;
}

template <int LPP, int CL>
__global__ void __launch_bounds__(HEX_BLOCK)
hexplane_fwd_kernel(const gsr_hexplane_field f, const int64_t n, const float* __restrict__ xyz, const int64_t xyz_stride,
                    const float* __restrict__ time, const int64_t time_stride, float* __restrict__ features)
{
    constexpr int C = 4 * LPP;
    const int sub = threadIdx.x % LPP;
    const int64_t i = (int64_t)blockIdx.x * (HEX_BLOCK / LPP) + threadIdx.x / LPP;
    if (i >= n) return;
    const HexPoint p = hex_point(f.aabb, xyz + i * xyz_stride, time + i * time_stride);
    float* out = features + i * ((int64_t)f.num_levels * C) + 4 * sub;
    for (int l = 0; l < f.num_levels; l++) {
        const gsr_hexplane_level& L = f.levels[l];
        HexAxis ax[4];
#pragma unroll
        for (int k = 0; k < 4; k++) ax[k] = hex_axis(p.c[k], L.res[k]);
        float4 prod = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int pl = 0; pl < 6; pl++) {                          // interp_space = interp_space * interp_out_plane (:103)
            const int c0 = hex_c0(pl), c1 = hex_c1(pl);
            prod = prod * hex_sample<CL>(L.planes[pl], ax[c0], ax[c1], L.res[c0], L.res[c1], C, sub);
        }
        *reinterpret_cast<float4*>(out + (size_t)l * C) = prod;   // torch.cat over levels (:111)
    }
}

// ---- the views of one mapping iteration in one launch --------------------------------------------------------------------------------
// The mapping back-end evaluates the field at the SAME positions for every keyframe of an iteration, each with its own time
// (gaussian_renderer/__init__.py:112,149-157: `time = torch.tensor(viewpoint_camera.time).repeat(P, 1)`). Of the six samples of a level the
// three spatial planes (xy, xz, yz) depend on the position only: they are gathered ONCE per point and level -- they are the scattered,
// HBM-sized part of the field (128 of its 136 MB) -- and only the three time planes, whose rows of one time fit in L2 (2 rows x 960
// columns x 128 B x 3 planes = 0.7 MB per view), are sampled per view. The product is formed in the reference's plane order
// (hexplane.py:93-103), so every view's features are bit-identical to hexplane_fwd_kernel's.
constexpr int HEX_MAX_VIEWS = 12;
typedef float hex_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void hex_stream_store4(float* dst, float4 v)
{
    hex_f4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<hex_f4*>(dst));
}
struct HexTimes { int V; float t[HEX_MAX_VIEWS]; };

template <int LPP, int CL>
__global__ void __launch_bounds__(HEX_BLOCK)
hexplane_fwd_views_kernel(const gsr_hexplane_field f, const int64_t n, const float* __restrict__ xyz, const int64_t xyz_stride,
                          const HexTimes tv, float* __restrict__ features)
{
    constexpr int C = 4 * LPP;
    const int sub = threadIdx.x % LPP;
    const int64_t i = (int64_t)blockIdx.x * (HEX_BLOCK / LPP) + threadIdx.x / LPP;
    if (i >= n) return;
    const float zero_time = 0.f;
    const HexPoint p = hex_point(f.aabb, xyz + i * xyz_stride, &zero_time);
    const int64_t row = (int64_t)f.num_levels * C;
    float* out = features + i * row + 4 * sub;
    for (int l = 0; l < f.num_levels; l++) {
        const gsr_hexplane_level& L = f.levels[l];
        HexAxis ax[3];
#pragma unroll
        for (int k = 0; k < 3; k++) ax[k] = hex_axis(p.c[k], L.res[k]);
        const float4 s0 = hex_sample<CL>(L.planes[0], ax[0], ax[1], L.res[0], L.res[1], C, sub);
        const float4 s1 = hex_sample<CL>(L.planes[1], ax[0], ax[2], L.res[0], L.res[2], C, sub);
        const float4 s3 = hex_sample<CL>(L.planes[3], ax[1], ax[2], L.res[1], L.res[2], C, sub);
        const float4 s01 = (make_float4(1.f, 1.f, 1.f, 1.f) * s0) * s1;
        for (int v = 0; v < tv.V; v++) {
            const HexAxis at = hex_axis(tv.t[v], L.res[3]);
            const float4 s2 = hex_sample<CL>(L.planes[2], ax[0], at, L.res[0], L.res[3], C, sub);
            const float4 s4 = hex_sample<CL>(L.planes[4], ax[1], at, L.res[1], L.res[3], C, sub);
            const float4 s5 = hex_sample<CL>(L.planes[5], ax[2], at, L.res[2], L.res[3], C, sub);
            const float4 prod = (((s01 * s2) * s3) * s4) * s5;
            hex_stream_store4(out + (int64_t)v * n * row + (size_t)l * C, prod);    // read once, by the MLP: keep the planes in L2
        }
    }
}

__device__ __forceinline__ void hex_atomic_add4(float* __restrict__ g, size_t off, size_t cstride, float4 v)
{
    unsafeAtomicAdd(g + off, v.x);
    unsafeAtomicAdd(g + off + cstride, v.y);
    unsafeAtomicAdd(g + off + 2 * cstride, v.z);
    unsafeAtomicAdd(g + off + 3 * cstride, v.w);
}

// Backward, four channels per lane (used for the reference's planar layout; channels-last planes take hexplane_bwd_lane_kernel
// below): recompute the six samples of a level, form dL/dsample_p = dL/dfeature * prod_{q != p} sample_q with prefix / suffix
// products (no division: a sample may be 0), scatter it to the four corners of plane p with hardware float atomics, and gather
// the coordinate gradient  d sample / d (u, v)  from the same four corner values (ATen grid_sampler_2d_backward).
template <int LPP, int CL>
__global__ void __launch_bounds__(HEX_BLOCK)
hexplane_bwd_kernel(const gsr_hexplane_field f, const int64_t n, const float* __restrict__ xyz, const int64_t xyz_stride,
                    const float* __restrict__ time, const int64_t time_stride, const float* __restrict__ dL_dfeatures,
                    float* __restrict__ dL_dxyz)
{
    constexpr int C = 4 * LPP;
    const int sub = threadIdx.x % LPP;
    const int64_t i = (int64_t)blockIdx.x * (HEX_BLOCK / LPP) + threadIdx.x / LPP;
    if (i >= n) return;                                           // whole LPP groups leave together (HEX_BLOCK % LPP == 0)
    const HexPoint p = hex_point(f.aabb, xyz + i * xyz_stride, time + i * time_stride);
    const float* gout = dL_dfeatures + i * ((int64_t)f.num_levels * C) + 4 * sub;
    float gc[3] = {0.f, 0.f, 0.f};                                // dL / d normalised (x, y, z), this lane's channels only
    for (int l = 0; l < f.num_levels; l++) {
        const gsr_hexplane_level& L = f.levels[l];
        HexAxis ax[4];
#pragma unroll
        for (int k = 0; k < 4; k++) ax[k] = hex_axis(p.c[k], L.res[k]);
        const float4 g = *reinterpret_cast<const float4*>(gout + (size_t)l * C);
        float4 s[6];
#pragma unroll
        for (int pl = 0; pl < 6; pl++) s[pl] = hex_sample<CL>(L.planes[pl], ax[hex_c0(pl)], ax[hex_c1(pl)], L.res[hex_c0(pl)], L.res[hex_c1(pl)], C, sub);
        float4 suffix[6];                                         // suffix[p] = prod_{q > p} s[q]
        suffix[5] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int pl = 4; pl >= 0; pl--) suffix[pl] = suffix[pl + 1] * s[pl + 1];
        float4 prefix = g;                                        // g * prod_{q < p} s[q]
#pragma unroll
        for (int pl = 0; pl < 6; pl++) {
            const int c0 = hex_c0(pl), c1 = hex_c1(pl);
            const int W = L.res[c0], H = L.res[c1];
            const HexAxis& X = ax[c0];
            const HexAxis& Y = ax[c1];
            const float4 gs = prefix * suffix[pl];               // dL / d sample_pl
            prefix = prefix * s[pl];
            const size_t cs = HexCorner<CL>::cstride(W, H, C);
            const int x1 = X.has1 ? X.i0 + 1 : X.i0, y1 = Y.has1 ? Y.i0 + 1 : Y.i0;
            const size_t o_nw = HexCorner<CL>::offset(Y.i0, X.i0, W, H, C, sub), o_ne = HexCorner<CL>::offset(Y.i0, x1, W, H, C, sub);
            const size_t o_sw = HexCorner<CL>::offset(y1, X.i0, W, H, C, sub), o_se = HexCorner<CL>::offset(y1, x1, W, H, C, sub);
            float* gp = L.grad_planes[pl];
            if (gp) {                                             // safe_add_2d: corners outside the plane receive nothing
                hex_atomic_add4(gp, o_nw, cs, gs * (X.w0 * Y.w0));
                if (X.has1) hex_atomic_add4(gp, o_ne, cs, gs * (X.w1 * Y.w0));
                if (Y.has1) hex_atomic_add4(gp, o_sw, cs, gs * (X.w0 * Y.w1));
                if (X.has1 && Y.has1) hex_atomic_add4(gp, o_se, cs, gs * (X.w1 * Y.w1));
            }
            if (dL_dxyz && (X.dmult != 0.f || Y.dmult != 0.f)) {  // uniform over the LPP lanes of a point
                const float* plane = L.planes[pl];
                const float zero1x = X.has1 ? 1.f : 0.f, zero1y = Y.has1 ? 1.f : 0.f;
                const float nw = dot4(hex_load4<CL>(plane, o_nw, cs), gs);
                const float ne = dot4(hex_load4<CL>(plane, o_ne, cs), gs) * zero1x;
                const float sw = dot4(hex_load4<CL>(plane, o_sw, cs), gs) * zero1y;
                const float se = dot4(hex_load4<CL>(plane, o_se, cs), gs) * (zero1x * zero1y);
                // ATen: gix = -nw (y1 - v) + ne (y1 - v) - sw (v - y0) + se (v - y0);  giy = -nw (x1 - u) - ne (u - x0) + sw (x1 - u) + se (u - x0)
                const float gix = (ne - nw) * Y.w0 + (se - sw) * Y.w1;
                const float giy = (sw - nw) * X.w0 + (se - ne) * X.w1;
                if (c0 < 3) gc[c0] += gix * X.dmult;
                if (c1 < 3) gc[c1] += giy * Y.dmult;
            }
        }
    }
    if (dL_dxyz) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int d = 1; d < LPP; d <<= 1) gc[k] += __shfl_xor(gc[k], d, 64);
        }
        if (sub == 0) {
            dL_dxyz[3 * i + 0] = gc[0] * p.dscale[0];
            dL_dxyz[3 * i + 1] = gc[1] * p.dscale[1];
            dL_dxyz[3 * i + 2] = gc[2] * p.dscale[2];
        }
    }
}


// The same backward with ONE channel per lane (C lanes per point, channels-last memory only): every corner is one dword per
// lane, so that a float atomic instruction covers the C*4 contiguous bytes of a texel -- the request shape the L2 atomic units
// are fastest at.  Measured at 200k points, shipped geometry, fwd+bwd: float4-per-lane ownership (8 dwords spread over a 128-byte
// line per request) 9.1 ms; 32 contiguous bytes per request 4.7 ms; this kernel 2.5 ms; torch's grid_sample autograd 56 ms.
// Per-XCD private accumulation buffers change nothing (8.6 ms vs 9.1 ms for the first variant): the cost is per request sector,
// not cross-XCD line migration.
template <int C>
__global__ void __launch_bounds__(HEX_BLOCK)
hexplane_bwd_lane_kernel(const gsr_hexplane_field f, const int64_t n, const float* __restrict__ xyz, const int64_t xyz_stride,
                         const float* __restrict__ time, const int64_t time_stride, const float* __restrict__ dL_dfeatures,
                         float* __restrict__ dL_dxyz)
{
    const int ch = threadIdx.x % C;
    const int64_t slot = (int64_t)blockIdx.x * (HEX_BLOCK / C) + threadIdx.x / C;
    if (slot >= n) return;
    // Waves that run at the same time should not work on neighbouring points: spatially sorted input makes them hammer the same
    // texels with atomics at the same moment (measured: sorted 200k points 5.0 ms, shuffled 2.5 ms).  Visit the points in the order
    // of a fixed bijection of [0, n): multiplication by the prime 2^31 - 1 (coprime to every n below it) modulo n.
    const int64_t i = n < 2147483647ll ? (int64_t)(((unsigned long long)slot * 2147483647ull) % (unsigned long long)n) : slot;
    const HexPoint p = hex_point(f.aabb, xyz + i * xyz_stride, time + i * time_stride);
    const float* gout = dL_dfeatures + i * ((int64_t)f.num_levels * C) + ch;
    float gc[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < f.num_levels; l++) {
        const gsr_hexplane_level& L = f.levels[l];
        HexAxis ax[4];
#pragma unroll
        for (int k = 0; k < 4; k++) ax[k] = hex_axis(p.c[k], L.res[k]);
        const float g = gout[(size_t)l * C];
        float s[6], corner[6][4];
        size_t off[6][4];
#pragma unroll
        for (int pl = 0; pl < 6; pl++) {
            const int c0 = hex_c0(pl), c1 = hex_c1(pl);
            const int W = L.res[c0];
            const HexAxis& X = ax[c0];
            const HexAxis& Y = ax[c1];
            const int x1 = X.has1 ? X.i0 + 1 : X.i0, y1 = Y.has1 ? Y.i0 + 1 : Y.i0;
            off[pl][0] = ((size_t)Y.i0 * W + X.i0) * C + ch;
            off[pl][1] = ((size_t)Y.i0 * W + x1) * C + ch;
            off[pl][2] = ((size_t)y1 * W + X.i0) * C + ch;
            off[pl][3] = ((size_t)y1 * W + x1) * C + ch;
            const float* plane = L.planes[pl];
#pragma unroll
            for (int k = 0; k < 4; k++) corner[pl][k] = plane[off[pl][k]];
        }
#pragma unroll
        for (int pl = 0; pl < 6; pl++) {
            const HexAxis& X = ax[hex_c0(pl)];
            const HexAxis& Y = ax[hex_c1(pl)];
            const float wx1 = X.has1 ? X.w1 : 0.f, wy1 = Y.has1 ? Y.w1 : 0.f;
            float v = corner[pl][0] * (X.w0 * Y.w0);
            v = fmaf(corner[pl][1], wx1 * Y.w0, v);
            v = fmaf(corner[pl][2], X.w0 * wy1, v);
            s[pl] = fmaf(corner[pl][3], wx1 * wy1, v);
        }
        float suffix[6];
        suffix[5] = 1.f;
#pragma unroll
        for (int pl = 4; pl >= 0; pl--) suffix[pl] = suffix[pl + 1] * s[pl + 1];
        float prefix = g;
#pragma unroll
        for (int pl = 0; pl < 6; pl++) {
            const int c0 = hex_c0(pl), c1 = hex_c1(pl);
            const HexAxis& X = ax[c0];
            const HexAxis& Y = ax[c1];
            const float gs = prefix * suffix[pl];
            prefix *= s[pl];
            float* gp = L.grad_planes[pl];
            if (gp) {
                unsafeAtomicAdd(gp + off[pl][0], gs * (X.w0 * Y.w0));
                if (X.has1) unsafeAtomicAdd(gp + off[pl][1], gs * (X.w1 * Y.w0));
                if (Y.has1) unsafeAtomicAdd(gp + off[pl][2], gs * (X.w0 * Y.w1));
                if (X.has1 && Y.has1) unsafeAtomicAdd(gp + off[pl][3], gs * (X.w1 * Y.w1));
            }
            if (dL_dxyz) {
                const float nw = corner[pl][0] * gs, ne = X.has1 ? corner[pl][1] * gs : 0.f, sw = Y.has1 ? corner[pl][2] * gs : 0.f;
                const float se = X.has1 && Y.has1 ? corner[pl][3] * gs : 0.f;
                const float gix = (ne - nw) * Y.w0 + (se - sw) * Y.w1;
                const float giy = (sw - nw) * X.w0 + (se - ne) * X.w1;
                if (c0 < 3) gc[c0] += gix * X.dmult;
                if (c1 < 3) gc[c1] += giy * Y.dmult;
            }
        }
    }
    if (dL_dxyz) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int d = 1; d < C; d <<= 1) gc[k] += __shfl_xor(gc[k], d, 64);
        }
        if (ch == 0) {
            dL_dxyz[3 * i + 0] = gc[0] * p.dscale[0];
            dL_dxyz[3 * i + 1] = gc[1] * p.dscale[1];
            dL_dxyz[3 * i + 2] = gc[2] * p.dscale[2];
        }
    }
}

}  // namespace gsr
