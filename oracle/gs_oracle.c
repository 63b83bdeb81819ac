/*
 * oracle/gs_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sequential CPU restatement of the reference's differentiable Gaussian
 * rasterizer (submodules/diff-gaussian-rasterization, abbreviated DGR/ below).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared library; the product path (4dgs-slam_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned by reference outputs". The reference path is
 * CUDA-only (needs nvcc, cuda_runtime.h, cub, cooperative_groups) and cannot be
 * compiled or run in this image, and the reference ships no tests / golden
 * vectors for it. This restatement is pinned instead by (tests/test_oracle_*.py):
 * analytic known-answer cases, an independent fp64 PyTorch-autograd forward,
 * finite differences through SE3_exp for the pose gradient, and golden vectors
 * generated from the importable Python pieces of the reference
 * (tests/golden/make_golden.py).
 *
 * The file is compiled twice (REAL=float -> *_f32, REAL=double -> *_f64) so the
 * fp32 result can be compared with a higher-precision evaluation of the same
 * algorithm. With REAL=float and -ffp-contract=off every operation is the fp32
 * operation the CUDA source spells out (nvcc may contract a*b+c into FMA; that
 * difference is inside the tolerances of BASELINE.md).
 *
 * Each function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
#ifndef SUFFIX
#define SUFFIX f32
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

#define BLOCK_X 16 /* DGR/cuda_rasterizer/config.h:15-17 */
#define BLOCK_Y 16
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)
#define NUM_CHANNELS 3

typedef REAL real;

/* DGR/cuda_rasterizer/auxiliary.h:22-39 */
/* GSO_OMP (oracle/Makefile target libgs_oracle_omp.so, bench.py's all-core cpu_baseline only): the tile loops and the per-Gaussian
 * loops run under OpenMP, per-Gaussian accumulations become atomic adds. The summation order is then not fixed, so the CHECKER is
 * always the serial build. */
#ifdef GSO_OMP
#define GSO_FOR_TILES _Pragma("omp parallel for collapse(2) schedule(dynamic, 2)")
#define GSO_FOR_POINTS _Pragma("omp parallel for schedule(static)")
#define GSO_ATOMIC _Pragma("omp atomic")
#else
#define GSO_FOR_TILES
#define GSO_FOR_POINTS
#define GSO_ATOMIC
#endif

static const real SH_C0 = (real)0.28209479177387814;
static const real SH_C1 = (real)0.4886025119029199;
static const real SH_C2[5] = {(real)1.0925484305920792, (real)-1.0925484305920792, (real)0.31539156525252005,
                              (real)-1.0925484305920792, (real)0.5462742152960396};
static const real SH_C3[7] = {(real)-0.5900435899266435, (real)2.890611442640554, (real)-0.4570457994644658,
                              (real)0.3731763325901154, (real)-0.4570457994644658, (real)1.445305721320277,
                              (real)-0.5900435899266435};

static inline real rmin(real a, real b) { return a < b ? a : b; }
static inline real rmax(real a, real b) { return a > b ? a : b; }
static inline real rsqrt_(real x) { return (real)sqrt((double)x); }
static inline real rexp_(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

typedef struct { real x, y, z; } v3;
typedef struct { v3 c[3]; } m33; /* column storage, like DGR/cuda_rasterizer/math.h:4-19 */

static inline v3 V3(real x, real y, real z) { v3 r = {x, y, z}; return r; }
static inline real dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* math.h:27-31 */
static inline m33 skew(v3 v) { m33 m; m.c[0] = V3(0, v.z, -v.y); m.c[1] = V3(-v.z, 0, v.x); m.c[2] = V3(v.y, -v.x, 0); return m; }
/* math.h:90-95 */
static inline m33 mneg(m33 a) { for (int i = 0; i < 3; i++) a.c[i] = V3(-a.c[i].x, -a.c[i].y, -a.c[i].z); return a; }
/* math.h:33-40 */
static inline m33 mtrans(m33 a) { m33 r; r.c[0] = V3(a.c[0].x, a.c[1].x, a.c[2].x); r.c[1] = V3(a.c[0].y, a.c[1].y, a.c[2].y); r.c[2] = V3(a.c[0].z, a.c[1].z, a.c[2].z); return r; }
/* math.h:81-88 */
static inline v3 mvec(m33 a, v3 v) { return V3(a.c[0].x * v.x + a.c[1].x * v.y + a.c[2].x * v.z, a.c[0].y * v.x + a.c[1].y * v.y + a.c[2].y * v.z, a.c[0].z * v.x + a.c[1].z * v.y + a.c[2].z * v.z); }
static inline m33 mident(void) { m33 m; m.c[0] = V3(1, 0, 0); m.c[1] = V3(0, 1, 0); m.c[2] = V3(0, 0, 1); return m; }

/* auxiliary.h:41-44 -- note the double-precision literals: evaluated in double, rounded once. */
static inline real ndc2Pix(real v, int S) { return (real)((((double)v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:46-56 */
static inline void getRect(real px, real py, int max_radius, int gx, int gy, int* rminx, int* rminy, int* rmaxx, int* rmaxy)
{
    *rminx = imin(gx, imax(0, (int)((px - max_radius) / BLOCK_X)));
    *rminy = imin(gy, imax(0, (int)((py - max_radius) / BLOCK_Y)));
    *rmaxx = imin(gx, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    *rmaxy = imin(gy, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}
/* auxiliary.h:58-66 */
static inline v3 transformPoint4x3(v3 p, const real* m) { return V3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13], m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]); }
/* auxiliary.h:89-97 */
static inline v3 transformVec4x3Transpose(v3 p, const real* m) { return V3(m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z, m[8] * p.x + m[9] * p.y + m[10] * p.z); }
/* auxiliary.h:107-117 */
static inline v3 dnormvdv(v3 v, v3 dv)
{
    real sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    real invsum32 = (real)1 / rsqrt_(sum2 * sum2 * sum2);
    v3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

/* ------------------------------------------------------------------------- */
/* State kept between forward and backward: the reference's GeometryState /
 * ImageState / BinningState (DGR/cuda_rasterizer/rasterizer_impl.h:29-65). */
typedef struct FN(gso_ctx) {
    int P, D, M, W, H, gx, gy, R;
    /* GeometryState */
    real* depths; uint8_t* clamped; real* means2D; real* cov3D; real* conic_opacity; real* rgb;
    uint32_t* tiles_touched; uint32_t* point_offsets; int* radii;
    /* ImageState */
    real* final_T; uint32_t* n_contrib; uint32_t* ranges;
    /* BinningState */
    uint64_t* keys; uint32_t* point_list;
} FN(gso_ctx);

void FN(gso_free)(FN(gso_ctx)* c)
{
    if (!c) return;
    free(c->depths); free(c->clamped); free(c->means2D); free(c->cov3D); free(c->conic_opacity); free(c->rgb);
    free(c->tiles_touched); free(c->point_offsets); free(c->radii); free(c->final_T); free(c->n_contrib);
    free(c->ranges); free(c->keys); free(c->point_list); free(c);
}

/* DGR/cuda_rasterizer/forward.cu:22-73 */
static void sh_to_rgb(int idx, int deg, int max_coeffs, const real* means, const real* campos, const real* shs, uint8_t* clamped, real* out)
{
    v3 pos = V3(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
    v3 dir = V3(pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]);
    real len = rsqrt_(dot3(dir, dir));
    dir = V3(dir.x / len, dir.y / len, dir.z / len);
    const real* sh = shs + (size_t)idx * max_coeffs * 3;
    real res[3];
    for (int k = 0; k < 3; k++) {
        real result = SH_C0 * sh[0 * 3 + k];
        if (deg > 0) {
            real x = dir.x, y = dir.y, z = dir.z;
            result = result - SH_C1 * y * sh[1 * 3 + k] + SH_C1 * z * sh[2 * 3 + k] - SH_C1 * x * sh[3 * 3 + k];
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                result = result + SH_C2[0] * xy * sh[4 * 3 + k] + SH_C2[1] * yz * sh[5 * 3 + k] +
                         SH_C2[2] * ((real)2 * zz - xx - yy) * sh[6 * 3 + k] + SH_C2[3] * xz * sh[7 * 3 + k] +
                         SH_C2[4] * (xx - yy) * sh[8 * 3 + k];
                if (deg > 2) {
                    result = result + SH_C3[0] * y * ((real)3 * xx - yy) * sh[9 * 3 + k] + SH_C3[1] * xy * z * sh[10 * 3 + k] +
                             SH_C3[2] * y * ((real)4 * zz - xx - yy) * sh[11 * 3 + k] +
                             SH_C3[3] * z * ((real)2 * zz - (real)3 * xx - (real)3 * yy) * sh[12 * 3 + k] +
                             SH_C3[4] * x * ((real)4 * zz - xx - yy) * sh[13 * 3 + k] + SH_C3[5] * z * (xx - yy) * sh[14 * 3 + k] +
                             SH_C3[6] * x * (xx - (real)3 * yy) * sh[15 * 3 + k];
                }
            }
        }
        result += (real)0.5;
        clamped[3 * idx + k] = (result < 0);
        res[k] = rmax(result, 0);
    }
    out[0] = res[0]; out[1] = res[1]; out[2] = res[2];
}

/* DGR/cuda_rasterizer/forward.cu:120-154. Sigma = Rq diag(s*mod)^2 Rq^T, quaternion NOT normalised (:129). */
static void computeCov3D(const real* scale, real mod, const real* rot, real* cov3D)
{
    real s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    /* Rq[i][j], standard rotation matrix from (r,x,y,z) */
    real Rq[3][3] = {
        {(real)1 - (real)2 * (y * y + z * z), (real)2 * (x * y - r * z), (real)2 * (x * z + r * y)},
        {(real)2 * (x * y + r * z), (real)1 - (real)2 * (x * x + z * z), (real)2 * (y * z - r * x)},
        {(real)2 * (x * z - r * y), (real)2 * (y * z + r * x), (real)1 - (real)2 * (x * x + y * y)}};
    /* M = S * Rq^T (glm: M = S * R with R stored transposed); Sigma = M^T M */
    real Mm[3][3];
    for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) Mm[k][j] = s[k] * Rq[j][k];
    real Sg[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Sg[i][j] = Mm[0][i] * Mm[0][j] + Mm[1][i] * Mm[1][j] + Mm[2][i] * Mm[2][j];
    cov3D[0] = Sg[0][0]; cov3D[1] = Sg[0][1]; cov3D[2] = Sg[0][2]; cov3D[3] = Sg[1][1]; cov3D[4] = Sg[1][2]; cov3D[5] = Sg[2][2];
}

/* Shared by forward (forward.cu:76-115) and backward (backward.cu:171-206): A = J * R_cw (2x3 used), cov2D = A Sigma A^T + 0.3 I */
static void cov2d_parts(v3 mean, real fx, real fy, real tan_fovx, real tan_fovy, const real* cov3D, const real* vm,
                        v3* t_out, real* txtz_o, real* tytz_o, real A[2][3], real Jm[2][3], real* a, real* b, real* c)
{
    v3 t = transformPoint4x3(mean, vm);
    const real limx = (real)1.3 * tan_fovx, limy = (real)1.3 * tan_fovy;
    const real txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = rmin(limx, rmax(-limx, txtz)) * t.z;
    t.y = rmin(limy, rmax(-limy, tytz)) * t.z;
    Jm[0][0] = fx / t.z; Jm[0][1] = 0; Jm[0][2] = -(fx * t.x) / (t.z * t.z);
    Jm[1][0] = 0; Jm[1][1] = fy / t.z; Jm[1][2] = -(fy * t.y) / (t.z * t.z);
    /* R_cw[i][k] = vm[i + 4k] */
    for (int i = 0; i < 2; i++) for (int k = 0; k < 3; k++) A[i][k] = Jm[i][0] * vm[0 + 4 * k] + Jm[i][1] * vm[1 + 4 * k] + Jm[i][2] * vm[2 + 4 * k];
    real V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
    real AV[2][3];
    for (int i = 0; i < 2; i++) for (int k = 0; k < 3; k++) AV[i][k] = A[i][0] * V[0][k] + A[i][1] * V[1][k] + A[i][2] * V[2][k];
    *a = AV[0][0] * A[0][0] + AV[0][1] * A[0][1] + AV[0][2] * A[0][2] + (real)0.3;
    *b = AV[0][0] * A[1][0] + AV[0][1] * A[1][1] + AV[0][2] * A[1][2];
    *c = AV[1][0] * A[1][0] + AV[1][1] * A[1][1] + AV[1][2] * A[1][2] + (real)0.3;
    *t_out = t; *txtz_o = txtz; *tytz_o = tytz;
}

/* rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

typedef struct { uint64_t key; uint32_t val; uint32_t seq; } kv_t;
static int kv_cmp(const void* a, const void* b)
{
    const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq); /* stable, like cub radix sort (rasterizer_impl.cu:306-311) */
}

/* rasterizer_impl.cu:54-66,141-153 + auxiliary.h:139-164 */
void FN(gso_mark_visible)(int P, const real* means3D, const real* viewmatrix, const real* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    for (int idx = 0; idx < P; idx++) {
        v3 p = V3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        v3 pv = transformPoint4x3(p, viewmatrix);
        present[idx] = !(pv.z <= (real)0.2);
    }
}

/*
 * Forward: DGR/cuda_rasterizer/rasterizer_impl.cu:198-344 (K1 preprocess forward.cu:157-258, K2 scan :280,
 * K3 duplicateWithKeys :70-111, K4 sort :306-311, K5 identifyTileRanges :116-138, K6 render forward.cu:263-392).
 * Output buffers are zero-filled by the caller as rasterize_points.cu:69-73 does.
 * Returns NULL if prefiltered is set and a point is culled (device __trap in the reference, auxiliary.h:156-160).
 */
FN(gso_ctx)* FN(gso_forward)(int P, int D, int M, const real* background, int W, int H,
    const real* means3D, const real* shs, const real* colors_precomp, const real* opacities,
    const real* scales, real scale_modifier, const real* rotations, const real* cov3D_precomp,
    const real* viewmatrix, const real* projmatrix, const real* cam_pos, real tan_fovx, real tan_fovy, int prefiltered,
    real* out_color, real* out_depth, real* out_opacity, int* radii_out, int* n_touched)
{
    FN(gso_ctx)* c = (FN(gso_ctx)*)calloc(1, sizeof(*c));
    const real focal_y = H / ((real)2 * tan_fovy), focal_x = W / ((real)2 * tan_fovx); /* rasterizer_impl.cu:225-226 */
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const int N = W * H;
    c->P = P; c->D = D; c->M = M; c->W = W; c->H = H; c->gx = gx; c->gy = gy;
    size_t Pa = P > 0 ? P : 1;
    c->depths = (real*)calloc(Pa, sizeof(real)); c->clamped = (uint8_t*)calloc(Pa * 3, 1);
    c->means2D = (real*)calloc(Pa * 2, sizeof(real)); c->cov3D = (real*)calloc(Pa * 6, sizeof(real));
    c->conic_opacity = (real*)calloc(Pa * 4, sizeof(real)); c->rgb = (real*)calloc(Pa * 3, sizeof(real));
    c->tiles_touched = (uint32_t*)calloc(Pa, 4); c->point_offsets = (uint32_t*)calloc(Pa, 4); c->radii = (int*)calloc(Pa, 4);
    c->final_T = (real*)calloc(N, sizeof(real)); c->n_contrib = (uint32_t*)calloc(N, 4);
    c->ranges = (uint32_t*)calloc((size_t)gx * gy * 2, 4);

    /* ---- K1: preprocessCUDA, forward.cu:184-257 ---- (serial in every build: it can return early) */
    for (int idx = 0; idx < P; idx++) {
        c->radii[idx] = 0; c->tiles_touched[idx] = 0;
        v3 p_orig = V3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        /* in_frustum, auxiliary.h:139-164 */
        v3 p_view = transformPoint4x3(p_orig, viewmatrix);
        if (p_view.z <= (real)0.2) {
            if (prefiltered) { FN(gso_free)(c); return NULL; }
            continue;
        }
        const real* pm = projmatrix;
        real hx = pm[0] * p_orig.x + pm[4] * p_orig.y + pm[8] * p_orig.z + pm[12];
        real hy = pm[1] * p_orig.x + pm[5] * p_orig.y + pm[9] * p_orig.z + pm[13];
        real hw = pm[3] * p_orig.x + pm[7] * p_orig.y + pm[11] * p_orig.z + pm[15];
        real p_w = (real)1 / (hw + (real)0.0000001);
        real projx = hx * p_w, projy = hy * p_w;

        const real* cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + (size_t)idx * 6;
        else { computeCov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, c->cov3D + (size_t)idx * 6); cov3D = c->cov3D + (size_t)idx * 6; }

        v3 t; real txtz, tytz, A[2][3], Jm[2][3], ca, cb, cc;
        cov2d_parts(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &t, &txtz, &tytz, A, Jm, &ca, &cb, &cc);
        real det = ca * cc - cb * cb; /* forward.cu:221-225 */
        if (det == 0) continue;
        real det_inv = (real)1 / det;
        real conic[3] = {cc * det_inv, -cb * det_inv, ca * det_inv};
        real mid = (real)0.5 * (ca + cc); /* forward.cu:231-234 */
        real lambda1 = mid + rsqrt_(rmax((real)0.1, mid * mid - det));
        real lambda2 = mid - rsqrt_(rmax((real)0.1, mid * mid - det));
        real my_radius = (real)ceil((double)((real)3 * rsqrt_(rmax(lambda1, lambda2))));
        real px = ndc2Pix(projx, W), py = ndc2Pix(projy, H);
        int rminx, rminy, rmaxx, rmaxy;
        getRect(px, py, (int)my_radius, gx, gy, &rminx, &rminy, &rmaxx, &rmaxy);
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;
        if (!colors_precomp) sh_to_rgb(idx, D, M, means3D, cam_pos, shs, c->clamped, c->rgb + (size_t)idx * 3);
        c->depths[idx] = p_view.z;
        c->radii[idx] = (int)my_radius;
        c->means2D[2 * idx] = px; c->means2D[2 * idx + 1] = py;
        c->conic_opacity[4 * idx + 0] = conic[0]; c->conic_opacity[4 * idx + 1] = conic[1];
        c->conic_opacity[4 * idx + 2] = conic[2]; c->conic_opacity[4 * idx + 3] = opacities[idx];
        c->tiles_touched[idx] = (uint32_t)((rmaxy - rminy) * (rmaxx - rminx));
    }
    if (radii_out) memcpy(radii_out, c->radii, sizeof(int) * (size_t)P);

    /* ---- K2: inclusive scan, rasterizer_impl.cu:280-284 ---- */
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += c->tiles_touched[i]; c->point_offsets[i] = run; }
    const int R = (int)run; c->R = R;

    /* ---- K3: duplicateWithKeys, rasterizer_impl.cu:70-111. Keys use the fp32 bit pattern of the depth. ---- */
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(R > 0 ? R : 1));
    for (int idx = 0; idx < P; idx++) {
        if (c->radii[idx] > 0) {
            uint32_t off = idx == 0 ? 0 : c->point_offsets[idx - 1];
            int rminx, rminy, rmaxx, rmaxy;
            getRect(c->means2D[2 * idx], c->means2D[2 * idx + 1], c->radii[idx], gx, gy, &rminx, &rminy, &rmaxx, &rmaxy);
            float df = (float)c->depths[idx]; uint32_t dbits; memcpy(&dbits, &df, 4);
            for (int y = rminy; y < rmaxy; y++) for (int x = rminx; x < rmaxx; x++) {
                uint64_t key = (uint64_t)(y * gx + x); key <<= 32; key |= dbits;
                kv[off].key = key; kv[off].val = (uint32_t)idx; kv[off].seq = off; off++;
            }
        }
    }
    /* ---- K4: stable sort on bits [0, 32+getHigherMsb(T)); tile ids < 2^bit so this is a full-key stable sort ---- */
    (void)getHigherMsb;
    qsort(kv, (size_t)R, sizeof(kv_t), kv_cmp);
    c->keys = (uint64_t*)malloc(8 * (size_t)(R > 0 ? R : 1)); c->point_list = (uint32_t*)malloc(4 * (size_t)(R > 0 ? R : 1));
    for (int i = 0; i < R; i++) { c->keys[i] = kv[i].key; c->point_list[i] = kv[i].val; }
    free(kv);
    /* ---- K5: identifyTileRanges, rasterizer_impl.cu:116-138 (ranges zeroed, :313) ---- */
    for (int i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(c->keys[i] >> 32);
        if (i == 0) c->ranges[2 * cur] = 0;
        else { uint32_t prev = (uint32_t)(c->keys[i - 1] >> 32); if (cur != prev) { c->ranges[2 * prev + 1] = (uint32_t)i; c->ranges[2 * cur] = (uint32_t)i; } }
        if (i == R - 1) c->ranges[2 * cur + 1] = (uint32_t)R;
    }

    /* ---- K6: renderCUDA, forward.cu:263-392 (per pixel; the block-level early-out :318-320 does not change results) ---- */
    const real* features = colors_precomp ? colors_precomp : c->rgb; /* rasterizer_impl.cu:324 */
    GSO_FOR_TILES
    for (int ty = 0; ty < gy; ty++) for (int tx = 0; tx < gx; tx++) {
        uint32_t r0 = c->ranges[2 * (ty * gx + tx)], r1 = c->ranges[2 * (ty * gx + tx) + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
            int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
            if (!(pxi < W && pyi < H)) continue;
            int pix_id = W * pyi + pxi;
            real pixfx = (real)pxi, pixfy = (real)pyi;
            real T = 1; uint32_t contributor = 0, last_contributor = 0; real C[3] = {0, 0, 0}; real Dd = 0;
            for (uint32_t k = r0; k < r1; k++) {
                contributor++;
                uint32_t g = c->point_list[k];
                real dx = c->means2D[2 * g] - pixfx, dy = c->means2D[2 * g + 1] - pixfy;
                const real* co = c->conic_opacity + 4 * (size_t)g;
                real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0) continue;
                real alpha = rmin((real)0.99, co[3] * rexp_(power));
                if (alpha < (real)1 / (real)255) continue;
                real test_T = T * (1 - alpha);
                if (test_T < (real)0.0001) break; /* done = true */
                for (int ch = 0; ch < 3; ch++) C[ch] += features[g * 3 + ch] * alpha * T;
                Dd += c->depths[g] * alpha * T;
                if (test_T > (real)0.5) { GSO_ATOMIC n_touched[g] += 1; }
                T = test_T;
                last_contributor = contributor;
            }
            c->final_T[pix_id] = T; c->n_contrib[pix_id] = last_contributor;
            for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * background[ch];
            out_depth[pix_id] = Dd; out_opacity[pix_id] = 1 - T;
        }
    }
    return c;
}

int FN(gso_num_rendered)(const FN(gso_ctx)* c) { return c->R; }

/* Copies of the intermediate state, so tests can localise a mismatch to a stage. Any pointer may be NULL. */
void FN(gso_get_state)(const FN(gso_ctx)* c, real* depths, uint8_t* clamped, real* means2D, real* cov3D, real* conic_opacity,
                       real* rgb, uint32_t* tiles_touched, uint32_t* point_offsets, real* final_T, uint32_t* n_contrib,
                       uint32_t* ranges, uint64_t* keys, uint32_t* point_list)
{
    size_t P = (size_t)c->P, N = (size_t)c->W * c->H, T = (size_t)c->gx * c->gy, R = (size_t)c->R;
    if (depths) memcpy(depths, c->depths, P * sizeof(real));
    if (clamped) memcpy(clamped, c->clamped, P * 3);
    if (means2D) memcpy(means2D, c->means2D, P * 2 * sizeof(real));
    if (cov3D) memcpy(cov3D, c->cov3D, P * 6 * sizeof(real));
    if (conic_opacity) memcpy(conic_opacity, c->conic_opacity, P * 4 * sizeof(real));
    if (rgb) memcpy(rgb, c->rgb, P * 3 * sizeof(real));
    if (tiles_touched) memcpy(tiles_touched, c->tiles_touched, P * 4);
    if (point_offsets) memcpy(point_offsets, c->point_offsets, P * 4);
    if (final_T) memcpy(final_T, c->final_T, N * sizeof(real));
    if (n_contrib) memcpy(n_contrib, c->n_contrib, N * 4);
    if (ranges) memcpy(ranges, c->ranges, T * 8);
    if (keys) memcpy(keys, c->keys, R * 8);
    if (point_list) memcpy(point_list, c->point_list, R * 4);
}

/* DGR/cuda_rasterizer/backward.cu:21-145 */
static void sh_backward(int idx, int deg, int max_coeffs, const real* means, const real* campos, const real* shs,
                        const uint8_t* clamped, const real* dL_dcolor, real* dL_dmeans, real* dL_dshs, real* dL_dtau)
{
    v3 pos = V3(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
    v3 dir_orig = V3(pos.x - campos[0], pos.y - campos[1], pos.z - campos[2]);
    real len = rsqrt_(dot3(dir_orig, dir_orig));
    v3 dir = V3(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
    const real* sh = shs + (size_t)idx * max_coeffs * 3;
    real dL_dRGB[3];
    for (int k = 0; k < 3; k++) dL_dRGB[k] = dL_dcolor[3 * idx + k] * (clamped[3 * idx + k] ? (real)0 : (real)1);
    real dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
    real x = dir.x, y = dir.y, z = dir.z;
    real* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
#define SHK(i, k) sh[(i) * 3 + (k)]
#define SETSH(i, coef) for (int k = 0; k < 3; k++) dL_dsh[(i) * 3 + k] = (coef) * dL_dRGB[k]
    SETSH(0, SH_C0);
    if (deg > 0) {
        real d1 = -SH_C1 * y, d2 = SH_C1 * z, d3 = -SH_C1 * x;
        SETSH(1, d1); SETSH(2, d2); SETSH(3, d3);
        for (int k = 0; k < 3; k++) { dRGBdx[k] = -SH_C1 * SHK(3, k); dRGBdy[k] = -SH_C1 * SHK(1, k); dRGBdz[k] = SH_C1 * SHK(2, k); }
        if (deg > 1) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            real d4 = SH_C2[0] * xy, d5 = SH_C2[1] * yz, d6 = SH_C2[2] * ((real)2 * zz - xx - yy), d7 = SH_C2[3] * xz, d8 = SH_C2[4] * (xx - yy);
            SETSH(4, d4); SETSH(5, d5); SETSH(6, d6); SETSH(7, d7); SETSH(8, d8);
            for (int k = 0; k < 3; k++) {
                dRGBdx[k] += SH_C2[0] * y * SHK(4, k) + SH_C2[2] * (real)2 * -x * SHK(6, k) + SH_C2[3] * z * SHK(7, k) + SH_C2[4] * (real)2 * x * SHK(8, k);
                dRGBdy[k] += SH_C2[0] * x * SHK(4, k) + SH_C2[1] * z * SHK(5, k) + SH_C2[2] * (real)2 * -y * SHK(6, k) + SH_C2[4] * (real)2 * -y * SHK(8, k);
                dRGBdz[k] += SH_C2[1] * y * SHK(5, k) + SH_C2[2] * (real)2 * (real)2 * z * SHK(6, k) + SH_C2[3] * x * SHK(7, k);
            }
            if (deg > 2) {
                real d9 = SH_C3[0] * y * ((real)3 * xx - yy), d10 = SH_C3[1] * xy * z, d11 = SH_C3[2] * y * ((real)4 * zz - xx - yy);
                real d12 = SH_C3[3] * z * ((real)2 * zz - (real)3 * xx - (real)3 * yy), d13 = SH_C3[4] * x * ((real)4 * zz - xx - yy);
                real d14 = SH_C3[5] * z * (xx - yy), d15 = SH_C3[6] * x * (xx - (real)3 * yy);
                SETSH(9, d9); SETSH(10, d10); SETSH(11, d11); SETSH(12, d12); SETSH(13, d13); SETSH(14, d14); SETSH(15, d15);
                for (int k = 0; k < 3; k++) {
                    dRGBdx[k] += (SH_C3[0] * SHK(9, k) * (real)3 * (real)2 * xy + SH_C3[1] * SHK(10, k) * yz + SH_C3[2] * SHK(11, k) * (real)-2 * xy +
                                  SH_C3[3] * SHK(12, k) * (real)-3 * (real)2 * xz + SH_C3[4] * SHK(13, k) * ((real)-3 * xx + (real)4 * zz - yy) +
                                  SH_C3[5] * SHK(14, k) * (real)2 * xz + SH_C3[6] * SHK(15, k) * (real)3 * (xx - yy));
                    dRGBdy[k] += (SH_C3[0] * SHK(9, k) * (real)3 * (xx - yy) + SH_C3[1] * SHK(10, k) * xz + SH_C3[2] * SHK(11, k) * ((real)-3 * yy + (real)4 * zz - xx) +
                                  SH_C3[3] * SHK(12, k) * (real)-3 * (real)2 * yz + SH_C3[4] * SHK(13, k) * (real)-2 * xy +
                                  SH_C3[5] * SHK(14, k) * (real)-2 * yz + SH_C3[6] * SHK(15, k) * (real)-3 * (real)2 * xy);
                    dRGBdz[k] += (SH_C3[1] * SHK(10, k) * xy + SH_C3[2] * SHK(11, k) * (real)4 * (real)2 * yz + SH_C3[3] * SHK(12, k) * (real)3 * ((real)2 * zz - xx - yy) +
                                  SH_C3[4] * SHK(13, k) * (real)4 * (real)2 * xz + SH_C3[5] * SHK(14, k) * (xx - yy));
                }
            }
        }
    }
#undef SHK
#undef SETSH
    v3 dL_ddir = V3(dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
                    dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
                    dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]);
    v3 dL_dmean = dnormvdv(dir_orig, dL_ddir);
    dL_dmeans[3 * idx + 0] += dL_dmean.x; dL_dmeans[3 * idx + 1] += dL_dmean.y; dL_dmeans[3 * idx + 2] += dL_dmean.z;
    dL_dtau[6 * idx + 0] += -dL_dmean.x; dL_dtau[6 * idx + 1] += -dL_dmean.y; dL_dtau[6 * idx + 2] += -dL_dmean.z; /* backward.cu:141-143 */
}

/* DGR/cuda_rasterizer/backward.cu:350-413 */
static void cov3d_backward(int idx, const real* scale, real mod, const real* rot, const real* dL_dcov3Ds, real* dL_dscales, real* dL_drots)
{
    real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    real Rq[3][3] = {
        {(real)1 - (real)2 * (y * y + z * z), (real)2 * (x * y - r * z), (real)2 * (x * z + r * y)},
        {(real)2 * (x * y + r * z), (real)1 - (real)2 * (x * x + z * z), (real)2 * (y * z - r * x)},
        {(real)2 * (x * z - r * y), (real)2 * (y * z + r * x), (real)1 - (real)2 * (x * x + y * y)}};
    real s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    /* M = S Rq^T (math); dL_dSigma symmetric with halved off-diagonals (:380-384) */
    real Mm[3][3];
    for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) Mm[k][j] = s[k] * Rq[j][k];
    const real* g = dL_dcov3Ds + 6 * (size_t)idx;
    real dS[3][3] = {{g[0], (real)0.5 * g[1], (real)0.5 * g[2]}, {(real)0.5 * g[1], g[3], (real)0.5 * g[4]}, {(real)0.5 * g[2], (real)0.5 * g[4], g[5]}};
    /* glm: dL_dM = 2.0f * M * dL_dSigma (:388): math dL_dM = 2 M dSigma */
    real dM[3][3];
    for (int k = 0; k < 3; k++) for (int j = 0; j < 3; j++) dM[k][j] = (real)2 * (Mm[k][0] * dS[0][j] + Mm[k][1] * dS[1][j] + Mm[k][2] * dS[2][j]);
    /* glm Rt[k] = column k of transpose(R_glm) = (Rq[0][k], Rq[1][k], Rq[2][k]); dL_dMt[k] = column k of transpose(dL_dM_glm) = math row k of dL_dM */
    for (int k = 0; k < 3; k++) dL_dscales[3 * idx + k] = Rq[0][k] * dM[k][0] + Rq[1][k] * dM[k][1] + Rq[2][k] * dM[k][2]; /* :394-397 */
    /* dL_dMt[k] *= s[k] (:399-401); glm dL_dMt[a][b] = math dM[a][b] * s[a] */
    real D_[3][3];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) D_[a][b] = dM[a][b] * s[a];
    real q0 = (real)2 * z * (D_[0][1] - D_[1][0]) + (real)2 * y * (D_[2][0] - D_[0][2]) + (real)2 * x * (D_[1][2] - D_[2][1]);
    real q1 = (real)2 * y * (D_[1][0] + D_[0][1]) + (real)2 * z * (D_[2][0] + D_[0][2]) + (real)2 * r * (D_[1][2] - D_[2][1]) - (real)4 * x * (D_[2][2] + D_[1][1]);
    real q2 = (real)2 * x * (D_[1][0] + D_[0][1]) + (real)2 * r * (D_[2][0] - D_[0][2]) + (real)2 * z * (D_[1][2] + D_[2][1]) - (real)4 * y * (D_[2][2] + D_[0][0]);
    real q3 = (real)2 * r * (D_[0][1] - D_[1][0]) + (real)2 * x * (D_[2][0] + D_[0][2]) + (real)2 * y * (D_[1][2] + D_[2][1]) - (real)4 * z * (D_[1][1] + D_[0][0]);
    dL_drots[4 * idx + 0] = q0; dL_drots[4 * idx + 1] = q1; dL_drots[4 * idx + 2] = q2; dL_drots[4 * idx + 3] = q3; /* :411-412, no normalisation backward */
}

/*
 * Backward: DGR/cuda_rasterizer/rasterizer_impl.cu:348-455 (K7 backward.cu:563-787, K8 :150-346, K9 :418-539).
 * All gradient outputs must be zero-filled by the caller (rasterize_points.cu:160-170).
 * dL_dmean2D is [P,3] (z never written), dL_dconic is [P,4] (.z never written; backward.cu:754-756 reduces an
 * uninitialised value that nothing reads, so the oracle leaves it 0).
 */
void FN(gso_backward)(const FN(gso_ctx)* c, const real* background, const real* means3D, const real* shs, const real* colors_precomp,
    const real* scales, real scale_modifier, const real* rotations, const real* cov3D_precomp,
    const real* viewmatrix, const real* projmatrix, const real* projmatrix_raw, const real* campos, real tan_fovx, real tan_fovy,
    const real* dL_dpix, const real* dL_dpix_depth,
    real* dL_dmean2D, real* dL_dconic, real* dL_dopacity, real* dL_dcolor, real* dL_ddepth,
    real* dL_dmean3D, real* dL_dcov3D, real* dL_dsh, real* dL_dscale, real* dL_drot, real* dL_dtau)
{
    const int P = c->P, D = c->D, M = c->M, W = c->W, H = c->H, gx = c->gx, gy = c->gy;
    const real focal_y = H / ((real)2 * tan_fovy), focal_x = W / ((real)2 * tan_fovx);
    const real* colors = colors_precomp ? colors_precomp : c->rgb; /* rasterizer_impl.cu:401 */

    /* ---- K7: renderCUDA backward, backward.cu:617-786 ---- */
    const real ddelx_dx = (real)0.5 * W, ddely_dy = (real)0.5 * H; /* :643-644 */
    GSO_FOR_TILES
    for (int ty = 0; ty < gy; ty++) for (int tx = 0; tx < gx; tx++) {
        uint32_t r0 = c->ranges[2 * (ty * gx + tx)], r1 = c->ranges[2 * (ty * gx + tx) + 1];
        if (r1 <= r0) continue;
        /* per-pixel running state of the 256 threads of the block */
        real T[BLOCK_SIZE], Tfin[BLOCK_SIZE], acc[BLOCK_SIZE][3], accd[BLOCK_SIZE], lastc[BLOCK_SIZE][3], lastd[BLOCK_SIZE], lasta[BLOCK_SIZE];
        real dpx[BLOCK_SIZE][3], dpd[BLOCK_SIZE];
        uint32_t contributor[BLOCK_SIZE]; int lastcontrib[BLOCK_SIZE]; int inside[BLOCK_SIZE];
        for (int t = 0; t < BLOCK_SIZE; t++) {
            int pxi = tx * BLOCK_X + (t % BLOCK_X), pyi = ty * BLOCK_Y + (t / BLOCK_X);
            inside[t] = pxi < W && pyi < H;
            int pix_id = W * pyi + pxi;
            Tfin[t] = inside[t] ? c->final_T[pix_id] : 0; T[t] = Tfin[t];
            contributor[t] = r1 - r0; lastcontrib[t] = inside[t] ? (int)c->n_contrib[pix_id] : 0;
            for (int ch = 0; ch < 3; ch++) { acc[t][ch] = 0; lastc[t][ch] = 0; dpx[t][ch] = inside[t] ? dL_dpix[(size_t)ch * H * W + pix_id] : 0; }
            accd[t] = 0; lastd[t] = 0; lasta[t] = 0; dpd[t] = inside[t] ? dL_dpix_depth[pix_id] : 0;
        }
        for (uint32_t k = r1; k-- > r0;) { /* back to front (:656) */
            const uint32_t g = c->point_list[k];
            const real* co = c->conic_opacity + 4 * (size_t)g;
            const real gxm = c->means2D[2 * g], gym = c->means2D[2 * g + 1], gdepth = c->depths[g];
            double s_m2x = 0, s_m2y = 0, s_cx = 0, s_cy = 0, s_cw = 0, s_op = 0, s_col[3] = {0, 0, 0}, s_dep = 0;
            for (int t = 0; t < BLOCK_SIZE; t++) {
                int done = !inside[t];
                int skip = done;
                contributor[t] = done ? contributor[t] : contributor[t] - 1; /* :677 */
                skip |= (contributor[t] >= (uint32_t)lastcontrib[t]);
                real pixfx = (real)(tx * BLOCK_X + (t % BLOCK_X)), pixfy = (real)(ty * BLOCK_Y + (t / BLOCK_X));
                real dx = gxm - pixfx, dy = gym - pixfy;
                real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                skip |= power > 0;
                real G = rexp_(power);
                real alpha = rmin((real)0.99, co[3] * G);
                skip |= alpha < (real)1 / (real)255;
                if (skip) continue; /* every contribution below is gated by !skip in the reference (:700-757) */
                T[t] = T[t] / ((real)1 - alpha);
                real dchannel_dcolor = alpha * T[t];
                real dL_dalpha = 0;
                for (int ch = 0; ch < 3; ch++) {
                    real cc_ = colors[g * 3 + ch];
                    acc[t][ch] = lasta[t] * lastc[t][ch] + ((real)1 - lasta[t]) * acc[t][ch];
                    lastc[t][ch] = cc_;
                    dL_dalpha += (cc_ - acc[t][ch]) * dpx[t][ch];
                    s_col[ch] += dchannel_dcolor * dpx[t][ch];
                }
                accd[t] = lasta[t] * lastd[t] + ((real)1 - lasta[t]) * accd[t];
                lastd[t] = gdepth;
                dL_dalpha += (gdepth - accd[t]) * dpd[t];
                s_dep += dchannel_dcolor * dpd[t];
                dL_dalpha *= T[t];
                lasta[t] = alpha;
                real bg_dot = 0;
                for (int ch = 0; ch < 3; ch++) bg_dot += background[ch] * dpx[t][ch];
                dL_dalpha += (-Tfin[t] / ((real)1 - alpha)) * bg_dot; /* :743 */
                real dL_dG = co[3] * dL_dalpha;
                real gdx = G * dx, gdy = G * dy;
                real dG_ddelx = -gdx * co[0] - gdy * co[1];
                real dG_ddely = -gdy * co[2] - gdx * co[1];
                s_m2x += dL_dG * dG_ddelx * ddelx_dx;
                s_m2y += dL_dG * dG_ddely * ddely_dy;
                s_cx += (real)-0.5 * gdx * dx * dL_dG;
                s_cy += (real)-0.5 * gdx * dy * dL_dG;
                s_cw += (real)-0.5 * gdy * dy * dL_dG;
                s_op += G * dL_dalpha;
            }
            /* block reduction + atomicAdd (:759-784) */
            GSO_ATOMIC dL_dmean2D[3 * g + 0] += (real)s_m2x;
            GSO_ATOMIC dL_dmean2D[3 * g + 1] += (real)s_m2y;
            GSO_ATOMIC dL_dconic[4 * g + 0] += (real)s_cx;
            GSO_ATOMIC dL_dconic[4 * g + 1] += (real)s_cy;
            GSO_ATOMIC dL_dconic[4 * g + 3] += (real)s_cw;
            GSO_ATOMIC dL_dopacity[g] += (real)s_op;
            for (int ch = 0; ch < 3; ch++) { GSO_ATOMIC dL_dcolor[3 * g + ch] += (real)s_col[ch]; }
            GSO_ATOMIC dL_ddepth[g] += (real)s_dep;
        }
    }

    /* ---- K8: computeCov2DCUDA, backward.cu:150-346 ---- */
    const real* cov3Ds = cov3D_precomp ? cov3D_precomp : c->cov3D; /* rasterizer_impl.cu:429 */
    GSO_FOR_POINTS
    for (int idx = 0; idx < P; idx++) {
        if (!(c->radii[idx] > 0)) continue;
        const real* cov3D = cov3Ds + 6 * (size_t)idx;
        v3 mean = V3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        real gcx = dL_dconic[4 * idx], gcy = dL_dconic[4 * idx + 1], gcz = dL_dconic[4 * idx + 3];
        v3 t; real txtz, tytz, A[2][3], Jm[2][3], a, b, cc;
        cov2d_parts(mean, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &t, &txtz, &tytz, A, Jm, &a, &b, &cc);
        const real limx = (real)1.3 * tan_fovx, limy = (real)1.3 * tan_fovy;
        const real x_grad_mul = (txtz < -limx || txtz > limx) ? 0 : 1;
        const real y_grad_mul = (tytz < -limy || tytz > limy) ? 0 : 1;
        real V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
        real denom = a * cc - b * b;
        real dL_da = 0, dL_db = 0, dL_dc = 0;
        real denom2inv = (real)1 / ((denom * denom) + (real)0.0000001);
        real* dcov = dL_dcov3D + 6 * (size_t)idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * gcx + 2 * b * cc * gcy + (denom - a * cc) * gcz);
            dL_dc = denom2inv * (-a * a * gcz + 2 * a * b * gcy + (denom - a * cc) * gcx);
            dL_db = denom2inv * 2 * (b * cc * gcx - (denom + 2 * b * b) * gcy + a * b * gcz);
            dcov[0] = (A[0][0] * A[0][0] * dL_da + A[0][0] * A[1][0] * dL_db + A[1][0] * A[1][0] * dL_dc);
            dcov[3] = (A[0][1] * A[0][1] * dL_da + A[0][1] * A[1][1] * dL_db + A[1][1] * A[1][1] * dL_dc);
            dcov[5] = (A[0][2] * A[0][2] * dL_da + A[0][2] * A[1][2] * dL_db + A[1][2] * A[1][2] * dL_dc);
            dcov[1] = 2 * A[0][0] * A[0][1] * dL_da + (A[0][0] * A[1][1] + A[0][1] * A[1][0]) * dL_db + 2 * A[1][0] * A[1][1] * dL_dc;
            dcov[2] = 2 * A[0][0] * A[0][2] * dL_da + (A[0][0] * A[1][2] + A[0][2] * A[1][0]) * dL_db + 2 * A[1][0] * A[1][2] * dL_dc;
            dcov[4] = 2 * A[0][2] * A[0][1] * dL_da + (A[0][1] * A[1][2] + A[0][2] * A[1][1]) * dL_db + 2 * A[1][1] * A[1][2] * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
        /* dL_dT (:244-255) */
        real dT0[3], dT1[3];
        for (int k = 0; k < 3; k++) {
            dT0[k] = 2 * (A[0][0] * V[k][0] + A[0][1] * V[k][1] + A[0][2] * V[k][2]) * dL_da + (A[1][0] * V[k][0] + A[1][1] * V[k][1] + A[1][2] * V[k][2]) * dL_db;
            dT1[k] = 2 * (A[1][0] * V[k][0] + A[1][1] * V[k][1] + A[1][2] * V[k][2]) * dL_dc + (A[0][0] * V[k][0] + A[0][1] * V[k][1] + A[0][2] * V[k][2]) * dL_db;
        }
        const real* vm = viewmatrix;
        /* W[i][k] (glm) = R_cw[i][k] = vm[i + 4k] (:259-262) */
        real dL_dJ00 = vm[0] * dT0[0] + vm[4] * dT0[1] + vm[8] * dT0[2];
        real dL_dJ02 = vm[2] * dT0[0] + vm[6] * dT0[1] + vm[10] * dT0[2];
        real dL_dJ11 = vm[1] * dT1[0] + vm[5] * dT1[1] + vm[9] * dT1[2];
        real dL_dJ12 = vm[2] * dT1[0] + vm[6] * dT1[1] + vm[10] * dT1[2];
        real tz = (real)1 / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        real dL_dtx = x_grad_mul * -focal_x * tz2 * dL_dJ02;
        real dL_dty = y_grad_mul * -focal_y * tz2 * dL_dJ12;
        real dL_dtz = -focal_x * tz2 * dL_dJ00 - focal_y * tz2 * dL_dJ11 + (2 * focal_x * t.x) * tz3 * dL_dJ02 + (2 * focal_y * t.y) * tz3 * dL_dJ12;
        /* pose: dpC/drho = I, dpC/dtheta = -[t]x with the CLAMPED t (:273-288) */
        m33 dpC_drho = mident();
        m33 dpC_dtheta = mneg(skew(t));
        for (int i = 0; i < 3; i++) {
            v3 cr = dpC_drho.c[i], ct = dpC_dtheta.c[i];
            dL_dtau[6 * idx + i] += dL_dtx * cr.x + dL_dty * cr.y + dL_dtz * cr.z;
            dL_dtau[6 * idx + i + 3] += dL_dtx * ct.x + dL_dty * ct.y + dL_dtz * ct.z;
        }
        v3 dL_dmean = transformVec4x3Transpose(V3(dL_dtx, dL_dty, dL_dtz), vm);
        dL_dmean3D[3 * idx + 0] = dL_dmean.x; dL_dmean3D[3 * idx + 1] = dL_dmean.y; dL_dmean3D[3 * idx + 2] = dL_dmean.z; /* assignment, :297 */
        /* rotation part through W (:299-343). glm J[0][0]=fx/tz, J[1][1]=fy/tz, J[0][2]=Jm[0][2], J[1][2]=Jm[1][2] */
        real dW00 = Jm[0][0] * dT0[0], dW01 = Jm[0][0] * dT0[1], dW02 = Jm[0][0] * dT0[2];
        real dW10 = Jm[1][1] * dT1[0], dW11 = Jm[1][1] * dT1[1], dW12 = Jm[1][1] * dT1[2];
        real dW20 = Jm[0][2] * dT0[0] + Jm[1][2] * dT1[0], dW21 = Jm[0][2] * dT0[1] + Jm[1][2] * dT1[1], dW22 = Jm[0][2] * dT0[2] + Jm[1][2] * dT1[2];
        v3 dWc1 = V3(dW00, dW10, dW20), dWc2 = V3(dW01, dW11, dW21), dWc3 = V3(dW02, dW12, dW22);
        v3 c1 = V3(vm[0], vm[1], vm[2]), c2 = V3(vm[4], vm[5], vm[6]), c3 = V3(vm[8], vm[9], vm[10]); /* columns of R_cw */
        m33 n1 = mneg(skew(c1)), n2 = mneg(skew(c2)), n3 = mneg(skew(c3));
        dL_dtau[6 * idx + 3] += dot3(dWc1, n1.c[0]) + dot3(dWc2, n2.c[0]) + dot3(dWc3, n3.c[0]);
        dL_dtau[6 * idx + 4] += dot3(dWc1, n1.c[1]) + dot3(dWc2, n2.c[1]) + dot3(dWc3, n3.c[1]);
        dL_dtau[6 * idx + 5] += dot3(dWc1, n1.c[2]) + dot3(dWc2, n2.c[2]) + dot3(dWc3, n3.c[2]);
    }

    /* ---- K9: preprocessCUDA backward, backward.cu:442-538 ---- */
    GSO_FOR_POINTS
    for (int idx = 0; idx < P; idx++) {
        if (!(c->radii[idx] > 0)) continue;
        v3 m = V3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
        const real* proj = projmatrix;
        real mhx = proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12];
        real mhy = proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13];
        real mhw = proj[3] * m.x + proj[7] * m.y + proj[11] * m.z + proj[15];
        real m_w = (real)1 / (mhw + (real)0.0000001);
        real mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
        real mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
        real g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
        dL_dmean3D[3 * idx + 0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dL_dmean3D[3 * idx + 1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dL_dmean3D[3 * idx + 2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        /* approximate pose Jacobian of the projection (:465-512): only proj_raw[0], [5], [11] are used */
        real alpha = (real)1 * m_w, beta = -mhx * m_w * m_w, gamma = -mhy * m_w * m_w;
        real pa = projmatrix_raw[0], pb = projmatrix_raw[5], pe = projmatrix_raw[11];
        const real* vm = viewmatrix;
        m33 Rm; Rm.c[0] = V3(vm[0], vm[1], vm[2]); Rm.c[1] = V3(vm[4], vm[5], vm[6]); Rm.c[2] = V3(vm[8], vm[9], vm[10]);
        v3 tt = V3(vm[12], vm[13], vm[14]);
        v3 pC = mvec(Rm, m); pC = V3(pC.x + tt.x, pC.y + tt.y, pC.z + tt.z); /* unclamped (:479) */
        m33 dp_rho = mident(), dp_theta = mneg(skew(pC));
        v3 d1 = V3(alpha * pa, 0, beta * pe), d2 = V3(0, alpha * pb, gamma * pe);
        v3 d1r = mvec(mtrans(dp_rho), d1), d2r = mvec(mtrans(dp_rho), d2);
        v3 d1t = mvec(mtrans(dp_theta), d1), d2t = mvec(mtrans(dp_theta), d2);
        real jx[6] = {d1r.x, d1r.y, d1r.z, d1t.x, d1t.y, d1t.z}, jy[6] = {d2r.x, d2r.y, d2r.z, d2t.x, d2t.y, d2t.z};
        for (int i = 0; i < 6; i++) dL_dtau[6 * idx + i] += g2x * jx[i] + g2y * jy[i];
        /* depth (:518-528) */
        real dL_dpCz = dL_ddepth[idx];
        dL_dmean3D[3 * idx + 0] += dL_dpCz * vm[2]; dL_dmean3D[3 * idx + 1] += dL_dpCz * vm[6]; dL_dmean3D[3 * idx + 2] += dL_dpCz * vm[10];
        for (int i = 0; i < 3; i++) { dL_dtau[6 * idx + i] += dL_dpCz * dp_rho.c[i].z; dL_dtau[6 * idx + i + 3] += dL_dpCz * dp_theta.c[i].z; }
        if (shs) sh_backward(idx, D, M, means3D, campos, shs, c->clamped, dL_dcolor, dL_dmean3D, dL_dsh, dL_dtau); /* :533-534 */
        if (scales) cov3d_backward(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot); /* :537-538 */
    }
}
