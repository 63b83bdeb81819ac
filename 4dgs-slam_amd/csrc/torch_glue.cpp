// torch_glue.cpp -- optional native host glue between PyTorch tensors and the C ABI of include/gs_rasterizer.h.
//
// It is the counterpart of the reference's torch/pybind layer (submodules/diff-gaussian-rasterization/ext.cpp:15-19 and
// rasterize_points.cu:35-232; submodules/simple-knn/ext.cpp, spatial.cu:15-26): same entry points, same argument order,
// same return tuples. All arithmetic stays behind the C ABI (libgs_rasterizer_hip.so); this file only allocates tensors,
// extracts raw pointers and forwards the caller's HIP stream. The ctypes binding in diff_gaussian_rasterization/_C.py does
// the same job in Python and remains the fallback; this one exists because at ~0.4 ms of GPU work per forward+backward the
// Python marshalling (~0.4 ms of host time) was the bottleneck.
//
// Pure C++ (no device code, no HIP headers): the stream is passed in as an integer by the Python caller.
#include <torch/extension.h>
#include <map>
#include <mutex>

#include <cstring>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/gs_rasterizer.h"
#include "../../include/simple_knn.h"
#include "../../include/slam_losses.h"
#include "../../include/control_nodes.h"
#include "../../include/deformation_field.h"

namespace {

constexpr int kChannels = GSR_NUM_CHANNELS;

const float* fptr(const torch::Tensor& t, const char* name)
{
    if (!t.defined() || t.numel() == 0) return nullptr;   // empty tensor == nullptr at the boundary
    TORCH_CHECK(t.is_cuda(), name, " is on '", t.device().str(),
                "': the MI355X rasterizer needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
    TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must be float32");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous (internal error: caller makes it so)");
    return t.data_ptr<float>();
}

torch::Tensor contig(const torch::Tensor& t) { return (t.defined() && t.numel() != 0 && !t.is_contiguous()) ? t.contiguous() : t; }

char* resize_cb(void* user, size_t n)   // the resizeFunctional of rasterize_points.cu:27-33
{
    auto* t = static_cast<torch::Tensor*>(user);
    t->resize_({(int64_t)n});
    return reinterpret_cast<char*>(t->data_ptr());
}

[[noreturn]] void fail(const char* what, int code)
{
    throw std::runtime_error(std::string(what) + " failed (code " + std::to_string(code) + "): " + gsr_last_error());
}

}  // namespace

// RasterizeGaussiansCUDA, rasterize_points.cu:35-122 (+ the stream)
std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_gaussians(const torch::Tensor& background, const torch::Tensor& means3D_, const torch::Tensor& colors_, const torch::Tensor& opacity_,
                    const torch::Tensor& scales_, const torch::Tensor& rotations_, double scale_modifier, const torch::Tensor& cov3D_,
                    const torch::Tensor& viewmatrix_, const torch::Tensor& projmatrix_, const torch::Tensor& projmatrix_raw,
                    double tan_fovx, double tan_fovy, int64_t image_height, int64_t image_width, const torch::Tensor& sh_, int64_t degree,
                    const torch::Tensor& campos_, bool prefiltered, bool debug, int64_t stream)
{
    (void)projmatrix_raw;
    TORCH_CHECK(means3D_.dim() == 2 && means3D_.size(1) == 3, "means3D must have dimensions (num_points, 3)");   // rasterize_points.cu:58-60
    TORCH_CHECK(means3D_.is_cuda(), "means3D is on '", means3D_.device().str(),
                "': the MI355X rasterizer needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
    const int P = (int)means3D_.size(0), H = (int)image_height, W = (int)image_width;
    auto fopt = means3D_.options().dtype(torch::kFloat32);
    auto iopt = means3D_.options().dtype(torch::kInt32);
    auto bopt = means3D_.options().dtype(torch::kUInt8);
    torch::Tensor img = torch::empty({kChannels + 2, H, W}, fopt);
    torch::Tensor ints = torch::empty({2, P}, iopt);
    torch::Tensor out_color = img.narrow(0, 0, kChannels), out_depth = img.narrow(0, kChannels, 1), out_opacity = img.narrow(0, kChannels + 1, 1);
    torch::Tensor radii = ints.select(0, 0), n_touched = ints.select(0, 1);
    torch::Tensor geomBuffer = torch::empty({0}, bopt), binningBuffer = torch::empty({0}, bopt), imgBuffer = torch::empty({0}, bopt);
    int rendered = 0;
    if (P != 0) {
        const int M = sh_.numel() != 0 ? (int)sh_.size(1) : 0;   // rasterize_points.cu:87-91
        const torch::Tensor bg = contig(background), means3D = contig(means3D_), colors = contig(colors_), opacity = contig(opacity_),
                            scales = contig(scales_), rotations = contig(rotations_), cov3D = contig(cov3D_), view = contig(viewmatrix_),
                            proj = contig(projmatrix_), sh = contig(sh_), campos = contig(campos_);
        rendered = gsr_forward(resize_cb, &geomBuffer, resize_cb, &binningBuffer, resize_cb, &imgBuffer, P, (int)degree, M,
                               fptr(bg, "bg"), W, H, fptr(means3D, "means3D"), fptr(sh, "shs"), fptr(colors, "colors_precomp"),
                               fptr(opacity, "opacities"), fptr(scales, "scales"), (float)scale_modifier, fptr(rotations, "rotations"),
                               fptr(cov3D, "cov3D_precomp"), fptr(view, "viewmatrix"), fptr(proj, "projmatrix"), fptr(campos, "campos"),
                               (float)tan_fovx, (float)tan_fovy, prefiltered ? 1 : 0, out_color.data_ptr<float>(), out_depth.data_ptr<float>(),
                               out_opacity.data_ptr<float>(), radii.data_ptr<int>(), n_touched.data_ptr<int>(), debug ? 1 : 0,
                               reinterpret_cast<void*>(stream));
        if (rendered < 0) fail("gsr_forward", rendered);
    } else {
        img.zero_();   // rasterize_points.cu:85: nothing is launched, outputs stay zero
    }
    return std::make_tuple(rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer, out_depth, out_opacity, n_touched);
}

// RasterizeGaussiansBackwardCUDA, rasterize_points.cu:124-211, plus a 10th result: dL_dtau summed over Gaussians, float32[6]
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor>
rasterize_gaussians_backward_fused(const torch::Tensor& background, const torch::Tensor& means3D_, const torch::Tensor& radii_,
                                   const torch::Tensor& colors_, const torch::Tensor& scales_, const torch::Tensor& rotations_,
                                   double scale_modifier, const torch::Tensor& cov3D_, const torch::Tensor& viewmatrix_,
                                   const torch::Tensor& projmatrix_, const torch::Tensor& projmatrix_raw_, double tan_fovx, double tan_fovy,
                                   const torch::Tensor& dL_dout_color_, const torch::Tensor& dL_dout_depths_, const torch::Tensor& sh_,
                                   int64_t degree, const torch::Tensor& campos_, const torch::Tensor& geomBuffer, int64_t R,
                                   const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, bool debug, bool lean,
                                   int64_t stream, const std::vector<torch::Tensor>& accumulate_into)
{
    TORCH_CHECK(means3D_.is_cuda(), "means3D is on '", means3D_.device().str(),
                "': the MI355X rasterizer needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
    const int64_t P = means3D_.size(0);
    const int H = (int)dL_dout_color_.size(1), W = (int)dL_dout_color_.size(2);
    const int64_t M = sh_.numel() != 0 ? sh_.size(1) : 0;
    auto fopt = means3D_.options().dtype(torch::kFloat32);
    // the kernels write every element: one uninitialised allocation carved into the eleven gradient tensors
    // (rasterize_points.cu:160-170 allocates and zero-fills eleven)
    // Layout: the five tensors that become the Gaussian parameters' .grad come first and back to back
    // (means3D, sh, opacity, scales, rotations), so a data-parallel caller can all-reduce them in place as ONE flat range
    // (mapping_shard.GradBucket detects this); the remaining six follow.
    // lean: gradients that only feed other gradients inside the kernel (conic, depth, per-Gaussian tau; colours when they
    // come from SH; cov3D when it comes from scales/rotations) are neither allocated nor written -- 80 of the 148 bytes the
    // geometry kernel stores per Gaussian. The autograd Function asks for this; the reference-shaped entry point does not.
    const bool sh_in = M > 0 && colors_.numel() == 0, cov_in = cov3D_.numel() != 0;
    // accumulate_into = the caller's five parameter-gradient buffers (means3D, sh, opacity, scales, rotations: fp32, contiguous, the
    // parameters' sizes): the kernels ADD this view's gradients to them (GSR_BACKWARD_ACCUMULATE) and nothing is allocated for them
    const bool acc = accumulate_into.size() == 5;
    TORCH_CHECK(acc || accumulate_into.empty(), "accumulate_into: five gradient buffers or none");
    const int64_t pw[5] = {3, 3 * M, 1, 3, 4};
    if (acc) {
        TORCH_CHECK(sh_in && !cov_in, "accumulate_into needs the SH + scales / rotations inputs");
        for (int i = 0; i < 5; i++)
            TORCH_CHECK(accumulate_into[i].is_cuda() && accumulate_into[i].scalar_type() == torch::kFloat32 && accumulate_into[i].is_contiguous() &&
                        accumulate_into[i].numel() == P * pw[i], "accumulate_into[", i, "]: contiguous fp32 device tensor of ", P * pw[i], " elements expected");
    }
    const int64_t widths[11] = {acc ? 0 : 3, acc ? 0 : 3 * M, acc ? 0 : 1, acc ? 0 : 3, acc ? 0 : 4, 3, (lean && sh_in) ? 0 : kChannels, lean ? 0 : 1,
                                lean ? 0 : 4, (lean && !cov_in) ? 0 : 6, lean ? 0 : 6};
    int64_t total = 6;
    for (int64_t w : widths) total += P * w;
    torch::Tensor flat = P == 0 ? torch::zeros({total}, fopt) : torch::empty({total}, fopt);
    torch::Tensor v[11];
    int64_t o = 0;
    for (int i = 0; i < 11; i++) { v[i] = flat.narrow(0, o, P * widths[i]); o += P * widths[i]; }
    auto shaped = [&](int i, std::vector<int64_t> shape) { return widths[i] ? v[i].view(shape) : v[i]; };   // skipped ones stay empty
    if (acc) for (int i = 0; i < 5; i++) v[i] = accumulate_into[i];
    torch::Tensor dL_dmeans3D = v[0].view({P, 3}), dL_dsh = v[1].view({P, M, 3}), dL_dopacity = v[2].view({P, 1}),
                  dL_dscales = v[3].view({P, 3}), dL_drotations = v[4].view({P, 4}), dL_dmeans2D = v[5].view({P, 3}),
                  dL_dcolors = shaped(6, {P, kChannels}), dL_ddepths = shaped(7, {P, 1}), dL_dconic = shaped(8, {P, 2, 2}),
                  dL_dcov3D = shaped(9, {P, 6}), dL_dtau = shaped(10, {P, 6});
    auto optr = [](const torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; };
    torch::Tensor tau_sum = flat.narrow(0, o, 6);
    if (P != 0) {
        const bool sh_path = M > 0 && colors_.numel() == 0;
        if (M > 0 && !sh_path) dL_dsh.zero_();   // colours were precomputed: the SH branch is not taken (backward.cu:533)
        const torch::Tensor bg = contig(background), means3D = contig(means3D_), colors = contig(colors_), scales = contig(scales_),
                            rotations = contig(rotations_), cov3D = contig(cov3D_), view = contig(viewmatrix_), proj = contig(projmatrix_),
                            proj_raw = contig(projmatrix_raw_), sh = contig(sh_), campos = contig(campos_), radii = contig(radii_),
                            gc = contig(dL_dout_color_.scalar_type() == torch::kFloat32 ? dL_dout_color_ : dL_dout_color_.to(torch::kFloat32)),
                            gd = contig(dL_dout_depths_.scalar_type() == torch::kFloat32 ? dL_dout_depths_ : dL_dout_depths_.to(torch::kFloat32));
        TORCH_CHECK(radii.is_cuda() && radii.scalar_type() == torch::kInt32, "radii must be an int32 device tensor");
        const int rc = gsr_backward_fused(
            (int)P, (int)degree, (int)M, (int)R, fptr(bg, "bg"), W, H, fptr(means3D, "means3D"), fptr(sh, "shs"), fptr(colors, "colors_precomp"),
            fptr(scales, "scales"), (float)scale_modifier, fptr(rotations, "rotations"), fptr(cov3D, "cov3D_precomp"), fptr(view, "viewmatrix"),
            fptr(proj, "projmatrix"), fptr(proj_raw, "projmatrix_raw"), fptr(campos, "campos"), (float)tan_fovx, (float)tan_fovy,
            radii.data_ptr<int>(), reinterpret_cast<char*>(geomBuffer.data_ptr()), reinterpret_cast<char*>(binningBuffer.data_ptr()),
            reinterpret_cast<char*>(imageBuffer.data_ptr()), fptr(gc, "dL_dout_color"), fptr(gd, "dL_dout_depth"), dL_dmeans2D.data_ptr<float>(),
            optr(dL_dconic), dL_dopacity.data_ptr<float>(), optr(dL_dcolors), optr(dL_ddepths),
            dL_dmeans3D.data_ptr<float>(), optr(dL_dcov3D), sh_path ? dL_dsh.data_ptr<float>() : nullptr, dL_dscales.data_ptr<float>(),
            dL_drotations.data_ptr<float>(), optr(dL_dtau), tau_sum.data_ptr<float>(), (debug ? 1 : 0) | (acc ? GSR_BACKWARD_ACCUMULATE : 0),
            reinterpret_cast<void*>(stream));
        if (rc < 0) fail("gsr_backward", rc);
    }
    return std::make_tuple(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dtau, tau_sum);
}

// ---- the autograd node of GaussianRasterizer in C++ (diff_gaussian_rasterization/autograd.py _RasterizeGaussians is the same node in
// Python and stays the reference path: debug mode, fused gradient accumulation and the ctypes binding go through it). A forward +
// backward of BASELINE config #2 is 0.24 ms of GPU time; the Python node costs ~75 us of host time per step on top of the two calls
// above (Function.apply, ctx bookkeeping, argument tuples, the backward trampoline), which a slow host cannot hide behind the 80 us of
// GPU work between the binning mailbox and the backward launches. Inputs 0-9 are the reference's differentiable inputs in its order
// (DGR/diff_gaussian_rasterization/__init__.py:44-56), the rest are the raster settings.
namespace {
struct RasterizeNode : public torch::autograd::Function<RasterizeNode> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const torch::Tensor& means3D, const torch::Tensor& means2D,
                                                  const torch::Tensor& sh, const torch::Tensor& colors, const torch::Tensor& opacities,
                                                  const torch::Tensor& scales, const torch::Tensor& rotations, const torch::Tensor& cov3D,
                                                  const torch::Tensor& theta, const torch::Tensor& rho, const torch::Tensor& bg,
                                                  double scale_modifier, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                                                  const torch::Tensor& projmatrix_raw, double tan_fovx, double tan_fovy, int64_t H, int64_t W,
                                                  int64_t degree, const torch::Tensor& campos, bool prefiltered, int64_t stream)
    {
        (void)means2D;
        auto r = rasterize_gaussians(bg, means3D, colors, opacities, scales, rotations, scale_modifier, cov3D, viewmatrix, projmatrix, projmatrix_raw,
                                     tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, false, stream);
        ctx->saved_data["R"] = (int64_t)std::get<0>(r);
        ctx->saved_data["scale_modifier"] = scale_modifier;
        ctx->saved_data["tan_fovx"] = tan_fovx;
        ctx->saved_data["tan_fovy"] = tan_fovy;
        ctx->saved_data["degree"] = degree;
        ctx->saved_data["stream"] = stream;
        ctx->saved_data["theta_3"] = theta.defined() && theta.numel() == 3;
        ctx->saved_data["rho_3"] = rho.defined() && rho.numel() == 3;
        ctx->saved_data["theta_shape"] = theta.defined() ? theta.sizes().vec() : std::vector<int64_t>{};
        ctx->saved_data["rho_shape"] = rho.defined() ? rho.sizes().vec() : std::vector<int64_t>{};
        const torch::Tensor &color = std::get<1>(r), &radii = std::get<2>(r), &depth = std::get<6>(r), &opacity = std::get<7>(r), &n_touched = std::get<8>(r);
        ctx->save_for_backward({colors, means3D, scales, rotations, cov3D, radii, sh, std::get<3>(r), std::get<4>(r), std::get<5>(r), bg, viewmatrix,
                                projmatrix, projmatrix_raw, campos});
        ctx->mark_non_differentiable({radii, n_touched});
        ctx->set_materialize_grads(false);        // unused cotangents (opacity, radii, n_touched) arrive undefined, not zero-filled
        return {color, radii, depth, opacity, n_touched};
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g)
    {
        const auto sv = ctx->get_saved_variables();
        const torch::Tensor &colors = sv[0], &means3D = sv[1], &scales = sv[2], &rotations = sv[3], &cov3D = sv[4], &radii = sv[5], &sh = sv[6],
                            &geom = sv[7], &binning = sv[8], &img = sv[9], &bg = sv[10], &view = sv[11], &proj = sv[12], &proj_raw = sv[13], &campos = sv[14];
        auto fopt = means3D.options().dtype(torch::kFloat32);
        torch::Tensor g_color = g[0], g_depth = g[2];
        // the image size is that of whichever cotangent arrived; with neither there is nothing to back-propagate
        if (!g_color.defined() && !g_depth.defined()) return torch::autograd::variable_list(23);
        const int64_t h = g_color.defined() ? g_color.size(1) : g_depth.size(1), w = g_color.defined() ? g_color.size(2) : g_depth.size(2);
        if (!g_color.defined()) g_color = torch::zeros({kChannels, h, w}, fopt);
        if (!g_depth.defined()) g_depth = torch::zeros({1, h, w}, fopt);
        auto r = rasterize_gaussians_backward_fused(bg, means3D, radii, colors, scales, rotations, ctx->saved_data["scale_modifier"].toDouble(), cov3D, view, proj,
                                                    proj_raw, ctx->saved_data["tan_fovx"].toDouble(), ctx->saved_data["tan_fovy"].toDouble(), g_color, g_depth, sh,
                                                    ctx->saved_data["degree"].toInt(), campos, geom, ctx->saved_data["R"].toInt(), binning, img, false, true,
                                                    ctx->saved_data["stream"].toInt(), {});
        const torch::Tensor& tau = std::get<9>(r);
        auto pose = [&](int64_t at, const char* is3, const char* shape) {     // a 3-element input gets its gradient in its own shape, else [1,3]
            torch::Tensor v = tau.narrow(0, at, 3);
            return ctx->saved_data[is3].toBool() ? v.view(ctx->saved_data[shape].toIntVector()) : v.view({1, 3});
        };
        auto some = [](const torch::Tensor& t) { return t.numel() ? t : torch::Tensor(); };     // gradients of inputs that were not given: undefined
        torch::autograd::variable_list out(23);
        out[0] = std::get<3>(r);            // means3D
        out[1] = std::get<0>(r);            // means2D
        out[2] = some(std::get<5>(r));      // sh
        out[3] = some(std::get<1>(r));      // colors_precomp
        out[4] = std::get<2>(r);            // opacities
        out[5] = some(std::get<6>(r));      // scales
        out[6] = some(std::get<7>(r));      // rotations
        out[7] = some(std::get<4>(r));      // cov3D_precomp
        out[8] = pose(3, "theta_3", "theta_shape");
        out[9] = pose(0, "rho_3", "rho_shape");
        return out;
    }
};
}  // namespace

std::vector<torch::Tensor> rasterize_autograd(const torch::Tensor& means3D, const torch::Tensor& means2D, const torch::Tensor& sh, const torch::Tensor& colors,
                                              const torch::Tensor& opacities, const torch::Tensor& scales, const torch::Tensor& rotations,
                                              const torch::Tensor& cov3D, const torch::Tensor& theta, const torch::Tensor& rho, const torch::Tensor& bg,
                                              double scale_modifier, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix,
                                              const torch::Tensor& projmatrix_raw, double tan_fovx, double tan_fovy, int64_t H, int64_t W, int64_t degree,
                                              const torch::Tensor& campos, bool prefiltered, int64_t stream)
{
    return RasterizeNode::apply(means3D, means2D, sh, colors, opacities, scales, rotations, cov3D, theta, rho, bg, scale_modifier, viewmatrix, projmatrix,
                                projmatrix_raw, tan_fovx, tan_fovy, H, W, degree, campos, prefiltered, stream);
}

// ---- fused prologue: gsr_forward_raw / gsr_backward_raw, see diff_gaussian_rasterization/raw.py ----
namespace {
const torch::Tensor& nz(const c10::optional<torch::Tensor>& t, const torch::Tensor& empty) { return t.has_value() ? *t : empty; }

gsr_raw_inputs describe(const torch::Tensor& xyz, const torch::Tensor& log_scales, const torch::Tensor& raw_rot, const torch::Tensor& logit,
                        const torch::Tensor& f_dc, const torch::Tensor& f_rest, const torch::Tensor& dyn_slot, const torch::Tensor& dx,
                        const torch::Tensor& ds, const torch::Tensor& dr, const torch::Tensor& gather)
{
    gsr_raw_inputs d{};
    if (gather.defined() && gather.numel() != 0) {
        TORCH_CHECK(gather.is_cuda() && gather.scalar_type() == torch::kInt32 && gather.is_contiguous(), "gather must be a contiguous int32 device tensor");
        d.gather = gather.data_ptr<int>();
    }
    d.xyz = fptr(xyz, "_xyz"); d.log_scales = fptr(log_scales, "_scaling"); d.scale_dim = (int)log_scales.size(-1);
    d.raw_rotations = fptr(raw_rot, "_rotation"); d.logit_opacity = fptr(logit, "_opacity");
    d.features_dc = fptr(f_dc, "_features_dc"); d.features_rest = fptr(f_rest, "_features_rest");
    if (dyn_slot.defined() && dyn_slot.numel() != 0) {
        TORCH_CHECK(dyn_slot.is_cuda() && dyn_slot.scalar_type() == torch::kInt32 && dyn_slot.is_contiguous(), "dyn_slot must be a contiguous int32 device tensor");
        d.dyn_slot = dyn_slot.data_ptr<int>();
    }
    d.dx = fptr(dx, "dx"); d.ds = fptr(ds, "ds"); d.dr = fptr(dr, "dr");
    return d;
}
}  // namespace

std::tuple<int, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_gaussians_raw(const torch::Tensor& background, const torch::Tensor& xyz_, const torch::Tensor& log_scales_, const torch::Tensor& raw_rot_,
                        const torch::Tensor& logit_, const torch::Tensor& f_dc_, const c10::optional<torch::Tensor>& f_rest_,
                        const c10::optional<torch::Tensor>& dyn_slot_, const c10::optional<torch::Tensor>& dx_,
                        const c10::optional<torch::Tensor>& ds_, const c10::optional<torch::Tensor>& dr_, double scale_modifier,
                        const torch::Tensor& viewmatrix_, const torch::Tensor& projmatrix_, double tan_fovx, double tan_fovy,
                        int64_t image_height, int64_t image_width, int64_t degree, const torch::Tensor& campos_, bool debug,
                        const c10::optional<torch::Tensor>& gather_, int64_t stream)
{
    TORCH_CHECK(xyz_.dim() == 2 && xyz_.size(1) == 3 && xyz_.size(0) > 0, "_xyz must have dimensions (num_points > 0, 3)");
    TORCH_CHECK(xyz_.is_cuda(), "_xyz is on '", xyz_.device().str(),
                "': the MI355X rasterizer needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
    const torch::Tensor none;
    const torch::Tensor gather = contig(nz(gather_, none));
    const bool masked = gather.defined() && gather.numel() != 0;
    const int P = (int)(masked ? gather.size(0) : xyz_.size(0)), H = (int)image_height, W = (int)image_width;   // rasterized Gaussians
    auto fopt = xyz_.options().dtype(torch::kFloat32);
    auto iopt = xyz_.options().dtype(torch::kInt32);
    auto bopt = xyz_.options().dtype(torch::kUInt8);
    torch::Tensor img = torch::empty({kChannels + 2, H, W}, fopt);
    torch::Tensor ints = torch::empty({2, P}, iopt);
    torch::Tensor out_color = img.narrow(0, 0, kChannels), out_depth = img.narrow(0, kChannels, 1), out_opacity = img.narrow(0, kChannels + 1, 1);
    torch::Tensor radii = ints.select(0, 0), n_touched = ints.select(0, 1);
    torch::Tensor geomBuffer = torch::empty({0}, bopt), binningBuffer = torch::empty({0}, bopt), imgBuffer = torch::empty({0}, bopt);
    const torch::Tensor bg = contig(background), xyz = contig(xyz_), ls = contig(log_scales_), rr = contig(raw_rot_), lo = contig(logit_),
                        fdc = contig(f_dc_), frest = contig(nz(f_rest_, none)), slot = contig(nz(dyn_slot_, none)), dx = contig(nz(dx_, none)),
                        ds = contig(nz(ds_, none)), dr = contig(nz(dr_, none)), view = contig(viewmatrix_), proj = contig(projmatrix_),
                        campos = contig(campos_);
    const int M = 1 + (frest.defined() && frest.numel() != 0 ? (int)frest.size(1) : 0);
    const gsr_raw_inputs in = describe(xyz, ls, rr, lo, fdc, frest, slot, dx, ds, dr, gather);
    const int rendered = gsr_forward_raw(resize_cb, &geomBuffer, resize_cb, &binningBuffer, resize_cb, &imgBuffer, P, (int)degree, M, fptr(bg, "bg"),
                                         W, H, &in, (float)scale_modifier, fptr(view, "viewmatrix"), fptr(proj, "projmatrix"), fptr(campos, "campos"),
                                         (float)tan_fovx, (float)tan_fovy, out_color.data_ptr<float>(), out_depth.data_ptr<float>(),
                                         out_opacity.data_ptr<float>(), radii.data_ptr<int>(), n_touched.data_ptr<int>(), debug ? 1 : 0,
                                         reinterpret_cast<void*>(stream));
    if (rendered < 0) fail("gsr_forward_raw", rendered);
    return std::make_tuple(rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer, out_depth, out_opacity, n_touched);
}

// returns (g_xyz, g_f_dc, g_f_rest, g_logit, g_log_scales, g_raw_rot, g_means2D, g_dx, g_ds, g_dr, tau_sum): the first six are
// views of one allocation in the optimizer's parameter order
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor,
           torch::Tensor, torch::Tensor>
rasterize_gaussians_raw_backward(const torch::Tensor& background, const torch::Tensor& xyz_, const torch::Tensor& log_scales_,
                                 const torch::Tensor& raw_rot_, const torch::Tensor& logit_, const torch::Tensor& f_dc_,
                                 const c10::optional<torch::Tensor>& f_rest_, const c10::optional<torch::Tensor>& dyn_slot_,
                                 const c10::optional<torch::Tensor>& dx_, const c10::optional<torch::Tensor>& ds_,
                                 const c10::optional<torch::Tensor>& dr_, double scale_modifier, const torch::Tensor& viewmatrix_,
                                 const torch::Tensor& projmatrix_, const torch::Tensor& projmatrix_raw_, double tan_fovx, double tan_fovy,
                                 const torch::Tensor& dL_dout_color_, const torch::Tensor& dL_dout_depths_, int64_t degree,
                                 const torch::Tensor& campos_, const torch::Tensor& radii_, const torch::Tensor& geomBuffer, int64_t R,
                                 const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer, bool debug,
                                 const c10::optional<torch::Tensor>& gather_, int64_t stream, const std::vector<torch::Tensor>& accumulate_into,
                                 bool pose_only)
{
    const torch::Tensor none;
    const torch::Tensor gather = contig(nz(gather_, none));
    const bool masked = gather.defined() && gather.numel() != 0;
    const int64_t P = xyz_.size(0), S = log_scales_.size(-1);   // P: rows of the raw tensors (= of every gradient)
    // accumulate_into = the model's six gradient buffers in the optimizer's order (xyz, f_dc, f_rest, opacity, scaling, rotation): the
    // kernels add this view's gradients to them (GSR_BACKWARD_ACCUMULATE); nothing is allocated or zero-filled for them
    // pose_only: no parameter gradients at all (GSR_BACKWARD_POSE_ONLY): nothing is allocated for them, empty tensors come back
    const bool acc = accumulate_into.size() == 6 && !pose_only;
    TORCH_CHECK(accumulate_into.size() == 6 || accumulate_into.empty(), "accumulate_into: six gradient buffers or none");
    const int H = (int)dL_dout_color_.size(1), W = (int)dL_dout_color_.size(2);
    const torch::Tensor bg = contig(background), xyz = contig(xyz_), ls = contig(log_scales_), rr = contig(raw_rot_), lo = contig(logit_),
                        fdc = contig(f_dc_), frest = contig(nz(f_rest_, none)), slot = contig(nz(dyn_slot_, none)), dx = contig(nz(dx_, none)),
                        ds = contig(nz(ds_, none)), dr = contig(nz(dr_, none)), view = contig(viewmatrix_), proj = contig(projmatrix_),
                        proj_raw = contig(projmatrix_raw_), campos = contig(campos_), radii = contig(radii_),
                        gc = contig(dL_dout_color_.scalar_type() == torch::kFloat32 ? dL_dout_color_ : dL_dout_color_.to(torch::kFloat32)),
                        gd = contig(dL_dout_depths_.scalar_type() == torch::kFloat32 ? dL_dout_depths_ : dL_dout_depths_.to(torch::kFloat32));
    const int64_t M = 1 + (frest.defined() && frest.numel() != 0 ? frest.size(1) : 0);
    auto fopt = xyz.options().dtype(torch::kFloat32);
    const int64_t pw[6] = {3, 3, 3 * (M - 1), 1, S, 4};
    if (acc)
        for (int i = 0; i < 6; i++)
            TORCH_CHECK(accumulate_into[i].is_cuda() && accumulate_into[i].scalar_type() == torch::kFloat32 && accumulate_into[i].is_contiguous() &&
                        accumulate_into[i].numel() == P * pw[i], "accumulate_into[", i, "]: contiguous fp32 device tensor of ", P * pw[i], " elements expected");
    const bool own = !acc && !pose_only;      // the six parameter gradients are allocated here
    const int64_t widths[7] = {own ? 3 : 0, own ? 3 : 0, own ? 3 * (M - 1) : 0, own ? 1 : 0, own ? S : 0, own ? 4 : 0, 3};
    int64_t total = 6;
    for (int64_t w : widths) total += P * w;
    torch::Tensor flat = masked ? torch::zeros({total}, fopt) : torch::empty({total}, fopt);   // with a mask only the selected rows are written
    torch::Tensor v[7];
    int64_t o = 0;
    for (int i = 0; i < 7; i++) { v[i] = flat.narrow(0, o, P * widths[i]); o += P * widths[i]; }
    if (acc) for (int i = 0; i < 6; i++) v[i] = accumulate_into[i].view({-1});
    torch::Tensor g_xyz = v[0], g_fdc = v[1], g_frest = v[2], g_logit = v[3], g_ls = v[4], g_rot = v[5], g_m2d = v[6].view({P, 3});
    if (!pose_only) {
        g_xyz = v[0].view({P, 3}); g_fdc = v[1].view({P, 1, 3}); g_frest = v[2].view({P, M - 1, 3}); g_logit = v[3].view(lo.sizes());
        g_ls = v[4].view({P, S}); g_rot = v[5].view({P, 4});
    }
    torch::Tensor tau_sum = flat.narrow(0, o, 6);
    auto zeros_like_opt = [&](const torch::Tensor& t) { return (t.defined() && t.numel() != 0) ? torch::zeros_like(t, fopt) : torch::Tensor(); };
    torch::Tensor g_dx = zeros_like_opt(dx), g_ds = zeros_like_opt(ds), g_dr = zeros_like_opt(dr);
    const gsr_raw_inputs in = describe(xyz, ls, rr, lo, fdc, frest, slot, dx, ds, dr, gather);
    gsr_raw_grads out{};
    if (!pose_only) {
        out.xyz = g_xyz.data_ptr<float>(); out.log_scales = g_ls.data_ptr<float>(); out.raw_rotations = g_rot.data_ptr<float>();
        out.logit_opacity = g_logit.data_ptr<float>(); out.features_dc = g_fdc.data_ptr<float>();
        out.features_rest = M > 1 ? g_frest.data_ptr<float>() : nullptr;
    }
    out.dx = g_dx.defined() ? g_dx.data_ptr<float>() : nullptr; out.ds = g_ds.defined() ? g_ds.data_ptr<float>() : nullptr;
    out.dr = g_dr.defined() ? g_dr.data_ptr<float>() : nullptr;
    TORCH_CHECK(radii.is_cuda() && radii.scalar_type() == torch::kInt32, "radii must be an int32 device tensor");
    const int rc = gsr_backward_raw((int)(masked ? gather.size(0) : P), (int)degree, (int)M, (int)R, fptr(bg, "bg"), W, H, &in, (float)scale_modifier, fptr(view, "viewmatrix"),
                                    fptr(proj, "projmatrix"), fptr(proj_raw, "projmatrix_raw"), fptr(campos, "campos"), (float)tan_fovx,
                                    (float)tan_fovy, radii.data_ptr<int>(), reinterpret_cast<char*>(geomBuffer.data_ptr()),
                                    reinterpret_cast<char*>(binningBuffer.data_ptr()), reinterpret_cast<char*>(imageBuffer.data_ptr()),
                                    fptr(gc, "dL_dout_color"), fptr(gd, "dL_dout_depth"), g_m2d.data_ptr<float>(), &out, tau_sum.data_ptr<float>(),
                                    (debug ? 1 : 0) | (acc ? GSR_BACKWARD_ACCUMULATE : 0) | (pose_only ? GSR_BACKWARD_POSE_ONLY : 0),
                                    reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_backward_raw", rc);
    return std::make_tuple(g_xyz, g_fdc, g_frest, g_logit, g_ls, g_rot, g_m2d, g_dx, g_ds, g_dr, tau_sum);
}

// ---- fused weighted L1 loss (include/slam_losses.h; slam_losses.py) ----------------------------------------------------------
namespace {
const float* optf(const c10::optional<torch::Tensor>& t, std::vector<torch::Tensor>& keep, const char* name)
{
    if (!t.has_value() || !t->defined() || t->numel() == 0) return nullptr;
    torch::Tensor c = t->scalar_type() == torch::kFloat32 ? *t : t->to(torch::kFloat32);
    c = c.contiguous();
    keep.push_back(c);
    return fptr(c, name);
}
}  // namespace

// returns (loss [] , workspace)
std::tuple<torch::Tensor, torch::Tensor> l1_loss_forward(const torch::Tensor& image, const torch::Tensor& depth, const torch::Tensor& gt_image,
                                                         const torch::Tensor& gt_depth, const c10::optional<torch::Tensor>& w_rgb,
                                                         const c10::optional<torch::Tensor>& w_depth, const c10::optional<torch::Tensor>& exposure_a,
                                                         const c10::optional<torch::Tensor>& exposure_b, double alpha,
                                                         const c10::optional<torch::Tensor>& opacity, double opacity_thr, int64_t stream)
{
    TORCH_CHECK(image.is_cuda(), "image is on '", image.device().str(),
                "': the MI355X rasterizer needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
    const int H = (int)image.size(-2), W = (int)image.size(-1);
    std::vector<torch::Tensor> keep;
    torch::Tensor loss = torch::empty({}, image.options().dtype(torch::kFloat32));
    torch::Tensor ws = torch::empty({(int64_t)gsr_l1_loss_workspace_size()}, image.options().dtype(torch::kUInt8));
    const int rc = gsr_l1_loss_forward(W, H, optf(image, keep, "image"), optf(depth, keep, "depth"), optf(gt_image, keep, "gt_image"),
                                       optf(gt_depth, keep, "gt_depth"), optf(w_rgb, keep, "w_rgb"), optf(w_depth, keep, "w_depth"),
                                       optf(exposure_a, keep, "exposure_a"), optf(exposure_b, keep, "exposure_b"), (float)alpha,
                                       optf(opacity, keep, "opacity"), (float)opacity_thr, loss.data_ptr<float>(), reinterpret_cast<char*>(ws.data_ptr()), reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_l1_loss_forward", rc);
    return std::make_tuple(loss, ws);
}

// returns (dL_dimage, dL_ddepth, dL_dexposure[2] or empty)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> l1_loss_backward(
    const torch::Tensor& image, const torch::Tensor& depth, const torch::Tensor& gt_image, const torch::Tensor& gt_depth,
    const c10::optional<torch::Tensor>& w_rgb, const c10::optional<torch::Tensor>& w_depth, const c10::optional<torch::Tensor>& exposure_a,
    const c10::optional<torch::Tensor>& exposure_b, double alpha, const c10::optional<torch::Tensor>& opacity, double opacity_thr,
    const torch::Tensor& upstream, const torch::Tensor& ws, int64_t stream)
{
    const int H = (int)image.size(-2), W = (int)image.size(-1);
    std::vector<torch::Tensor> keep;
    auto fopt = image.options().dtype(torch::kFloat32);
    torch::Tensor g_image = torch::empty(image.sizes(), fopt), g_depth = torch::empty(depth.sizes(), fopt);
    const bool has_exp = exposure_a.has_value() && exposure_a->defined() && exposure_a->numel() != 0;
    torch::Tensor g_exp = has_exp ? torch::empty({2}, fopt) : torch::Tensor();
    const int rc = gsr_l1_loss_backward(W, H, optf(image, keep, "image"), optf(depth, keep, "depth"), optf(gt_image, keep, "gt_image"),
                                        optf(gt_depth, keep, "gt_depth"), optf(w_rgb, keep, "w_rgb"), optf(w_depth, keep, "w_depth"),
                                        optf(exposure_a, keep, "exposure_a"), optf(exposure_b, keep, "exposure_b"), (float)alpha,
                                        optf(opacity, keep, "opacity"), (float)opacity_thr, optf(upstream, keep, "upstream"), g_image.data_ptr<float>(), g_depth.data_ptr<float>(),
                                        has_exp ? g_exp.data_ptr<float>() : nullptr, reinterpret_cast<char*>(ws.data_ptr()),
                                        reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_l1_loss_backward", rc);
    return std::make_tuple(g_image, g_depth, g_exp);
}

// ---- the multi-view entry point (gsr_forward_views / gsr_backward_views): the marshalling of diff_gaussian_rasterization/views.py in C++.
// The autograd Functions stay in Python (they resolve the fused-accumulation targets and hold the per-call context); what moves here is
// the per-view descriptor table, the three allocation callbacks per view and ~25 pointer extractions per view -- 0.3 ms of host time per
// call with ctypes at 6-10 views, once forward and once backward per mapping iteration. Per-view optional tensors travel as lists of
// optionals (None = not given); flow_* lists are empty unless the call renders flow views.
namespace {
using TensorVec = std::vector<torch::Tensor>;
using OptVec = std::vector<c10::optional<torch::Tensor>>;

const float* opt_at(const OptVec& v, size_t i, TensorVec& keep, const char* name)
{
    if (i >= v.size() || !v[i].has_value() || !v[i]->defined() || v[i]->numel() == 0) return nullptr;
    torch::Tensor t = v[i]->detach();
    if (t.scalar_type() != torch::kFloat32) t = t.to(torch::kFloat32);
    t = t.contiguous();
    keep.push_back(t);
    return fptr(t, name);
}
const float* need_at(const TensorVec& v, size_t i, TensorVec& keep, const char* name)
{
    TORCH_CHECK(i < v.size() && v[i].defined(), name, ": one tensor per view expected");
    torch::Tensor t = v[i].detach().contiguous();
    keep.push_back(t);
    return fptr(t, name);
}
void fill_view_cameras(std::vector<gsr_view>& views, const TensorVec& view, const TensorVec& proj, const TensorVec& proj_raw, const TensorVec& campos,
                       const OptVec& dx, const OptVec& ds, const OptVec& dr, const OptVec& flow_dx2, const OptVec& flow_proj1, const OptVec& flow_proj2,
                       TensorVec& keep)
{
    for (size_t v = 0; v < views.size(); v++) {
        gsr_view& w = views[v];
        w.viewmatrix = need_at(view, v, keep, "viewmatrix"); w.projmatrix = need_at(proj, v, keep, "projmatrix");
        w.projmatrix_raw = need_at(proj_raw, v, keep, "projmatrix_raw"); w.cam_pos = need_at(campos, v, keep, "campos");
        w.dx = opt_at(dx, v, keep, "dx"); w.ds = opt_at(ds, v, keep, "ds"); w.dr = opt_at(dr, v, keep, "dr");
        w.flow_dx2 = opt_at(flow_dx2, v, keep, "d_xyz2"); w.flow_proj1 = opt_at(flow_proj1, v, keep, "proj1"); w.flow_proj2 = opt_at(flow_proj2, v, keep, "proj2");
    }
}
}  // namespace

// returns (img [V, C+2, H, W]: colour | depth | opacity per view, ints [V, 2, P]: radii | n_touched, num_rendered per view, the 3 V scratch buffers)
std::tuple<torch::Tensor, torch::Tensor, std::vector<int64_t>, TensorVec>
rasterize_views_forward(const torch::Tensor& background, const torch::Tensor& xyz_, const torch::Tensor& log_scales_, const torch::Tensor& raw_rot_,
                        const torch::Tensor& logit_, const c10::optional<torch::Tensor>& f_dc_, const c10::optional<torch::Tensor>& f_rest_,
                        const c10::optional<torch::Tensor>& dyn_slot_, const TensorVec& view, const TensorVec& proj, const TensorVec& proj_raw,
                        const TensorVec& campos, const OptVec& dx, const OptVec& ds, const OptVec& dr, const OptVec& flow_dx2, const OptVec& flow_proj1,
                        const OptVec& flow_proj2, double scale_modifier, double tan_fovx, double tan_fovy, int64_t H, int64_t W, int64_t degree, bool debug,
                        int64_t stream, const OptVec& flow_clips)
{
    TORCH_CHECK(xyz_.dim() == 2 && xyz_.size(1) == 3 && xyz_.size(0) > 0 && xyz_.is_cuda(), "_xyz must be a (num_points > 0, 3) tensor on a HIP device");
    const int V = (int)view.size(), P = (int)xyz_.size(0);
    const torch::Tensor none;
    TensorVec keep;
    auto fopt = xyz_.options().dtype(torch::kFloat32);
    torch::Tensor img = torch::empty({V, kChannels + 2, H, W}, fopt);
    torch::Tensor ints = torch::empty({V, 2, P}, xyz_.options().dtype(torch::kInt32));
    TensorVec state(3 * (size_t)V);
    for (auto& t : state) t = torch::empty({0}, xyz_.options().dtype(torch::kUInt8));
    const torch::Tensor bg = contig(background.detach()), xyz = contig(xyz_.detach()), ls = contig(log_scales_.detach()), rr = contig(raw_rot_.detach()),
                        lo = contig(logit_.detach()), fdc = contig(nz(f_dc_, none)), frest = contig(nz(f_rest_, none)), slot = contig(nz(dyn_slot_, none));
    const int M = 1 + (frest.defined() && frest.numel() != 0 ? (int)frest.size(1) : 0);
    const gsr_raw_inputs in = describe(xyz, ls, rr, lo, fdc.defined() ? fdc.detach() : fdc, frest.defined() ? frest.detach() : frest, slot, none, none, none, none);
    std::vector<gsr_view> views((size_t)V);
    memset(views.data(), 0, sizeof(gsr_view) * (size_t)V);
    fill_view_cameras(views, view, proj, proj_raw, campos, dx, ds, dr, flow_dx2, flow_proj1, flow_proj2, keep);
    for (int v = 0; v < V; v++) {
        gsr_view& w = views[(size_t)v];
        float* base = img.data_ptr<float>() + (size_t)v * (kChannels + 2) * H * W;
        w.out_color = base; w.out_depth = base + (size_t)kChannels * H * W; w.out_opacity = base + (size_t)(kChannels + 1) * H * W;
        w.radii = ints.data_ptr<int>() + (size_t)v * 2 * P; w.n_touched = w.radii + P;
        w.geometry_user = &state[3 * (size_t)v]; w.binning_user = &state[3 * (size_t)v + 1]; w.image_user = &state[3 * (size_t)v + 2];
        if ((size_t)v < flow_clips.size() && flow_clips[(size_t)v].has_value() && flow_clips[(size_t)v]->defined()) {   // gsr_view.flow_clip: int32 [4] on the device
            const torch::Tensor& c = *flow_clips[(size_t)v];
            TORCH_CHECK(c.is_cuda() && c.scalar_type() == torch::kInt32 && c.numel() == 4 && c.is_contiguous(), "flow clip: int32 [4] contiguous device tensor");
            w.flow_clip = c.data_ptr<int>();
        }
    }
    const int rc = gsr_forward_views(V, views.data(), resize_cb, resize_cb, resize_cb, P, (int)degree, M, fptr(bg, "bg"), (int)W, (int)H, &in, (float)scale_modifier,
                                     (float)tan_fovx, (float)tan_fovy, debug ? 1 : 0, reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_forward_views", rc);
    std::vector<int64_t> rendered((size_t)V);
    for (int v = 0; v < V; v++) rendered[(size_t)v] = views[(size_t)v].num_rendered;
    return std::make_tuple(img, ints, rendered, state);
}

// targets: the six parameter-gradient buffers the kernels add to / write (xyz, f_dc, f_rest, logit, log_scales, raw_rot; f_rest may be empty),
// or an empty list = allocate them here (flow views: xyz only). Returns (the six parameter gradients or undefined, per_view [V, 3 P + 6]:
// screen-space gradient | pose sum, the delta gradients per view: dx, dx2, ds, dr (undefined where the view has no such input)).
std::tuple<TensorVec, torch::Tensor, TensorVec>
rasterize_views_backward(const torch::Tensor& background, const torch::Tensor& xyz_, const torch::Tensor& log_scales_, const torch::Tensor& raw_rot_,
                         const torch::Tensor& logit_, const c10::optional<torch::Tensor>& f_dc_, const c10::optional<torch::Tensor>& f_rest_,
                         const c10::optional<torch::Tensor>& dyn_slot_, const TensorVec& view, const TensorVec& proj, const TensorVec& proj_raw,
                         const TensorVec& campos, const OptVec& dx, const OptVec& ds, const OptVec& dr, const OptVec& flow_dx2, const OptVec& flow_proj1,
                         const OptVec& flow_proj2, double scale_modifier, double tan_fovx, double tan_fovy, int64_t H, int64_t W, int64_t degree,
                         const torch::Tensor& ints, const TensorVec& state, const std::vector<int64_t>& rendered, const TensorVec& g_color,
                         const TensorVec& g_depth, const TensorVec& targets, bool accumulate, bool pose_only, bool debug, int64_t stream)
{
    const int V = (int)view.size(), P = (int)xyz_.size(0), S = (int)log_scales_.size(-1);
    const bool flow = !flow_proj1.empty();
    const torch::Tensor none;
    TensorVec keep;
    auto fopt = xyz_.options().dtype(torch::kFloat32);
    const torch::Tensor bg = contig(background.detach()), xyz = contig(xyz_.detach()), ls = contig(log_scales_.detach()), rr = contig(raw_rot_.detach()),
                        lo = contig(logit_.detach()), fdc = contig(nz(f_dc_, none)), frest = contig(nz(f_rest_, none)), slot = contig(nz(dyn_slot_, none));
    const int M = 1 + (frest.defined() && frest.numel() != 0 ? (int)frest.size(1) : 0);
    const gsr_raw_inputs in = describe(xyz, ls, rr, lo, fdc.defined() ? fdc.detach() : fdc, frest.defined() ? frest.detach() : frest, slot, none, none, none, none);
    // parameter gradients
    TensorVec grads(6);
    gsr_raw_grads out{};
    if (!pose_only) {
        if (!targets.empty()) {
            TORCH_CHECK(targets.size() == 6, "targets: six gradient buffers or none");
            out.xyz = targets[0].data_ptr<float>(); out.features_dc = targets[1].data_ptr<float>();
            out.features_rest = (M > 1 && targets[2].numel()) ? targets[2].data_ptr<float>() : nullptr;
            out.logit_opacity = targets[3].data_ptr<float>(); out.log_scales = targets[4].data_ptr<float>(); out.raw_rotations = targets[5].data_ptr<float>();
        } else if (flow) {
            grads[0] = torch::empty({P, 3}, fopt);
            out.xyz = grads[0].data_ptr<float>();
        } else {
            const int64_t widths[6] = {3, 3, 3 * (M - 1), 1, S, 4};
            int64_t total = 0;
            for (int64_t w_ : widths) total += (int64_t)P * w_;
            torch::Tensor own = torch::empty({total}, fopt);
            int64_t o = 0;
            for (int i = 0; i < 6; i++) { grads[(size_t)i] = own.narrow(0, o, (int64_t)P * widths[i]); o += (int64_t)P * widths[i]; }
            out.xyz = grads[0].data_ptr<float>(); out.features_dc = grads[1].data_ptr<float>(); out.features_rest = M > 1 ? grads[2].data_ptr<float>() : nullptr;
            out.logit_opacity = grads[3].data_ptr<float>(); out.log_scales = grads[4].data_ptr<float>(); out.raw_rotations = grads[5].data_ptr<float>();
        }
    }
    torch::Tensor per_view = torch::empty({V, (int64_t)P * 3 + 6}, fopt);
    // delta gradients: one allocation, one fill for all views (the kernels only write the rows of visible Gaussians)
    TensorVec delta(4 * (size_t)V);
    if (!pose_only) {
        int64_t total = 0;
        auto count = [&](const OptVec& v, size_t i) { return (i < v.size() && v[i].has_value() && v[i]->defined()) ? v[i]->numel() : (int64_t)0; };
        for (size_t v = 0; v < (size_t)V; v++) total += count(dx, v) + count(flow_dx2, v) + count(ds, v) + count(dr, v);
        if (total) {
            torch::Tensor flat = torch::zeros({total}, fopt);
            int64_t o = 0;
            auto take = [&](const OptVec& src, size_t i) {
                const int64_t n = count(src, i);
                if (!n) return torch::Tensor();
                torch::Tensor t = flat.narrow(0, o, n).view((*src[i]).sizes());
                o += n;
                return t;
            };
            for (size_t v = 0; v < (size_t)V; v++) { delta[4 * v] = take(dx, v); delta[4 * v + 1] = take(flow_dx2, v); delta[4 * v + 2] = take(ds, v); delta[4 * v + 3] = take(dr, v); }
        }
    }
    std::vector<gsr_view> views((size_t)V);
    memset(views.data(), 0, sizeof(gsr_view) * (size_t)V);
    fill_view_cameras(views, view, proj, proj_raw, campos, dx, ds, dr, flow_dx2, flow_proj1, flow_proj2, keep);
    TORCH_CHECK((int)state.size() == 3 * V && (int)rendered.size() == V && (int)g_color.size() == V && (int)g_depth.size() == V, "one entry per view expected");
    for (int v = 0; v < V; v++) {
        gsr_view& w = views[(size_t)v];
        w.radii = const_cast<int*>(ints.data_ptr<int>()) + (size_t)v * 2 * P;
        w.geom_buffer = reinterpret_cast<char*>(state[3 * (size_t)v].data_ptr()); w.binning_buffer = reinterpret_cast<char*>(state[3 * (size_t)v + 1].data_ptr());
        w.image_buffer = reinterpret_cast<char*>(state[3 * (size_t)v + 2].data_ptr());
        w.num_rendered = (int)rendered[(size_t)v];
        auto cot = [&](const torch::Tensor& t, const char* name) {
            torch::Tensor c = t.scalar_type() == torch::kFloat32 ? t : t.to(torch::kFloat32);
            c = c.contiguous();
            keep.push_back(c);
            return fptr(c, name);
        };
        w.dL_dcolor = cot(g_color[(size_t)v], "dL_dcolor"); w.dL_ddepth = cot(g_depth[(size_t)v], "dL_ddepth");
        w.dL_dmean2D = per_view.data_ptr<float>() + (size_t)v * ((size_t)P * 3 + 6); w.dL_dtau_sum = w.dL_dmean2D + (size_t)P * 3;
        auto dp = [&](const torch::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; };
        w.ddx = dp(delta[4 * (size_t)v]); w.ddx2 = dp(delta[4 * (size_t)v + 1]); w.dds = dp(delta[4 * (size_t)v + 2]); w.ddr = dp(delta[4 * (size_t)v + 3]);
    }
    torch::Tensor scratch;
    if (!pose_only) scratch = torch::empty({(int64_t)gsr_views_scratch_size(V, P, M, S)}, xyz_.options().dtype(torch::kUInt8));
    const int flags = (debug ? 1 : 0) | (accumulate ? GSR_BACKWARD_ACCUMULATE : 0) | (pose_only ? GSR_BACKWARD_POSE_ONLY : 0);
    const int rc = gsr_backward_views(V, views.data(), P, (int)degree, M, fptr(bg, "bg"), (int)W, (int)H, &in, (float)scale_modifier, (float)tan_fovx, (float)tan_fovy,
                                      &out, scratch.defined() ? reinterpret_cast<char*>(scratch.data_ptr()) : nullptr, flags, reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_backward_views", rc);
    return std::make_tuple(grads, per_view, delta);
}

// The autograd node of slam_losses.weighted_l1_loss in C++ (the Python node _WeightedL1 stays the reference path and serves the ctypes
// binding): the mapping loops call it once per view and iteration, so the Python node's apply / ctx / backward trampoline (~30 us each
// way) was a quarter of a static mapping iteration's host time. Undefined tensors stand for the optional inputs that were not given.
namespace {
using OptTensor = c10::optional<torch::Tensor>;
OptTensor opt_of(const torch::Tensor& t) { return t.defined() ? OptTensor(t) : c10::nullopt; }
torch::Tensor detached(const OptTensor& t) { return (t.has_value() && t->defined()) ? t->detach() : torch::Tensor(); }

// The value of a loss evaluated with compute_value = false: a zero scalar without a launch -- a fresh alias of ONE zero per device (a defined
// placeholder, NOT the loss; nobody may write into it).
static torch::Tensor loss_placeholder(const torch::Tensor& like)
{
    static std::mutex mu;
    static std::map<int, torch::Tensor> zeros;
    std::lock_guard<std::mutex> lock(mu);
    const int dev = like.get_device();
    auto it = zeros.find(dev);
    if (it == zeros.end()) it = zeros.emplace(dev, torch::zeros({}, like.options().dtype(torch::kFloat32))).first;
    return it->second.detach();
}

struct WeightedL1Node : public torch::autograd::Function<WeightedL1Node> {
    // (optional inputs travel as c10::optional: the autograd machinery records layout / device of every Tensor argument, an undefined one has neither)
    static torch::Tensor forward(torch::autograd::AutogradContext* ctx, const torch::Tensor& image, const torch::Tensor& depth, const torch::Tensor& gt_image,
                                 const torch::Tensor& gt_depth, const OptTensor& w_rgb, const OptTensor& w_depth, const OptTensor& exposure_a,
                                 const OptTensor& exposure_b, double alpha, const OptTensor& opacity, double opacity_thr, bool compute_value, int64_t stream)
    {
        const torch::Tensor img = image.detach(), dep = depth.detach(), ea = detached(exposure_a), eb = detached(exposure_b), op = detached(opacity);
        const torch::Tensor wr = detached(w_rgb), wd = detached(w_depth);
        torch::Tensor loss, ws;
        if (compute_value) {
            std::tie(loss, ws) = l1_loss_forward(img, dep, gt_image, gt_depth, opt_of(wr), opt_of(wd), opt_of(ea), opt_of(eb), alpha, opt_of(op), opacity_thr, stream);
        } else {        // the caller only back-propagates: a defined placeholder (NOT the loss), the backward kernels need nothing from the forward pass
            TORCH_CHECK(image.is_cuda(), "image is on '", image.device().str(),
                        "': the MI355X rasterizer needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
            loss = loss_placeholder(image);
            ws = torch::empty({(int64_t)gsr_l1_loss_workspace_size()}, image.options().dtype(torch::kUInt8));
        }
        ctx->saved_data["alpha"] = alpha;
        ctx->saved_data["opacity_thr"] = opacity_thr;
        ctx->saved_data["stream"] = stream;
        ctx->saved_data["mask"] = (int64_t)((wr.defined() ? 1 : 0) | (wd.defined() ? 2 : 0) | (ea.defined() ? 4 : 0) | (op.defined() ? 8 : 0));
        std::vector<torch::Tensor> keep = {img, dep, gt_image, gt_depth, ws};
        for (const torch::Tensor& t : {wr, wd, ea, eb, op}) if (t.defined()) keep.push_back(t);
        ctx->save_for_backward(keep);
        return loss;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g)
    {
        const auto sv = ctx->get_saved_variables();
        torch::autograd::variable_list out(13);
        if (!g[0].defined()) return out;
        const int64_t mask = ctx->saved_data["mask"].toInt();
        size_t at = 5;
        auto next = [&](int64_t bit) { return (mask & bit) ? sv[at++] : torch::Tensor(); };
        const torch::Tensor wr = next(1), wd = next(2), ea = next(4), eb = (mask & 4) ? sv[at++] : torch::Tensor(), op = next(8);
        auto r = l1_loss_backward(sv[0], sv[1], sv[2], sv[3], opt_of(wr), opt_of(wd), opt_of(ea), opt_of(eb), ctx->saved_data["alpha"].toDouble(), opt_of(op),
                                  ctx->saved_data["opacity_thr"].toDouble(), g[0], sv[4], ctx->saved_data["stream"].toInt());
        out[0] = std::get<0>(r);
        out[1] = std::get<1>(r);
        const torch::Tensor& g_exp = std::get<2>(r);
        if (g_exp.defined() && ea.defined()) {
            out[6] = g_exp.narrow(0, 0, 1).view(ea.sizes());
            out[7] = g_exp.narrow(0, 1, 1).view(eb.sizes());
        }
        return out;
    }
};
}  // namespace

torch::Tensor weighted_l1_autograd(const torch::Tensor& image, const torch::Tensor& depth, const torch::Tensor& gt_image, const torch::Tensor& gt_depth,
                                   const c10::optional<torch::Tensor>& w_rgb, const c10::optional<torch::Tensor>& w_depth,
                                   const c10::optional<torch::Tensor>& exposure_a, const c10::optional<torch::Tensor>& exposure_b, double alpha,
                                   const c10::optional<torch::Tensor>& opacity, double opacity_thr, bool compute_value, int64_t stream)
{
    return WeightedL1Node::apply(image, depth, gt_image, gt_depth, w_rgb, w_depth, exposure_a, exposure_b, alpha, opacity, opacity_thr, compute_value, stream);
}

// ---- fused SSIM (include/slam_losses.h) ----
std::tuple<torch::Tensor, torch::Tensor> ssim_forward(const torch::Tensor& img1_, const torch::Tensor& img2_, const c10::optional<torch::Tensor>& mask_,
                                                       int64_t stream)
{
    TORCH_CHECK(img1_.is_cuda(), "img1 is on '", img1_.device().str(),
                "': the MI355X rasterizer needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
    const int C = (int)img1_.size(-3), H = (int)img1_.size(-2), W = (int)img1_.size(-1);
    std::vector<torch::Tensor> keep;
    const float* a = optf(img1_, keep, "img1");
    const float* b = optf(img2_, keep, "img2");
    torch::Tensor m8;
    if (mask_.has_value() && mask_->defined() && mask_->numel() != 0) {
        m8 = mask_->to(torch::kUInt8).contiguous();
        TORCH_CHECK(m8.numel() == (int64_t)H * W, "ssim: mask must have H*W elements");
    }
    torch::Tensor out = torch::empty({}, img1_.options().dtype(torch::kFloat32));
    torch::Tensor ws = torch::empty({(int64_t)gsr_ssim_workspace_size(W, H, C)}, img1_.options().dtype(torch::kUInt8));
    const int rc = gsr_ssim_forward(W, H, C, a, b, m8.defined() ? m8.data_ptr<unsigned char>() : nullptr, out.data_ptr<float>(),
                                    reinterpret_cast<char*>(ws.data_ptr()), reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_ssim_forward", rc);
    return std::make_tuple(out, ws);
}

torch::Tensor ssim_backward(const torch::Tensor& img1_, const torch::Tensor& img2_, const c10::optional<torch::Tensor>& mask_,
                            const torch::Tensor& upstream, const torch::Tensor& ws, int64_t stream)
{
    const int C = (int)img1_.size(-3), H = (int)img1_.size(-2), W = (int)img1_.size(-1);
    std::vector<torch::Tensor> keep;
    const float* a = optf(img1_, keep, "img1");
    const float* b = optf(img2_, keep, "img2");
    torch::Tensor m8;
    if (mask_.has_value() && mask_->defined() && mask_->numel() != 0) m8 = mask_->to(torch::kUInt8).contiguous();
    torch::Tensor grad = torch::empty(img1_.sizes(), img1_.options().dtype(torch::kFloat32));
    const int rc = gsr_ssim_backward(W, H, C, a, b, m8.defined() ? m8.data_ptr<unsigned char>() : nullptr, optf(upstream, keep, "upstream"),
                                     grad.data_ptr<float>(), reinterpret_cast<char*>(ws.data_ptr()), reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_ssim_backward", rc);
    return grad;
}

// markVisible, rasterize_points.cu:213-232
torch::Tensor mark_visible(const torch::Tensor& means3D_, const torch::Tensor& viewmatrix_, const torch::Tensor& projmatrix_, int64_t stream)
{
    TORCH_CHECK(means3D_.is_cuda(), "means3D is on '", means3D_.device().str(),
                "': the MI355X rasterizer needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
    const int P = (int)means3D_.size(0);
    torch::Tensor present = torch::zeros({P}, means3D_.options().dtype(torch::kBool));
    if (P != 0) {
        const torch::Tensor m = contig(means3D_), v = contig(viewmatrix_), pr = contig(projmatrix_);
        const int rc = gsr_mark_visible(P, fptr(m, "means3D"), fptr(v, "viewmatrix"), fptr(pr, "projmatrix"),
                                        reinterpret_cast<unsigned char*>(present.data_ptr()), reinterpret_cast<void*>(stream));
        if (rc < 0) fail("gsr_mark_visible", rc);
    }
    return present;
}

// distCUDA2, simple-knn/spatial.cu:15-26
torch::Tensor dist_cuda2(const torch::Tensor& points_, int64_t stream)
{
    TORCH_CHECK(points_.is_cuda(), "points is on '", points_.device().str(),
                "': the MI355X rasterizer needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
    const int P = (int)points_.size(0);
    torch::Tensor means = torch::zeros({P}, points_.options().dtype(torch::kFloat32));
    if (P != 0) {
        const torch::Tensor pts = contig(points_);
        torch::Tensor ws = torch::empty({(int64_t)gsr_knn_workspace_size(P)}, points_.options().dtype(torch::kUInt8));
        const int rc = gsr_knn_mean_dist2(P, fptr(pts, "points"), means.data_ptr<float>(), reinterpret_cast<char*>(ws.data_ptr()),
                                          reinterpret_cast<void*>(stream));
        if (rc < 0) fail("gsr_knn_mean_dist2", rc);
    }
    return means;
}

// ---- SC-GS control nodes (include/control_nodes.h) ----
namespace {
const float* optp(const c10::optional<torch::Tensor>& t, std::vector<torch::Tensor>& keep, const char* name)
{
    if (!t.has_value() || !t->defined() || t->numel() == 0) return nullptr;
    keep.push_back(t->contiguous());
    return fptr(keep.back(), name);
}

void fill_blend(gsr_node_blend& a, std::vector<torch::Tensor>& keep, const torch::Tensor& x, const c10::optional<torch::Tensor>& mask,
                const torch::Tensor& nodes, const torch::Tensor& radius, const c10::optional<torch::Tensor>& weight,
                const c10::optional<torch::Tensor>& trans, const c10::optional<torch::Tensor>& rot, const c10::optional<torch::Tensor>& scale,
                const c10::optional<torch::Tensor>& local_rot, int64_t K, bool rot_as_residual, int64_t flags)
{
    TORCH_CHECK(x.dim() == 2 && x.size(1) == 3 && nodes.dim() == 2 && nodes.size(1) >= 3, "node blend expects x [N, 3] and nodes [M, >=3]");
    TORCH_CHECK(x.is_cuda(), "x is on '", x.device().str(),
                "': the MI355X library needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
    a = gsr_node_blend{};
    a.n = x.size(0); a.m = (int32_t)nodes.size(0); a.K = (int32_t)K; a.rot_as_residual = rot_as_residual; a.node_stride = (int32_t)nodes.size(1);
    a.flags = (int32_t)flags;
    keep.push_back(x.contiguous()); a.x = keep.back().numel() ? fptr(keep.back(), "x") : nullptr;
    keep.push_back(nodes.contiguous()); a.nodes = fptr(keep.back(), "nodes");
    keep.push_back(radius.contiguous()); a.node_radius = fptr(keep.back(), "node_radius");
    TORCH_CHECK(radius.numel() == a.m, "node_radius must have one value per node");
    a.motion_mask = optp(mask, keep, "motion_mask");
    TORCH_CHECK(!a.motion_mask || keep.back().numel() == a.n, "motion_mask must have one value per Gaussian");
    a.node_weight = optp(weight, keep, "node_weight");
    TORCH_CHECK(!a.node_weight || keep.back().numel() == a.m, "node_weight must have one value per node");
    auto shaped = [&](const c10::optional<torch::Tensor>& t, int64_t c, const char* name) {
        const float* p = optp(t, keep, name);
        TORCH_CHECK(!p || (keep.back().dim() == 2 && keep.back().size(0) == a.m && keep.back().size(1) == c), name, " must have shape (", a.m, ", ", c, ")");
        return p;
    };
    a.node_trans = shaped(trans, 3, "node_trans");
    a.node_rot = shaped(rot, 4, "node_rot");
    a.node_scale = shaped(scale, 3, "node_scale");
    a.node_local_rotation = a.node_trans ? shaped(local_rot, 4, "local_rotation") : nullptr;
    a.local_frame = a.node_local_rotation != nullptr;
}
}  // namespace

std::vector<torch::Tensor> node_blend_forward(const torch::Tensor& x, const c10::optional<torch::Tensor>& mask, const torch::Tensor& nodes,
                                              const torch::Tensor& radius, const c10::optional<torch::Tensor>& weight,
                                              const c10::optional<torch::Tensor>& trans, const c10::optional<torch::Tensor>& rot,
                                              const c10::optional<torch::Tensor>& scale, const c10::optional<torch::Tensor>& local_rot, int64_t K,
                                              bool rot_as_residual, int64_t flags, int64_t stream)
{
    gsr_node_blend a{};
    std::vector<torch::Tensor> keep;
    fill_blend(a, keep, x, mask, nodes, radius, weight, trans, rot, scale, local_rot, K, rot_as_residual, flags);
    auto fopt = x.options().dtype(torch::kFloat32);
    torch::Tensor w = torch::empty({a.n, K}, fopt), dist = torch::empty({a.n, K}, fopt), idx = torch::empty({a.n, K}, x.options().dtype(torch::kInt64));
    const bool blend = a.node_trans != nullptr;
    torch::Tensor d_xyz = torch::empty({blend ? a.n : 0, 3}, fopt), d_rot = torch::empty({blend ? a.n : 0, 4}, fopt), d_scale = torch::empty({blend ? a.n : 0, 3}, fopt);
    const int rc = gsr_node_blend_forward(&a, w.data_ptr<float>(), dist.data_ptr<float>(), idx.data_ptr<int64_t>(), blend ? d_xyz.data_ptr<float>() : nullptr,
                                          blend ? d_rot.data_ptr<float>() : nullptr, blend ? d_scale.data_ptr<float>() : nullptr, reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_node_blend_forward", rc);
    return {w, dist, idx, d_xyz, d_rot, d_scale};
}

// returns (g_radius, g_weight, g_trans, g_rot, g_scale, g_local_rotation); undefined tensors where the input was absent
std::vector<torch::Tensor> node_blend_backward(const torch::Tensor& x, const c10::optional<torch::Tensor>& mask, const torch::Tensor& nodes,
                                               const torch::Tensor& radius, const c10::optional<torch::Tensor>& weight,
                                               const c10::optional<torch::Tensor>& trans, const c10::optional<torch::Tensor>& rot,
                                               const c10::optional<torch::Tensor>& scale, const c10::optional<torch::Tensor>& local_rot, int64_t K,
                                               bool rot_as_residual, int64_t flags, const torch::Tensor& w, const torch::Tensor& dist,
                                               const torch::Tensor& idx, const c10::optional<torch::Tensor>& g_w, const c10::optional<torch::Tensor>& g_xyz,
                                               const c10::optional<torch::Tensor>& g_rot, const c10::optional<torch::Tensor>& g_scale, int64_t stream)
{
    gsr_node_blend a{};
    std::vector<torch::Tensor> keep;
    fill_blend(a, keep, x, mask, nodes, radius, weight, trans, rot, scale, local_rot, K, rot_as_residual, flags);
    auto fopt = x.options().dtype(torch::kFloat32);
    const bool blend = a.node_trans != nullptr;
    torch::Tensor g_radius = torch::empty({a.m}, fopt), g_weight, g_trans, g_nrot, g_nscale, g_local;
    if (a.node_weight) g_weight = torch::empty({a.m}, fopt);
    if (blend) { g_trans = torch::empty({a.m, 3}, fopt); g_nrot = torch::empty({a.m, 4}, fopt); g_nscale = torch::empty({a.m, 3}, fopt); }
    if (a.node_local_rotation) g_local = torch::empty({a.m, 4}, fopt);
    torch::Tensor ws = torch::empty({(int64_t)gsr_node_blend_workspace_size(a.n, a.m)}, x.options().dtype(torch::kUInt8));
    auto p = [](torch::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; };
    const float* cw = optp(g_w, keep, "g_nn_weight");
    const float* cx = blend ? optp(g_xyz, keep, "g_xyz") : nullptr;
    const float* cr = blend ? optp(g_rot, keep, "g_rotation") : nullptr;
    const float* cs = blend ? optp(g_scale, keep, "g_scaling") : nullptr;
    const int rc = gsr_node_blend_backward(&a, a.n ? w.data_ptr<float>() : nullptr, a.n ? dist.data_ptr<float>() : nullptr,
                                           a.n ? idx.data_ptr<int64_t>() : nullptr, cx, cr, cs, cw, p(g_trans), p(g_nrot), p(g_nscale), p(g_local),
                                           p(g_radius), p(g_weight), reinterpret_cast<char*>(ws.data_ptr()), reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_node_blend_backward", rc);
    return {g_radius, g_weight, g_trans, g_nrot, g_nscale, g_local};
}

// ---- HexPlane field + deformation MLP (include/deformation_field.h) ----
namespace {
// 1 = channels_last ([H][W][C] memory), 0 = contiguous ([C][H][W]); anything else is rejected
int plane_layout(const torch::Tensor& p)
{
    const int64_t C = p.size(1), H = p.size(2), W = p.size(3);
    const auto st = p.strides();
    if ((C == 1 || st[1] == 1) && (W == 1 || st[3] == C) && (H == 1 || st[2] == W * C)) return 1;
    if ((W == 1 || st[3] == 1) && (H == 1 || st[2] == W) && (C == 1 || st[1] == H * W)) return 0;
    TORCH_CHECK_VALUE(false, "HexPlane plane has strides that are neither channels_last nor contiguous");
    return -1;
}

void describe_field(gsr_hexplane_field& f, const std::vector<torch::Tensor>& planes, int64_t n_levels, const c10::optional<torch::Tensor>& aabb)
{
    static const int C0[6] = {0, 0, 0, 1, 1, 2}, C1[6] = {1, 2, 3, 2, 3, 3};   // itertools.combinations(range(4), 2)
    TORCH_CHECK_VALUE(n_levels >= 1 && n_levels <= GSR_HEXPLANE_MAX_LEVELS, "HexPlane field with ", n_levels, " levels (1..8 supported)");
    TORCH_CHECK_VALUE((int64_t)planes.size() == 6 * n_levels, "a HexPlane level has six planes (grid_dimensions=2 over 4 input coordinates)");
    f = gsr_hexplane_field{};
    f.num_levels = (int32_t)n_levels;
    const int64_t C = planes[0].size(1);
    f.feat_dim = (int32_t)C;
    f.channels_last = plane_layout(planes[0]);
    f.aabb = (aabb.has_value() && aabb->defined()) ? aabb->data_ptr<float>() : nullptr;
    for (int64_t l = 0; l < n_levels; l++) {
        int res[4] = {0, 0, 0, 0};
        for (int p = 0; p < 6; p++) {
            const torch::Tensor& t = planes[6 * l + p];
            TORCH_CHECK(t.is_cuda(), "HexPlane plane is on '", t.device().str(),
                        "': the MI355X library needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
            TORCH_CHECK_VALUE(t.scalar_type() == torch::kFloat32 && t.dim() == 4 && t.size(0) == 1 && t.size(1) == C, "HexPlane plane must be fp32 [1, ", C, ", H, W]");
            TORCH_CHECK_VALUE(plane_layout(t) == f.channels_last, "all HexPlane planes must share one memory layout");
            const int sz[2] = {(int)t.size(3), (int)t.size(2)}, cc[2] = {C0[p], C1[p]};   // the first coordinate indexes the width
            for (int k = 0; k < 2; k++) {
                TORCH_CHECK_VALUE(res[cc[k]] == 0 || res[cc[k]] == sz[k], "level ", l, ": inconsistent resolution along coordinate ", cc[k], ": ", res[cc[k]], " vs ", sz[k]);
                res[cc[k]] = sz[k];
            }
            f.levels[l].planes[p] = t.data_ptr<float>();
        }
        for (int k = 0; k < 4; k++) f.levels[l].res[k] = res[k];
    }
}

void check_points(const torch::Tensor& xyz, const torch::Tensor& time)
{
    TORCH_CHECK(xyz.is_cuda(), "pts is on '", xyz.device().str(),
                "': the MI355X library needs tensors on a HIP device (device='cuda'); there is no CPU fallback in the product path.");
    TORCH_CHECK_VALUE(xyz.scalar_type() == torch::kFloat32 && time.scalar_type() == torch::kFloat32, "HexPlane inputs must be fp32");
    TORCH_CHECK_VALUE(xyz.dim() == 2 && xyz.size(1) >= 3 && time.dim() == 2 && time.size(0) == xyz.size(0) && xyz.stride(1) == 1,
                      "HexPlane expects pts [n, 3] and timestamps [n, 1]");
}
}  // namespace

torch::Tensor hexplane_forward(const std::vector<torch::Tensor>& planes, int64_t n_levels, const torch::Tensor& xyz, const torch::Tensor& time,
                               const c10::optional<torch::Tensor>& aabb, int64_t stream)
{
    check_points(xyz, time);
    gsr_hexplane_field f;
    describe_field(f, planes, n_levels, aabb);
    torch::Tensor out = torch::empty({xyz.size(0), n_levels * (int64_t)f.feat_dim}, xyz.options().dtype(torch::kFloat32));
    const int rc = gsr_hexplane_forward(&f, xyz.size(0), xyz.data_ptr<float>(), xyz.stride(0), time.data_ptr<float>(), time.stride(0),
                                        out.data_ptr<float>(), reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_hexplane_forward", rc);
    return out;
}

// returns [dL_dxyz (or undefined), gradient of plane 0, plane 1, ... (undefined where not needed)]; the plane gradients are views of ONE
// zeroed buffer with the planes' own strides
std::vector<torch::Tensor> hexplane_backward(const std::vector<torch::Tensor>& planes, int64_t n_levels, const torch::Tensor& xyz,
                                             const torch::Tensor& time, const c10::optional<torch::Tensor>& aabb, const torch::Tensor& g_,
                                             const std::vector<bool>& need_plane, bool need_xyz, bool sorted, int64_t stream)
{
    check_points(xyz, time);
    gsr_hexplane_field f;
    describe_field(f, planes, n_levels, aabb);
    const torch::Tensor g = g_.contiguous();
    int64_t total = 0;
    for (size_t k = 0; k < planes.size(); k++) if (need_plane[k]) total += planes[k].numel();
    torch::Tensor flat = torch::zeros({total}, xyz.options().dtype(torch::kFloat32));
    std::vector<torch::Tensor> out(1 + planes.size());
    int64_t off = 0;
    for (size_t k = 0; k < planes.size(); k++) {
        if (!need_plane[k]) continue;
        out[1 + k] = flat.as_strided(planes[k].sizes(), planes[k].strides(), off);
        f.levels[k / 6].grad_planes[k % 6] = flat.data_ptr<float>() + off;
        off += planes[k].numel();
    }
    if (need_xyz) out[0] = torch::empty({xyz.size(0), 3}, xyz.options().dtype(torch::kFloat32));
    torch::Tensor ws;
    if (sorted && total > 0) ws = torch::empty({(int64_t)gsr_hexplane_backward_workspace_size(&f, xyz.size(0))}, xyz.options().dtype(torch::kUInt8));
    const int rc = gsr_hexplane_backward(&f, xyz.size(0), xyz.data_ptr<float>(), xyz.stride(0), time.data_ptr<float>(), time.stride(0),
                                         g.data_ptr<float>(), need_xyz ? out[0].data_ptr<float>() : nullptr,
                                         ws.defined() ? reinterpret_cast<char*>(ws.data_ptr()) : nullptr, reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_hexplane_backward", rc);
    return out;
}

namespace {
void describe_mlp(gsr_deform_mlp& m, const std::vector<torch::Tensor>& params, int64_t in_dim)
{
    TORCH_CHECK_VALUE(params.size() == 14, "the fused deformation MLP takes W0, b0 and (W1, b1, W2, b2) for three heads");
    for (const auto& p : params) TORCH_CHECK(p.is_cuda() && p.scalar_type() == torch::kFloat32 && p.is_contiguous(), "MLP parameters must be contiguous fp32 HIP tensors");
    m = gsr_deform_mlp{};
    m.W0 = params[0].data_ptr<float>(); m.b0 = params[1].data_ptr<float>(); m.in_dim = (int32_t)in_dim;
    for (int j = 0; j < 3; j++) {
        m.W1[j] = params[2 + 4 * j].data_ptr<float>(); m.b1[j] = params[3 + 4 * j].data_ptr<float>();
        m.W2[j] = params[4 + 4 * j].data_ptr<float>(); m.b2[j] = params[5 + 4 * j].data_ptr<float>();
    }
}
}  // namespace

torch::Tensor deform_mlp_forward(const torch::Tensor& feat_, const std::vector<torch::Tensor>& params, int64_t stream)
{
    const torch::Tensor feat = feat_.contiguous();
    gsr_deform_mlp m;
    describe_mlp(m, params, feat.size(1));
    torch::Tensor out = torch::empty({feat.size(0), 10}, feat.options());
    const int rc = gsr_deform_mlp_forward(&m, feat.size(0), feat.data_ptr<float>(), out.data_ptr<float>(), reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_deform_mlp_forward", rc);
    return out;
}

// returns [dfeat, dW0, db0, (dW1, db1, dW2, db2) x 3]: the parameter gradients are views of one flat buffer in parameter order
std::vector<torch::Tensor> deform_mlp_backward(const torch::Tensor& feat, const std::vector<torch::Tensor>& params, const torch::Tensor& dout_, int64_t stream)
{
    const torch::Tensor dout = dout_.contiguous();
    gsr_deform_mlp m;
    describe_mlp(m, params, feat.size(1));
    const int in_dim = (int)feat.size(1);
    torch::Tensor dfeat = torch::empty_like(feat);
    torch::Tensor flat = torch::empty({(int64_t)gsr_deform_mlp_grad_count(in_dim)}, feat.options());
    torch::Tensor ws = torch::empty({(int64_t)gsr_deform_mlp_workspace_size(in_dim)}, feat.options().dtype(torch::kUInt8));
    const int rc = gsr_deform_mlp_backward(&m, feat.size(0), feat.data_ptr<float>(), dout.data_ptr<float>(), dfeat.data_ptr<float>(),
                                           flat.data_ptr<float>(), reinterpret_cast<char*>(ws.data_ptr()), reinterpret_cast<void*>(stream));
    if (rc < 0) fail("gsr_deform_mlp_backward", rc);
    std::vector<torch::Tensor> out{dfeat};
    int64_t off = 0;
    for (const auto& p : params) { out.push_back(flat.narrow(0, off, p.numel()).view(p.sizes())); off += p.numel(); }
    return out;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("rasterize_gaussians", &rasterize_gaussians);
    m.def("rasterize_gaussians_backward_fused", &rasterize_gaussians_backward_fused);
    m.def("rasterize_autograd", &rasterize_autograd);
    m.def("rasterize_gaussians_raw", &rasterize_gaussians_raw);
    m.def("rasterize_gaussians_raw_backward", &rasterize_gaussians_raw_backward);
    m.def("l1_loss_forward", &l1_loss_forward);
    m.def("l1_loss_backward", &l1_loss_backward);
    m.def("weighted_l1_autograd", &weighted_l1_autograd);
    m.def("rasterize_views_forward", &rasterize_views_forward);
    m.def("rasterize_views_backward", &rasterize_views_backward);
    m.def("ssim_forward", &ssim_forward);
    m.def("ssim_backward", &ssim_backward);
    m.def("mark_visible", &mark_visible);
    m.def("dist_cuda2", &dist_cuda2);
    m.def("node_blend_forward", &node_blend_forward);
    m.def("node_blend_backward", &node_blend_backward);
    m.def("hexplane_forward", &hexplane_forward);
    m.def("hexplane_backward", &hexplane_backward);
    m.def("deform_mlp_forward", &deform_mlp_forward);
    m.def("deform_mlp_backward", &deform_mlp_backward);
}
