"""CPU-side checks of the product's boundary: the shared library loads without a GPU and exports every symbol the
headers declare; the Python drop-in mirrors the reference API; and the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from util import REPO

import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C


def _declared_symbols():
    syms = set()
    for h in sorted(f for f in os.listdir(os.path.join(REPO, "include")) if f.endswith(".h")):
        txt = open(os.path.join(REPO, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        syms |= set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", txt))
    syms.discard("gsr_alloc_fn")
    return syms


def test_library_loads_and_exports_all_declared_symbols():
    lib = _C.load_library()
    syms = _declared_symbols()
    assert {"gsr_forward", "gsr_backward", "gsr_mark_visible", "gsr_knn_mean_dist2", "gsr_forward_raw", "gsr_backward_raw",
            "gsr_l1_loss_forward", "gsr_l1_loss_backward", "gsr_adam_step", "gsr_forward_views", "gsr_backward_views", "gsr_views_scratch_size"} <= syms
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/*.h but not exported"
    assert b"gfx950" in lib.gsr_version()
    # sizes are pure host functions
    assert lib.gsr_geometry_buffer_size(1000) > 1000 * 70
    assert lib.gsr_image_buffer_size(640, 480, 200000) >= 640 * 480 * 8
    assert lib.gsr_binning_buffer_size(1000) >= 1000 * (4 + 40 + 8)


def test_work_item_map_is_a_bijection_with_full_pieces_first_and_short_pieces_last():
    """render_bwd's block b takes work item b; the forward pass places a frame's pieces (csrc/gs_device.h item_block_*, a host + device
    function exposed through gsr_debug_item_block): onto [0, n_items) exactly; XCD b % 8 runs its blocks in the order of b / 8 -- first full
    pieces with consecutive ranks (tile order: locality), then partial pieces in ascending rank (= descending length), dealt round robin."""
    lib = _C.load_library()
    lib.gsr_debug_item_block.restype = ctypes.c_int
    lib.gsr_debug_item_block.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_int]
    for n_items, n_partial in [(1, 0), (1, 1), (7, 3), (8, 8), (9, 1), (10, 1), (16, 5), (17, 0), (23, 23), (100, 37), (5548, 1193), (4999, 1200), (64, 1)]:
        n_full = n_items - n_partial
        blocks = [lib.gsr_debug_item_block(n_items, n_partial, f, 0) for f in range(n_full)] + \
                 [lib.gsr_debug_item_block(n_items, n_partial, r, 1) for r in range(n_partial)]
        assert sorted(blocks) == list(range(n_items)), (n_items, n_partial)
        for x in range(8):
            seq = sorted((b // 8, kind, rank) for kind, ranks in ((0, range(n_full)), (1, range(n_partial)))
                         for rank in ranks for b in [blocks[rank + kind * n_full]] if b % 8 == x)
            kinds = [k for _, k, _ in seq]
            assert kinds == sorted(kinds), "an XCD runs its full pieces before its partial ones"
            fulls, parts = [r for _, k, r in seq if k == 0], [r for _, k, r in seq if k == 1]
            assert fulls == list(range(fulls[0], fulls[0] + len(fulls))) if fulls else True
            assert parts == [x + 8 * j for j in range(len(parts))]
    assert lib.gsr_debug_item_block(10, 11, 0, 0) < 0 and lib.gsr_debug_item_block(10, 3, 7, 0) < 0 and lib.gsr_debug_item_block(10, 3, 3, 1) < 0


def test_c_entry_points_reject_bad_arguments_before_touching_the_gpu():
    """Argument validation is host code: negative GSR_ERR_INVALID_ARGUMENT (-1) and a message, no device needed."""
    lib = _C.load_library()
    lib.gsr_last_error.restype = ctypes.c_char_p
    vp, i, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.gsr_forward_raw.restype = i
    lib.gsr_forward_raw.argtypes = [vp] * 6 + [i, i, i, vp, i, i, vp, f, vp, vp, vp, f, f, vp, vp, vp, vp, vp, i, vp]
    assert lib.gsr_forward_raw(None, None, None, None, None, None, 10, 0, 1, None, 64, 64, None, 1.0, None, None, None, 1.0, 1.0,
                               None, None, None, None, None, 0, None) == -1
    assert b"gsr_forward" in lib.gsr_last_error()
    lib.gsr_l1_loss_forward.restype = i
    lib.gsr_l1_loss_forward.argtypes = [i, i] + [vp] * 8 + [f, vp, f, vp, vp, vp]
    assert lib.gsr_l1_loss_forward(64, 64, None, None, None, None, None, None, None, None, 0.9, None, 0.95, None, None, None) == -1
    assert b"gsr_l1_loss_forward" in lib.gsr_last_error()
    import slam_losses
    sl = slam_losses._lib()
    assert sl.gsr_masked_l1_forward(5, None, 64, 64, 2, 3, 1.0, None, None, None) == -1 and b"gsr_masked_l1_forward" in lib.gsr_last_error()
    term = (slam_losses._MaskedTerm * 1)()
    assert sl.gsr_masked_l1_backward(1, term, 64, 64, 4, 3, 1.0, None, None) == -1 and b"gsr_masked_l1_backward" in lib.gsr_last_error()
    lib.gsr_adam_step.restype = i
    lib.gsr_adam_step.argtypes = [i, vp, vp]
    assert lib.gsr_adam_step(9, None, None) == -1 and lib.gsr_adam_step(0, None, None) == 0
    assert lib.gsr_l1_loss_workspace_size() > 0
    # the multi-view entry point rejects bad arguments before it touches the device
    lib.gsr_forward_views.restype = i
    lib.gsr_forward_views.argtypes = [i, vp, vp, vp, vp, i, i, i, vp, i, i, vp, f, f, f, i, vp]
    assert lib.gsr_forward_views(0, None, None, None, None, 10, 0, 1, None, 64, 64, None, 1.0, 1.0, 1.0, 0, None) == -1
    assert b"gsr_forward_views" in lib.gsr_last_error()
    lib.gsr_views_scratch_size.restype = ctypes.c_size_t
    lib.gsr_views_scratch_size.argtypes = [i, i, i, i]
    assert lib.gsr_views_scratch_size(10, 1000, 1, 3) >= 10 * 1000 * 14 * 4
    # deformation_field.h
    import hexplane
    i64 = ctypes.c_int64
    hl = hexplane._lib()
    field = hexplane._Field()
    assert hl.gsr_hexplane_forward(None, 4, None, 3, None, 1, None, None) == -1 and b"null field" in lib.gsr_last_error()
    field.num_levels, field.feat_dim = 0, 32
    assert hl.gsr_hexplane_forward(ctypes.byref(field), 4, None, 3, None, 1, None, None) == -1 and b"num_levels" in lib.gsr_last_error()
    field.num_levels, field.feat_dim = 1, 12
    assert hl.gsr_hexplane_backward(ctypes.byref(field), 4, None, 3, None, 1, None, None, None, None) == -1 and b"feat_dim" in lib.gsr_last_error()
    field.feat_dim = 32
    assert hl.gsr_hexplane_forward(ctypes.byref(field), 4, None, 3, None, 1, None, None) == -1 and b"resolution" in lib.gsr_last_error()
    lib.gsr_linear_wgrad.restype = i
    lib.gsr_linear_wgrad.argtypes = [i64, i, i, vp, i64, vp, i64, vp, vp, vp, vp]
    assert lib.gsr_linear_wgrad(10, 129, 4, None, 129, None, 4, None, None, None, None) == -1 and b"gsr_linear_wgrad" in lib.gsr_last_error()
    lib.gsr_linear_wgrad_workspace_size.restype = ctypes.c_size_t
    lib.gsr_linear_wgrad_workspace_size.argtypes = [i64, i, i]
    assert lib.gsr_linear_wgrad_workspace_size(200000, 128, 64) >= 64 * 129 * 4
    import deformation
    dl = deformation._lib()
    assert dl.gsr_deform_mlp_forward(None, 4, None, None, None) == -1 and b"gsr_deform_mlp_forward" in lib.gsr_last_error()
    mlp = deformation._Mlp(in_dim=100)
    assert dl.gsr_deform_mlp_backward(ctypes.byref(mlp), 4, None, None, None, None, None, None) == -1 and b"multiple of 16" in lib.gsr_last_error()
    assert dl.gsr_deform_mlp_grad_count(128) == 64 * 128 + 64 + 3 * (64 * 64 + 64) + 10 * 64 + 10
    assert dl.gsr_deform_mlp_workspace_size(128) >= 256 * dl.gsr_deform_mlp_grad_count(128) * 4
    # control_nodes.h
    import control_nodes
    cl = control_nodes._lib()
    assert cl.gsr_knn_points(10, 10, 3, 33, None, None, None, None, None) == -1 and b"gsr_knn_points" in lib.gsr_last_error()
    assert cl.gsr_knn_points(0, 10, 3, 3, None, None, None, None, None) == -1          # p2 null with m > 0
    assert cl.gsr_knn_points_batch(2, 10, 10, 3, 33, None, None, None, None, None) == -1 and b"gsr_knn_points_batch" in lib.gsr_last_error()
    assert cl.gsr_knn_points_batch(0, 10, 10, 3, 3, None, None, None, None, None) == 0     # an empty batch is not an error
    blend = control_nodes._Blend(n=4, m=0, K=3, node_stride=3)
    assert cl.gsr_node_blend_forward(ctypes.byref(blend), None, None, None, None, None, None, None) == -1 and b"no control nodes" in lib.gsr_last_error()
    blend = control_nodes._Blend(n=4, m=8, K=9, node_stride=3)
    assert cl.gsr_node_blend_backward(ctypes.byref(blend), *([None] * 15)) == -1 and b"K outside" in lib.gsr_last_error()
    assert cl.gsr_node_blend_workspace_size(100000, 512) >= 512 * 21 * 4 * 2
    # the batched forms: B sets of node attributes per call
    blend = control_nodes._Blend(n=4, m=8, K=3, node_stride=3, x=4096, nodes=4096, node_radius=4096)      # (never dereferenced: the calls return at the batch checks)
    assert cl.gsr_node_blend_forward_batch(ctypes.byref(blend), 0, *([None] * 7)) == -1 and b"1 <= B" in lib.gsr_last_error()
    assert cl.gsr_node_blend_forward_batch(ctypes.byref(blend), 3, *([None] * 7)) == -1 and b"needs node attributes" in lib.gsr_last_error()
    assert cl.gsr_node_blend_backward_batch(ctypes.byref(blend), 2, *([None] * 15)) == -1 and b"gsr_node_blend_backward_batch" in lib.gsr_last_error()
    one, twelve = cl.gsr_node_blend_workspace_size_batch(1000, 512, 1), cl.gsr_node_blend_workspace_size_batch(1000, 512, 12)
    assert twelve >= 12 * (1000 * 3 + 512) * 21 * 4 and one < twelve < 12 * one          # per element: contributions + summed row; the reverse lists once per call
    # round 4: ordered scatter sums, the device-side schedule / keyframe slots of the mapping graph, batched camera steps, scheduled Adam
    cl.gsr_index_csr_workspace_size.restype = ctypes.c_size_t
    assert cl.gsr_index_csr_workspace_size(7, 5120, 512) >= 7 * (5120 + 2 * 512) * 4
    assert cl.gsr_index_csr(0, 10, 4, None, None, None) == -1 and b"gsr_index_csr" in lib.gsr_last_error()
    assert cl.gsr_segment_sum(2, 1, 10, 3, 4, None, None, None, None, None) == -1 and b"gsr_segment_sum" in lib.gsr_last_error()
    cl.gsr_relu_backward_bias_workspace_size.restype = ctypes.c_size_t
    assert cl.gsr_relu_backward_bias_workspace_size(50_000, 256) == ((50_000 + 63) // 64) * 256 * 4 and cl.gsr_relu_backward_bias_workspace_size(0, 256) == 0
    assert cl.gsr_relu_backward_bias(10, 100, None, None, None, None, None, None) == -1 and b"gsr_relu_backward_bias" in lib.gsr_last_error()
    assert cl.gsr_relu_backward_bias(10, 256, None, None, None, None, None, None) == -1
    assert cl.gsr_schedule_advance(None, None, 4, 2, None, None) == -1 and b"gsr_schedule_advance" in lib.gsr_last_error()
    assert cl.gsr_slot_gather(5, None, None, None, 100, None) == -1 and b"gsr_slot_gather" in lib.gsr_last_error()
    assert cl.gsr_slot_gather(0, None, None, None, 100, None) == 0                        # no slots: nothing to do
    assert cl.gsr_camera_steps_launch(13, None, None) == -1 and b"0..12 cameras" in lib.gsr_last_error()
    assert cl.gsr_camera_steps_launch(0, None, None) == 0
    assert cl.gsr_adam_step_scheduled(1, None, None, None) == -1
    out2 = (ctypes.c_float * 2)()
    cl.gsr_adam_coefficients.restype = None
    cl.gsr_adam_coefficients.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    cl.gsr_adam_coefficients(0.01, 0.9, 0.999, 3, out2)
    assert abs(out2[0] - 0.01 / (1 - 0.9 ** 3)) < 1e-8 and abs(out2[1] - (1 - 0.999 ** 3) ** -0.5) < 1e-4
    assert cl.gsr_forward_status_views(None) == 0
    # round 5: strided node attributes, several small sums in one launch, the dense layers' masked input gradient / batched split, Adam with
    # device-side step counts
    blend = control_nodes._Blend(n=4, m=8, K=3, node_stride=3, x=4096, nodes=4096, node_radius=4096, attr_stride=2)
    assert cl.gsr_node_blend_forward(ctypes.byref(blend), None, None, None, None, None, None, None) == -1 and b"attr_stride" in lib.gsr_last_error()
    assert cl.gsr_multi_add(65, None, None) == -1 and b"gsr_multi_add" in lib.gsr_last_error()
    assert cl.gsr_multi_add(0, None, None) == 0
    assert cl.gsr_multi_add(2, None, None) == -1
    import dense_layers
    dn = dense_layers._lib()
    assert dn.gsr_dense_split_many(25, None, None) == -1 and b"gsr_dense_split_many" in lib.gsr_last_error()
    assert dn.gsr_dense_split_many(0, None, None) == 0
    assert dn.gsr_dense_backward_input(10, 256, 256, None, 256, None, None, 0, None, 256, None, None, None) == -1 and b"gsr_dense_backward_input" in lib.gsr_last_error()
    assert dn.gsr_dense_backward_input_workspace_size(33280, 256) >= (33280 // 32) * 256 * 4
    assert dn.gsr_dense_forward(10, 256, 256, None, 256, None, 0, None, None, 0, None, 256, None) == -1
    import fused_adam
    fa = fused_adam._lib()
    assert fa.gsr_adam_step_device_count(33, None, None, None, None) == -1 and b"gsr_adam_step_device_count" in lib.gsr_last_error()
    assert fa.gsr_adam_step_device_count(1, None, None, None, None) == -1
    assert fa.gsr_adam_step_device_count(0, None, None, None, None) == 0


def test_public_names_and_settings_fields_match_reference():
    assert set(dgr.__all__) == {"GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_RasterizeGaussians"}
    assert dgr.GaussianRasterizationSettings._fields == (            # DGR/diff_gaussian_rasterization/__init__.py:173-186
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "projmatrix_raw",
        "sh_degree", "campos", "prefiltered", "debug")
    import inspect
    sig = inspect.signature(dgr.GaussianRasterizer.forward)
    assert list(sig.parameters)[1:] == ["means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations",
                                        "cov3D_precomp", "theta", "rho"]
    from simple_knn._C import distCUDA2  # noqa: F401  (gaussian_model.py:18)


def test_argument_validation_messages():
    r = dgr.GaussianRasterizer(None)
    z = torch.zeros(1, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(z, z, z)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(z, z, z, shs=z, colors_precomp=z)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(z, z, z, shs=z, scales=z)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(z, z, z, shs=z, scales=z, rotations=z, cov3D_precomp=z)


def test_no_cpu_fallback():
    """The product path must not silently compute on the CPU."""
    rs = dgr.GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), torch.eye(4), 0,
                                           torch.zeros(3), False, False)
    r = dgr.GaussianRasterizer(rs)
    P = 4
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(torch.zeros(P, 3), torch.zeros(P, 3), torch.ones(P, 1), shs=torch.zeros(P, 1, 3), scales=torch.ones(P, 3), rotations=torch.ones(P, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r.markVisible(torch.zeros(P, 3))
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        distCUDA2(torch.zeros(P, 3))
    with pytest.raises(RuntimeError, match="num_points, 3"):
        _C.rasterize_gaussians(torch.zeros(3), torch.zeros(P, 4), *([torch.Tensor([])] * 4), 1.0, torch.Tensor([]), torch.eye(4), torch.eye(4),
                               torch.eye(4), 1.0, 1.0, 16, 16, torch.Tensor([]), 0, torch.zeros(3), False, False)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(REPO, "4dgs-slam_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), os.path.join(root, f)
                assert "libgs_oracle" not in txt


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _C.load_library()


def test_render_package_forms_the_visibility_filter_when_it_is_read():
    """gaussian_renderer.render_views / render_flow_views return render()'s dict without the `radii > 0` launch nobody in the batched
    mapping loop reads; indexing the key forms (and keeps) it."""
    import torch
    import gaussian_renderer as gr
    pkg = gr._RenderPackage({"radii": torch.tensor([0, 3, 0, 1])})
    assert "visibility_filter" not in pkg
    assert pkg["visibility_filter"].tolist() == [False, True, False, True] and "visibility_filter" in pkg
    with pytest.raises(KeyError):
        pkg["depth"]


def test_hexplane_workspace_follows_the_ordered_option():
    """gsr_set_option("hex_ordered") (include/gs_rasterizer.h, include/deformation_field.h): process-wide, default 1, value < 0 only reads; the
    sorted backward's workspace carries 8 bytes per plane texel more in ordered mode (host-side arithmetic only: no GPU needed)."""
    import hexplane
    lib = _C.load_library()
    hl = hexplane._lib()
    assert _C.set_option("hex_ordered") == 1 and _C.set_option("hex_ordered", -1) == 1
    C, n, V = 32, 100_000, 8
    res = [[64 * m, 64 * m, 64 * m, 25] for m in (1, 2, 4, 8)]
    field = hexplane._Field()
    field.num_levels, field.feat_dim, field.channels_last = 4, C, 1
    for l, r in enumerate(res):
        for k in range(4):
            field.levels[l].res[k] = r[k]
    combos = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    texels_all = sum(r[a] * r[b] * C for r in res for a, b in combos)
    texels_spatial = sum(r[a] * r[b] * C for r in res for a, b in combos if b != 3)
    columns = sum(r[k] for r in res for k in range(3)) * V * C
    try:
        sizes = {}
        for mode in (0, 1):
            _C.set_option("hex_ordered", mode)
            sizes[mode] = (hl.gsr_hexplane_backward_workspace_size(ctypes.byref(field), n),
                           hl.gsr_hexplane_backward_views_workspace_size(ctypes.byref(field), n, V))
    finally:
        _C.set_option("hex_ordered", 1)
    pad = 512                                                        # the carve rounds every region up to 256 bytes
    assert 0 <= sizes[1][0] - sizes[0][0] - 8 * texels_all <= pad
    assert 0 <= sizes[1][1] - sizes[0][1] - 8 * (texels_spatial + columns) <= pad
    assert lib is not None
