import sys, os
import numpy as np
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/4dgs-slam_amd")
from util import oracle_run, hip_run, make_camera, make_gaussians, make_cotangents, rel_l1
kind = sys.argv[1]
rng = np.random.default_rng(41)
W, H, P = 200, 136, 3000
cam = make_camera(W, H)
g = make_gaussians(P, cam, seed=42, sh_degree=1, scale_mean=0.01)
s = g["scales"].copy(); s[:, 0] *= 40.0 if kind == "needles" else 400.0; s[:, 1:] *= 0.3; g["scales"] = s
if kind == "far_needles":
    g["opacities"] = np.clip(g["opacities"] * 0 + rng.uniform(0.5, 0.99, g["opacities"].shape), 0, 0.99).astype(np.float32)
gc, gd = make_cotangents(cam, seed=43)
bg = np.array([0.3, 0.6, 0.9], np.float32)
o32, _, g32 = oracle_run(g, cam, bg, gc, gd)
o64, _, g64 = oracle_run(g, cam, bg, gc, gd, dtype=np.float64)
oh, gh = hip_run(g, cam, bg, gc, gd)
print(kind, os.environ.get("GSR_LIB", "default").split("/")[-1])
print(" color: hip-vs-o32 %.2e  hip-vs-o64 %.2e  o32-vs-o64 %.2e" % (rel_l1(oh["color"], o32["color"]), rel_l1(oh["color"], o64["color"]), rel_l1(o32["color"], o64["color"])))
for kh, ko in (("means3D", "dL_dmeans3D"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"), ("opacities", "dL_dopacity"), ("means2D", "dL_dmeans2D")):
    print(" %-10s hip-vs-o32 %.2e  hip-vs-o64 %.2e  o32-vs-o64 %.2e" % (kh, rel_l1(gh[kh].reshape(-1), g32[ko].reshape(-1)), rel_l1(gh[kh].reshape(-1), g64[ko].reshape(-1)), rel_l1(g32[ko].reshape(-1), g64[ko].reshape(-1))))
print(" radii mismatch", int((oh["radii"] != o32["radii"]).sum()), "max |color diff| %.3e" % np.abs(oh["color"] - o32["color"]).max())
