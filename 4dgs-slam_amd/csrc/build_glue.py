#!/usr/bin/env python
"""Builds the optional native host glue diff_gaussian_rasterization/_glue*.so (csrc/torch_glue.cpp: torch tensors -> C ABI).
Plain g++ against the installed PyTorch headers; no device code. libgs_rasterizer_hip.so must have been built first."""
import os
import subprocess
import sys
import sysconfig

import torch
from torch.utils import cpp_extension

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)


def main():
    out = os.path.join(PKG, "diff_gaussian_rasterization", "_glue" + sysconfig.get_config_var("EXT_SUFFIX"))
    src = os.path.join(HERE, "torch_glue.cpp")
    deps = [src, os.path.join(PKG, "..", "include", "gs_rasterizer.h"), os.path.join(PKG, "..", "include", "simple_knn.h"),
            os.path.join(PKG, "..", "include", "slam_losses.h"), os.path.join(PKG, "..", "include", "control_nodes.h"),
            os.path.join(PKG, "..", "include", "deformation_field.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps) and "--force" not in sys.argv:
        print("up to date:", out)
        return
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", out,
           "-DTORCH_EXTENSION_NAME=_glue", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I" + sysconfig.get_paths()["include"]]
    cmd += ["-isystem" + p for p in cpp_extension.include_paths()]
    cmd += ["-L" + tl, "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-L" + PKG, "-lgs_rasterizer_hip",
            "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + tl, "-Wno-unused-function"]
    print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    print("built", out)


if __name__ == "__main__":
    main()
