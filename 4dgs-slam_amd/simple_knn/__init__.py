"""Drop-in replacement for the reference's ``simple_knn`` package (submodules/simple-knn): ``simple_knn._C.distCUDA2``
(imported at gaussian_splatting/scene/gaussian_model.py:18, called at :235-241,381), backed by the HIP spatial-hash
k-NN of libgs_rasterizer_hip.so (include/simple_knn.h)."""
