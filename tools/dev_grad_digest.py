"""Dev: sha256 of the forward outputs and of every gradient of the bench's step (200k Gaussians @640x480), at a few sizes / poses, for the
library named by GSR_LIB (GSR_GLUE=ctypes). Two kernel variants that claim identical arithmetic must print identical lines."""
import hashlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "4dgs-slam_amd")]
import torch
import bench

dev = torch.device("cuda:0")
def digest(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]
for P, kfs in ((200000, (0, 3)), (30000, (1,)), (1000, (0,))):
    sc = bench.Scene(P, dev, keyframes=kfs)
    for k in kfs:
        for p in sc.params + [sc.theta, sc.rho, sc.means2D]:
            p.grad = None
        sc.fwd_bwd(k)
        torch.cuda.synchronize()
        print(P, k, " ".join(digest(p.grad) for p in sc.params + [sc.theta, sc.rho, sc.means2D]))
