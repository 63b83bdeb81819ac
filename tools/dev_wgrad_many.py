"""Dev: the node network's D + 1 weight gradients (utils/time_utils.py:428-452's backward) three ways, at R rows:
  library   the rounds 4-5 path: torch.bmm over row groups of ~2000 + a sum over the groups (hipBLASLt fp32)
  single    gsr_dense_wgrad, one launch pair per product (round 5)
  many      gsr_dense_wgrad_many, all products in one launch pair (round 6)
Prints one JSON line per R; under rocprofv3 --kernel-trace --stats the per-kernel split is in the stats file."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "4dgs-slam_amd"))
import torch

import dense_layers as dl
from slam.deform_model import _grad_weight, _row_groups

dev = "cuda:0"
SHAPES = [(256, 84)] + [(256, 256)] * 3 + [(256, 340)] + [(256, 256)] * 3 + [(14, 256)]


def bench(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    which = os.environ.get("WGM_WHICH", "library,single,many").split(",")
    for R in [int(a) for a in sys.argv[1:]] or [33280, 66560]:
        g = torch.Generator(device="cpu").manual_seed(R)
        pairs = [(torch.randn((R, n), generator=g).to(dev), torch.randn((R, k), generator=g).to(dev)) for n, k in SHAPES]
        groups = _row_groups(R)
        same = [i for i, s in enumerate(SHAPES) if s == (256, 256)]
        buf = torch.empty((len(same), groups, 256, 256), device=dev)

        def library():
            out = [None] * len(pairs)
            for i, (G, X) in enumerate(pairs):
                if i in same and groups > 1:
                    torch.bmm(G.view(groups, R // groups, -1).transpose(1, 2), X.view(groups, R // groups, -1), out=buf[same.index(i)])
                else:
                    out[i] = _grad_weight(G, X)
            if groups > 1:
                for j, dW in enumerate(buf.sum(1).unbind(0)):
                    out[same[j]] = dW
            return out

        single = lambda: [dl.dense_wgrad(G, X) for G, X in pairs]
        many = lambda: dl.dense_wgrad_many(pairs)
        row = {"rows": R, "products": len(pairs), "flop": sum(2 * R * n * k for n, k in SHAPES)}
        ref = [G.double().t() @ X.double() for G, X in pairs]
        rel = lambda a, b: float((a.double() - b).abs().sum() / b.abs().sum())
        for name, f in (("library", library), ("single", single), ("many", many)):
            if name not in which:
                continue
            row[name + "_us"] = round(bench(f), 1)
            row[name + "_max_rel_err"] = max(rel(a, b) for a, b in zip(f(), ref))
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
