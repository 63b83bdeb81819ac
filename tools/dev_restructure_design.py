"""One-off (round 6): move the superseded per-round material of DESIGN.md into DESIGN_HISTORY.md and put the new top matter /
section-4 summaries / open list (read from tools/_design_r06/*.md) in its place. Line numbers are those of the round-5 DESIGN.md."""
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = open(os.path.join(R, "DESIGN.md")).read().split("\n")
assert src[135].startswith("## 1. The path"), src[135]
assert src[262].startswith("### What bounds the tile kernels"), src[262]
assert src[551].startswith("## 5. Host orchestration"), src[551]
assert src[1208].startswith("## 11. Open items"), src[1208]
part = lambda name: open(os.path.join(R, "tools", "_design_r06", name)).read().rstrip("\n").split("\n")

history = (["# DESIGN_HISTORY — superseded per-round material of DESIGN.md",
            "",
            "Moved out of DESIGN.md in round 6 (VERDICT r05 item 8) so that its body states the current design and numbers once. Nothing here is needed to",
            "read the code; it is the record of what was measured, built and dropped in rounds 1–5, in the words of those rounds. Section numbers refer to",
            "DESIGN.md's.",
            "",
            "## A. Per-round change summaries and review tables (rounds 2–5)",
            ""] + src[8:135] +
           ["", "## B. §4 as rounds 2–5 wrote it: microbenchmark reading, cycle accounting, per-round kernel changes and measured tables", ""] + src[262:551] +
           ["", "## C. §11 (open items) as rounds 3–5 left it", ""] + src[1209:])
new = src[0:8] + [""] + part("top.md") + [""] + src[135:262] + part("sec4.md") + [""] + src[551:1208] + part("sec11.md") + [""]
open(os.path.join(R, "DESIGN_HISTORY.md"), "w").write("\n".join(history) + "\n")
open(os.path.join(R, "DESIGN.md"), "w").write("\n".join(new))
print(len(src), "->", len(new), "lines; history", len(history))
