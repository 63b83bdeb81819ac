"""dev: share of a file's non-blank lines (whitespace-normalised) that occur verbatim in a reference file."""
import re, sys
def lines(p):
    out = []
    for l in open(p, errors="ignore"):
        l = re.sub(r"\s+", " ", l.strip())
        if l and not l.startswith("#") and len(l) > 3:
            out.append(l)
    return out
for mine, ref in zip(sys.argv[1::2], sys.argv[2::2]):
    a, b = lines(mine), set(lines(ref))
    hit = [l for l in a if l in b]
    print(f"{mine}: {len(hit)}/{len(a)} = {100 * len(hit) / max(1, len(a)):.1f}% of its lines occur in {ref}")
    if "-v" in sys.argv:
        print("\n".join("    " + h for h in hit))
