#!/usr/bin/env python3
"""similarity.py mine[:a-b] ref[:a-b[,c-d...]] — share of the normalised lines of MINE that also occur, in order, in REF.

The normalisation follows the round-5 review: identifiers lower-cased, '_' dropped, float3 / make_float3 -> v3, the
fabs / sqrt spellings folded, blanks removed, comment-only and brace-only lines dropped.  Needs the reference tree, so it
runs in the build container only; the figures it printed are kept in profiles/r06/similarity.txt.
"""
import difflib, re, sys


def norm(line):
    line = re.sub(r"//.*", "", line)
    line = line.strip().lower().replace("_", "")
    line = re.sub(r"makefloat([234])\(", r"v\1(", line)
    line = re.sub(r"float([234])", r"v\1", line)
    line = re.sub(r"fabsf|fabs", "fabs", line)
    line = re.sub(r"sqrtrn|sqrtf|sqrt", "sqrt", line)
    line = re.sub(r"\bconst\b|forceinline|device|host|inline", "", line)
    line = re.sub(r"\s+", "", line)
    line = line.replace("1.f", "1").replace("0.f", "0").replace(".f", "").replace("f,", ",").replace("f)", ")")
    return line


def load(spec):
    path, _, rng = spec.partition(":")
    lines = open(path, errors="replace").read().splitlines()
    if rng:
        picked = []
        for part in rng.split(","):
            a, b = part.split("-")
            picked += lines[int(a) - 1:int(b)]
        lines = picked
    out = [norm(l) for l in lines]
    return [l for l in out if l and l not in ("{", "}", "};", "break;", "else{", "}else{", "else", "return;")]


def share(mine, ref):
    sm = difflib.SequenceMatcher(None, mine, ref, autojunk=False)
    same = sum(b.size for b in sm.get_matching_blocks())
    return same, len(mine)


if __name__ == "__main__":
    mine, ref = load(sys.argv[1]), load(sys.argv[2])
    same, n = share(mine, ref)
    refset = set(ref)
    anyorder = sum(1 for l in mine if l in refset)
    print(f"{sys.argv[1]} vs {sys.argv[2]}: {same}/{n} = {100.0 * same / max(n, 1):.1f} % in order, {100.0 * anyorder / max(n, 1):.1f} % in any order")
