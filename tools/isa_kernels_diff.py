#!/usr/bin/env python3
"""isa_kernels_diff.py base.s new.s: which kernels of pt_kernel.hip differ between two `tools/isa_diff.sh --save` dumps, after
renumbering the labels whose numbers are file-global (.Lpost_getpcN): a change to one kernel renumbers them in all the others."""
import re, sys


def kernels(path):
    s = open(path).read()
    d = {}
    for m in re.finditer(r'^(_ZN2pt\w+):.*$', s, re.M):
        if m.group(1) in d:
            continue
        body = s[m.end():s.find('s_endpgm', m.end())]
        seen = {}
        d[m.group(1)] = re.sub(r'\.Lpost_getpc(\d+)', lambda g: '.Lpost_getpc#%d' % seen.setdefault(g.group(1), len(seen)), body)
    return d


a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
changed = [k for k in a if a[k] != b.get(k)] + [k for k in b if k not in a]
print(f"{len(a)} kernels, {len(changed)} differ")
for k in changed:
    print("  ", k)
