"""tools/hot_loop_spills.py <file.s> <kernel-substring>: scratch (spill) accesses inside each innermost loop of a kernel
(blocks the compiler annotates with 'Depth=2' or deeper), grouped by loop header."""
import re
import sys
from collections import defaultdict
s = open(sys.argv[1]).read()
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n', s):
    name = m.group(1)
    if sys.argv[2] not in name:
        continue
    body = s[m.end():]
    body = body[:body.index('.Lfunc_end')]
    loops = defaultdict(lambda: [0, 0, 0])
    hdr = None
    for ln in body.split('\n'):
        mm = re.match(r'^(\.LBB\d+_\d+):\s*;\s*(.*)', ln)
        if mm:
            c = mm.group(2)
            d = re.search(r'Depth=(\d+)', c)
            h = re.search(r'Header=(BB\d+_\d+)', c)
            if d and int(d.group(1)) >= 2:
                hdr = (h.group(1) if h else mm.group(1).lstrip('.L')) + f" depth {d.group(1)}"
            else:
                hdr = None
            continue
        if re.match(r'^\.LBB', ln):
            hdr = None
            continue
        t = ln.strip()
        if hdr and t and not t.startswith((';', '.')):
            loops[hdr][0] += 1
            loops[hdr][1] += 'scratch_' in t
            loops[hdr][2] += 'v_fma' in t
    print(name)
    for k, (n, sc, f) in loops.items():
        if n > 100:
            print(f'   loop {k}: {n} instructions, {f} fp64 fma, {sc} scratch accesses')
