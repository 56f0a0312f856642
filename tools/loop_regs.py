"""tools/loop_regs.py <file.s> <kernel-substring> <loop header label, e.g. BB7_1664>: VGPRs never referenced inside a loop
(values that only live THROUGH it) and the scratch accesses per block of the loop."""
import re
import sys
s = open(sys.argv[1]).read()
m = re.search(r'\n(_Z\w*' + re.escape(sys.argv[2]) + r'\w*):', s)
body = s[m.end():]
body = body[:body.index('.Lfunc_end')]
hdr = sys.argv[3]
blocks, cur = [], ['entry', '', []]
for ln in body.split('\n'):
    mm = re.match(r'^(\.LBB\d+_\d+):\s*;?\s*(.*)', ln)
    if mm:
        blocks.append(cur)
        cur = [mm.group(1), mm.group(2), []]
    else:
        t = ln.strip()
        if t and not t.startswith((';', '.')):
            cur[2].append(t)
blocks.append(cur)
inl = [b for b in blocks if hdr in b[1] or b[0] == '.L' + hdr]
regs = set()
for lab, c, ins in inl:
    for x in ins:
        for mm in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', x.split(';')[0]):
            if mm.group(1):
                regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
            else:
                regs.add(int(mm.group(3)))
    sc = [x.split(';')[0].strip() for x in ins if 'scratch_' in x]
    if sc or any('s_barrier' in x for x in ins):
        print(lab, 'n', len(ins), 'barriers', sum('s_barrier' in x for x in ins), 'scratch', len(sc), sc[:8])
print('VGPRs referenced in the loop:', len(regs), ' live-through only:', sorted(set(range(256)) - regs))
