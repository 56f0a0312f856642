"""tools/asm_blocks.py <file.s> <function-substring>: per basic block of a kernel -- instruction count, fp64 FMAs, LDS ops,
barriers, MFMAs and SCRATCH (spill) accesses.  Used to check that hot loops are spill-free."""
import re
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n', s):
    name = m.group(1)
    if pat not in name or name.endswith('.kd'):
        continue
    body = s[m.end():]
    body = body[:body.index('.Lfunc_end')]
    blocks, cur = [], ['entry', []]
    for ln in body.split('\n'):
        mm = re.match(r'^(\.LBB\d+_\d+):', ln)
        if mm:
            blocks.append(cur)
            cur = [mm.group(1), []]
        else:
            t = ln.strip()
            if t and not t.startswith((';', '.', '//')):
                cur[1].append(t)
    blocks.append(cur)
    print(name, 'blocks', len(blocks), 'instructions', sum(len(b) for _, b in blocks))
    for lab, b in blocks:
        c = lambda k: sum(k in x for x in b)
        if len(b) >= 40 or c('s_barrier'):
            print(f'  {lab:12s} n={len(b):4d} fma64={c("v_fma_f64"):3d} mul/add64={c("v_mul_f64")+c("v_add_f64"):3d} ds={c("ds_"):3d} '
                  f'dpp={c("dpp"):3d} cnd={c("v_cndmask"):3d} barrier={c("s_barrier")} mfma={c("v_mfma")} scratch={c("scratch_")} '
                  f'branch_to={[x.split()[-1] for x in b if x.startswith(("s_cbranch", "s_branch"))][:3]}')
