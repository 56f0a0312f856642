"""Per basic block of a device function: instruction counts by class (compiler ISA from `hipcc -S --cuda-device-only`).
usage: python tools/wave_isa_blocks.py file.s function-substring [min-instructions]"""
import re, sys
txt = open(sys.argv[1]).read()
sub = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n', txt):
    if sub not in m.group(1):
        continue
    body = txt[m.end():]
    body = body[:body.find('.Lfunc_end')]
    print(m.group(1)[:100])
    for b in re.split(r'\n(?=\.LBB\d+_\d+:)', body):
        hdr = b.split('\n')[0]
        lines = [l.strip() for l in b.split('\n')[1:] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        if len(lines) < mn:
            continue
        c = lambda pat: sum(1 for l in lines if re.search(pat, l))
        print(hdr[:80].ljust(80), 'n', len(lines), 'f64', c(r'v_(fma|fmac|mul|add|max|min)_f64'), 'acc', c('v_accvgpr'), 'scr', c('scratch_'),
              'ds', c(r'^ds_'), 'cnd', c('v_cndmask'), 'dpp', c('dpp'), 'mfma', c('v_mfma'), 'wait', c('s_waitcnt'))
