"""Per-class instruction table of a kernel's loops (compiler ISA, `hipcc -S --cuda-device-only`, product flags).
usage: python tools/isa_classes.py <file.hip> <kernel-substring> [min-depth] [extra flags ...]
Classes: fp64 math | conversions | selects | other VALU (integer, address, compares, moves, DPP) | SALU | LDS | VMEM | waitcnt/nop |
branches.  Blocks are listed in program order with their loop depth; the totals line sums the blocks at depth >= min-depth."""
import re, subprocess, sys
src, sub = sys.argv[1], sys.argv[2]
mind = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-DF16_FAST_TAN", "-DF16_FAST_POW",
       "-DF16_FAST_TRIG", "-DF16_FAST_DIV", "-S", "--cuda-device-only", "-o", "/tmp/isa_classes.s", src] + sys.argv[4:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
txt = open("/tmp/isa_classes.s").read()
CL = ["fp64", "cvt", "select", "valu_other", "salu", "lds", "vmem", "wait", "branch"]


def classify(l):
    op = l.split()[0]
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")): return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_setpc", "s_swappc")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_cndmask"): return "select"
    if re.match(r"v_(fma|fmac|mul|add|max|min|rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|trig_preop|ldexp|frexp|fract|floor|ceil|rndne|trunc)\w*_f64", op): return "fp64"
    return "valu_other"


for m in re.finditer(r'\n(_Z\w+):[^\n]*\n', txt):
    if sub not in m.group(1) or m.group(1).endswith(".kd"):
        continue
    body = txt[m.end():]; body = body[:body.find('.Lfunc_end')]
    print(m.group(1)[:110])
    tot = dict.fromkeys(CL, 0)
    for b in re.split(r'\n(?=\.LBB\d+_\d+:)', body):
        d = re.search(r'Depth=(\d+)', b[:400]); d = int(d.group(1)) if d else 0
        lines = [l.strip() for l in b.split('\n')[1:] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        c = dict.fromkeys(CL, 0)
        for l in lines: c[classify(l)] += 1
        if d >= mind:
            for k in CL: tot[k] += c[k]
            if len(lines) >= 25:
                print(f"  {b.split(':')[0][:12]:12s} depth {d} n {len(lines):5d} | " + " ".join(f"{k} {c[k]}" for k in CL))
    n = sum(tot.values())
    print(f"  TOTAL depth >= {mind}: {n} | " + " ".join(f"{k} {tot[k]} ({100 * tot[k] / max(n, 1):.0f} %)" for k in CL))
