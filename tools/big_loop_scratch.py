"""Static check of k_mpc_big: scratch (spill) instructions per loop depth in the compiler's ISA.
usage: python tools/big_loop_scratch.py [extra hipcc flags ...]"""
import re, subprocess, sys
src = "f16_mpc_oop_py_amd/csrc/f16_mpc_big.hip"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-DF16_FAST_TAN", "-DF16_FAST_POW",
       "-DF16_FAST_TRIG", "-DF16_FAST_DIV", "-S", "--cuda-device-only", "-o", "/tmp/big_ls.s", src] + sys.argv[1:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
txt = open("/tmp/big_ls.s").read()
m = re.search(r'\n(_ZN3f163big9k_mpc_big\w+):', txt)
body = txt[m.end():]; body = body[:body.find('.Lfunc_end')]
depth = {}
for b in re.split(r'\n(?=\.LBB\d+_\d+:)', body):
    hdr = b.split('\n')[0]
    d = re.search(r'Depth=(\d+)', b[:400])
    d = int(d.group(1)) if d else 0
    lines = [l.strip() for l in b.split('\n')[1:] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    a = depth.setdefault(d, [0, 0, 0])
    a[0] += len(lines); a[1] += sum(l.startswith('scratch_load') for l in lines); a[2] += sum(l.startswith('scratch_store') for l in lines)
for d in sorted(depth):
    print(f"depth {d}: instructions {depth[d][0]:6d} scratch loads {depth[d][1]:4d} stores {depth[d][2]:4d}")
print(re.search(r'; ScratchSize: \d+', txt[m.end():]).group(0))
