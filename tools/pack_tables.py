#!/usr/bin/env python3
"""Pack the F-16 aerodynamic lookup DATA into one generated C include.

Runs in the build container only (needs /root/reference).  It reads

  * the 5 breakpoint vectors + 43 hifi tables under <ref>/C/*.dat
    (file <-> table function <-> grid map: C/hifi_F16_AeroData.c:7-105,109-1861,
    SURVEY.md Appendix A), and
  * the numeric initialisers of the Stevens-Lewis lofi tables
    (C/lofi_F16_AeroData.c:17-26, 66-104, 192-206, 271-283, 343-344)

and writes `f16_mpc_oop_py_amd/csrc/f16_tables_data.inc`: plain integer arrays.
Every hifi value is an exact multiple of 1e-5 and every lofi value of 1e-3, and
`int / scale` (one correctly rounded IEEE division) reproduces the reference's
`fscanf("%lf")` / C literal bit for bit -- this script asserts that for every
number it packs.  Only numbers are emitted; no reference code is copied.
"""
import argparse
import os
import re
import sys

import numpy as np

# (table id, .dat file, grid axes)  -- order = table id used by kernels + oracle
A1, A2, B1, D1, D2 = "ALPHA1", "ALPHA2", "BETA1", "DH1", "DH2"
BREAKPOINTS = [(A1, 20), (A2, 14), (B1, 19), (D1, 5), (D2, 3)]
HIFI = [
    # 3-D on ALPHA1 x BETA1 x DH1
    ("Cx", "CX0120_ALPHA1_BETA1_DH1_201", (A1, B1, D1)),
    ("Cz", "CZ0120_ALPHA1_BETA1_DH1_301", (A1, B1, D1)),
    ("Cm", "CM0120_ALPHA1_BETA1_DH1_101", (A1, B1, D1)),
    # 3-D on ALPHA1 x BETA1 x DH2
    ("Cn", "CN0120_ALPHA1_BETA1_DH2_501", (A1, B1, D2)),
    ("Cl", "CL0120_ALPHA1_BETA1_DH2_601", (A1, B1, D2)),
    # 2-D on ALPHA1 x BETA1
    ("Cy", "CY0320_ALPHA1_BETA1_401", (A1, B1)),
    ("Cy_r30", "CY0720_ALPHA1_BETA1_405", (A1, B1)),
    ("Cn_r30", "CN0720_ALPHA1_BETA1_503", (A1, B1)),
    ("Cl_r30", "CL0720_ALPHA1_BETA1_603", (A1, B1)),
    ("Cy_a20", "CY0620_ALPHA1_BETA1_403", (A1, B1)),
    ("Cn_a20", "CN0620_ALPHA1_BETA1_504", (A1, B1)),
    ("Cl_a20", "CL0620_ALPHA1_BETA1_604", (A1, B1)),
    # 2-D on ALPHA2 x BETA1
    ("Cx_lef", "CX0820_ALPHA2_BETA1_202", (A2, B1)),
    ("Cz_lef", "CZ0820_ALPHA2_BETA1_302", (A2, B1)),
    ("Cm_lef", "CM0820_ALPHA2_BETA1_102", (A2, B1)),
    ("Cy_lef", "CY0820_ALPHA2_BETA1_402", (A2, B1)),
    ("Cn_lef", "CN0820_ALPHA2_BETA1_502", (A2, B1)),
    ("Cl_lef", "CL0820_ALPHA2_BETA1_602", (A2, B1)),
    ("Cy_a20_lef", "CY0920_ALPHA2_BETA1_404", (A2, B1)),
    ("Cn_a20_lef", "CN0920_ALPHA2_BETA1_505", (A2, B1)),
    ("Cl_a20_lef", "CL0920_ALPHA2_BETA1_605", (A2, B1)),
    # 1-D on ALPHA1 (order of hifi_damping's retVal, then hifi_other_coeffs)
    ("CXq", "CX1120_ALPHA1_204", (A1,)),
    ("CYr", "CY1320_ALPHA1_406", (A1,)),
    ("CYp", "CY1220_ALPHA1_408", (A1,)),
    ("CZq", "CZ1120_ALPHA1_304", (A1,)),
    ("CLr", "CL1320_ALPHA1_606", (A1,)),
    ("CLp", "CL1220_ALPHA1_608", (A1,)),
    ("CMq", "CM1120_ALPHA1_104", (A1,)),
    ("CNr", "CN1320_ALPHA1_506", (A1,)),
    ("CNp", "CN1220_ALPHA1_508", (A1,)),
    ("dCNbeta", "CN9999_ALPHA1_brett", (A1,)),
    ("dCLbeta", "CL9999_ALPHA1_brett", (A1,)),
    ("dCm", "CM9999_ALPHA1_brett", (A1,)),
    # 1-D on ALPHA2 (order of hifi_damping_lef's retVal)
    ("dCXq_lef", "CX1420_ALPHA2_205", (A2,)),
    ("dCYr_lef", "CY1620_ALPHA2_407", (A2,)),
    ("dCYp_lef", "CY1520_ALPHA2_409", (A2,)),
    ("dCZq_lef", "CZ1420_ALPHA2_305", (A2,)),
    ("dCLr_lef", "CL1620_ALPHA2_607", (A2,)),
    ("dCLp_lef", "CL1520_ALPHA2_609", (A2,)),
    ("dCMq_lef", "CM1420_ALPHA2_105", (A2,)),
    ("dCNr_lef", "CN1620_ALPHA2_507", (A2,)),
    ("dCNp_lef", "CN1520_ALPHA2_509", (A2,)),
    # 1-D on DH1
    ("eta_el", "ETA_DH1_brett", (D1,)),
]
# lofi arrays: (id, C array name, occurrence index of that name, rows, cols)
LOFI = [
    ("f16_lofi_damp", "A", 0, 9, 12),
    ("f16_lofi_dlda", "ALA", 0, 7, 12),
    ("f16_lofi_dldr", "ALR", 0, 7, 12),
    ("f16_lofi_dnda", "ANA", 0, 7, 12),
    ("f16_lofi_dndr", "ANR", 0, 7, 12),
    ("f16_lofi_cl", "AL", 0, 7, 12),
    ("f16_lofi_cn", "AN", 0, 7, 12),
    ("f16_lofi_cx", "AX", 0, 5, 12),
    ("f16_lofi_cm", "AM", 0, 5, 12),
    ("f16_lofi_cz", "A", 1, 1, 12),
]
HIFI_SCALE = 100000
LOFI_SCALE = 1000


def to_scaled_ints(tokens, scale, what):
    vals = np.array([float(t) for t in tokens], dtype=np.float64)
    ints = np.rint(vals * scale).astype(np.int64)
    back = ints.astype(np.float64) / float(scale)
    bad = np.nonzero(back != vals)[0]  # (-0.0 == +0.0: a signed zero is the only bit pattern allowed to differ)
    if bad.size:
        raise SystemExit(f"{what}: {bad.size} values are not exact int/{scale} (first idx {bad[0]}: {tokens[bad[0]]})")
    assert np.abs(ints).max() < 2 ** 31
    return ints


def c_array(name, ints, per_line=12):
    lines = [f"static const int32_t {name}[{len(ints)}] = {{"]
    for i in range(0, len(ints), per_line):
        lines.append("  " + ", ".join(str(int(v)) for v in ints[i:i + per_line]) + ",")
    lines.append("};")
    return "\n".join(lines)


def parse_lofi(src):
    """Pull the numeric initialiser of every `double NAME[..] = {...};` in order."""
    out = {}
    for m in re.finditer(r"double\s+(\w+)\s*(?:\[\d*\])+\s*=\s*\{(.*?)\};", src, re.S):
        nums = re.findall(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)", m.group(2))
        out.setdefault(m.group(1), []).append(nums)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..",
                                                  "f16_mpc_oop_py_amd", "csrc", "f16_tables_data.inc"))
    args = ap.parse_args()
    cdir = os.path.join(args.ref, "C")
    npts = dict(BREAKPOINTS)

    chunks = ["/* GENERATED by tools/pack_tables.py -- DATA ONLY (scaled integers).",
              " * hifi value = (double)int / 1e5, lofi value = (double)int / 1e3 (IEEE division;",
              " * asserted at pack time to equal strtod() of the reference text bit for bit).",
              " * hifi element order inside a table = reference file order: alpha fastest, then beta, then el. */",
              "#include <stdint.h>",
              f"#define F16_HIFI_SCALE {HIFI_SCALE}.0",
              f"#define F16_LOFI_SCALE {LOFI_SCALE}.0",
              f"#define F16_N_HIFI_TABLES {len(HIFI)}"]
    for name, n in BREAKPOINTS:
        toks = open(os.path.join(cdir, name + ".dat")).read().split()
        assert len(toks) == n, (name, len(toks))
        chunks.append(c_array(f"f16_bp_{name.lower()}", to_scaled_ints(toks, HIFI_SCALE, name)))

    total = 0
    offsets = []
    for tid, (tname, fname, axes) in enumerate(HIFI):
        toks = open(os.path.join(cdir, fname + ".dat")).read().split()
        want = int(np.prod([npts[a] for a in axes]))
        assert len(toks) == want, (tname, len(toks), want)
        chunks.append(f"/* table {tid}: {tname}  <- C/{fname}.dat  grid {' x '.join(axes)} */")
        chunks.append(c_array(f"f16_tab_{tname}", to_scaled_ints(toks, HIFI_SCALE, tname)))
        offsets.append(total)
        total += want
    assert total == 13405, total
    chunks.append("enum f16_table_id {")
    chunks += [f"  F16_T_{t[0]} = {i}," for i, t in enumerate(HIFI)]
    chunks.append("};")
    chunks.append(f"static const int32_t *const f16_hifi_tables[{len(HIFI)}] = {{")
    chunks += [f"  f16_tab_{t[0]}," for t in HIFI]
    chunks.append("};")
    chunks.append(f"static const int f16_hifi_sizes[{len(HIFI)}] = {{"
                  + ", ".join(str(int(np.prod([npts[a] for a in t[2]]))) for t in HIFI) + "};")

    lofi = parse_lofi(open(os.path.join(cdir, "lofi_F16_AeroData.c")).read())
    for lid, cname, occ, rows, cols in LOFI:
        toks = lofi[cname][occ]
        assert len(toks) == rows * cols, (lid, len(toks))
        chunks.append(f"/* {lid}: [{rows}][{cols}] (Stevens & Lewis table) */")
        chunks.append(c_array(lid, to_scaled_ints(toks, LOFI_SCALE, lid)))

    out = os.path.abspath(args.out)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        f.write("\n".join(chunks) + "\n")
    print(f"wrote {out}: {total} hifi values + {sum(r * c for *_, r, c in LOFI)} lofi values")


if __name__ == "__main__":
    sys.exit(main())
