#!/usr/bin/env python3
"""Static LDS bank-conflict count of ONE ADMM iteration of k_mpc_wave (csrc/f16_mpc_wave.hip), by the banking rules of
MI355X_MICROARCH.md (LDS): reads of 8 / 16 bytes bank on (a / 4) mod 64, everything else on mod 32; lane groups per instruction;
identical addresses broadcast; each extra distinct address on a busy bank of a group adds one LDS cycle.  Prints, per access of the
iteration, the conflict-free cycles and the extra cycles, so that a layout change can be costed before it is built.

   python tools/wave_lds_conflicts.py [--p1rec 22] [--p3rec 30] [--mvxrec 14]
"""
import argparse
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("--p1rec", type=int, default=22)
ap.add_argument("--p3rec", type=int, default=30)
ap.add_argument("--mvxrec", type=int, default=14)
ap.add_argument("--ws6", type=int, default=6, help="doubles per step of the state-row vector")
ap.add_argument("--wspad", type=int, default=8, help="extra doubles per 5-step block of the state-row vector (0: round 4)")
ap.add_argument("--clamp", type=int, default=24, help="idle lanes 60..63 read the B blocks of lane l - clamp (0: of lane 59, round 4)")
ap.add_argument("-N", type=int, default=30)
args = ap.parse_args()
N = args.N
NB, NL, TB = 15, 60, 5
KI_SIZE = 36 * NL * 2
WS_OFF, WS_SIZE = KI_SIZE, 444
WC_OFF = WS_OFF + WS_SIZE
WR_OFF = WC_OFF + 92
XT_OFF = WR_OFF + 96
XPADS = 7
P3_OFF, P3_REC = 0, args.p3rec
P1_OFF, P1_REC = P3_OFF + 22 * P3_REC, args.p1rec
ZP_OFF = P1_OFF + 64 * P1_REC
MVX_OFF, MVX_REC = 0, args.mvxrec
assert ZP_OFF + 8 <= 18 * NL * 2, "records overrun the B blocks"

G128R = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128R = G128R + [[l + 32 for l in g] for g in G128R]
G32x2 = [list(range(0, 32)), list(range(32, 64))]
G16x4 = [list(range(16 * g, 16 * g + 16)) for g in range(4)]
G8x8 = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def role(l):
    o, t = l >> 3, l & 7
    b2, par, estar = (t >> 2) & 1, t & 1, t >> 1
    h = par ^ b2
    istep = 4 * o + estar
    r, s = l >> 2, l & 3
    rr = min(r, NB - 1)
    cA = (rr + NB - (2 * s + 1)) % NB
    cB = (rr + NB - (2 * s + 2)) % NB if s < 3 else rr
    return dict(l=l, o=o, t=t, par=par, h=h, istep=istep, act=istep < N, r=r, s=s, rr=rr, cA=cA, cB=cB)


def tjob(l):
    live = l < 63
    ll = l if live else 0
    s = ll // 21
    j = ll - 21 * s
    I = 5 if j >= 15 else 4 if j >= 10 else 3 if j >= 6 else 2 if j >= 3 else 1 if j >= 1 else 0
    T = j - I * (I + 1) // 2
    return dict(s=s, I=I, T=T, D=I - T, rec3=j if live else 21)


R = [role(l) for l in range(64)]
J = [tjob(l) for l in range(64)]


def cost(addr, width, kind):
    """addr: {lane: double index} of the active lanes; width in bytes (8 / 16); kind 'r' / 'w' -> (base cycles, extra cycles)"""
    if kind == "r":
        mod, groups = 64, (G128R if width == 16 else G32x2)
    else:
        mod, groups = 32, (G8x8 if width == 16 else G16x4)
    base = extra = 0
    for g in groups:
        per_bank = defaultdict(set)
        for l in g:
            if l in addr:
                d = addr[l] * 2                          # dword address
                for k in range(width // 4):
                    per_bank[(d + k) % mod].add(d)       # distinct addresses on this bank
        if per_bank:
            base += 1
            extra += max(len(v) for v in per_bank.values()) - 1
    return base, extra


rows = []


def acc(name, width, kind, fn, n=1):
    b = e = 0
    for i in range(n):
        a = {}
        for l in range(64):
            v = fn(l, i)
            if v is not None:
                a[l] = v
        bb, ee = cost(a, width, kind)
        b += bb; e += ee
    rows.append((name, kind, width, n, b, e))


def a1(l, k):
    r = R[l]
    inn = r["istep"] < 6 * TB
    i = r["istep"] if inn else 0
    I, e = i // TB, i % TB
    T, sp = 3 * r["par"] + k // 3, k % 3
    II = I + T
    return P1_OFF + (21 * sp + II * (II + 1) // 2 + T) * P1_REC + 4 * e if (inn and II <= 5) else ZP_OFF


def a3(l, T):
    r = R[l]
    inn = r["istep"] < 6 * TB
    i = r["istep"] if inn else 0
    I, e = i // TB, i % TB
    return P3_OFF + (I * (I + 1) // 2 + T) * P3_REC + 6 * e + 3 * r["h"] if (inn and T <= I) else ZP_OFF


x4 = lambda step: XT_OFF + 4 * (step + XPADS)
WS = args.ws6
wsi = lambda step: WS * step + args.wspad * (step // TB)
acc("stage1_partials: state-row vector (9 x b128)", 16, "r", lambda l, m: WS_OFF + wsi(TB * J[l]["I"] + m) + 2 * J[l]["s"], 9)
acc("stage1_partials: P1 record b128", 16, "w", lambda l, e: P1_OFF + l * P1_REC + 4 * e, 5)
acc("stage1_partials: P1 record b64", 8, "w", lambda l, e: P1_OFF + l * P1_REC + 4 * e + 2, 5)
acc("stage1_totals: records b128", 16, "r", a1, 9)
acc("stage1_totals: records b64", 8, "r", lambda l, k: a1(l, k) + 2, 9)
kxa = lambda l: 3 * R[l]["istep"] if R[l]["act"] else 0
acc("rhs: wc / wr / wr+3 (9 x b64)", 8, "r", lambda l, i: (WC_OFF + kxa(l) + i if i < 3 else WR_OFF + kxa(l) + (i - 3)), 9)
acc("rhs: write (3 x b64)", 8, "w", lambda l, c: WC_OFF + 3 * R[l]["istep"] + c if (R[l]["act"] and R[l]["par"] == 0) else None, 3)
acc("kkt_matvec: B blocks (18 x b128)", 16, "r", lambda l, m: 2 * ((18 + m) * NL + (l if l < NL else (l - args.clamp if args.clamp else NL - 1))), 18)
acc("kkt_matvec: rhs rows r / cA / cB (9 x b128)", 16, "r", lambda l, i: WC_OFF + 6 * (R[l]["rr"], R[l]["cA"], R[l]["cB"])[i // 3] + 2 * (i % 3), 9)
acc("kkt_matvec: hand-over write (6 x b128)", 16, "w", lambda l, m: MVX_OFF + l * MVX_REC + 2 * m, 6)
srcA = lambda l: 4 * ((R[l]["rr"] + 2 * R[l]["s"] + 1) % NB) + R[l]["s"]
srcB = lambda l: 4 * ((R[l]["rr"] + 2 * R[l]["s"] + 2) % NB) + R[l]["s"]
acc("kkt_matvec: hand-over read A (3 x b128)", 16, "r", lambda l, m: MVX_OFF + srcA(l) * MVX_REC + 2 * m, 3)
acc("kkt_matvec: hand-over read B (3 x b128)", 16, "r", lambda l, m: MVX_OFF + srcB(l) * MVX_REC + 6 + 2 * m, 3)
xw = lambda l: R[l]["s"] == 0 and R[l]["r"] < NB
acc("x~ write b128 (2)", 16, "w", lambda l, i: x4(2 * R[l]["r"]) + 4 * i if xw(l) else None, 2)
acc("x~ write b64 (2)", 8, "w", lambda l, i: x4(2 * R[l]["r"]) + 4 * i + 2 if xw(l) else None, 2)
acc("stage3_partials: x~ b128 (9)", 16, "r", lambda l, m: XT_OFF + 4 * (TB * J[l]["D"] - 4 + XPADS) + 4 * m, 9)
acc("stage3_partials: x~ b64 (9)", 8, "r", lambda l, m: XT_OFF + 4 * (TB * J[l]["D"] - 4 + XPADS) + 4 * m + 2, 9)
acc("stage3_partials: P3 record (5 x b128)", 16, "w", lambda l, e: P3_OFF + J[l]["rec3"] * P3_REC + 2 * J[l]["s"] + 6 * e, 5)
acc("stage3_totals: records (18 x b64)", 8, "r", lambda l, i: a3(l, i // 3) + i % 3, 18)
xk = lambda l: x4(R[l]["istep"] if R[l]["act"] else 0)
acc("x~ of the own step b128 (2)", 16, "r", lambda l, i: xk(l) - 4 * i, 2)
acc("x~ of the own step b64 (2)", 8, "r", lambda l, i: xk(l) - 4 * i + 2, 2)
acc("w: state rows (3 x b64)", 8, "w", lambda l, c: WS_OFF + wsi(R[l]["istep"]) + 3 * R[l]["h"] + c if R[l]["act"] else None, 3)
acc("w: command / rate rows (3 x b64)", 8, "w", lambda l, c: (WR_OFF if R[l]["par"] else WC_OFF) + 3 * R[l]["istep"] + c if R[l]["act"] else None, 3)
tb = te = 0
print("%-52s %2s %3s %3s %6s %6s" % ("access", "", "B", "n", "cycles", "extra"))
for name, kind, width, n, b, e in rows:
    print("%-52s %2s %3d %3d %6d %6d" % (name, kind, width, n, b, e))
    tb += b; te += e
print("one iteration: %d conflict-free LDS cycles + %d conflict cycles = %.3f of the total" % (tb, te, te / (tb + te)))
