#!/usr/bin/env python3
"""Index tables + a lane-level emulation of the data movement of the one-wavefront-per-aircraft ADMM solver
(csrc/f16_mpc_wave.hip).  Run in the build container (no GPU):

    python tools/wave_tables.py            # checks the emulation against dense numpy, writes csrc/f16_wave_tables.inc

Layouts (one 64-lane wavefront per aircraft, n = 3 N <= 90 variables = 15 block rows of 6 = two horizon steps):

* K^-1 (and P at the termination test) is used as a SYMMETRIC matrix of 15 x 15 blocks (6 x 6), each unordered pair of
  block rows stored once, by the circulant rule: lane (r, s) -- quad r = lane / 4 is block row r, s = lane % 4 -- holds
      block A  B_A = M[6r.., 6cA..],  cA = (r - (2s+1)) mod 15
      block B  B_B = M[6r.., 6cB..],  cB = (r - (2s+2)) mod 15   (s < 3);   the diagonal block (cB = r) for s = 3
  For 15 block rows the offsets k = 1..7 meet every unordered pair exactly once.  A block serves twice: y_r += B x_c
  (direct) and y_c += B' x_r (transposed; fetched by the owner quad with ds_bpermute from lane (c + k, s)).
* the table KSCAT sends element (row 4q + l/16 of tile row w, column l%16 of tile column J) of the lower-triangular tile
  image of the KKT inverse (MFMA accumulator layout of the blocked sweep) to its slot(s) of that block image in LDS.
* the block image in LDS: chunk m (16 bytes = elements 2m, 2m+1 of a lane's 72) at double2 index m * 60 + lane (lanes 0..59).
"""
import os
import sys

import numpy as np

NB = 15                      # block rows (6 variables = two horizon steps each)
NL = 60                      # lanes that hold blocks
NT = 6                       # 16 x 16 tiles per side of the 96 x 96 tile image


def lane_blocks(lane):
    """[(c, transposed_use)] for the two blocks of a lane: (cA, True), (cB, s < 3)."""
    r, s = lane >> 2, lane & 3
    cA = (r - (2 * s + 1)) % NB
    cB = (r - (2 * s + 2)) % NB if s < 3 else r
    return r, s, cA, cB


def kinv_slot(lane, blk, i, j):
    """double index of element [i][j] of block `blk` (0 = A, 1 = B) of `lane` in the LDS block image."""
    e = blk * 36 + i * 6 + j
    return ((e >> 1) * NL + lane) * 2 + (e & 1)


def build_block_owner():
    """(R, C) ordered block coordinates -> (lane, blk, transposed): where M[6R.., 6C..] lives; transposed means the lane
    stores M[6C.., 6R..] (the mirror block)."""
    own = {}
    for lane in range(NL):
        r, s, cA, cB = lane_blocks(lane)
        own[(r, cA)] = (lane, 0, False)
        own[(cA, r)] = (lane, 0, True)
        own[(r, cB)] = (lane, 1, False)
        if cB != r:
            own[(cB, r)] = (lane, 1, True)
    assert len(own) == NB * NB, len(own)
    return own


def build_kscat():
    """KSCAT[tile t = w(w+1)/2 + J][q][lane] -> (dst0, dst1) double indices (0xFFFF = none) for tile element
    (i, j) = (16 w + 4 q + lane / 16, 16 J + lane % 16), stored tiles J <= w; only i >= j is scattered (the mirror goes along)."""
    own = build_block_owner()
    tab = np.full((NT * (NT + 1) // 2, 4, 64, 2), 0xFFFF, dtype=np.uint16)
    for w in range(NT):
        for J in range(w + 1):
            t = w * (w + 1) // 2 + J
            for q in range(4):
                for l in range(64):
                    i, j = 16 * w + 4 * q + (l >> 4), 16 * J + (l & 15)
                    if i < j or i >= 6 * NB:
                        continue
                    R, C = i // 6, j // 6
                    lane, blk, tr = own[(R, C)]
                    ii, jj = (j % 6, i % 6) if tr else (i % 6, j % 6)
                    tab[t, q, l, 0] = kinv_slot(lane, blk, ii, jj)
                    if R == C and i != j:                      # diagonal block: the mirror element too
                        tab[t, q, l, 1] = kinv_slot(lane, blk, jj, ii)
    return tab


def build_pgath():
    """PGATH[e (0..71)][lane] -> (i << 8) | j with i >= j: the matrix element behind element e of the lane's 72."""
    tab = np.zeros((72, 64), dtype=np.uint16)
    for lane in range(64):
        r, s, cA, cB = lane_blocks(min(lane, NL - 1))
        for e in range(72):
            blk, ii, jj = e // 36, (e % 36) // 6, e % 6
            c = cA if blk == 0 else cB
            i, j = 6 * r + ii, 6 * c + jj
            if i < j:
                i, j = j, i
            tab[e, lane] = (i << 8) | j
    return tab


# ------------------------------------------------------------------------------------------------ emulation
def emul_symv(M, x):
    """y = M x by the block image: direct + transposed products, pulls, quad sums (what stage 2 of the kernel does)."""
    n = 6 * NB
    img = np.zeros(36 * NL * 2)
    pg = build_pgath()
    for lane in range(NL):
        for e in range(72):
            i, j = pg[e, lane] >> 8, pg[e, lane] & 255
            img[((e >> 1) * NL + lane) * 2 + (e & 1)] = M[i, j]
    yd = np.zeros((64, 6)); ytA = np.zeros((64, 6)); ytB = np.zeros((64, 6))
    for lane in range(NL):
        r, s, cA, cB = lane_blocks(lane)
        A = np.array([[img[kinv_slot(lane, 0, i, j)] for j in range(6)] for i in range(6)])
        Bk = np.array([[img[kinv_slot(lane, 1, i, j)] for j in range(6)] for i in range(6)])
        xr, xa, xb = x[6 * r:6 * r + 6], x[6 * cA:6 * cA + 6], x[6 * cB:6 * cB + 6]
        yd[lane] = A @ xa + Bk @ xb
        ytA[lane] = A.T @ xr
        ytB[lane] = Bk.T @ xr if s < 3 else 0.0
    y = np.zeros(n)
    for c in range(NB):
        tot = np.zeros(6)
        for s in range(4):
            lane = 4 * c + s
            srcA = 4 * ((c + 2 * s + 1) % NB) + s
            srcB = 4 * ((c + 2 * s + 2) % NB) + s
            tot += yd[lane] + ytA[srcA] + (ytB[srcB] if s < 3 else 0.0)
        y[6 * c:6 * c + 6] = tot
    return y


def emul_scatter(Kinv):
    """Tile image (lower tiles of the 96 x 96 padded matrix) -> block image through KSCAT; returns the image."""
    tab = build_kscat()
    img = np.full(36 * NL * 2, np.nan)
    for w in range(NT):
        for J in range(w + 1):
            t = w * (w + 1) // 2 + J
            for q in range(4):
                for l in range(64):
                    i, j = 16 * w + 4 * q + (l >> 4), 16 * J + (l & 15)
                    v = Kinv[i, j]
                    for d in tab[t, q, l]:
                        if d != 0xFFFF:
                            img[d] = v
    return img


def octet_roles(lane):
    o, t = lane >> 3, lane & 7
    b2, b1, b0 = (t >> 2) & 1, (t >> 1) & 1, t & 1
    return dict(o=o, t=t, estar=t >> 1, h=b0 ^ b2, par=b0, b1=b1, b2=b2)


def emul_stage3(G6, x, N):
    """z[i][rr] = sum_{d <= i} G6[d][rr][:] . x[i-d][:]  (6 kept rows) by the octet layout + the reduction tree; returns
    z [N][6] assembled from the owner lanes."""
    acc = np.zeros((64, 4, 6))
    for lane in range(64):
        R = octet_roles(lane)
        for e in range(4):
            i = 4 * R["o"] + e
            for u in range(4):
                d = 4 * R["t"] + u
                if d < N and i - d >= 0 and i < N:
                    acc[lane, e] += G6[d] @ x[i - d]
    # step 1: rows split with the xor-1 partner (h decides which half a lane keeps)
    v1 = np.zeros((64, 4, 3))
    for lane in range(64):
        R = octet_roles(lane)
        p = lane ^ 1
        v1[lane] = acc[lane, :, 3 * R["h"]:3 * R["h"] + 3] + acc[p, :, 3 * R["h"]:3 * R["h"] + 3]
    # step 2: half-mirror partner (7 - t): keep e in {2 b2, 2 b2 + 1}
    v2 = np.zeros((64, 2, 3))
    for lane in range(64):
        R = octet_roles(lane)
        p = (lane & ~7) | (7 - R["t"])
        assert octet_roles(p)["h"] == R["h"]
        v2[lane] = v1[lane, 2 * R["b2"]:2 * R["b2"] + 2] + v1[p, 2 * R["b2"]:2 * R["b2"] + 2]
    # step 3: xor-2 partner: keep e = 2 b2 + b1
    z = np.zeros((N, 6))
    for lane in range(64):
        R = octet_roles(lane)
        p = lane ^ 2
        assert octet_roles(p)["h"] == R["h"] and octet_roles(p)["b2"] == R["b2"]
        val = v2[lane, R["b1"]] + v2[p, R["b1"]]
        i = 4 * R["o"] + R["estar"]
        if i < N:
            z[i, 3 * R["h"]:3 * R["h"] + 3] = val
    return z


def emul_stage1(G6, w, N):
    """t[j][c] = sum_{d, j+d < N} G6[d][:, c] . w[j+d][:] by the octet layout; reduction tree of 12 values."""
    acc = np.zeros((64, 4, 3))
    for lane in range(64):
        R = octet_roles(lane)
        for e in range(4):
            j = 4 * R["o"] + e
            for u in range(4):
                d = 4 * R["t"] + u
                if d < N and j + d < N:
                    acc[lane, e] += G6[d].T @ w[j + d]
    v1 = np.zeros((64, 2, 3))
    for lane in range(64):
        R = octet_roles(lane)
        p = (lane & ~7) | (7 - R["t"])
        v1[lane] = acc[lane, 2 * R["b2"]:2 * R["b2"] + 2] + acc[p, 2 * R["b2"]:2 * R["b2"] + 2]
    v2 = np.zeros((64, 3))
    for lane in range(64):
        R = octet_roles(lane)
        v2[lane] = v1[lane, R["b1"]] + v1[lane ^ 2, R["b1"]]
    tt = np.zeros((N, 3))
    for lane in range(64):
        R = octet_roles(lane)
        val = v2[lane] + v2[lane ^ 1]
        j = 4 * R["o"] + R["estar"]
        if j < N:
            tt[j] = val
    return tt


def selfcheck():
    rng = np.random.default_rng(0)
    n = 6 * NB
    M = rng.standard_normal((n, n)); M = M + M.T
    x = rng.standard_normal(n)
    assert np.allclose(emul_symv(M, x), M @ x, rtol=0, atol=1e-12), "block-image symmetric product"
    Kp = np.eye(96); Kp[:n, :n] = M
    img = emul_scatter(Kp)
    pg = build_pgath()
    for lane in range(NL):
        for e in range(72):
            i, j = pg[e, lane] >> 8, pg[e, lane] & 255
            assert img[((e >> 1) * NL + lane) * 2 + (e & 1)] == M[i, j], (lane, e)
    for N in (30, 29, 17, 4, 1):
        G6 = rng.standard_normal((N, 6, 3))
        xs = rng.standard_normal((N, 3))
        zref = np.array([sum(G6[d] @ xs[i - d] for d in range(i + 1)) for i in range(N)])
        assert np.allclose(emul_stage3(G6, xs, N), zref, rtol=0, atol=1e-12), ("stage 3", N)
        ws = rng.standard_normal((N, 6))
        tref = np.array([sum(G6[d].T @ ws[j + d] for d in range(N - j)) for j in range(N)])
        assert np.allclose(emul_stage1(G6, ws, N), tref, rtol=0, atol=1e-12), ("stage 1", N)
    print("wave layout emulation: ok")


def write_inc(path):
    ks, pg = build_kscat(), build_pgath()
    ks32 = ks[..., 0].astype(np.uint32) | (ks[..., 1].astype(np.uint32) << 16)
    with open(path, "w") as f:
        f.write("// generated by tools/wave_tables.py -- index tables of the one-wavefront-per-aircraft ADMM solver (do not edit)\n")
        f.write("// F16_WAVE_KSCAT[(tile w(w+1)/2+J) * 4 + q][lane]: slots (double index into the LDS block image, 0xFFFF = none; low half |\n")
        f.write("//   high half << 16) of tile element (16w + 4q + lane/16, 16J + lane%16) of the lower-triangular tile image of the KKT inverse\n")
        f.write("// F16_WAVE_PGATH[e][lane]: (i << 8) | j, i >= j: the matrix element behind element e (0..71) of a lane's two 6x6 blocks\n")
        f.write("__device__ const unsigned F16_WAVE_KSCAT[%d] = {\n" % ks32.size)
        flat = ks32.reshape(-1)
        for k in range(0, flat.size, 16):
            f.write(" " + ",".join(str(int(v)) for v in flat[k:k + 16]) + ",\n")
        f.write("};\n__device__ const unsigned short F16_WAVE_PGATH[%d] = {\n" % pg.size)
        flat = pg.reshape(-1)
        for k in range(0, flat.size, 24):
            f.write(" " + ",".join(str(int(v)) for v in flat[k:k + 24]) + ",\n")
        f.write("};\n")


if __name__ == "__main__":
    selfcheck()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "f16_mpc_oop_py_amd", "csrc", "f16_wave_tables.inc")
    write_inc(out)
    print("wrote", os.path.normpath(out))
