"""CPU checks of the one-wavefront-per-aircraft ADMM solver's layouts (csrc/f16_mpc_wave.hip): the lane-level emulation of its
data movement (symmetric block image of the KKT inverse with direct + transposed use and ds_bpermute pulls, octet layout of the
block-Toeplitz stages with the recursive-halving reductions, scatter table from the matrix-core tile image) against dense
numpy, and that the committed index tables are what the generator writes."""
import importlib.util
import os

import numpy as np

from conftest import REPO


def _tables():
    spec = importlib.util.spec_from_file_location("wave_tables", os.path.join(REPO, "tools", "wave_tables.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_lane_level_emulation_of_the_wavefront_solver_layouts():
    _tables().selfcheck()


def test_committed_index_tables_are_current(tmp_path):
    wt = _tables()
    out = tmp_path / "t.inc"
    wt.write_inc(str(out))
    assert out.read_text() == open(os.path.join(REPO, "f16_mpc_oop_py_amd", "csrc", "f16_wave_tables.inc")).read()


def test_every_unordered_pair_of_block_rows_is_stored_once():
    wt = _tables()
    seen = {}
    for lane in range(wt.NL):
        r, s, cA, cB = wt.lane_blocks(lane)
        for c in (cA, cB):
            key = (min(r, c), max(r, c))
            assert key not in seen, (key, lane, seen[key])
            seen[key] = lane
    assert len(seen) == wt.NB * (wt.NB + 1) // 2
    ks = wt.build_kscat()
    used = ks[ks != 0xFFFF]
    assert len(np.unique(used)) == len(used) == 36 * wt.NL * 2          # every slot of the image written exactly once
