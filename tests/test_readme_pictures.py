"""The reference's README pictures as known answers.

/root/reference/screenshots/fsk-{1..5}.png and ook-1.png are terminal pictures of the REAL reference's `sparkfft` output on its
two committed example recordings (README.md:27-97, 113-118).  tests/golden/make_readme_grids.py decoded them into grids of
glyph codes (tests/golden/readme_glyph_grids.npz); the recordings are tests/golden/{fsk-example.sr21M.fc32,
cupboard-superdec.sr400.cf32}.  A picture is a crop: which output row / FFT bin its first cell is, is found by search.

What they pin:
  * fsk-1 (no shift, no lowpass, -width 128, default range): 39 rows x 127 bins = 4953 cells, EVERY one equal — the window
    loop, the 128-point Radix4 FFT (base 8 + two radix-4 layers), hypot and the glyph ladder, on real data;
  * ook-1 (-width 4 -stride 2 -range 0.001:0.01): 79 rows x 4 bins, every one equal (beside the README string at :167);
  * fsk-2..5 (the chains with `lowpass`): the pictures were taken with an EARLIER revision of the reference's filter code — no
    tap count, cutoff or decimation phase reproduces them exactly.  (SURVEY section 4 lists the `uniq -c` excerpt at README.md:135-140
    as stale too; it is not — tests/test_oracle_golden.py::test_readme_ook_run_lengths finds it in today's output.)
    Today's code agrees with them in 94.6-99.5 % of the cells at the best alignment, the tones in the same columns, and EVERY
    differing cell differs by exactly one glyph level (the oracle's norm there lies within 0.52 of a glyph step of a threshold,
    median 0.03-0.2).  Kept as exact counts of that agreement plus a strict xfail on exact equality, so a change of either side shows.
    Round 4: a picture is a terminal screenshot, so what lies beside the 128 glyphs of a line is blank background — the alignment search
    pads the output with blank columns.  That is what fsk-3 needed: its crop starts at output column 2 (its 127th column is background),
    which the unpadded search could not reach; read at column 1 it showed 528 differing cells of up to four levels (the shifted DC bar,
    one bin off), read where it belongs 241, all of one level — the same residue as the other three.
"""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (chain, width, stride, range, expected best alignment (row, col), cells that agree there out of all)
PICTURES = {
    "fsk_1": (dict(), 128, 128, None, (1484, 0), (4953, 4953)),
    "fsk_2": (dict(lowpass=(2_000_000, 16, 40)), 128, 128, None, (44, 0), (6429, 6477)),
    "fsk_3": (dict(shift=280000, lowpass=(2_000_000, 16, 40)), 128, 128, None, (38, 2), (4204, 4445)),
    "fsk_4": (dict(shift=280000, lowpass=(200_000, 16, 400)), 128, 128, None, (32, 1), (3887, 3937)),
    "fsk_5": (dict(shift=280000, lowpass=(200_000, 32, 400)), 64, 16, None, (29, 0), (2757, 2772)),
    "ook_1": (dict(), 4, 2, (0.001, 0.01), (343, 0), (316, 316)),
}


@pytest.fixture(scope="module")
def grids():
    return np.load(os.path.join(GOLDEN, "readme_glyph_grids.npz"))


@pytest.fixture(scope="module")
def recordings():
    return {"fsk": open(os.path.join(GOLDEN, "fsk-example.sr21M.fc32"), "rb").read(),
            "cup": open(os.path.join(GOLDEN, "cupboard-superdec.sr400.cf32"), "rb").read()}


def oracle_codes(oracle, recordings, name):
    chain, W, S, rng, _, _ = PICTURES[name]
    data, sr = (recordings["cup"], 400) if name == "ook_1" else (recordings["fsk"], 21_000_000)
    ch = oracle.Chain.from_bytes(data, oracle.FMT_CF32, sr)
    if "shift" in chain:
        ch = ch.shift(chain["shift"])
    if "lowpass" in chain:
        ch = ch.lowpass(*chain["lowpass"])
    return ch.spark_fft(W, S, rng) if rng else ch.spark_fft(W, S)


PAD = 3      # blank terminal columns either side of a line of glyphs


def padded(codes):
    return np.pad(codes, ((0, 0), (PAD, PAD)))


def best_alignment(codes, grid):
    """(cells agreeing, row, column) of the best placement of the picture over the output, the output padded with PAD blank columns
    either side; the column is reported in output coordinates (negative: the crop starts in the background)."""
    R, C = grid.shape
    codes = padded(codes)
    m, r0, c0 = _best(codes, grid)
    return m, r0, c0 - PAD


def _best(codes, grid):
    R, C = grid.shape
    best = (-1, 0, 0)
    for c0 in range(codes.shape[1] - C + 1):
        sub = codes[:, c0:c0 + C]
        for r0 in range(codes.shape[0] - R + 1):
            m = int((sub[r0:r0 + R] == grid).sum())
            if m > best[0]:
                best = (m, r0, c0)
    return best


@pytest.mark.parametrize("name", ["fsk_1", "ook_1"])
def test_oracle_reproduces_the_picture_exactly(oracle, grids, recordings, name):
    norms, codes = oracle_codes(oracle, recordings, name)
    g = grids[name]
    m, r0, c0 = best_alignment(codes, g)
    assert (m, (r0, c0)) == (g.size, PICTURES[name][4]), (name, m, g.size, r0, c0)
    assert (g != 0).sum() >= 59                                   # the picture is not blank: 59+ lit cells to get right
    # ... and nowhere else in the output (the alignment is unique, so this is a real known answer)
    R, C = g.shape
    cp = padded(codes)
    hits = [(r, c - PAD) for c in range(cp.shape[1] - C + 1) for r in range(cp.shape[0] - R + 1) if np.array_equal(cp[r:r + R, c:c + C], g)]
    assert hits == [PICTURES[name][4]]


@pytest.mark.parametrize("name", ["fsk_2", "fsk_3", "fsk_4", "fsk_5"])
def test_lowpass_pictures_agree_as_far_as_todays_reference_does(oracle, grids, recordings, name):
    """Pictures taken with an earlier filter revision (module docstring): the agreement of TODAY's algorithm is pinned."""
    norms, codes = oracle_codes(oracle, recordings, name)
    g = grids[name]
    m, r0, c0 = best_alignment(codes, g)
    want_m, total = PICTURES[name][5]
    assert total == g.size and (r0, c0) == PICTURES[name][4]
    assert m == want_m, (name, m, want_m)                         # exactly as many cells as the restatement of today's code gives
    # ... and every differing cell by ONE glyph level, its norm within 0.55 of a glyph step of a threshold
    cp, npad = padded(codes), np.pad(norms, ((0, 0), (PAD, PAD)))
    sub = cp[r0:r0 + g.shape[0], c0 + PAD:c0 + PAD + g.shape[1]].astype(int)
    nn = npad[r0:r0 + g.shape[0], c0 + PAD:c0 + PAD + g.shape[1]]
    d = sub - g.astype(int)
    assert set(np.unique(d).tolist()) <= {-1, 0, 1}, (name, np.unique(d))
    step = (1.0 - 0.08) / 7
    dist = np.abs(nn[d != 0][:, None] - (0.08 + step * np.arange(8))[None, :]).min(axis=1) / step
    assert dist.max() <= 0.55, (name, float(dist.max()))
    # the tones sit in the same columns: compare the column histogram of lit cells
    lit_pic, lit_now = (g != 0).sum(axis=0), (sub != 0).sum(axis=0)
    top = lambda v: set(np.argsort(v)[-2:].tolist())
    assert len(top(lit_pic) & top(lit_now)) >= 1, (top(lit_pic), top(lit_now))


@pytest.mark.xfail(strict=True, reason="README pictures fsk-2..5 predate today's lowpass code: 48 / 241 / 50 / 15 cells differ, each by one glyph level")
@pytest.mark.parametrize("name", ["fsk_2", "fsk_3", "fsk_4", "fsk_5"])
def test_lowpass_pictures_exactly(oracle, grids, recordings, name):
    norms, codes = oracle_codes(oracle, recordings, name)
    g = grids[name]
    m, r0, c0 = best_alignment(codes, g)
    assert m == g.size


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PICTURES))
def test_gpu_glyphs_against_the_pictures(engine, oracle, grids, recordings, name):
    """The HIP chain with the glyph epilogue on the same recordings: every cell equals the picture for fsk-1 / ook-1, and equals
    the oracle (edge-aware) for all six chains — so the GPU stands exactly where today's reference stands on fsk-2..5."""
    from test_gpu_parity import assert_codes_edge_aware
    chain, W, S, rng, (r0, c0), _ = PICTURES[name]
    data, sr = (recordings["cup"], 400) if name == "ook_1" else (recordings["fsk"], 21_000_000)
    p = engine.Plan(engine.FMT_CF32, sr, len(data) // 8, shift_hz=chain.get("shift"), lowpass=chain.get("lowpass"), width=W, stride=S,
                    epilogue=engine.EPI_GLYPH_U8, rng=rng)
    got = p.run_host(data)
    norms, codes = oracle_codes(oracle, recordings, name)
    assert got.shape == codes.shape
    rmin, rmax = rng if rng else (0.08, 1.0)
    assert_codes_edge_aware(codes, got, norms, rmin, rmax, f"README picture chain {name}")
    g = grids[name]
    if name in ("fsk_1", "ook_1"):
        assert np.array_equal(got[r0:r0 + g.shape[0], c0:c0 + g.shape[1]], g)
