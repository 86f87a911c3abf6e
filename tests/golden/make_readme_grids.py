"""Decodes the reference's README screenshots (terminal pictures of `sparkfft` output, /root/reference/screenshots/
fsk-{1..5}.png and ook-1.png, README.md:27-97,113-118) into grids of glyph codes — 0 blank, 1..8 = '▁'..'█' — and writes
tests/golden/readme_glyph_grids.npz.  These are OUTPUTS OF THE REAL REFERENCE on its two committed example recordings, i.e.
externally authored known answers for the whole chain (shift, lowpass, FFT, hypot, glyph ladder), unlike the oracle's own
vectors.  Run in the build container (needs /root/reference and PIL); the .npz is the committed fixture.

A terminal cell is `cw` x `pitch` pixels; a glyph of code k is a bar of ceil(k * pitch / 8) pixel rows at the bottom of its
cell (measured: the only heights that occur are 2,3,5,6,8,12 at pitch 12; 2,3,4,6,9 at pitch 9; 3 at pitch 19), so
k = floor(8 h / pitch).  The pictures are crops: the first complete text row and the first complete cell column are found from
the pixel phase of the bars; WHICH output row / FFT bin they are is not in the picture — the test searches for it."""
import os
import sys

import numpy as np
from PIL import Image

SHOTS = "/root/reference/screenshots"
HERE = os.path.dirname(os.path.abspath(__file__))
# name: (cell width, row pitch, has the │ border columns)
SPEC = {"fsk-1": (5, 12, False), "fsk-2": (5, 9, False), "fsk-3": (5, 12, False), "fsk-4": (5, 12, False), "fsk-5": (10, 19, False),
        "ook-1": (5, 9, True)}


def decode(name):
    cw, pitch, border = SPEC[name]
    a = np.array(Image.open(os.path.join(SHOTS, name + ".png")).convert("L")) > 127
    H, W = a.shape
    if border:                                   # vertical frame lines are 1 px wide: drop them, cells are aligned at x = 0
        line_cols = [x for x in range(W) if a[:, x].mean() > 0.9]
        a = a.copy()
        a[:, line_cols] = False
        x0 = 0
    else:                                        # phase of the bars' left edges
        starts = np.zeros(cw, dtype=int)
        for y in range(H):
            row = a[y]
            edge = np.flatnonzero(row & ~np.r_[False, row[:-1]])
            for x in edge:
                starts[x % cw] += 1
        x0 = int(starts.argmax())
    # bar bottoms: pixel rows where a white run ends
    bottoms = [y for y in range(H - 1) if (a[y] & ~a[y + 1]).any()] + ([H - 1] if a[H - 1].any() else [])
    phase = np.bincount(np.array(bottoms) % pitch, minlength=pitch).argmax()
    first = int(phase) if phase >= pitch - 1 else int(phase)          # first bottom row with a whole cell above it
    while first - (pitch - 1) < 0:
        first += pitch
    n_rows = (H - 1 - first) // pitch + 1
    n_cols = (W - x0) // cw
    grid = np.zeros((n_rows, n_cols), dtype=np.uint8)
    for r in range(n_rows):
        yb = first + r * pitch
        for c in range(n_cols):
            cell = a[yb - pitch + 1:yb + 1, x0 + c * cw:x0 + (c + 1) * cw]
            rows = cell.any(axis=1)
            h = 0
            for y in range(pitch - 1, -1, -1):
                if rows[y]:
                    h += 1
                else:
                    break
            assert not rows[:pitch - h].any(), (name, r, c, "a bar that does not sit on the cell's bottom")
            grid[r, c] = (8 * h) // pitch
    if border:
        grid = grid[:, 1:-1]                     # the frame's own cells
    return grid


def main():
    out = {}
    for name in SPEC:
        g = decode(name)
        out[name.replace("-", "_")] = g
        print(name, g.shape, "codes", dict(zip(*np.unique(g, return_counts=True))))
    np.savez_compressed(os.path.join(HERE, "readme_glyph_grids.npz"), **out)


if __name__ == "__main__":
    main()
