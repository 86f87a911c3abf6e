"""Development: the same search as lds_swizzle_search.py for k_spark2's transform buffer (base butterflies out of registers).
LDS instructions per tile: the base pass's 16-byte piece WRITES (lane (g, xp), column u: run g W + rev4(2 xp + u) base), the radix-4 layers
(reads; writes except in the last layer, which feeds |X| from registers).  Prints cycles for the identity, for the hand-made swizzle the
kernel shipped with, and for the best GF(2) map found.   usage: python scripts/lds_swizzle_search2.py W"""
import itertools, sys
import numpy as np

W = int(sys.argv[1])
logW = W.bit_length() - 1
base = 16 if logW % 2 == 0 else 8
log_base = base.bit_length() - 1
layers = (logW - log_base) // 2
width = W // base
LPW = width // 2
GW = 64 // LPW
TS = GW * W
lanes = np.arange(64)


def rev4(x, digits):
    r = 0
    for _ in range(digits):
        r = (r << 2) | (x & 3)
        x >>= 2
    return r


instrs = []
g, xp = lanes // LPW, lanes % LPW
for u in range(2):
    run = g * W + (np.array([rev4(int(2 * x + u), layers) for x in xp]) << log_base)
    for q in range(base // 2):
        instrs.append(("w128", run + 2 * q))
cols = base
for l in range(layers):
    for t0 in range(0, TS // 4, 64):
        t = t0 + lanes
        chunk, i = t // cols, t % cols
        for q in range(4):
            a = chunk * 4 * cols + i + q * cols
            instrs.append(("r64", a))
            if l + 1 < layers:
                instrs.append(("w64", a))
    cols *= 4
GROUPS = {"w64": [list(range(k * 16, k * 16 + 16)) for k in range(4)], "r64": [list(range(0, 32)), list(range(32, 64))],
          "w128": [list(range(k * 8, k * 8 + 8)) for k in range(8)]}
UNIT = {"w64": (0, 16), "r64": (0, 32), "w128": (1, 8)}
rows, shifts, mods, kinds = [], [], [], []
for kind, a in instrs:
    for grp in GROUPS[kind]:
        v = a[grp]
        rows.append(np.pad(v, (0, 32 - len(v)), constant_values=-1)); shifts.append(UNIT[kind][0]); mods.append(UNIT[kind][1]); kinds.append(kind)
rows = np.array(rows); shifts = np.array(shifts)[:, None]; mods = np.array(mods)[:, None]; kinds = np.array(kinds)
nbits = TS.bit_length() - 1


def cycles(M, sel=None):
    r = rows if sel is None else rows[sel]
    s = r.copy()
    for dst, srcs in M.items():
        for sb in srcs:
            s ^= ((r >> sb) & 1) << dst
    bank = np.where(r >= 0, (s >> (shifts if sel is None else shifts[sel])) % (mods if sel is None else mods[sel]), -1)
    mx = np.zeros(len(r), dtype=np.int64)
    for b in range(32):
        mx = np.maximum(mx, (bank == b).sum(axis=1))
    return int(mx.sum())


def n_offsets(M):
    return len({sb - d for d, srcs in M.items() for sb in srcs})


L2 = 2 * layers
if base == 16:
    shipped = {4: (6,), 1: (log_base + L2 - 1,), 2: (log_base + L2 - 4,), 3: (log_base + L2 - 3,)}
else:
    shipped = {3: (5,), 4: (6,)}
ideal = len(rows)
print(f"W={W}: base {base}, layers {layers}, tile {TS}; conflict-free {ideal} lane-group cycles; identity {cycles({})}; shipped swizzle {cycles(shipped)} {shipped}")
options = {}
for dst in (1, 2, 3, 4):
    srcs = list(range(max(dst + 1, log_base), nbits))
    options[dst] = [()] + [(a,) for a in srcs] + list(itertools.combinations(srcs, 2))
rng = np.random.default_rng(1)
best = None
for start in range(10):
    M = {d: (options[d][rng.integers(len(options[d]))] if start > 1 else (shipped.get(d, ()) if start == 1 else ())) for d in (1, 2, 3, 4)}
    cur = cycles(M)
    improved = True
    while improved:
        improved = False
        for dst in (4, 3, 2, 1):
            for o in options[dst]:
                T = dict(M); T[dst] = o
                c = cycles(T)
                if c < cur or (c == cur and n_offsets(T) < n_offsets(M)):
                    cur, M, improved = c, T, True
    if best is None or cur < best[0] or (cur == best[0] and n_offsets(M) < n_offsets(best[1])):
        best = (cur, M)
print(f"best: {best[0]} ({best[0] / ideal:.2f}x): {best[1]}")
for kind in ("w128", "r64", "w64"):
    ids = np.nonzero(kinds == kind)[0]
    if len(ids):
        print(f"  {kind}: {len(ids)} groups: identity {cycles({}, ids)}  shipped {cycles(shipped, ids)}  best {cycles(best[1], ids)}")
masks = [sum(1 << sb for sb in best[1].get(d, ())) for d in (1, 2, 3, 4)]
print(f"template <> struct Spark2Swz<{W}> {{ static constexpr uint32_t m[4] = {{{', '.join(hex(m) for m in masks)}}}; }};")
