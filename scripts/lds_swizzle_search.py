"""Development: XOR swizzles for the wave-local kernel's (k_spark, lean path) LDS transform buffer.

The wave's buffer holds TS complex samples (8 B elements).  Per tile it is touched by
  A  the scatter of the unpacked samples into rustfft's transposed order      ds_write_b64   (groups of 16 contiguous lanes, 128-B bank window)
  B  the base butterflies: lane t owns the run [t * base, (t + 1) * base)      ds_read_b128 / ds_write_b128 in 16-byte pieces
                                                                               (reads: 4 groups of 16 lanes, 256-B window; writes: 8 groups of 8, 128-B window)
  C  the radix-4 layers: butterfly t reads / writes chunk * 4 cols + i + q cols  ds_read_b64 (2 groups of 32, 256-B window) / ds_write_b64
  D  the epilogue: element (lane + 64 k) ^ (W / 2)                              ds_read_b64
(bank rules: MI355X_MICROARCH.md, LDS).  A swizzle sigma(p) = p ^ M p with M strictly "higher bits into lower bits" (a bijection) and bit 0
untouched (16-byte pieces stay whole) is searched by coordinate descent: for every destination bit 1 ... 4 the XOR of at most two
source bits.  Prints LDS-array cycles per tile for the identity and for the best map found.
The map is searched per (W, SPL) for both tile sizes the kernel uses (512 samples with a shift, 1024 without) at once, with source bits
>= log2(base) (so that a base butterfly's run sees ONE XOR value) and, among equals, the fewest distinct (source - destination) offsets
(each costs a shift and an AND where the XOR value is formed at run time).  Prints the C++ specialisation for qd_chain.h.
usage: python scripts/lds_swizzle_search.py W SPL"""
import itertools, sys
import numpy as np

W, SPL = (int(x) for x in sys.argv[1:3])
logW = W.bit_length() - 1
base = 16 if logW % 2 == 0 else 8
if W < 8:
    raise SystemExit("W >= 8")
log_base = base.bit_length() - 1
layers = (logW - log_base) // 2
CH = 64 * SPL
lanes = np.arange(64)


def rev4(x, digits):
    r = 0
    for _ in range(digits):
        r = (r << 2) | (x & 3)
        x >>= 2
    return r


READ128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
READ128_GROUPS += [[l + 32 for l in g] for g in READ128_GROUPS]

def instructions(TS):
    NCH = TS // CH
    # every wave-instruction as (kind, addresses[64]) with element addresses (8-B units); -1 = lane inactive
    instrs = []
    for c in range(NCH):
        for u in range(SPL):
            m = c * CH + lanes * SPL + u
            k = m & (W - 1)
            xx, yy = k & ((1 << (2 * layers)) - 1), k >> (2 * layers)
            pos = (m & ~(W - 1)) + yy + (np.array([rev4(int(v), layers) for v in xx]) << log_base)
            instrs.append(("w64", pos))
    n_task = TS // base
    for t0 in range(0, n_task, 64):
        t = t0 + lanes
        act = t < n_task
        for q in range(base // 2):
            a = np.where(act, t * base + 2 * q, -1)
            instrs.append(("r128", a)); instrs.append(("w128", a))
    cols = base
    for _ in range(layers):
        for t0 in range(0, TS // 4, 64):
            t = t0 + lanes
            chunk, i = t // cols, t % cols
            for q in range(4):
                a = chunk * 4 * cols + i + q * cols
                instrs.append(("r64", a)); instrs.append(("w64", a))
        cols *= 4
    for k in range(TS // 64):
        instrs.append(("r64", (lanes + 64 * k) ^ (W // 2)))

    return instrs

GROUPS = {"w64": [list(range(g * 16, g * 16 + 16)) for g in range(4)], "r64": [list(range(0, 32)), list(range(32, 64))],
          "r128": READ128_GROUPS, "w128": [list(range(g * 8, g * 8 + 8)) for g in range(8)]}
UNIT = {"w64": (0, 16), "r64": (0, 32), "r128": (1, 16), "w128": (1, 8)}     # (shift, modulus): bank unit of an element address

# flatten: one row per (instruction, group) with up to 32 addresses, both tile sizes
rows, shifts, mods, kinds = [], [], [], []
for TS in (512, 1024):
    if TS < CH:
        continue
    for kind, a in instructions(TS):
        for g in GROUPS[kind]:
            v = a[g]
            rows.append(np.pad(v, (0, 32 - len(v)), constant_values=-1)); shifts.append(UNIT[kind][0]); mods.append(UNIT[kind][1]); kinds.append(kind)
rows = np.array(rows); shifts = np.array(shifts)[:, None]; mods = np.array(mods)[:, None]; kinds = np.array(kinds)
valid = rows >= 0
nbits = 9                                   # source bits 0 ... 8 exist in both tiles


def cycles(M):
    """M: dict dst bit -> tuple of source bits"""
    p = rows.copy()
    s = p.copy()
    for dst, srcs in M.items():
        for sb in srcs:
            s ^= ((p >> sb) & 1) << dst
    bank = (s >> shifts) % mods
    key = np.where(valid, bank, -1)
    # multiplicity per row: max count of a bank value
    mx = np.zeros(len(rows), dtype=np.int64)
    for b in range(32):
        mx = np.maximum(mx, (key == b).sum(axis=1))
    return int(mx.sum())


def n_offsets(M):
    return len({sb - d for d, srcs in M.items() for sb in srcs})


ident = cycles({})
ideal = len(rows)
print(f"W={W} SPL={SPL}: base {base}, layers {layers}; {ideal} lane-group cycles when conflict-free (tiles of 512 + 1024 samples); identity layout {ident}")
best = None
options = {}
for dst in (1, 2, 3, 4):
    srcs = list(range(max(dst + 1, log_base), nbits))
    options[dst] = [()] + [(a,) for a in srcs] + list(itertools.combinations(srcs, 2))
rng = np.random.default_rng(1)
for start in range(10):
    M = {d: (options[d][rng.integers(len(options[d]))] if start else ()) for d in (1, 2, 3, 4)}
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
    print(f"  start {start}: {cur} cycles with {M}", flush=True)
print(f"best: {best[0]} cycles ({best[0] / ideal:.2f}x conflict-free, identity {ident / ideal:.2f}x): {best[1]}")
print("breakdown (identity -> best) per access kind:")
for kind in ("w64", "r128", "w128", "r64"):
    ids = np.nonzero(kinds == kind)[0]
    sub_rows, sub_sh, sub_md = rows[ids], shifts[ids], mods[ids]
    def cyc(M):
        s = sub_rows.copy()
        for dst, srcs in M.items():
            for sb in srcs:
                s ^= ((sub_rows >> sb) & 1) << dst
        bank = np.where(sub_rows >= 0, (s >> sub_sh) % sub_md, -1)
        mx = np.zeros(len(sub_rows), dtype=np.int64)
        for b in range(32):
            mx = np.maximum(mx, (bank == b).sum(axis=1))
        return int(mx.sum())
    print(f"  {kind}: {len(ids)} groups: {cyc({})} -> {cyc(best[1])}")
masks = [sum(1 << sb for sb in best[1].get(d, ())) for d in (1, 2, 3, 4)]
print(f"template <> struct SparkSwz<{W}, {SPL}> {{ static constexpr uint32_t m[4] = {{{', '.join(hex(m) for m in masks)}}}; }};   // {ident} -> {best[0]} LDS lane-group cycles per 512 + 1024 tile (conflict-free: {ideal})")
