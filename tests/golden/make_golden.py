"""Regenerates tests/golden/oracle_vectors.npz from the CPU oracle (oracle/quadrs_oracle.c).

The reference (Rust) cannot be built in this image, and its own tests hold no vectors on this
path, so these are outputs of the *restatement*, pinned so that (a) a change in the oracle or
in the platform libm is noticed, and (b) the GPU is checked against stored bits even where the
oracle .so is not rebuilt.  Inputs are tiny and deterministic.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    # (iii) taps for the four (T, fc, sr) of SURVEY §8(c)
    for T, fc, sr in ((40, 2_000_000, 21_000_000), (200, 200_000, 21_000_000), (400, 200_000, 21_000_000),
                      (512, 5_000_000, 100_000_000)):
        out[f"taps_{T}_{fc}_{sr}"] = O.taps(fc, sr, T)
    # (iv) unpack tables: all 256 byte values for the 8-bit formats, every 16-bit value for cs16
    b = np.arange(256, dtype=np.uint8)
    pairs8 = np.stack([b, b[::-1]], axis=1).reshape(-1).tobytes()
    out["unpack_cs8"] = O.unpack(O.FMT_CS8, pairs8)
    out["unpack_cu8"] = O.unpack(O.FMT_CU8, pairs8)
    h = np.arange(65536, dtype=np.uint16)
    pairs16 = np.stack([h, h[::-1]], axis=1).reshape(-1).astype("<u2").tobytes()
    full = O.unpack(O.FMT_CS16, pairs16)
    out["unpack_cs16_every257"] = full[::257]
    out["unpack_cs16_xor"] = np.bitwise_xor.reduce(full.view(np.uint32).reshape(-1))[None]
    # (v) NCO multipliers for cfg 2's ratio at assorted n
    ratio = O.shift_ratio(280000, 21_000_000)
    ns = np.array([0, 1, 2, 74, 75, 76, 511, 512, 513, 12345, 2**24 + 1, 2**27 - 1, 2**31 - 1, 2**31, 2**33 - 1,
                   2**34 - 1], dtype=np.uint64)
    out["nco_n"] = ns
    out["nco_ratio"] = np.array([ratio])
    out["nco_mul"] = O.shift_multipliers(ratio, ns)
    # (vi) gen: first 64 samples of the cfg 4 tone list
    tones = [(k - 32) * 1_562_500 + 390_625 for k in range(64)]
    g = O.Chain.gen(tones, 100_000_000, 42.94967296)
    out["gen_tones"] = np.array(tones, dtype=np.int64)
    out["gen_len"] = np.array([g.len()], dtype=np.uint64)
    out["gen_first64"] = g.read_at(0, 64)[1]
    out["gen_far64"] = g.read_at(2**32 - 64, 64)[1]
    # (ii) FSK README chain: first 64 / last 16 rows of norms on the committed 65536-sample head
    fsk = open(os.path.join(HERE, "fsk-example-head65536.sr21M.cf32"), "rb").read()
    ch = O.Chain.from_bytes(fsk, O.FMT_CF32, 21_000_000).shift(280000).lowpass(200000, 32, 400)
    norms, _ = ch.spark_fft(64, 16)
    out["fsk_norms_first64"] = norms[:64]
    out["fsk_norms_last16"] = norms[-16:]
    out["fsk_nwin"] = np.array([norms.shape[0]], dtype=np.uint64)
    # (vii) write after lowpass: the 0x1000-block truncation pattern (3 blocks) on synthetic data
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((3 * 4096 * 4 + 40 + 100, 2)) * 0.05).astype(np.float32)
    # the input is regenerated from seed 7 by the test (PCG64 streams are stable), not stored
    w = O.Chain.from_bytes(x.tobytes(), O.FMT_CF32, 1_000_000).lowpass(100_000, 4, 40)
    rc, n, samples = w.do_write(4 * 4096)
    out["write_rc_n"] = np.array([rc, n], dtype=np.int64)
    out["write_samples"] = samples
    # FFT twiddles + a fixed-input FFT for W in {4, 64, 128, 1024}
    rng = np.random.default_rng(11)
    for W in (4, 64, 128, 1024):
        xin = rng.standard_normal((W, 2)).astype(np.float32)
        out[f"fft_in_{W}"] = xin
        out[f"fft_out_{W}"] = O.fft(xin)
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_vectors.npz"), sum(v.nbytes for v in out.values()), "bytes raw")


if __name__ == "__main__":
    main()
