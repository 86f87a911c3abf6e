"""The C++ host driver (quadrs_amd/quadrs-hip): argument grammar on CPU, stdout parity on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from util import README_OOK, bits_equal, ook_pipeline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def cli(engine):
    from quadrs_amd import build as B
    if os.environ.get("QD_CLI_BIN"):          # an instrumented build of the driver (scripts/sanitize_cpu.sh)
        return os.environ["QD_CLI_BIN"]
    return B.build_cli()


def run(cli, *args, env=None, cwd=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([cli, *args], capture_output=True, env=e, cwd=cwd, timeout=300)


# ---------------- CPU: grammar and errors (nothing here reaches a kernel)

def test_usage_and_parse_errors(cli):
    r = run(cli)
    assert r.returncode != 0 and b"usage:" in r.stderr and r.stdout == b""
    for argv, msg in ((["frobnicate"], b"unrecognised command"),
                      (["shift"], b"'shift' requires a frequency argument"),
                      (["sparkfft", "-width"], b"-width requires an argument"),
                      (["sparkfft", "-width", "4", "-width", "8"], b"specified more than once"),
                      (["sparkfft", "-bogus", "1"], b"invalid flags"),
                      (["from", "nosuffix.bin"], b"unable to guess sample rate"),
                      (["from", "x.sr400.xyz"], b"unable to guess format"),
                      (["bucket", "-by", "time", "2"], b"must bucket -by freq"),
                      (["gen", "100"], b"gen requires at least one operation"),
                      (["ui"], b"out of scope")):
        r = run(cli, *argv)
        assert r.returncode != 0 and msg in r.stderr, (argv, r.stderr)


def test_chain_errors_before_any_kernel(cli, tmp_path):
    f = os.path.join(GOLDEN, "cupboard-superdec.sr400.cf32")
    r = run(cli, "shift", "100")
    assert r.returncode == 1 and b"shift requires an input" in r.stderr
    r = run(cli, "from", f, "shift", "200")                      # |f| < sr/2 (src/shift.rs:20-23)
    assert r.returncode == 1 and b"half the sample rate" in r.stderr
    r = run(cli, "from", f, "shift", "-199", "sparkfft", "-width", "3")
    assert r.returncode == 1 and b"power-of-two" in r.stderr
    assert r.stdout == b"sparkfft sample_rate=400\n"            # the header precedes the failure (src/fft.rs:19)
    r = run(cli, "from", str(tmp_path / "missing.sr1k.cf32"), "sparkfft")
    assert r.returncode == 1 and b"No such file" in r.stderr


def test_si_numbers_and_the_minus_rule(cli):
    """The reference's own unit test for its number parser (src/args.rs:457-465: "123", "1k", "47k", "0M") and the rule that
    decides whether an argument starting with '-' is an option or a negative number (src/args.rs:420-425: a number only if its
    THIRD character is a digit — so `shift -50k` and `shift -12` are frequencies, `shift -5k` and `shift -5` are options without a
    value), through `-parse-only`."""
    f = os.path.join(GOLDEN, "cupboard-superdec.sr400.cf32")
    for text, want in (("123", 123), ("1k", 1_000), ("47k", 47_000), ("0M", 0), ("3G", 3_000_000_000)):
        r = run(cli, "-parse-only", "from", "-sr", text, "-format", "cf32", f)
        assert r.returncode == 0 and f"sample_rate={want} ".encode() in r.stdout, (text, r.stdout, r.stderr)
    for text in ("2.5M", "1e3", "1k5", "7m", "5K", "3T", "-5"):                       # u64::from_str fails on all of these (only k / M / G are suffixes)
        r = run(cli, "-parse-only", "from", "-sr", text, "-format", "cf32", f)
        assert r.returncode != 0, text
    for text, want in (("-50k", -50_000), ("-280000", -280_000), ("-12", -12), ("5k", 5_000)):
        r = run(cli, "-parse-only", "from", "-sr", "1M", "-format", "cf32", f, "shift", text)
        assert r.returncode == 0 and f"shift {want}\n".encode() in r.stdout, (text, r.stdout, r.stderr)
    for text in ("-5k", "-5", "-x"):                                                  # taken for an option name: an error, as in the reference
        r = run(cli, "-parse-only", "from", "-sr", "1M", "-format", "cf32", f, "shift", text)
        assert r.returncode != 0, text


def test_filename_guessing_and_overrides(cli):
    """guess_format_from_name / guess_sample_rate (src/args.rs:100-135, 328-333, 392-402): the `srNNN[kMG]` word, gqrx and rtl_433
    capture names, the extension table, and the -sr / -format overrides (src/args.rs:65-98), through `-parse-only` (nothing is
    opened).  Expected values are what the reference's regexes and parse_si give for these names."""
    cases = [
        ("examples/cupboard-superdec.sr400.cf32", [], "sample_rate=400 format=cf32"),            # the reference's own example files
        ("examples/fsk-example.sr21M.fc32", [], "sample_rate=21000000 format=cf32"),
        ("a.sr2k.c8", [], "sample_rate=2000 format=cs8"),
        ("capture.sr1G.sc16", [], "sample_rate=1000000000 format=cs16"),
        ("x.sr250k.su8", [], "sample_rate=250000 format=cu8"),
        ("gqrx_20180126_111922_868000000_8000000_fc.raw", [], "sample_rate=8000000 format=cf32"),   # src/args.rs:110-117 (its comment's example)
        ("/data/gqrx_20200101_000000_433920000_2400000_fc.raw", [], "sample_rate=2400000 format=cf32"),
        ("g001_433.92M_250k.cu8", [], "sample_rate=250000 format=cu8"),                            # rtl_433 (src/args.rs:119-125)
        ("g017_868M_1024k.cu8", [], "sample_rate=1024000 format=cu8"),
        ("g001_433.92M_250k.cu8", ["-sr", "1M"], "sample_rate=1000000 format=cu8"),                # overrides win
        ("noext.sr8M", ["-format", "cs16"], "sample_rate=8000000 format=cs16"),
        ("blob.bin", ["-sr", "48k", "-format", "fc32"], "sample_rate=48000 format=cf32"),
        ("gqrx_1_2_3_4_fc.raw", ["-format", "c8"], "sample_rate=4 format=cs8"),
    ]
    for name, flags, want in cases:
        r = run(cli, "-parse-only", "from", *flags, name, "sparkfft")
        assert r.returncode == 0, (name, r.stderr)
        first = r.stdout.decode().splitlines()[0]
        assert first == f"from file={name} {want}", (name, first)
    for name, flags, msg in (("sr400.cf32x", [], b"unable to guess format"),          # `sr400` is a word here, the extension is unknown
                             ("xsr400.cf32", [], b"unable to guess sample rate"),      # no word boundary in front of `sr`
                             ("noext_sr8M.cf32", [], b"unable to guess sample rate"),  # `_` is a word character for \\b, in Rust's regex as in ECMAScript
                             ("a.sr400.cf32", ["-format", "wav"], b"unrecognised extension"),
                             ("g001_433M_250.cu8", [], b"unable to guess sample rate")):   # rtl_433 names end in `k`
        r = run(cli, "-parse-only", "from", *flags, name, "sparkfft")
        assert r.returncode != 0 and msg in r.stderr, (name, r.stderr)
    # defaults of the other operators (src/args.rs:151-219): -power 20 => 40 taps, -decimate 8, -width 128, stride = width
    r = run(cli, "-parse-only", "from", "a.sr1M.cf32", "shift", "-25k", "lowpass", "100k", "sparkfft", "bucket", "-by", "freq", "2")
    assert r.returncode == 0 and r.stdout.decode().splitlines()[1:] == [
        "shift -25000", "lowpass frequency=100000 decimate=8 size=40", "sparkfft width=128 stride=128 range=no", "bucket width=128 stride=128 levels=2"], r.stdout


# ---------------- GPU: byte-identical stdout / files

@pytest.mark.gpu
@pytest.mark.parametrize("nofuse", ["", "1"])
def test_readme_ook_stdout(cli, oracle, cupboard, nofuse):
    f = os.path.join(GOLDEN, "cupboard-superdec.sr400.cf32")
    r = run(cli, "from", f, "sparkfft", "-width", "4", "-stride", "2", "-range", "0.001:0.01",
            env={"QUADRS_HIP_NO_FUSE": nofuse} if nofuse else None)
    assert r.returncode == 0, r.stderr
    want = oracle.Chain.from_bytes(cupboard, oracle.FMT_CF32, 400).spark_text(4, 2, (0.001, 0.01))
    assert r.stdout == want
    assert ook_pipeline(r.stdout) == README_OOK


@pytest.mark.gpu
@pytest.mark.parametrize("nofuse", ["", "1"])
def test_readme_fsk_stdout(cli, oracle, fsk, nofuse):
    f = os.path.join(GOLDEN, "fsk-example-head65536.sr21M.cf32")
    args = ["from", f, "shift", "280000", "lowpass", "-power", "200", "-decimate", "32", "200000",
            "sparkfft", "-width", "64", "-stride", "16", "-range", "0.002:0.2"]
    r = run(cli, *args, env={"QUADRS_HIP_NO_FUSE": nofuse} if nofuse else None)
    assert r.returncode == 0, r.stderr
    ch = oracle.Chain.from_bytes(fsk, oracle.FMT_CF32, 21_000_000).shift(280000).lowpass(200000, 32, 400)
    want = ch.spark_text(64, 16, (0.002, 0.2))
    got_lines, want_lines = r.stdout.split(b"\n"), want.split(b"\n")
    assert len(got_lines) == len(want_lines) and got_lines[0] == b"sparkfft sample_rate=656250"
    same = sum(a == b for a, b in zip(got_lines, want_lines))
    assert same >= len(want_lines) - 1        # a 1-ulp NCO event may flip one glyph at a bin edge
    assert len(set(want_lines)) > 20           # the range really draws the spectrum


@pytest.mark.gpu
@pytest.mark.parametrize("stride", [1024, 512])
def test_windows_larger_than_the_lds_tile_print_the_reference_bytes(cli, oracle, tmp_path, stride):
    """`lowpass -decimate 32 -power 100 ... sparkfft -width 1024`: 32 968 source samples per window, more than one workgroup's LDS holds.
    stride == width runs as the library's two-stage plan; overlapping windows (stride 512) have no fused plan and the driver pulls them
    through the iterator chain (read_exact_at per window, as src/fft.rs:30 does) — both print the oracle's bytes."""
    n = 6 * 1024 * 32 + 900
    t = np.arange(n)
    rng = np.random.default_rng(9)
    z = 0.2 * np.exp(2j * np.pi * 0.0004 * t) * np.sign(np.sin(t * 0.0003) + 1e-9) + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    data = np.stack([z.real, z.imag], 1).astype(np.float32).tobytes()
    f = tmp_path / "big-window.sr21M.cf32"
    f.write_bytes(data)
    ch = oracle.Chain.from_bytes(data, oracle.FMT_CF32, 21_000_000).lowpass(150000, 32, 200)
    norms, _ = ch.spark_fft(1024, stride)
    rmin, rmax = float(np.percentile(norms, 20)), float(np.percentile(norms, 99))
    r = run(cli, "from", str(f), "lowpass", "-power", "100", "-decimate", "32", "150000", "sparkfft", "-width", "1024", "-stride", str(stride),
            "-range", f"{rmin!r}:{rmax!r}")
    assert r.returncode == 0, r.stderr
    want = ch.spark_text(1024, stride, (np.float32(rmin), np.float32(rmax)))
    assert r.stdout == want


@pytest.mark.gpu
def test_bucket_and_default_flags(cli, oracle, fsk):
    f = os.path.join(GOLDEN, "fsk-example-head65536.sr21M.cf32")
    r = run(cli, "from", f, "shift", "280000", "lowpass", "2000000", "bucket", "-by", "freq", "2")
    assert r.returncode == 0, r.stderr
    ch = oracle.Chain.from_bytes(fsk, oracle.FMT_CF32, 21_000_000).shift(280000).lowpass(2_000_000, 8, 40)  # defaults
    want = "".join(str(v) for v in ch.freq_levels(128, 128)) + "\n"
    assert r.stdout.decode() == want


@pytest.mark.gpu
@pytest.mark.parametrize("nofuse", ["", "1"])
def test_write_after_lowpass_matches_reference_bytes(cli, oracle, tmp_path, nofuse):
    """do_write (src/lib.rs:178-213): same bytes, then the reference's end-of-stream assert as exit 1."""
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((3 * 4096 * 4 + 40 + 100, 2)) * 0.05).astype(np.float32)
    src = tmp_path / "in.sr1M.cf32"
    src.write_bytes(x.tobytes())
    r = run(cli, "from", str(src), "lowpass", "-decimate", "4", "100000", "write", "out", cwd=str(tmp_path),
            env={"QUADRS_HIP_NO_FUSE": nofuse} if nofuse else None)
    rc, n, samples = oracle.Chain.from_bytes(x.tobytes(), 0, 1_000_000).lowpass(100_000, 4, 40).do_write(4 * 4096)
    assert rc == 2 and r.returncode == 1 and b"short read" in r.stderr        # LowPass::len over-reports by one
    got = np.frombuffer((tmp_path / "out.sr250000.cf32").read_bytes(), dtype=np.float32).reshape(-1, 2)
    assert got.shape[0] == n and bits_equal(got, samples)
    r2 = run(cli, "from", str(src), "lowpass", "-decimate", "4", "100000", "write", "out", cwd=str(tmp_path))
    assert r2.returncode == 1 and b"exists" in r2.stderr.lower()               # create_new unless -overwrite yes
    r3 = run(cli, "from", str(src), "write", "-overwrite", "yes", "copy", cwd=str(tmp_path))
    assert r3.returncode == 0 and (tmp_path / "copy.sr1000000.cf32").read_bytes() == x.tobytes()


@pytest.mark.gpu
def test_gen_chain(cli, oracle):
    r = run(cli, "gen", "-cos", "1000", "-cos", "-3k", "-len", "0.01", "48k", "sparkfft", "-width", "16", "-range", "0.5:20")
    assert r.returncode == 0, r.stderr
    want = oracle.Chain.gen([1000, -3000], 48000, 0.01).spark_text(16, 16, (0.5, 20.0))
    assert r.stdout == want


@pytest.mark.gpu
def test_gen_source_runs_fused_on_the_device(cli, oracle):
    """`gen ... | shift | lowpass | sparkfft` and `... | bucket`: the tones are generated in HBM (qd_gen on a device buffer),
    the chain runs as one plan on them and only glyphs / digits come back — same text as the block-iterator path and as
    the oracle."""
    chain = ["gen", "-cos", "1000", "-cos", "-3k", "-cos", "7500", "-len", "0.25", "48k",
             "shift", "-1200", "lowpass", "-power", "12", "-decimate", "4", "4000"]
    for sink in (["sparkfft", "-width", "16", "-stride", "8", "-range", "0.02:3"], ["bucket", "-width", "32", "-by", "freq", "2"]):
        fused = run(cli, *chain, *sink)
        slow = run(cli, *chain, *sink, env={"QUADRS_HIP_NO_FUSE": "1"})
        assert fused.returncode == 0 and slow.returncode == 0, (fused.stderr, slow.stderr)
        assert fused.stdout == slow.stdout and len(fused.stdout) > 50
    ch = oracle.Chain.gen([1000, -3000, 7500], 48000, 0.25).shift(-1200).lowpass(4000, 4, 24)
    want = ch.spark_text(16, 8, (0.02, 3.0))
    got = run(cli, *chain, "sparkfft", "-width", "16", "-stride", "8", "-range", "0.02:3").stdout
    assert got.decode() == want if isinstance(want, str) else got == want


@pytest.mark.gpu
def test_gpus_flag_shards_inside_one_process(cli, tmp_path):
    """`quadrs-hip -gpus N ...`: the sink's windows go over N shards (devices repeat on a one-GPU box) through
    qd_plan_run_sharded; stdout and written files are byte-identical to the one-device run.  The source file is mapped and,
    when large enough, registered with the HIP runtime (QD_MEM_HOST_PINNED)."""
    rng = np.random.default_rng(11)
    n = 9_000_000                                                     # 72 MB cf32: above the 64 MiB pinning threshold
    t = np.arange(n)
    z = 0.02 * np.exp(2j * np.pi * (-280000.0 / 21e6) * t) * np.sign(np.sin(2 * np.pi * 9600.0 / 21e6 * t) + 1e-9)
    x = np.stack([z.real, z.imag], axis=1).astype(np.float32) + (rng.standard_normal((n, 2)) * 0.002).astype(np.float32)
    src = tmp_path / "big.sr21M.cf32"
    x.tofile(src)
    chain = ["from", str(src), "shift", "280000", "lowpass", "-power", "100", "-decimate", "32", "200000"]
    one = run(cli, *chain, "sparkfft", "-width", "128", "-range", "0.0005:0.05")
    assert one.returncode == 0 and one.stdout.count(b"\n") > 2000, one.stderr[-500:]
    for g in ("2", "3", "8"):
        r = run(cli, "-gpus", g, *chain, "sparkfft", "-width", "128", "-range", "0.0005:0.05")
        assert r.returncode == 0 and r.stdout == one.stdout, (g, r.stderr[-500:])
    r = run(cli, "-gpus", "4", *chain, "bucket", "-width", "64", "-by", "freq", "2")
    r1 = run(cli, *chain, "bucket", "-width", "64", "-by", "freq", "2")
    assert r.returncode == 0 and r.stdout == r1.stdout and len(r.stdout) > 1000
    a = run(cli, *chain, "write", "-overwrite", "yes", str(tmp_path / "one"))
    b = run(cli, "-gpus", "2", *chain, "write", "-overwrite", "yes", str(tmp_path / "two"))
    assert a.returncode == b.returncode
    fa = (tmp_path / "one.sr656250.cf32").read_bytes()
    assert fa == (tmp_path / "two.sr656250.cf32").read_bytes() and len(fa) > 2_000_000
    bad = run(cli, "-gpus", "99", *chain, "sparkfft")
    assert bad.returncode == 2 and b"-gpus takes" in bad.stderr
