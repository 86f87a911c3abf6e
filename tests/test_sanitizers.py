"""CPU sanitizer pass over the test infrastructure and the host code (scripts/sanitize_cpu.sh): the oracle and the C++ driver's
parser built with -fsanitize=address,undefined and their CPU tests run under them.  The third leg of the script (the host side
of the C ABI through hipcc, ~80 s of compile) is run by hand per round: profiles/r03/sanitize_cpu.log."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_driver_are_clean_under_asan_ubsan(engine, tmp_path):
    env = dict(os.environ, QD_SAN_LEGS="12")
    env.pop("QD_ORACLE_SO", None); env.pop("QD_CLI_BIN", None)
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "sanitize_cpu.sh"), str(tmp_path)], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0 and "clean" in r.stdout.splitlines()[-1], r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("sanitizer reports: 0") == 2, r.stdout
