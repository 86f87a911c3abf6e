"""bench.py's own multi-rank launch path (`python bench.py --gpus N` with no launcher), on the CPU with a stub step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "3", "--warmup", "1", *extra],
                          env=env, capture_output=True, text=True, timeout=180)


def test_self_spawn_two_ranks_prints_one_line():
    r = _run("--gpus", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"


def test_self_spawn_fails_when_a_rank_fails():
    r = _run("--gpus", "2", "--stub-fail-rank", "1")
    assert r.returncode != 0
    assert "rank exit codes" in r.stderr


def test_single_rank_needs_no_spawn():
    r = _run("--gpus", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_valu_roof_model():
    """The VALU roof bench.py reports: cfg3' is HBM-bound on paper, cfg3 / cfg4 are bound by the exact-order arithmetic."""
    sys.path.insert(0, ROOT)
    import bench
    roofs = {}
    for name, order in (("cfg2", 1), ("cfg3p", 1), ("cfg3", 2), ("cfg4", 0)):
        cfg = bench.WORKLOADS[name]
        valu, f32, f64 = bench.valu_roof_msamples(cfg, order)
        hbm = bench.HBM_PEAK_GBPS * 1e9 / (bench.BPS[cfg["fmt"]] + cfg["W"] * 4 / (cfg["S"] * cfg["lp"][1])) / 1e6
        roofs[name] = (valu, hbm)
        assert f32 >= 4.0 * cfg["lp"][2] / cfg["lp"][1]
    assert roofs["cfg2"][0] > roofs["cfg2"][1] and roofs["cfg3p"][0] > roofs["cfg3p"][1]     # hbm-bound
    assert roofs["cfg3"][0] < roofs["cfg3"][1] and roofs["cfg4"][0] < roofs["cfg4"][1]       # valu-bound
