#!/bin/bash
# Development: timing-only ablations of the wave-local kernel k_spark2 (16 GiB cf32, W = 128 by default), one process per variant so that
# the plan-time compiler sees each -DQD_SPARK_ABL value.  bits: 1 no output stores, 2 no |X|, 4 last layer only, 8 no base butterflies.
# usage: scripts/spark_ablate.sh [W] [shift]      (needs quadrs_amd/libquadrs_hip_dev.so: python quadrs_amd/build.py --dev)
W=${1:-128}; SH=${2:-None}
export QD_LIB_PATH=$PWD/quadrs_amd/libquadrs_hip_dev.so
for abl in 0 1 2 3 4 8 15; do
  QD_JIT_CACHE=off QD_JIT_FLAGS="-DQD_SPARK_ABL=$abl" python - "$W" "$SH" "$abl" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
import quadrs_amd as Q
W, shift, abl = int(sys.argv[1]), (None if sys.argv[2] == "None" else int(sys.argv[2])), int(sys.argv[3])
n = 1 << 31
dev = torch.device("cuda", 0)
src = bench.synth_slab(torch, 0, 0, n, 0x5EED0002, dev)
p = Q.Plan(0, 21_000_000, n, shift_hz=shift, width=W, stride=W)
out = torch.empty(p.n_windows, W, dtype=torch.float32, device=dev)
for _ in range(3): p.run_device(src, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8): p.run_device(src, out)
e1.record(); torch.cuda.synchronize()
print(f"W={W} shift={shift} ablation {abl:2d}: kind {p.info.kernel_kind} flags {p.info.kernel_flags}: {e0.elapsed_time(e1) / 8:.3f} ms", flush=True)
PY
done
