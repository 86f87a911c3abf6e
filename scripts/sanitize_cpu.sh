#!/bin/bash
# CPU sanitizer pass ("race detection / sanitizers"; GPU AddressSanitizer is not available on this pool):
#   1. the oracle (oracle/quadrs_oracle.c) built with gcc -fsanitize=address,undefined, tests/test_oracle_golden.py under it;
#   2. the C++ driver's parser and Samples classes (quadrs_amd/cli/quadrs_hip_cli.cpp) built with g++ -fsanitize=address,undefined,
#      the CPU half of tests/test_cli.py under it (grammar, filename guessing, errors before any kernel);
#   3. the host side of the C ABI (quadrs_hip.hip: validation, plan bookkeeping, taps / FFT layout, partitioning) built by hipcc with
#      host-only instrumentation (-fsanitize=address,undefined -fno-gpu-sanitize), tests/test_abi_cpu.py under it.
# Runs in the GPU-less container; prints one summary line per leg.  usage: scripts/sanitize_cpu.sh [outdir]
set -u
cd "$(dirname "$0")/.."
out=${1:-/tmp/qd_sanitize}
mkdir -p "$out"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
GCC_ASAN=$(gcc -print-file-name=libasan.so)
status=0

echo "== 1. oracle under ASan + UBSan"
gcc -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-fast-math -ffp-contract=off -fPIC -std=c11 \
    -shared -o "$out/libquadrs_oracle_san.so" oracle/quadrs_oracle.c -lm || status=1
LD_PRELOAD=$GCC_ASAN QD_ORACLE_SO="$out/libquadrs_oracle_san.so" timeout 1500 python -m pytest tests/test_oracle_golden.py tests/test_unpack_division.py -x -q -p no:cacheprovider > "$out/oracle.log" 2>&1
rc=$?; tail -1 "$out/oracle.log"; [ $rc -ne 0 ] && status=1
grep -c "ERROR: AddressSanitizer\|runtime error:" "$out/oracle.log" | sed 's/^/   sanitizer reports: /'

echo "== 2. quadrs-hip driver (parser, Samples classes) under ASan + UBSan"
g++ -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 -Wall -I include -o "$out/quadrs-hip-san" \
    quadrs_amd/cli/quadrs_hip_cli.cpp -L quadrs_amd -lquadrs_hip -Wl,-rpath,"$PWD/quadrs_amd" -Wl,-rpath-link,/opt/rocm/lib || status=1
QD_CLI_BIN="$out/quadrs-hip-san" timeout 900 python -m pytest tests/test_cli.py -x -q -m "not gpu" -p no:cacheprovider > "$out/cli.log" 2>&1
rc=$?; tail -1 "$out/cli.log"; [ $rc -ne 0 ] && status=1
grep -c "ERROR: AddressSanitizer\|runtime error:" "$out/cli.log" | sed 's/^/   sanitizer reports: /'

if [ "${QD_SAN_LEGS:-123}" = "12" ]; then echo "sanitize_cpu (legs 1-2): $([ $status = 0 ] && echo clean || echo FAILED)"; exit $status; fi
echo "== 3. host side of the C ABI under ASan + UBSan (device code uninstrumented)"
CLANG_ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p "$out/obj"
ok=1
for f in quadrs_hip; do
  extra=""
  /opt/rocm/bin/hipcc -O1 -g -fno-omit-frame-pointer --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -fno-fast-math -I include \
      -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -mllvm -amdgpu-atomic-optimizer-strategy=None $extra \
      -c -o "$out/obj/$f.o" quadrs_amd/csrc/$f.hip > "$out/abi_build_$f.log" 2>&1 || ok=0
done
if [ $ok = 1 ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o "$out/libquadrs_hip_san.so" "$out/obj/quadrs_hip.o" -lhiprtc -ldl > "$out/abi_link.log" 2>&1 || ok=0
fi
if [ $ok = 1 ]; then
  mkdir -p "$out/csrc_link"; ln -sfn "$PWD/quadrs_amd/csrc" "$out/csrc"
  LD_PRELOAD=$CLANG_ASAN QD_LIB_PATH="$out/libquadrs_hip_san.so" timeout 900 python -m pytest tests/test_abi_cpu.py -x -q -p no:cacheprovider \
      --deselect tests/test_abi_cpu.py::test_builtin_kernels_do_not_spill --deselect tests/test_abi_cpu.py::test_shipped_library_reads_no_tuning_environment > "$out/abi.log" 2>&1
  rc=$?; tail -1 "$out/abi.log"; [ $rc -ne 0 ] && status=1
  grep -c "ERROR: AddressSanitizer\|runtime error:" "$out/abi.log" | sed 's/^/   sanitizer reports: /'
else
  echo "   instrumented build of the ABI library failed (see $out/abi_build_*.log): leg skipped"; status=1
fi
echo "sanitize_cpu: $([ $status = 0 ] && echo clean || echo FAILED)"
exit $status
