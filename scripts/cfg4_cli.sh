#!/bin/bash
# BASELINE configs[3] end to end through the C++ driver: 64 cosines @100 Msps, 2^32 samples generated in HBM,
# 512-tap FIR /8, 1024-point windows, one bucket digit per window back to the host.
tones=""
for k in $(seq 0 63); do tones="$tones -cos $(( (k - 32) * 1562500 + 390625 ))"; done
time quadrs_amd/quadrs-hip gen $tones -len 42.94967296 100M lowpass -power 256 -decimate 8 5M bucket -width 1024 -by freq 2 > gpurun_out/cfg4_digits.txt
wc -c gpurun_out/cfg4_digits.txt; head -c 80 gpurun_out/cfg4_digits.txt; echo
