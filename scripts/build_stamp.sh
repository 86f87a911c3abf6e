#!/bin/bash
# Diagnostic build with per-phase cycle stamps (-DQD_DEVELOP -DQD_STAMP): quadrs_amd/libquadrs_hip_stamp.so, used through
# QD_LIB_PATH.  It sits next to the main library so that the plan-time compiler finds csrc/.
set -e
cd "$(dirname "$0")/.."
python quadrs_amd/build.py --dev --stamp >/dev/null
mv quadrs_amd/libquadrs_hip_dev.so quadrs_amd/libquadrs_hip_stamp.so
echo quadrs_amd/libquadrs_hip_stamp.so
