#!/bin/bash
# CPU: build the library of another commit as libcimbar_amd/variants/libcimbar_hip_<name>.so for same-box A/B runs (CIMBAR_HIP_LIB=...).
# Usage: tools/ab_variant.sh <commit> <name>
set -e
C=${1:-HEAD}; N=${2:-base}
R=$(git rev-parse --show-toplevel)
T=$(mktemp -d)
git -C $R worktree add -f $T $C > /dev/null 2>&1
mkdir -p $R/libcimbar_amd/variants
(cd $T && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -o $R/libcimbar_amd/variants/libcimbar_hip_$N.so libcimbar_amd/csrc/cimbar_hip.hip)
git -C $R worktree remove --force $T
ls -la $R/libcimbar_amd/variants/libcimbar_hip_$N.so
