#!/bin/bash
# scripts/build_variant.sh ab/libX.so [-DNAME=VAL ...] [SOURCE.hip]: a variant of libtde_hip.so for scripts/ab_rollout.py
OUT=$1; shift
SRC="$(dirname "$0")/../torchdriveenv_amd/csrc/tde_kernels.hip"
ARGS=()
for a in "$@"; do case "$a" in *.hip) SRC=$a;; *) ARGS+=("$a");; esac; done
mkdir -p "$(dirname "$OUT")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fvisibility=hidden \
  -fno-slp-vectorize -fno-vectorize -Wl,-soname,libtde_hip.so -I"$(dirname "$0")/../include" -I"$(dirname "$0")/../torchdriveenv_amd/csrc" "${ARGS[@]}" -o "$OUT" "$SRC"
