#!/bin/bash
# build-container helper: compile one .hip of regda_amd/csrc to gfx950 assembly and print resources (+ optionally one kernel's text)
#   bash scripts/dev/isa.sh conv_kernels [mangled-substring [first-line-offset count]]
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
f=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -S --cuda-device-only \
  "$ROOT/regda_amd/csrc/$f.hip" -I"$ROOT/regda_amd/csrc" -o /tmp/$f.s 2>&1 | grep -v "warning: argument unused" || true
python "$ROOT/scripts/lib/kres.py" /tmp/$f.s
if [ -n "${2:-}" ]; then
  a=$(grep -n "^_Z.*$2.*:" /tmp/$f.s | head -1 | cut -d: -f1)
  b=$(awk -v a=$a 'NR>a && /^\.Lfunc_end/ {print NR; exit}' /tmp/$f.s)
  sed -n "${a},${b}p" /tmp/$f.s > /tmp/kernel.s
  echo "kernel text: /tmp/kernel.s ($((b-a)) lines)"
fi
