#!/bin/bash
# usage (in the build container): bash scripts/ab_snapshot.sh [git ref, default HEAD]
# Exports bench.py + regda_amd/ + oracle/ + include/ + configs/ of a git ref into .ab_base/ (git-ignored, travels with
# the gpurun snapshot) and builds its library there: the "A" arm of scripts/ab.sh.  A whole tree, not just a library:
# the ABI may differ between the two arms.
set -e
cd "$(dirname "$0")/.."
ref=${1:-HEAD}
rm -rf .ab_base && mkdir .ab_base
git archive "$ref" bench.py regda_amd oracle include configs profiles/pmc_traffic.json | tar -x -C .ab_base
make -C .ab_base/regda_amd/csrc -j8 > /dev/null
echo "snapshot of $ref ($(git rev-parse --short "$ref")) built in .ab_base/"
