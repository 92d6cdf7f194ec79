#!/bin/bash
# Builds libnextsearch_hip_base.so from the csrc/ of a git revision (default HEAD) next to the working tree's library,
# for same-box A/B runs (tools/gpu/ab.sh): box-to-box spread of the kernel times is ~5 %, more than most changes.
set -e
REV=${1:-HEAD}
R=$(cd $(dirname $0)/../.. && pwd)
T=$(mktemp -d)
git -C $R archive $REV nextsearch-api_amd/csrc include | tar -x -C $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed -I$T/include -I$T/nextsearch-api_amd/csrc -shared -o $R/nextsearch-api_amd/libnextsearch_hip_base.so $T/nextsearch-api_amd/csrc/ns_api.hip
rm -rf $T
echo "built libnextsearch_hip_base.so from $REV"
