#!/bin/bash
# usage: bash tools/ab_libs.sh outdir lib1 lib2 ...   (checksums of each build, then the alternating timing A/B)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=$1; shift; mkdir -p $O
for l in "$@"; do echo "== $l" >> $O/checksums.txt; DD_HIP_LIB=$PWD/$l python tools/lib_checksum.py >> $O/checksums.txt 2>&1; done
grep -v amdgpu.ids $O/checksums.txt
python tools/ab_builds.py "$@" 3 2>&1 | tee $O/ab.txt
