#!/bin/bash
# build the library; non-zero exit status (and the error lines) when a compile fails
cd "$(dirname "$0")/.." && out=$(python -m decompdiff_amd.build 2>&1); rc=$?
echo "$out" | grep -E "error" | head -20
exit $rc
