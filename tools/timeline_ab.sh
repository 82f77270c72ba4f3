#!/bin/bash
# kernel timeline of one step with and without an option of the measurement build.   usage: tools/timeline_ab.sh key=value [rows]
cd /tmp && export TMPDIR=/tmp
for v in "$1" "99=0"; do
  rm -rf /tmp/tl
  if [ "$v" = "99=0" ]; then args=""; echo "=== default"; else args="$v"; echo "=== $v"; fi
  DD_HIP_LIB=$GRAFT_REPO_ROOT/decompdiff_amd/lib/libdecompdiff_hip_dbg.so timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -- python $GRAFT_REPO_ROOT/tools/run_steps.py 30 $args > /dev/null 2>&1
  f=$(find /tmp/tl -name "*.db" | head -1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/timeline.py $f 20 ${2:-60}
done
