#!/bin/bash
# kernel timeline of one step under the current default schedule (and option overrides passed as key=value)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -- python $GRAFT_REPO_ROOT/tools/run_steps.py 30 "$@" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/tl -name "*.db" | head -1); [ -n "$f" ] && python tools/timeline.py $f 20 70
