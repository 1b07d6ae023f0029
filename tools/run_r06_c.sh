#!/bin/bash
# round 6, third GPU call: the whole GPU suite (no -x) on the builtin-fdot2 epilogue with the SPLIT+QKN experiment reverted
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06c; mkdir -p $out
( time timeout 1500 python -m pytest tests -q -m gpu --durations=12 ) > $out/gputests.log 2>&1
tail -30 $out/gputests.log
