#!/bin/bash
# round 4, run 6: device-side hand-off latency between two waves of one launch (tools/handoff_bench.hip)
source "$(dirname "$0")/_common.sh"
echo "== hand-off"; timeout -k 10 120 tools/_build/handoff_bench > $O/handoff.json 2> $O/handoff.err; echo "rc=$?"; cat $O/handoff.json
