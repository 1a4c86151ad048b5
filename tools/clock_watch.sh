#!/bin/bash
# Sample GPU clock / power while a command runs (GPU box): tools/clock_watch.sh <out.log> -- <cmd...>
out=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > "$out" &
wpid=$!
"$@"
rc=$?
kill $wpid
exit $rc
