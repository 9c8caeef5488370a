#!/bin/bash
# Samples the GPU's shader clock and power while tools/prof_pbs.py runs a wide bootstrap batch (what clock does the
# kernel really run at?).  usage: tools/clock_watch.sh [gates]
G=${1:-65536}
(for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.1; done) > gpurun_out/clock_watch.log 2>&1 &
W=$!
timeout -k 10 120 python3 tools/prof_pbs.py $G
kill $W 2>/dev/null
sort gpurun_out/clock_watch.log | uniq -c | sort -rn | head -12
