#!/bin/bash
# Sample the GPU's shader clock and package power while a command runs (is the launch clock-limited by power?).
#   tools/clock_probe.sh <logfile> <command...>
log=$1; shift
"$@" > "$log.cmd" 2>&1 &
pid=$!
: > "$log"
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power|Average Graphics" >> "$log"
  echo "--" >> "$log"
  sleep 0.5
done
wait $pid
