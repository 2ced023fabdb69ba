#!/bin/bash
# Samples socket power / clocks with rocm-smi while a command runs (diagnostics only).
# usage: scripts/power_probe.sh out.txt -- <command...>
out=$1; shift; shift
"$@" > "${out%.txt}.cmd.log" 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse --csv 2>/dev/null | tail -n +1 >> "$out"
  sleep 0.3
done
wait $pid
