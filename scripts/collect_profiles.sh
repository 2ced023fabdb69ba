#!/bin/bash
# Collects the round's measurement artefacts on a 1-GPU MI355X box (run from the repo root through gpurun):
#   bash scripts/collect_profiles.sh <step> [<step> ...]
# steps: bench | stats | fetch | write | mfma | infer | ifetch | iwrite | dist | bf16 | bf16stats | bf16mfma
#
# Hygiene (VERDICT r3 item 9 / ADVICE r3): EVERY profiler pass runs in a directory of its own, named by round, step and UTC time
# (gpurun_out/<round>/<step>_<time>/), and leaves there its stdout, its stderr, the exact command (`cmd.txt`) and a `status.txt`
# with the exit code and what it means (0 = complete, 124 = our `timeout` expired, 137 = killed after -k, anything else = the
# tool or the program died).  Nothing is overwritten by a retry; a failed pass stops the script (no later GPU step is
# started behind a hung or killed one); scripts/publish_profiles.sh copies the NEWEST complete pass of each step into profiles/.
# Counter passes (--pmc) never share a pass with --stats / trace domains other than --kernel-trace.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
RD=${Y4_ROUND:-r04}
O=$R/gpurun_out/$RD
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
STEPS="--steps 5 --warmup 2 --no-cpu-baseline --no-infer-leg"
BF="--conv-mode bf16 --batch 128"

run_pass() {            # run_pass <step> <seconds> <command...>: one pass, its own directory, status recorded
  local step=$1 secs=$2; shift 2
  local d="$O/${step}_$(date -u +%Y%m%dT%H%M%SZ)"
  mkdir -p "$d"
  printf '%s\n' "$*" > "$d/cmd.txt"
  ( cd "$d" && timeout -k 10 "$secs" "$@" > "$d/stdout.txt" 2> "$d/stderr.txt" )
  local rc=$?
  local what="complete"
  [ $rc -eq 124 ] && what="our timeout ($secs s) expired: the pass hung or was too slow"
  [ $rc -eq 137 ] && what="killed (SIGKILL after the timeout's grace period)"
  [ $rc -ne 0 ] && [ $rc -ne 124 ] && [ $rc -ne 137 ] && what="the tool or the program exited with an error"
  echo "exit $rc: $what; finished $(date -u +%Y-%m-%dT%H:%M:%SZ)" > "$d/status.txt"
  echo "[$step] $(cat "$d/status.txt") -> $d"
  LAST_DIR="$d"
  return $rc
}

for what in "$@"; do
  case $what in
    bench)     run_pass bench 400 python3 $R/bench.py || exit 1 ;;
    stats)     run_pass stats 400 rocprofv3 --kernel-trace --stats -d . -o train --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-infer-leg --conv-table conv_table.txt || exit 1 ;;
    fetch)     run_pass fetch 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d . -o f --output-format csv -- python3 $R/bench.py $STEPS || exit 1 ;;
    write)     run_pass write 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d . -o w --output-format csv -- python3 $R/bench.py $STEPS || exit 1 ;;
    # The SQ-counter pass: counters only for the kernels that have MFMAs to count, and 3 steps instead of 7.  In rounds 3 and 4 the
    # UNRESTRICTED pass (7 steps, ~11 600 serialized dispatches) hung in 4 of 11 attempts -- in round 4, with this script's
    # per-pass records, at dispatch 11 376 of 11 625 (counter file complete up to there), in the middle of the seventh identical
    # backward pass: inside the profiler's per-dispatch counter start / stop, not at a particular kernel of the library
    # (profiles/r04_profiler_passes/mfma_20261005T071003Z/).  ~1 000 serialized dispatches instead.
    mfma)      run_pass mfma 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "conv|wgrad" -d . -o m --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer-leg || exit 1 ;;
    infer)     run_pass infer 400 rocprofv3 --kernel-trace --stats -d . -o infer --output-format csv -- python3 $R/bench.py --infer --steps 10 --warmup 2 || exit 1 ;;
    ifetch)    run_pass ifetch 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d . -o f --output-format csv -- python3 $R/bench.py --infer --steps 3 --warmup 1 || exit 1 ;;
    iwrite)    run_pass iwrite 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d . -o w --output-format csv -- python3 $R/bench.py --infer --steps 3 --warmup 1 || exit 1 ;;
    dist)      export Y4_FORCE_DIST=1
               run_pass dist 400 rocprofv3 --kernel-trace -d . -o dist --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-infer-leg --ddp-timeline || exit 1
               unset Y4_FORCE_DIST ;;
    bf16)      run_pass bf16 400 python3 $R/bench.py $BF --steps 6 --warmup 2 --no-cpu-baseline --no-infer-leg || exit 1 ;;
    bf16stats) run_pass bf16stats 400 rocprofv3 --kernel-trace --stats -d . -o train --output-format csv -- python3 $R/bench.py $BF --steps 6 --warmup 2 --no-cpu-baseline --no-infer-leg --conv-table conv_table.txt || exit 1 ;;
    bf16mfma)  run_pass bf16mfma 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "conv|wgrad" -d . -o m --output-format csv -- python3 $R/bench.py $BF --steps 2 --warmup 1 --no-cpu-baseline --no-infer-leg || exit 1 ;;
    *) echo "unknown step $what"; exit 2 ;;
  esac
  sleep 3
done
ls "$O"
