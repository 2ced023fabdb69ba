#!/bin/bash
# Copies the summaries worth keeping from gpurun_out/<round>/ (scratch, merged back by gpurun) into profiles/ (tracked).
# Run in the build container after scripts/collect_profiles.sh passes came back.  For every step the NEWEST pass whose
# status.txt says "exit 0" is published; the status line, command and stderr of EVERY pass of the round -- complete or not --
# go to profiles/<round>_profiler_passes/ (one small directory per pass), so that a pass that died keeps its evidence.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
RD=${Y4_ROUND:-r04}
O=$R/gpurun_out/$RD
P=$R/profiles
newest() {   # newest complete pass directory of a step, or nothing
  for d in $(ls -d $O/$1_2* 2>/dev/null | sort -r); do
    if grep -q '^exit 0' $d/status.txt 2>/dev/null; then echo $d; return; fi
  done
}
jsonline() { python3 - "$1" "$2" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{')][-1]
json.dump(json.loads(line), open(sys.argv[2], 'w'), indent=1)
PY
}
d=$(newest bench);  [ -n "$d" ] && jsonline $d/stdout.txt $P/${RD}_bench_bs64.json
d=$(newest stats)
if [ -n "$d" ]; then
  cp $d/train_kernel_stats.csv $P/${RD}_bench_bs64_kernel_stats.csv
  cp $d/conv_table.txt $P/${RD}_conv_table.txt
  jsonline $d/stdout.txt $P/${RD}_bench_bs64_under_rocprof_all_events.json
  python3 $R/scripts/trace_gaps.py $d/train_kernel_trace.csv 0.6 > $P/${RD}_trace_gaps.txt
  STATS=$d
fi
f=$(newest fetch); w=$(newest write)
if [ -n "$f" ] && [ -n "$w" ]; then
  python3 $R/scripts/pmc_traffic.py $f/f_counter_collection.csv $w/w_counter_collection.csv $P/${RD}_pmc_hbm_traffic_per_kernel.json > /dev/null
  [ -n "$STATS" ] && python3 $R/scripts/pmc_hbm_rates.py $P/${RD}_pmc_hbm_traffic_per_kernel.json $P/${RD}_bench_bs64_kernel_stats.csv $P/${RD}_pmc_hbm_rate_per_kernel.json > $P/${RD}_pmc_hbm_rate_per_kernel.txt
fi
d=$(newest mfma);   [ -n "$d" ] && python3 $R/scripts/pmc_mfma_util.py $d/m_counter_collection.csv $P/${RD}_pmc_mfma_util_per_kernel.json > /dev/null
d=$(newest infer)
if [ -n "$d" ]; then
  cp $d/infer_kernel_stats.csv $P/${RD}_infer_bs32_kernel_stats.csv
  jsonline $d/stdout.txt $P/${RD}_bench_infer_bs32.json
fi
f=$(newest ifetch); w=$(newest iwrite)
[ -n "$f" ] && [ -n "$w" ] && python3 $R/scripts/pmc_traffic.py $f/f_counter_collection.csv $w/w_counter_collection.csv $P/${RD}_infer_bs32_pmc_hbm_traffic_per_kernel.json > /dev/null
d=$(newest dist)
if [ -n "$d" ]; then
  jsonline $d/stdout.txt $P/${RD}_bench_force_dist_1rank.json
  python3 $R/scripts/ddp_overlap.py $d/dist_kernel_trace.csv $P/${RD}_ddp_1rank_rccl_trace_summary.json > /dev/null
fi
d=$(newest bf16);      [ -n "$d" ] && jsonline $d/stdout.txt $P/${RD}_bench_bf16_bs128.json
d=$(newest bf16stats)
if [ -n "$d" ]; then
  cp $d/train_kernel_stats.csv $P/${RD}_bench_bf16_bs128_kernel_stats.csv
  cp $d/conv_table.txt $P/${RD}_conv_table_bf16_bs128.txt
fi
d=$(newest bf16mfma);  [ -n "$d" ] && python3 $R/scripts/pmc_mfma_util.py $d/m_counter_collection.csv $P/${RD}_pmc_mfma_util_per_kernel_bf16_bs128.json > /dev/null
# every pass of the round, complete or not: status, command, stderr
mkdir -p $P/${RD}_profiler_passes
for d in $(ls -d $O/*_2* 2>/dev/null); do
  n=$(basename $d); mkdir -p $P/${RD}_profiler_passes/$n
  for f in status.txt cmd.txt stderr.txt; do [ -f $d/$f ] && cp $d/$f $P/${RD}_profiler_passes/$n/$f; done
done
for f in "$@"; do cp $O/$f $P/${RD}_$f; done
ls $P | grep "^${RD}_"
