#!/bin/bash
# Copies the summaries worth keeping from gpurun_out/<round>/ (scratch, merged back by gpurun) into profiles/ (tracked).
# Run in the build container after scripts/collect_profiles.sh (+ planes_micro / bn_micro / pmc_lds runs) came back.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
RD=${Y4_ROUND:-r03}
O=$R/gpurun_out/$RD
P=$R/profiles
cp $O/stats/train_kernel_stats.csv $P/${RD}_bench_bs64_kernel_stats.csv
cp $O/conv_table.txt $P/${RD}_conv_table.txt
python3 - "$O/train_stats.json" "$P/${RD}_bench_bs64_under_rocprof_all_events.json" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{')][-1]
json.dump(json.loads(line), open(sys.argv[2], 'w'), indent=1)
PY
python3 $R/scripts/pmc_traffic.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $P/${RD}_pmc_hbm_traffic_per_kernel.json > /dev/null
MF=$O/mfma; [ -f $O/mfma2/m_counter_collection.csv ] && [ $O/mfma2/m_counter_collection.csv -nt $O/mfma/m_counter_collection.csv ] && MF=$O/mfma2
python3 $R/scripts/pmc_mfma_util.py $MF/m_counter_collection.csv $P/${RD}_pmc_mfma_util_per_kernel.json > /dev/null
python3 $R/scripts/pmc_hbm_rates.py $P/${RD}_pmc_hbm_traffic_per_kernel.json $P/${RD}_bench_bs64_kernel_stats.csv $P/${RD}_pmc_hbm_rate_per_kernel.json > $P/${RD}_pmc_hbm_rate_per_kernel.txt
python3 $R/scripts/trace_gaps.py $O/stats/train_kernel_trace.csv 0.6 > $P/${RD}_trace_gaps.txt
cp $O/istats/infer_kernel_stats.csv $P/${RD}_infer_bs32_kernel_stats.csv
cp $O/infer_pmc_hbm_traffic_per_kernel.json $P/${RD}_infer_bs32_pmc_hbm_traffic_per_kernel.json
python3 - "$O/dist_bench.json" "$P/${RD}_bench_force_dist_1rank.json" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith('{')][-1]
json.dump(json.loads(line), open(sys.argv[2], 'w'), indent=1)
PY
python3 $R/scripts/ddp_overlap.py $O/dist/dist_kernel_trace.csv $P/${RD}_ddp_1rank_rccl_trace_summary.json > /dev/null
for m in fwd wgrad; do [ -d $O/lds_pmc/${m}_g1 ] && python3 $R/scripts/pmc_lds_report.py $O/lds_pmc $m > $P/${RD}_pmc_lds_$m.txt; done
mkdir -p $P/${RD}_profiler_stderr
for f in train_stats fetch write mfma mfma2 infer_stats ifetch iwrite dist; do [ -f $O/$f.err ] && cp $O/$f.err $P/${RD}_profiler_stderr/$f.err; done
for f in "$@"; do cp $O/$f $P/${RD}_$f; done
ls $P | grep "^${RD}_"
