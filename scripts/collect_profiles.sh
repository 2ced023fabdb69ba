#!/bin/bash
# Collects the round's measurement artefacts on a 1-GPU MI355X box (run from the repo root through gpurun):
# kernel stats + PMC passes of the training bench, the inference leg, and the one-rank RCCL rehearsal.
# Outputs land in gpurun_out/r03/ (every pass keeps its own stderr file); the summaries worth keeping are copied into profiles/ by hand.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${Y4_ROUND:-r03}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEPS="--steps 5 --warmup 2 --no-cpu-baseline"
KSEL=${Y4_PMC_KERNELS:-}      # e.g. --kernel-include-regex conv
what=${1:-all}

if [ $what = mfma ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE $KSEL -d $O/mfma -o m --output-format csv -- python3 $R/bench.py $STEPS > /dev/null 2> $O/mfma.err || exit 1
  python3 $R/scripts/pmc_mfma_util.py $(find $O/mfma -name "*counter_collection.csv") $O/pmc_mfma_util_per_kernel.json > $O/mfma.txt
  echo "mfma done"
fi
if [ $what = all ] || [ $what = train ] || [ $what = pmc ]; then
  [ $what = pmc ] || timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o train --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --conv-table $O/conv_table.txt > $O/train_stats.json 2> $O/train_stats.err || exit 1
  echo "stats done"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE $KSEL -d $O/fetch -o f --output-format csv -- python3 $R/bench.py $STEPS > /dev/null 2> $O/fetch.err || exit 1
  sleep 5
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE $KSEL -d $O/write -o w --output-format csv -- python3 $R/bench.py $STEPS > /dev/null 2> $O/write.err || exit 1
  echo "traffic done"
  # (the third counter pass of one session -- SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- died or hung at tool start-up three
  # times out of four when it followed the two traffic passes in the same gpurun call, with or without a pause, and never
  # when it ran first: it is its own step now, `collect_profiles.sh mfma`, to be run in a separate call)
  python3 $R/scripts/pmc_traffic.py $(find $O/fetch -name "*counter_collection.csv") $(find $O/write -name "*counter_collection.csv") $O/pmc_hbm_traffic_per_kernel.json > $O/traffic.txt
fi
if [ $what = all ] || [ $what = infer ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/istats -o infer --output-format csv -- python3 $R/bench.py --infer --steps 10 --warmup 2 > $O/infer_stats.json 2> $O/infer_stats.err || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/ifetch -o f --output-format csv -- python3 $R/bench.py --infer --steps 3 --warmup 1 > /dev/null 2> $O/ifetch.err || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/iwrite -o w --output-format csv -- python3 $R/bench.py --infer --steps 3 --warmup 1 > /dev/null 2> $O/iwrite.err || exit 1
  python3 $R/scripts/pmc_traffic.py $(find $O/ifetch -name "*counter_collection.csv") $(find $O/iwrite -name "*counter_collection.csv") $O/infer_pmc_hbm_traffic_per_kernel.json > $O/itraffic.txt
  echo "infer done"
fi
if [ $what = all ] || [ $what = dist ]; then
  export Y4_FORCE_DIST=1
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/dist -o dist --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --ddp-timeline > $O/dist_bench.json 2> $O/dist.err || exit 1
  unset Y4_FORCE_DIST
  echo "dist done"
fi
ls $O
