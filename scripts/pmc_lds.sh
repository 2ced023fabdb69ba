#!/bin/bash
# SQ / LDS counters of the plane conv kernels on the layer shapes of scripts/planes_micro.py (one rocprofv3 --pmc pass per
# counter group; run from the repo root through gpurun).  Output: gpurun_out/<round>/lds_pmc/<group>/…counter_collection.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${Y4_ROUND:-r03}/lds_pmc
MODE=${1:-fwd}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "planes_mfma" -d $O/${MODE}_g$i -o p --output-format csv -- python3 $R/scripts/planes_micro.py 64 $MODE > $O/${MODE}_g$i.log 2> $O/${MODE}_g$i.err || exit 1
  echo "group $i done"
done
