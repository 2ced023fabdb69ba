# -*- coding: utf-8 -*-
"""Per (kernel, grid) medians of the counters collected by scripts/pmc_lds.sh, plus the derived ratios.

    python scripts/pmc_lds_report.py gpurun_out/r03/lds_pmc fwd
"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    root, mode = sys.argv[1], sys.argv[2]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(root, f'{mode}_g*'))):
        f = os.path.join(d, 'p_counter_collection.csv')
        if not os.path.isfile(f):
            continue
        for r in csv.DictReader(open(f)):
            name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']).split('(')[0]
            key = (name, int(r['Grid_Size']))
            vals[key][r['Counter_Name']].append(float(r['Counter_Value']))
            durs[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    med = lambda v: sorted(v)[len(v) // 2] if v else float('nan')
    names = sorted({c for k in vals for c in vals[k]})
    for key in sorted(vals):
        m = {c: med(vals[key][c]) for c in names}
        print(f'{key[0]} grid {key[1]}  ({len(durs[key]) // max(1, len(names))} launches)  dur {med(durs[key]):.1f} us (under the profiler)')
        for c in names:
            print(f'    {c:28s} {m[c]:16.0f}')
        busy = m.get('SQ_BUSY_CYCLES', float('nan'))
        # SQ_BUSY_CYCLES is summed over shader engines / XCDs; per-CU quantities are compared with SQ_BUSY_CU_CYCLES-like
        # normalisation below only through RATIOS of counters of the same scope
        if m.get('SQ_LDS_IDX_ACTIVE'):
            print(f"    -> bank-conflict cycles / LDS active cycles : {m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.3f}")
        if m.get('SQ_WAVE_CYCLES'):
            wc = m['SQ_WAVE_CYCLES']
            for c in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_ANY', 'SQ_LDS_IDX_ACTIVE',
                      'SQ_LDS_DATA_FIFO_FULL', 'SQ_LDS_CMD_FIFO_FULL', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INST_LEVEL_LDS'):
                if c in m:
                    print(f'    -> {c} / SQ_WAVE_CYCLES : {m[c] / wc:.3f}')
        _ = busy


if __name__ == '__main__':
    main()
