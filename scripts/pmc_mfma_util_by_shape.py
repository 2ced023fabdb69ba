# -*- coding: utf-8 -*-
"""MFMA pipe utilisation per (kernel symbol, grid size) -- i.e. per layer shape of scripts/planes_micro.py -- from a
`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass: busy cycles over the 1024 SIMDs / (128 SIMDs per XCD x
GRBM_GUI_ACTIVE summed over the 8 XCDs), as scripts/pmc_mfma_util.py does per symbol.

    python scripts/pmc_mfma_util_by_shape.py counter_collection.csv
"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(anonymous namespace\)::|^void ', '', r['Kernel_Name']).split('(')[0]
    key = (name, int(r['Grid_Size']))
    acc[key][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in n[key]:
        n[key].add(r['Dispatch_Id'])
        dur[key] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
print(f'{"kernel":40s} {"grid":>9s} {"blocks":>7s} {"launches":>8s} {"avg us":>8s} {"mfma util":>9s}')
for key in sorted(acc):
    c = acc[key]
    if not c.get('GRBM_GUI_ACTIVE') or not c.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        continue
    wg = 512 if ('256, 128, 4, 2' in key[0] or 'wgrad_planes' in key[0]) else 256
    print(f'{key[0][:40]:40s} {key[1]:9d} {key[1] // wg:7d} {len(n[key]):8d} {dur[key] / len(n[key]) / 1e3:8.1f} '
          f'{c["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * c["GRBM_GUI_ACTIVE"]):9.3f}')
