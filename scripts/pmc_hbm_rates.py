# -*- coding: utf-8 -*-
"""Achieved HBM rate per kernel symbol: bytes per launch from the PMC passes (scripts/pmc_traffic.py output: 2 x FETCH_SIZE +
WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md) divided by the average launch duration of the SAME command's
`rocprofv3 --kernel-trace --stats` run (durations under --pmc are inflated by the counter reads and are not used).

    python scripts/pmc_hbm_rates.py <pmc_hbm_traffic_per_kernel.json> <kernel_stats.csv> <out.json>
"""
import csv
import json
import re
import sys


def norm(n):
    n = re.sub(r'\(anonymous namespace\)::|^void ', '', n.strip())
    return n.split('(')[0]


def main():
    traffic = json.load(open(sys.argv[1]))
    stats = {}
    for r in csv.DictReader(open(sys.argv[2])):
        stats[norm(r['Name'])] = (float(r['AverageNs']), int(r['Calls']), float(r['TotalDurationNs']))
    out = {}
    rows = []
    for k, v in traffic.items():
        name = norm(k)
        if name not in stats or 'hbm_bytes_per_launch_corrected' not in v:
            continue
        avg_ns, calls, tot = stats[name]
        b = v['hbm_bytes_per_launch_corrected']
        out[name] = {'hbm_bytes_per_launch_pmc': b, 'avg_launch_us_kernel_trace': avg_ns / 1e3, 'launches_in_trace': calls,
                     'achieved_TBps': b / avg_ns / 1e3, 'frac_of_8TBps': b / avg_ns / 1e3 / 8.0}
        rows.append((tot, name))
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(f'{"kernel":58s} {"MB/launch":>10s} {"avg us":>9s} {"TB/s":>6s}')
    for _, name in sorted(rows, reverse=True):
        o = out[name]
        print(f"{name[:58]:58s} {o['hbm_bytes_per_launch_pmc'] / 1e6:10.1f} {o['avg_launch_us_kernel_trace']:9.1f} {o['achieved_TBps']:6.2f}")


if __name__ == '__main__':
    main()
