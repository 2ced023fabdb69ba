# -*- coding: utf-8 -*-
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (one stream): per kernel symbol, the gap in FRONT of
its launches, and the busy / idle split of the steady part of the run (the last `frac` of the launches).

    python scripts/trace_gaps.py <kernel_trace.csv> [frac=0.6]
"""
import collections
import csv
import re
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.split(r'\(', n)[0][:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    rows = rows[int(len(rows) * (1 - frac)):]
    gaps, dur = collections.defaultdict(list), collections.defaultdict(list)
    prev_end = None
    for r in rows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        k = short(r['Kernel_Name'])
        if prev_end is not None:
            gaps[k].append(max(0, s - prev_end) / 1e3)
        dur[k].append((e - s) / 1e3)
        prev_end = max(prev_end or 0, e)
    span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
    busy = sum(sum(v) for v in dur.values())
    idle = sum(sum(v) for v in gaps.values())
    print(f'launches {len(rows)}  span {span / 1e3:.2f} ms  busy {busy / 1e3:.2f} ms  idle {idle / 1e3:.2f} ms ({100 * idle / span:.1f} %)')
    print(f'{"kernel":60s} {"n":>6s} {"dur med":>8s} {"dur sum":>9s} {"gap med":>8s} {"gap sum":>9s}   (us, ms)')
    for k in sorted(dur, key=lambda k: -(sum(dur[k]) + sum(gaps.get(k, [0])))):
        d, g = sorted(dur[k]), sorted(gaps.get(k, [0]))
        print(f'{k:60s} {len(d):6d} {d[len(d) // 2]:8.1f} {sum(d) / 1e3:9.2f} {g[len(g) // 2]:8.2f} {sum(g) / 1e3:9.2f}')


if __name__ == '__main__':
    main()
