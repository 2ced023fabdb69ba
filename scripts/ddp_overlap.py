"""Where the gradient-bucket collectives sit inside the backward pass, from a rocprofv3 --kernel-trace of
`Y4_FORCE_DIST=1 python3 bench.py --ddp-timeline` (kernel_trace.csv): for every RCCL kernel of the LAST step its start
relative to the step's first backward kernel, its duration, the kernels of other streams that ran during it, and the
exposed tail (end of the last collective - end of the last compute kernel of the backward pass).
usage: python scripts/ddp_overlap.py kernel_trace.csv [out.json]"""
import csv, json, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    r['n'] = re.sub(r'\(anonymous namespace\)::|^void ', '', r['Kernel_Name'])
    r['n'] = re.sub(r'\((y4::|float|int|unsigned|const|ncclDevKernelArgs).*$', '', r['n'])[:70]
rows.sort(key=lambda r: r['s'])
is_coll = lambda r: 'oneRankReduce' in r['n'] or 'ncclDevKernel' in r['n'] or 'nccl' in r['n'].lower()
# a step begins with the stem's forward kernel; take the last complete one
stems = [r['s'] for r in rows if r['n'].startswith('conv_stem_fwd')]
step_start, step_end = stems[-2], stems[-1]
step = [r for r in rows if step_start <= r['s'] < step_end]
colls = [r for r in step if is_coll(r)]
bwd_first = next(r for r in step if 'yolo_loss_bwd' in r['n'] or 'yolo_decode_bwd' in r['n'])
compute = [r for r in step if not is_coll(r) and r['s'] >= bwd_first['s']]
out = {'step_ms': (step_end - step_start) / 1e6, 'backward_ms': (max(r['e'] for r in compute) - bwd_first['s']) / 1e6, 'collectives': []}
for c in colls:
    during = {}
    for r in compute:
        ov = min(r['e'], c['e']) - max(r['s'], c['s'])
        if ov > 0 and r['Stream_Id'] != c['Stream_Id']:
            during[r['n']] = during.get(r['n'], 0) + ov
    top = sorted(during.items(), key=lambda kv: -kv[1])[:3]
    out['collectives'].append({'kernel': c['n'], 'stream': c['Stream_Id'], 'start_ms_after_backward_start': round((c['s'] - bwd_first['s']) / 1e6, 3),
                               'duration_ms': round((c['e'] - c['s']) / 1e6, 3),
                               'compute_kernels_running_meanwhile': [f'{k} ({v / 1e3:.0f} us)' for k, v in top]})
last_compute_end = max(r['e'] for r in compute)
out['exposed_tail_ms'] = round(max(0, max(c['e'] for c in colls) - last_compute_end) / 1e6, 3) if colls else None
out['compute_streams'] = sorted({r['Stream_Id'] for r in compute})
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
