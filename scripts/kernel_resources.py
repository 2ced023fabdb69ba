# -*- coding: utf-8 -*-
"""Register / scratch / LDS use of every kernel of one HIP source, as hipcc reports it for gfx950
(`-Rpass-analysis=kernel-resource-usage`, device-only compile; no GPU needed).

    python scripts/kernel_resources.py yolov4_amd/csrc/conv_f16x2.hip            # table
    python scripts/kernel_resources.py yolov4_amd/csrc/conv_f16x2.hip --scratch  # only kernels with scratch / spills

tests/test_kernel_resources.py asserts on the parsed table (no scratch in the kernels the training step launches)."""
import os
import re
import subprocess
import sys

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CXXFILT = 'c++filt'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_resources(src, extra_flags=()):
    """[{name (demangled), vgprs, agprs, sgprs, scratch, vgpr_spill, sgpr_spill, occupancy, lds}] for every kernel in src."""
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '--cuda-device-only',
           '-Rpass-analysis=kernel-resource-usage', '-I', os.path.join(ROOT, 'include'), *extra_flags,
           '-c', src, '-o', os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-4000:])
    blocks = re.split(r'remark: Function Name: ', r.stderr)[1:]
    rows = []
    for b in blocks:
        name = b.split(' ', 1)[0].strip()

        def num(key):
            m = re.search(r'remark:\s+' + re.escape(key) + r': (\d+)', b)
            return int(m.group(1)) if m else -1
        rows.append({'mangled': name, 'vgprs': num('VGPRs'), 'agprs': num('AGPRs'), 'sgprs': num('TotalSGPRs'),
                     'scratch': num('ScratchSize [bytes/lane]'), 'vgpr_spill': num('VGPRs Spill'),
                     'sgpr_spill': num('SGPRs Spill'), 'occupancy': num('Occupancy [waves/SIMD]'),
                     'lds': num('LDS Size [bytes/block]')})
    if rows:
        dem = subprocess.run([CXXFILT] + [x['mangled'] for x in rows], capture_output=True, text=True).stdout.split('\n')
        for x, d in zip(rows, dem):
            x['name'] = d.replace('(anonymous namespace)::', '')
    return rows


def main():
    src = sys.argv[1]
    only = '--scratch' in sys.argv
    flags = [a for a in sys.argv[2:] if a.startswith('-f') or a.startswith('-D')]
    print('vgpr agpr sgpr scratch vspill occ  lds   kernel')
    for x in kernel_resources(src, flags):
        if only and x['scratch'] == 0 and x['vgpr_spill'] == 0:
            continue
        print(f"{x['vgprs']:4d} {x['agprs']:4d} {x['sgprs']:4d} {x['scratch']:7d} {x['vgpr_spill']:6d} {x['occupancy']:3d} {x['lds']:5d}  "
              f"{x['name'].split('(')[0]}")


if __name__ == '__main__':
    main()
