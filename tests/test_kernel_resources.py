# -*- coding: utf-8 -*-
"""CPU-side check of the compiled kernels' resources (hipcc cross-compiles gfx950 without a GPU): no kernel of the
library may use scratch memory.  Round 2 shipped dgrad kernels with 322 spilled VGPRs / 648 B per lane of scratch (an
opt-in epilogue compiled into every instance): 2.7x the algorithmic HBM traffic.  `-Rpass-analysis=kernel-resource-usage`
is what `scripts/kernel_resources.py` parses."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from kernel_resources import kernel_resources  # noqa: E402

CSRC = os.path.join(ROOT, 'yolov4_amd', 'csrc')


@pytest.mark.parametrize('src,flags', [('conv_f16x2.hip', ()), ('conv_planes.hip', ()), ('conv_tile.hip', ()), ('pointwise.hip', ()),
                                        ('yolo_head.hip', ('-ffp-contract=off',))])
def test_no_kernel_uses_scratch(src, flags):
    path = os.path.join(CSRC, src)
    if not os.path.isfile(path):
        pytest.skip(f'{src} not in this tree')
    rows = kernel_resources(path, flags)
    assert rows, 'no kernels parsed'
    bad = [(r['name'].split('(')[0], r['scratch'], r['vgpr_spill']) for r in rows if r['scratch'] != 0 or r['vgpr_spill'] != 0]
    assert not bad, bad
    # the kernels the bs=64 training step launches most are in the table and within the 2-blocks-per-CU budget
    if src == 'conv_f16x2.hip':
        names = {r['name'].split('(')[0]: r for r in rows}
        for k in ('void conv_gather_f16x2<128, 128, 2, 2, true, 16>', 'void conv_gather_f16x2<128, 128, 2, 2, false, 16>',
                  'void conv3x3_halo_f16x2<128, true, 16>', 'void conv3x3_halo_f16x2<128, false, 16>',
                  'void conv_wgrad_f16x2<128, 128, 16>'):
            assert k in names, k
            assert names[k]['vgprs'] + names[k]['agprs'] <= 256, (k, names[k])
    if src == 'conv_planes.hip':                      # one 8-wave block per CU: 256 registers per lane
        for r in rows:
            if 'planes_mfma' in r['name']:
                assert r['vgprs'] + r['agprs'] <= 256 and r['occupancy'] >= 2, r
