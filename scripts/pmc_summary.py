"""Per-kernel sums of the counters in a rocprofv3 --pmc counter_collection.csv, with the ratios used in DESIGN.md:
usage: python scripts/pmc_summary.py counter_collection.csv [name filter]"""
import collections, csv, re, sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(path)):
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\((y4::|StemGeom|StemWgradGeom|float|int|unsigned).*$', '', name).strip()
    if flt and flt not in name:
        continue
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    key = (name, r['Dispatch_Id'])
    if key not in seen:
        seen.add(key)
        n[name] += 1
        dur[name] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
for name in sorted(acc, key=lambda k: -dur[k])[:16]:
    c = acc[name]
    line = f'{name[:64]:64s} n={n[name]:4d} avg {dur[name] / n[name] / 1e3:8.1f} us'
    wc = c.get('SQ_WAVE_CYCLES')
    if wc:
        for k, lab in (('SQ_WAIT_ANY', 'wait'), ('SQ_WAIT_INST_ANY', 'issue-stall'), ('SQ_ACTIVE_INST_ANY', 'active'),
                       ('SQ_WAIT_INST_LDS', 'lds-stall'), ('SQ_ACTIVE_INST_VALU', 'valu'), ('SQ_ACTIVE_INST_LDS', 'lds'),
                       ('SQ_ACTIVE_INST_VMEM', 'vmem'), ('SQ_ACTIVE_INST_SCA', 'scalar'), ('SQ_LDS_BANK_CONFLICT', 'bank-conf')):
            if k in c:
                line += f' {lab} {c[k] / wc:.2f}'
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c and c['GRBM_GUI_ACTIVE']:
        # busy cycles summed over 1024 SIMDs / (GUI_ACTIVE summed over 8 XCDs x 128 SIMDs per XCD)
        line += f" mfma-util {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (128.0 * c['GRBM_GUI_ACTIVE']):.3f}"
    for k in ('SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_MFMA'):
        if k in c:
            line += f' {k[9:].lower()}/launch {c[k] / n[name]:.3g}'
    print(line)
    extra = ''
    if c.get('TCC_HIT_sum') is not None and (c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0)) > 0:
        extra += f" L2-hit {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f} L2-req/launch {(c['TCC_HIT_sum'] + c['TCC_MISS_sum']) / n[name]:.3g}"
    for k in ('TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum', 'TCC_REQ_sum', 'TCC_READ_sum', 'TCP_TCC_READ_REQ_sum', 'TCP_TOTAL_CACHE_ACCESSES_sum',
              'TCP_TCC_READ_REQ_LATENCY_sum', 'TCP_PENDING_STALL_CYCLES_sum', 'TCP_GATE_EN1_sum', 'TCP_GATE_EN2_sum', 'TCP_TA_TCP_STATE_READ_sum',
              'TA_BUSY_avr', 'TA_ADDR_STALLED_BY_TC_CYCLES_sum', 'TA_DATA_STALLED_BY_TC_CYCLES_sum', 'TCC_BUSY_avr', 'TCC_TAG_STALL_sum',
              'TCP_READ_TAGCONFLICT_STALL_CYCLES_sum', 'TCP_TCR_TCP_STALL_CYCLES_sum', 'FETCH_SIZE', 'WRITE_SIZE', 'MemUnitStalled', 'L2CacheHit'):
        if k in c:
            extra += f' {k} {c[k] / n[name]:.4g}'
    if extra:
        print('      ' + extra)
