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
