"""Per-kernel MFMA pipe utilisation from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass
(counter_collection.csv): busy cycles summed over the 1024 SIMDs / (128 SIMDs per XCD x GRBM_GUI_ACTIVE summed over the
8 XCDs), MI355X_MICROARCH.md's recipe.
usage: python scripts/pmc_mfma_util.py counter_collection.csv out.json"""
import collections, csv, json, re, sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\((y4::|StemGeom|StemWgradGeom|float|int|unsigned|const).*$', '', name).strip()
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in disp[name]:
        disp[name].add(r['Dispatch_Id'])
        dur[name] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
out = {}
for name in sorted(acc, key=lambda k: -dur[k]):
    c = acc[name]
    if not c.get('GRBM_GUI_ACTIVE') or 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or c['SQ_VALU_MFMA_BUSY_CYCLES'] == 0:
        continue
    n = len(disp[name])
    out[name] = {'launches': n, 'avg_us_under_pmc': dur[name] / n / 1e3,
                 'mfma_util': c['SQ_VALU_MFMA_BUSY_CYCLES'] / (128.0 * c['GRBM_GUI_ACTIVE'])}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
for k in list(out)[:16]:
    print(f"{k[:72]:72s} n={out[k]['launches']:4d} avg {out[k]['avg_us_under_pmc']:8.1f} us  mfma-util {out[k]['mfma_util']:.3f}")
