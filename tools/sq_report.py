"""Prints per-kernel SQ counter averages and kernel durations from gpurun_out/sq_now + gpurun_out/trace_now (tools/sq_now.sh)."""
import collections, csv, glob
f = glob.glob('gpurun_out/sq_now/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'swfr::' in k:
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
dur = {}
for r in csv.DictReader(open(glob.glob('gpurun_out/trace_now/*/*kernel_stats.csv')[0])):
    dur[r['Name'].split('(')[0].replace('void ', '')] = float(r['AverageNs']) / 1000
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = max(m['SQ_WAVES'], 1)
    print('%-24s %6.1f us  waves %6d  valu/w %6.0f salu/w %6.0f lds/w %5.0f  cycles/w %6d  active %.2f' % (
        k, dur.get(k, 0), w, m['SQ_INSTS_VALU'] / w, m['SQ_INSTS_SALU'] / w, m['SQ_INSTS_LDS'] / w, 4 * m['SQ_WAVE_CYCLES'] / w,
        m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']))
