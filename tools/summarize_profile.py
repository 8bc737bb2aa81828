"""Copy a rocprofv3 --kernel-trace --stats run's kernel_stats.csv into profiles/ with a readable summary.

    python tools/summarize_profile.py gpurun_out/prof2 profiles/r01_bench_cfg2 "command line that was profiled" [skip]

The summary's `avg us` is taken over the per-dispatch trace (kernel_trace.csv) with the first `skip` launches of every
kernel dropped (default 2: the warm-up steps of bench.py run with cold instruction caches, first-touch page faults and
lazily loaded code objects, and used to pull the average 4-5 % above what bench.py's event timing of the timed region
reports).  Kernels with <= skip launches keep all of them.  The untouched rocprofv3 statistics (all launches) are copied
beside the summary as *_kernel_stats.csv.
"""
import collections
import csv
import glob
import shutil
import sys

src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 2
fn = glob.glob(src + '/*/*kernel_stats.csv')[0]
shutil.copy(fn, dst + '_kernel_stats.csv')
trace = glob.glob(src + '/*/*kernel_trace.csv')
per = collections.OrderedDict()
if trace:
    rows = list(csv.DictReader(open(trace[0])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    for r in rows:
        per.setdefault(r['Kernel_Name'], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
else:  # no per-dispatch trace in the output directory: fall back to rocprofv3's own aggregate (all launches)
    skip = 0
    for r in csv.DictReader(open(fn)):
        per[r['Name']] = [float(r['AverageNs']) / 1e3] * int(r['Calls'])
stats = []
for name, us in per.items():
    kept = us[skip:] if len(us) > skip else us
    stats.append((name, len(us), len(kept), sum(kept) / len(kept), min(kept), max(kept), sum(kept)))
stats.sort(key=lambda t: -t[6])
tot = sum(t[6] for t in stats)
with open(dst + '_summary.md', 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats\n\ncommand: `{cmd}`\n\n')
    f.write(f'per-kernel statistics over the dispatch trace with the first {skip} launches of each kernel dropped '
            f'(warm-up); sum of the kept kernel time: {tot / 1e3:.3f} ms\n\n')
    f.write('| kernel | calls (kept) | avg us | min us | max us | % of kernel time |\n|---|---|---|---|---|---|\n')
    for name, n, k, avg, mn, mx, total in stats[:40]:
        f.write(f'| `{name[:110]}` | {n} ({k}) | {avg:.1f} | {mn:.1f} | {mx:.1f} | {total / tot * 100:.1f} |\n')
print('wrote', dst + '_summary.md')
