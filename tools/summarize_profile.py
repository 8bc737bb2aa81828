"""Copy a rocprofv3 --kernel-trace --stats run's kernel_stats.csv into profiles/ with a readable summary.

    python tools/summarize_profile.py gpurun_out/prof2 profiles/r01_bench_cfg2 "command line that was profiled"
"""
import csv
import glob
import shutil
import sys

src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
fn = glob.glob(src + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(fn)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
shutil.copy(fn, dst + '_kernel_stats.csv')
with open(dst + '_summary.md', 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats\n\ncommand: `{cmd}`\n\nsum of kernel time: {tot / 1e6:.3f} ms\n\n')
    f.write('| kernel | calls | avg us | min us | max us | % of kernel time |\n|---|---|---|---|---|---|\n')
    for r in rows[:40]:
        f.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
                f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / tot * 100:.1f} |\n")
print('wrote', dst + '_summary.md')
