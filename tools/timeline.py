"""Per-kernel timeline of ONE training step from a rocprofv3 --kernel-trace CSV (diagnostic).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras
    python3 tools/timeline.py /tmp/tl > gpurun_out/timeline.txt

Prints, for the last complete step (from one label_attn_fwd launch to the next): start offset (us), duration, queue, gap to
the previous kernel's end on any queue, kernel name; then the busy / idle split of the step.
"""
import csv
import glob
import sys


def main():
    files = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)
    rows = list(csv.DictReader(open(files[0])))
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows))
    fw = [i for i, e in enumerate(ev) if 'label_attn_fwd' in e[2]]
    a, b = fw[-2], fw[-1]
    step = ev[a:b]
    t0 = step[0][0]
    last_end = t0
    busy = 0
    cover_end = t0
    for s, e, name, q in step:
        gap = s - last_end
        print(f'{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  q{q:>3}  gap {gap / 1e3:7.1f}  {name[:90]}')
        last_end = max(last_end, e)
        if e > cover_end:
            busy += e - max(s, cover_end)
            cover_end = e
    total = ev[b][0] - t0
    print(f'step {total / 1e3:.1f} us, some kernel running {busy / 1e3:.1f} us, idle {(total - busy) / 1e3:.1f} us')


if __name__ == '__main__':
    main()
