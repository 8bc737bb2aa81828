"""Prints launches / time per WGAN-GP step from the kernel-stats CSVs tools/profile_gan.sh leaves (steps = timed + warm-up)."""
import csv
import sys

for f, steps in ((sys.argv[1] + '/k4096.csv', 12), (sys.argv[1] + '/k8.csv', 22)):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows)
    print(f, 'kernel ms/step %.3f' % (tot / steps / 1e6), 'launches/step %.1f' % (calls / steps))
    for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
        print('%-84s calls/step %6.1f  avg us %8.2f  ms/step %7.3f  %5.1f%%' % (
            r['Name'][:84], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / steps / 1e6,
            float(r['Percentage'])))
