"""Summarise a rocprofv3 --pmc run (rocpd sqlite output): per kernel, mean counters per dispatch and a few ratios.

    python tools/pmc_summary.py gpurun_out/pmc1/pmc_results.db [name-substring]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts
cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, rocprofv3 PMC slots).
"""
import collections
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    rows = db.execute("select kernel_name, counter_name, avg(value), count(*), avg(end-start) from counters_collection "
                      "group by kernel_name, counter_name").fetchall()
    d = collections.defaultdict(dict)
    for k, cn, v, n, dur in rows:
        if pat in k:
            d[k][cn] = v
            d[k]['_n'] = n
            d[k]['_us'] = dur / 1e3
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]['_us'] * kv[1]['_n']):
        print(f"{k[:90]}  n={v['_n']}  {v['_us']:.1f} us/launch")
        line = []
        gui = v.get('GRBM_GUI_ACTIVE')
        if gui:
            cyc = gui / 8.0
            line.append(f"clock {cyc / v['_us'] / 1e3:.2f} GHz")
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
                line.append(f"MFMA busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cyc * 100:.1f} % of SIMD-cycles")
        wc = v.get('SQ_WAVE_CYCLES')
        if wc:
            for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_INST_LDS',
                      'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_INST_CYCLES_VMEM'):
                if c in v:
                    line.append(f"{c[3:]} {v[c] / wc * 100:.1f} %")
        for c in sorted(v):
            if not c.startswith('_') and not c.startswith('SQ_W') and not c.startswith('SQ_ACTIVE'):
                line.append(f"{c}={v[c]:.4g}")
        print('    ' + '; '.join(line))


if __name__ == '__main__':
    main()
