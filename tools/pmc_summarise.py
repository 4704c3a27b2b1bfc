#!/usr/bin/env python3
"""tools/pmc_summarise.py -- turn rocprofv3 --pmc CSV passes into profiles/<round>_traffic.json (ROUND=r05).

    python tools/pmc_summarise.py <dir with pmc_<map>_{fetch,write,sq}/.../*_counter_collection.csv> [out.json]

Per map kind (the fan sweep k_mbes_sweep: <2,false> on the mesh, <0,false> on the grid) and for the dominant MBES kernel: mean FETCH_SIZE / WRITE_SIZE per dispatch (KB),
HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes (gfx950 reports half of a wide
streaming read: MI355X_MICROARCH.md, HBM section; calibrated in round 1 on k_predict), VALU
wave-instructions per launch, VALU lane utilisation (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)) and
the share of wave cycles spent waiting.  The file is stamped with the hash of the kernel sources
(bench.py:source_hash) so that bench.py attaches it only to a library built from the same sources."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rows(path_glob):
    for p in glob.glob(path_glob, recursive=True):
        with open(p) as f:
            for r in csv.DictReader(f):
                yield r


def mean_counter(d, pass_name, kernel_prefix, counter):
    vals = {}
    for r in rows(os.path.join(d, pass_name, '**', '*counter_collection.csv*')):
        if r['Kernel_Name'].startswith(kernel_prefix) and r['Counter_Name'] == counter:
            vals.setdefault(r['Dispatch_Id'], 0.0)
            vals[r['Dispatch_Id']] += float(r['Counter_Value'])
    v = list(vals.values())
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def main():
    import bench
    d = sys.argv[1]
    rnd = os.environ.get('ROUND', 'r05')
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'profiles', '%s_traffic.json' % rnd)
    res = {'_how': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (three separate passes) on '
                   '`python3 bench.py --map <m> --steps 4 --warmup 1 --only-main`, dominant kernel k_mbes_sweep<2,false,false> (mesh) / k_mbes_sweep<0,false,false> (grid): the fan sweep (one pass since round 4), '
                   'mean over dispatches; KB -> bytes x1024; FETCH_SIZE doubled (gfx950 reports half of a wide streaming read)',
           '_round': int(rnd.lstrip('r')), 'source_hash': bench.source_hash()}
    for kind, prefix in (('mesh', 'void k_mbes_sweep<2, false, false>'), ('grid', 'void k_mbes_sweep<0, false, false>')):
        e = {}
        f, nf = mean_counter(d, 'pmc_%s_fetch' % kind, prefix, 'FETCH_SIZE')
        w, nw = mean_counter(d, 'pmc_%s_write' % kind, prefix, 'WRITE_SIZE')
        if f is None or w is None:
            continue
        e['FETCH_SIZE_KB'], e['WRITE_SIZE_KB'], e['dispatches'] = f, w, nf
        e['traffic_bytes_per_launch'] = int(round((2.0 * f + w) * 1024.0))
        sq = {}
        for c in ('SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_THREAD_CYCLES_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY',
                  'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_WAVES', 'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES', 'SQ_INSTS_SALU'):
            v, _ = mean_counter(d, 'pmc_%s_sq' % kind, prefix, c)
            if v is not None:
                sq[c] = v
        if sq.get('SQ_INSTS_VALU'):
            e['valu_insts_per_launch'] = sq['SQ_INSTS_VALU']
        if sq.get('SQ_THREAD_CYCLES_VALU') and sq.get('SQ_ACTIVE_INST_VALU'):
            e['valu_lane_utilisation'] = sq['SQ_THREAD_CYCLES_VALU'] / (64.0 * sq['SQ_ACTIVE_INST_VALU'])
        if sq.get('SQ_WAIT_ANY') and sq.get('SQ_WAVE_CYCLES'):
            e['wait_frac_of_wave_cycles'] = sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']
        if sq.get('SQ_INSTS_LDS'):
            e['lds_insts_per_launch'] = sq['SQ_INSTS_LDS']
        if 'SQ_LDS_BANK_CONFLICT' in sq:
            e['lds_bank_conflict_cycles'] = sq['SQ_LDS_BANK_CONFLICT']
        e['sq_raw'] = sq
        # in-kernel average of the same kernel from the --kernel-trace --stats pass of the same command
        for p in glob.glob(os.path.join(d, 'stats_%s' % kind, '**', '*kernel_stats.csv'), recursive=True):
            with open(p) as f:
                for r in csv.DictReader(f):
                    if r['Name'].startswith(prefix):
                        e['kernel_avg_us'] = float(r['AverageNs']) / 1e3
                        e['kernel_calls'] = int(r['Calls'])
        res[kind] = e
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
