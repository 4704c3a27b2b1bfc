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
    rnd = os.environ.get('ROUND', 'r06')
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'profiles', '%s_traffic.json' % rnd)
    res = {'_how': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (three separate passes) on '
                   '`python3 bench.py --map <m> --steps 4 --warmup 1 --only-main`, dominant kernel k_mbes_sweep<2,false,false> (mesh) / k_mbes_sweep<0,false,false> (grid): the fan sweep (one pass since round 4), '
                   'mean over dispatches; KB -> bytes x1024; FETCH_SIZE doubled (gfx950 reports half of a wide streaming read)',
           '_round': int(rnd.lstrip('r')), 'source_hash': bench.source_hash()}
    # (round 6: the adjacency walk over an irregular TIN -- the path of any real survey mesh -- with its cache counters, and the
    #  lattice sweep on a filter that keeps a healthy spread; keys = bench.py's map kinds, so that a line of --map mesh-tin
    #  finds its own entry)
    for kind, prefix in (('mesh', 'void k_mbes_sweep<2, false, false>'), ('grid', 'void k_mbes_sweep<0, false, false>'),
                         ('mesh-tin', 'void k_mbes_sweep<5, false, false>'), ('mesh_tempered', 'void k_mbes_sweep<2, false, false>'),
                         ('mesh-tin_tempered', 'void k_mbes_sweep<5, false, false>')):
        e = {}
        f, nf = mean_counter(d, 'pmc_%s_fetch' % kind, prefix, 'FETCH_SIZE')
        w, nw = mean_counter(d, 'pmc_%s_write' % kind, prefix, 'WRITE_SIZE')
        if f is None or w is None:
            if mean_counter(d, 'pmc_%s_sq' % kind, prefix, 'SQ_INSTS_VALU')[0] is None:
                continue
        else:
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
        # cache behaviour (round 6): L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS) (MI355X_MICROARCH.md, L2); vector-L1: requests
        # that went on to L2 per L1 access
        cache = {}
        for c in ('TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_REQ_sum', 'TCP_TOTAL_CACHE_ACCESSES_sum', 'TCP_TCC_READ_REQ_sum', 'TCP_TOTAL_ACCESSES_sum'):
            v, _ = mean_counter(d, 'pmc_%s_cache' % kind, prefix, c)
            if v is not None:
                cache[c] = v
        if cache.get('TCC_HIT_sum') is not None and cache.get('TCC_MISS_sum') is not None and cache['TCC_HIT_sum'] + cache['TCC_MISS_sum'] > 0:
            cache['l2_hit_rate'] = cache['TCC_HIT_sum'] / (cache['TCC_HIT_sum'] + cache['TCC_MISS_sum'])
        if cache.get('TCP_TOTAL_CACHE_ACCESSES_sum') and cache.get('TCP_TCC_READ_REQ_sum') is not None:
            cache['l1_read_requests_to_l2_per_l1_access'] = cache['TCP_TCC_READ_REQ_sum'] / cache['TCP_TOTAL_CACHE_ACCESSES_sum']
        if cache:
            e['cache'] = cache
        # in-kernel average of the same kernel from the --kernel-trace --stats pass of the same command
        for p in glob.glob(os.path.join(d, 'stats_%s' % kind, '**', '*kernel_stats.csv'), recursive=True):
            with open(p) as f:
                for r in csv.DictReader(f):
                    if r['Name'].startswith(prefix):
                        e['kernel_avg_us'] = float(r['AverageNs']) / 1e3
                        e['kernel_calls'] = int(r['Calls'])
        res[kind] = e
    # ---- the streaming kernels of the step: what stops each of them, from the same SQ pass on the mesh leg
    table = {}
    for name, prefix in (('k_predict_pose', 'void k_predict_pose<false>'), ('k_quantise_tiles', 'k_quantise_tiles'),
                         ('k_cdf_expand', 'void k_cdf_expand<true>'), ('k_resample_gather', 'void k_resample_gather<true, true, true>'),
                         ('k_visit_scan', 'k_visit_scan'), ('k_mbes_cast (empty hand-over launch)', 'void k_mbes_cast<')):
        row = {}
        for c in ('SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES',
                  'SQ_ACTIVE_INST_VALU', 'SQ_THREAD_CYCLES_VALU'):
            v, _ = mean_counter(d, 'pmc_mesh_sq', prefix, c)
            if v is None:
                v, _ = mean_counter(d, 'pmc_mesh_sq2', prefix, c)
            if v is not None:
                row[c] = v
        if not row:
            continue
        if row.get('SQ_WAVES'):
            row['valu_per_wave'] = row.get('SQ_INSTS_VALU', 0.0) / row['SQ_WAVES']
        if row.get('SQ_WAVE_CYCLES') and row.get('SQ_WAIT_ANY') is not None:
            row['wait_frac_of_wave_cycles'] = row['SQ_WAIT_ANY'] / row['SQ_WAVE_CYCLES']
        if row.get('SQ_THREAD_CYCLES_VALU') and row.get('SQ_ACTIVE_INST_VALU'):
            row['valu_lane_utilisation'] = row['SQ_THREAD_CYCLES_VALU'] / (64.0 * row['SQ_ACTIVE_INST_VALU'])
        for p in glob.glob(os.path.join(d, 'stats_mesh', '**', '*kernel_stats.csv'), recursive=True):
            with open(p) as f:
                for r in csv.DictReader(f):
                    if r['Name'].startswith(prefix):
                        row['kernel_avg_us'] = float(r['AverageNs']) / 1e3
        if row.get('kernel_avg_us') and row.get('SQ_INSTS_VALU'):
            # share of the chip's vector-issue slots (wave64 VALU = 2 cycles of a SIMD, 1 024 SIMDs at 2.4 GHz)
            row['valu_issue_frac'] = row['SQ_INSTS_VALU'] / (row['kernel_avg_us'] * 1e-6) / (1024 * 2.4e9 / 2.0)
        table[name] = row
    if table:
        res['streaming_kernels'] = table
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
