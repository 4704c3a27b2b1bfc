#!/usr/bin/env python3
"""tools/bench_summary.py <bench line file | /dev/stdin> -- the few numbers of a bench.py JSON line one looks at: value, ms per step,
the dominant launch, every extra leg with its hand-overs and per-phase microseconds."""
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print({k:d[k] for k in ['value','ms_per_step','first_block_ms']}, 'main_us', d['roofline']['launch_us'], 'frac', d['roofline']['frac'])
print({k:d[k] for k in d if k.startswith('value_') and not k.endswith('_what') and k != 'value_definition'})
for k,v in d.get('extra',{}).items():
    if isinstance(v,dict): print('%-28s'%k, v.get('ms_per_step'), v.get('steps_per_s'), v.get('mbes_path',{}).get('particles_handed_to_traversal'), {a:round(b*1000,1) for a,b in (v.get('kernels') or {}).items()})
print({k:round(v['avg_ms']*1000,1) for k,v in d.get('kernels').items()})
cb = dict(d.get('cpu_baseline') or {})
cb.pop('reference_python', None)
print(cb)
