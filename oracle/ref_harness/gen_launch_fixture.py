#!/usr/bin/env python3
"""Interface fixture of the node's launch file: the NAME / VALUE pairs of
/root/reference/auv_particle_filter/launch/auv_pf.launch (arguments with their defaults, the node's
pkg / type / name, every <param> with its value expression and type) as JSON.  Data only -- no text of the
launch file is kept.  Also records the code defaults auv_pf.py gives rospy.get_param (auv_pf.py:27-110),
found by a regular expression over the calls.  Re-run:  python oracle/ref_harness/gen_launch_fixture.py"""
import json
import os
import re
import sys
import xml.etree.ElementTree as ET

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
import manifest  # noqa: E402  (this directory: where to write, and the fixture hashes)

REF = '/root/reference/auv_particle_filter'
OUT = os.path.join(manifest.golden_dir(), 'auv_pf_launch_params.json')


def main():
    root = ET.parse(os.path.join(REF, 'launch', 'auv_pf.launch')).getroot()
    args = {a.get('name'): a.get('default') for a in root.iter('arg')}
    node = next(root.iter('node'))
    params = {p.get('name'): {'value': p.get('value'), 'type': (p.get('type') or '').strip() or None} for p in node.iter('param')}
    group = next(root.iter('group'))
    src = open(os.path.join(REF, 'scripts', 'auv_pf.py')).read()
    code = {}
    for m in re.finditer(r"rospy\.get_param\(\s*['\"]~?([A-Za-z_]+)['\"]\s*(?:,\s*([^)]+?))?\s*\)", src):
        name, default = m.group(1), m.group(2)
        if default is not None:
            default = default.strip()
            if default[:1] in '\'"':
                default = default[1:-1]
            else:
                try:
                    default = json.loads(default)
                except ValueError:
                    pass
        code[name] = default
    out = {'args': args, 'node': {'pkg': node.get('pkg'), 'type': node.get('type'), 'name': node.get('name')},
           'group_ns': group.get('ns'), 'params': params, 'code_defaults': code}
    with open(OUT, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    manifest.record(os.path.dirname(OUT), ['auv_pf_launch_params.json'], 'oracle/ref_harness/gen_launch_fixture.py', needs_reference=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
