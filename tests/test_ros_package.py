"""The droppable catkin package (ros/auv_particle_filter_hip) against the reference's launch interface.

tests/golden/auv_pf_launch_params.json holds the NAME / VALUE pairs of the reference's auv_pf.launch and the
code defaults of auv_pf.py's rospy.get_param calls (oracle/ref_harness/gen_launch_fixture.py; data only).
Every argument and parameter of the reference launch must exist here under the same name with the same default
/ value expression, every parameter the launch sets must be consumed by the node mirror (DEFAULT_PARAMS), and
the code defaults must be the reference's."""
import json
import os
import re
import xml.etree.ElementTree as ET

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'ros', 'auv_particle_filter_hip')


def _fixture():
    with open(os.path.join(ROOT, 'tests', 'golden', 'auv_pf_launch_params.json')) as f:
        return json.load(f)


def _launch():
    root = ET.parse(os.path.join(PKG, 'launch', 'auv_pf.launch')).getroot()
    args = {a.get('name'): a.get('default') for a in root.iter('arg')}
    node = next(root.iter('node'))
    params = {p.get('name'): {'value': p.get('value'), 'type': (p.get('type') or '').strip() or None}
              for p in node.iter('param')}
    return root, args, node, params


def _resolve(expr, args):
    """substitute $(arg x) with the argument defaults, recursively"""
    for _ in range(8):
        m = re.search(r'\$\(arg ([A-Za-z_]+)\)', expr)
        if not m:
            return expr
        expr = expr.replace(m.group(0), args[m.group(1)])
    raise AssertionError('unresolved ' + expr)


def test_launch_file_mirrors_every_reference_argument_and_parameter():
    ref = _fixture()
    root, args, node, params = _launch()
    for name, default in ref['args'].items():
        assert name in args, 'missing <arg> ' + name
        assert args[name] == default, (name, args[name], default)
    assert node.get('name') == ref['node']['name'] == 'auv_pf'
    assert next(root.iter('group')).get('ns') == ref['group_ns']
    for name, pv in ref['params'].items():
        assert name in params, 'missing <param> ' + name
        assert params[name]['value'] == pv['value'], (name, params[name], pv)
        assert params[name]['type'] == pv['type'], (name, params[name], pv)
        # and it resolves to the same string with the default arguments
        assert _resolve(params[name]['value'], args) == _resolve(pv['value'], ref['args'])
    # the only intended differences: the package and the executable
    assert (node.get('pkg'), node.get('type')) == ('auv_particle_filter_hip', 'ros_node.py')
    assert os.access(os.path.join(PKG, 'scripts', node.get('type')), os.X_OK)


def test_every_launch_parameter_is_consumed_by_the_node_with_the_reference_code_default():
    from smarc_navigation_amd.auv_pf import DEFAULT_PARAMS, parse_cov_string
    ref = _fixture()
    _, args, _, params = _launch()
    for name in params:
        assert name in DEFAULT_PARAMS, 'the launch file sets %s but the node never reads it' % name
    for name, default in ref['code_defaults'].items():
        assert name in DEFAULT_PARAMS, 'the reference reads %s, the mirror does not' % name
        if default is None:
            # no code default in the reference (the three covariance strings): the mirror falls back to the
            # launch default, which must parse the reference's way
            assert DEFAULT_PARAMS[name] == ref['args'][name]
            assert len(parse_cov_string(DEFAULT_PARAMS[name])) == 6
        else:
            assert DEFAULT_PARAMS[name] == default, (name, DEFAULT_PARAMS[name], default)
    # the launch values are usable by the node as they stand
    for name in ('init_covariance', 'motion_covariance', 'resampling_noise_covariance', 'mbes_sensor_offset'):
        assert len(parse_cov_string(_resolve(params[name]['value'], args))) == 6
    assert int(_resolve(params['particle_count']['value'], args)) == 50
    assert float(_resolve(params['measurement_std']['value'], args)) == 1.0


def test_package_manifest_declares_what_the_node_imports():
    man = ET.parse(os.path.join(PKG, 'package.xml')).getroot()
    deps = {d.text for d in man.iter('exec_depend')}
    for need in ('rospy', 'tf', 'tf2_ros', 'geometry_msgs', 'nav_msgs', 'std_msgs', 'sensor_msgs'):
        assert need in deps
    cm = open(os.path.join(PKG, 'CMakeLists.txt')).read()
    assert 'catkin_install_python' in cm and 'scripts/ros_node.py' in cm and 'launch' in cm
    # the wrapper module exposes the entry point the script calls
    src = open(os.path.join(ROOT, 'smarc_navigation_amd', 'ros_node.py')).read()
    assert re.search(r'^def main\(', src, re.M)
