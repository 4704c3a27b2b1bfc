#!/usr/bin/env python3
"""tests/golden/MANIFEST.json -- one SHA-256 per committed fixture, written by the generator that made it.

TEST INFRASTRUCTURE.  Every gen_*.py of this directory ends with manifest.record(...): the hash of what it has just
written goes into the manifest beside the name of the script (and the case) it came from.  tests/test_golden_manifest.py
checks, on every CPU run, that the committed fixtures are exactly what the manifest says -- no file without an entry, no
entry without a file, no drift -- and re-runs the generators into a scratch directory (MCL_GOLDEN_OUT) to compare bit for
bit: all the reference-derived ones where /root/reference is present (this container), a quick subset of the self-oracle
MBES cases everywhere.  `python oracle/ref_harness/manifest.py --regen-check` re-runs EVERY generator (about ten minutes:
the independent MBES implementation samples its rays densely) and reports fixture by fixture.

The hash is taken over the CONTENT of a fixture, not over its container: for an .npz the sorted (name, dtype, shape,
C-order bytes) of its arrays -- independent of zip timestamps and compression level; for anything else the file's bytes."""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
GOLDEN = os.path.join(REPO, 'tests', 'golden')
NAME = 'MANIFEST.json'


def golden_dir():
    """Where the generators write: tests/golden, or the scratch directory of a regeneration check."""
    out = os.environ.get('MCL_GOLDEN_OUT') or GOLDEN
    os.makedirs(out, exist_ok=True)
    return out


def content_sha256(path):
    h = hashlib.sha256()
    if path.endswith('.npz'):
        import numpy as np
        with np.load(path, allow_pickle=False) as z:
            for key in sorted(z.files):
                a = np.ascontiguousarray(z[key])
                h.update(('%s|%s|%s|' % (key, a.dtype.str, a.shape)).encode())
                h.update(a.tobytes())
    else:
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def load(directory=GOLDEN):
    p = os.path.join(directory, NAME)
    if not os.path.exists(p):
        return {}
    with open(p) as f:
        return json.load(f)


def record(directory, files, generator, needs_reference):
    """Called by a generator for the files it has just written into `directory`."""
    m = load(directory)
    for name in files:
        m[name] = {'sha256': content_sha256(os.path.join(directory, name)), 'generator': generator,
                   'needs_reference': bool(needs_reference)}
    with open(os.path.join(directory, NAME), 'w') as f:
        json.dump(m, f, indent=1, sort_keys=True)
        f.write('\n')


def fixtures(directory=GOLDEN):
    return sorted(f for f in os.listdir(directory) if f != NAME and not f.startswith('.'))


def verify(directory=GOLDEN, against=None):
    """-> list of problems (empty: every fixture of `directory` is what `against` -- default: its own manifest -- says)."""
    m = load(GOLDEN if against is None else against)
    bad = []
    for name in fixtures(directory):
        if name not in m:
            bad.append('%s: not in the manifest' % name)
        elif content_sha256(os.path.join(directory, name)) != m[name]['sha256']:
            bad.append('%s: content differs from the manifest (generator: %s)' % (name, m[name]['generator']))
    return bad


# (script, arguments, needs /root/reference)
GENERATORS = (('gen_golden.py', (), True), ('gen_golden_dr.py', (), True), ('gen_golden_stats.py', (), True),
              ('gen_launch_fixture.py', (), True), ('gen_golden_mbes.py', (), False))


def regenerate(out_dir, scripts=None, mbes_cases=None):
    """Run generators into out_dir (their own manifest goes there too).  scripts: names to run (default: all that can run
    here); mbes_cases: restrict gen_golden_mbes.py to these cases."""
    env = dict(os.environ, MCL_GOLDEN_OUT=out_dir)
    have_ref = os.path.isdir('/root/reference')
    for script, args, needs_ref in GENERATORS:
        if scripts is not None and script not in scripts:
            continue
        if needs_ref and not have_ref:
            continue
        if script == 'gen_golden_mbes.py' and mbes_cases is not None:
            args = tuple(mbes_cases)
        subprocess.run([sys.executable, os.path.join(HERE, script)] + list(args), env=env, check=True,
                       stdout=subprocess.DEVNULL)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--regen-check':
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            regenerate(tmp)
            committed, fresh = load(GOLDEN), load(tmp)
            ok = True
            for name in sorted(committed):
                if name not in fresh:
                    print('%-36s not regenerated here (%s)' % (name, committed[name]['generator']))
                elif fresh[name]['sha256'] == committed[name]['sha256']:
                    print('%-36s identical' % name)
                else:
                    print('%-36s DIFFERS' % name)
                    ok = False
            sys.exit(0 if ok else 1)
    bad = verify()
    for b in bad:
        print(b)
    print('%d fixtures, %d problems' % (len(fixtures()), len(bad)))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
