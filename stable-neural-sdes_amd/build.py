"""Build libsnsde.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
OUT = os.path.join(HERE, 'libsnsde.so')


def hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (expected /opt/rocm/bin/hipcc)')


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    srcs = glob.glob(os.path.join(CSRC, '*')) + [os.path.join(INCLUDE, 'snsde.h')]
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-I', INCLUDE, '-I', CSRC,
           '-o', OUT + '.tmp'] + srcs
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
    os.replace(OUT + '.tmp', OUT)
    return OUT


if __name__ == '__main__':
    print(build(force=True, verbose=True))
