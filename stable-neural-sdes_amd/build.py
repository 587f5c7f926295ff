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


def build(force=False, verbose=False, defines=(), out=None):
    """Compile every csrc/*.hip to an object file (in parallel: the MFMA kernels are split by hidden size) and
    link libsnsde.so."""
    from concurrent.futures import ThreadPoolExecutor
    out = out or OUT
    if not force and out == OUT and not needs_build():
        return out
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    objdir = os.path.join(HERE, 'build_' + ('_'.join(d.strip('-D') for d in defines) or 'obj'))
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    base = [cc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', INCLUDE, '-I', CSRC] + list(defines)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        cmd = base + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for ' + src + ':\n' + r.stdout + r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out + '.tmp'] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout + r.stderr)
    os.replace(out + '.tmp', out)
    return out


if __name__ == '__main__':
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == 'trace':
        print(build(force=True, defines=('-DSNSDE_TRACE',), out=os.path.join(HERE, 'libsnsde_trace.so')))
    else:
        print(build(force=True, verbose=True))
