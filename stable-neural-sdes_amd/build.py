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


LEAN_RESOURCES = {}      # kernel name -> (VGPRs, scratch bytes per lane) of the lean 4-row-tile instantiations, last build


def check_lean_resources(src, remarks):
    """The lean kernel (csrc/snsde_m4_kernel.h) issues its LDS / global prefetches from inline asm and waits for them itself:
    the compiler believes their destination registers are written at the asm statement.  A register spill of one of them
    would save stale data, so every instantiation must compile WITHOUT scratch (csrc: lean_fits() keeps the register-heavy
    configurations on the general kernel).  Checked here from hipcc's kernel-resource-usage remarks; a violation fails
    the build instead of shipping a kernel that is wrong under register pressure."""
    import re
    name = None
    for line in remarks.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            continue
        if name and ('snsde_m4_kernel' in name or 'snsde_m4s_kernel' in name or 'snsde_m4s2_kernel' in name or 'snsde_m4s2_reverse_kernel' in name):
            m = re.search(r'remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill): (\d+)', line)
            if m:
                LEAN_RESOURCES.setdefault(name, {})[m.group(1)] = int(m.group(2))
    bad = {k: v for k, v in LEAN_RESOURCES.items() if v.get('ScratchSize [bytes/lane]', 0) or v.get('VGPRs Spill', 0)}
    if bad:
        raise RuntimeError(f'{os.path.basename(src)}: lean kernel instantiations spill registers (tighten lean_fits() in '
                           f'csrc/snsde_mfma_kernels.h): {bad}')


def build(force=False, verbose=False, defines=(), out=None, check_resources=True, incremental=False):
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
    # --offload-compress: the device code objects (~24 of the library's 27 MB uncompressed) are stored zstd-compressed
    # and inflated by the HIP runtime when the library is loaded
    # -fvisibility=hidden: only the SNSDE_API entry points of include/snsde.h are in the dynamic symbol table
    base = [cc, '--offload-arch=gfx950', '--offload-compress', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-I', INCLUDE, '-I', CSRC] + list(defines)

    hdr_time = max(os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(INCLUDE, 'snsde.h')])

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + '.o')
        if incremental and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_time):
            return obj          # (development: `python build.py inc`; the lean resource check only sees what was recompiled)
        lean = os.path.basename(src).startswith(('snsde_m4_h', 'snsde_m4s_h', 'snsde_m4s2_h', 'snsde_m4s2_rev_h'))
        cmd = base + (['-Rpass-analysis=kernel-resource-usage'] if lean else []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for ' + src + ':\n' + r.stdout + r.stderr)
        if lean and check_resources:
            check_lean_resources(src, r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    # dynamic symbol table = the SNSDE_API entry points declared in include/snsde.h, nothing else: -fvisibility=hidden covers the
    # internal launchers / dispatchers, the version script also localises the weak kernel-handle objects hipcc emits for template
    # kernels (they are only referenced from inside the library)
    import re
    # (only declarations: lines that start with SNSDE_API, not names mentioned in the header's prose)
    names = sorted(set(re.findall(r'^SNSDE_API\s[^;(]*?\b(snsde_[a-z_0-9]+)\s*\(', open(os.path.join(INCLUDE, 'snsde.h')).read(), re.M)))
    vs = os.path.join(objdir, 'exports.map')
    with open(vs, 'w') as f:
        f.write('{\n  global:\n' + ''.join(f'    {n};\n' for n in names) + '  local:\n    *;\n};\n')
    cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,--version-script=' + vs, '-Wl,--no-undefined-version', '-o', out + '.tmp'] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout + r.stderr)
    os.replace(out + '.tmp', out)
    return out


if __name__ == '__main__':
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == 'trace':
        print(build(force=True, defines=('-DSNSDE_TRACE',), out=os.path.join(HERE, 'libsnsde_trace.so')))
    elif len(sys.argv) > 1 and sys.argv[1] == 'leantrace':
        # cycle timeline of the lean kernel (tools/lean_trace.py): headline configurations only; the training-mode variant of the
        # trace build spills two registers and is not used by the tool, hence no resource check
        print(build(force=True, defines=('-DLEAN_TRACE', '-DSNSDE_DEV_SUBSET'), out=os.path.join(HERE, 'libsnsde_leantrace.so'),
                    check_resources=False))
    elif len(sys.argv) > 1 and sys.argv[1] == 'w4trace':
        print(build(force=True, defines=('-DW4_TRACE', '-DSNSDE_DEV_SUBSET'), out=os.path.join(HERE, 'libsnsde_w4trace.so'), check_resources=False))
    elif len(sys.argv) > 2 and sys.argv[1] == 'variant':
        # development experiments: python build.py variant NAME -DFLAG ... -> libsnsde_NAME.so (headline / K4 instantiations only)
        print(build(force=True, defines=tuple(sys.argv[3:]) + ('-DSNSDE_DEV_SUBSET',), out=os.path.join(HERE, f'libsnsde_{sys.argv[2]}.so'),
                    check_resources=False))
    elif len(sys.argv) > 1 and sys.argv[1] == 'inc':
        print(build(force=True, verbose=True, incremental=True))
    elif len(sys.argv) > 1 and sys.argv[1] == 'devtuning':
        # the weight-gradient split heuristic's sweep knobs (tools/sweep_wgrad.sh); the product library reads no environment
        print(build(force=True, defines=('-DSNSDE_DEV_TUNING',), out=os.path.join(HERE, 'libsnsde_devtuning.so')))
    else:
        print(build(force=True, verbose=True))
