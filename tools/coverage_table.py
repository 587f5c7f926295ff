#!/usr/bin/env python3
"""Coverage table of the fused paths (host-side queries only, runs without a GPU): for every (input_option, noise_option,
method) of the reference's Diffusion_model at the BASELINE hidden sizes, which kernel family the forward solve takes
(snsde_forward_path) and which backward a training step takes (snsde_backward_supported: 1 = MFMA adjoint + native
weight-gradient pass, 2 = generic adjoint kernels + batched parameter pass, 0 = autograd through the tensor-op loop).
usage: python tools/coverage_table.py > profiles/rNN_coverage.txt"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_neural_sdes_amd import _lib, engine

L = _lib.lib()
METHODS = (('euler', _lib.EULER), ('milstein', _lib.MILSTEIN), ('srk', _lib.SRK))
FWD = {0: 'LOOP', 1: 'gen', 2: 'M16', 3: 'M4', 4: 'lean', 5: 'leanS', 6: 'genS', 7: 'M4S', 8: 'W4'}
BWD = {0: 'loop', 1: 'mfma', 2: 'gen'}


def query(H, C_, io, no, NL, B, method, N=50, knots=51):
    s = _lib.Solve()
    s.model = engine.model_struct(C_, H, H, NL, io, no)
    s.batch, s.knots, s.n_steps, s.n_out, s.method = B, knots, N, 2, method
    return L.snsde_forward_path(C.byref(s)), L.snsde_backward_supported(C.byref(s))


def main():
    print('Fused-path coverage (host-side query of the C ABI; no GPU involved).  Cell = forward kernel family / backward mode.')
    print('forward: lean = lean 4-row-tile MFMA kernel, leanS = its streamed-weight variant (H = 256), W4 = wave-owns-rows kernels (H = 64, diffusion nets), M4 / M16 = general MFMA')
    print('         kernel with 4- / 16-row tiles, M4S = SRK on MFMA 4-row tiles, gen / genS = generic kernels (any option),')
    print('         LOOP = no kernel: the host layer integrates with its graph-captured tensor-op stepper.')
    print('backward: mfma = MFMA adjoint kernel + native weight-gradient pass, gen = generic adjoint kernels + batched')
    print('          parameter pass, loop = autograd through the tensor-op loop (options={"strict": True} raises instead).')
    for (H, C_, B, NL, what) in ((128, 21, 1024, 2, 'K2 / K3 shape'), (64, 69, 2048, 2, 'K4 shape (sepsis channels)'),
                                 (256, 14, 128, 2, 'K5 per-GPU shard'), (32, 2, 256, 1, 'K1 / tutorial shape'),
                                 (48, 5, 64, 2, 'a hidden size without MFMA instantiation'), (128, 21, 16384, 4, 'large batch, 3 hidden layers')):
        print(f'\nH={H} C={C_} B={B} num_hidden_layers={NL}  ({what})')
        print('  io\\no ' + ' '.join(f'{no:>10d}' for no in range(20)))
        for mname, mval in METHODS:
            for io in range(7):
                cells = []
                for no in range(20):
                    f, b = query(H, C_, io, no, NL, B, mval)
                    cells.append(f'{FWD[f]}/{BWD[b] if f else "loop"}')
                print(f'  {mname[:4]:4s} {io} ' + ' '.join(f'{c:>10s}' for c in cells))
    # summary over the reference's named models (the factory's (input_option, noise_option) pairs)
    print('\nNamed models of the reference factory at H=128, C=21, B=1024 (forward / backward per method):')
    named = {'neuralsde_0_0 (ODE-like)': (0, 0), 'neurallsde (2,16)': (2, 16), 'neurallnsde (4,17)': (4, 17), 'neuralgsde (6,17)': (6, 17),
             'neuralsde_3_18 (K4)': (3, 18), 'neuralsde_1_14': (1, 14), 'neuralsde_5_19': (5, 19), 'neuralsde_4_7 (sqrt y)': (4, 7)}
    for name, (io, no) in named.items():
        row = []
        for mname, mval in METHODS:
            f, b = query(128, 21, io, no, 2, 1024, mval)
            row.append(f'{mname}: {FWD[f]}/{BWD[b] if f else "loop"}')
        print(f'  {name:28s} ' + '   '.join(row))
    fall = sum(1 for H in (32, 64, 128, 256) for io in range(7) for no in range(20) for _, mv in METHODS
               if query(H, 21, io, no, 2, 1024, mv)[0] == 0)
    tot = 4 * 7 * 20 * 3
    print(f'\nforward requests without a kernel at H in (32, 64, 128, 256), C=21: {fall} of {tot} '
          '(Milstein with sqrt(y), noise_option 7: no finite dg/dy at the clipped values)')
    # the diffusion nets (noise_option 14 / 15 / 18 / 19) on the MFMA net kernels (snsde_m4n_kernel.h), by shape
    print('\nDiffusion nets under SRK / Milstein, input_option 1..6 x noise_option 14, 15, 18, 19 (forward/backward), per hidden size and depth:')
    for (H, C_, B) in ((16, 3, 256), (32, 5, 256), (64, 69, 2048), (128, 21, 1024), (128, 69, 1024), (256, 14, 128)):
        for NL in (1, 2, 3, 4):
            for mname, mval in METHODS[1:]:
                cells = {}
                for io in range(1, 7):
                    for no in (14, 15, 18, 19):
                        f, b = query(H, C_, io, no, NL, B, mval)
                        cells.setdefault(f'{FWD[f]}/{BWD[b] if f else "loop"}', []).append((io, no))
                desc = '; '.join(f'{k}: {len(v)} of 24' + ('' if len(v) in (24,) else ' ' + str(sorted(set(v)))[:120]) for k, v in sorted(cells.items()))
                print(f'  H={H:3d} C={C_:2d} NL={NL} {mname:8s} {desc}')


if __name__ == '__main__':
    main()
