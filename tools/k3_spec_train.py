"""K3 (BASELINE configs[2]: GSDE (6,17), H = 128, 200 Euler steps, Hermite coefficients) TRAINING at the SPECIFIED weight scale (unit
nn.Linear init): per row, is dL/dy0 finite in the fused fp32 adjoint, in fp32 autograd through the tensor loop, in the fp64 loop -
and how close are the finite ones?  Driver: tests/bigcase.py (k3_spec_run).  usage: python tools/k3_spec_train.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.bigcase import k3_spec_run, k3_spec_summary      # noqa: E402

if __name__ == '__main__':
    rep = k3_spec_summary(k3_spec_run(int(sys.argv[1]) if len(sys.argv) > 1 else 256))
    for k, v in rep.items():
        print(f'{k:36s} {v}')
