#!/usr/bin/env python3
"""Measured gradient margins at the BASELINE sizes (tests/bigcase.py): HIP fp32 and the fp32 tensor loop against fp64 autograd.
Output kept as profiles/rNN_grad_margins.txt; the tolerances of tests/test_gpu_backward_sizes.py are set from it."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import bigcase
dev = torch.device('cuda:0')
for name in (sys.argv[1:] or list(bigcase.CASES)):
    kern = 'auto'
    if ':' in name:
        name, kern = name.split(':')
    rep = bigcase.run_case(name, dev, kernel=kern)
    print(bigcase.format_report(name + ' [' + kern + ']', rep), flush=True)
    del rep
    torch.cuda.empty_cache()
