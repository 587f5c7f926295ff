"""The torch-CPU restatement used as bench.py's cpu_baseline must agree with the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import sde_oracle as O
from oracle import torch_loop as T
from tests.helpers import draw_dW, make_problem


@pytest.mark.parametrize('io,no', [(4, 17), (6, 17), (2, 16), (1, 18), (3, 18), (0, 5), (5, 15), (1, 7)])
def test_torch_loop_matches_numpy_oracle(io, no):
    pr = make_problem(3, io, no, 2, 6, 16, 4, 9)
    dW = draw_dW(3, [0, 8], 1.0, 6, 16)
    p = {k: torch.from_numpy(v) for k, v in pr['params'].items()}
    y = T.euler_solve(p, io, no, torch.from_numpy(pr['coeffs']), torch.from_numpy(pr['times']),
                      torch.from_numpy(pr['y0']), 0.0, 8, 1.0, dW=torch.from_numpy(dW))
    ref, _ = O.solve_diffusion_model(pr['params'], io, no, pr['coeffs'], pr['times'], pr['y0'],
                                     np.array([0, 8], np.float32), 1.0, dW, dtype=np.float64)
    np.testing.assert_allclose(y.numpy(), ref[-1], rtol=2e-5, atol=2e-5)
