SNSDE_LIB=$PWD/stable-neural-sdes_amd/libsnsde_leantrace.so python tools/lean_trace.py train 2>&1 | grep -v amdgpu.ids
for i in 1 2; do python tools/time_train.py 2>&1 | grep -E "sdeint|training-mode|adjoint|native"; done
python -m pytest tests/test_gpu_parity.py -x -q -k "backward or train or recompute or graph" 2>&1 | tail -3
