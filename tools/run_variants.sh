for v in "$@"; do SNSDE_LIB=$PWD/stable-neural-sdes_amd/libsnsde_v_$v.so python tools/lean_check.py time 2>&1 | grep -v amdgpu.ids; done
