"""ctypes binding of libsnsde.so (C ABI: include/snsde.h).

The shared library is built in-tree by ``build.py`` (``__graft_entry__.build()``).  There is no
Python/CPU substitute for it: ``lib()`` raises if it is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SNSDE_LIB', os.path.join(_HERE, 'libsnsde.so'))   # SNSDE_LIB: debug/trace builds

SNSDE_STEP_STRIDE = 12
EULER, MILSTEIN, SRK = 0, 1, 2
SNSDE_SRK_STRIDE = 8
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_MFMA = 0, 1, 2
FLAG_REUSE_PREPARED = 1
FLAG_EXACT_ORDER = 2
FLAG_STREAM_ALL = 4
FLAG_TWO_TILE = 8
BWD_ADJ0_ONLY = 1
PATHS = ('none', 'generic', 'mfma16', 'mfma4', 'lean', 'lean-streamed', 'generic-srk', 'mfma-srk', 'w4')
KERNELS = {'auto': KERNEL_AUTO, 'generic': KERNEL_GENERIC, 'mfma': KERNEL_MFMA, 'mfma16': 3, 'mfma4': 4, 'w4': 5}


class Model(C.Structure):
    _fields_ = [('input_channels', C.c_int32), ('hidden_channels', C.c_int32),
                ('hidden_hidden_channels', C.c_int32), ('num_hidden_layers', C.c_int32),
                ('input_option', C.c_int32), ('noise_option', C.c_int32), ('activation', C.c_int32),
                ('drift_output', C.c_int32), ('diffusion_output', C.c_int32), ('time_feature', C.c_int32)]


class _Sized(C.Structure):
    """Descriptor structs start with struct_size = sizeof(the struct as THIS binding declares it): the library compares it with
    its own sizeof and returns SNSDE_ERR_ABI (-10) for a stale binding instead of reading past the struct."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(type(self))


class Solve(_Sized):
    _fields_ = [('struct_size', C.c_uint32), ('model', Model), ('batch', C.c_int32), ('knots', C.c_int32), ('n_steps', C.c_int32),
                ('n_out', C.c_int32), ('method', C.c_int32), ('kernel', C.c_int32), ('flags', C.c_int32),
                ('reserved', C.c_int32),
                ('row_offset', C.c_int64), ('seed', C.c_uint64),
                ('params', C.c_void_p), ('coeffs', C.c_void_p), ('step_tab', C.c_void_p),
                ('out_step', C.c_void_p), ('out_w', C.c_void_p), ('y0', C.c_void_p), ('dW', C.c_void_p),
                ('ys', C.c_void_p), ('traj', C.c_void_p), ('dW_out', C.c_void_p), ('srk_tab', C.c_void_p), ('dU', C.c_void_p),
                ('dU_out', C.c_void_p), ('act_save', C.c_void_p), ('stage_save', C.c_void_p), ('seed_dev', C.c_void_p), ('noise_table', C.c_void_p), ('row_out', C.c_void_p),
                ('z0_weight', C.c_void_p), ('z0_bias', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t),
                ('kl_column1', C.c_int32), ('kl_prior_a', C.c_float), ('kl_prior_b', C.c_float), ('reserved2', C.c_int32)]


class Backward(_Sized):
    _fields_ = [('struct_size', C.c_uint32), ('fwd', Solve), ('grad_ys', C.c_void_p), ('adj', C.c_void_p), ('delta_save', C.c_void_p),
                ('workspace', C.c_void_p),
                ('workspace_bytes', C.c_size_t), ('grad_noise_table', C.c_void_p), ('flags', C.c_int32), ('reserved', C.c_int32)]


class Head(_Sized):
    _fields_ = [('struct_size', C.c_uint32), ('rows', C.c_int32), ('in_features', C.c_int32), ('hidden', C.c_int32), ('out_features', C.c_int32),
                ('input_tanh', C.c_int32), ('bn_eps', C.c_float),
                ('x', C.c_void_p), ('w1', C.c_void_p), ('b1', C.c_void_p), ('bn_mean', C.c_void_p), ('bn_var', C.c_void_p),
                ('bn_weight', C.c_void_p), ('bn_bias', C.c_void_p), ('w2', C.c_void_p), ('b2', C.c_void_p), ('out', C.c_void_p)]


class SnsdeError(RuntimeError):
    def __init__(self, code, what=''):
        self.code = code
        msg = lib().snsde_strerror(code).decode()
        super().__init__(f'libsnsde: {msg} (code {code}){": " + what if what else ""}')


_lib = None

ABI_VERSION = 2
EXPORTS = ('snsde_version', 'snsde_abi_check', 'snsde_strerror', 'snsde_param_count', 'snsde_param_numel', 'snsde_param_info',
           'snsde_grid_count', 'snsde_grid_build', 'snsde_grid_srk_build', 'snsde_workspace_bytes', 'snsde_solve_forward',
           'snsde_spline_evaluate', 'snsde_eval_fg', 'snsde_act_slots', 'snsde_backward_supported',
           'snsde_backward_workspace_bytes', 'snsde_solve_backward', 'snsde_spline_workspace_bytes',
           'snsde_natural_cubic_coeffs', 'snsde_hermite_coeffs', 'snsde_param_gradients_workspace_bytes',
           'snsde_param_gradients', 'snsde_backward_with_gradients', 'snsde_forward_path', 'snsde_readout_head', 'snsde_save_layout',
           'snsde_affine_compose', 'snsde_affine_compose_backward')


MAX_AFFINE_JOBS = 12


class AffineJob(C.Structure):
    """snsde_affine_job (include/snsde.h): one (weight, bias) pair of a composed parameter block."""
    _fields_ = [('w_outer', C.c_void_p), ('b_outer', C.c_void_p), ('w_inner', C.c_void_p), ('b_inner', C.c_void_p),
                ('R', C.c_int32), ('K', C.c_int32), ('Cin', C.c_int32), ('zero_col', C.c_int32),
                ('dst_w', C.c_int64), ('dst_b', C.c_int64),
                ('g_w_outer', C.c_int64), ('g_b_outer', C.c_int64), ('g_w_inner', C.c_int64), ('g_b_inner', C.c_int64)]


def lib():
    """Load libsnsde.so once.  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the HIP engine is not built. Run `python -c "import __graft_entry__ as g; '
            f'g.build()"` (hipcc --offload-arch=gfx950) from the repository root. There is no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    L.snsde_version.restype = C.c_int
    L.snsde_strerror.restype = C.c_char_p
    L.snsde_strerror.argtypes = [C.c_int]
    L.snsde_param_count.argtypes = [C.POINTER(Model)]
    L.snsde_param_numel.argtypes = [C.POINTER(Model)]
    L.snsde_param_numel.restype = C.c_int64
    L.snsde_param_info.argtypes = [C.POINTER(Model), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.snsde_grid_count.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.POINTER(C.c_int32)]
    L.snsde_grid_build.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    L.snsde_grid_srk_build.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.snsde_workspace_bytes.argtypes = [C.POINTER(Solve)]
    L.snsde_workspace_bytes.restype = C.c_size_t
    L.snsde_solve_forward.argtypes = [C.POINTER(Solve), C.c_void_p]
    L.snsde_spline_evaluate.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                        C.c_int32, C.c_void_p, C.c_void_p]
    L.snsde_eval_fg.argtypes = [C.POINTER(Solve), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.snsde_spline_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.snsde_spline_workspace_bytes.restype = C.c_size_t
    L.snsde_natural_cubic_coeffs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_void_p]
    L.snsde_hermite_coeffs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.snsde_act_slots.argtypes = [C.POINTER(Model)]
    L.snsde_save_layout.argtypes = [C.POINTER(Solve), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.snsde_param_gradients_workspace_bytes.argtypes = [C.POINTER(Backward)]
    L.snsde_param_gradients_workspace_bytes.restype = C.c_size_t
    L.snsde_param_gradients.argtypes = [C.POINTER(Backward), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.snsde_backward_with_gradients.argtypes = [C.POINTER(Backward), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.snsde_backward_supported.argtypes = [C.POINTER(Solve)]
    L.snsde_forward_path.argtypes = [C.POINTER(Solve)]
    L.snsde_readout_head.argtypes = [C.POINTER(Head), C.c_void_p]
    L.snsde_affine_compose.argtypes = [C.POINTER(AffineJob), C.c_int32, C.c_void_p, C.c_void_p]
    L.snsde_affine_compose_backward.argtypes = [C.POINTER(AffineJob), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.snsde_backward_workspace_bytes.argtypes = [C.POINTER(Backward)]
    L.snsde_backward_workspace_bytes.restype = C.c_size_t
    L.snsde_solve_backward.argtypes = [C.POINTER(Backward), C.c_void_p]
    L.snsde_abi_check.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    rc = L.snsde_abi_check(ABI_VERSION, C.sizeof(Model), C.sizeof(Solve), C.sizeof(Backward), C.sizeof(Head))
    if rc != 0:
        raise RuntimeError(f'{LIB_PATH}: ABI mismatch (library version {L.snsde_version()}, binding version {ABI_VERSION}, code {rc}): '
                           f'rebuild the library (__graft_entry__.build()) or update the binding to include/snsde.h')
    _lib = L
    return L


def check(code, what=''):
    if code != 0:
        raise SnsdeError(code, what)


_LAYOUT_CACHE = {}


def param_layout(model):
    """[(name, offset, shape)] of the flat parameter block, in state_dict order (memoised per model shape)."""
    key = (model.input_channels, model.hidden_channels, model.hidden_hidden_channels, model.num_hidden_layers,
           model.input_option, model.noise_option)
    if key in _LAYOUT_CACHE:
        return _LAYOUT_CACHE[key]
    L = lib()
    n = L.snsde_param_count(C.byref(model))
    check(min(n, 0), 'snsde_param_count')
    out = []
    buf = C.create_string_buffer(64)
    off, rows, cols = C.c_int64(), C.c_int32(), C.c_int32()
    for i in range(n):
        check(L.snsde_param_info(C.byref(model), i, buf, 64, C.byref(off), C.byref(rows), C.byref(cols)))
        name = buf.value.decode()
        shape = (rows.value, cols.value) if cols.value > 0 else (rows.value,)
        out.append((name, off.value, shape))
    _LAYOUT_CACHE[key] = (out, int(L.snsde_param_numel(C.byref(model))))
    return _LAYOUT_CACHE[key]
