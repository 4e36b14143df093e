"""ctypes binding of libfibhip.so (include/fibhip.h) — the only door between the Python API and
the HIP kernels.  There is deliberately no fallback: if the shared library is missing or no HIP
device is present, every entry point raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, 'libfibhip.so')
SRC = os.path.join(HERE, 'csrc', 'fibhip.hip')
HDR = os.path.join(ROOT, 'include', 'fibhip.h')

FENTON4V, BR, COURT, COURT_US, CUSTOM = 0, 1, 2, 3, 4
CHEBY, SKIP, CHRONIC, FAST, ALLVARS, ROW_INTERLEAVED, ZEROPAD, HOLD = 1, 2, 4, 8, 16, 32, 64, 128

# -ffp-contract=off: FMAs appear only where the source writes them (policy hook P::mad).
# -fno-slp-vectorize: SLP packs pairs of f32 ops into v_pk_* instructions, which issue at half rate on
#   gfx950 (tools/ubench/valu.hip) and need v_mov shuffles: 11 % slower on the Fenton kernel.
# -fvisibility=hidden: the library exports the C ABI of include/fibhip.h and nothing else (the header pushes default
#   visibility for its declarations).
HIPCC_FLAGS = ['-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize', '-fvisibility=hidden', '-fPIC', '-shared',
               '-std=c++17', '-Wall', '-Wno-unused-value', '-Wno-unused-result']


def _build_tag(so_path):
    """-DFIB_BUILD_TAG for a non-stock build: every kernel symbol of the build carries it (csrc/models.hpp)"""
    import re
    return '-DFIB_BUILD_TAG=b_' + re.sub(r'[^0-9A-Za-z_]', '_', os.path.splitext(os.path.basename(so_path))[0])


class FibhipError(RuntimeError):
    pass


# ---- arrays on page-locked host memory ------------------------------------------------------------------------------
# eval() / image() return ordinary NumPy arrays; behind single-array read-backs sits a small pool of page-locked buffers
# (fibhip_host_alloc) the device writes directly, which saves the staging copy of fibhip_get_state.  A buffer returns to
# the pool when the last view of its array is gone; with more than PINNED_MAX arrays alive at once (a caller keeping
# every frame) further read-backs fall back to pageable arrays.
PINNED_MAX = 8
_pinned_free = {}          # nbytes -> [address, ...]
_pinned_out = [0]


def _pinned_release(nbytes, addr):
    _pinned_out[0] -= 1
    _pinned_free.setdefault(nbytes, []).append(addr)


def _pinned_array(L, shape):
    import weakref
    n = int(np.prod(shape))
    nbytes = 4 * n
    free = _pinned_free.get(nbytes)
    if free:
        addr = free.pop()
    else:
        if _pinned_out[0] + sum(len(v) for v in _pinned_free.values()) >= PINNED_MAX:
            other = next((v for v in _pinned_free.values() if v), None)     # an idle buffer of another size makes room
            if other is None:
                return None
            L.fibhip_host_free(other.pop())
        p = C.c_void_p()
        if L.fibhip_host_alloc(nbytes, C.byref(p)) != 0 or not p.value:
            return None
        addr = p.value
    raw = (C.c_float * n).from_address(addr)
    _pinned_out[0] += 1
    f = weakref.finalize(raw, _pinned_release, nbytes, addr)
    f.atexit = False
    return np.ctypeslib.as_array(raw).reshape(shape)


class Desc(C.Structure):
    _fields_ = [('struct_size', C.c_int), ('model', C.c_int), ('height', C.c_int), ('width', C.c_int),
                ('dt', C.c_double), ('diff', C.c_double), ('flags', C.c_uint), ('device', C.c_int),
                ('steps_per_tick', C.c_int), ('global_height', C.c_int), ('row_offset', C.c_int),
                ('ghost_top', C.c_int), ('ghost_bottom', C.c_int), ('stream', C.c_void_p),
                ('ext_slab', C.c_void_p * 2), ('module', C.c_void_p)]


class ModuleKernel(C.Structure):
    _fields_ = [('symbol', C.c_char_p), ('kind', C.c_int), ('mode', C.c_int), ('fast', C.c_int), ('phase', C.c_int),
                ('K', C.c_int), ('TX', C.c_int), ('TY', C.c_int), ('NT', C.c_int)]


class ModuleDesc(C.Structure):
    _fields_ = [('struct_size', C.c_int), ('nvar', C.c_int), ('steps_per_tick', C.c_int), ('nmodes', C.c_int),
                ('masks', C.c_uint * 8), ('consts_bytes', C.c_int)] + [
                    (k, C.c_int) for k in ('K', 'TX', 'TY', 'R', 'TYB', 'K2', 'TX2', 'TY2', 'R2')] + [
                    ('nkernels', C.c_int), ('kernels', C.POINTER(ModuleKernel))]


class TraceEvent(C.Structure):
    _fields_ = [('name', C.c_char * 96), ('start_us', C.c_double), ('dur_us', C.c_double), ('K', C.c_int), ('tile_w', C.c_int),
                ('tile_h', C.c_int), ('rows_per_wave', C.c_int), ('ticks', C.c_int)]


class HaloMsg(C.Structure):
    _fields_ = [('offset', C.c_longlong), ('count', C.c_longlong), ('peer', C.c_int), ('send', C.c_int)]


_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_h = C.c_void_p

# name -> (argtypes, restype): every symbol include/fibhip.h declares
SYMBOLS = {
    'fibhip_nvar': ([C.c_int], C.c_int),
    'fibhip_default_steps_per_tick': ([C.c_int], C.c_int),
    'fibhip_abi_version': ([], C.c_int),
    'fibhip_device_count': ([], C.c_int),
    'fibhip_create': ([C.POINTER(Desc), C.POINTER(_h)], C.c_int),
    'fibhip_destroy': ([_h], C.c_int),
    'fibhip_set_phase': ([_h, _fp], C.c_int),
    'fibhip_set_state': ([_h, C.c_int, _fp], C.c_int),
    'fibhip_get_state': ([_h, C.c_int, _fp], C.c_int),
    'fibhip_set_consts': ([_h, _fp, C.c_int], C.c_int),
    'fibhip_step': ([_h, C.c_int], C.c_int),
    'fibhip_step_slow': ([_h], C.c_int),
    'fibhip_step_mode': ([_h, C.c_int], C.c_int),
    'fibhip_pace': ([_h, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float], C.c_int),
    'fibhip_probe': ([_h, C.c_int, C.c_int, C.c_int, _fp], C.c_int),
    'fibhip_sync': ([_h], C.c_int),
    'fibhip_time_steps': ([_h, C.c_int, _fp, _ip], C.c_int),
    'fibhip_time_begin': ([_h], C.c_int),
    'fibhip_time_end': ([_h, _fp, _ip], C.c_int),
    'fibhip_step_edges': ([_h], C.c_int),
    'fibhip_step_interior': ([_h], C.c_int),
    'fibhip_step_commit': ([_h], C.c_int),
    'fibhip_state_ptr': ([_h, C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    'fibhip_next_ptr': ([_h, C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    'fibhip_halo_vars': ([_h], C.c_int),
    'fibhip_halo_due': ([_h], C.c_int),
    'fibhip_unit_op': ([C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_double, C.c_int, _fp],
                       C.c_int),
    'fibhip_court_inter': ([C.c_int, C.c_int, _fp, C.c_int, _fp], C.c_int),
    'fibhip_halo_plan': ([_h, C.c_int, C.c_int, C.POINTER(HaloMsg), _ip], C.c_int),
    'fibhip_comm_open': ([C.c_char_p], C.c_int),
    'fibhip_comm_unique_id': ([C.c_char_p], C.c_int),
    'fibhip_comm_check': ([_h, C.c_int, C.c_int], C.c_int),
    'fibhip_comm_init': ([_h, C.c_char_p, C.c_int, C.c_int], C.c_int),
    'fibhip_comm_exchange': ([_h, C.c_int, C.c_int], C.c_int),
    'fibhip_comm_free': ([_h], C.c_int),
    'fibhip_copy_bandwidth': ([C.c_int, C.c_size_t, C.c_int, _fp], C.c_int),
    'fibhip_warm': ([C.c_int], C.c_int),
    'fibhip_module_load': ([C.c_int, C.c_void_p, C.c_size_t, C.POINTER(ModuleDesc), C.POINTER(C.c_void_p)], C.c_int),
    'fibhip_module_unload': ([C.c_void_p], C.c_int),
    'fibhip_launch_plan': ([_h, _ip, _ip], C.c_int),
    'fibhip_get_state_direct': ([_h, C.c_int, _fp], C.c_int),
    'fibhip_host_alloc': ([C.c_size_t, C.POINTER(C.c_void_p)], C.c_int),
    'fibhip_host_free': ([C.c_void_p], C.c_int),
    'fibhip_ticks_per_launch': ([_h], C.c_int),
    'fibhip_launch_stats': ([_h, C.POINTER(C.c_longlong)], C.c_int),
    'fibhip_spec_stats': ([_h, C.POINTER(C.c_longlong)], C.c_int),
    'fibhip_fallbacks': ([_h, C.POINTER(C.c_longlong)], C.c_int),
    'fibhip_set_mt_wait_ms': ([_h, C.c_int], C.c_int),
    'fibhip_expect': ([_h, C.c_int], C.c_int),
    'fibhip_trace_begin': ([_h], C.c_int),
    'fibhip_trace_end': ([_h, C.POINTER(TraceEvent), C.c_int], C.c_int),
    'fibhip_plan_tile': ([_h, _ip, _ip, _ip], C.c_int),
    'fibhip_last_error': ([], C.c_char_p),
}

_lib = None


DEPS = [SRC, HDR] + [os.path.join(HERE, "csrc", f) for f in ("kernels.hpp", "models.hpp", "fenton_step.inc", "br_step.inc",
                                                             "court_step.inc", "court_inter.inc")]


def build(force=False, verbose=False):
    """compile csrc/fibhip.hip for gfx950 in-tree (hipcc cross-compiles without a GPU)"""
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in DEPS):
        return SO
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + HIPCC_FLAGS + [SRC, '-o', SO]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return SO


def build_custom(inc_path, so_path, verbose=False):
    """the same translation unit with ONE traced model compiled in (generated header `inc_path`,
    fib_tf_amd/traced.py) and the stock models' kernels left out: a few seconds instead of a minute"""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    tmp = '%s.%d.tmp' % (so_path, os.getpid())
    cmd = [hipcc] + HIPCC_FLAGS + ['-DFIB_CUSTOM_ONLY', '-DFIB_CUSTOM_MODEL_INC="%s"' % os.path.abspath(inc_path),
                                   _build_tag(so_path), SRC, '-o', tmp]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, so_path)            # atomic: several ranks may build the same model at once
    return so_path


def build_specialised(defines, so_path, verbose=False):
    """the same translation unit with model constants baked in as literals (`defines`: list of -D options);
    used by BeelerReuter for its Chebyshev table: a VALU operand from a literal issues at full rate, one from an
    SGPR (kernel argument) at ~60 % (tools/ubench/valu2.hip), and the table feeds 96 multiply-adds per cell-step"""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    tmp = '%s.%d.tmp' % (so_path, os.getpid())
    cmd = [hipcc] + HIPCC_FLAGS + list(defines) + [_build_tag(so_path), SRC, '-o', tmp]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, so_path)
    return so_path


def load(path):
    """dlopen a build of the library and declare every symbol's signature"""
    L = C.CDLL(path)
    for name, (args, res) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.argtypes, fn.restype = args, res
    if L.fibhip_abi_version() != 1:
        raise FibhipError('%s: ABI version mismatch' % os.path.basename(path))
    return L


def lib():
    global _lib
    if _lib is None:
        alt = os.environ.get('FIBHIP_LIBRARY')          # tuning experiments: another build of the same source
        if alt:
            _lib = load(alt)
            return _lib
        if not os.path.exists(SO):
            raise FibhipError('libfibhip.so is not built (run `python -c "import __graft_entry__ as g; g.build()"`); '
                              'fib_tf_amd has no CPU fallback')
        _lib = load(SO)
    return _lib


_warmed = set()


def warm(device=0):
    """the stock library's code object first: called before any OTHER build of the library launches a kernel in this
    process (include/fibhip.h fibhip_warm: under rocprofv3 the opposite order dies inside the HIP runtime)"""
    if device not in _warmed:
        check(lib().fibhip_warm(device))
        _warmed.add(device)


# ---- in-process builds of traced models: hiprtc -> code object -> fibhip_module_load ------------------------------
HIPRTC = os.environ.get('FIBTF_HIPRTC', '/opt/rocm/lib/libhiprtc.so')
_rtc = None


def hiprtc():
    """libhiprtc bound through ctypes, or None when the runtime does not ship it"""
    global _rtc
    if _rtc is None:
        try:
            R = C.CDLL(HIPRTC)
        except OSError:
            _rtc = False
            return None
        pp = C.POINTER(C.c_char_p)
        R.hiprtcCreateProgram.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, pp, pp]
        R.hiprtcAddNameExpression.argtypes = [C.c_void_p, C.c_char_p]
        R.hiprtcCompileProgram.argtypes = [C.c_void_p, C.c_int, pp]
        R.hiprtcGetLoweredName.argtypes = [C.c_void_p, C.c_char_p, pp]
        R.hiprtcGetProgramLogSize.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        R.hiprtcGetProgramLog.argtypes = [C.c_void_p, C.c_char_p]
        R.hiprtcGetCodeSize.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        R.hiprtcGetCode.argtypes = [C.c_void_p, C.c_char_p]
        R.hiprtcDestroyProgram.argtypes = [C.POINTER(C.c_void_p)]
        _rtc = R
    return _rtc or None


def rtc_compile(source, headers, name_exprs, options):
    """compile device code in this process: (code object bytes, {name expression: lowered symbol}).
    headers: {include name: text} handed over in memory; hiprtc needs no GPU (gfx950 is an option)."""
    R = hiprtc()
    if R is None:
        raise FibhipError('libhiprtc is not available (%s)' % HIPRTC)
    prog = C.c_void_p()
    names = list(headers)
    harr = (C.c_char_p * len(names))(*[headers[n].encode() for n in names])
    narr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    rc = R.hiprtcCreateProgram(C.byref(prog), source.encode(), b'fibhip_traced.hip', len(names), harr, narr)
    if rc:
        raise FibhipError('hiprtcCreateProgram failed (%d)' % rc)
    try:
        for e in name_exprs:
            if R.hiprtcAddNameExpression(prog, e.encode()):
                raise FibhipError('hiprtcAddNameExpression refused %r' % e)
        oarr = (C.c_char_p * len(options))(*[o.encode() for o in options])
        rc = R.hiprtcCompileProgram(prog, len(options), oarr)
        if rc:
            n = C.c_size_t()
            R.hiprtcGetProgramLogSize(prog, C.byref(n))
            log = C.create_string_buffer(max(1, n.value))
            R.hiprtcGetProgramLog(prog, log)
            raise FibhipError('hiprtc could not compile the generated model (%d):\n%s' % (rc, log.value.decode('utf-8', 'replace')[-4000:]))
        lowered = {}
        for e in name_exprs:
            p = C.c_char_p()
            if R.hiprtcGetLoweredName(prog, e.encode(), C.byref(p)) or not p.value:
                raise FibhipError('hiprtcGetLoweredName failed for %r' % e)
            lowered[e] = p.value.decode()
        n = C.c_size_t()
        R.hiprtcGetCodeSize(prog, C.byref(n))
        code = C.create_string_buffer(n.value)
        R.hiprtcGetCode(prog, code)
        return code.raw, lowered
    finally:
        R.hiprtcDestroyProgram(C.byref(prog))


RTC_OPTIONS = ['--offload-arch=gfx950', '-O3', '-ffp-contract=off', '-fno-slp-vectorize', '-std=c++17',
               '-I' + os.path.join(HERE, 'csrc')]


class ModuleLibrary:
    """the stock library + ONE traced model's code object loaded into it (fibhip_module_load): stands where a
    per-model build of libfibhip stood — `Stepper(..., library=this)` creates FIBHIP_CUSTOM handles on the module"""

    def __init__(self, code, meta, device, name):
        L = lib()
        self._L, self._code, self.meta, self.device = L, code, meta, device
        self._name = '%s#module:%s' % (SO, name)
        kern = (ModuleKernel * len(meta['kernels']))()
        self._keep = []
        for i, k in enumerate(meta['kernels']):
            sym = k['symbol'].encode()
            self._keep.append(sym)
            kern[i] = ModuleKernel(sym, k['kind'], k['mode'], k['fast'], k['phase'], k['K'], k['TX'], k['TY'], k['NT'])
        d = ModuleDesc()
        d.struct_size = C.sizeof(ModuleDesc)
        d.nvar, d.steps_per_tick, d.nmodes, d.consts_bytes = meta['nvar'], meta['spt'], meta['nmodes'], meta['consts_bytes']
        for i, m in enumerate(meta['masks']):
            d.masks[i] = m
        for k in ('K', 'TX', 'TY', 'R', 'TYB', 'K2', 'TX2', 'TY2', 'R2'):
            setattr(d, k, meta['plan'][k])
        d.nkernels, d.kernels = len(kern), kern
        self.module = C.c_void_p()
        check(L.fibhip_module_load(device, code, len(code), C.byref(d), C.byref(self.module)), L)

    # model facts the stock library cannot know
    def fibhip_nvar(self, model):
        return self.meta['nvar'] if model == CUSTOM else self._L.fibhip_nvar(model)

    def fibhip_default_steps_per_tick(self, model):
        return self.meta['spt'] if model == CUSTOM else self._L.fibhip_default_steps_per_tick(model)

    def __getattr__(self, name):
        return getattr(self._L, name)


def check(rc, L=None):
    if rc < 0:
        raise FibhipError((L or lib()).fibhip_last_error().decode('utf-8', 'replace'))
    return rc


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_fp)


def unit_op(op, a, b=None, c=None, phi=None, dt=0.0, fast=False, device=0):
    a = _f32(a)
    H, W = a.shape
    b = None if b is None else _f32(b)
    c = None if c is None else _f32(c)
    phi = None if phi is None else _f32(phi)
    out = np.empty_like(a)
    check(lib().fibhip_unit_op(device, op, H, W, _ptr(a), _ptr(b), _ptr(c), _ptr(phi), float(dt), int(fast),
                               _ptr(out)))
    return out


COURT_INTER_KEYS = ('d_infinity', 'tau_d', 'f_infinity', 'tau_f', 'tau_w', 'w_infinity', 'm_inf', 'tau_m', 'h_inf',
                    'tau_h', 'j_inf', 'tau_j', 'tau_oa', 'oa_infinity', 'tau_oi', 'oi_infinity', 'tau_ua',
                    'ua_infinity', 'tau_ui', 'ui_infinity', 'tau_xr', 'xr_infinity', 'tau_xs', 'xs_infinity', 'g_Kur',
                    'f_NaK', 'i_NaCaa', 'i_NaCab', 'i_K1a', 'i_Kra', 'us_infinity', 'tau_us')


def copy_bandwidth(nbytes=1 << 30, reps=5, device=0, library=None):
    """GB/s (read + written) of a plain streaming copy on the device: the achievable-HBM yardstick.
    `library`: the build of libfibhip the caller is already running kernels from (a specialised or traced build)"""
    L = library or lib()
    out = C.c_float()
    check(L.fibhip_copy_bandwidth(device, nbytes, reps, C.byref(out)), L)
    return out.value


def court_inter(V, fast=False, device=0):
    """the voltage-only intermediates of the Courtemanche model for an array (or scalar) of voltages:
    dict key -> float32 array shaped like V (court.py:273-429, court_ultra.py:445-450)"""
    v = _f32(V)
    out = np.empty((len(COURT_INTER_KEYS), v.size), np.float32)
    check(lib().fibhip_court_inter(device, v.size, _ptr(v.reshape(-1)), int(fast), _ptr(out)))
    return {k: out[i].reshape(v.shape) for i, k in enumerate(COURT_INTER_KEYS)}


class Stepper:
    """One fibhip handle: a grid (or a row block of it) resident on one MI355X."""

    def __init__(self, model, height, width, dt, diff, flags=0, device=0, steps_per_tick=0,
                 global_height=0, row_offset=0, ghost_top=0, ghost_bottom=0, stream=None, ext_slabs=None,
                 library=None):
        L = library or lib()
        # ANY other build of the library (a Beeler-Reuter table baked in, a traced model compiled in, an experiment's
        # FIBHIP_BR_LIBRARY): the stock library's code object comes up first on this device — see `warm` — whoever the caller is
        if library is not None and not isinstance(library, ModuleLibrary) and library is not lib():
            warm(device)
        d = Desc()
        d.struct_size = C.sizeof(Desc)
        d.model, d.height, d.width = model, height, width
        d.dt, d.diff, d.flags, d.device = float(dt), float(diff), flags, device
        d.steps_per_tick = steps_per_tick
        d.global_height, d.row_offset = global_height, row_offset
        d.ghost_top, d.ghost_bottom = ghost_top, ghost_bottom
        d.stream = stream
        if ext_slabs is not None:
            d.ext_slab[0], d.ext_slab[1] = ext_slabs
        d.module = getattr(L, 'module', None)          # a traced model's run-time module (ModuleLibrary), or NULL
        self._h = _h()
        self._L = L
        self._warned = False
        self.nvar = self._ck(L.fibhip_nvar(model))
        self.height, self.width = height, width
        self.steps_per_tick = steps_per_tick or self._ck(L.fibhip_default_steps_per_tick(model))
        self._ck(L.fibhip_create(C.byref(d), C.byref(self._h)))

    def _ck(self, rc):
        return check(rc, self._L)

    def close(self):
        if getattr(self, '_h', None) and self._h.value:
            self._L.fibhip_destroy(self._h)
            self._h = _h()

    __del__ = close

    def set_phase(self, phi):
        if phi is None:
            self._ck(self._L.fibhip_set_phase(self._h, None))
        else:
            phi = _f32(phi)
            assert phi.shape == (self.height, self.width)
            self._ck(self._L.fibhip_set_phase(self._h, _ptr(phi)))

    def set_state(self, var, arr):
        arr = _f32(arr)
        want = (self.nvar, self.height, self.width) if var < 0 else (self.height, self.width)
        assert arr.shape == want, (arr.shape, want)
        self._ck(self._L.fibhip_set_state(self._h, var, _ptr(arr)))

    def get_state(self, var=-1):
        shape = (self.nvar, self.height, self.width) if var < 0 else (self.height, self.width)
        if var >= 0:                                    # one array (eval(), image()): straight into page-locked memory
            out = _pinned_array(self._L, shape)
            if out is not None:
                self._ck(self._L.fibhip_get_state_direct(self._h, var, _ptr(out)))
                self._fallback_warning()
                return out
        out = np.empty(shape, np.float32)
        self._ck(self._L.fibhip_get_state(self._h, var, _ptr(out)))
        self._fallback_warning()
        return out

    def set_consts(self, tbl):
        tbl = _f32(tbl).ravel()
        self._ck(self._L.fibhip_set_consts(self._h, _ptr(tbl), tbl.size))

    def step(self, nticks=1):
        self._ck(self._L.fibhip_step(self._h, nticks))

    def step_slow(self):
        self._ck(self._L.fibhip_step_slow(self._h))

    def step_mode(self, mode):
        self._ck(self._L.fibhip_step_mode(self._h, mode))

    def pace(self, r0, r1, c0, c1, v, min_v):
        self._ck(self._L.fibhip_pace(self._h, r0, r1, c0, c1, v, min_v))

    def probe(self, var, row, col):
        out = C.c_float()
        self._ck(self._L.fibhip_probe(self._h, var, row, col, C.byref(out)))
        return np.float32(out.value)

    def sync(self):
        self._ck(self._L.fibhip_sync(self._h))
        self._fallback_warning()

    def expect(self, nticks):
        """declares the caller's next series: `nticks` ticks without an observation in between (include/fibhip.h)"""
        self._ck(self._L.fibhip_expect(self._h, int(nticks)))

    def set_mt_wait_ms(self, ms):
        self._ck(self._L.fibhip_set_mt_wait_ms(self._h, int(ms)))

    def fallbacks(self):
        """(multi-tick launches that gave up and were recovered, ticks recomputed one launch per tick)"""
        out = (C.c_longlong * 2)()
        self._ck(self._L.fibhip_fallbacks(self._h, out))
        return int(out[0]), int(out[1])

    def _fallback_warning(self):
        if not self._warned:
            n, ticks = self.fallbacks()
            if n:
                self._warned = True
                import warnings
                warnings.warn('fib_tf_amd: a launch advancing several ticks gave up waiting for a neighbouring tile (is another '
                              'process holding the GPU, or a CU mask set?); the handle went back to the state that launch started '
                              'from, recomputed %d tick(s) and runs one launch per tick from here on — same results, slower '
                              '(FIBHIP_MT=0 selects that mode from the start)' % ticks, RuntimeWarning, stacklevel=3)

    def time_steps(self, nticks):
        ms, n = C.c_float(), C.c_int()
        self._ck(self._L.fibhip_time_steps(self._h, nticks, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def time_begin(self):
        self._ck(self._L.fibhip_time_begin(self._h))

    def time_end(self):
        """(milliseconds, launches) since time_begin, by HIP events on the handle's stream"""
        ms, n = C.c_float(), C.c_int()
        self._ck(self._L.fibhip_time_end(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def step_edges(self):
        self._ck(self._L.fibhip_step_edges(self._h))

    def step_interior(self):
        self._ck(self._L.fibhip_step_interior(self._h))

    def step_commit(self):
        self._ck(self._L.fibhip_step_commit(self._h))

    def state_buf(self, var):
        p = C.c_void_p()
        return self._ck(self._L.fibhip_state_ptr(self._h, var, C.byref(p))), p.value

    def next_buf(self, var):
        p = C.c_void_p()
        return self._ck(self._L.fibhip_next_ptr(self._h, var, C.byref(p))), p.value

    def halo_vars(self):
        return self._ck(self._L.fibhip_halo_vars(self._h))

    def halo_due(self):
        return bool(self._ck(self._L.fibhip_halo_due(self._h)))

    # ---- direct halo exchange (RCCL issued by the library on this handle's stream) ------------------------
    def comm_open(self, librccl_path=None):
        self._ck(self._L.fibhip_comm_open(librccl_path.encode() if librccl_path else None))

    def comm_unique_id(self, librccl_path=None):
        self.comm_open(librccl_path)
        buf = C.create_string_buffer(128)
        self._ck(self._L.fibhip_comm_unique_id(buf))
        return buf.raw

    def comm_check(self, rank, nranks):
        """the local checks of comm_init (no collective): raises if this rank could not join"""
        self._ck(self._L.fibhip_comm_check(self._h, rank, nranks))

    def comm_init(self, unique_id, rank, nranks, librccl_path=None):
        self._ck(self._L.fibhip_comm_open(librccl_path.encode() if librccl_path else None))
        assert len(unique_id) == 128
        self._ck(self._L.fibhip_comm_init(self._h, unique_id, rank, nranks))

    def halo_plan(self, up, down):
        """[(offset, count, peer, send)], slab index: the messages of this exchange tick (include/fibhip.h)"""
        msg = (HaloMsg * 4)()
        idx = C.c_int()
        n = self._ck(self._L.fibhip_halo_plan(self._h, -1 if up is None else up, -1 if down is None else down, msg,
                                              C.byref(idx)))
        return [(m.offset, m.count, m.peer, bool(m.send)) for m in msg[:n]], idx.value

    def comm_free(self):
        self._L.fibhip_comm_free(self._h)

    def comm_exchange(self, up, down):
        self._ck(self._L.fibhip_comm_exchange(self._h, -1 if up is None else up, -1 if down is None else down))

    def launch_plan(self):
        k, n = C.c_int(), C.c_int()
        self._ck(self._L.fibhip_launch_plan(self._h, C.byref(k), C.byref(n)))
        return k.value, n.value

    def plan_tile(self):
        """(tile width, tile height, rows per wave of a strip kernel | -threads of a flat tile kernel) of the dominant launch"""
        w, t, r = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.fibhip_plan_tile(self._h, C.byref(w), C.byref(t), C.byref(r)))
        return w.value, t.value, r.value

    def trace_begin(self):
        self._ck(self._L.fibhip_trace_begin(self._h))

    def trace_end(self):
        """[{'name', 'ts' (us from the first launch), 'dur' (us), 'K', 'tile', 'ticks'}] of the launches since trace_begin"""
        cap = 1024
        ev = (TraceEvent * cap)()
        n = self._ck(self._L.fibhip_trace_end(self._h, ev, cap))
        self.trace_truncated = n > cap                  # (more launches than the buffer holds: the first `cap` are returned)
        return [{'name': e.name.decode(), 'ts': e.start_us, 'dur': e.dur_us, 'K': e.K, 'tile': (e.tile_w, e.tile_h, e.rows_per_wave),
                 'ticks': e.ticks} for e in ev[:min(n, cap)]]

    def trace_tick(self):
        """one tick with every launch between two HIP events (the timeline of ionic.py:231-241)"""
        self.trace_begin()
        self.step(1)
        return self.trace_end()

    def launch_stats(self):
        """{'launches', 'ticks', 'mt_launches', 'mt_ticks'} since the handle was created"""
        out = (C.c_longlong * 4)()
        self._ck(self._L.fibhip_launch_stats(self._h, out))
        d = dict(zip(('launches', 'ticks', 'mt_launches', 'mt_ticks'), [int(x) for x in out]))
        sp = (C.c_longlong * 2)()
        self._ck(self._L.fibhip_spec_stats(self._h, sp))
        d.update(ahead_stopped_in_time=int(sp[0]), ahead_recomputed=int(sp[1]))
        fb = self.fallbacks()
        d.update(gave_up_recovered=fb[0], ticks_recomputed_after_give_up=fb[1])
        return d

    def ticks_per_launch(self):
        """consecutive ticks one launch can cover (Courtemanche, fast policy, one device: 3; Fenton / Beeler-Reuter on a
        grid whose tiles are all resident at once: FIBHIP_MT_MAX, default 32; otherwise 1)"""
        return self._ck(self._L.fibhip_ticks_per_launch(self._h))
