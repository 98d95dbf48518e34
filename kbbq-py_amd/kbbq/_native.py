"""
ctypes binding of libkbbq_hip.so (C ABI: include/kbbq_hip.h).

There is NO CPU fallback: if the shared library is missing or cannot be loaded,
or no gfx950 device is present when a context is requested, the error is raised
to the caller.  Build the library with ``make -C kbbq-py_amd/csrc`` (or
``python -c "import __graft_entry__ as g; g.build()"`` from the repo root).
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# KBBQ_HIP_LIB selects another build of the same ABI (timing-only ablation builds, scripts/)
LIB_PATH = os.environ.get('KBBQ_HIP_LIB') or os.path.join(HERE, 'libkbbq_hip.so')

KBBQ_OK = 0
KBBQ_E_HIP = -1
KBBQ_E_INDEX = -2
KBBQ_E_TYPE = -3
KBBQ_E_ARG = -4
KBBQ_E_RANGE = -5
KBBQ_E_NAME = -6
KBBQ_E_LUT = -7
KBBQ_E_MEANQ = -8
APPLY_CHECKED, APPLY_FAST = 0, 1
ROWS_PAIRS, ROWS_NIBBLES, ROWS_TWINS = 1, 2, 4

NQ = 43
NDINUC = 16

_c = ctypes
_vp, _i, _i64, _u64, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_uint64, _c.c_size_t

# name -> (restype, argtypes); every symbol include/kbbq_hip.h declares
PROTOTYPES = {
    'kbbq_abi_version': (_i, []),
    'kbbq_last_error': (_c.c_char_p, []),
    'kbbq_device_count': (_i, [_c.POINTER(_i)]),
    'kbbq_host_threads': (_i, [_sz]),
    'kbbq_host_advise_huge': (_i, [_vp, _sz]),
    'kbbq_bind_host_to_pci': (_i, [_c.c_char_p, _c.POINTER(_i), _c.POINTER(_i)]),
    'kbbq_bind_host_to_device': (_i, [_i, _c.POINTER(_i), _c.POINTER(_i)]),
    'kbbq_ctx_create': (_i, [_i, _c.POINTER(_vp)]),
    'kbbq_ctx_destroy': (_i, [_vp]),
    'kbbq_ctx_set_stream': (_i, [_vp, _vp]),
    'kbbq_ctx_sync': (_i, [_vp]),
    'kbbq_ctx_status': (_i, [_vp, _c.POINTER(_i64)]),
    'kbbq_ctx_info': (_i, [_vp, _c.POINTER(_i), _c.POINTER(_i), _c.c_char_p, _i]),
    'kbbq_dev_alloc': (_i, [_vp, _sz, _c.POINTER(_vp)]),
    'kbbq_dev_free': (_i, [_vp, _vp]),
    'kbbq_dev_zero': (_i, [_vp, _vp, _sz]),
    'kbbq_dev_upload': (_i, [_vp, _vp, _vp, _sz]),
    'kbbq_dev_download': (_i, [_vp, _vp, _vp, _sz]),
    'kbbq_dev_mem_info': (_i, [_vp, _c.POINTER(_sz), _c.POINTER(_sz)]),
    'kbbq_host_alloc': (_i, [_sz, _c.POINTER(_vp)]),
    'kbbq_host_free': (_i, [_vp]),
    'kbbq_dev_copy_async': (_i, [_vp, _vp, _vp, _sz, _i]),
    'kbbq_event_create': (_i, [_vp, _c.POINTER(_vp)]),
    'kbbq_event_record': (_i, [_vp, _vp]),
    'kbbq_event_sync': (_i, [_vp]),
    'kbbq_event_destroy': (_i, [_vp]),
    'kbbq_tables_count': (_sz, [_i, _i]),
    'kbbq_lut_count': (_sz, [_i, _i, _i]),
    'kbbq_accumulate_dev': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    'kbbq_accumulate_ex_dev': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    'kbbq_accumulate_band_dev': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'kbbq_accumulate': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'kbbq_lut_row_stride': (_i, [_i]),
    'kbbq_full_lut_bytes': (_sz, [_i, _i, _i]),
    'kbbq_lut_bytes': (_sz, [_i, _i, _i]),
    'kbbq_build_lut': (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _c.POINTER(_i)]),
    'kbbq_apply_dev': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    'kbbq_apply': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i,
                        _vp, _vp, _vp, _vp, _vp, _vp]),
    'kbbq_delta_q_dev': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'kbbq_combiln_host': (_i, [_vp, _vp, _i64, _vp, _i]),
    'kbbq_solve_prep_host': (_i, [_vp, _i, _i, _vp, _vp, _i]),
    'kbbq_gammaln_host': (_i, [_vp, _i64, _vp]),
    'kbbq_xlogy_tables_host': (_i, [_vp, _i, _vp, _vp]),
    'kbbq_libm_log_data': (_i, [_vp, _i]),
    'kbbq_gammaln_restated_host': (_i, [_vp, _i64, _vp, _vp]),
    'kbbq_gammaln_dev': (_i, [_vp, _vp, _i64, _vp, _vp]),
    'kbbq_solve_device_dev': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'kbbq_posterior_q_dev': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'kbbq_solve_aux_count': (_sz, [_i, _i]),
    'kbbq_solve_dq_count': (_sz, [_i, _i]),
    'kbbq_solve_dev': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'kbbq_find_errors_dev': (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    'kbbq_accumulate_aligned_dev': (_i, [_vp] * 7 + [_i64, _i, _i, _i, _i, _i, _vp]),
    'kbbq_tally_aligned_dev': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i] + [_vp] * 6 + [_i64] + [_vp] * 4 + [_i, _i, _i, _vp]),
    'kbbq_count_q_dev': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    'kbbq_pair_pitch': (_i, [_i]),
    'kbbq_pair_lut_bytes': (_sz, [_i, _i, _i]),
    'kbbq_pack_pairs_dev': (_i, [_vp] * 5 + [_i64, _i, _i] + [_vp] * 4),
    'kbbq_unpack_pairs_dev': (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    'kbbq_accumulate_pairs_dev': (_i, [_vp] * 5 + [_i64, _i, _i, _i, _i, _vp]),
    'kbbq_pair_lut_dev': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'kbbq_pair_lut_rows_dev': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'kbbq_apply_pairs_dev': (_i, [_vp] * 4 + [_i64, _i, _i, _i] + [_vp] * 3),
    'kbbq_accumulate_grouped_dev': (_i, [_vp] * 5 + [_i64, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'kbbq_apply_grouped_dev': (_i, [_vp] * 4 + [_i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'kbbq_accumulate_rows_dev': (_i, [_vp] * 5 + [_i64, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'kbbq_apply_rows_dev': (_i, [_vp] * 4 + [_i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'kbbq_meta_stats_dev': (_i, [_vp, _vp, _i64, _vp]),
    'kbbq_group_rows_work_bytes': (_sz, [_i64, _i]),
    'kbbq_group_rows_dev': (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    'kbbq_lay_out_dev': (_i, [_vp] * 5 + [_i64, _i, _i, _i] + [_vp] * 5),
    'kbbq_unpack_nibbles_dev': (_i, [_vp, _vp, _i64, _vp]),
    'kbbq_tables_add_dev': (_i, [_vp, _vp, _vp, _sz]),
    'kbbq_comm_library': (_i, [_c.c_char_p]),
    'kbbq_comm_unique_id': (_i, [_vp]),
    'kbbq_comm_create': (_i, [_vp, _vp, _i, _i, _c.POINTER(_vp)]),
    'kbbq_allreduce_tables': (_i, [_vp, _vp, _sz]),
    'kbbq_comm_destroy': (_i, [_vp]),
    'kbbq_sam_open': (_i, [_c.c_char_p, _vp]),
    'kbbq_sam_close': (_i, [_vp]),
    'kbbq_sam_info': (_i, [_vp, _vp]),
    'kbbq_sam_fields': (_i, [_vp] * 13),
    'kbbq_sam_cigar': (_i, [_vp, _vp]),
    'kbbq_sam_adaptor_trim': (_i, [_vp, _vp]),
    'kbbq_text_open': (_i, [_c.c_char_p, _c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_sz)]),
    'kbbq_text_close': (_i, [_vp]),
    'kbbq_sam_fill': (_i, [_vp, _i64, _i64, _i, _i, _vp]),
    'kbbq_sam_text': (_i, [_vp, _i, _i64, _vp, _vp]),
    'kbbq_sam_match_fastq': (_i, [_vp, _vp, _vp]),
    'kbbq_canonical_reads_dev': (_i, [_vp] * 9 + [_i64, _i, _i, _i, _i] + [_vp] * 4),
    'kbbq_canonical_reads_rows_dev': (_i, [_vp] * 9 + [_i64, _i, _i, _i, _i, _i] + [_vp] * 4),
    'kbbq_fastq_open': (_i, [_c.c_char_p, _c.POINTER(_vp)]),
    'kbbq_fastq_close': (_i, [_vp]),
    'kbbq_fastq_open_range': (_i, [_c.c_char_p, _i64, _i64, _c.POINTER(_vp)]),
    'kbbq_fastq_record_offset': (_i64, [_vp, _i64]),
    'kbbq_fastq_sync_offset': (_i64, [_c.c_char_p, _i64]),
    'kbbq_fastq_sync_offset_ex': (_i64, [_c.c_char_p, _i64, _i]),
    'kbbq_fastq_is_plain': (_i, [_vp]),
    'kbbq_fastq_set_rg_names': (_i, [_vp, _c.c_char_p, _i]),
    'kbbq_fastq_count': (_i64, [_vp]),
    'kbbq_fastq_name': (_i, [_vp, _i64, _c.POINTER(_vp), _c.POINTER(_i)]),
    'kbbq_fastq_rg_count': (_i, [_vp]),
    'kbbq_fastq_rg_name': (_c.c_char_p, [_vp, _i]),
    'kbbq_fastq_scan': (_i, [_vp, _vp, _i, _vp]),
    'kbbq_fastq_scan_next': (_i, [_vp, _vp, _i, _i64, _vp]),
    'kbbq_fastq_stream_open': (_i, [_c.c_char_p, _c.POINTER(_vp)]),
    'kbbq_fastq_stream_is_regular': (_i, [_vp]),
    'kbbq_fastq_stream_tee': (_i, [_vp, _i]),
    'kbbq_fastq_stream_next': (_i, [_vp, _sz, _i64, _c.POINTER(_vp), _c.POINTER(_i)]),
    'kbbq_fastq_stream_prefetch': (_i, [_vp, _sz]),
    'kbbq_fastq_stream_close': (_i, [_vp]),
    'kbbq_fastq_pair_begin': (_i, [_c.c_char_p, _c.c_char_p, _i, _c.POINTER(_vp)]),
    'kbbq_fastq_pair_wait': (_i, [_vp, _c.POINTER(_vp), _c.POINTER(_vp), _vp]),
    'kbbq_fastq_lengths': (_i, [_vp, _i64, _i64, _vp]),
    'kbbq_fastq_length_runs': (_i, [_vp, _i64, _i64, _vp, _i, _i, _vp]),
    'kbbq_fastq_fill': (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    'kbbq_fastq_fill_range': (_i, [_vp, _vp, _i, _i64, _i64, _i, _vp, _vp, _vp, _vp]),
    'kbbq_fastq_format': (_i64, [_vp, _i64, _i64, _i, _vp, _vp, _i64]),
    'kbbq_fastq_meta': (_i, [_vp, _i, _i64, _i64, _vp, _vp]),
    'kbbq_group_rows_host': (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    'kbbq_fastq_fill_rows': (_i, [_vp, _vp, _i64, _i64, _vp, _i, _i, _i, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _c.POINTER(_i)]),
    'kbbq_fastq_format_rows': (_i64, [_vp, _i64, _i64, _i, _i, _i, _vp, _vp, _i64]),
    'kbbq_synth_dev': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i, _u64,
                            _i, _i, _i, _i, _i, _vp]),
    'kbbq_ctx_timing': (_i, [_vp, _i]),
    'kbbq_ctx_kernel_ms': (_i, [_vp, _i, _c.POINTER(_c.c_double), _c.POINTER(_i64), _i]),
}



class Band(ctypes.Structure):
    """kbbq_band (include/kbbq_hip.h): one length band of a mixed-length input for kbbq_accumulate_bands_dev / kbbq_apply_bands_dev."""
    _fields_ = [('d_seq', _vp), ('d_cseq', _vp), ('d_qual', _vp), ('d_meta', _vp), ('nrows', _i64), ('pitch', _c.c_int32),
                ('flags', _c.c_int32), ('S_band', _c.c_int32), ('S_min', _c.c_int32), ('d_seg', _vp), ('d_perm', _vp),
                ('d_out', _vp), ('d_pair_lut', _vp)]


PROTOTYPES['kbbq_accumulate_bands_dev'] = (_i, [_vp, _c.POINTER(Band), _i, _i, _i, _i, _i, _vp])
PROTOTYPES['kbbq_apply_bands_dev'] = (_i, [_vp, _c.POINTER(Band), _i, _i, _i, _i, _vp])

_lib = None


class KbbqHipError(RuntimeError):
    """A HIP runtime failure or a bad argument reported by libkbbq_hip."""


class LutNeedsCheckedApply(KbbqHipError):
    """kbbq_ctx_status: the device-built LUT is not usable by the table-driven apply kernel."""


class MeanqNeedsHost(KbbqHipError):
    """kbbq_ctx_status: the all-device solve met a meanq on a truncation boundary; solve with the host's longdouble meanq."""


def _preload_torch_hip_runtime():
    import sys
    if 'torch' in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')          # locates the package without importing it
        path = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        import torch  # noqa: F401 -- an unexpected layout: the import maps the runtime itself


def load():
    """dlopen libkbbq_hip.so and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libkbbq_hip.so is not built (%s).  Run `make -C kbbq-py_amd/csrc`; '
            'this package has no CPU fallback.' % LIB_PATH)
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP runtimes in one
    # process cannot both open the GPU, so torch's must be the one already mapped when
    # libkbbq_hip.so resolves its DT_NEEDED libamdhip64.so.7.  Mapping that file is enough (torch finds it mapped
    # when it is imported later): the host-side readers can start before the second-long `import torch`.
    _preload_torch_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.kbbq_abi_version() != 1:
        raise ImportError('libkbbq_hip.so ABI version mismatch')
    _lib = lib
    return lib


def last_error():
    msg = load().kbbq_last_error()
    return msg.decode('utf-8', 'replace') if msg else ''


def check(rc):
    """Map a C return code to the exception the reference raises in that situation."""
    if rc == KBBQ_OK:
        return
    msg = last_error()
    if rc == KBBQ_E_INDEX:
        raise IndexError(msg)
    if rc == KBBQ_E_TYPE:
        raise TypeError(msg)
    if rc == KBBQ_E_RANGE:
        raise ValueError(msg)
    if rc == KBBQ_E_NAME:
        raise AssertionError(msg)
    if rc == KBBQ_E_ARG:
        raise ValueError(msg)
    if rc == KBBQ_E_LUT:
        raise LutNeedsCheckedApply(msg)
    if rc == KBBQ_E_MEANQ:
        raise MeanqNeedsHost(msg)
    raise KbbqHipError(msg or ('libkbbq_hip error %d' % rc))


def ptr(x):
    """Device or host address of a torch tensor / numpy array / int / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if hasattr(x, 'data_ptr'):
        return ctypes.c_void_p(x.data_ptr())
    return ctypes.c_void_p(x.ctypes.data)


class Context:
    """One per device (kbbq_ctx).  Not thread-safe, except that device allocations, asynchronous copies and events (what the
    output pipeline's copy stage uses) may run on a second thread beside the one that launches kernels: include/kbbq_hip.h."""

    def __init__(self, device=0):
        lib = load()
        n = ctypes.c_int(0)
        rc = lib.kbbq_device_count(ctypes.byref(n))
        if rc != KBBQ_OK or n.value == 0:
            raise KbbqHipError('no HIP device available (%s); kbbq needs an MI355X (gfx950) -- '
                               'there is no CPU fallback' % last_error())
        h = ctypes.c_void_p()
        check(lib.kbbq_ctx_create(device, ctypes.byref(h)))
        self._h = h
        self.device = device
        cus, lds = ctypes.c_int(0), ctypes.c_int(0)
        name = ctypes.create_string_buffer(128)
        check(lib.kbbq_ctx_info(h, ctypes.byref(cus), ctypes.byref(lds), name, 128))
        self.compute_units, self.lds_bytes = cus.value, lds.value
        self.name = name.value.decode()

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, '_h', None):
            load().kbbq_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        check(load().kbbq_ctx_set_stream(self._h, ctypes.c_void_p(stream_ptr) if stream_ptr else None))   # 0 = null stream

    def sync(self):
        check(load().kbbq_ctx_sync(self._h))

    def status(self):
        """Raise the error the reference would have raised, if a kernel flagged one."""
        idx = ctypes.c_int64(-1)
        try:
            check(load().kbbq_ctx_status(self._h, ctypes.byref(idx)))
        except Exception as exc:
            exc.read_index = int(idx.value)          # row of the batch the kernel flagged (multi-rank error agreement)
            raise

    def timing(self, enable):
        check(load().kbbq_ctx_timing(self._h, 1 if enable else 0))

    def kernel_ms(self, which, reset=False):
        ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        check(load().kbbq_ctx_kernel_ms(self._h, which, ctypes.byref(ms), ctypes.byref(n), 1 if reset else 0))
        return ms.value, n.value
