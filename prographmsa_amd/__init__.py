"""prographmsa_amd — MI355X (gfx950) accelerator for ProGraphMSA's graph-vs-graph DP hot path.

The product is the C-ABI shared library ``lib/libpgm_hip.so`` (HIP kernels, see ``include/pgm_hip.h``)
plus the C++ host mirror of the reference call surface (``host/``, driver ``bin/pgmsa``).  This Python
package is plumbing only: a ctypes binding of the C ABI for the tests and ``bench.py``.

There is no CPU fallback: importing works without a GPU (symbols can be inspected), but every compute
entry point needs a gfx950 device, and a missing library raises ``ImportError`` at import time.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PGM_TOOLS_LIB=1: the tools build of the same library, `make -C prographmsa_amd/csrc tools` — experiment switches compiled in)
LIB_PATH = os.path.join(_HERE, "lib", "libpgm_hip_tools.so" if os.environ.get("PGM_TOOLS_LIB") else "libpgm_hip.so")
PGMSA_PATH = os.path.join(_HERE, "bin", "pgmsa")

PGM_OK, PGM_ERR_INVALID, PGM_ERR_DEVICE, PGM_ERR_BACKTRACK, PGM_ERR_NOMEM = 0, 1, 2, 3, 4
PGM_GAP = 0xFFFFFFFF
PGM_BATCH_KEEP_MATRICES = 1
PGM_NW_REDUCED = 1
PGM_MERGE_RESIDENT = 1


class pgm_graph(C.Structure):
    _fields_ = [("n", C.c_uint32), ("dim", C.c_uint32), ("sites", C.POINTER(C.c_double)),
                ("e_rowptr", C.POINTER(C.c_int32)), ("e_col", C.POINTER(C.c_uint32)), ("e_val", C.POINTER(C.c_float)),
                ("r_rowptr", C.POINTER(C.c_int32)), ("r_col", C.POINTER(C.c_uint32)), ("r_units", C.POINTER(C.c_uint32))]


class pgm_model(C.Structure):
    _fields_ = [("M", C.POINTER(C.c_double)), ("pi", C.POINTER(C.c_double))]


class pgm_scores(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("gap_init", "gap_extend", "match_init", "end_match", "end_gap", "end_skip",
                                         "start_gap", "start_init", "repeat_init", "repeat_ext")]


class pgm_mldist_model(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("Q", C.POINTER(C.c_double)), ("V", C.POINTER(C.c_double)), ("Vi", C.POINTER(C.c_double)),
                ("sigma", C.POINTER(C.c_double))] + [(n, C.c_double) for n in ("dist_max", "var_max", "var_min", "cutoff_dist", "min_dist", "max_dist", "indel_rate")] \
        + [("mldist", C.c_int32), ("mldist_gap", C.c_int32)]


class pgm_merge_job(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("n1", C.c_uint32), ("n2", C.c_uint32), ("nnodes", C.c_uint32),
                ("sites1", C.POINTER(C.c_double)), ("sites2", C.POINTER(C.c_double)), ("P1", C.POINTER(C.c_double)), ("P2", C.POINTER(C.c_double)),
                ("k1", C.POINTER(C.c_uint32)), ("k2", C.POINTER(C.c_uint32)), ("g2_with_P1", C.POINTER(C.c_uint8)), ("profiles", C.POINTER(C.c_double))]


class pgm_site_ref(C.Structure):
    _fields_ = [("dev_sites", C.POINTER(C.c_double)), ("node_map", C.POINTER(C.c_uint32)), ("ncols", C.c_uint32)]


class pgm_align_out(C.Structure):
    _fields_ = [("score", C.c_float), ("n_tr_indels", C.c_uint32), ("len", C.c_uint32), ("status", C.c_int32),
                ("map1", C.POINTER(C.c_uint32)), ("map2", C.POINTER(C.c_uint32))]


# every symbol include/pgm_hip.h declares
EXPORTS = [
    "pgm_device_count", "pgm_ctx_create", "pgm_ctx_destroy", "pgm_last_error", "pgm_ctx_device_info",
    "pgm_align_graphs_batch", "pgm_align_batch_create", "pgm_align_batch_create_ex", "pgm_align_batch_create_res", "pgm_align_graphs_batch_res", "pgm_align_batch_run", "pgm_align_batch_fetch",
    "pgm_align_batch_destroy", "pgm_align_batch_cells", "pgm_align_batch_test_stall", "pgm_test_cu_shares", "pgm_align_batch_stage_times", "pgm_align_batch_job_times", "pgm_align_batch_time", "pgm_align_batch_read_matrices",
    "pgm_nw_pairs_batch", "pgm_nw_pairs_submit", "pgm_nw_pairs_wait", "pgm_nw_last_kernel_ms", "pgm_host_alloc", "pgm_host_free", "pgm_csprofile_load", "pgm_csprofile_create_batch", "pgm_csprofile_create_batch_res",
    "pgm_csprofile_last_kernel_ms", "pgm_mldist_batch", "pgm_prealigned_counts_batch", "pgm_kmer_cosine", "pgm_dist_last_kernel_ms",
    "pgm_merge_profiles_batch", "pgm_merge_profiles_batch_ex", "pgm_resident_reset", "pgm_resident_onehot", "pgm_resident_import", "pgm_merge_last_kernel_ms",
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("libpgm_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "— there is no CPU fallback for the hot path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
    PG, PM = C.POINTER(C.POINTER(pgm_graph)), C.POINTER(C.POINTER(pgm_model))
    sig = {
        "pgm_device_count": (C.c_int, []),
        "pgm_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "pgm_ctx_destroy": (None, [vp]),
        "pgm_last_error": (C.c_char_p, []),
        "pgm_ctx_device_info": (C.c_int, [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
        "pgm_align_graphs_batch": (C.c_int, [vp, u32, PG, PG, PM, C.POINTER(pgm_scores), C.POINTER(pgm_align_out)]),
        "pgm_align_batch_create": (C.c_int, [vp, u32, PG, PG, PM, C.POINTER(pgm_scores), C.POINTER(vp)]),
        "pgm_align_batch_create_ex": (C.c_int, [vp, u32, PG, PG, PM, C.POINTER(pgm_scores), u32, C.POINTER(vp)]),
        "pgm_align_batch_create_res": (C.c_int, [vp, u32, PG, PG, PM, C.POINTER(pgm_scores), u32, C.POINTER(pgm_site_ref), C.POINTER(pgm_site_ref), C.POINTER(vp)]),
        "pgm_align_graphs_batch_res": (C.c_int, [vp, u32, PG, PG, PM, C.POINTER(pgm_scores), C.POINTER(pgm_site_ref), C.POINTER(pgm_site_ref), C.POINTER(pgm_align_out)]),
        "pgm_align_batch_run": (C.c_int, [vp, vp]),
        "pgm_align_batch_fetch": (C.c_int, [vp, vp, C.POINTER(pgm_align_out)]),
        "pgm_align_batch_destroy": (None, [vp, vp]),
        "pgm_align_batch_cells": (C.c_uint64, [vp]),
        "pgm_align_batch_test_stall": (C.c_int, [vp, u32, u32, u32]),
        "pgm_test_cu_shares": (C.c_int, [u32, C.c_double, u32, C.c_double, u32, C.c_double, u32, u32, C.c_double, C.c_double, u32, C.POINTER(C.c_uint32)]),
        "pgm_align_batch_stage_times": (C.c_int, [vp, C.c_int] + [C.POINTER(C.c_float)] * 3 + [C.POINTER(u32)]),
        "pgm_align_batch_job_times": (C.c_int, [vp, vp, C.POINTER(C.c_uint64)]),
        "pgm_align_batch_time": (C.c_int, [vp, vp, C.c_int] + [C.POINTER(C.c_float)] * 4),
        "pgm_align_batch_read_matrices": (C.c_int, [vp, vp, u32] + [C.POINTER(C.c_float)] * 5),
        "pgm_nw_pairs_batch": (C.c_int, [vp, u32, C.POINTER(i32), i32, i32, u32, C.POINTER(C.c_int8), C.POINTER(u32), u32,
                                         C.POINTER(u32), C.POINTER(u32), C.POINTER(i32), C.POINTER(u32)]),
        "pgm_nw_pairs_submit": (C.c_int, [vp, u32, C.POINTER(i32), i32, i32, u32, C.POINTER(C.c_int8), C.POINTER(u32), u32,
                                          C.POINTER(u32), C.POINTER(u32), u32, C.POINTER(i32), C.POINTER(u32), C.POINTER(C.c_int)]),
        "pgm_nw_pairs_wait": (C.c_int, [vp, C.c_int]),
        "pgm_host_alloc": (vp, [C.c_size_t]),
        "pgm_host_free": (None, [vp]),
        "pgm_nw_last_kernel_ms": (C.c_float, [vp]),
        "pgm_csprofile_load": (C.c_int, [vp, u32, u32] + [C.POINTER(C.c_double)] * 3),
        "pgm_csprofile_create_batch": (C.c_int, [vp, u32, C.POINTER(C.c_int8), C.POINTER(u32)] + [C.POINTER(C.c_double)] * 4
                                       + [C.POINTER(C.c_uint64)]),
        "pgm_csprofile_create_batch_res": (C.c_int, [vp, u32, C.POINTER(C.c_int8), C.POINTER(u32)] + [C.POINTER(C.c_double)] * 3
                                           + [C.POINTER(C.POINTER(C.c_double))]),
        "pgm_csprofile_last_kernel_ms": (C.c_float, [vp]),
        "pgm_mldist_batch": (C.c_int, [vp, C.POINTER(pgm_mldist_model), u32, C.POINTER(i32), C.POINTER(u32)] + [C.POINTER(C.c_double)] * 3),
        "pgm_prealigned_counts_batch": (C.c_int, [vp, u32, u32, u32, C.POINTER(C.c_int8), u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(i32), C.POINTER(u32)]),
        "pgm_kmer_cosine": (C.c_int, [vp, u32, u32, C.POINTER(i32), C.POINTER(C.c_double)]),
        "pgm_dist_last_kernel_ms": (C.c_float, [vp]),
        "pgm_merge_profiles_batch": (C.c_int, [vp, u32, C.POINTER(pgm_merge_job)]),
        "pgm_merge_profiles_batch_ex": (C.c_int, [vp, u32, C.POINTER(pgm_merge_job), u32, C.POINTER(C.POINTER(C.c_double))]),
        "pgm_resident_reset": (C.c_int, [vp]),
        "pgm_resident_onehot": (C.c_int, [vp, u32, u32, C.POINTER(C.c_int8), C.POINTER(u32), C.POINTER(C.POINTER(C.c_double))]),
        "pgm_resident_import": (C.c_int, [vp, vp, C.POINTER(C.c_double), C.c_uint64, C.POINTER(C.POINTER(C.c_double))]),
        "pgm_merge_last_kernel_ms": (C.c_float, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here = the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class PgmError(RuntimeError):
    pass


def check(rc, what="libpgm_hip"):
    if rc != PGM_OK:
        raise PgmError("%s failed (%d): %s" % (what, rc, lib.pgm_last_error().decode()))


class Context:
    """pgm_ctx wrapper.  Raises PgmError when no gfx950 device is usable (no CPU fallback)."""

    def __init__(self, device=0):
        self.handle = C.c_void_p()
        check(lib.pgm_ctx_create(device, C.byref(self.handle)), "pgm_ctx_create")

    def device_info(self):
        buf = C.create_string_buffer(256)
        cu = C.c_int()
        check(lib.pgm_ctx_device_info(self.handle, buf, 256, C.byref(cu)))
        return buf.value.decode(), cu.value

    def close(self):
        if self.handle:
            lib.pgm_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
