"""ctypes binding of libraffthip.so (include/rafft_hip.h).  No CPU fallback: if the
HIP library or a GPU is missing every call raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAFFT_LIB") or os.path.join(_HERE, "libraffthip.so")      # (RAFFT_LIB: another build of the library, for A/B measurements)

OK, ERR_BAD_CHAR, ERR_EMPTY, ERR_TOO_LONG, ERR_TEMP, ERR_CAPACITY, ERR_PARAM, ERR_HIP, ERR_STRUCT, ERR_NO_DEVICE = range(10)


class Params(C.Structure):
    _fields_ = [("nb_mode", C.c_int32), ("max_stack", C.c_int32), ("max_branch", C.c_int32), ("min_hp", C.c_int32),
                ("min_nrj", C.c_double), ("traj", C.c_int32), ("_pad", C.c_int32), ("temp", C.c_double),
                ("gc_wei", C.c_double), ("au_wei", C.c_double), ("gu_wei", C.c_double)]


class SeqResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("length", C.c_int32), ("n_steps", C.c_int32), ("n_structs", C.c_int32),
                ("step_size", C.POINTER(C.c_int32)), ("step_off", C.POINTER(C.c_int32)),
                ("db", C.POINTER(C.c_char)), ("dcal", C.POINTER(C.c_int32))]


class Result(C.Structure):
    _fields_ = [("n_seq", C.c_int32), ("n_failed", C.c_int32), ("seq", C.POINTER(SeqResult)), ("_owner", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_expand", C.c_double), ("ms_expand_c1", C.c_double),
                ("ms_expand_c2", C.c_double), ("ms_expand_c3", C.c_double), ("ms_expand_wall", C.c_double), ("ms_beam", C.c_double),
                ("ms_materialize", C.c_double), ("ms_output", C.c_double),
                ("n_expand_launches", C.c_int64), ("n_steps", C.c_int64), ("n_node_expansions", C.c_int64),
                ("n_nodes_created", C.c_int64), ("n_nodes_aliased", C.c_int64),
                ("sum_node_len", C.c_int64), ("sum_lags", C.c_int64), ("n_structs", C.c_int64),
                ("n_children", C.c_int64), ("sum_struct_len", C.c_int64), ("alg_bytes", C.c_int64),
                ("alg_bytes_expand", C.c_int64), ("alg_bytes_expand_all", C.c_int64), ("n_regrows", C.c_int64),
                ("alg_bytes_expand_small", C.c_int64), ("alg_bytes_expand_c2", C.c_int64), ("alg_bytes_expand_c3", C.c_int64),
                ("alg_bytes_beam", C.c_int64), ("n_node_instances", C.c_int64), ("n_dE_evals", C.c_int64), ("n_dE_guessed", C.c_int64),
                ("n_kept_guessed", C.c_int64), ("n_regrows_prod", C.c_int64), ("n_waves_long_lists", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


EXPORTS = ["rafft_init", "rafft_fold_batch", "rafft_fold_submit", "rafft_fold_wait", "rafft_free_result", "rafft_last_error", "rafft_eval_structure",
           "rafft_eval_structures", "rafft_eval_structures_at", "rafft_expand_node", "rafft_get_stats", "rafft_version",
           "rafft_load_params", "rafft_load_params_text", "rafft_reset_params", "rafft_save_params", "rafft_params_info",
           "rafft_param_value", "rafft_kin_rate_matrix", "rafft_shutdown", "rafft_alloc_counters", "rafft_eval_structures_info", "rafft_params_unpinned"]

_lib = None


class RafftError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libraffthip error {code}: {msg}")
        self.code = code


def _one_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's); if
    libraffthip.so pulled in the system one first, a later `import torch` would load a second runtime that finds no
    GPU ("No HIP GPUs are available" - measured).  So when torch is installed its runtime is loaded first and
    libraffthip.so binds to it; without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(hip):
        try:
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load the HIP library (building it is __graft_entry__.build()'s / rafft_amd.build's job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing - run `python -m rafft_amd.build` (hipcc, gfx950). "
                          "rafft_amd has no CPU fallback.")
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.rafft_init.argtypes = [C.c_int]
    L.rafft_fold_batch.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int,
                                   C.POINTER(C.POINTER(Result))]
    L.rafft_fold_submit.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int,
                                    C.POINTER(C.c_void_p)]
    L.rafft_fold_wait.argtypes = [C.c_void_p, C.POINTER(C.POINTER(Result))]
    L.rafft_free_result.argtypes = [C.POINTER(Result)]
    L.rafft_free_result.restype = None
    L.rafft_shutdown.restype = None
    L.rafft_shutdown.argtypes = []
    L.rafft_alloc_counters.restype = None
    L.rafft_alloc_counters.argtypes = [C.POINTER(C.c_ulonglong * 5)]
    L.rafft_last_error.restype = C.c_char_p
    L.rafft_version.restype = C.c_char_p
    L.rafft_eval_structure.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    L.rafft_eval_structures.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
    L.rafft_expand_node.argtypes = [C.POINTER(Params), C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.rafft_eval_structures_at.argtypes = [C.c_double] + L.rafft_eval_structures.argtypes
    L.rafft_eval_structures_info.argtypes = L.rafft_eval_structures.argtypes + [C.POINTER(C.c_int)]
    L.rafft_params_unpinned.argtypes = [C.POINTER(C.c_int * 3)]
    L.rafft_load_params.argtypes = [C.c_char_p]
    L.rafft_load_params_text.argtypes = [C.c_char_p, C.c_char_p]
    L.rafft_save_params.argtypes = [C.c_char_p]
    L.rafft_params_info.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.rafft_param_value.argtypes = [C.c_char_p, C.c_int, C.c_long, C.POINTER(C.c_int)]
    L.rafft_get_stats.argtypes = [C.POINTER(Stats)]
    L.rafft_kin_rate_matrix.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_char_p, C.POINTER(C.c_int), C.c_int,
                                        C.POINTER(C.c_double), C.c_double, C.c_void_p]
    _lib = L
    return L


def check(rc):
    if rc:
        raise RafftError(rc, lib().rafft_last_error().decode())
