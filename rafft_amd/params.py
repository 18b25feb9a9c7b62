"""Energy parameters of the fold engine.

The reference takes them from ViennaRNA: `Glob_parms` builds `RNA.md()`, sets `md.temperature` and
`RNA.fold_compound(sequence, md)` (rafft/utils.py:17-21) - the parameter set currently loaded in ViennaRNA,
rescaled to the temperature.  Here ViennaRNA is touched once, up front, for the tables and never inside the fold:

* `load_params(path)` reads a ViennaRNA 2.x parameter file (misc/rna_turner2004.par, or what `RNA.params_save`
  writes) into libraffthip.so; with it `temp != 37` works (the file carries the enthalpies) and the energies are
  ViennaRNA's by construction;
* `load_params_from_viennarna()` does that through an installed `RNA` module (`RNA.params_save` to a temporary
  file) - called automatically before the first fold when `RNA` is importable, so a host that has ViennaRNA gets
  ViennaRNA's own tables;
* the environment variable RAFFT_PARAMS=<file> is honoured before that;
* otherwise the built-in 37 C tables are used (the published Turner-2004 model arbitrated by the reference's
  11 505 energy rows - DESIGN.md section 2 says which entries those rows pin).
"""
import ctypes as C
import os
import tempfile

from . import _native as N

_auto_done = False


def load_params(path):
    """counterpart of RNA.params_load(path) / RNA.read_parameter_file(path)"""
    global _auto_done
    N.check(N.lib().rafft_load_params(os.fspath(path).encode()))
    _auto_done = True


def load_params_text(text, source_name="<memory>"):
    global _auto_done
    N.check(N.lib().rafft_load_params_text(text.encode(), source_name.encode()))
    _auto_done = True


def reset_params():
    """back to the built-in 37 C tables (and no automatic loading afterwards)"""
    global _auto_done
    N.check(N.lib().rafft_reset_params())
    _auto_done = True


def save_params(path):
    """counterpart of RNA.params_save(path): the current set in ViennaRNA's parameter file format"""
    N.check(N.lib().rafft_save_params(os.fspath(path).encode()))


def params_info():
    buf = C.create_string_buffer(1024)
    has = C.c_int()
    N.check(N.lib().rafft_params_info(buf, len(buf), C.byref(has)))
    return {"source": buf.value.decode(), "has_enthalpies": bool(has.value)}


def unpinned_entries():
    """How many entries of the current 1x1 / 2x1 / 2x2 interior-loop tables are rule / model values that no reference-held
    energy row exercises (the built-in set; all zero with a loaded ViennaRNA parameter file)."""
    c = (C.c_int * 3)()
    N.check(N.lib().rafft_params_unpinned(C.byref(c)))
    return {"int11": c[0], "int21": c[1], "int22": c[2]}


def param_value(table, index, enthalpy=False):
    v = C.c_int()
    N.check(N.lib().rafft_param_value(table.encode(), 1 if enthalpy else 0, int(index), C.byref(v)))
    return v.value


def load_params_from_viennarna():
    """The tables of the installed ViennaRNA (`import RNA`), through RNA.params_save.  Returns False when ViennaRNA
    is not importable or has no params_save (nothing is changed then)."""
    try:
        import RNA
    except ImportError:
        return False
    save = getattr(RNA, "params_save", None)
    if save is None:
        return False
    fd, path = tempfile.mkstemp(suffix=".par", prefix="rafft_vrna_")
    os.close(fd)
    try:
        save(path)
        N.check(N.lib().rafft_load_params_text(open(path).read().encode(),
                                               f"ViennaRNA {getattr(RNA, '__version__', '?')} (RNA.params_save)".encode()))
    finally:
        os.unlink(path)
    return True


def ensure_default_params():
    """Called before the first fold of a process: RAFFT_PARAMS, else an installed ViennaRNA, else built-in."""
    global _auto_done
    if _auto_done:
        return
    _auto_done = True
    env = os.environ.get("RAFFT_PARAMS")
    if env:
        # a parameter file that was asked for and cannot be read is an error at EVERY fold - never a silent fall back to the built-in
        # tables (whose unexercised entries are rule / model values: the caller asked for ViennaRNA's own)
        try:
            load_params(env)
        except Exception as e:
            _auto_done = False
            raise RuntimeError(f"RAFFT_PARAMS={env!r}: the parameter file could not be loaded ({e}); unset RAFFT_PARAMS to fold with "
                               "the built-in 37 C tables") from e
    elif os.environ.get("RAFFT_NO_VIENNARNA") is None:
        load_params_from_viennarna()
