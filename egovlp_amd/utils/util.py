"""Checkpoint plumbing of the drop-in boundary (reference: utils/util.py:25-51, base/base_trainer.py:399-480).

The reference's checkpoints are `torch.save({'arch', 'epoch', 'state_dict', 'optimizer', 'monitor_best', 'config'})`
where `config` is the live `parse_config.ConfigParser` object (base/base_trainer.py:407-414).  Unpickling such a file
needs (a) `weights_only=False` -- torch >= 2.6 defaults to the safe unpickler, which rejects the ConfigParser global --
and (b) a `parse_config` module on the path, which this package does not ship.  `load_checkpoint_file` therefore reads
TRUSTED checkpoint files with an unpickler that maps every global it cannot import to an inert placeholder: tensors,
optimizer state and plain containers load normally, the pickled config object becomes a `_Placeholder` that still exposes
its `_config` dict (so `checkpoint['config']['arch']` keeps working as in `_resume_checkpoint`).
"""
from __future__ import annotations

import pickle

import torch


def state_dict_data_parallel_fix(load_state_dict, curr_state_dict):
    """utils/util.py:25-51: add / strip the DDP 'module.' prefix so that key sets line up."""
    load_keys = list(load_state_dict.keys())
    curr_keys = list(curr_state_dict.keys())
    redo_dp = False
    undo_dp = False
    if not curr_keys[0].startswith('module.') and load_keys[0].startswith('module.'):
        undo_dp = True
    elif curr_keys[0].startswith('module.') and not load_keys[0].startswith('module.'):
        redo_dp = True
    if undo_dp:
        return type(load_state_dict)((k[7:], v) for k, v in load_state_dict.items())
    if redo_dp:
        return type(load_state_dict)(('module.' + k, v) for k, v in load_state_dict.items())
    return load_state_dict


class _Placeholder:
    """Stands in for a pickled object whose class is not importable here (e.g. parse_config.ConfigParser)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__['_state'] = state

    # ConfigParser-like read access: checkpoint['config']['arch'] (base/base_trainer.py:438,471)
    def __getitem__(self, name):
        cfg = self.__dict__.get('_config')
        if cfg is None:
            raise KeyError(name)
        return cfg[name]


# Globals the lenient unpickler resolves for real: an EXACT (module, name) allow-list -- what tensors, optimizer state, numpy
# scalars / arrays and plain containers are made of: torch's own `weights_only` table plus the numpy reconstructors and a few
# inert value types.  EVERYTHING else -- the reference's ConfigParser, loggers, and whatever a hostile file names (os.system,
# builtins.eval, but also torch.hub.load, torch.load, numpy.load, torch.utils.cpp_extension.load: REDUCE may call any global it
# is handed, so a module-prefix rule is not an allow-list) -- becomes an inert _Placeholder class: it is constructed and given
# its state, it never runs code.
_ALLOWED_BUILTINS = {"set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes", "bytearray", "complex",
                     "slice", "range", "object"}
_NUMPY_NAMES = {"ndarray", "dtype", "bool_", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32", "uint64", "float16",
                "float32", "float64", "complex64", "complex128"}
_EXTRA_ALLOWED = {("collections", "OrderedDict"), ("collections", "defaultdict"), ("collections", "Counter"), ("collections", "deque"),
                  ("_codecs", "encode"), ("pathlib", "PosixPath"), ("pathlib", "PurePosixPath"), ("pathlib", "Path"), ("pathlib", "PurePath"),
                  ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
                  ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar")} | {("numpy", n) for n in _NUMPY_NAMES}
_TORCH_TABLE = None


def _torch_safe_globals():
    """torch's `weights_only=True` allow-list ("module.name" -> object): rebuild functions, storages, dtypes, OrderedDict, ..."""
    global _TORCH_TABLE
    if _TORCH_TABLE is None:
        try:
            from torch._weights_only_unpickler import _get_allowed_globals
            _TORCH_TABLE = dict(_get_allowed_globals())
        except Exception:        # pragma: no cover  (a torch without the private table: only the explicit names below)
            _TORCH_TABLE = {}
        import torch._utils as tu
        for n in ("_rebuild_tensor_v2", "_rebuild_parameter", "_rebuild_tensor", "_rebuild_parameter_with_state"):
            if hasattr(tu, n):
                _TORCH_TABLE.setdefault("torch._utils." + n, getattr(tu, n))
    return _TORCH_TABLE


class _LenientUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        key = f"{module}.{name}"
        table = _torch_safe_globals()
        if key in table:
            return table[key]
        if (module == "builtins" and name in _ALLOWED_BUILTINS) or (module, name) in _EXTRA_ALLOWED:
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                pass
        return type(name, (_Placeholder,), {'__module__': module})


class _LenientPickle:
    """`pickle_module` for torch.load: the stdlib pickle with the lenient Unpickler."""
    __name__ = 'pickle'
    Unpickler = _LenientUnpickler
    load = staticmethod(lambda f, **kw: _LenientUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    Pickler = pickle.Pickler
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def load_checkpoint_file(path, map_location=None, trusted=False):
    """torch.load for checkpoint files.  Plain state_dict files (and any file the safe `weights_only=True` unpickler accepts)
    load on the safe path.  A file the safe unpickler REJECTS -- the reference's own checkpoints pickle their ConfigParser next
    to the weights -- is read with the lenient unpickler only when the caller says the file is `trusted` (a checkpoint the
    user named in the config / on the command line); a warning is logged.  A missing or corrupt file raises as usual: only
    the safe unpickler's refusal (pickle.UnpicklingError) triggers the fallback."""
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:
        if not trusted:
            raise pickle.UnpicklingError(
                f"{path}: not loadable with weights_only=True ({str(e).splitlines()[0][:200]}); pass trusted=True to "
                f"load_checkpoint_file for a reference-format checkpoint you trust") from e
        import warnings
        warnings.warn(f"{path}: the safe unpickler refused this file; reading it with the allow-list unpickler "
                      f"(unknown classes become inert placeholders)")
        return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_LenientPickle)
