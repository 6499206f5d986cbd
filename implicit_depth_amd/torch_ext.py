"""torch_ext.py — loader of the pybind11 torch-extension shim (csrc/lidf_torch_ext.cpp), the
counterpart of the reference's `torch.utils.cpp_extension.load(...)` modules
(extensions/ray_aabb/jit.py:2-3, extensions/pcl_aabb/jit.py): the same C ABI as the ctypes binding
(implicit_depth_amd/_lib.py), with tensor checks, output allocation and the current HIP stream
handled in C++. Functions: ray_aabb, pcl_aabb, compute_ray_aabb, forward_decoders, forward_query.
"""
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(_HERE, "csrc", "lidf_torch_ext.so")
_mod = None


def ext():
    """The extension module; raises if it has not been built (__graft_entry__.build())."""
    global _mod
    if _mod is None:
        if not os.path.exists(EXT_PATH):
            raise RuntimeError("lidf_torch_ext.so not found at %s — run __graft_entry__.build()" % EXT_PATH)
        import torch  # noqa: F401  (libtorch must be loaded first)
        spec = importlib.util.spec_from_file_location("lidf_torch_ext", EXT_PATH)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        _mod = m
    return _mod


def decoder_weights(mod):
    """[w1,b1,...,w4,b4(,offset_enc.weight,offset_enc.bias)] of an IMNet / IEF module."""
    from .decoders import _PARAM_ORDER, _get, _has
    return [_get(mod, k).detach().contiguous() for k in _PARAM_ORDER if _has(mod, k)]
