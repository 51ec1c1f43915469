"""MI355X-native per-frame dense path of "A Simple Stereo SLAM System with Deep Loop Closing".

The product is csrc/ (hand-written HIP for gfx950) behind the C ABI in include/myslam_hip.h, built
into libmyslam_hip.so by build.py.  This package directory is not an importable Python identifier
(it carries the repository's name); load it with `load_package()` from __graft_entry__.py, or

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("myslam_amd", "<dir>/__init__.py",
                                                  submodule_search_locations=["<dir>"])
    mod = importlib.util.module_from_spec(spec); sys.modules["myslam_amd"] = mod; spec.loader.exec_module(mod)

`api` is the ctypes mirror of the reference's operator interface (ORBextractor, DeepLCD, ...) used by
the parity tests and bench.py; it never falls back to a CPU path.
"""
from . import build as _build          # noqa: F401
from . import synth                    # noqa: F401
from . import api                      # noqa: F401
from . import chain                    # noqa: F401  (Frontend / Backend / LoopClosing of the reference over the operators: configs[0]'s host side)


def __getattr__(name):
    if name == "sharded_db":          # imports torch: only needed on the multi-GPU path
        import importlib
        return importlib.import_module(".sharded_db", __name__)
    raise AttributeError(name)

build_library = _build.build
