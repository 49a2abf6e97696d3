import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Kernel tests are written once and parametrised over a `dev` fixture:
    dev='emu' (CPU SIMT emulator build of the same .hip sources; not gpu) and dev='gpu'."""
    for item in items:
        if "dev" in getattr(item, "fixturenames", ()):
            if item.callspec.params.get("dev") == "gpu":
                item.add_marker(pytest.mark.gpu)


@pytest.fixture(params=["emu", "gpu"])
def dev(request):
    """Binds pase_amd's kernel wrappers either to the emulator .so (tensors on CPU) or to the real
    gfx950 .so (tensors on cuda:0).  Returns the torch device to allocate on."""
    import torch
    from pase_amd import _lib
    if request.param == "emu":
        from pase_amd import build
        so = build.build_emu()
        _lib.use_library(so, "cpu")
        yield torch.device("cpu")
        _lib.use_library(None, "cuda")
    else:
        if not torch.cuda.is_available():
            pytest.skip("no GPU")
        _lib.use_library(None, "cuda")
        _lib.lib()   # raises loudly if libpase_hip.so is missing
        yield torch.device("cuda:0")
