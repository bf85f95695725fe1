import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built(request):
    """Build (if stale) every native library; hipcc cross-compiles gfx950 without a GPU."""
    from vecchio_amd import build
    build.build_host()
    build.build_oracle()
    build.build_emu()
    build.build_device()
    # the debug library (instrumented kernels, arithmetic probe) only serves -m gpu tests: a CPU-only run does not compile the device
    # sources a second time (ffi.load_debug_lib builds it on first use anyway)
    if "not gpu" not in (request.config.getoption("-m") or ""):
        build.build_device_debug()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    import oracle_ffi
    oracle_ffi.load()
    return oracle_ffi


@pytest.fixture(scope="session")
def emu(built):
    import emu_ffi
    emu_ffi.load()
    return emu_ffi


@pytest.fixture(scope="session")
def host_scenes(built):
    from vecchio_amd import HostScene
    cache = {}

    def get(name, seed=1):
        key = (name, seed)
        if key not in cache:
            hs = HostScene(name, seed)
            cache[key] = (hs, hs.next_camera())
        return cache[key]

    return get


@pytest.fixture(scope="session")
def device(built):
    """The HIP library on a real GPU; fails loudly (never falls back) when it is unusable."""
    from vecchio_amd import ffi
    # torch first: its wheel carries its own HIP runtime, which finds no GPU when it is initialised after the system one
    # that libvecchio_amd.so links (tests that hand torch tensors to vk_render_device would then depend on test order)
    import torch
    torch.cuda.init()
    lib = ffi.load_device_lib()
    n = lib.vk_device_count()
    assert n >= 1, "no gfx950 device visible: -m gpu tests need the MI355X box"
    return lib
