import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def wm_lib():
    """libwholegraph.so, built if missing. The product library — never the oracle."""
    from wholegraph_amd import binding
    if not os.path.exists(binding.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    elif os.path.isdir(os.path.join(ROOT, "build", "obj")):
        # development tree (objects present; the GPU box only receives the built .so): keep the in-tree library in step
        # with the sources — a no-op when it is up to date
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "wholegraph_amd", "csrc"), "-j8"], stdout=subprocess.DEVNULL)
    return binding.lib()


@pytest.fixture
def knobs(wm_lib):
    """Environment knobs of the library, changed mid-process: the library reads each WM_* variable once, so every change
    is followed by wholememory_ext_reload_knobs(); the environment is restored (and reloaded) after the test."""
    from wholegraph_amd import binding
    saved = {}

    class Knobs:
        def set(self, name, value):
            saved.setdefault(name, os.environ.get(name))
            os.environ[name] = str(value)
            binding.reload_knobs()

        def unset(self, name):
            saved.setdefault(name, os.environ.get(name))
            os.environ.pop(name, None)
            binding.reload_knobs()

    yield Knobs()
    for name, old in saved.items():
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old
    binding.reload_knobs()


@pytest.fixture(scope="session")
def gpu_env(wm_lib):
    """Initialised library + single-rank communicator on cuda:0."""
    import torch
    from wholegraph_amd import binding
    import wholegraph_amd.torch as wgth
    assert torch.cuda.is_available(), "gpu tests need a GPU; the library has no CPU fallback"
    torch.cuda.set_device(0)
    binding.check(wm_lib.wholememory_init(0, binding.LEVEL_WARN))
    assert wm_lib.wholememory_ext_backend_name() == b"hip-gfx950"
    comm = wgth.create_group_communicator(1)
    yield comm
