import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_addoption(parser):
    parser.addoption("--gpt-opt", action="append", default=[], metavar="NAME=VALUE",
                     help="renderer option applied to every Renderer of the run (gpt_set_option), e.g. lds_scene=0 sends "
                          "every scene through the global-memory kernels")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    from gpu_pathtracer_amd import api
    for kv in config.getoption("--gpt-opt"):
        k, v = kv.split("=")
        api.DEFAULT_OPTIONS[k] = int(v)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    import oracle_lib
    oracle_lib.build_oracle()


@pytest.fixture(scope="session")
def gpt():
    """The HIP library through its C ABI; the GPU tests fail loudly if it is not built."""
    from gpu_pathtracer_amd import api
    api.load()
    return api


@pytest.fixture(scope="session")
def gpt_host():
    """The same library for its host-side entry points (scene preparation: no GPU needed)."""
    from gpu_pathtracer_amd import api
    api.load()
    return api
