import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


import os
import subprocess
import sys

import pytest

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DBG_LIB = os.path.join(_ROOT, "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so")


@pytest.fixture
def debug_options(request):
    """The alternative launch schedules / kernel variants behind dd_debug_set_option exist only in the measurement build
    (lib/libdecompdiff_hip_dbg.so, -DDD_DEBUG_OPTIONS=1; the default library compiles none of them).  A test that compares
    them with the shipped path asks for this fixture: inside a process that runs on the measurement build it is True;
    otherwise the test is re-run in a subprocess on that build (DD_HIP_LIB), its verdict is asserted, and the fixture is
    False -- the in-process body returns at once."""
    if os.environ.get("DD_HIP_LIB", "").endswith("_dbg.so"):
        return True
    assert os.path.exists(_DBG_LIB), "measurement build missing: python -m decompdiff_amd.build --debug-options"
    env = dict(os.environ, DD_HIP_LIB=_DBG_LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", request.node.nodeid],
                       cwd=_ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-1500:])
    return False
