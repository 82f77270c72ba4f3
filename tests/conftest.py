import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle (torch CPU, fp32) is what the GPU suite spends its time in.  On a 256-CPU GPU host torch's default thread
    # count makes its small ops SLOWER (bench.py's probe: 16 threads 5.6 s per oracle step, 64 threads 9.3 s): cap it.
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 16)))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The measured parity numbers of the chain / step tests, printed at every verbosity (golden_utils.note_parity)."""
    try:
        import golden_utils as GU
    except Exception:                                              # noqa: BLE001
        return
    if GU.PARITY_LINES:
        terminalreporter.write_line("")
        terminalreporter.write_line("parity against the reference's fixtures, as measured in this run:")
        for line in GU.PARITY_LINES:
            terminalreporter.write_line("  " + line)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


import os
import subprocess
import sys

import pytest

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DBG_LIB = os.path.join(_ROOT, "decompdiff_amd", "lib", "libdecompdiff_hip_dbg.so")


_DBG_RESULTS = {}


def _run_debug_options_tests(session):
    """ONE subprocess on the measurement build for every test of the session that asks for `debug_options` (each used to
    spawn its own: an interpreter start, a torch import and a library load per test)."""
    ids = [it.nodeid for it in session.items if "debug_options" in getattr(it, "fixturenames", ())]
    env = dict(os.environ, DD_HIP_LIB=_DBG_LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-rA"] + ids,
                       cwd=_ROOT, env=env, capture_output=True, text=True, timeout=1500)
    for line in r.stdout.splitlines():
        for verdict in ("PASSED", "FAILED", "ERROR", "SKIPPED"):
            if line.startswith(verdict + " "):
                _DBG_RESULTS[line[len(verdict) + 1:].split(" - ")[0].strip()] = verdict
    _DBG_RESULTS["__log__"] = r.stdout[-6000:] + r.stderr[-2000:]
    _DBG_RESULTS["__rc__"] = r.returncode


@pytest.fixture
def debug_options(request):
    """The alternative launch schedules / kernel variants behind dd_debug_set_option exist only in the measurement build
    (lib/libdecompdiff_hip_dbg.so, -DDD_DEBUG_OPTIONS=1; the default library compiles none of them).  A test that compares
    them with the shipped path asks for this fixture: inside a process that runs on the measurement build it is True;
    otherwise all such tests of the session are run once, together, in a subprocess on that build (DD_HIP_LIB), this
    test's verdict there is asserted, and the fixture is False -- the in-process body returns at once."""
    if os.environ.get("DD_HIP_LIB", "").endswith("_dbg.so"):
        return True
    assert os.path.exists(_DBG_LIB), "measurement build missing: python -m decompdiff_amd.build --debug-options"
    if not _DBG_RESULTS:
        _run_debug_options_tests(request.session)
    verdict = _DBG_RESULTS.get(request.node.nodeid)
    assert verdict == "PASSED", (f"{request.node.nodeid} on the measurement build: {verdict} (subprocess rc "
                                 f"{_DBG_RESULTS.get('__rc__')})\n" + _DBG_RESULTS.get("__log__", ""))
    return False
