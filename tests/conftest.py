"""Shared pytest configuration.

* ``gpu`` marker: tests that need a real MI355X (run with ``-m gpu``); everything else runs on CPU.
* The CPU suite never touches a GPU: it checks the oracle against the reference's own test properties and the
  committed golden vectors, the host-side logic, and that the C-ABI library loads and exports every declared symbol.
"""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT / "orthogonal-additive-gaussian-processes_amd"), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = Path(__file__).resolve().parent / "golden"

# Single-node runs: RCCL's bootstrap needs no real network; pin it to loopback so that a box with an odd interface list cannot
# stall communicator creation (the data path is xGMI / device memory either way).
import os as _os0
_os0.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); deselected by -m 'not gpu'")


# Watchdog: a test that makes no progress for WATCHDOG_S seconds (a wedged device call cannot be interrupted from Python)
# dumps every thread's stack and ends the process with a failure instead of hanging the run.
import os as _os
WATCHDOG_S = int(_os.environ.get("OAK_TEST_WATCHDOG_S", "420"))


def _post_mortem(nodeid):
    """Ten seconds before the watchdog ends the process: every thread's Python stack and the library's own view of its contexts
    (which stream is still busy, the last phases / collectives enqueued) into gpurun_out/watchdog/, so that a stall leaves
    evidence behind even when the terminal log is lost."""
    import faulthandler
    try:
        out = ROOT / "gpurun_out" / "watchdog"
        out.mkdir(parents=True, exist_ok=True)
        name = "".join(ch if ch.isalnum() or ch in "-_." else "_" for ch in nodeid)[-150:]
        with open(out / f"{name}.txt", "w") as f:
            f.write(f"watchdog: {nodeid} made no progress for {WATCHDOG_S - 10} s (pid {_os.getpid()})\n\n")
            try:
                from oak import _capi
                f.write(_capi.debug_state() + "\n")
            except Exception as ex:                                  # noqa: BLE001
                f.write(f"(no library state: {ex!r})\n")
            f.flush()
            faulthandler.dump_traceback(file=f, all_threads=True)
    except Exception:                                                # noqa: BLE001
        pass


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_protocol(item, nextitem):
    import faulthandler
    import threading
    faulthandler.dump_traceback_later(WATCHDOG_S, exit=True)
    timer = threading.Timer(max(WATCHDOG_S - 10, 1), _post_mortem, args=(item.nodeid,))
    timer.daemon = True
    timer.start()
    try:
        yield
    finally:
        timer.cancel()
        faulthandler.cancel_dump_traceback_later()


@pytest.fixture
def concrete_normalised_10_rows_data():
    """10 rows x 7 columns of normalised UCI-concrete inputs + targets: the data fixture of the reference's
    tests/conftest.py:11-41, stored as data in tests/golden/concrete_10rows.json."""
    d = json.loads((GOLDEN / "concrete_10rows.json").read_text())
    return np.array(d["X"]), np.array(d["y"])


@pytest.fixture(scope="session")
def hip():
    """Session-wide HIP context; the test FAILS (not skips) if the native library or the device is missing."""
    from oak import _capi
    return _capi.HipContext(0)
