import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) is mostly the kernel sources under the fiber emulation: one core per test, ~15 minutes in a row.  When pytest-xdist is there and
    the caller asked for no worker count, the tests are spread over a few worker processes (round 5; TH_TEST_WORKERS=0 keeps one process, =N asks for N).  The GPU suite
    is never parallelised: its tests share one device.  (Runs before xdist's own pytest_cmdline_main, which turns the worker count into its worker specs.)"""
    want = os.environ.get("TH_TEST_WORKERS", "")
    opt = config.option
    if want == "0" or "not gpu" not in (getattr(opt, "markexpr", "") or "") or not hasattr(opt, "numprocesses") or opt.numprocesses or getattr(opt, "collectonly", False):
        return None
    if getattr(opt, "dist", "no") != "no" or hasattr(config, "workerinput"):
        return None
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = int(want) if want.isdigit() else min(4, cpus // 2)
    if n >= 2:
        # what the workers would otherwise build at the same moment: the emulation and oracle libraries, once, here
        for d in ("oracle", os.path.join("tests", "cpu_emul")):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, d)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        opt.numprocesses = n
        opt.dist = "load"
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def refso():
    """oracle/_ref/libthref.so: the reference's own ksw2/edlib/seq code compiled where it lies."""
    import oracle_lib
    lib = oracle_lib.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libthref.so not built (needs /root/reference)")
    return lib
