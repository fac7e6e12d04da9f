import os
import subprocess
import sys
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# An index builds its presence filters and position-sorted lists on SECOND use by default (option lazy_aux: they do not
# pay back within one pass).  Most tests want every first call on the filtered / list-assisted paths -- that is where
# the code is -- so the default here is 0; the tests named in LAZY_BOTH run under BOTH values (fixture lazy_aux), so
# that the builds a search call makes for itself when it is the second of its kind -- index_prepare_sap /
# index_prepare_filter under acquire_all, the path two stalled fuzz processes of round 4 were in -- run inside the
# parity, tier, shard and passes tests as well, not only in test_lazy_filter_and_lists_same_results.
os.environ.setdefault("ASGART_LAZY_AUX", "0")
LAZY_BOTH = {
    "test_battery_families_and_csr", "test_random_sweep_default_and_forced_tiers",
    "test_passes_entry_point_equals_single_calls", "test_specialised_wave_kernel_with_generation_wraps",
    "test_escalation_tiers_give_identical_results", "test_shards_concatenate_to_unsharded",
    "test_progress_array_and_pipelined_calls", "test_k8_free_counts_do_not_depend_on_timing",
}
TEST_TIMEOUT_S = 900


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_generate_tests(metafunc):
    if metafunc.function.__name__ in LAZY_BOTH:
        if "lazy_aux" not in metafunc.fixturenames:
            metafunc.fixturenames.append("lazy_aux")
        metafunc.parametrize("lazy_aux", ["0", "1"], indirect=True, ids=["eager", "lazy"])


@pytest.fixture
def lazy_aux(request, monkeypatch):
    monkeypatch.setenv("ASGART_LAZY_AUX", request.param)
    return request.param


def pytest_collection_modifyitems(config, items):
    """A test that hangs (a kernel that never returns, a lost rendezvous) ends the run with the stacks of every thread
    instead of stalling it: pytest-timeout, when it is installed, bounds each test ("thread" method: a main thread
    blocked inside a HIP call never returns to the interpreter, so a signal handler would not run)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(TEST_TIMEOUT_S, method="thread"))


def dump_native_stacks(reason, out=sys.stderr):
    """The NATIVE stacks of every thread of this process (a Python-level dump shows a wait inside the HIP runtime as
    one ctypes frame): first a debugger from a child process -- gdb / rocgdb `thread apply all bt`, which also resolves
    symbols of stripped runtime libraries' exports and shows kernel-side waits --, then the library's own in-process dump
    (asgart_debug_dump_stacks: glibc backtrace from a signal handler), which needs no tool and no ptrace permission."""
    print(f"\n==== native stacks ({reason}) ====", file=out, flush=True)
    try:
        import ctypes

        ctypes.CDLL(None).prctl(0x59616D61, ctypes.c_ulong(-1 & 0xFFFFFFFFFFFFFFFF), 0, 0, 0)  # PR_SET_PTRACER_ANY
    except Exception:
        pass
    for dbg in ("gdb", "/opt/rocm/bin/rocgdb", "rocgdb"):
        try:
            r = subprocess.run([dbg, "-p", str(os.getpid()), "-batch", "-ex", "set pagination off", "-ex",
                                "thread apply all bt"], capture_output=True, text=True, timeout=90)
        except Exception as e:  # not installed, not allowed, too slow
            print(f"[{dbg}: {type(e).__name__}]", file=out, flush=True)
            continue
        print(r.stdout[-20000:], file=out, flush=True)
        if "Thread" in r.stdout:
            break
        print(r.stderr[-2000:], file=out, flush=True)
    try:
        import asgart_amd

        n = asgart_amd.load_library().asgart_debug_dump_stacks()
        print(f"[asgart_debug_dump_stacks: {n} threads]", file=out, flush=True)
    except Exception as e:
        print(f"[asgart_debug_dump_stacks: {type(e).__name__}: {e}]", file=out, flush=True)


@pytest.fixture(autouse=True)
def _native_stacks_before_the_timeout(request):
    """A minute before pytest-timeout ends a stalled run, a timer thread writes the native stacks: the next stall names
    the runtime call it sits in (the three of round 4 ended in `asgart_index_destroy` / a build's read-back and nobody
    could say which call)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    t = threading.Timer(TEST_TIMEOUT_S - 100, dump_native_stacks, args=(f"{request.node.nodeid}: {TEST_TIMEOUT_S - 100} s",))
    t.daemon = True
    t.start()
    yield
    t.cancel()


@pytest.fixture(scope="session")
def hiplib():
    """The product library.  GPU tests must exercise it, never a fallback."""
    import asgart_amd

    return asgart_amd.load_library()
