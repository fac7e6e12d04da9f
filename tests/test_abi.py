"""The drop-in boundary: libasgart_hip.so loads without a GPU, exports every symbol declared in
include/asgart_hip.h, has the documented struct layouts, and refuses to compute without a
device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import asgart_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "asgart_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(asgart_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(hiplib):
    declared = _declared_functions()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(hiplib, name), name
    assert set(declared) == set(asgart_amd.ABI_SYMBOLS)


def test_struct_layouts():
    assert C.sizeof(asgart_amd._Settings) == 40           # 8 + 4(+4) + 8 + 8 + 1 + 1 (+6)
    assert asgart_amd._Settings.max_gap_size.offset == 8
    assert asgart_amd._Settings.min_duplication_length.offset == 16
    assert asgart_amd._Settings.reverse.offset == 32
    assert C.sizeof(asgart_amd.Stats) == 5 * 8 + 13 * 8 + 8 + 8 + 8 + 2 * 8 + 8 + 8 + 8 + 8 + 2 * 8


def test_version_and_settings_from_cli(hiplib):
    assert b"gfx950" in hiplib.asgart_version()
    s = asgart_amd.RunSettings.from_cli(k=20, gap=100)
    assert s.max_gap_size == 120                          # src/bin/asgart.rs:681


def test_no_cpu_fallback_without_device(hiplib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; this checks the GPU-less behaviour")
    with pytest.raises(asgart_amd.AsgartError) as e:
        asgart_amd.Index(b"ACGTACGTACGT$")
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)


def test_bad_arguments_are_reported(hiplib):
    h = C.c_void_p()
    assert hiplib.asgart_index_create(None, 0, None, 0, 0, C.byref(h)) == -1
    assert b"empty" in hiplib.asgart_last_error()
    assert hiplib.asgart_index_prepare(None, 20) == -1


def test_the_library_carries_the_gfx950_kernels_of_the_path(hiplib):
    """The shared object is the product: the hand-written kernels of the hot path must be IN it (a build that silently
    dropped the device code would still export every host symbol).  Their mangled names are in the embedded code object."""
    blob = open(hiplib._name, "rb").read()
    for kernel in (b"probe_count_kernel", b"collect_pending_kernel", b"extend_kernel", b"extend_fast_kernel",
                   b"extend_k8_kernel", b"extend_heavy_kernel", b"scan_segments_kernel", b"cluster_barren_kernel",
                   b"plan_ranges_kernel", b"validate_cuts_kernel"):
        assert kernel in blob, kernel
    assert b"gfx950" in blob


def test_native_stack_dump_names_the_threads(hiplib, capfd):
    """asgart_debug_dump_stacks (what tests/conftest.py calls a minute before a stalled GPU test is ended): every thread of
    the process writes its native stack to stderr, a thread blocked in a system call included."""
    import threading
    import time

    stop = threading.Event()
    t = threading.Thread(target=lambda: stop.wait(30))
    t.start()
    time.sleep(0.05)
    n = hiplib.asgart_debug_dump_stacks()
    stop.set()
    t.join()
    err = capfd.readouterr().err
    assert n >= 2 and err.count("---- native stack of thread") == n
    assert "libasgart_hip" in err       # the dumping thread itself is inside the library
