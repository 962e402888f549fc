"""In-tree build of the native pieces: libgba.so (HIP, gfx950) and libgbasynth.so (C)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIBGBA = os.path.join(_HERE, "libgba.so")
SRC = os.path.join(_HERE, "csrc", "gba.hip")
HDR = os.path.join(ROOT, "include", "gba.h")
INC = os.path.join(_HERE, "csrc", "gba_kband.inc")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


ISA = os.path.join(_HERE, "csrc", "gba_gfx950.s")      # device assembly of the same source and flags (read by tests/test_isa.py)


def build_libgba(force=False, extra=()):
    if force or _stale(LIBGBA, [SRC, HDR, INC]):
        cmd = ["hipcc"] + HIPCC_FLAGS + list(extra) + ["-o", LIBGBA, SRC]
        subprocess.check_call(cmd)
    if force or _stale(ISA, [SRC, HDR, INC]):
        flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.check_call(["hipcc"] + flags + list(extra) + ["-S", "--cuda-device-only", "-o", ISA, SRC])
    return LIBGBA


HOST_DEMO = os.path.join(_HERE, "host", "refine_demo")


def build_host_demo(force=False):
    """C++ driver of the host shim (blasr_amd/host/gba_refine.hpp), used by the tests."""
    src = os.path.join(_HERE, "host", "refine_demo.cpp")
    hdr = os.path.join(_HERE, "host", "gba_refine.hpp")
    if force or _stale(HOST_DEMO, [src, hdr, HDR, LIBGBA]):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-o", HOST_DEMO, src,
                               "-L" + _HERE, "-lgba", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"])
    return HOST_DEMO


READ_HARNESS = os.path.join(_HERE, "host", "read_harness")


def build_read_harness(force=False):
    """The read-level harness (blasr_amd/host/read_harness.cpp): reads in, filtered and scored alignments out, over the C ABI;
    links the synthetic generator for its input."""
    from . import synth
    src = os.path.join(_HERE, "host", "read_harness.cpp")
    hdr = os.path.join(_HERE, "host", "gba_refine.hpp")
    so = synth.build()
    if force or _stale(READ_HARNESS, [src, hdr, HDR, LIBGBA, so]):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-o", READ_HARNESS, src,
                               "-L" + _HERE, "-lgba", "-L" + os.path.dirname(so), "-lgbasynth",
                               "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,$ORIGIN/../synth", "-Wl,-rpath,/opt/rocm/lib"])
    return READ_HARNESS


def build_all(force=False):
    from . import synth
    build_libgba(force)
    synth.build(force)
    build_host_demo(force)
    build_read_harness(force)


if __name__ == "__main__":
    import sys
    build_all(force="--force" in sys.argv)
    print(LIBGBA)
