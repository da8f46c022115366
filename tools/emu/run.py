"""Runs the GPU parity tests (tests/test_gpu_parity.py) against tools/emu/libswfr_emu.so -- the kernels compiled as plain C++ and
executed by the lock-step wavefront emulator -- instead of libswfr.so.  A development aid: it finds kernel bugs (divergent
barriers, out-of-bounds indices, wrong pixels) without a GPU; it proves nothing about the GPU build, whose parity is tested
on the GPU.   usage: python tools/emu/run.py [pytest args ...]      e.g.  python tools/emu/run.py -k scenario -x -q
"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)


def use_emulator():
    import build as emu_build
    lib = emu_build.build(sanitize=bool(os.environ.get("EMU_SANITIZE")))
    from swf_renderer_amd import api
    real = api.library_path
    api.library_path = lambda: lib            # the loader below reads this
    api._LIB = None
    L = api.load_library()
    api.library_path = real
    return L


if __name__ == "__main__":
    use_emulator()
    import pytest
    import torch
    torch.cuda.is_available = lambda: True   # the `gpu` fixture of tests/conftest.py: the emulator plays the device
    os.environ["SWFR_EMULATOR"] = "1"        # (tests skip the cases that are only a matter of time on the emulator)
    args = sys.argv[1:] or ["-x", "-q"]
    sys.exit(pytest.main([os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-p", "no:cacheprovider"] + args))
