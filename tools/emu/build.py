"""Builds tools/emu/libswfr_emu.so: the product's sources compiled as plain C++ against the lock-step wavefront emulator.
Development aid only (kernel debugging, CPU sanitizers); never loaded by the swf_renderer_amd package."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "swf_renderer_amd", "csrc")
LIB = os.path.join(HERE, "libswfr_emu.so")


def build(sanitize=False, opt="-O1"):
    """Compiles what is out of date and links; concurrent callers (the ranks of a multi-process test) take turns."""
    import fcntl
    os.makedirs(os.path.join(HERE, "obj"), exist_ok=True)
    with open(os.path.join(HERE, "obj", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(sanitize, opt)


def _build(sanitize, opt):
    sys.path.insert(0, ROOT)
    from swf_renderer_amd.build import SOURCES
    objs = []
    flags = ["-std=c++17", opt, "-g", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math", "-fno-inline", "-fno-omit-frame-pointer",
             "-fno-optimize-sibling-calls", "-fno-reorder-blocks", "-fno-reorder-blocks-and-partition", "-fno-reorder-functions", "-Wno-attributes",
             "-Wno-unknown-pragmas", "-DSWFR_BUILD", "-I" + os.path.join(HERE, "include")]
    if sanitize:
        flags += ["-fsanitize=undefined", "-fno-sanitize-recover=undefined"]
    os.makedirs(os.path.join(HERE, "obj"), exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "emu_rt.cpp")]
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "obj", os.path.basename(s) + ".o")
        objs.append(o)
        deps = [s] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".hip"))] + [os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "swfr.h")]
        if os.path.exists(o) and all(os.path.getmtime(d) < os.path.getmtime(o) for d in deps) and not sanitize:
            continue
        procs.append(subprocess.Popen(["g++", "-x", "c++"] + flags + ["-c", s, "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("emu build failed")
    if procs or not os.path.exists(LIB):
        tmp = LIB + ".tmp%d" % os.getpid()
        subprocess.check_call(["g++", "-shared", "-o", tmp] + objs + ["-ldl"] + (["-fsanitize=undefined"] if sanitize else []))
        os.replace(tmp, LIB)                              # (atomic: a process that has the old library mapped keeps it)
    return LIB


if __name__ == "__main__":
    print(build(sanitize="--sanitize" in sys.argv, opt="-O0" if "--O0" in sys.argv else "-O1"))
