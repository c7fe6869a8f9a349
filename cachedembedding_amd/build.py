"""Builds libce_hip.so (gfx950) in-tree with hipcc.  No torch types cross the boundary:
the library is plain C ABI (include/ce_api.h) linked against the HIP runtime only."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "libce_hip.so"
# the same library with the fault / delay injection of tests/test_gpu_worker.py compiled in (-DCE_TEST_HOOKS): loaded
# only by the child process of tests/test_gpu_worker.py::test_hook_tests_on_the_test_hooks_build (CE_LIBRARY=...)
LIB_TEST_HOOKS = PKG / "libce_hip_testhooks.so"
HOOKED_SOURCES = ["ce_cache.hip"]          # the sources that read CE_TEST_HOOKS
SOURCES = ["ce_host.hip", "ce_bag.hip", "ce_bag_extra.hip", "ce_cache.hip", "ce_sort.hip", "ce_rowcopy.cpp"]
HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-x", "c++"]       # plain C++ sources (host only)
HEADERS = [ROOT / "include" / "ce_api.h", CSRC / "ce_common.h", CSRC / "ce_cache_index.h", CSRC / "ce_cache_select.h",
           CSRC / "ce_cache_rows.h", CSRC / "ce_cache_fused.h", CSRC / "ce_cache_worker.h"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
         f"--offload-arch={ARCH}", f"-I{ROOT / 'include'}", f"-I{CSRC}"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _extra_flags():
    # CE_BUILD_ABLATIONS=1: compile the backward kernels' ablation switches in (CE_BWD_DEBUG; wrong results by design)
    return ["-DCE_ABLATIONS"] if os.environ.get("CE_BUILD_ABLATIONS") else []


def _digest() -> str:
    h = hashlib.sha256()
    h.update(" ".join(_extra_flags()).encode())
    for p in [CSRC / s for s in SOURCES] + HEADERS + [Path(__file__)]:
        h.update(p.read_bytes())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> Path:
    # two stamps: csrc/.build_stamp is TRACKED (the digest of the sources a commit's numbers were measured on:
    # profiles/traffic.json refers to it); csrc/build/lib.stamp lies with the objects, untracked like the library itself,
    # and is what says which sources the library on disk was built from -- a `git checkout` can put an old tracked
    # stamp back beside a newer library, and the next build() must not take that library for current
    stamp = PKG / "csrc" / ".build_stamp"
    objdir = PKG / "csrc" / "build"
    lib_stamp = objdir / "lib.stamp"
    dig = _digest()
    if LIB.exists() and not force and lib_stamp.exists() and lib_stamp.read_text().strip() == dig:
        if not stamp.exists() or stamp.read_text().strip() != dig:
            stamp.write_text(dig)
        return LIB
    hipcc = _hipcc()
    objdir.mkdir(exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = objdir / (src.rsplit(".", 1)[0] + ".o")
        flags = HOST_FLAGS if src.endswith(".cpp") else FLAGS
        cmd = [hipcc, *flags, *_extra_flags(), "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(str(obj))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    # bind to whatever libamdhip64.so.7 the process already has (torch's bundled copy when
    # loaded from Python, /opt/rocm/lib for a standalone C/C++ host program)
    link = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", str(LIB), "-lpthread",
            "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    stamp.write_text(dig)
    lib_stamp.write_text(dig)
    return LIB


def build_test_hooks(force: bool = False) -> Path:
    """libce_hip_testhooks.so: the objects of build() with HOOKED_SOURCES recompiled under -DCE_TEST_HOOKS"""
    build(force=force)
    objdir = PKG / "csrc" / "build"
    hook_stamp = objdir / "lib_testhooks.stamp"
    dig = _digest()
    if LIB_TEST_HOOKS.exists() and not force and hook_stamp.exists() and hook_stamp.read_text().strip() == dig:
        return LIB_TEST_HOOKS
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        stem = src.rsplit(".", 1)[0]
        if src in HOOKED_SOURCES:
            obj = objdir / (stem + "_testhooks.o")
            cmd = [hipcc, *FLAGS, *_extra_flags(), "-DCE_TEST_HOOKS", "-c", str(CSRC / src), "-o", str(obj)]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src} (-DCE_TEST_HOOKS):\n{r.stdout}")
        else:
            obj = objdir / (stem + ".o")
        objs.append(str(obj))
    link = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", str(LIB_TEST_HOOKS), "-lpthread",
            "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    hook_stamp.write_text(dig)
    return LIB_TEST_HOOKS


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_test_hooks(force="--force" in sys.argv))
