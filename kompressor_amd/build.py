"""Builds the in-tree native libraries (no torch extension machinery needed:
the product boundary is a plain C ABI).

  libkompressor_hip.so  hipcc, gfx950 only   -- the product
  libkmpcorpus.so       gcc                   -- seeded synthetic corpus (bench + tests)
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")

HIP_LIB = os.path.join(HERE, "libkompressor_hip.so")
CORPUS_LIB = os.path.join(HERE, "libkmpcorpus.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the kompressor_amd backend needs ROCm to build")


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "kompressor_hip.h"))
    if not force and not _newer(HIP_LIB, srcs):
        return HIP_LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
           "-o", HIP_LIB, os.path.join(CSRC, "kmp_api.hip")]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return HIP_LIB


def build_corpus(force=False):
    src = os.path.join(CSRC, "corpus.c")
    if not force and not _newer(CORPUS_LIB, [src]):
        return CORPUS_LIB
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-fvisibility=hidden", "-o", CORPUS_LIB, src], check=True)
    return CORPUS_LIB


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_corpus(force)


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv, verbose="-v" in sys.argv))
