"""Builds the in-tree native libraries (no torch extension machinery needed:
the product boundary is a plain C ABI).

  libkompressor_hip.so      hipcc, gfx950 only   -- the product: csrc/kmp_batch.hip + kmp_deflate.hip + kmp_stream.hip
  libkompressor_hip_abl.so  the same sources with -DKMP_ABLATIONS: the second level-3 parser, the fused kernel and the
                            experiment knobs as environment variables (what the tests of those paths and the A/B tools load)
  libkmpcorpus.so           gcc                   -- seeded synthetic corpus (bench + tests)
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")

HIP_LIB = os.path.join(HERE, "libkompressor_hip.so")
HIP_LIB_ABL = os.path.join(HERE, "libkompressor_hip_abl.so")
HIP_UNITS = ("kmp_batch.hip", "kmp_deflate.hip", "kmp_stream.hip")
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
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


def build_hip(force=False, verbose=False, ablations=False):
    """The translation units are compiled side by side (one hipcc each) and linked into one shared object."""
    target = HIP_LIB_ABL if ablations else HIP_LIB
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "kompressor_hip.h"))
    if not force and not _newer(target, srcs):
        return target
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"] + (["-DKMP_ABLATIONS"] if ablations else [])
    if verbose:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    objs, procs = [], []
    for unit in HIP_UNITS:
        obj = os.path.join(OBJ_DIR, unit.replace(".hip", "_abl.o" if ablations else ".o"))
        objs.append(obj)
        procs.append((unit, subprocess.Popen([_hipcc()] + flags + ["-c", "-o", obj, os.path.join(CSRC, unit)])))
    for unit, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, f"hipcc -c {unit}")
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", target] + objs, check=True)
    return target


def build_corpus(force=False):
    src = os.path.join(CSRC, "corpus.c")
    if not force and not _newer(CORPUS_LIB, [src]):
        return CORPUS_LIB
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-fvisibility=hidden", "-o", CORPUS_LIB, src], check=True)
    return CORPUS_LIB


def find_jni_include():
    """Directories holding a JDK's jni.h and jni_md.h, or None (the build image has no JDK: SURVEY.md section 8c)."""
    roots = [os.environ.get("JAVA_HOME"), "/usr/lib/jvm/default-java", "/usr/lib/jvm/default"]
    if os.path.isdir("/usr/lib/jvm"):
        roots += sorted(os.path.join("/usr/lib/jvm", d) for d in os.listdir("/usr/lib/jvm"))
    for r in roots:
        inc = r and os.path.join(r, "include")
        if inc and os.path.exists(os.path.join(inc, "jni.h")):
            return [inc] + [os.path.join(inc, d) for d in ("linux", "darwin") if os.path.isdir(os.path.join(inc, d))]
    return None


JNI_SHIMS = {"libzstd-jni.so": [os.path.join(ROOT, "jni", "zstd", "Wrapper.cpp"),      # the names ZstdWrapper.kt:10-17 /
                                os.path.join(ROOT, "jni", "zstd", "BatchWrapper.cpp")],  # (+ the batch exports)
             "libz-jni.so": [os.path.join(ROOT, "jni", "zlib", "Wrapper.cpp")]}         # ZlibWrapper.kt load


def build_jni(force=False):
    """The JNI shims over libkompressor_hip.so (jni/*/Wrapper.cpp), built only where a JDK provides jni.h.
    Returns the list of libraries built (empty without a JDK)."""
    inc = find_jni_include()
    if inc is None:
        return []
    out = []
    for name, srcs in JNI_SHIMS.items():
        target = os.path.join(HERE, name)
        deps = srcs + [os.path.join(ROOT, "jni", "common", "kmp_jni.h"), os.path.join(ROOT, "include", "kompressor_hip.h")]
        if force or _newer(target, deps):
            cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden"] + [f"-I{d}" for d in inc] + \
                  ["-o", target] + srcs + [f"-L{HERE}", "-lkompressor_hip", "-Wl,-rpath,$ORIGIN"]
            subprocess.run(cmd, check=True)
        out.append(target)
    return out


def build_all(force=False, verbose=False):
    libs = build_hip(force, verbose), build_corpus(force)
    build_hip(force, verbose, ablations=True)
    build_jni(force)
    return libs


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv, verbose="-v" in sys.argv))
