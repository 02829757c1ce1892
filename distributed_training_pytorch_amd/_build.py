"""In-tree build of the native libraries with hipcc for gfx950 (cross-compiles without a GPU).

  lib/libdtp_hip.so      — all HIP kernels (csrc/*.hip), plain C ABI, no torch headers
  lib/libdtp_runtime.so  — host runtime (csrc/runtime/*.cpp): RCCL bucket transport + bucket
                           scheduler (all-reduce and sharded reduce-scatter/optimizer/all-gather),
                           pinned host ring for the input pipeline; links librccl + libamdhip64

``python -m distributed_training_pytorch_amd._build [--clean]`` or ``build_all(clean=...)``.  Without
``clean`` objects are rebuilt only when their source (or a header) is newer.  The .so files are
git-ignored but travel with the ``gpurun`` snapshot, so GPU boxes never compile.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build command failed:\n  %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def _compile_hip(src, headers, extra=()):
    obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
    if _newer([src] + headers, obj):
        # DTP_BUILD_DEFS: extra defines for diagnostic builds (e.g. "-DDTP_C64_STAMPS": in-kernel time
        # stamps of conv_c64, bench/stamp_c64.py); touch the source to force its rebuild
        defs = os.environ.get("DTP_BUILD_DEFS", "").split()
        _run([HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
              "-I", CSRC, *extra, *defs, "-c", src, "-o", obj])
    return obj


def build_hip(verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h"))
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources under %s" % CSRC)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile_hip(s, headers), srcs))
    out = os.path.join(LIBDIR, "libdtp_hip.so")
    if _newer(objs, out):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out, *objs])
    if verbose:
        print("built", out)
    return out


def build_runtime(verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "runtime", "*.cpp")))
    if not srcs:
        return None
    headers = glob.glob(os.path.join(CSRC, "runtime", "*.h")) + glob.glob(os.path.join(CSRC, "*.h"))
    objs = []
    for s in srcs:
        obj = os.path.join(OBJDIR, "rt_" + os.path.basename(s) + ".o")
        if _newer([s] + headers, obj):
            _run([HIPCC, "-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I", CSRC,
                  "-I", os.path.join(ROCM, "include"), "-c", s, "-o", obj])
        objs.append(obj)
    out = os.path.join(LIBDIR, "libdtp_runtime.so")
    if _newer(objs, out):
        _run([HIPCC, "-shared", "-fPIC", "-o", out, *objs, "-L", os.path.join(ROCM, "lib"),
              "-lrccl", "-lamdhip64", "-lpthread", "-ldl", "-Wl,-rpath," + os.path.join(ROCM, "lib")])
    if verbose:
        print("built", out)
    return out


def clean():
    """Remove every object file and library: the next build compiles everything from source."""
    import shutil
    shutil.rmtree(OBJDIR, ignore_errors=True)
    for name in ("libdtp_hip.so", "libdtp_runtime.so"):
        try:
            os.remove(os.path.join(LIBDIR, name))
        except FileNotFoundError:
            pass


def build_all(verbose=False, clean_first=False):
    if clean_first:
        clean()
    a = build_hip(verbose)
    b = build_runtime(verbose)
    return [p for p in (a, b) if p]


if __name__ == "__main__":
    for p in build_all(verbose=True, clean_first="--clean" in sys.argv[1:]):
        print(p)
    sys.exit(0)
