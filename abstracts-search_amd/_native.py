"""Build + load the C-ABI HIP libraries (ctypes).

The libraries are built in-tree with hipcc for gfx950 (no JIT cache, so the
.so travels with the repository snapshot).  Loading never falls back to a CPU
implementation: a missing library is an ImportError with the build command.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
_INCLUDE = os.path.join(os.path.dirname(_PKG), "include")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-shared", "-Wall", "-Wextra", "-Wl,-rpath,/opt/rocm/lib"]

LIBS = {
    "ivfpq": ("libmi_ivfpq.so", ["ivfpq.hip"], ["ivfpq_kernels.h", "encoder_kernels.h", "common.h"]),   # ivfpq.hip includes the ring GEMM
    "encoder": ("libmi_encoder.so", ["encoder.hip"], ["encoder_kernels.h", "encoder_few.h", "encoder_mid.h", "common.h"]),
}

_loaded: dict[str, ctypes.CDLL] = {}


def lib_path(name: str) -> str:
    # MI_ENCODER_LIB / MI_IVFPQ_LIB: load another in-tree build of the same library (A/B runs of
    # compile-time variants, tools/ only); it must sit next to the default one
    alt = os.environ.get(f"MI_{name.upper()}_LIB")
    if alt:
        return os.path.join(_PKG, os.path.basename(alt))
    return os.path.join(_PKG, LIBS[name][0])


def _stale(name: str) -> bool:
    so = lib_path(name)
    if not os.path.exists(so):
        return True
    t = os.path.getmtime(so)
    _, srcs, hdrs = LIBS[name]
    deps = [os.path.join(_CSRC, f) for f in srcs + hdrs]
    deps += [os.path.join(_INCLUDE, f) for f in os.listdir(_INCLUDE)] if os.path.isdir(_INCLUDE) else []
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in deps)


def build(name: str, force: bool = False, verbose: bool = False) -> str:
    """hipcc cross-compiles gfx950 without a GPU (seconds)."""
    so = lib_path(name)
    srcs = [os.path.join(_CSRC, f) for f in LIBS[name][1]]
    if not all(os.path.exists(s) for s in srcs):
        raise FileNotFoundError(f"sources of {name} missing: {srcs}")
    if not force and not _stale(name):
        return so
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *HIPCC_FLAGS, *srcs, "-o", so + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(so + ".tmp", so)
    return so


def oa_jsonl_path() -> str:
    return os.path.join(_PKG, "oa_jsonl_mt")


def build_oa_jsonl(force: bool = False) -> str:
    """The host-side OpenAlex text filter (csrc/oa_jsonl_mt.c, plain C + pthreads)."""
    src, exe = os.path.join(_CSRC, "oa_jsonl_mt.c"), oa_jsonl_path()
    if force or not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        cc = shutil.which("gcc") or shutil.which("cc")
        subprocess.check_call([cc, "-O2", "-Wall", "-Wextra", "-pthread", "-o", exe + ".tmp", src])
        os.replace(exe + ".tmp", exe)
    return exe


def build_all(force: bool = False, verbose: bool = False) -> list[str]:
    out = []
    for name, (_, srcs, _) in LIBS.items():
        if all(os.path.exists(os.path.join(_CSRC, s)) for s in srcs):
            out.append(build(name, force=force, verbose=verbose))
    return out


def load(name: str) -> ctypes.CDLL:
    """dlopen the library.  torch is imported first so that its bundled HIP
    runtime (SONAME libamdhip64.so.7) is the one the library binds to: torch
    device pointers and streams are then valid inside our kernels."""
    if name in _loaded:
        return _loaded[name]
    so = lib_path(name)
    if not os.path.exists(so):
        raise ImportError(
            f"{so} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    try:
        import torch  # noqa: F401  (loads libamdhip64 before our library)
    except Exception:  # pragma: no cover - torch is plumbing, the library also works without it
        pass
    lib = ctypes.CDLL(so, mode=ctypes.RTLD_GLOBAL)
    _loaded[name] = lib
    return lib
