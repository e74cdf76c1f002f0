"""Loader of the C-ABI shared library (stella_vslam_amd/libsvgpu.so).

There is NO fallback: if the HIP library is missing or fails to load, importing the product path
raises.  (The CPU oracle under oracle/ is test infrastructure and is never used here.)
"""
from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import os

_HERE = pathlib.Path(__file__).resolve().parent
# SVGPU_LIB_PATH selects an A/B build of the SAME library (tools/build_variant.sh) for kernel experiments; there is still no fallback
LIB_PATH = pathlib.Path(os.environ["SVGPU_LIB_PATH"]) if os.environ.get("SVGPU_LIB_PATH") else _HERE / "libsvgpu.so"
_LIB = None


class SvgpuError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: svgpu status {status} ({status_string(status)}) {detail}".rstrip())


def build(force: bool = False) -> pathlib.Path:
    """(Re)build libsvgpu.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = list((_HERE / "csrc").glob("*.hip")) + list((_HERE / "csrc").glob("*.h")) + list((_HERE / "csrc").glob("*.inc"))
    srcs.append(_HERE.parent / "include" / "svgpu.h")
    stale = force or not LIB_PATH.exists() or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", str(_HERE / "csrc"), "-j8"] + (["-B"] if force else []))
    return LIB_PATH


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _LIB = C.CDLL(str(LIB_PATH))
        _LIB.svgpu_last_error.restype = C.c_char_p
        _LIB.svgpu_status_string.restype = C.c_char_p
        _LIB.svgpu_stream.restype = C.c_void_p
        if _LIB.svgpu_abi_version() != 1:
            raise ImportError("libsvgpu.so ABI version mismatch")
    return _LIB


def status_string(status: int) -> str:
    try:
        return lib().svgpu_status_string(status).decode()
    except Exception:
        return "?"
