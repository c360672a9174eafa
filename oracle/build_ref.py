#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY. Build the reference's own CPU c++ext (nms, crop_forward,
crop_backward) from the sources where they lie under /root/reference, into oracle/_ref/.

 * Sources compiled in place, unmodified: c++ext/maskrcnn/csrc/{vision.cpp, cpu/nms_cpu.cpp,
   cpu/crop_cpu.cpp} (the CUDA half needs THC + nvcc and is unbuildable here).
 * One compatibility overload is force-included (oracle/ref_compat.h) because the torch-1.0
   dispatch call at nms_cpu.cpp:75 no longer resolves on torch 2.x.
 * Output: oracle/_ref/maskrcnn_ref_C.so (a pybind11 module exporting nms, crop_forward,
   crop_backward exactly as csrc/vision.cpp:11-15 defines them). oracle/_ref/ is git-ignored
   but travels to the GPU box with the snapshot.

Nothing is copied from the reference into this repo. If /root/reference is absent (GPU box)
this script is a no-op and the prebuilt .so is used.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MASKRCNN_REFERENCE", "/root/reference")
CSRC = os.path.join(REF, "c++ext", "maskrcnn", "csrc")
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "maskrcnn_ref_C.so")
SOURCES = ["vision.cpp", os.path.join("cpu", "nms_cpu.cpp"), os.path.join("cpu", "crop_cpu.cpp")]


def build(force: bool = False) -> str | None:
    if not os.path.isdir(CSRC):
        return OUT if os.path.exists(OUT) else None
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(HERE, "ref_compat.h"), os.path.abspath(__file__)]
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps)):
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-w",
           "-include", os.path.join(HERE, "ref_compat.h"),
           "-DTORCH_EXTENSION_NAME=maskrcnn_ref_C", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I" + CSRC, "-I" + sysconfig.get_paths()["include"]]
    cmd += ["-I" + p for p in ce.include_paths()]
    cmd += srcs
    libdir = ce.library_paths()[0]
    cmd += ["-L" + libdir, "-Wl,-rpath," + libdir,
            "-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-o", OUT]
    subprocess.run(cmd, check=True)
    return OUT


def load():
    """Import the compiled reference module (torch must be imported first)."""
    import importlib.util
    import torch  # noqa: F401
    if not os.path.exists(OUT):
        return None
    spec = importlib.util.spec_from_file_location("maskrcnn_ref_C", OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print(p if p else "reference not present and no prebuilt oracle/_ref: skipped")
