"""Build maskrcnn/_C_native.so: the reference's pybind module (c++ext/maskrcnn/csrc/vision.cpp:11-15) on libmaskrcnn_hip.so, plus
the TORCH_LIBRARY registration maskrcnn_native::* (maskrcnn/csrc/vision_hip.cpp). Plain g++ against torch's headers; in-tree.
    python maskrcnn/build_native.py [--force]"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "vision_hip.cpp")
OUT = os.path.join(HERE, "_C_native.so")
HIPLIB_DIR = os.path.join(ROOT, "maskrcnn_amd")


def build(force: bool = False) -> str:
    deps = [SRC, os.path.join(ROOT, "include", "maskrcnn_hip.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(d) for d in deps):
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    libdir = ce.library_paths()[0]
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-w", "-DTORCH_EXTENSION_NAME=_C_native", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-I" + os.path.join(ROOT, "include"), "-I" + sysconfig.get_paths()["include"], "-I/opt/rocm/include"]
    cmd += ["-I" + p for p in ce.include_paths()]
    cmd += [SRC, "-L" + libdir, "-Wl,-rpath," + libdir, "-L" + HIPLIB_DIR, "-Wl,-rpath,$ORIGIN/../maskrcnn_amd", "-lmaskrcnn_hip",
            "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python", "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed: {' '.join(cmd)}\n{r.stderr[-4000:]}")
    return OUT


def load():
    """import maskrcnn._C_native (torch first: the module links against libtorch)."""
    import importlib.util
    import torch  # noqa: F401
    if not os.path.exists(OUT):
        raise ImportError(f"{OUT} is missing: run python maskrcnn/build_native.py")
    spec = importlib.util.spec_from_file_location("maskrcnn._C_native", OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
