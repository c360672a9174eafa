"""Build libmaskrcnn_hip.so (hipcc, gfx950 only) in-tree: maskrcnn_amd/libmaskrcnn_hip.so.

    python maskrcnn_amd/build.py [--force] [--verbose]   (run as a script: importing the package needs the built library)
    python maskrcnn_amd/build.py --variant NAME [--only a.hip,b.hip] [-DMACRO ...]   an experiment build: objects and library under
        maskrcnn_amd/csrc/build/variants/NAME/ (printed; load it with MRCNN_LIB=<path>); the product library is not touched

hipcc cross-compiles without a GPU. Objects are cached by mtime under maskrcnn_amd/csrc/build/.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libmaskrcnn_hip.so")
ARCH = "gfx950"

# MRCNN_ABLATIONS=1: also build the kernels that lost their A/B measurements (four-wave F(2x2) Winograd kernel, linear-tile
# heads variant, direct-kernel fused RPN level: include/maskrcnn_hip_ablations.h) and the F(4x4) kernel's timing variants
ABL = ["-DMRCNN_ABLATIONS"] if os.environ.get("MRCNN_ABLATIONS") else []
# per-source extra flags. nms/crop reproduce the reference's separately-rounded fp32 arithmetic:
# never let the compiler contract a*b+c into an FMA there.
SOURCES = {
    "common.hip": [],
    "nms.hip": ["-ffp-contract=off"],
    "nms_general.hip": ["-ffp-contract=off"],
    "crop.hip": ["-ffp-contract=off"] + (["-DMRCNN_CROP_STAMPS"] if os.environ.get("MRCNN_CROP_STAMPS") else []),
    "conv.hip": ABL,
    "conv_f16.hip": [],
    "conv_f16p.hip": [],
    "conv_wino.hip": ABL,
    "conv_wino4.hip": (["-DMRCNN_W4_ABLATIONS"] if (os.environ.get("MRCNN_W4_ABLATIONS") or os.environ.get("MRCNN_ABLATIONS")) else [])
                      + ([f"-DMRCNN_W4_WALK_SHIFT={int(os.environ['MRCNN_W4_WALK_SHIFT'])}"] if os.environ.get("MRCNN_W4_WALK_SHIFT") else []),
    "stem.hip": [],
    "bottleneck.hip": [],
    "bottleneck_op.hip": [],
    "mask_tail_f16.hip": [],
    "bottleneck_f16.hip": ([f"-DMRCNN_BF16_ABL={int(os.environ['MRCNN_BF16_ABL'])}"] if os.environ.get("MRCNN_BF16_ABL") else []),
    "misc.hip": ["-ffp-contract=off"],
    "select.hip": ["-ffp-contract=off"],
    "image.hip": ["-ffp-contract=off"],   # Pillow's coefficient arithmetic, operation by operation in fp64
}
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fno-fast-math",
          "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _newer(target: str, deps: list[str]) -> bool:
    return os.path.exists(target) and os.path.getmtime(target) >= max(os.path.getmtime(d) for d in deps)


def _plan(force, variant, defines, only):
    """(compile commands, flag stamps, object list, library path) of one build; compiles nothing."""
    obj_dir, lib_path = OBJ, LIB
    if variant:
        obj_dir = os.path.join(OBJ, "variants", variant)
        lib_path = os.path.join(obj_dir, "libmaskrcnn_hip.so")
    os.makedirs(obj_dir, exist_ok=True)
    cc = hipcc()
    headers = [os.path.join(ROOT, "include", "maskrcnn_hip.h"), os.path.abspath(__file__)]
    headers += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    jobs, objs, stamps = [], [], []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        mine = not (variant and only) or src in only
        o = os.path.join(obj_dir if mine else OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if not mine:
            continue
        extra = [*extra, *defines]
        # an object is reused only when it was built with the same flags (e.g. MRCNN_W4_ABLATIONS toggles variants)
        flags = " ".join([*COMMON, *extra])
        stamp = o + ".flags"
        same_flags = os.path.exists(stamp) and open(stamp).read() == flags
        if force or not same_flags or not _newer(o, [s] + headers):
            jobs.append([cc, *COMMON, *extra, "-c", s, "-o", o])
            stamps.append((stamp, flags))
    return jobs, stamps, objs, lib_path


def build_many(builds, force: bool = False, verbose: bool = False) -> list:
    """Several builds — [(variant or None, defines, only), ...] — with ALL their translation units in one pool (the two builds
    __graft_entry__.build() makes, product + schedule-fuzz variant, are 34 compiles: one after the other they leave the machine idle
    behind each build's longest unit). Returns the library paths."""
    plans = [_plan(force, v, tuple(d), tuple(o)) for v, d, o in builds]
    cc = hipcc()

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    jobs = [j for p in plans for j in p[0]]
    # longest units first (the Winograd and direct kernels: tens of seconds each), so that they do not end up last in the pool
    jobs.sort(key=lambda c: -os.path.getsize(c[-3]))
    with cf.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    out = []
    for pj, stamps, objs, lib_path in plans:
        for stamp, flags in stamps:   # only after every compile succeeded
            with open(stamp, "w") as fh:
                fh.write(flags)
        if pj or force or not _newer(lib_path, objs):
            run([cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib_path])
        out.append(lib_path)
    return out


def build(force: bool = False, verbose: bool = False, variant: str | None = None, defines: tuple = (),
          only: tuple = ()) -> str:
    """variant builds: `only` names the sources the defines apply to (compiled into the variant directory); every other
    object is the product's (built first if need be)."""
    if variant and only:
        build(force=False, verbose=verbose)
    return build_many([(variant, defines, only)], force=force, verbose=verbose)[0]


if __name__ == "__main__":
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    _only = tuple(sys.argv[sys.argv.index("--only") + 1].split(",")) if "--only" in sys.argv else ()
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv, variant=_variant,
                defines=tuple(a for a in sys.argv[1:] if a.startswith("-D")), only=_only))
