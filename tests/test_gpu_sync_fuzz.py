"""Schedule fuzzing (round 6): the kernel-level parity tests on a build in which every wave sleeps a pseudo-random 0 .. 24 576
cycles behind every workgroup barrier (maskrcnn_amd/csrc/common.hpp, MRCNN_SYNC_FUZZ; built by __graft_entry__.build() as
maskrcnn_amd/csrc/build/variants/sync_fuzz/). A kernel whose LDS hand-offs are ordered by barriers and counted waits computes the
same bits at any skew between its waves; one that relies on "the other waves cannot be that far ahead" — round 5's
conv3x3_wino4_f32: ten MFMA slots where a barrier belonged, wrong once per ~5 000 launches — is wrong on every launch here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZ_LIB = os.path.join(ROOT, "maskrcnn_amd", "csrc", "build", "variants", "sync_fuzz", "libmaskrcnn_hip.so")
pytestmark = pytest.mark.gpu


def _env():
    assert os.path.exists(FUZZ_LIB), f"{FUZZ_LIB} is missing: run __graft_entry__.build() (or maskrcnn_amd/build.py --variant sync_fuzz -DMRCNN_SYNC_FUZZ -DMRCNN_W4_ABLATIONS)"
    env = dict(os.environ, MRCNN_LIB=FUZZ_LIB, MRCNN_SYNC_FUZZ_CHILD="1")
    env.pop("MRCNN_W4_DEBUG", None)
    return env


@pytest.mark.skipif(os.environ.get("MRCNN_SYNC_FUZZ_CHILD") == "1", reason="this IS the fuzzed child run")
def test_kernel_parity_tests_pass_under_schedule_fuzzing():
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_conv.py", "tests/test_gpu_ops.py", "tests/test_gpu_image.py",
                        "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=_env(), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]


@pytest.mark.skipif(os.environ.get("MRCNN_SYNC_FUZZ_CHILD") == "1", reason="this IS the fuzzed child run")
def test_full_size_and_pipeline_tests_pass_under_schedule_fuzzing():
    """The same at BASELINE's full sizes: the whole trunk / RPN / heads / detections against the oracle, batch-slice bit identity, the
    fp16 mode's agreement tests and the pipeline tests (80 tests, about two minutes) on the fuzzed build."""
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_fullsize.py", "tests/test_gpu_pipeline.py",
                        "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=_env(), capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]


@pytest.mark.skipif(os.environ.get("MRCNN_SYNC_FUZZ_CHILD") == "1", reason="this IS the fuzzed child run")
def test_fuzzing_finds_round5_kernel():
    """The detector detects: the same build with the barrier behind the prologue's operand reads left out (MRCNN_W4_DEBUG=8192,
    an ablation variant of the plain F(4x4) kernel = round 5's kernel) is wrong on every launch, the shipped kernel on none."""
    code = r'''
import os, sys, json, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import w4_forensics as f
from maskrcnn_amd import ops
dev = torch.device("cuda:0")
x, w, shift = f.operands()
xk, u4, shift = ops.nhwc_to_kblocked(x.to(dev)), ops.winograd4_weights(w.to(dev)), shift.to(dev)
ref = ops.conv3x3_winograd(xk, ops.winograd_weights(w.to(dev)), None, shift, False)
out = {}
for name, dbg in (("shipped", 0), ("round5", 8192)):
    os.environ["MRCNN_W4_DEBUG"] = str(dbg)
    out[name] = [float((ops.conv3x3_winograd4(xk, u4, None, shift, False, None, "nhwc") - ref).abs().max()) for _ in range(4)]
print("RESULT " + json.dumps(out))
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert max(res["shipped"]) <= 1e-4, res           # F(4x4) vs F(2x2): rounding only
    assert min(res["round5"]) > 0.1, res              # wave 0 multiplies k tile 0 with k tile 2's operands: O(1) errors
