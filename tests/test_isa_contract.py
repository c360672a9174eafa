"""The inline-asm contracts of the hand-scheduled kernels, re-derived from the generated code (hipcc -S; no GPU): tools/isa_audit.py.

hipcc neither counts nor pads what is inside an asm string. conv3x3_wino4_f32 has asm MFMAs the hazard recogniser cannot see,
asm LDS reads behind hand-counted lgkmcnt(N) and staging waits vmcnt(3) counted by hand; conv_f16p has asm ds_read_b128 and counted
vmcnt(N). Round 6 met both failure modes this guards against: a build whose register allocation copied an accumulator 6 wait
states behind an asm MFMA (wrong sums; check C) and compiler copies of a[0:15] at the head of the k loop (check D)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_audit  # noqa: E402


@pytest.mark.parametrize("source", sorted(isa_audit.SOURCES))
def test_asm_contracts_hold(source):
    rep = isa_audit.audit_source(source)
    assert rep, f"{source}: no kernel with an asm contract found — the audit is not looking at anything"
    for name, r in rep.items():
        assert not r["violations"], f"{source}: {name}: {r['violations'][:5]} (meta {r['meta']})"
    if source == "conv_wino4.hip":
        assert len(rep) == 5   # plain (act / no act), heads (act / no act), conv2 + conv3
        for name, r in rep.items():
            # 4 waves x 2 k tiles of the loop x 36 MFMAs; every LDS-DMA of a k tile behind a hand-counted wait
            assert r["asm_mfma"] == 288 and r["asm_lds_reads"] >= 4 * (25 + 18) * 3 and r["asm_vmcnt_waits"] >= 12, (name, r)


def test_broken_contracts_are_reported():
    """Edits of the generated code that break ONE contract each (a staging wait one too loose, a spill of an asm load's
    destination before its wait, an accumulator read behind an asm MFMA, a counted LDS wait one too loose, a compiler copy of
    a live accumulator inside the k loop): the audit must report every one — a test that cannot fail proves nothing."""
    text = isa_audit.compile_asm("conv_wino4.hip")
    for what, (check, mutated) in isa_audit.mutations(text).items():
        assert mutated != text, what
        rep = isa_audit.audit_source("conv_wino4.hip", mutated)
        fired = {v[0] for r in rep.values() for v in r["violations"]}
        assert check in fired, f"mutation '{what}' not reported (checks that fired: {sorted(fired)})"
