#!/usr/bin/env python3
"""Static audit of the hand-scheduled kernels' ISA (hipcc -S, no GPU): the contracts inline asm takes over from the compiler.

hipcc neither counts nor pads what is inside an asm string (cdna_hip_programming.md 'What hipcc does not do'): an asm load's
destination counts as written at ;;#ASMEND, an asm MFMA is invisible to the hazard recogniser, an asm s_waitcnt vmcnt(N) is a
number somebody counted by hand. This tool re-derives those counts from the generated code:

  A  asm LDS reads   no instruction reads or writes the destination of an asm ds_read between the read and the s_waitcnt lgkmcnt(N)
                     that retires it (LDS operations return in order; an outstanding scalar load makes only lgkmcnt(0) count);
  B  asm vmcnt       behind every asm s_waitcnt vmcnt(N) no LDS-DMA (buffer_load ... lds) of the wave is still outstanding —
                     vector-memory operations retire in issue order, so the N youngest must all be register loads / stores
                     (compiler spill code can only make such a wait stricter: it adds operations, it cannot make a DMA younger);
  C  asm MFMA        >= 18 wait states (16-pass v_mfma_f32_32x32x2_f32) between an asm MFMA and any other instruction that reads
                     or writes its D registers, except an MFMA that takes D whole as its C (the accumulate chain);
  D  accumulators    inside the innermost loop that holds asm MFMAs (the k loop) the compiler touches no accumulator register
                     (no v_accvgpr_*, no copy) and places no spill code;
  E  M0              at least one instruction between a write of M0 and the LDS-DMA that reads it;
  F  metadata        vgpr_spill_count == 0 and no scratch in every kernel under contract.

Usage: tools/isa_audit.py [--mutations]    (tests/test_isa_contract.py runs both)
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "maskrcnn_amd", "csrc")
OUT = os.path.join(CSRC, "build", "isa")
FLAGS = ["-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC, "-S", "--cuda-device-only"]


def compile_asm(source: str, extra=()) -> str:
    """hipcc -S of maskrcnn_amd/csrc/<source>, cached by mtime of the source and the headers."""
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(CSRC, source)
    dst = os.path.join(OUT, source.replace(".hip", "") + ("_" + "_".join(e.lstrip("-D") for e in extra) if extra else "") + ".s")
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "maskrcnn_hip.h")]
    if not os.path.exists(dst) or os.path.getmtime(dst) < max(os.path.getmtime(d) for d in deps):
        hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
        r = subprocess.run([hipcc, *FLAGS, *extra, src, "-o", dst], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
    return open(dst).read()


REG = re.compile(r"\b([vas])(\d+)\b|\b([vas])\[(\d+):(\d+)\]|\b(m0|vcc_lo|vcc_hi|vcc|exec_lo|exec_hi|exec|scc)\b")


def regs_of(tok: str) -> set:
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        elif m.group(3):
            out.update(m.group(3) + str(i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
        else:
            out.add(m.group(6))
    return out


class Inst:
    __slots__ = ("line", "text", "op", "ops", "asm", "dst", "src", "label")

    def __init__(self, line, text, asm):
        self.line, self.text, self.asm, self.label = line, text, asm, None
        body = text.split(";")[0].strip()
        parts = body.split(None, 1)
        self.op = parts[0]
        self.ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        op = self.op
        all_regs = [regs_of(o) for o in self.ops]
        stores = op.startswith(("ds_write", "ds_store", "buffer_store", "global_store", "scratch_store", "flat_store")) or \
            (op.startswith("buffer_load") and body.endswith(" lds")) or op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch", "s_cmp", "s_bitcmp", "s_sleep", "s_setprio", "s_endpgm"))
        if stores or not all_regs:
            self.dst, self.src = set(), set().union(*all_regs) if all_regs else set()
        else:
            self.dst, self.src = set(all_regs[0]), set().union(*all_regs[1:]) if len(all_regs) > 1 else set()
            if op.startswith(("v_mfma", "v_smfma")):
                pass
            elif len(all_regs) > 1 and self.ops[1] in ("vcc", "vcc_lo") or (len(self.ops) > 1 and re.fullmatch(r"s\[\d+:\d+\]", self.ops[1]) and op.startswith(("v_add_co", "v_sub_co", "v_addc", "v_subb", "v_div_scale", "v_mad_u64", "v_mad_i64"))):
                self.dst |= all_regs[1]

    @property
    def is_dma(self):
        return self.op.startswith("buffer_load") and self.text.split(";")[0].rstrip().endswith(" lds") or self.op.startswith("global_load_lds")

    @property
    def is_vmem(self):
        return self.op.startswith(("buffer_", "global_", "scratch_", "flat_")) and not self.op.startswith(("buffer_wbl2", "buffer_inv", "global_wb", "global_inv"))

    @property
    def is_lds(self):
        return self.op.startswith("ds_")

    @property
    def is_smem(self):
        return self.op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime"))


def kernels(asm_text: str) -> dict:
    """{mangled kernel name: {"insts": [Inst...], "meta": {...}}} for every kernel of the translation unit."""
    lines = asm_text.split("\n")
    meta, cur = {}, None
    for l in lines:
        m = re.match(r"\s+\.name:\s+(\S+)", l)
        if m:
            cur = m.group(1)
            meta[cur] = {}
        m = re.match(r"\s+\.(vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|vgpr_count|agpr_count|sgpr_count):\s+(\d+)", l)
        if m and cur:
            meta[cur][m.group(1)] = int(m.group(2))
    out = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s", lines[i])
        if m and m.group(1) in meta:
            name, insts, in_asm, pending_label = m.group(1), [], False, None
            i += 1
            while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
                l = lines[i]
                s = l.strip()
                if s.startswith(";;#ASMSTART"):
                    in_asm = True
                elif s.startswith(";;#ASMEND"):
                    in_asm = False
                elif re.match(r"^\.LBB\d+_\d+:", l):
                    pending_label = l.split(":")[0]
                elif s and not s.startswith((";", ".", "//")) and not re.match(r"^\d+:$", s):
                    ins = Inst(i + 1, s, in_asm)
                    ins.label, pending_label = pending_label, None
                    insts.append(ins)
                i += 1
            out[name] = {"insts": insts, "meta": meta[name]}
        i += 1
    return out


def lgkm_n(ins):
    m = re.search(r"lgkmcnt\((\d+)\)", ins.text)
    return int(m.group(1)) if m else None


def vm_n(ins):
    m = re.search(r"vmcnt\((\d+)\)", ins.text)
    return int(m.group(1)) if m else None


def loops(insts):
    """[(first index, last index)] of the backward branches (label .. branch)."""
    at = {ins.label: k for k, ins in enumerate(insts) if ins.label}
    out = []
    for k, ins in enumerate(insts):
        if ins.op.startswith(("s_cbranch", "s_branch")) and ins.ops and ins.ops[0] in at and at[ins.ops[0]] <= k:
            out.append((at[ins.ops[0]], k))
    return out


MFMA_PASSES = {"v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_32x32x2f32": 16}


def audit(insts, require_no_dma_behind_asm_vmcnt=True) -> list:
    """The violations of checks A-E in one kernel's instruction list (one linear walk, then every loop body once more, seeded
    with the state at its backward branch)."""
    viol = []

    def walk(lo, hi, lgkm, vm, mfma, m0_age):
        at_label = {}   # MFMA state carried along forward branches to their labels
        for k in range(lo, hi + 1):
            ins = insts[k]
            if ins.label in at_label:
                for regs, left in at_label.pop(ins.label):
                    mfma.append([set(regs), left])
            touched = ins.dst | ins.src
            # A
            if ins.op == "s_waitcnt" or ins.op.startswith("s_waitcnt"):
                n = lgkm_n(ins)
                if n is not None:
                    if any(e["smem"] for e in lgkm):
                        if n == 0:
                            lgkm.clear()
                    else:
                        del lgkm[:max(0, len(lgkm) - n)]
                n = vm_n(ins)
                if n is not None:
                    del vm[:max(0, len(vm) - n)]
                    if ins.asm and require_no_dma_behind_asm_vmcnt and any(e["kind"] == "dma" for e in vm):
                        viol.append(("B", ins.line, f"asm '{ins.text}' leaves an LDS-DMA outstanding (issued at line {[e['line'] for e in vm if e['kind'] == 'dma'][0]})"))
            else:
                pend = set().union(*[e["dst"] for e in lgkm if e["asm"]]) if lgkm else set()
                hit = touched & pend
                if hit and not ins.op.startswith("s_nop"):
                    viol.append(("A", ins.line, f"'{ins.text}' touches {sorted(hit)[:4]} while an asm LDS read of it is outstanding"))
            if ins.is_lds:
                lgkm.append({"asm": ins.asm, "dst": set(ins.dst) if ins.op.startswith(("ds_read", "ds_load")) else set(), "smem": False})
            elif ins.is_smem:
                lgkm.append({"asm": ins.asm, "dst": set(), "smem": True})
            # B
            if ins.is_vmem:
                kind = "dma" if ins.is_dma else "scratch" if ins.op.startswith("scratch_") else "other"
                vm.append({"kind": kind, "line": ins.line})
            # C / D
            if not (ins.op.startswith("s_waitcnt")):
                for d in list(mfma):
                    regs, left = d
                    is_chain = ins.op in MFMA_PASSES and ins.asm and regs_of(ins.ops[-1]) == regs and ins.dst == regs
                    if touched & regs and not is_chain and left > 0:
                        viol.append(("C", ins.line, f"'{ins.text}' touches {sorted(touched & regs)[:3]} {18 - left} wait states behind an asm MFMA (18 needed)"))
                        mfma.remove(d)
                step = 1
                if ins.op == "s_nop":
                    step = int(ins.ops[0], 0) + 1
                for d in mfma:
                    d[1] -= step
                mfma[:] = [d for d in mfma if d[1] > 0]
                if ins.asm and ins.op in MFMA_PASSES:
                    mfma[:] = [d for d in mfma if d[0] != ins.dst]
                    mfma.append([set(ins.dst), MFMA_PASSES[ins.op] + 2])
            # E
            if ins.is_dma and m0_age[0] == 0:
                viol.append(("E", ins.line, f"'{ins.text}' directly behind a write of M0"))
            m0_age[0] = 0 if "m0" in ins.dst else m0_age[0] + (int(ins.ops[0], 0) + 1 if ins.op == "s_nop" else 1)
            if ins.op.startswith(("s_cbranch", "s_branch")) and ins.ops and mfma:
                at_label.setdefault(ins.ops[0], []).extend([set(r), l] for r, l in mfma)   # (a backward target is never popped)
            if ins.op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                # what follows in the text is reached only through its label: the MFMA / M0 state of THIS path does not apply
                # (the zero-trip path around the k loop sits behind the loop's closing branch)
                del mfma[:]
                m0_age[0] = 9
        return lgkm, vm, mfma, m0_age

    state = walk(0, len(insts) - 1, [], [], [], [9])
    for lo, hi in loops(insts):
        if any(insts[k].asm for k in range(lo, hi + 1)):
            # state at the branch: re-derive by walking up to the branch, then once more through the body
            st = walk(0, hi, [], [], [], [9])
            before = len(viol)
            walk(lo, hi, *st)
            del viol[before:before]  # (kept: loop-carried violations)
    # D: compiler accesses to accumulators (and spill code) inside the innermost loops with asm MFMAs
    all_loops = loops(insts)
    for lo, hi in all_loops:
        if any((l2, h2) != (lo, hi) and lo <= l2 and h2 <= hi for l2, h2 in all_loops):
            continue
        acc = set()
        for k in range(lo, hi + 1):
            if insts[k].asm and insts[k].op in MFMA_PASSES:
                acc |= insts[k].dst
        if acc:
            for k in range(lo, hi + 1):
                ins = insts[k]
                if not ins.asm and (ins.dst | ins.src) & acc:
                    viol.append(("D", ins.line, f"compiler instruction '{ins.text}' touches accumulator registers inside the MFMA loop"))
                if ins.op.startswith("scratch_"):
                    viol.append(("D", ins.line, f"spill code '{ins.text}' inside the MFMA loop"))
    # de-duplicate (the loop re-walk reports steady-state violations twice)
    seen, uniq = set(), []
    for v in viol:
        if (v[0], v[1]) not in seen:
            seen.add((v[0], v[1]))
            uniq.append(v)
    return uniq


def has_asm_contract(insts) -> bool:
    return any(i.asm and (i.is_lds or i.op in MFMA_PASSES or (i.op.startswith("s_waitcnt") and vm_n(i) is not None)) for i in insts)


# ---- the kernels under contract: source, per-kernel expectations
SOURCES = {
    # conv3x3_wino4_f32: asm MFMAs with pinned accumulator classes, asm ds_read_b64 with counted lgkmcnt, staging waits vmcnt(3)
    "conv_wino4.hip": {"dma_free_behind_asm_vmcnt": True,
                       # no instantiation spills (round 6: the exchange-area writes as ds_write2st64_b32 with immediate component /
                       # position offsets freed the ~20 address registers that had cost HEADS 2 and CONV3 15 spilled values)
                       "spills": lambda name, meta: meta["vgpr_spill_count"] == 0 and meta["private_segment_fixed_size"] == 0},
    # conv_f16p: asm ds_read_b128 behind lgkmcnt(0), counted vmcnt(N) that deliberately leaves later tiles' DMAs in flight
    "conv_f16p.hip": {"dma_free_behind_asm_vmcnt": False,
                      "spills": lambda name, meta: meta["vgpr_spill_count"] == 0 and meta["private_segment_fixed_size"] == 0},
}


def audit_source(source: str, text: str | None = None) -> dict:
    spec = SOURCES[source]
    ks = kernels(text if text is not None else compile_asm(source))
    report = {}
    for name, k in ks.items():
        if not has_asm_contract(k["insts"]):
            continue
        v = audit(k["insts"], spec["dma_free_behind_asm_vmcnt"])
        if not spec["spills"](name, k["meta"]):
            v.append(("F", 0, f"spills: {k['meta']}"))
        report[name] = {"violations": v, "meta": k["meta"], "instructions": len(k["insts"]),
                        "asm_mfma": sum(1 for i in k["insts"] if i.asm and i.op in MFMA_PASSES),
                        "asm_lds_reads": sum(1 for i in k["insts"] if i.asm and i.is_lds),
                        "lds_dma": sum(1 for i in k["insts"] if i.is_dma),
                        "asm_vmcnt_waits": sum(1 for i in k["insts"] if i.asm and i.op.startswith("s_waitcnt") and vm_n(i) is not None)}
    return report


def mutations(text: str) -> dict:
    """Edits of conv_wino4's generated code that break one contract each; every one must be reported."""
    out = {}
    # the hand-counted staging wait one too loose: a DMA stays outstanding
    out["vmcnt(3) -> vmcnt(4) in the staging waits"] = ("B", text.replace("s_waitcnt vmcnt(3) lgkmcnt(0)", "s_waitcnt vmcnt(4) lgkmcnt(0)"))
    # a forced spill of an asm-read destination between the read and its wait
    m = re.search(r"(;;#ASMSTART\n\tds_read_b64 (v\[(\d+):\d+\]), [^\n]*\n\t;;#ASMEND\n)", text)
    out["spill of an asm LDS read's destination before its wait"] = ("A", text.replace(m.group(1), m.group(1) + f"\tscratch_store_dword off, v{m.group(3)}, off offset:60\n", 1))
    # an LDS store of an accumulator two instructions behind the MFMA that writes it (the k loop's last)
    i = text.index("\ts_nop 15\n\ts_nop 3\n")
    j = text.rindex(";;#ASMEND\n", 0, text.rindex("v_mfma_f32_32x32x2_f32", 0, i))
    m = re.search(r"v_mfma_f32_32x32x2_f32 (v\[(\d+):\d+\])", text[text.rindex("v_mfma_f32_32x32x2_f32", 0, i):i])
    k = text.index(";;#ASMEND\n", text.rindex("v_mfma_f32_32x32x2_f32", 0, i)) + len(";;#ASMEND\n")
    out["an LDS store of an accumulator directly behind the k loop's last MFMA"] = ("C", text[:k] + f"\tds_write_b32 v0, v{m.group(2)}\n" + text[k:])
    # a counted LDS wait one too loose
    out["lgkmcnt(2) -> lgkmcnt(3) in front of the input transform"] = ("A", text.replace("s_waitcnt lgkmcnt(2)", "s_waitcnt lgkmcnt(3)"))
    # a compiler copy of an accumulator inside the k loop
    m = re.search(r"(;;#ASMSTART\n\tv_mfma_f32_32x32x2_f32 a\[0:15\][^\n]*\n\t;;#ASMEND\n)", text)
    out["a compiler v_accvgpr_read of a live accumulator in the k loop"] = ("D", text.replace(m.group(1), m.group(1) + "\tv_accvgpr_read_b32 v0, a3\n"))
    return out


if __name__ == "__main__":
    bad = 0
    for source in SOURCES:
        rep = audit_source(source)
        for name, r in rep.items():
            print(f"{source}: {name[:70]}  {r['instructions']} instructions, {r['asm_mfma']} asm MFMAs, {r['asm_lds_reads']} asm LDS reads, "
                  f"{r['lds_dma']} LDS-DMAs, {r['asm_vmcnt_waits']} asm vmcnt waits, meta {r['meta']}: {len(r['violations'])} violations")
            for v in r["violations"][:12]:
                print("    ", v)
            bad += len(r["violations"])
    if "--mutations" in sys.argv:
        text = compile_asm("conv_wino4.hip")
        for what, (check, mutated) in mutations(text).items():
            assert mutated != text, what
            rep = audit_source("conv_wino4.hip", mutated)
            found = sorted({v[0] for r in rep.values() for v in r["violations"]})
            print(f"mutation '{what}': checks that fire {found} (expected {check})")
            bad += check not in found
    sys.exit(1 if bad else 0)
