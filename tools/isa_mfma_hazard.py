#!/usr/bin/env python3
"""Build-time guard for the kernels that issue their matrix instructions through INLINE ASM (csrc/species_linear.hip,
species_linear_rows.hip): the compiler's hazard recogniser does not look inside inline asm, so nothing but the sources' own
`s_nop` pads keeps a store from reading an accumulator before the matrix pipeline has written it (csrc/Makefile: the
-simplifycfg-sink-common=false story; the failure was accumulators kept in scratch memory, load - MFMA - scratch_store).

What is checked on the gfx950 ISA of an object file (llvm-objdump --offloading + -d):
  over the control-flow graph of every kernel, from each v_mfma to every instruction that hands one of its destination
  registers to the memory pipelines (an operand of a global_ / buffer_ / flat_ / scratch_ / ds_ instruction -- a spill of an
  accumulator is a scratch_store): at least REQUIRED wait states on EVERY path (an instruction = 1, `s_nop N` = N + 1).
  CDNA3/4 ISA guide, "XDL write VGPR -> VMEM / LDS / FLAT read" for an 8-pass instruction (v_mfma_f32_16x16x4_f32): 11; the
  sources' mfma_drain() gives 16.  VALU reads of an accumulator are interlocked in hardware on gfx950
  (tools/ubench/mfma_overlap.hip) and not checked.  (A few prologue spills of scalars in species_linear_kernel are fine:
  what matters is the distance, not the instruction.)

    python3 tools/isa_mfma_hazard.py matten_amd/csrc/build/species_linear.o [...]      exit code 1 on a finding
Used by tests/test_host.py::test_inline_asm_mfma_objects_keep_their_hazard_distance."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("MATTEN_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
REQUIRED = 10
MEM_PREFIXES = ("global_", "buffer_", "flat_", "scratch_", "ds_")
_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def device_disassembly(obj: str) -> str:
    """gfx950 code object of a hipcc -c object -> llvm-objdump -d text (the extraction writes next to its input: a copy)"""
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=td)
        cos = [f for f in os.listdir(td) if "amdgcn" in f]
        if len(cos) != 1:
            raise RuntimeError(f"{obj}: expected one gfx950 bundle, found {cos}")
        return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", os.path.join(td, cos[0])], check=True,
                              capture_output=True, text=True).stdout


def _vregs(operands: str):
    out = set()
    for m in _VREG.finditer(operands):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse_functions(text: str):
    """{symbol: [(addr, mnemonic, operands)]} in address order"""
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        m = _INS.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def check_function(name: str, ins, required: int = REQUIRED):
    """-> list of findings (strings).  Forward data-flow over the CFG: state = {vgpr: wait states since an MFMA wrote it},
    absent = settled; merge = minimum per register."""
    findings = []
    index = {a: i for i, (a, _, _) in enumerate(ins)}
    succ = []
    for i, (addr, mn, ops) in enumerate(ins):
        nxt = [i + 1] if i + 1 < len(ins) else []
        if mn in ("s_endpgm", "s_setpc_b64"):
            nxt = []
        elif mn.startswith("s_branch") or mn.startswith("s_cbranch"):
            # simm16 in dwords relative to the next instruction (objdump prints it as an unsigned decimal)
            off = int(ops.split()[0])
            off = off - 65536 if off >= 32768 else off
            tgt = addr + 4 + 4 * off
            if tgt not in index:
                findings.append(f"{name}: branch at {addr:#x} leaves the function (target {tgt:#x}): cannot follow")
                tgt = None
            nxt = ([index[tgt]] if tgt is not None else []) + ([] if mn.startswith("s_branch") else nxt)
        succ.append(nxt)
    state_in = [None] * len(ins)
    state_in[0] = {}
    work = [0]
    reported = set()
    while work:
        i = work.pop()
        st = dict(state_in[i])
        addr, mn, ops = ins[i]
        if mn.startswith(MEM_PREFIXES):
            hot = {r: w for r, w in st.items() if r in _vregs(ops)}
            if hot and addr not in reported:
                reported.add(addr)
                r, w = min(hot.items(), key=lambda kv: kv[1])
                findings.append(f"{name}: {mn} {ops} at {addr:#x} takes v{r} {w} wait states after a v_mfma wrote it "
                                f"(needs >= {required})")
        ws = 1
        if mn == "s_nop":
            ws = int(ops.split()[0], 0) + 1
        st = {r: w + ws for r, w in st.items() if w + ws < required}
        if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
            for r in _vregs(ops.split(",")[0]):
                st[r] = 0
        for j in succ[i]:
            old = state_in[j]
            if old is None:
                state_in[j] = dict(st)
                work.append(j)
            else:
                new = dict(old)
                changed = False
                for r, w in st.items():
                    if r not in new or w < new[r]:
                        new[r] = w
                        changed = True
                if changed:
                    state_in[j] = new
                    work.append(j)
    return findings


def check_object(obj: str, required: int = REQUIRED):
    text = device_disassembly(obj)
    funcs = parse_functions(text)
    findings, n_mfma = [], 0
    for name, ins in funcs.items():
        if not ins:
            continue
        n_mfma += sum(1 for _, mn, _ in ins if mn.startswith("v_mfma"))
        findings += check_function(name, ins, required)
    return findings, n_mfma, len(funcs)


if __name__ == "__main__":
    bad = 0
    for obj in sys.argv[1:]:
        f, n, k = check_object(obj)
        print(f"{obj}: {k} functions, {n} matrix instructions, {len(f)} findings")
        for line in f:
            print("   ", line)
        bad += len(f)
    sys.exit(1 if bad else 0)
