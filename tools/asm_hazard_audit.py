#!/usr/bin/env python3
"""Audit of a hipcc -save-temps .s file for the hazards hipcc does not pad INSIDE inline asm (cdna_hip_programming.md section 5.7):

  H1  an SGPR written by a VALU instruction (v_readlane_b32 = the reload of an SGPR the allocator spilled to a VGPR lane,
      v_readfirstlane_b32, v_cmp writing an SGPR pair) and read as the scalar base / offset of a vector-memory instruction, or moved
      into M0 for an LDS-DMA, within 5 wait states, where the reader sits between ;;#ASMSTART and ;;#ASMEND (the hazard recogniser
      pads compiler-emitted readers only);
  H2  a compiler-inserted v_mov / v_accvgpr_* whose source or destination is the destination of an inline-asm global_load that has
      not been waited for (the asm loads are invisible to hipcc's vmcnt bookkeeping: a copy made in flight copies the OLD value,
      and the load then lands in a register that may have been given to something else).

Usage: asm_hazard_audit.py file.s [kernel-name-substring]
Prints every finding with its line number; exit code 1 if there is any.
"""
import re
import sys

VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)")
SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def vregs(text):
    out = set()
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def audit(lines, start, end, name):
    findings = []
    in_asm = False
    recent = []          # (wait states since the write, sgpr set, line number, text) of VALU writes of SGPRs
    queue = []           # outstanding vector-memory operations, oldest first: (line, VGPRs an inline-asm load will write)
    for ln in range(start, end):
        raw = lines[ln]
        t = raw.split("//")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            if t.endswith(":"):
                pass  # (labels do not add wait states)
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith(";"):
            continue
        op = t.split()[0]
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            queue = []   # (linear scan: what follows an unconditional branch is not reached from here; H2 is reported for fall-through paths only)
        states = 1
        m = re.match(r"s_nop\s+(\d+)", t)
        if m:
            states = int(m.group(1)) + 1
        # ---- H1 readers
        reads_s = set()
        if VMEM.match(t):
            reads_s = sregs(t)
        elif re.match(r"s_mov_b32\s+m0", t):
            reads_s = set()   # an SALU read of a VALU-written SGPR needs no wait state on gfx950 (cdna guide 5.7 item 2)
        if reads_s and in_asm:
            for age, regs, wl, wt in recent:
                hit = regs & reads_s
                if hit and age < 5:
                    findings.append(f"H1 {name}: line {ln + 1} `{t}` (inline asm) reads s{sorted(hit)} written by VALU at line {wl + 1} `{wt}` only {age} wait states earlier (needs 5)")
        # ---- H2: compiler copies of in-flight asm load destinations.  `queue` = the wave's outstanding vector-memory operations in
        # issue order (loads return in order, cdna guide: stores / atomics / LDS-DMA count together with them); an entry carries the
        # VGPRs an INLINE-ASM load will write (compiler loads are waited for by the compiler itself)
        if VMEM.match(t):
            dst = set()
            if in_asm and re.match(r"(global|buffer)_load_\w+\s+v", t) and " lds" not in t and "_lds_" not in t:
                dst = vregs(t.split(",")[0])
            queue.append((ln, dst))
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            queue[:] = queue[len(queue) - n:] if n else []
        if not in_asm and re.match(r"(v_mov_b32|v_mov_b64|v_accvgpr_write|v_accvgpr_read|v_pk_mov_b32|scratch_store|scratch_load)", op):
            touched = vregs(t)
            for qln, dst in queue:
                hit = touched & dst
                if hit:
                    findings.append(f"H2 {name}: line {ln + 1} `{t}` touches v{sorted(hit)} while the asm load of line {qln + 1} may be in flight")
                    break
        # ---- age the VALU->SGPR writes
        recent = [(age + states, regs, wl, wt) for age, regs, wl, wt in recent if age + states < 8]
        if re.match(r"(v_readlane_b32|v_readfirstlane_b32)\s+s", t):
            recent.append((0, sregs(t.split(",")[0]), ln, t))
        elif re.match(r"v_cmp\w*\s+s\[", t):
            recent.append((0, sregs(t.split(",")[0]), ln, t))
    return findings


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    lines = open(path).read().splitlines()
    starts = [(i, l[:-1]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s*(;.*)?$", l.split("//")[0].strip() + "") and not l.startswith(".")]
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z[\w$.]+:", l)]
    total = 0
    for k, (i, nm) in enumerate(starts):
        if want and want not in nm:
            continue
        end = len(lines)
        for j in range(i, len(lines)):
            if lines[j].strip().startswith("s_endpgm"):
                end = j + 1
                break
        f = audit(lines, i, end, nm[:70])
        nrl = sum(1 for l in lines[i:end] if "v_readlane_b32" in l)
        nwl = sum(1 for l in lines[i:end] if "v_writelane_b32" in l)
        print(f"{nm[:90]}: {end - i} lines, v_readlane {nrl}, v_writelane {nwl}, findings {len(f)}")
        for x in f[:40]:
            print("   ", x)
        total += len(f)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
