"""ISA gate of the gfx950 library (round 5).

The fast kernels here carry their vector-memory bookkeeping BY HAND: weight fragments and scale / shift tables are fetched by
inline-asm loads that hipcc knows nothing of, retired by hand-counted `s_waitcnt vmcnt(N)`, and bricks arrive by LDS-DMA.  That is
only correct while the compiler adds no vector-memory traffic of its own to those streams and does not break the hazards it
does not pad inside an `asm` statement (cdna_hip_programming.md section 5.7).  `check_asm_file` reads the `.s` that hipcc
leaves beside an object (`-save-temps=obj`) and fails the build when

  R   a hand-counted kernel (HAND_COUNTED) has scratch (`.private_segment_fixed_size`) or spilled VGPRs: spill code is scratch_load /
      scratch_store, i.e. vector-memory operations inside the counted streams;
  H1  an SGPR written by a VALU instruction - `v_readlane_b32` (the reload of an SGPR the allocator spilled to a VGPR lane),
      `v_readfirstlane_b32`, a `v_cmp` writing an SGPR pair - is read as scalar base / offset by a vector-memory instruction INSIDE an
      inline-asm statement fewer than 5 wait states later (the hazard recogniser pads compiler-emitted readers only).  This is
      what the abandoned two-body build of conv3_f32_wino3_kernel<2, true> did (round 4, "Memory access fault"): the weight base of
      a step was reloaded from its spill lane one instruction in front of `global_load_dwordx4 v[..], v, s[0:1]`, which then read
      the PREVIOUS contents of s[0:1] - a wild global address;
  H2  a compiler-inserted copy (v_mov / v_accvgpr_* / scratch access) touches the destination of an inline-asm load that may still be
      in flight (linear scan, fall-through paths only: what follows an unconditional branch is not reached from above).
"""
from __future__ import annotations

import re
import subprocess

# kernels whose vmcnt waits are counted by hand (name prefixes of the demangled kernel)
HAND_COUNTED = ("conv3_f32_wino3_kernel", "conv3_f32_wino2_kernel", "conv3_f32_wino_kernel", "conv3_f32_s2dma_kernel",
                "conv3_f16_dma_kernel", "conv3_f16_s2dma_kernel", "conv3_f16_c32_kernel")

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


def kernels(lines):
    """[(first line, end line, mangled name)] of every function in an assembly listing"""
    out = []
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z[\w$.]+:", l)]
    for i, nm in starts:
        end = len(lines)
        for j in range(i, len(lines)):
            if lines[j].strip().startswith("s_endpgm"):
                end = j + 1
                break
        out.append((i, end, nm))
    return out


def resources(text):
    """{mangled name: dict(scratch, vgpr_spill, sgpr_spill, vgpr)} from the amdhsa.kernels metadata of a listing"""
    res = {}
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+)"
                         r".*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", text, re.S):
        res[m.group(2)] = dict(scratch=int(m.group(3)), sgpr_spill=int(m.group(4)), vgpr=int(m.group(5)), vgpr_spill=int(m.group(6)))
    return res


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True, timeout=10).stdout.strip() or name
    except Exception:
        return name


def check_asm_text(text, label=""):
    """All findings (strings) of one assembly listing; empty list = the gate passes."""
    lines = text.splitlines()
    findings = []
    for name, r in resources(text).items():
        plain = demangle(name).replace("void ", "").replace("mi355::", "")
        if plain.startswith(HAND_COUNTED) and (r["scratch"] or r["vgpr_spill"]):
            findings.append(f"R {label}{plain[:80]}: scratch {r['scratch']} B, {r['vgpr_spill']} spilled VGPRs in a kernel with hand-counted vmcnt waits")
    for i, end, nm in kernels(lines):
        findings += audit(lines, i, end, label + nm[:70])
    return findings


def check_asm_file(path):
    with open(path) as fh:
        return check_asm_text(fh.read(), label="")
