#!/usr/bin/env python3
"""Command-line front end of the library's ISA gate (<pkg>/_isa_gate.py): hazards hipcc does not pad inside inline asm, compiler
copies of in-flight inline-asm load destinations, scratch in the hand-counted kernels.

    tools/asm_hazard_audit.py file.s [kernel-name-substring]      (a listing from hipcc -S --cuda-device-only or -save-temps)
Prints a line per kernel and every finding; exit code 1 if there is any.
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gate = importlib.import_module("automated-brain-mri-analysis-and-report-generation-with-retrieval-augmented-clinical-assistance_amd._isa_gate")


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    text = open(path).read()
    lines = text.splitlines()
    res = gate.resources(text)
    total = 0
    for i, end, nm in gate.kernels(lines):
        if want and want not in nm:
            continue
        f = gate.audit(lines, i, end, nm[:70])
        r = res.get(nm, {})
        nrl = sum(1 for l in lines[i:end] if "v_readlane_b32" in l)
        print(f"{gate.demangle(nm)[:100]}: {end - i} lines, scratch {r.get('scratch', '?')} B, spilled VGPRs {r.get('vgpr_spill', '?')}, v_readlane {nrl}, findings {len(f)}")
        for x in f[:40]:
            print("   ", x)
        total += len(f)
    rfind = [x for x in gate.check_asm_text(text) if x.startswith("R ")]
    for x in rfind:
        print(x)
    return 1 if total + len(rfind) else 0


if __name__ == "__main__":
    sys.exit(main())
