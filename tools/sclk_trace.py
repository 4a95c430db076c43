#!/usr/bin/env python3
"""Shader clock / board power trace while a command runs on the GPU (GPU box): samples the amdgpu sysfs files of every
card every 50 ms from THIS process (which never touches HIP) while the command runs as a child, and calls `rocm-smi
--showclocks --showpower --showuse` a few times on the way.  Output: a text table (time, per card: sclk of hwmon
freq1_input or the starred pp_dpm_sclk level, power1_average / power1_input, gpu_busy_percent) and a summary of the busiest
card over the samples where it was busy.

    python tools/sclk_trace.py gpurun_out/r03/sclk_config3_f16.txt -- python bench.py --config 3 --dtype f16 --steps 10 --warmup 1 --no-cpu-baseline

(MI355X_MICROARCH.md, DVFS give-back: the clock a dense MFMA kernel really runs at reads up to ~10 % below what sysfs
reports; tools/kernel_clocks.py derives it per kernel from GRBM_GUI_ACTIVE.  This trace is the coarse, tool-independent view.)
"""
import glob
import os
import re
import subprocess
import sys
import time


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def cards():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if read(dev + "/vendor") != "0x1002":
            continue
        hw = sorted(glob.glob(dev + "/hwmon/hwmon*"))
        out.append((os.path.basename(os.path.dirname(dev)), dev, hw[0] if hw else None))
    return out


def sample(dev, hw):
    sclk = None
    if hw:
        f = read(hw + "/freq1_input")
        if f and f.isdigit():
            sclk = int(f) / 1e6
    if sclk is None:
        levels = read(dev + "/pp_dpm_sclk") or ""
        m = re.search(r"(\d+)Mhz \*", levels)
        sclk = float(m.group(1)) if m else float("nan")
    power = None
    if hw:
        for n in ("power1_average", "power1_input"):
            p = read(hw + "/" + n)
            if p and p.isdigit():
                power = int(p) / 1e6
                break
    busy = read(dev + "/gpu_busy_percent")
    return sclk, power if power is not None else float("nan"), int(busy) if busy and busy.isdigit() else -1


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    cs = cards()
    rows = []
    smi = []
    t0 = time.time()
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    next_smi = 8.0
    while child.poll() is None:
        t = time.time() - t0
        rows.append((t, [sample(dev, hw) for _, dev, hw in cs]))
        if t >= next_smi and len(smi) < 4:
            next_smi += 4.0
            try:
                r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showuse"], capture_output=True, text=True, timeout=20)
                smi.append((t, r.stdout.strip() or r.stderr.strip()))
            except Exception as e:  # rocm-smi missing or refused: the sysfs samples stand on their own
                smi.append((t, f"rocm-smi failed: {e}"))
        time.sleep(0.05)
    child_out = child.stdout.read()
    with open(out_path, "w") as fh:
        fh.write("# " + " ".join(cmd) + f"\n# exit code {child.returncode}; {len(rows)} samples at ~50 ms; cards: {[c[0] for c in cs]}\n")
        busiest, best = None, -1.0
        for i, (name, _, _) in enumerate(cs):
            b = [r[1][i][2] for r in rows if r[1][i][2] >= 0]
            if b and sum(b) / len(b) > best:
                best, busiest = sum(b) / len(b), i
        if busiest is not None:
            act = [r[1][busiest] for r in rows if r[1][busiest][2] >= 50]
            idle = [r[1][busiest] for r in rows if 0 <= r[1][busiest][2] < 5]
            def stats(v):
                v = sorted(x for x in v if x == x)
                return f"min {v[0]:.0f} median {v[len(v) // 2]:.0f} max {v[-1]:.0f}" if v else "n/a"
            fh.write(f"# busiest card: {cs[busiest][0]} ({len(act)} samples at >= 50 % busy, {len(idle)} idle)\n")
            fh.write(f"#   busy: sclk MHz {stats([a[0] for a in act])}; power W {stats([a[1] for a in act])}\n")
            fh.write(f"#   idle: sclk MHz {stats([a[0] for a in idle])}; power W {stats([a[1] for a in idle])}\n")
        fh.write("# t[s]  " + "  ".join(f"{c[0]}: sclk[MHz] power[W] busy[%]" for c in cs) + "\n")
        for t, vals in rows:
            fh.write(f"{t:7.2f}  " + "  ".join(f"{v[0]:8.0f} {v[1]:7.1f} {v[2]:4d}" for v in vals) + "\n")
        for t, text in smi:
            fh.write(f"\n# ---- rocm-smi --showclocks --showpower --showuse at t = {t:.1f} s\n{text}\n")
        fh.write("\n# ---- output of the command (tail)\n" + "\n".join(child_out.splitlines()[-3:]) + "\n")
    sys.exit(child.returncode)


if __name__ == "__main__":
    main()
