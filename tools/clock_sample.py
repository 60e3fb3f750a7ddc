#!/usr/bin/env python3
"""Shader clock / power of the GPU while a command runs (is the step power-bound?).

usage: python3 tools/clock_sample.py [--period 0.1] -- python3 bench.py --steps 3000 ...
Starts the command as a child, samples the amdgpu sysfs files (pp_dpm_sclk's active level, hwmon power / frequency
inputs) every `period` seconds until it exits, prints the child's stdout last line and a histogram of the samples.
Read-only: needs no privileges.  Nothing here touches the HIP runtime (the child owns the GPU).
"""
import glob
import subprocess
import sys
import time
from collections import Counter


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def main():
    argv = sys.argv[1:]
    period = 0.1
    if argv and argv[0] == "--period":
        period = float(argv[1])
        argv = argv[2:]
    if argv and argv[0] == "--":
        argv = argv[1:]
    cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    freq_in = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
    pow_in = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) or sorted(
        glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
    print("sysfs:", cards, freq_in, pow_in, flush=True)
    # the child's stdout goes to a temporary file (a pipe read only after exit blocks a child that prints more than the pipe
    # buffer -- bench.py's JSON line can); its stderr is inherited (bench.py's early `headline:` line stays visible)
    import tempfile

    sink = tempfile.TemporaryFile(mode="w+")
    child = subprocess.Popen(argv, stdout=sink, stderr=None, text=True)
    sclk, freq, power = [], [], []
    t0 = time.time()
    while child.poll() is None:
        for c in cards:
            s = read(c)
            if s:
                for line in s.splitlines():
                    if line.rstrip().endswith("*"):
                        sclk.append((time.time() - t0, line.split(":")[1].strip(" *")))
        for f in freq_in:
            s = read(f)
            if s:
                freq.append((time.time() - t0, int(s) / 1e6))
        for p in pow_in:
            s = read(p)
            if s:
                power.append((time.time() - t0, int(s) / 1e6))
        time.sleep(period)
    sink.seek(0)
    out = sink.read().strip().splitlines()
    print("child rc", child.returncode, "| last line:", (out[-1][:400] if out else ""))
    # per-card view: the busy card is the one that drew the most power
    ncard = max(1, len(pow_in))
    if power and ncard > 1:
        per = [[x for i, (_, x) in enumerate(power) if i % ncard == c] for c in range(ncard)]
        busy = max(range(ncard), key=lambda c: sum(per[c]) / max(1, len(per[c])))
        print("busy card index", busy, "mean W per card:", [round(sum(v) / max(1, len(v))) for v in per])
        power = [(t, x) for i, (t, x) in enumerate(power) if i % ncard == busy]
        if len(freq_in) == ncard:
            freq = [(t, x) for i, (t, x) in enumerate(freq) if i % ncard == busy]
        if len(cards) == ncard:
            sclk = [(t, x) for i, (t, x) in enumerate(sclk) if i % ncard == busy]
    print("pp_dpm_sclk active level histogram:", Counter(v for _, v in sclk).most_common(12))
    if freq:
        v = sorted(x for _, x in freq)
        print(f"freq1_input MHz: n={len(v)} min {v[0]:.0f} p10 {v[len(v)//10]:.0f} median {v[len(v)//2]:.0f} p90 {v[9*len(v)//10]:.0f} max {v[-1]:.0f}")
        # timeline in 1-second buckets
        b = {}
        for t, x in freq:
            b.setdefault(int(t), []).append(x)
        print("per-second mean MHz:", [round(sum(x) / len(x)) for _, x in sorted(b.items())])
    if power:
        v = sorted(x for _, x in power)
        print(f"power W: n={len(v)} min {v[0]:.0f} median {v[len(v)//2]:.0f} p90 {v[9*len(v)//10]:.0f} max {v[-1]:.0f}")
        b = {}
        for t, x in power:
            b.setdefault(int(t), []).append(x)
        print("per-second mean W:", [round(sum(x) / len(x)) for _, x in sorted(b.items())])


if __name__ == "__main__":
    main()
