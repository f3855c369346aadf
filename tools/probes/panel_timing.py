#!/usr/bin/env python3
"""Reads the s_memtime records of conv2d_panel_n64_kernel<., TIMING> (OSP_PANEL_TIMING=<file>) and prints, per recorded wave, the clocks
between the loop's wait points: median over the recorded tiles.  s_memtime ticks are shader cycles (MI355X_MICROARCH.md).
    OSP_PANEL_TIMING=/tmp/pt.bin ONLY=1 python tools/probes/panel_probe.py; python tools/probes/panel_timing.py /tmp/pt.bin"""
import struct, sys
import numpy as np
E = 72
data = open(sys.argv[1], "rb").read()
off, rec = 0, 0
while off < len(data):
    gy, nph, sw, taps = struct.unpack_from("4q", data, off); off += 32
    a = np.frombuffer(data, dtype=np.uint64, count=64 * 2 * E, offset=off).reshape(64, 2, E).astype(np.int64); off += 64 * 2 * E * 8
    rec += 1
    want = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3]
    if rec not in want:
        continue
    n = a[:, :, E - 1]
    ok = n[:, 0] > 0
    if not ok.any():
        continue
    t = a[ok]                                   # (tiles, 2 waves, events)
    nev = int(n[ok][0, 0])
    t0 = t[:, :, 0:1]
    rel = (t - t0)[:, :, :nev]
    start = t[:, 0, 0] - t[:, 0, 0].min()
    print(f"launch {rec}: grid {gy} tiles, phases {nph}, SW {sw}, taps {taps}; {ok.sum()} tiles recorded, {nev} events; tile start spread {start.min()}..{start.max()} ticks")
    med = np.median(rel, axis=0)                # (2, nev)
    names = ["start", "maps", "first landed", "first barrier"]
    for k in range(taps):
        names += [f"t{k} frags", f"t{k} mfma issued", f"t{k} vmcnt", f"t{k} barrier"]
    names += ["epilogue"]
    for w in range(2):
        print(f"  wave {0 if w == 0 else 3}: total {med[w, nev - 1]:.0f} ticks")
        prev = 0
        line = []
        for i in range(nev):
            line.append(f"{names[i] if i < len(names) else i}:+{med[w, i] - prev:.0f}")
            prev = med[w, i]
        print("    " + "  ".join(line))
