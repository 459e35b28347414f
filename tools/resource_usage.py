#!/usr/bin/env python3
"""Compiler view (gfx950) of VGPRs / scratch / occupancy per kernel: tools/resource_usage.py file.hip"""
import re, subprocess, sys
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "-c", sys.argv[1], "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            if "rocprim" not in cur["name"]:
                print(f"{cur['name'][:100]:100s} vgpr={cur.get('VGPRs')} agpr={cur.get('AGPRs')} scratch={cur.get('ScratchSize [bytes/lane]')} occ={cur.get('Occupancy [waves/SIMD]')} lds={v.strip()}")
