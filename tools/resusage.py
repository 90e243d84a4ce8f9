#!/usr/bin/env python3
"""Compile dsg_api.hip with -Rpass-analysis=kernel-resource-usage and print one compact line per kernel."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "diffsg_amd", "csrc", "dsg_api.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
                      "-o", "/tmp/_resusage.so", src, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = {}
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPRs Spill", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur:
            cur[key] = int(m.group(1))
            if key == "LDS Size [bytes/block]":
                if pat in cur["name"]:
                    print(f"{cur['name'][:78]:78s} v={cur.get('VGPRs')} a={cur.get('AGPRs')} scratch={cur.get('ScratchSize [bytes/lane]')} "
                          f"occ={cur.get('Occupancy [waves/SIMD]')} spill={cur.get('VGPRs Spill')} lds={cur[key]}")
    if "error" in line:
        print(line)
