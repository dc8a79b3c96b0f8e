#!/usr/bin/env python3
"""VGPRs / scratch / LDS of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).  usage: kernel_resources.py file.hip [-D...]"""
import re, subprocess, sys
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
                      "-c", sys.argv[1], "-o", "/dev/null"] + sys.argv[2:], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.split("\n"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
    for key in ("VGPRs", "AGPRs", r"ScratchSize \[bytes/lane\]", r"LDS Size \[bytes/block\]", "SGPRs"):
        m = re.search(r"\s" + key + r": (\d+)", line)
        if m and cur:
            rows[cur][key.split(" ")[0].replace("\\", "")] = int(m.group(1))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", name)
    print(f"{name:70s} VGPR {v.get('VGPRs'):4d}  scratch {v.get('ScratchSize'):4d}  LDS {v.get('LDS'):7d}  SGPR {v.get('SGPRs')}")
