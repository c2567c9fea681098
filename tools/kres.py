#!/usr/bin/env python3
"""Register / spill / scratch summary of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kres.py csrc/file.hip [extra hipcc flags...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "domain-specific-image-compression_amd", "csrc")
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{root}/include",
       f"-I{csrc}", *sys.argv[2:], "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
def flush():
    if cur:
        name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        print(f"{name[:70]:70s} sgpr {cur.get('TotalSGPRs','?'):>4} vgpr {cur.get('VGPRs','?'):>4} agpr {cur.get('AGPRs','?'):>4} "
              f"spill s/v {cur.get('SGPRs Spill','?')}/{cur.get('VGPRs Spill','?')} scratch {cur.get('ScratchSize [bytes/lane]','?')} "
              f"occ {cur.get('Occupancy [waves/SIMD]','?')} lds {cur.get('LDS Size [bytes/block]','?')}")
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    body = m.group(1)
    if body.startswith("Function Name:"):
        flush(); cur = {"name": body.split(":", 1)[1].strip()}
    elif ":" in body:
        k, v = body.split(":", 1); cur[k.strip()] = v.strip()
flush()
