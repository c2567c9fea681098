#!/usr/bin/env python3
"""Turn rocprofv3 outputs (gpurun_out/p1..p3) into the tracked summaries under profiles/.

  p1: --kernel-trace --stats -f csv          -> profiles/<tag>_kernel_stats.csv
  p2: --pmc FETCH_SIZE --kernel-trace -f csv  } -> profiles/pmc_traffic.json (bytes per launch, per kernel)
  p3: --pmc WRITE_SIZE --kernel-trace -f csv  }
  p5: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -f csv (-o mfma)
      -> profiles/<tag>_mfma_util.json: per kernel, MFMA-busy cycles / (active cycles x 1024 SIMDs).
      GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is 64 cycles per
      v_mfma_f32_32x32x2_f32 (32 per 16x16x4), summed over all SIMDs.
FETCH_SIZE/WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide
coalesced reads (MI355X_MICROARCH.md, HBM section) and is doubled here.
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
commit = sys.argv[2] if len(sys.argv) > 2 else "unknown"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def short(name):
    m = re.search(r"conv_igemm_kernel<([^>]*)>", name)
    if m:
        args = [a.strip() for a in m.group(1).split(",")]
        args = ["1" if a == "true" else "0" if a == "false" else a for a in args]
        return "conv_igemm_kernel<" + ",".join(args) + ">"
    m = re.search(r"(conv_wino_bf16m_kernel|conv_wino_bf16_kernel|conv_wino_kernel|conv_first_kernel)<([^>]*)>", name)
    if m:
        arg = m.group(2).split(",")[0].strip()          # conv_first_kernel<3, false> -> <3>
        arg = {"true": "1", "false": "0"}.get(arg, arg)
        return f"{m.group(1)}<{arg}>"
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("dsic::", "")


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            d[k][0] += 1
            d[k][1] += float(r["Counter_Value"])
    return d


stats = [f for f in os.listdir(os.path.join(G, "p1")) if f.endswith("kernel_stats.csv")][0]
shutil.copy(os.path.join(G, "p1", stats), os.path.join(P, f"{tag}_kernel_stats.csv"))
fetch = agg(os.path.join(G, "p2", "fetch_counter_collection.csv"), "FETCH_SIZE")
write = agg(os.path.join(G, "p3", "write_counter_collection.csv"), "WRITE_SIZE")
out = {}
detail = {}
for k in sorted(set(fetch) | set(write)):
    if not (k.startswith("conv_") or k.startswith("convT_") or k in ("rate_kernel", "ssim_level_kernel", "range_encode_kernel", "tables_kernel",
                                                  "hyper_params_kernel", "image_to_nhwc8_kernel")):
        continue
    fn, fv = fetch.get(k, [0, 0.0])
    wn, wv = write.get(k, [0, 0.0])
    fb = 2.0 * fv / max(fn, 1) * 1024.0
    wb = wv / max(wn, 1) * 1024.0
    out[k] = fb + wb
    detail[k] = {"launches_sampled": fn, "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb}
out["_commit"] = commit
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
json.dump(detail, open(os.path.join(P, f"{tag}_pmc_detail.json"), "w"), indent=1)
for src, dst in (("p1_bench.log", f"{tag}_bench_under_rocprof.log"),):
    lines = [l for l in open(os.path.join(G, src)) if l.startswith("{")]
    open(os.path.join(P, dst), "w").writelines(lines)
print(json.dumps(out, indent=1))

p5 = os.path.join(G, "p5", "mfma_counter_collection.csv")
if os.path.exists(p5):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for r in csv.DictReader(open(p5)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            launches[k] += 1
    util = {}
    for k, v in acc.items():
        busy, act = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
        if busy <= 0 or act <= 0:
            continue
        cycles = act / 8.0
        util[k] = {"launches_sampled": launches[k], "active_cycles_per_launch": cycles / launches[k],
                   "mfma_busy_cycles_per_launch": busy / launches[k], "mfma_util": busy / (cycles * 1024.0),
                   "mfma_flop_per_launch": v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0 / launches[k]}
    json.dump(util, open(os.path.join(P, f"{tag}_mfma_util.json"), "w"), indent=1)
    print(json.dumps({k: round(v["mfma_util"], 3) for k, v in util.items()}, indent=1))
    util["_commit"] = commit
    json.dump(util, open(os.path.join(P, "mfma_util.json"), "w"), indent=1)
