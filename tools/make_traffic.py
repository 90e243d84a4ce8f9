#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py output for ONE kernel section):
    python tools/make_traffic.py <pmc_summary.txt> "<section header substring>" <out.json> [summary path to cite]
HBM bytes per launch = FETCH_SIZE x 2 (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM) + WRITE_SIZE, both
reported in KiB by rocprofv3; valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA."""
import json, re, sys
src, pat, out = sys.argv[1], sys.argv[2], sys.argv[3]
cite = sys.argv[4] if len(sys.argv) > 4 else src
sec, vals = None, {}
for line in open(src):
    if line.startswith("== "):
        sec = line[3:].strip()
        continue
    if sec is not None and pat in sec:
        m = re.match(r"(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
        if m:
            vals[m.group(1)] = float(m.group(3))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
d = {"kernel": pat, "summary": cite,
     "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_bench.sh), mean per dispatch; FETCH_SIZE "
               "doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); counters are in KiB",
     "note": "FETCH_SIZE counts the L2's fabric-side read requests (Infinity-Cache hits included): an upper bound of HBM bytes.  The up "
             "block reads its concat input twice (stage 1, then the Linear shortcut): both reads are in the figure.  Round 5 sweep "
             "(profiles/r05_concat_read_sweep.txt): at 65 536 / 131 072 / 262 144 rows -- the last past the 256 MB Infinity Cache -- FETCH per row "
             "stays flat and the time per row falls, so the second read costs no time at any size (the kernel is not bound by these "
             "bytes); whether it is served on-die is neither shown nor needed; DESIGN.md 3.4",
     "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
     "valu_per_mfma": vals["SQ_INSTS_VALU"] / vals["SQ_INSTS_MFMA"] if "SQ_INSTS_MFMA" in vals else None,
     "vmem_rd_per_wave": vals.get("SQ_INSTS_VMEM_RD", 0) / vals["SQ_WAVES"] if "SQ_WAVES" in vals else None,
     "wait_any_share": vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in vals else None,
     "wait_inst_share": vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in vals else None}
# whole step: sum over the sections of the summary, each times its launches per reverse step ("name=count" arguments after the citation)
step = {}
if len(sys.argv) > 5:
    want = dict(a.rsplit("=", 1) for a in sys.argv[5:])
    sec, per = None, {}
    for line in open(src):
        if line.startswith("== "):
            sec = line[3:].strip(); continue
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+mean=(\S+)", line)
        if sec in want and m:
            per.setdefault(sec, {})[m.group(1)] = float(m.group(3))
    total, parts = 0.0, {}
    for k, cnt in want.items():
        if k in per and "FETCH_SIZE" in per[k] and "WRITE_SIZE" in per[k]:
            b = (2 * per[k]["FETCH_SIZE"] + per[k]["WRITE_SIZE"]) * 1024 * int(cnt)
            parts[k] = int(b); total += b
    step = {"bytes_per_step": int(total), "per_kernel_bytes": parts, "missing": [k for k in want if k not in parts],
            "note": "sum of (2 x FETCH_SIZE + WRITE_SIZE) per dispatch x launches per reverse step over the step's large kernels; fabric-side "
                    "traffic (Infinity-Cache hits included), an upper bound of the HBM bytes"}
d["step_traffic"] = step
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d))
