#!/usr/bin/env python3
"""profiles/traffic_train.json from a training PMC summary (tools/pmc_train.sh): fabric-side bytes per launch and per step of the
training step's large kernels, FETCH_SIZE doubled (gfx950 counts half of a 16-byte-per-lane stream) + WRITE_SIZE, both in KiB.
    python tools/make_traffic_train.py profiles/r05_pmc_train_summary.txt 7 profiles/traffic_train.json
(7 = steps the profiled command ran: tools/train_prof.py 4 -> 3 warm-up + 4 timed)."""
import json, re, sys
src, steps, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cur, d = None, {}
for line in open(src):
    if line.startswith("== "):
        cur = line[3:].strip(); d[cur] = {}
    else:
        m = re.match(r"(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
        if m and cur:
            d[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)))
rows, tot = {}, 0.0
for k, v in d.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        n, f = v["FETCH_SIZE"]; _, w = v["WRITE_SIZE"]
        per_launch = (2 * f + w) * 1024
        rows[k] = {"launches_per_step": n / steps, "bytes_per_launch": per_launch, "bytes_per_step": n / steps * per_launch}
        tot += n / steps * per_launch
out = {"summary": src, "steps_profiled": steps, "kernels": rows, "bytes_per_step_listed_kernels": tot,
       "k_wgrad_h_bytes_per_launch_mean": rows.get("k_wgrad_h", {}).get("bytes_per_launch"),
       "note": "fabric side (L2 <-> memory, Infinity-Cache hits included): an upper bound of HBM bytes; 2 x FETCH_SIZE + WRITE_SIZE; "
               "k_wgrad_h is launched five times per step with different unit lists, the figure is the mean launch"}
json.dump(out, open(dst, "w"), indent=1)
print(f"{tot / 1e9:.3f} GB per step over {len(rows)} kernels -> {dst}")
