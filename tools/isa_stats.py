#!/usr/bin/env python3
"""Per-kernel instruction mix of the gfx950 ISA of dsg_api.hip (static): python tools/isa_stats.py [substring]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/dsg_api.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                os.path.join(root, "diffsg_amd", "csrc", "dsg_api.hip")], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
pat = sys.argv[1] if len(sys.argv) > 1 else "k_resblock"
for m in re.finditer(r"\n(_ZN[^\n:]*):[^\n]*\n(.*?)\n\s+s_endpgm", s, re.S):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat not in dem:
        continue
    c = lambda r: len(re.findall(r, body))
    valu, ds = c(r"\n\s+v_(?!mfma|accvgpr)"), c(r"\n\s+ds_")
    print(f"{dem[:60]:60s} mfma={c('v_mfma')} valu={valu} exp={c('v_exp_f32')} rcp={c('v_rcp_f32')} "
          f"accmov={c('v_accvgpr')} gload={c('global_load')} gstore={c('global_store')} ds={ds} wait={c('s_waitcnt')} "
          f"nop={c('s_nop')} lines={body.count(chr(10))}")
