#!/usr/bin/env python3
"""Static look at the gfx950 code object of the built library for the patterns that cost time without showing up as scratch or registers
(DESIGN.md 9):  python tools/isa_lint.py [substring of kernel names ...]
per kernel: instructions, FLAT accesses (a pointer read from memory: as_global it), global loads that are followed within four
instructions by `s_waitcnt vmcnt(0)` (a load inside a wave-uniform branch, or a load-use pair the scheduler could not separate), full
LDS drains (`lgkmcnt(0)`), register moves, branches, MFMAs, and the register moves that sit inside LOOP bodies (a prefetch the compiler rotates
through moves every iteration: the round-1 exact-float32 block kernel spent 56 per k-group that way).  Runs on the CPU (llvm-objdump from the ROCm tree)."""
import os, re, shutil, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FILT = next((f for f in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/usr/bin/c++filt") if os.path.exists(f)), "")


def disassemble(so):
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(so, os.path.join(tmp, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
        return subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


LOOP_MOVES = {}    # kernel symbol -> register moves (v_mov / v_accvgpr) inside loop bodies, i.e. between a backward branch and its target


def kernels(dis):
    syms = [(m.start(), m.group(1)) for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:$", dis, re.M)]
    for k, (pos, name) in enumerate(syms):
        body = dis[pos:syms[k + 1][0] if k + 1 < len(syms) else len(dis)]
        ins, addr = [], []
        for l in body.split("\n"):
            if "\t" not in l:
                continue
            txt = l.split("\t")[1].split("//")[0].strip()
            m = re.search(r"//\s*([0-9A-F]+):", l)
            if txt:
                ins.append(txt)
                addr.append(int(m.group(1), 16) if m else -1)
        # loop bodies: a branch with a negative offset closes a loop that starts at its target (a prefetch rotated through register
        # moves every iteration shows up here: round 5, k_resblock<128, true> -- 56 moves per k-group)
        idx = {a: i for i, a in enumerate(addr) if a >= 0}
        inloop = set()
        for i, t in enumerate(ins):
            if t.startswith("s_cbranch") or t.startswith("s_branch"):
                try:
                    off = int(t.split()[-1])
                except ValueError:
                    continue
                off = off - 65536 if off >= 32768 else off
                tgt = addr[i] + 4 + 4 * off
                if off < 0 and tgt in idx:
                    inloop.update(range(idx[tgt], i + 1))
        LOOP_MOVES[name] = sum(1 for i in inloop if ins[i].startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr")))
        yield name, ins


def _vregs(operand_text):
    """VGPR numbers named in an operand list: v7, v[4:7] (accumulators a[..] are a different file)."""
    regs = set()
    for m in re.finditer(r"(?<![a-z_0-9])v(\d+)\b", operand_text):
        regs.add(int(m.group(1)))
    for m in re.finditer(r"(?<![a-z_0-9])v\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return regs


def mixhi_hazards(ins):
    """gfx940+ "dst_sel forwarding" hazard: v_fma_mixhi_f16 writes HALF of its destination register; a vector or matrix
    instruction issued directly behind it that names that register gets the old contents.  hipcc pads the pairs it emits itself, not
    inline asm (split_pair closes with s_nop 0; split_pair_nowait relies on its call site): list every v_fma_mixhi_f16 whose NEXT
    instruction is a v_* instruction mentioning its vdst.  Any instruction in between (s_nop, scalar, memory) is the wait state."""
    bad = []
    for k, t in enumerate(ins[:-1]):
        if not t.startswith("v_fma_mixhi_f16"):
            continue
        m = re.match(r"v_fma_mixhi_f16\s+v(\d+)\s*,", t)
        nxt = ins[k + 1]
        if m and nxt.startswith("v_") and int(m.group(1)) in _vregs(nxt.split(None, 1)[1] if " " in nxt else ""):
            bad.append((k, t, nxt))
    return bad


def demangle(names):
    if not FILT:
        return {n: n for n in names}
    out = subprocess.run([FILT], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    want = sys.argv[1:]
    dis = disassemble(os.path.join(ROOT, "diffsg_amd", "libdiffsg_hip.so"))
    rows = []
    hazards = {}
    for name, ins in kernels(dis):
        if not name.startswith("_ZN3dsg"):
            continue
        hz = mixhi_hazards(ins)
        if hz:
            hazards[name] = hz
        ops = Counter(i.split()[0] for i in ins)
        waits = [k for k, i in enumerate(ins) if i.startswith("s_waitcnt")]
        vm0 = [k for k in waits if "vmcnt(0)" in ins[k]]
        load_then_wait = sum(1 for k in vm0 if any(x.startswith("global_load") or x.startswith("flat_load") for x in ins[max(0, k - 4):k]))
        rows.append((name, len(ins), sum(v for o, v in ops.items() if o.startswith("flat_")), load_then_wait, len(vm0),
                     sum(1 for k in waits if "lgkmcnt(0)" in ins[k]), ops.get("v_mov_b32_e32", 0) + ops.get("v_mov_b64_e32", 0),
                     sum(v for o, v in ops.items() if o.startswith("s_cbranch")), sum(v for o, v in ops.items() if o.startswith("v_mfma")), LOOP_MOVES.get(name, 0)))
    names = demangle([r[0] for r in rows])
    print(f"{'kernel':72s} {'instr':>6s} {'flat':>5s} {'ld->vm0':>7s} {'vm0':>4s} {'lgkm0':>5s} {'v_mov':>5s} {'br':>4s} {'mfma':>5s} {'mov@loop':>8s}")
    for r in sorted(rows, key=lambda r: -r[1]):
        dn = re.sub(r"\(.*", "", names[r[0]]).replace("void ", "").replace("dsg::", "")
        if want and not any(w in dn for w in want):
            continue
        print(f"{dn[:72]:72s} {r[1]:6d} {r[2]:5d} {r[3]:7d} {r[4]:4d} {r[5]:5d} {r[6]:5d} {r[7]:4d} {r[8]:5d} {r[9]:8d}")


    print(f"v_fma_mixhi_f16 -> dependent vector instruction with no wait state between them: {sum(len(v) for v in hazards.values())} in {len(hazards)} kernel(s)")
    for name, hz in hazards.items():
        for k, a, b in hz[:4]:
            print(f"  HAZARD {names.get(name, name)[:60]} @{k}: {a}  ->  {b}")


if __name__ == "__main__":
    main()
