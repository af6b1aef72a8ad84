"""The three-term bf16 kernels issue their MFMAs as asm statements, so the compiler pads no wait states around them.  This reads the ISA of those files and
checks, for every v_mfma, that no VALU instruction within the WAIT issue slots in front of it writes one of its source registers, and that no VALU / LDS / VMEM
instruction within WAIT slots behind the LAST mfma of an accumulator chain reads its destination (s_nop N counts N + 1 slots).
Also: every v_dot2c_f32_bf16 of the split (lfsr_split_pair) must take its selector from an SGPR -- as an inline constant (-1.0) the hardware reads the f32 pattern, i.e. the
other half of the pair.
usage: python tools/check_asm_mfma_hazards.py [file.hip ...]   (default: rowgemm_b3.hip ffn_b3.hip up_tail.hip epi_b3.hip lnlin_b3.hip)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = [d for d in os.listdir(ROOT) if d.endswith("_amd")][0] + "/csrc"
files = sys.argv[1:] or ["rowgemm_b3.hip", "ffn_b3.hip", "up_tail.hip", "epi_b3.hip", "lnlin_b3.hip"]
NEED_BEFORE, NEED_AFTER = 2, 18

def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()

bad = 0
for f in files:
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", t.name, os.path.join(ROOT, CSRC, f)],
                              stderr=subprocess.DEVNULL)
        lines = [l.split(";")[0].strip() for l in open(t.name)]
    ins = [l for l in lines if l and not l.startswith((".", "//")) and not l.endswith(":")]
    n_mfma = 0
    n_dot = 0
    for l in ins:
        if l.startswith("v_dot2c_f32_bf16"):
            n_dot += 1
            ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
            if not re.fullmatch(r"s\d+", ops[1]):
                print(f"{f}: split selector is not an SGPR:\n    {l}"); bad += 1
    for i, l in enumerate(ins):
        if not l.startswith("v_mfma") or "bf16" not in l: continue
        n_mfma += 1
        ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
        dst, srcs = regs(ops[0]), set().union(*[regs(o) for o in ops[1:4]])
        # backwards: VALU writers of the sources
        slots, j = 0, i - 1
        while j >= 0 and slots < NEED_BEFORE:
            p = ins[j]
            if p.startswith("s_nop"): slots += int(p.split()[1]) + 1
            elif p.startswith("v_mfma"): slots += 4
            else:
                if p.startswith("v_") and regs([o.strip() for o in p.split(None, 1)[1].split(",")][0]) & srcs:
                    print(f"{f}: VALU write {slots} slot(s) before its MFMA use:\n    {p}\n    {l}"); bad += 1
                slots += 1
            j -= 1
        # forwards: readers of the destination that are not MFMAs accumulating into it
        slots, j = 0, i + 1
        while j < len(ins) and slots < NEED_AFTER:
            p = ins[j]
            if p.startswith("s_nop"): slots += int(p.split()[1]) + 1
            elif p.startswith("v_mfma"):
                slots += 4
                if regs([o.strip() for o in p.split(None, 1)[1].split(",")][0]) == dst: break      # the chain goes on: checked at its last link
            elif p.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier")): break
            else:
                toks = [o.strip() for o in p.split(None, 1)[1].split(",")] if " " in p else []
                rd = set().union(*[regs(o.split()[0]) for o in (toks[1:] if p.startswith("v_") else toks) if o]) if toks else set()
                if rd & dst and p.startswith(("v_", "ds_", "buffer_", "global_")):
                    print(f"{f}: read of an MFMA result {slots} slot(s) behind it:\n    {l}\n    {p}"); bad += 1
                slots += 1
            j += 1
    print(f"{f}: {n_mfma} bf16 MFMAs checked, {n_dot} v_dot2c selectors in SGPRs")
sys.exit(1 if bad else 0)
