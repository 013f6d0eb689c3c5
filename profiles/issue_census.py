#!/usr/bin/env python3
"""Static census of the traversal kernel's instruction mix, for the issue-slot roofline.

Compiles n-body-simulation_amd/csrc/nb_bh.hip to gfx950 assembly (hipcc cross-compiles without a
GPU), takes the plain instantiation of k_bh_force (no statistics, no softening) and counts, over the
blocks of its main loop that hold no FP64 arithmetic (the FP64 re-decisions are cold), the vector
instructions by what they cost to issue on an MI355X SIMD, as bin/microbench
(n-body-simulation_amd/csrc/microbench.hip) measured it: a plain FP32 / integer vector op 1 slot, a packed FP32 op (v_pk_*) 2, v_rsq /
v_rcp / v_sqrt 2, an op that writes a lane mask or moves a lane to a scalar register (v_cmp*,
v_readfirstlane, v_readlane) 1.6.  The mean of those weights is what one counted vector
instruction (SQ_INSTS_VALU) is priced at; an LDS instruction is priced at 3.3 plain ops (a
ds_read_b128 per visit cost 10 cycles where four v_fma cost 12).

Prints a JSON object; profiles/summarize.py stores it in profiles/traffic.json."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN2nb10k_bh_forceILi0ELb0EEEvNS_9ForceArgsE"
WEIGHTS = {"plain": 1.0, "packed": 2.0, "transcendental": 2.0, "mask": 1.6}
LDS_SLOTS = 3.3
ISSUE_RATE = 62.0e12 / 64.0  # wave-instruction slots per second an FMA-only kernel sustains: 62 T lane-ops/s of v_fma_f32 (bin/microbench, 8 waves per SIMD)


def census():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "nb_bh.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "n-body-simulation_amd", "csrc"),
                               os.path.join(ROOT, "n-body-simulation_amd", "csrc", "nb_bh.hip"), "--offload-device-only", "-S",
                               "-o", out], stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], None
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            cur = {"loop": "Loop" in m.group(2), "ins": []}
            blocks.append(cur)
        elif cur is not None and l.startswith("\t") and not l.strip().startswith((".", ";")):
            op = l.strip().split()[0]
            cur["ins"].append(op)
            if op.startswith(("s_cbranch", "s_branch")):  # a basic block ends here, label or not
                cur = {"loop": cur["loop"], "ins": []}
                blocks.append(cur)
    kinds = {k: 0 for k in WEIGHTS}
    lds = salu = 0
    for b in blocks:
        if not b["loop"] or any("_f64" in i for i in b["ins"]):
            continue
        for i in b["ins"]:
            if i.startswith("v_pk_"):
                kinds["packed"] += 1
            elif i.startswith(("v_rsq", "v_rcp", "v_sqrt")):
                kinds["transcendental"] += 1
            elif i.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
                kinds["mask"] += 1
            elif i.startswith("v_"):
                kinds["plain"] += 1
            elif i.startswith("ds_"):
                lds += 1
            elif i.startswith("s_"):
                salu += 1
    valu = sum(kinds.values())
    w = sum(WEIGHTS[k] * v for k, v in kinds.items()) / valu
    return {"kernel": "k_bh_force<FM_NOEPS, false>", "hot_loop_vector_instructions": kinds, "hot_loop_lds_instructions": lds,
            "hot_loop_scalar_instructions": salu, "slot_weights": WEIGHTS, "slots_per_vector_instruction": w,
            "slots_per_lds_instruction": LDS_SLOTS, "issue_rate_wave_slots_per_s": ISSUE_RATE,
            "note": "static mix of the main loop's blocks without FP64 arithmetic; weights and issue rate measured by bin/microbench"}


if __name__ == "__main__":
    json.dump(census(), sys.stdout, indent=1)
    print()
