#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of profiles/collect.sh (gpurun_out/prof_<round>/) into the
committed summaries:

    profiles/<round>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of `python3 bench.py`
    profiles/<round>_bench.json           the JSON line that run printed
    profiles/<round>_pmc.json             per-kernel averages of the counter passes
    profiles/traffic.json                 HBM bytes per launch of the dominant kernels (read by bench.py)

HBM bytes follow MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE are in KiB, from
separate --pmc passes; on gfx950 FETCH_SIZE reports half the bytes of 16 B/lane loads, so it is
doubled; WRITE_SIZE is taken as it is.  `traffic` = 2 x FETCH_SIZE + WRITE_SIZE per launch (the
traversal's loads are 16 B/lane gathers and 8 B/lane loads: the doubling is exact for the former
and an upper estimate for the latter; the raw figure is kept beside it).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.split("(")[0]
    name = name.replace("void ", "").replace("nb::", "")
    return name.strip()


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + rnd)
    out = os.path.join(ROOT, "profiles")
    stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    shutil.copy(stats, os.path.join(out, rnd + "_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(out, rnd + "_bench.json"))
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # The counter passes run the headline workload alone: its first three steps, then on to step 300
    # and two more.  The launches of the traversal kernel are taken apart accordingly (in dispatch
    # order): "@early" = the first three, "@late" = the last two.
    for k in [k for k in pmc if k.startswith("k_bh_force<") and "false" in k]:
        for c, v in list(pmc[k].items()):
            if len(v) >= 8:
                pmc[k + "@early"][c] = v[:3]
                pmc[k + "@late"][c] = v[-2:]
    summary = {}
    for k, cs in sorted(pmc.items()):
        summary[k] = {c: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in sorted(cs.items())}
        h, m = cs.get("TCC_HIT_sum"), cs.get("TCC_MISS_sum")
        if h and m:
            summary[k]["l2_hit_rate"] = sum(h) / (sum(h) + sum(m))
    # instructions per node visit of the traversal: the counters of its plain instantiation over the
    # wave-node visits the bench counted for the same workload (its statistics run), this round and,
    # for comparison, the committed figures of the round before
    def per_visit(counters, bench_line):
        visits = (bench_line.get("tree") or {}).get("group_node_visits")
        if not visits:
            return None
        return {c.replace("SQ_INSTS_", "").lower(): counters[c]["mean"] / visits
                for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if c in counters} | {"wave_node_visits": visits}

    plain = [k for k in summary if k.startswith("k_bh_force<") and "false" in k and k.endswith("@early")]
    if plain:
        line = json.loads(open(os.path.join(src, "bench.json")).read().strip().split("\n")[-1])
        summary["k_bh_force_per_visit"] = {rnd: per_visit(summary[plain[0]], line)}
        accepted = (line.get("tree") or {}).get("accepted_interactions")
        if accepted:
            summary["k_bh_force_per_accepted_interaction"] = {
                c.replace("SQ_INSTS_", "").lower() + "_lane_slots": summary[plain[0]][c]["mean"] * 64 / accepted
                for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS") if c in summary[plain[0]]}
        prev = "r%02d" % (int(rnd[1:]) - 1)
        try:
            old = json.load(open(os.path.join(out, prev + "_pmc.json")))
            old_line = json.loads(open(os.path.join(out, prev + "_bench.json")).read().strip().split("\n")[-1])
            old_plain = [k for k in old if k.startswith("k_bh_force<") and "false" in k and "@" not in k]
            summary["k_bh_force_per_visit"][prev] = per_visit(old[old_plain[0]], old_line)
        except (OSError, ValueError, IndexError, KeyError):
            pass
    json.dump(summary, open(os.path.join(out, rnd + "_pmc.json"), "w"), indent=1, sort_keys=True)
    traffic = {}
    # the plain traversal instantiation (the other is its statistics variant), early and late
    base = [k[: -len("@early")] for k in summary if k.startswith("k_bh_force<") and "false" in k and k.endswith("@early")]
    for k, tag, what in ((base[0] + "@early" if base else "", "k_bh_force", "steps 2-3 after the bodies are set"),
                         (base[0] + "@late" if base else "", "k_bh_force_late", "steps 300-301")):
        if k in summary and "FETCH_SIZE" in summary[k] and "WRITE_SIZE" in summary[k]:
            fetch = summary[k]["FETCH_SIZE"]["mean"] * 1024
            write = summary[k]["WRITE_SIZE"]["mean"] * 1024
            traffic[tag] = {
                "hbm_bytes_per_launch": 2 * fetch + write,
                "fetch_bytes_raw": fetch,
                "fetch_bytes_doubled": 2 * fetch,
                "write_bytes": write,
                "l2_hit_rate": summary[k].get("l2_hit_rate"),
                "workload": "galaxy collision N=5,000,000, %s (bench.py --steps 2 --warmup 1 --headline-only --no-prewarm "
                            "--late-step 300 --later-step 0)" % what,
                "round": rnd,
            }
            for c, name in (("SQ_INSTS_VALU", "valu_wave_insts"), ("SQ_INSTS_SALU", "salu_wave_insts"),
                            ("SQ_INSTS_LDS", "lds_wave_insts")):
                if c in summary[k]:
                    traffic[tag][name] = summary[k][c]["mean"]
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from issue_census import census

        traffic["issue_census"] = census()
    except Exception as e:  # no hipcc here: the counters stand without the weights
        traffic["issue_census"] = {"error": str(e)}
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
