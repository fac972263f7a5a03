#!/usr/bin/env python3
"""Instruction budget of the neighbour-search kernel by phase ("what's missing" item 5): wave-instructions per keypoint —
VALU / SALU / LDS / VMEM read / VMEM write / SMEM — of every phase of k_accumulate_rows, for the first, the second and a late search of a
solve. Measured, not estimated: the kernel's ablation bits switch phases off one after the other, the SQ instruction counters of each
variant's launch are read with `rocprofv3 --pmc`, and a phase's cost is the difference between two variants (telescoping):

    full                           everything
    - hand-over (bit 4)            B4: count, bounds, offsets to the per-keypoint record
    - final selection (bit 2)      B3: row_select at the end of a search round
    - admission (bit 128)          B2b: ballot + compaction of the accepted candidates into the row's list (and the in-stream prunes they cause)
    - streaming (bit 1)            B2a: chunk fetches, distances
    - probes (bit 16)              B1b: hash probes issued / resolved (the chunk lists stay: they are built from what the probes return)
    - search rounds (bit 1024)     B1a: reach tests, probe batches' bookkeeping, chunk lists, round control
    = what is left                 A + V: transform, bounds, slab masks, the pool check

Two modes:
  isa_budget.py run <workload>          the workload driver (runs under rocprofv3): for every (variant, launch j) one fresh solve whose first j - 1
                                        iterations run un-ablated and whose j-th runs with the variant's mask; prints the schedule as JSON.
  isa_budget.py report <schedule.json> <counter_collection.csv> [...]      the table.
Measurement script."""
import argparse
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [("full", 0), ("no hand-over", 4), ("no final selection", 4 | 2), ("no admission", 4 | 2 | 128), ("no streaming", 4 | 2 | 128 | 1),
            ("no probes", 4 | 2 | 128 | 1 | 16), ("no search rounds", 4 | 2 | 128 | 1 | 16 | 1024)]
PHASES = ["B4 hand-over", "B3 final selection", "B2b admission + list", "B2a stream fetch + distances", "B1b hash probes", "B1a reach tests + chunk lists + round control",
          "A + V transform, bounds, pool check"]
LAUNCHES = [1, 2, 4]


def run(workload):
    import bench
    import ct_icp_amd as cia
    from ct_icp_amd import se3, synthetic as syn
    args = argparse.Namespace(map_frames=20, d_sweeps=8, d_radius=100.0, local_rank=0)
    W = bench.build_workload(workload, 0, 1, args, cia, syn, se3)
    s = cia.GnSolver(W["gm"])
    s.set_rewind(True)
    s.set_keypoints(W["raw"], W["world0"], W["t"])
    o = cia.CTICPOptions(solver=cia.GN, num_iters_icp=W["ipf"], min_number_neighbors=W["min_nb"], threshold_orientation_norm=0.0, debug_print=False)
    schedule = []
    NO_SPLIT, NO_GUESS = 1 << 19, 1 << 24
    # every variant runs the FUSED kernel (pool check inside k_accumulate_rows: an ablated launch never takes the split path, so the
    # un-ablated one must not either) and, for the first search, the radius instead of the guessed bound (a guess whose final selection is
    # ablated counts as failed and is searched again: the differences would be of two different computations)
    for j in LAUNCHES:
        base = NO_SPLIT | (NO_GUESS if j == 1 else 0)
        for name, mask in VARIANTS:
            s.set_ablation(NO_SPLIT)
            s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"])
            if j > 1:
                s.gn_iterate(j - 1)
            s.set_ablation(mask | base)
            s.gn_iterate(1)
            s.gn_end()
            schedule.append(dict(launch=j, variant=name, mask=mask | base, search_dispatches=j, measured_dispatches=1))
    # the default launches as they run (guessed first search; from the third search on, k_pool_check + k_accumulate_rows on the list it leaves
    # when the frame is large enough for the split): totals only
    for j in LAUNCHES:
        s.set_ablation(0)
        s.rewind(); s.gn_begin(W["pose0"], W["inp"]["tbe"], o, W["mm"])
        s.gn_iterate(j)
        s.gn_end()
        split = len(W["t"]) >= 400000
        disp = sum(2 if (split and i >= 3) else 1 for i in range(1, j + 1))
        schedule.append(dict(launch=j, variant="default", mask=0, search_dispatches=disp, measured_dispatches=2 if (split and j >= 3) else 1))
    s.set_ablation(0)
    print(json.dumps(dict(workload=workload, keypoints=int(len(W["t"])), schedule=schedule)))


def report(schedule_path, csv_paths):
    sched = json.load(open(schedule_path))
    n = sched["keypoints"]
    per = collections.defaultdict(dict)          # (launch, variant) -> counter -> value
    for path in csv_paths:
        rows = [r for r in csv.DictReader(open(path)) if "k_accumulate_rows" in r.get("Kernel_Name", "") or "k_pool_check" in r.get("Kernel_Name", "")]
        by_counter = collections.defaultdict(list)
        for r in rows:
            by_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, vals in by_counter.items():
            vals.sort()
            at = 0
            for e in sched["schedule"]:
                at += e["search_dispatches"]
                per[(e["launch"], e["variant"])][c] = sum(v for _, v in vals[at - e["measured_dispatches"]:at])    # the last search dispatch(es) of the solve
            assert at == len(vals), (c, at, len(vals))
    counters = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"]
    print(f"workload {sched['workload']}, {n} keypoints: wave-instructions per keypoint of k_accumulate_rows, by phase (PMC counters of ablated launches, differenced)")
    for j in LAUNCHES:
        print(f"\nsearch {j} of a fresh solve" + (" (no carried-over bound; on the radius: the default launch starts from a guessed bound, last line)" if j == 1 else " (carried-over bound, builds the pools)" if j == 2 else " (pool check + the searches it leaves)"))
        print(f"  {'phase':52s}" + "".join(f"{c[9:]:>10s}" for c in counters))
        names = [v[0] for v in VARIANTS]
        tot = per[(j, "full")]
        for i, ph in enumerate(PHASES):
            a = per[(j, names[i])]
            b = per[(j, names[i + 1])] if i + 1 < len(names) else {c: 0.0 for c in counters}
            print(f"  {ph:52s}" + "".join(f"{(a.get(c, 0.0) - b.get(c, 0.0)) / n:10.2f}" for c in counters))
        print(f"  {'TOTAL (un-ablated launch of this variant set)':52s}" + "".join(f"{tot.get(c, 0.0) / n:10.2f}" for c in counters))
        d = per[(j, "default")]
        print(f"  {'the DEFAULT launch (guess / split as they run)':52s}" + "".join(f"{d.get(c, 0.0) / n:10.2f}" for c in counters))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        report(sys.argv[2], sys.argv[3:])
