#!/usr/bin/env python3
"""Turns the stderr of ONE bench step of the profiling build (tools/build_variant.sh prof -DTH_ALN_PROF=1 -DTH_POA_PROF=1 -DTH_CHAIN_PROF=1;
TH_BENCH_LIB=tests/cpu_emul/libth_prof.so python3 bench.py --quick --steps 1 --warmup 0 2> prof.err) into the two committed profiles of round 6:

  profiles/rNN_dp_cells.json      DP cells per launch of the three big kernels -- what bench.py's `roofline.secondary` divides the instruction counts by:
                                  th_k_poa: band cells (sum over graph rows of e - b + 1), th_k_align: cells the engines computed (anti-diagonals x 128 x registers of
                                  every register band run), th_k_chain_dp: predecessor evaluations (64 per trip).  A property of workload + algorithm, not of the build.
  profiles/rNN_align_phases.json  th_k_align's launch by class of item: items, wave time, cells computed, cells of the band the item's final score certifies.

usage: tools/prof_report.py prof.err reads_per_launch config out_prefix      (e.g. ... 100000 c3 profiles/r06)"""
import json
import re
import sys


def lines(tag, text):
    return [json.loads(m.group(1)) for m in re.finditer(r"^\[%s\] (\{.*\})\s*$" % re.escape(tag), text, re.M)]


def main():
    err, reads, config, prefix = open(sys.argv[1]).read(), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    # the LAST batch's lines (earlier ones belong to warm-up steps, which process the same batch)
    poa = lines("poa_prof", err)[-1]
    chain = lines("chain_prof", err)[-1]
    aln_l = lines("aln_prof", err)
    cls_l = lines("aln_prof_classes", err)
    # the launches of the last batch: as many as distinct launch names, taken from the end
    names = []
    for x in reversed(cls_l):
        if x["launch"] in names:
            break
        names.append(x["launch"])
    n_l = len(names)
    aln_l, cls_l = aln_l[-n_l:], cls_l[-n_l:]
    rows = sum(v["rows"] if isinstance(v, dict) else 0 for v in poa["dp_rows"].values())
    poa_cells = sum(v["cycles"] if isinstance(v, dict) else 0 for v in poa["dp_rows"].values())      # (-DTH_POA_PROF=1: the field carries band cells)
    aln_cells = sum(c["cells_computed"] for L in cls_l for c in L["classes"])
    aln_need = sum(c["cells_certified_band"] for L in cls_l for c in L["classes"])
    cells = {
        "what": __doc__.split("usage:")[0].strip(),
        "config": config, "reads_per_launch": reads,
        "kernels": {
            "th_k_poa": {"dp_cells_per_launch": poa_cells, "graph_rows": rows, "mean_band_columns": poa_cells / max(rows, 1), "jobs": poa["jobs"],
                         "cell": "one (graph row, query column) cell of the banded convex-gap recurrence: H, E1, E2, F1, F2 and its direction word"},
            "th_k_align": {"dp_cells_per_launch": aln_cells, "cells_of_the_certified_bands": aln_need,
                           "cell": "one cell of ksw2's affine recurrence (H, E, F; the unit alignments also carry the identity count); strip-engine and full-matrix items (0.0003 % of the items here) not counted"},
            "th_k_chain_dp": {"dp_cells_per_launch": chain["pair_evaluations"], "heads_scanned": chain["heads_scanned"], "trips": chain["trips"], "hits": chain["hits"], "followers": chain["followers"],
                              "cell": "one (hit, predecessor) evaluation of get_con_score + the DP update (src/tandem_chain.c:151-166,325-356); followers need none"},
        },
    }
    json.dump(cells, open(prefix + "_dp_cells.json", "w"), indent=1)
    out = {"what": "th_k_align at %s, %d reads per launch: where the launch's wave time goes by class of item (the -DTH_ALN_PROF=1 build; ticks = s_memtime, 100 MHz; a wave works on ONE item "
                   "at a time, so the classes' ticks sum to the waves' busy time)" % (config, reads), "launches": []}
    for a, c in zip(aln_l, cls_l):
        busy = sum(x["ticks"] for x in c["classes"])
        L = {"launch": a["launch"], "waves_that_took_items": a["waves_that_took_items"], "items": a["items"],
             "busy_ticks": {"shortest": a["busy_ticks_shortest"], "mean": a["busy_ticks_mean"], "longest": a["busy_ticks_longest"],
                            "mean_over_longest": a["busy_ticks_mean"] / max(a["busy_ticks_longest"], 1)},
             "costliest_item": {"ticks": a["costliest_item_ticks"], "flank_bases": a["its_tlen"], "consensus_bases": a["its_cl"], "share_of_longest_wave": a["costliest_item_ticks"] / max(a["busy_ticks_longest"], 1)},
             "classes": []}
        for x in c["classes"]:
            if not x["items"]:
                continue
            L["classes"].append({"class": x["class"], "items": x["items"], "share_of_items": x["items"] / max(a["items"], 1), "ticks": x["ticks"], "share_of_wave_time": x["ticks"] / max(busy, 1),
                                 "ticks_per_item": x["ticks"] / x["items"], "cells_computed": x["cells_computed"], "cells_of_the_certified_band": x["cells_certified_band"],
                                 "computed_over_certified": (x["cells_computed"] / x["cells_certified_band"]) if x["cells_certified_band"] else None})
        out["launches"].append(L)
    json.dump(out, open(prefix + "_align_phases.json", "w"), indent=1)
    print(json.dumps({k: v["dp_cells_per_launch"] for k, v in cells["kernels"].items()}))


if __name__ == "__main__":
    main()
