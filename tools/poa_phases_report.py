#!/usr/bin/env python3
"""BUILD CONTAINER: profiles/rNN_poa_phases.json from the GPU box's gpurun_out/poa_phases/poa_phases_raw.json (tools/poa_phases.sh) and the ISA of the
`base` variant (tools/build_variant.sh base -> /tmp/th_variant_base/*.s).

What it holds, per launch of thpoa_n::th_k_poa<3072, short, true> over 100 000 headline reads:
  phases        vector / scalar wave-instructions, kernel milliseconds and SQ wave-cycles of the row tables, the DP, the traceback and the consensus, each the
                difference between the variant that runs that phase twice and the variant that runs none twice (same code: opaque one-trip loops), and
                "merge + seed + job set-up" as what is left of the plain variant's totals; the sum is checked against the PMC totals
  dp_rows       rows per path of the DP's row loop and further trips of packed rows, counted by the -DTH_POA_PROF=1 build (one LDS add per row)
  row_tables    chunks and relaxation rounds of the row-table phase by loop kind
  traceback     tiles loaded, run trips / steps, single steps
  isa           static instruction counts of the packed row's trip bodies (basic blocks with the three interleaved 6-step DPP scans) in the product's ISA
  before        the same phase split at the end of the round before
usage: tools/poa_phases_report.py [raw.json] [isa.s] > profiles/r06_poa_phases.json"""
import glob, json, re, subprocess, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "gpurun_out/poa_phases/poa_phases_raw.json")))
isa = sys.argv[2] if len(sys.argv) > 2 else (glob.glob("/tmp/th_variant_base/*gfx950*.s") or [None])[0]
pmc, ms = raw["pmc"], raw["poa_ms"]
mean = lambda v: sum(v) / len(v) if v else None
def d(a, b, k): return pmc[a].get(k, 0.0) - pmc[b].get(k, 0.0)
names = {"dup0": "row_tables", "dup1": "dp", "dup2": "traceback", "dup3": "consensus"}
phases, tot = {}, {"valu": 0.0, "salu": 0.0, "ms": 0.0}
for v, n in names.items():
    phases[n] = {"valu_wave_instr": d(v, "dup9", "SQ_INSTS_VALU"), "salu_wave_instr": d(v, "dup9", "SQ_INSTS_SALU"), "wave_cycles": d(v, "dup9", "SQ_WAVE_CYCLES"),
                 "ms_marginal": mean(ms[v]) - mean(ms["dup9"])}
    tot["valu"] += phases[n]["valu_wave_instr"]; tot["salu"] += phases[n]["salu_wave_instr"]; tot["ms"] += phases[n]["ms_marginal"]
ref = "dup9"
phases["merge_seed_setup"] = {"valu_wave_instr": pmc[ref]["SQ_INSTS_VALU"] - tot["valu"], "salu_wave_instr": pmc[ref]["SQ_INSTS_SALU"] - tot["salu"],
                              "ms_marginal": None, "note": "what is left of the no-phase-repeated variant's totals (times are not additive: phases of different waves overlap)"}
out = {
    "what": __doc__.split("\n\n")[1],
    "kernel": "thpoa_n::th_k_poa<3072, short, true>, 100 000 reads of config 3 per launch (bench.py --quick)",
    "totals": {"product_build": {"valu_wave_instr": pmc["base"]["SQ_INSTS_VALU"], "salu_wave_instr": pmc["base"]["SQ_INSTS_SALU"], "wave_cycles": pmc["base"]["SQ_WAVE_CYCLES"],
                                 "ms": mean(ms["base"]), "seconds_under_counters": pmc["base"]["seconds_under_counters"]},
               "attribution_build_no_phase_repeated": {"valu_wave_instr": pmc[ref]["SQ_INSTS_VALU"], "salu_wave_instr": pmc[ref]["SQ_INSTS_SALU"], "ms": mean(ms[ref])}},
    "phases": phases,
    "check": {"phase_sum_over_totals_valu": (tot["valu"] + phases["merge_seed_setup"]["valu_wave_instr"]) / pmc[ref]["SQ_INSTS_VALU"],
              "four_phases_share_of_valu": tot["valu"] / pmc[ref]["SQ_INSTS_VALU"], "four_phases_share_of_salu": tot["salu"] / pmc[ref]["SQ_INSTS_SALU"],
              "product_vs_attribution_build_valu": pmc["base"]["SQ_INSTS_VALU"] / pmc[ref]["SQ_INSTS_VALU"]},
}
pe = raw.get("prof_events_build")
if pe:
    rows = {k: v["rows"] for k, v in pe["dp_rows"].items()}
    n_rows = sum(rows.values())
    out["dp_rows"] = {"rows_per_launch": n_rows, "by_path": rows, "share": {k: v / n_rows for k, v in rows.items()}, "packed_further_trips": pe["packed_further_trips"],
                      "per_row": {"dp_valu": phases["dp"]["valu_wave_instr"] / n_rows, "dp_salu": phases["dp"]["salu_wave_instr"] / n_rows,
                                  "kernel_valu": pmc[ref]["SQ_INSTS_VALU"] / n_rows, "kernel_salu": pmc[ref]["SQ_INSTS_SALU"] / n_rows},
                      "paths": "common_packed: one predecessor = the previous row, straight-line set-up; packed_one_pred / two / three_four: the bubble path (every predecessor in the ring of "
                               "the last three rows); packed_far: one predecessor or a reader outside the ring (full cells in HBM); general_*: one column per lane in 32 bits"}
    out["traceback"] = pe["traceback"]; out["jobs"] = pe["jobs"]
    if raw.get("prof_rows"): out["row_tables"] = raw["prof_rows"]
if isa:
    txt = subprocess.run([sys.executable, os.path.join(R, "tools/isa_blocks.py"), isa, "_ZN7thpoa_n8th_k_poaILi3072EsLb1E", "60"], capture_output=True, text=True).stdout
    trips = []
    for l in txt.splitlines():
        m = re.match(r"(\S+)\s+n\s+(\d+)\s+VALU\s+(\d+) \(lane\s+(\d+) dpp\s+(\d+) pk\s+(\d+)\)\s+SALU\s+(\d+) \(wait\s+(\d+) nop\s+(\d+)\)", l)
        if m and int(m.group(5)) == 20:
            trips.append({"block": m.group(1), "instructions": int(m.group(2)), "valu": int(m.group(3)), "lane_ops": int(m.group(4)), "packed": int(m.group(6)), "salu": int(m.group(7)), "s_nop": int(m.group(9))})
    out["isa"] = {"trip_bodies": trips, "note": "basic blocks of the product kernel that hold the three interleaved 6-step DPP scans = the trip bodies of th_poa_pk_row's instantiations, in address order: "
                  "(common row: first trip incl. the row's scalar set-up, further trip), (bubble row, one / two / three-four predecessors: first, further), (far row: first, further), general path's packed row"}
# the same split at the end of the round before, for the record of where this round's cuts went
try:
    prev = json.load(open(os.path.join(R, "profiles", "r05_poa_phases.json")))
    out["before"] = {"note": "end of round 5 (profiles/r05_poa_phases.json), same method", "totals": prev["totals"]["product_build"], "phases": prev["phases"], "traceback": prev.get("traceback")}
except (OSError, ValueError, KeyError):
    pass
json.dump(out, sys.stdout, indent=1)
print()
