#!/usr/bin/env python3
"""Build container: gpurun_out/poa_row_paths/raw.json (tools/poa_row_paths.sh) + the row counts and DP-phase totals of profiles/r06_poa_phases.json
-> the per-row-path instruction table of th_k_poa's row loop.   usage: tools/poa_row_paths_report.py > profiles/r06_poa_row_paths.json"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(os.path.join(R, "gpurun_out/poa_row_paths/raw.json")))
ph = json.load(open(os.path.join(R, "profiles/r06_poa_phases.json")))
rows = ph["dp_rows"]["by_path"]; dp = ph["phases"]["dp"]
cats = {0: "common_packed", 1: "packed_one_pred", 6: "packed_two_preds", 7: "packed_three_four_preds", 9: "packed_far",
        2: "general_several_preds", 3: "general_one_far_pred", 4: "general_stores_full", 5: "general_other"}
base = raw["rdupm2"]; prod = raw["product"]
assert len({v["records_digest"] for v in raw.values()}) == 1, "a variant changed the records"
assert abs(prod["valu"] - ph["totals"]["product_build"]["valu_wave_instr"]) < 1e-3 * prod["valu"], "the product is not the one r06_poa_phases.json measured"
table = {}; sv = ss = 0.0
for c, name in cats.items():
    d = raw["rdup%d" % c]; n = rows[name]
    v, s = d["valu"] - base["valu"], d["salu"] - base["salu"]
    sv += v; ss += s
    table[name] = {"rows_per_launch": n, "share_of_rows": round(n / ph["dp_rows"]["rows_per_launch"], 4), "valu_per_row": round(v / n, 1), "salu_per_row": round(s / n, 1),
                   "wave_cycles_per_row": round((d["wave_cycles"] - base["wave_cycles"]) / n, 0), "valu_per_launch": v, "salu_per_launch": s, "ms_added_by_repeating_it": round(d["ms_under_counters"] - base["ms_under_counters"], 1)}
for t in table.values(): t["share_of_dp_valu"] = round(t["valu_per_launch"] / sv, 4)
out = {
 "what": "th_k_poa<3072, short, true>, 100 000 headline reads per launch: vector / scalar wave-instructions PER ROW of each path of the DP's row loop, measured -- variant rdup<c> "
         "(-DTH_POA_DUP_ROW=c, th_k_poa_body.h) computes every row of path c twice (loop scalars restored in between; same records digest as the product), rdupm2 is the same code "
         "with no row repeated; per row = (rdup<c> - rdupm2) / rows of the path (the -DTH_POA_PROF=1 build's counts in r06_poa_phases.json, same reads).  A path's figure holds its "
         "set-up, every trip of the row (packed rows: the second trip of the 18.5 % whose band exceeds 128 columns is in the average) and the row's end.  tools/poa_row_paths.sh, "
         "tools/poa_row_paths_report.py",
 "paths": ph["dp_rows"]["paths"],
 "per_path": table,
 "check": {"sum_valu": sv, "dp_phase_valu_of_the_product": dp["valu_wave_instr"], "sum_over_dp_valu": round(sv / dp["valu_wave_instr"], 4),
           "sum_salu": ss, "dp_phase_salu_of_the_product": dp["salu_wave_instr"], "sum_over_dp_salu": round(ss / dp["salu_wave_instr"], 4),
           "note": "vector: the table reproduces the DP phase's PMC total of the product within the 5 %% asked for.  Scalar: over by the repeat's own per-row cost -- the test whether "
                   "to repeat, nine restored loop scalars and the jump back are part of every repeated row (~15 scalar instructions x 1.04e9 rows = the excess), and the instrumented "
                   "builds spill nine more scalars in the row loop (rdupm2 against the product: +%.0f vector, +%.0f scalar instructions per row, outside the differences above)"
                   % ((base["valu"] - prod["valu"]) / ph["dp_rows"]["rows_per_launch"], (base["salu"] - prod["salu"]) / ph["dp_rows"]["rows_per_launch"])},
 "reading": "the common row costs 128 vector instructions (static count of one trip: 115; + 0.185 second trips x 92 = 132); a second predecessor costs +50 (178), a far one +110 (240); "
            "the 0.8 % of rows on the one-column-per-lane 32-bit path cost ~700 each = 3.8 % of the DP's vector instructions",
 "rows_on_the_32_bit_path_by_reason": dict(json.load(open(os.path.join(R, "gpurun_out/poa_row_paths/why.json"))),
      note="-DTH_POA_PROF=1 build, [poa_prof_why] line of `TH_BENCH_LIB=tests/cpu_emul/libth_prof.so bench.py --quick --steps 1 --warmup 0`: first reason that applies, in this order; "
           "a band of 192 columns or more does not fit the ring (a 256-column ring does not fit the LDS at six waves per SIMD), two predecessors outside the ring would need a second "
           "far operand in the packed row (~3 ms if built)"),
 "raw": raw}
json.dump(out, sys.stdout, indent=1); print()
