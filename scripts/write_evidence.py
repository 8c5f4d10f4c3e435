#!/usr/bin/env python3
"""profiles/<tag>_snr_ber_hip.{json,md} and profiles/<tag>_run_benchmark_hip{.json,_raw.json,.md} from the tools' outputs under
gpurun_out/ (viterbidecodercpp_amd.tools.run_snr_ber / run_benchmark > gpurun_out/<tag>_snr_ber_hip.json / <tag>_run_benchmark_hip.json
on the GPU box).  usage: write_evidence.py <tag> [round number for the headings]"""
import json, os, shutil, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r6"
ROUND = sys.argv[2] if len(sys.argv) > 2 else TAG.lstrip("r")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = lambda *p: os.path.join(ROOT, *p)
if os.path.exists(g("gpurun_out", f"{TAG}_snr_ber_hip.json")):
    ber = json.load(open(g("gpurun_out", f"{TAG}_snr_ber_hip.json")))
    shutil.copy(g("gpurun_out", f"{TAG}_snr_ber_hip.json"), g("profiles", f"{TAG}_snr_ber_hip.json"))
    lines = ["# BER vs Eb/N0 on one MI355X, round " + ROUND + " (`python -m viterbidecodercpp_amd.tools.run_snr_ber --codes 2 5 7 --decode-types SOFT16 HARD8 --bits-scale 4`)", "",
             "Frames generated (`vit_hip_synth_batch`), decoded (`vit_hip_decode_batch`: this round's kernels -- lane-basis branch-metric ring, K15 table build) and compared",
             "(`vit_hip_count_bit_errors`) on the device; the sweep and its stopping rules are those of `examples/run_snr_ber.cpp:220-400`; JSON record in the reference's",
             "format: `" + TAG + "_snr_ber_hip.json`.  `tests/test_gpu_synth.py::test_ber_point_equals_the_oracle_decoding_the_same_frames` pins that a point's error count is exactly",
             "the CPU checker's on the same frames (green in this round's GPU run), so the GPU curve IS the oracle's curve on these frames.", ""]
    for r in ber:
        pts = ", ".join(f"{e:g} dB {b:.3g}" for e, b in zip(r["EbNo_dB"], r["ber"]))
        lines.append(f"**{r['name']} K={r['K']} R=1/{r['R']} {r['decode_type']}**: {pts}")
        lines.append("")
    lines += ["Against the reference's committed curves (`examples/data_snr_ber_x86.txt`, SCALAR rows :72-80 Voyager SOFT16, :122-130 Voyager HARD8, :242-250 CDMA IS-95A SOFT16,",
              ":482-490 Cassini SOFT16).  They come from an older sweep whose Eb/N0 axis lacks the `+3 dB for a real signal` term of today's `run_snr_ber.cpp:323`, i.e. they sit 3 dB",
              "to the left; different random streams:", "",
              "| code | published (Eb/N0 of the file) | here (Eb/N0 + 3 dB) |", "|---|---|---|"]
    def at(name, dt, e):
        for r in ber:
            if r["name"] == name and r["decode_type"] == dt and e in r["EbNo_dB"]:
                return r["ber"][r["EbNo_dB"].index(e)]
        return float("nan")
    for name, dt, pub in (("Voyager", "SOFT16", [(-1.0, 6.281e-3), (0.0, 4.072e-4), (1.0, 2.188e-5), (2.0, 3.839e-7)]),
                          ("Voyager", "HARD8", [(0.0, 7.019e-3), (1.0, 9.076e-4), (2.0, 6.385e-5), (3.0, 2.687e-6)]),
                          ("CDMA IS-95A", "SOFT16", [(-1.0, 2.212e-3), (0.0, 1.223e-4), (1.0, 1.024e-6)])):
        lines.append(f"| {name} {dt} | " + ", ".join(f"{e:g} dB {b:.3g}" for e, b in pub) + " | " + ", ".join(f"{e + 3:g} dB {at(name, dt, e + 3):.3g}" for e, _ in pub) + " |")
    lines += ["", "(Cassini SOFT16, `:482-490`: the published points lie between -17 and -12 dB on that file's axis, below where this sweep starts (0 dB): no common points; the curve above",
              "is pinned by the oracle on the same frames, like the others.)", ""]
    open(g("profiles", f"{TAG}_snr_ber_hip.md"), "w").write("\n".join(lines))

rb = json.load(open(g("gpurun_out", f"{TAG}_run_benchmark_hip.json")))
slim = [{k: (v if not isinstance(v, list) or k == "G" else {"n": len(v), "median": sorted(v)[len(v) // 2], "min": min(v), "max": max(v)}) for k, v in r.items()} for r in rb]
json.dump(slim, open(g("profiles", f"{TAG}_run_benchmark_hip.json"), "w"), indent=1)
# ... and the tool's own format with the sample lists cut to 64 integer entries: what the reference's examples/parse_benchmark.py reads
# (tests/test_host.py::test_reference_parse_benchmark_reads_the_tools_records feeds it this file)
raw = [{k: ([int(round(x)) for x in v[:64]] if isinstance(v, list) and k != "G" else v) for k, v in r.items()} for r in rb]
json.dump(raw, open(g("profiles", f"{TAG}_run_benchmark_hip_raw.json"), "w"))
lines = ["# viterbidecodercpp_amd.tools.run_benchmark -T 0.3 on one MI355X, round " + ROUND + " (noise-free 2048-bit frames, the reference's protocol)", "",
         "JSON with the reference's field names (`examples/run_benchmark.cpp:297-327`, readable by `examples/parse_benchmark.py`): the tool's stdout (`" + TAG + "_run_benchmark_hip.json` keeps",
         "the sample arrays as count / median / min / max); 24 records (8 codes x 3 decode types), update and chainback timed APART with HIP events as the reference times its two",
         "phases (:272-281).  The end-to-end figures of the same matrix through the pipeline are `" + TAG + "_matrix.json`.", "",
         "| code | decode | frames/batch | update Gsym/s | chainback Gbit/s |", "|---|---|---|---|---|"]
for r in rb:
    med = lambda v: sorted(v)[len(v) // 2]
    lines.append(f"| {r['name']} | {r['decode_type']} | {r['frames']} | {r['total_symbols'] / med(r['update_symbols_ns']):.2f} | {r['total_input_bits'] / med(r['chainback_bits_ns']):.2f} |")
lines += ["", "Round 2 for comparison: `profiles/r2_run_benchmark_hip.md` (superseded).", ""]
open(g("profiles", f"{TAG}_run_benchmark_hip.md"), "w").write("\n".join(lines))
print("written")
