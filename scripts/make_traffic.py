#!/usr/bin/env python3
"""Regenerate profiles/traffic.json from the rocprofv3 summaries it names.

bench.py reads `roofline.traffic` (HBM bytes per update launch, from the FETCH_SIZE / WRITE_SIZE counter passes) and the VALU
instruction count of `roofline_valu` from profiles/traffic.json.  Every entry of that file is DERIVED: its `source` is a
`profiles/*_summary.md` written by scripts/summarize_profile.py from one scripts/profile.sh run (separate --pmc passes, per the
guide's HBM / rocprofv3 section), and this script is the only writer -- the numbers follow from the "PMC, per launch" tables of
the summary:

    HBM bytes = FETCH_SIZE x 1024 x 2 (gfx950: the counter's unit is 64 B where the tool assumes 32) + WRITE_SIZE x 1024
    VALU      = SQ_INSTS_VALU

    python scripts/make_traffic.py            rewrite profiles/traffic.json
    python scripts/make_traffic.py --check    exit 1 if the committed file differs from what the summaries give
                                              (tests/test_host.py::test_traffic_json_follows_from_its_sources)

A new profile: add its (key, summary) pair to ENTRIES.  key = "<code name>|<decode type>|<frames per launch>x<bits>|<plan>".
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")

ENTRIES = [
    ("Voyager|SOFT16|65536x8192|reg", "r6_k7_default_summary.md"),
    ("Voyager|SOFT16|16384x8192|lds", "r1_k7_lds_summary.md"),
    ("Cassini|SOFT16|1024x8192|lds2", "r1_k15_lds2_summary.md"),
    ("CDMA IS-95A|SOFT16|65536x8192|reg", "r6_k9_summary.md"),
    ("Cassini|SOFT16|4096x8192|lds2", "r6_k15_summary.md"),
    ("Voyager|HARD8|32768x8192|reg", "r6_hard8_summary.md"),
]


def parse_summary(path):
    """{kernel name: {counter: value per launch}} from the '## PMC, per launch: `kernel`' sections"""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"## PMC, per launch: `(.+)`", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if line.startswith("## "):
            cur = None
            continue
        m = re.match(r"\| ([A-Za-z0-9_]+) \| ([-+0-9.eE]+) \|", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return out


def entry(summary):
    kernels = parse_summary(os.path.join(PROFILES, summary))
    detail = {}
    for name, c in sorted(kernels.items()):
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        rd_raw = c["FETCH_SIZE"] * 1024.0
        wr = c["WRITE_SIZE"] * 1024.0
        detail[name] = {"read_raw": rd_raw, "read_corrected": 2.0 * rd_raw, "write": wr, "total": 2.0 * rd_raw + wr}
    upd = [k for k in detail if "update_kernel" in k]
    cb = [k for k in detail if "chainback_kernel" in k or "chainback_alt_kernel" in k]
    e = {"source": f"profiles/{summary}",
         "update_kernel_hbm_bytes_per_launch": detail[upd[0]]["total"] if upd else None,
         "chainback_kernel_hbm_bytes_per_launch": detail[cb[0]]["total"] if cb else None}
    if upd and "SQ_INSTS_VALU" in kernels[upd[0]]:
        e["update_kernel_valu_insts_per_launch"] = kernels[upd[0]]["SQ_INSTS_VALU"]
    if cb and "SQ_INSTS_VALU" in kernels[cb[0]]:
        e["chainback_kernel_valu_insts_per_launch"] = kernels[cb[0]]["SQ_INSTS_VALU"]
    e["detail"] = detail
    return e


def build():
    return {key: entry(summary) for key, summary in ENTRIES}


def differences(a, b, path=""):
    if isinstance(a, dict) and isinstance(b, dict):
        out = []
        for k in sorted(set(a) | set(b)):
            if k not in a or k not in b:
                out.append(f"{path}/{k}: only in {'the file' if k in a else 'the summaries'}")
            else:
                out += differences(a[k], b[k], f"{path}/{k}")
        return out
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        return [] if abs(a - b) <= 1e-5 * max(abs(a), abs(b), 1.0) else [f"{path}: file {a!r} != summaries {b!r}"]
    return [] if a == b else [f"{path}: file {a!r} != summaries {b!r}"]


def main():
    want = build()
    tpath = os.path.join(PROFILES, "traffic.json")
    if "--check" in sys.argv[1:]:
        have = json.load(open(tpath))
        diff = differences(have, want)
        for d in diff:
            print(d)
        return 1 if diff else 0
    with open(tpath, "w") as f:
        json.dump(want, f, indent=1)
        f.write("\n")
    print(f"wrote {tpath}: {len(want)} entries")
    return 0


if __name__ == "__main__":
    sys.exit(main())
