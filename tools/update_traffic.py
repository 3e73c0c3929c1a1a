#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of one profile directory.

    tools/update_traffic.py profiles/r5 [--table profiles/traffic.json] [--stamp <sources_sha256>]

Reads <dir>/pmc_fetch.summary.txt (FETCH_SIZE, KiB) and <dir>/pmc_write.summary.txt (WRITE_SIZE, KiB; TCC_HIT_sum,
TCC_MISS_sum) - the files tools/profile_bench.sh + tools/rocpd_summary.py write - and records, per kernel,
HBM-side bytes per launch = FETCH_SIZE x 1024 x 2 (gfx950 wide-read under-count, MI355X_MICROARCH.md HBM section) +
WRITE_SIZE x 1024, the L2 hit rate, the profile directory and the sha256 of the sources the profiled library was built
from (<dir>/library.stamp.json, copied there by profile_bench.sh; --stamp overrides).  bench.py reports roofline.traffic
only when that hash equals the loaded library's.
ONE stamp per table (round 5): a table that already holds records of ANOTHER library is not merged into - its records are
dropped (and listed) and the table starts again from this profile, so that no line of traffic.json can belong to a library
other than the one named in the same directory's library.stamp.json."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(r"^(?:void )?am::(\w+).*?\s+(FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCC_MISS_sum)\s+(\d+)\s+([0-9.]+)\s*$")


def counters(path):
    out = {}
    with open(path) as f:
        for line in f:
            m = LINE.match(line.rstrip("\n"))
            if m:
                out.setdefault(m.group(1), {})[m.group(2)] = float(m.group(4))
    return out


def main():
    args = sys.argv[1:]
    stamp = None
    if "--stamp" in args:
        i = args.index("--stamp")
        stamp = args[i + 1]
        del args[i:i + 2]
    table_arg = None
    if "--table" in args:
        i = args.index("--table")
        table_arg = args[i + 1]
        del args[i:i + 2]
    prof = args[0].rstrip("/")
    if stamp is None:
        with open(os.path.join(prof, "library.stamp.json")) as f:
            stamp = json.load(f)["sources_sha256"]
    fetch = counters(os.path.join(prof, "pmc_fetch.summary.txt"))
    write = counters(os.path.join(prof, "pmc_write.summary.txt"))
    table_path = table_arg or os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(table_path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        table = {}
    stale = {name: rec for name, rec in table.get("kernels", {}).items() if rec.get("sources_sha256") != stamp}
    if stale:
        print(f"dropping {len(stale)} record(s) of other libraries: " + ", ".join(f"{n} ({r.get('sources_sha256', '?')[:12]})" for n, r in sorted(stale.items())))
        table["kernels"] = {name: rec for name, rec in table.get("kernels", {}).items() if name not in stale}
    table["sources_sha256"] = stamp
    kernels = table.setdefault("kernels", {})
    for name in sorted(set(fetch) & set(write)):
        if "FETCH_SIZE" not in fetch[name] or "WRITE_SIZE" not in write[name]:
            continue
        hit, miss = write[name].get("TCC_HIT_sum"), write[name].get("TCC_MISS_sum")
        kernels[name] = {
            "bytes_per_launch": int(round(fetch[name]["FETCH_SIZE"] * 1024 * 2 + write[name]["WRITE_SIZE"] * 1024)),
            "l2_hit": None if not hit or miss is None else round(hit / (hit + miss), 4),
            "profile": os.path.relpath(os.path.abspath(prof), ROOT), "sources_sha256": stamp}
        print(name, kernels[name])
    table["source"] = ("per kernel: FETCH_SIZE (KiB) x 1024 x 2 (gfx950 wide-read under-count, MI355X_MICROARCH.md HBM section) + "
                       "WRITE_SIZE (KiB) x 1024 per dispatch, separate --pmc passes of bench.py at 2 x 100k x 512, k = 5 "
                       "(tools/profile_bench.sh); l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS); the counter includes Infinity-Cache "
                       "hits.  A record belongs to the library whose sources hash to sources_sha256; bench.py reports it as "
                       "roofline.traffic only for that library (tools/update_traffic.py).")
    table.pop("bytes_per_launch", None)
    with open(table_path, "w") as f:
        json.dump(table, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
