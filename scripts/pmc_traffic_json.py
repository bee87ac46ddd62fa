#!/usr/bin/env python3
"""Assemble profiles/rNN_pmc_traffic_*.json from rocprofv3 --pmc passes over scripts/prof_step.py:

    python scripts/pmc_traffic_json.py OUT.json DIR_FETCH DIR_WRITE [DIR_SQ ...]

Per kernel: mean counter value per dispatch; hbm_bytes_per_launch = 2 * FETCH_SIZE KiB (gfx950 reports half of a
wide coalesced stream, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402

NAMES = [("k_fwd<256", "k_fwd"), ("k_fwd_pipe<", "k_fwd"), ("k_bwd<32, 256", "k_bwd_last"), ("k_bwd<256, 256, 2, 4", "k_bwd_hidden"),
         ("k_bwd<256, 256, 2, 2", "k_bwd_hidden_layer1"), ("k_dw0<256", "k_dw_first"), ("k_dw0_8<256", "k_dw_first"),
         # scratch formats 12 / 8 (siren_s8.hip): LAST, hidden (P0 = false) and layer-1 (P0 = true) forms
         ("k_bwd8<32, 256", "k_bwd_last"), ("k_bwd8<256, 256, 2, 4, false, false", "k_bwd_hidden_r2"),
         ("k_bwd8<256, 256, 2, 2, false, false", "k_bwd_hidden"), ("k_bwd8<256, 256, 2, 4, false, true", "k_bwd_hidden_layer1"),
         ("k_bwd8<256, 256, 2, 2, false, true", "k_bwd_hidden_layer1"),
         # round 3: the slot-per-MFMA pipeline of the hidden layers (siren_s8h.hip)
         ("k_bwd8h<", "k_bwd_hidden")]


def main():
    if len(sys.argv) < 3 or sys.argv[1].startswith("-"):
        sys.exit("usage: pmc_traffic_json.py OUT.json PMC_DIR [PMC_DIR ...]")
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    full = {}
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                for pat, short in NAMES:
                    if pat in r["Kernel_Name"]:
                        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                        full[short] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    kernels = {}
    for k, cs in acc.items():
        e = {"kernel": full[k]}
        for c, v in sorted(cs.items()):
            e[c + ("_KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")] = sum(v) / len(v)
        if "FETCH_SIZE_KiB" in e and "WRITE_SIZE_KiB" in e:
            e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE_KiB"] + e["WRITE_SIZE_KiB"]) * 1024
        kernels[k] = e
    json.dump({"kernel_source_sha256": kernel_source_hash(),
               "source": "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_*} (separate passes) -- python3 scripts/prof_step.py 2048 2",
               "note": "per launch at one 4 Mi-pixel chunk (2048x2048, SIREN 256x8): identical launch geometry to bench.py's "
                       "4096x4096 run with 4 Mi-pixel chunks. FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 reports "
                       "half of a wide coalesced stream, MI355X_MICROARCH.md HBM section).",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, e in kernels.items():
        print(k, {c: round(v, 1) if isinstance(v, float) else v for c, v in e.items() if c != "kernel"})


if __name__ == "__main__":
    main()
