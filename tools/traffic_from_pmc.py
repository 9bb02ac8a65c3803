#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: traffic_from_pmc.py <fetch_pass.csv> <write_pass.csv> <workload> <out.json> [kernel substring]
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte read requests of a wide coalesced
stream at 64 bytes, so the read side is doubled (an upper bound for narrow accesses, which are not calibrated).
"""
import csv
import json
import sys


def total(path, counter, kernel):
    s, disp = 0.0, set()
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            s += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return s, len(disp)


def main():
    fetch_csv, write_csv, workload, out = sys.argv[1:5]
    kernel = sys.argv[5] if len(sys.argv) > 5 else "k_wave_trace"
    f, nf = total(fetch_csv, "FETCH_SIZE", kernel)
    w, nw = total(write_csv, "WRITE_SIZE", kernel)
    if not nf or not nw:
        raise SystemExit("kernel not found in the counter files")
    read_b = 2.0 * f * 1024.0 / nf
    write_b = w * 1024.0 / nw
    json.dump({"workload": workload, "kernel": kernel, "launches": nf,
               "fetch_size_kib_per_launch_raw": f / nf, "write_size_kib_per_launch_raw": w / nw,
               "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b,
               "hbm_bytes_per_launch": read_b + write_b,
               "note": "read side = 2 x FETCH_SIZE (gfx950 tallies 128-byte requests at 64 bytes)"}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
