"""python profiles/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [skip_launches [iterations [drop_last]]]

iterations: lattice passes the profiled command made in all (bench.py: warmup + steps + the 10 passes of its "alone" measurement).
With it the collide kernel's bytes are summed over ALL its launches and divided by that count (bytes per lattice pass, however
many launches a pass is made of); without it: average per launch after the first skip_launches.

HBM traffic per launch of the kernels of bench.py from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
counter values are KiB per dispatch).  gfx950 correction as prescribed in MI355X_MICROARCH.md (HBM section) and
calibrated in profiles/r01_pmc_collide_traffic.md: FETCH_SIZE reports half of the bytes of a streamed read (x2),
WRITE_SIZE is exact."""
import csv
import hashlib
import json
import os
import sys

NODES = 256 ** 3


def per_kernel(path, counter, skip, total=False, drop_last=0):
    acc = {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "").strip()
        if "anonymous" in row["Kernel_Name"] and not name:
            name = row["Kernel_Name"].split("::")[1].split("(")[0]
        acc.setdefault(name, []).append(float(row["Counter_Value"]))
    if total:
        return {k: sum(v) for k, v in acc.items()}
    cut = lambda k, v: v[skip:len(v) - drop_last] if ("collide_stream_kernel" in k and drop_last) else v[skip:]
    return {k: sum(cut(k, v)) / max(1, len(cut(k, v))) for k, v in acc.items() if len(v) > skip + drop_last}


def main():
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    iterations = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    drop_last = int(sys.argv[6]) if len(sys.argv) > 6 else 0     # bench.py ends with 10 collide launches alone (no spread before them)
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE", skip, drop_last=drop_last)
    write = per_kernel(sys.argv[2], "WRITE_SIZE", skip, drop_last=drop_last)
    key = [k for k in fetch if "collide_stream_kernel" in k][0]
    if iterations:   # bytes per lattice pass = all launches of the kernel together / passes made
        fetch[key] = per_kernel(sys.argv[1], "FETCH_SIZE", 0, True)[key] / iterations
        write[key] = per_kernel(sys.argv[2], "WRITE_SIZE", 0, True)[key] / iterations
    total = fetch[key] * 1024 * 2 + write[key] * 1024
    out = {
        "kernel": "collide_stream_kernel",
        "workload": "pipe 256x256x256, R=127, 1937 RBC (bench.py default)",
        "nodes": NODES,
        # build of the kernel these counters belong to (= hc_build_tag(): SHA-256 of csrc/lattice.hip, 16 hex digits);
        # bench.py quotes the figure only next to timings of the same build
        "kernel_tag": hashlib.sha256(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hemocell_amd", "csrc", "lattice.hip"), "rb").read()).hexdigest()[:16],
        "per": "iteration (all collide launches of one lattice pass together)" if iterations else "launch",
        "FETCH_SIZE_KiB_avg": fetch[key], "WRITE_SIZE_KiB_avg": write[key],
        "correction": "gfx950: FETCH_SIZE counts 1/2 of streamed read bytes (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE exact; "
                      "calibrated on the all-fluid box (177/176 B per node, profiles/r01_pmc_collide_traffic.md)",
        "hbm_bytes_per_launch": total, "hbm_bytes_per_node": total / NODES,
        "other_kernels_KiB": {k: {"FETCH_SIZE": fetch[k], "WRITE_SIZE": write.get(k)} for k in fetch
                              if k != key and ("ibm_" in k or "advance" in k or "mechanics" in k)},
    }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
