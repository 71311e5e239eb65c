"""Turn two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) into profiles/*gemm_hbm_traffic.json.

    python tools/hbm_traffic.py gpurun_out/pmc_h_fetch gpurun_out/pmc_h_write profiles/r01_h_gemm_hbm_traffic.json

Per-launch average over every dispatch whose kernel name contains `gemm_nt_f32`.  Units and the gfx950 correction follow
/opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KiB; FETCH_SIZE counts 128-byte
requests as 64 bytes on gfx950 and is doubled; WRITE_SIZE is taken as read.
"""
import csv
import glob
import json
import sys


def per_launch(directory, counter, match):
    total, n = 0.0, 0
    for path in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter and match in row["Kernel_Name"]:
                    total += float(row["Counter_Value"])
                    n += 1
    if n == 0:
        raise SystemExit("no %s rows for %s under %s" % (counter, match, directory))
    return total / n, n


def avg_duration_us(directory, match):
    """average dispatch duration of the matching kernels from the kernel trace of the same (counter) run"""
    tot, n = 0.0, 0
    for path in glob.glob(directory + "/**/*kernel_trace.csv", recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if match in row["Kernel_Name"]:
                    tot += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3
                    n += 1
    return (tot / n) if n else None


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    match = sys.argv[4] if len(sys.argv) > 4 else "gemm_nt_f32"
    bench_args = sys.argv[5] if len(sys.argv) > 5 else "--steps 2 --warmup 1 --no-cpu --no-secondary"
    fetch_kb, n = per_launch(fetch_dir, "FETCH_SIZE", match)
    write_kb, _ = per_launch(write_dir, "WRITE_SIZE", match)
    dur = avg_duration_us(fetch_dir, match)
    doc = {
        "avg_duration_us_in_counter_run": dur,
        "hbm_gbs_in_counter_run": ((2.0 * fetch_kb + write_kb) * 1024.0 / (dur * 1e-6) / 1e9) if dur else None,
        "kernel": match,
        "launches": n,
        "fetch_size_kb_per_launch": fetch_kb,
        "write_size_kb_per_launch": write_kb,
        "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
        "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md HBM section); "
                      "WRITE_SIZE as read; the counter is fabric-side: Infinity-Cache hits (activation re-reads) are included.  Both statements "
                      "calibrated on this chip for this kernel's request type (global_load_lds_dwordx4) and for per-XCD re-reads of one buffer: "
                      "profiles/r06_c_fetch_size_calibration.txt (tools/fetch_calib.hip)",
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py " + bench_args,
    }
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
