"""Read the counter CSVs of `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./tools/fetch_calib` (and a WRITE_SIZE pass if given) and print,
per kernel, the counter (KiB -> bytes) beside the bytes the kernel is known to move: the ratio is the correction factor for that
request type.      python tools/fetch_calib.py <dir> > profiles/r06_c_fetch_size_calibration.txt"""
import csv, glob, sys
known = {"k_stream_vgpr": (512 << 20, "512 MiB once, global_load_dwordx4"),
         "k_stream_lds": (512 << 20, "512 MiB once, global_load_lds_dwordx4"),
         "k_xcd_reread_lds": (80 << 20, "10 MiB unique x 8 XCDs = 80 MiB requested, global_load_lds_dwordx4")}
rows = {}
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        for k in known:
            if r["Kernel_Name"].startswith(k):
                rows.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(rows.items()):
    b, what = known[k]
    for i, x in enumerate(v):
        print("%-18s %-11s launch %d: counter %12.0f KiB = %8.1f MiB   known %6.0f MiB (%s)   counter / known = %.3f"
              % (k, c, i, x, x / 1024, b / 2**20, what, x * 1024 / b))
