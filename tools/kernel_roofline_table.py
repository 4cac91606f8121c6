"""Per-kernel table of one configuration from committed records: average duration (rocprofv3 --kernel-trace --stats CSV) and HBM
traffic per launch (separate --pmc passes; KiB counters; gfx950's FETCH_SIZE reports half of wide coalesced reads, so reads
count twice — the correction bench.py applies, MI355X_MICROARCH.md "HBM") -> GB/s moved and the fraction of the 8 TB/s peak.
  python tools/kernel_roofline_table.py profiles/r04_bench_c3_kernel_stats.csv profiles/r04_pmc_c3 [min_calls]
Measurement helper, not product code."""
import csv, sys

stats, pmc, min_calls = sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1000
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(", ", "; ")


def pmc_rows(path, col):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["kernel"].replace("void ", "")] = float(r[col])
    return out


fetch, write = pmc_rows(pmc + "_FETCH_SIZE.csv", "mean_FETCH_SIZE"), pmc_rows(pmc + "_WRITE_SIZE.csv", "mean_WRITE_SIZE")
print("%-32s %8s %10s %10s %10s %9s %7s" % ("kernel", "calls", "avg us", "read MB", "write MB", "GB/s", "of 8T"))
for r in csv.DictReader(open(stats)):
    name = short(r["Name"])
    if not name.startswith("k_") or int(r["Calls"]) < min_calls:
        continue
    us = float(r["AverageNs"]) / 1e3
    f, w = fetch.get(name), write.get(name)
    if f is None or w is None:
        print("%-32s %8s %10.2f %10s %10s" % (name, r["Calls"], us, "-", "-"))
        continue
    rd, wr = 2.0 * f * 1024 / 1e6, w * 1024 / 1e6
    gbs = (rd + wr) * 1e6 / (us * 1e-6) / 1e9
    print("%-32s %8s %10.2f %10.1f %10.1f %9.0f %7.3f" % (name, r["Calls"], us, rd, wr, gbs, gbs / 8000.0))
