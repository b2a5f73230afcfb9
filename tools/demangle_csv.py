"""Rewrites the kernel-name column of a rocprofv3 csv with demangled names (rocprofv3 leaves every fp16 kernel mangled: its
demangler does not know `DF16_`).   python tools/demangle_csv.py in.csv out.csv"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv, argv = [sys.argv[0], os.devnull, os.devnull], sys.argv   # pmc_traffic runs its command line on import: give it empty inputs


def main(src, dst):
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_traffic_names", os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.py"))
    text = open(spec.origin).read().split("\nfetch, n1 = total_kb")[0]       # the helpers only, not the command-line part
    ns = {"__name__": "pmc_traffic_names", "__file__": spec.origin}
    exec(compile(text, spec.origin, "exec"), ns)
    rows = list(csv.reader(open(src)))
    col = rows[0].index("Name") if "Name" in rows[0] else rows[0].index("Kernel_Name")
    for r in rows[1:]:
        r[col] = ns["demangle"](r[col])
    csv.writer(open(dst, "w", newline=""), quoting=csv.QUOTE_NONNUMERIC).writerows(rows)


if __name__ == "__main__":
    main(argv[1], argv[2])
