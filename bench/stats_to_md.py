"""rocprofv3 --kernel-trace --stats CSV -> the markdown table kept under profiles/.

  python bench/stats_to_md.py gpurun_out/prof_b1024 "title" > profiles/NN_*.md
"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)( \[clone.*)?$", "", name)
    return name[:90]


def main():
    d, title = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_stats.csv under %s" % d)
    rows = list(csv.DictReader(open(files[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# %s\n" % title)
    print("| kernel | calls | total ms | % | avg us |\n|---|---|---|---|---|")
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:int(os.environ.get("TOP", "40"))]:
        t = float(r["TotalDurationNs"])
        print("| `%s` | %s | %.2f | %.1f | %.1f |" % (short(r["Name"]), r["Calls"], t / 1e6, 100 * t / tot,
                                                     t / 1e3 / int(r["Calls"])))
    print("\nsum of kernel time: %.1f ms over %d kernels" % (tot / 1e6, len(rows)))


if __name__ == "__main__":
    main()
