"""Fold rocprofv3 --pmc passes (one directory per counter set, produced with bench/pmc_kernels.py)
into one per-kernel table with derived metrics.

  python bench/summarize_pmc.py gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 > profiles/NN_pmc.md

Per kernel (name shortened, keyed with its grid size so that the same template on two layer
shapes stays apart) the LAST dispatch of each pass is used (the earlier ones warm caches/clocks).
Derived columns:
  us            kernel duration from the pass's own timestamps
  mfma_busy%    SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs): share of all matrix-pipe
                cycles of the chip that issued MFMA work (== achieved / peak dense bf16 rate)
  clk_GHz       GRBM_GUI_ACTIVE / 8 XCDs / duration of the pass that counted it: the clock the chip held
                under this kernel (profiled passes clock a few % lower than plain runs)
  mfma@clk%     the same MFMA cycles against the cycles the chip actually ran (duration x clk_GHz x 1024
                SIMDs): how busy the matrix pipes were, independent of the power-limited clock
  lds_conf%     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE    (share of LDS cycles lost to conflicts)
  wait%         SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES           (wave-cycles stalled on a waitcnt)
  l2_hit%       TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  hbm_rd_MB / hbm_wr_MB   FETCH_SIZE / WRITE_SIZE (KiB units as reported by rocprofv3)
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)
    return name[:64]


def load(d):
    out = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if k.startswith("void at::") or "at::native" in k or "rccl" in k.lower():
                continue
            key = (short(k), int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            e = out.setdefault(key, {})
            did = int(r["Dispatch_Id"])
            if did >= e.get("_did", -1):
                if did > e.get("_did", -1):
                    e.clear()
                    e["_did"] = did
                e[r["Counter_Name"]] = float(r["Counter_Value"])
                e["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                e["_regs"] = "%s+%s" % (r["VGPR_Count"], r["Accum_VGPR_Count"])
                e["_lds"] = int(r["LDS_Block_Size"])
    return out


def main(dirs):
    merged = collections.OrderedDict()
    for d in dirs:
        for key, e in load(d).items():
            m = merged.setdefault(key, {})
            for k, v in e.items():
                if k == "_us":
                    m.setdefault("_us_all", []).append(v)
                    if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                        m["_us_pass"] = v
                    if "GRBM_GUI_ACTIVE" in e:
                        m["_us_grbm"] = v
                else:
                    m[k] = v

    def ratio(m, a, b, scale=100.0):
        if a in m and b in m and m[b] > 0:
            return "%.1f" % (scale * m[a] / m[b])
        return "-"

    print("| kernel | grid x wg | regs (v+a) | LDS B | us | mfma_busy% | clk_GHz | mfma@clk% | lds_conf% | wait% | l2_hit% | hbm_rd_MB | hbm_wr_MB |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for (name, grid, wg), m in merged.items():
        us = min(m.get("_us_all", [0.0]))
        hit = "-"
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
            hit = "%.1f" % (100.0 * m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]))
        rd = "%.1f" % (m["FETCH_SIZE"] / 1024.0) if "FETCH_SIZE" in m else "-"
        wr = "%.1f" % (m["WRITE_SIZE"] / 1024.0) if "WRITE_SIZE" in m else "-"
        clk = None
        if "GRBM_GUI_ACTIVE" in m and m.get("_us_grbm", 0) > 50.0:      # the quotient reads high on short dispatches
            clk = m["GRBM_GUI_ACTIVE"] / 8.0 / (m["_us_grbm"] * 1e3)
        at_clk = "-"
        if clk and "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("_us_pass", 0) > 0 and m["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
            at_clk = "%.1f" % (100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["_us_pass"] * clk * 1e3 * 1024.0))
        print("| `%s` | %dx%d | %s | %d | %.1f | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            name, grid // max(wg, 1), wg, m.get("_regs", "-"), m.get("_lds", 0), us,
            ("%.1f" % (100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["_us_pass"] * 2400.0 * 1024.0))
             if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("_us_pass", 0) > 0 else "-"),
            ("%.2f" % clk) if clk else "-", at_clk,
            ratio(m, "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
            ratio(m, "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"), hit, rd, wr))


if __name__ == "__main__":
    main(sys.argv[1:])
