"""Cuts rocprofv3 counter CSVs of tools/pmc_workloads.py into per-workload figures and writes
profiles/kernel_counters.json (read by bench.py) plus a markdown summary.

    python tools/pmc_collect.py <dir with one sub-directory per pass> <out json> <out md> <source label>

Each pass directory holds p_counter_collection.csv (one row per dispatch and counter).  FETCH_SIZE
and WRITE_SIZE are in KB as rocprofv3 derives them (TCC_EA0_RDREQ / WRREQ based); the raw values
are kept -- see the note the markdown carries about the gfx950 byte-load correction."""
import collections
import csv
import glob
import json
import os
import sys

src, out_json, out_md, label = sys.argv[1:5]
order = json.load(open(os.path.join(src, "workloads.json")))


def clean(k):
    return k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


# counter -> kernel -> list over dispatches (in dispatch order) of (value, duration ns)
vals = collections.defaultdict(lambda: collections.defaultdict(dict))
for path in sorted(glob.glob(os.path.join(src, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(path)):
        k = clean(r["Kernel_Name"])
        d = int(r["Dispatch_Id"])
        vals[r["Counter_Name"]][k][d] = (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

res = {}
for cname, kernels in vals.items():
    cursor = collections.defaultdict(int)
    for w in order:
        k = w["kernel"]
        ds = sorted(kernels.get(k, {}))
        take = w["launches_per_call"] * w["calls"]
        chunk = ds[cursor[k]:cursor[k] + take]
        cursor[k] += take
        if len(chunk) != take:
            continue
        per_call = [sum(kernels[k][d][0] for d in chunk[i * w["launches_per_call"]:(i + 1) * w["launches_per_call"]])
                    for i in range(w["calls"])]
        dur = [sum(kernels[k][d][1] for d in chunk[i * w["launches_per_call"]:(i + 1) * w["launches_per_call"]])
               for i in range(w["calls"])]
        e = res.setdefault(w["key"], {"kernel": k, "source": label, "counters": {}})
        e["counters"][cname] = sum(per_call) / len(per_call)
        e.setdefault("kernel_ms_under_pmc", {})[cname] = sum(dur) / len(dur) / 1e6

for key, e in res.items():
    c = e["counters"]
    if "SQ_INSTS_VALU" in c:
        e["valu_insts"] = c["SQ_INSTS_VALU"]
    if "FETCH_SIZE" in c:
        e["fetch_bytes"] = c["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in c:
        e["write_bytes"] = c["WRITE_SIZE"] * 1024.0
    if "SQ_INSTS_VALU" in e.get("kernel_ms_under_pmc", {}):
        e["kernel_ms"] = e["kernel_ms_under_pmc"]["SQ_INSTS_VALU"]
json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)

L = ["# Per-workload PMC counters (%s)" % label,
     "Source: rocprofv3 --pmc passes over `python3 tools/pmc_workloads.py` (the kernels, inputs and sizes of bench.py, "
     "one workload after the other), cut per workload by tools/pmc_collect.py.  Values are per CALL "
     "(multi-pass scans: the sum of their launches), mean of %d calls.  FETCH_SIZE / WRITE_SIZE raw (KB -> bytes); "
     "the disparity kernels read with byte loads, for which the guide's gfx950 x2 correction (calibrated on 16 B/lane "
     "streams) does not apply, so raw values are reported.\n" % order[0]["calls"],
     "| workload | kernel | VALU wave-instr | LDS instr | fetch MB | write MB | ms under PMC |", "|---|---|---|---|---|---|---|"]
for key in [w["key"] for w in order]:
    e = res.get(key)
    if not e:
        continue
    c = e["counters"]
    f = lambda n, s=1.0: ("%.4g" % (c[n] * s)) if n in c else "-"
    L.append("| %s | %s | %s | %s | %s | %s | %s |" % (key, e["kernel"], f("SQ_INSTS_VALU"), f("SQ_INSTS_LDS"),
                                                      f("FETCH_SIZE", 1024 / 1e6), f("WRITE_SIZE", 1024 / 1e6),
                                                      "%.4g" % e["kernel_ms"] if "kernel_ms" in e else "-"))
others = sorted({n for e in res.values() for n in e["counters"]} - {"SQ_INSTS_VALU", "SQ_INSTS_LDS", "FETCH_SIZE", "WRITE_SIZE"})
if others:
    L += ["", "Other counters (per call):", "", "| workload | " + " | ".join(others) + " |", "|---|" + "---|" * len(others)]
    for key in [w["key"] for w in order]:
        e = res.get(key)
        if e:
            L.append("| %s | " % key + " | ".join(("%.4g" % e["counters"][n]) if n in e["counters"] else "-" for n in others) + " |")
open(out_md, "w").write("\n".join(L) + "\n")
print("wrote", out_json, out_md, len(res), "workloads")
