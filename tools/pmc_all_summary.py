"""Per-kernel summary (markdown + raw json) of a rocprofv3 --pmc SQ_* counter_collection.csv of bench.py.
Usage: python tools/pmc_all_summary.py <counter_collection.csv> <out_prefix>"""
import collections
import csv
import json
import sys

src, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"]
    if "at::native" in k or "rocclr" in k:
        continue
    k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
raw = {k: {c: {"mean": sum(v) / len(v), "min": min(v), "max": max(v), "launches": len(v)} for c, v in d.items()}
       for k, d in acc.items()}
json.dump(raw, open(out + ".json", "w"), indent=1)


def g(k, c, w="mean"):
    return raw[k][c][w]


L = ["# rocprofv3 --pmc SQ_* per launch, all kernels of bench.py at the end of round 1 (raw: %s.json)" % out.split("/")[-1],
     "Command: rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES",
     "         SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu",
     "(counters only, no other trace domain; values are per launch; where a kernel runs several workloads in bench.py the",
     "launch named in the row is picked by its instruction count)\n",
     "| kernel | workload | SQ_INSTS_VALU | derived |", "|---|---|---|---|"]
units = 2880 * 1988 * 256 - 1988 * (255 * 256 // 2)
v = g("disp_band_kernel<4, false>", "SQ_INSTS_VALU", "min")
L.append("| disp_band_kernel<4,false> | 2880x1988, 9x9 SAD, D=256 (%.3g units; smallest launch) | %.3g | %.2f VALU wave-instr per 64 units |" % (units, v, v / (units / 64)))
v = g("disp_band_kernel<4, true>", "SQ_INSTS_VALU")
L.append("| disp_band_kernel<4,true> | same pair, 9x9 SSD | %.3g | %.2f per 64 units |" % (v, v / (units / 64)))
v = g("disp_band_kernel<0, false>", "SQ_INSTS_VALU")
L.append("| disp_band_kernel<0,false> | same pair, 1x1, D=256 | %.3g | %.2f per 64 units |" % (v, v / (units / 64)))
for r_, name in ((5, "11x11"), (6, "13x13"), (7, "15x15")):
    k = "disp_band_kernel<%d, false>" % r_
    if k in raw:
        v = g(k, "SQ_INSTS_VALU")
        L.append("| disp_band_kernel<%d,false> | same pair, %s SAD | %.3g | %.2f per 64 units |" % (r_, name, v, v / (units / 64)))
u = 2880 * 2880 * 1988
v = g("disp_ref_kernel", "SQ_INSTS_VALU")
L.append("| disp_ref_kernel | same pair, whole-row search (%.3g units) | %.3g | %.2f per 64 units |" % (u, v, v / (u / 64)))
px = 8 * 1920 * 1080
v = g("rectify_kernel<true>", "SQ_INSTS_VALU")
L.append("| rectify_kernel<true> | 8 x 1920x1080 | %.3g | %.0f lane-instr per pixel |" % (v, v * 64 / px))
n = 200000
mf = g("match_mfma_bf16x1_kernel", "SQ_VALU_MFMA_BUSY_CYCLES")
v = g("match_mfma_bf16x1_kernel", "SQ_INSTS_VALU")
L.append("| match_mfma_bf16x1_kernel | 200k x 200k x 128 | %.3g | SQ_VALU_MFMA_BUSY_CYCLES %.3g = %.3g MFMAs x 32 cycles (2 n^2 K / 32768 = %.3g: one bf16 product per fp32 product); %.1f VALU per MFMA; LDS instr %.3g |"
         % (v, mf, mf / 32, 2 * n * n * 128 / 32768, v / (mf / 32), g("match_mfma_bf16x1_kernel", "SQ_INSTS_LDS")))
L.append("| match_replay_kernel | 200k queries | %.3g | exact fp32 replay of the candidate lists |" % g("match_replay_kernel", "SQ_INSTS_VALU"))
ev = 200 * 200000
v = g("ransac_error_kernel", "SQ_INSTS_VALU")
L.append("| ransac_error_kernel | 200 hypotheses x 200k pairs | %.3g | %.0f VALU wave-instr per 64 evaluations (22 IEEE divides, 4 square roots): VALU bound |" % (v, v / (ev / 64)))
L.append("| ransac_reduce_kernel | same, 400 waves | %.3g | serial chain: SQ_WAVE_CYCLES %.3g over 400 waves, SQ_BUSY_CYCLES %.3g |"
         % (g("ransac_reduce_kernel", "SQ_INSTS_VALU"), g("ransac_reduce_kernel", "SQ_WAVE_CYCLES"), g("ransac_reduce_kernel", "SQ_BUSY_CYCLES")))
a, b = g("fast_flags_kernel", "SQ_INSTS_VALU"), g("fast_scatter_kernel", "SQ_INSTS_VALU")
L.append("| fast_flags_kernel + fast_scatter_kernel | 2880x1988 | %.3g + %.3g | %.1f lane-instr per pixel |" % (a, b, (a + b) * 64 / (2880 * 1988)))
L.append("| nms_kernel / rank_kernel | 57 578 corners, O(n^2) tiles | %.3g / %.3g | |" % (g("nms_kernel", "SQ_INSTS_VALU"), g("rank_kernel", "SQ_INSTS_VALU")))
v = g("sift_kernel", "SQ_INSTS_VALU")
L.append("| sift_kernel | 12 238 features, sixteen lanes each | %.3g | %.0f lane-instr per feature (incl. the lanes' redundant chains) |" % (v, v * 64 / 12238))
open(out + ".md", "w").write("\n".join(L) + "\n")
print("\n".join(L))
