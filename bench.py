#!/usr/bin/env python3
"""bench.py -- headline benchmark of the stereo hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): disparity Mpix*disp/s = pixels x candidate positions
evaluated per second.  A step is one pass of the dense disparity scan over one
synthetic rectified pair of BASELINE config 2 (2880x1988, 256 disparities) per
GPU; inputs are resident in HBM before the timed region.  Ranks each own their
pair (pair sharding, no data-path collective), so scaling is weak.

The JSON line also carries the roofline of the dominant kernel, the CPU
baseline (oracle restatement on a bounded sample, rank 0 at N=1 only) and an
"extra" object with the other kernels of the path (reference-mode scan,
rectification warp, FAST-12, descriptor matching) and their CPU timings.

north_star's multi-GPU workloads run at every N (backend nccl = RCCL):
  config 3  one 4K pair, 512 disparities: each rank scans its row band (+ halo rows from its own
            copy of the pair), bands gathered to rank 0 by point-to-point sends   (strong scaling)
  config 4  1024 1080p pairs, 192 disparities: 1024 / N pairs per rank, timed with and without
            the gather of the maps to rank 0                                      (strong scaling)
  config 5  200k x 200k matching: list1's queries sharded, list2 replicated       (strong scaling)
Rank 0 checks every gathered result against the single-GPU result of the same input.  The other
extras (secondary kernels, CPU timings) run at N = 1 only.

Counters that cannot be measured inside this process (VALU instructions and HBM bytes per
launch) are read from profiles/kernel_counters.json, which tools/pmc_collect.py writes from
rocprofv3 --pmc passes over this same command; every figure derived from them names that file.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X constants (/opt/skills/guides/MI355X_MICROARCH.md)
CUS = 256
CLOCK_HZ = 2.4e9
# Integer VALU roofline: 256 CU x 64 lanes/clk x 2.4 GHz = 3.93e13 lane-ops/s.
# tools/ubench.hip measures 3.5e13 (3-operand integer ops such as v_sad_u32 / v_min3_u32,
# 8 waves per SIMD, clock throttled to ~1.5 GHz under that load) to 5.2e13 (v_add_u32).
VALU_LANEOPS = CUS * 64 * CLOCK_HZ
# What the chip measurably sustains, 8 waves per SIMD, all CUs (tools/ubench.hip,
# profiles/r01_ubench_instruction_rates.txt; the clock drops to 1.2 - 1.5 GHz under these loads):
#   2-operand VALU (v_add_u32 806, v_sub_u32 833, v_fma_f32 794 G wave-instr/s): the fastest
#     instruction streams measured -- the issue CEILING, no kernel can exceed it;
#   3-operand VALU (v_sad_u32 548, v_sad_u8 553, v_min3_u32 554, v_add3_u32 518): what the
#     disparity kernels are mostly made of; a kernel that mixes in 2-operand instructions
#     (DPP subtractions, compares) can run above this rate.
ISSUE_CEILING_WAVE_INSTR = 833e9
ISSUE_RATE_3OP_WAVE_INSTR = 550e9
HBM_PEAK_GBS = 8000.0
MFMA_F32_TFLOPS = 157.3
MFMA_BF16_TFLOPS = 2500.0

W, H, D, WIN = 2880, 1988, 256, 9           # BASELINE config 2, 9x9 SAD window
SETTLE_MS = 150.0                           # untimed launches before the warm-up steps (clock ramp of a fresh box)
OPS_PER_UNIT_WINDOWED = 7                   # sliding column/row sums
OPS_PER_UNIT_WIN1 = 3                       # subtract, abs, compare-select


def event_time_ms(torch, fn, steps, warmup):
    """Average ms per call of fn over `steps` calls, HIP events on the current stream."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def load_counters():
    """profiles/kernel_counters.json: per kernel and workload, VALU wave-instructions and HBM
    bytes per launch from rocprofv3 --pmc passes (tools/pmc_collect.py).  {} if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "kernel_counters.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def valu_roofline(counters, key, units, ops_per_unit, ms, label=None):
    """Roofline object of a VALU-bound kernel: the SURVEY 8(d) op model against the nominal
    integer peak, and -- when the PMC instruction count of this workload is on file -- the
    instructions actually issued against the measured issue ceiling."""
    achieved = units * ops_per_unit / (ms * 1e-3)
    r = {"bound": "valu", "achieved": achieved / 1e12, "peak": VALU_LANEOPS / 1e12, "unit": "Tlane-op/s",
         "frac": achieved / VALU_LANEOPS, "kernel_ms": ms,
         "model": "%g integer lane-ops per unit (SURVEY 8d) x %.4g units; peak = %d CU x 64 lanes/clk x %.1f GHz"
                  % (ops_per_unit, units, CUS, CLOCK_HZ / 1e9)}
    if label:
        r["kernel"] = label
    c = counters.get(key)
    if c and c.get("valu_insts"):
        rate = c["valu_insts"] / (ms * 1e-3)
        r["issue"] = {"valu_wave_instr_per_launch": c["valu_insts"],
                      "wave_instr_per_64_units": c["valu_insts"] / (units / 64.0),
                      "achieved_Ginstr_per_s": rate / 1e9, "ceiling_Ginstr_per_s": ISSUE_CEILING_WAVE_INSTR / 1e9,
                      "frac": rate / ISSUE_CEILING_WAVE_INSTR,
                      "three_operand_stream_Ginstr_per_s": ISSUE_RATE_3OP_WAVE_INSTR / 1e9,
                      "vs_three_operand_stream": rate / ISSUE_RATE_3OP_WAVE_INSTR,
                      "note": "frac is against the fastest VALU stream measured on this chip (2-operand, 833 G/s); "
                              "vs_three_operand_stream is against a pure v_sad / v_min3 stream (550 G/s) and can "
                              "pass 1 for kernels that mix in 2-operand instructions",
                      "source": "profiles/kernel_counters.json (%s) / profiles/r01_ubench_instruction_rates.txt"
                                % c.get("source", "?")}
    return r


def units_evaluated(w, h, d):
    """Candidate positions actually evaluated: pixel x has min(x+1, d) candidates."""
    d = min(d, w)
    per_row = d * (w - d) + d * (d + 1) // 2
    return per_row * h


def config1_chain(ctx, orc, api, synth, reps=5):
    """BASELINE config 1 (main.cpp:188-380 on a quarter-res pair) stage by stage: the host-pointer
    calls of the library -- what the drop-in delivers, PCIe included -- beside the oracle on one
    host thread, same inputs, outputs compared.  Stand-ins as in tests/test_gpu_configs.py: a
    synthetic 741x500 pair, box-blur / central-difference gradient images instead of OpenCV's,
    seeded K R K^-1 homographies instead of ComputeRectificationRotations, and the 200 RANSAC
    hypotheses drawn once on the host (the eight-point solves stay the reference's host code)."""
    from stereo_amd import ransac as rsc
    h, w = 500, 741
    imgs = synth.rectified_pair(1001, h, w, 48, zero_frac=0.0)
    grads = [synth.gradient_stand_ins(im) for im in imgs]
    kern = ctx.gaussian_kernel()
    f = 3979.911 / 4
    K = np.array([[f, 0, 311.0], [0, f, 255.0], [0, 0, 1]], np.float32)
    hinv = [ctx.invert_homography(synth.homography(5, w, h, deg_y=0.8, deg_x=0.2)),
            ctx.invert_homography(synth.homography(6, w, h, deg_y=-0.6, deg_x=0.3))]

    def best_ms(fn, n=reps):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            r = fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts), r

    def once_ms(fn):
        t0 = time.perf_counter()
        r = fn()
        return (time.perf_counter() - t0) * 1e3, r

    gpu, cpu, same = {}, {}, {}
    # features of both images
    g_fast = [None, None]; g_sc = [None, None]; g_desc = [None, None]
    c_fast = [None, None]; c_sc = [None, None]; c_desc = [None, None]
    gpu["fast12"], _ = best_ms(lambda: [g_fast.__setitem__(k, ctx.fast_detect(imgs[k])[0]) for k in (0, 1)])
    cpu["fast12"], _ = once_ms(lambda: [c_fast.__setitem__(k, orc.fast_features(imgs[k])) for k in (0, 1)])
    same["fast12"] = all(np.array_equal(g_fast[k], np.asarray(c_fast[k], np.float32).reshape(-1, 2)) for k in (0, 1))
    gpu["score_nms"], _ = best_ms(lambda: [g_sc.__setitem__(k, ctx.score_cluster(grads[k][0], grads[k][1], g_fast[k], kern,
                                                                                 api.ST_THRESH, api.NMS_WINDOW)) for k in (0, 1)])
    cpu["score_nms"], _ = once_ms(lambda: [c_sc.__setitem__(k, orc.score_and_cluster(grads[k][0], grads[k][1], c_fast[k], kern,
                                                                                     api.ST_THRESH, api.NMS_WINDOW)) for k in (0, 1)])
    same["score_nms"] = all(g_sc[k][0].tobytes() == np.asarray(c_sc[k][0], np.float32).tobytes()
                            and g_sc[k][1].tobytes() == np.asarray(c_sc[k][1], np.float32).tobytes() for k in (0, 1))
    gpu["sift_descriptors"], _ = best_ms(lambda: [g_desc.__setitem__(k, ctx.sift_descriptors(grads[k][0], grads[k][1], g_sc[k][0])[1])
                                                  for k in (0, 1)])
    cpu["sift_descriptors"], _ = once_ms(lambda: [c_desc.__setitem__(k, orc.sift_descriptors(grads[k][0], grads[k][1], c_sc[k][0])[1])
                                                  for k in (0, 1)])
    same["sift_descriptors"] = all(np.array_equal(g_desc[k].view(np.uint32), c_desc[k].view(np.uint32)) for k in (0, 1))
    # MatchDescriptors
    gpu["match"], gm = best_ms(lambda: ctx.match(g_desc[0], g_sc[0][1], g_desc[1], g_sc[1][1], 1000.0))
    cpu["match"], cm = once_ms(lambda: orc.match(c_desc[0], c_sc[0][1], c_desc[1], c_sc[1][1], 1000.0))
    same["match"] = bool(np.array_equal(gm[0], cm[0]) and np.array_equal(gm[1], cm[1]) and np.array_equal(gm[4], cm[4])
                         and gm[2].tobytes() == cm[2].tobytes())
    acc = np.nonzero(cm[4] == api.MATCH_ACCEPTED)[0]
    p1 = np.ascontiguousarray(c_sc[0][0][acc], np.float32)
    p2 = np.ascontiguousarray(c_sc[1][0][cm[0][acc]], np.float32)
    # the pair loop of FindFundamentalMatrixWithRANSAC over 200 hypotheses
    rng = rsc.msvc_rand(7)
    recs, excl = [], []
    for _ in range(4 * api.FUNDAMENTAL_RANSAC_ITERATIONS):
        if len(recs) == api.FUNDAMENTAL_RANSAC_ITERATIONS or len(p1) < 8:
            break
        ch = rsc.draw_eight(len(p1), rng)
        okf, f8 = rsc.find_fundamental_matrix(p1[ch], p2[ch])
        rec = rsc.hypothesis_record(f8, K, K) if okf and np.all(np.isfinite(f8)) else None
        if rec is not None:
            recs.append(rec)
            excl.append(ch)
    if recs:
        recs, excl = np.stack(recs), np.array(excl, np.int32)
        gpu["ransac_score"], gr = best_ms(lambda: ctx.ransac_score(recs, K, K, p1, p2, exclude=excl))
        cpu["ransac_score"], cr = once_ms(lambda: orc.ransac_score(recs, K, K, p1, p2, excl,
                                                                  api.FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD))
        same["ransac_score"] = bool(np.array_equal(gr[0], cr[0]) and gr[1].tobytes() == np.asarray(cr[1], np.float32).tobytes())
    # RectifyImage x 2 + ComputeDepthImage as written (1x1, whole row)
    gpu["rectify_x2_disparity"], gd = best_ms(lambda: ctx.rectify_disparity(imgs[0], imgs[1], hinv[0], hinv[1], win=1, max_disp=0))

    def cpu_rd():
        r0 = orc.rectify(imgs[0], np.zeros((h, w), np.uint8), hinv[0])
        r1 = orc.rectify(imgs[1], np.zeros((h, w), np.uint8), hinv[1])
        return orc.disparity_ref(r0, r1)
    cpu["rectify_x2_disparity"], cd = once_ms(cpu_rd)
    gd = gd[0] if isinstance(gd, tuple) else gd
    same["rectify_x2_disparity"] = bool((gd == cd).all())
    return {
        "workload": "741x500 synthetic pair: FAST-12, Shi-Tomasi + NMS and descriptors of both images (%d / %d features), "
                    "MatchDescriptors (%d accepted), %d RANSAC hypotheses scored, RectifyImage x 2 + ComputeDepthImage "
                    "(1x1, whole row)" % (len(g_sc[0][0]), len(g_sc[1][0]), len(acc), len(recs)),
        "gpu_ms": gpu, "gpu_ms_total": sum(gpu.values()),
        "cpu_ms": cpu, "cpu_ms_total": sum(cpu.values()), "cpu_cores": 1,
        "gpu_equals_cpu": same,
        "note": "gpu_ms: best of %d blocking host-pointer calls per stage (uploads and downloads included); cpu_ms: the "
                "oracle, one call, one thread (ransac_score: one thread per hypothesis).  Host-only steps of the chain "
                "(eight-point solves, rotations) are the reference's own code and not timed." % reps,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary kernels")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--match-n", type=int, default=200000, help="n1 = n2 of the matching leg (BASELINE config 5)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)
    # Rehearsal on a one-GPU box: STEREO_BENCH_SHARE_GPU=1 maps every rank to cuda:0 and uses
    # gloo for the barrier / max-reduce (RCCL cannot put two ranks on one device).
    share = os.environ.get("STEREO_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if share else "cuda"

    import stereo_amd
    from stereo_amd import synth, api

    ctx = stereo_amd.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---- headline: windowed bounded-D disparity scan, one pair per GPU ----------
    left, right = synth.rectified_pair(2024 + rank, H, W, D)
    d_l = torch.from_numpy(left).cuda()
    d_r = torch.from_numpy(right).cuda()
    d_o = torch.empty_like(d_l)

    def step():
        ctx.disparity_dev(d_l, d_r, d_o, win=WIN, max_disp=D, cost_mode=api.COST_SAD)

    # Before the W warm-up steps: let the GPU's clock settle.  A fresh box idles at a low clock and
    # needs some tens of milliseconds under load to reach its sustained one; 5 warm-up steps are
    # 1.5 ms (round 2's driver run read 0.300 ms per launch where the profile of the same command
    # read 0.281).  The same step, untimed, for SETTLE_MS of wall time; reported as "settle".
    t_settle = time.perf_counter()
    settle_steps = 0
    while (time.perf_counter() - t_settle) * 1e3 < SETTLE_MS and settle_steps < 4000:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        settle_steps += 20
    settle_ms = (time.perf_counter() - t_settle) * 1e3
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / args.steps  # one launch per step on this stream
    if world > 1:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    units = units_evaluated(W, H, D)
    value = units * world * args.steps / elapsed / 1e6

    out = {
        "metric": "disparity Mpix*disp/s",
        "value": value,
        "unit": "Mpix*disp/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "settle": {"steps": settle_steps, "ms": settle_ms,
                   "note": "untimed launches of the same step before the warm-up steps, until the clock has settled"},
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE config 2: %dx%d u8 rectified pair per GPU, %d disparities, "
                        "%dx%d SAD window (D1-ext), pair-sharded" % (W, H, D, WIN, WIN),
            "pairs_per_step": world, "width": W, "height": H, "max_disp": D, "win": WIN,
            "units_per_pair": units, "parallelism": "pairs x%d" % world,
        },
    }
    counters = load_counters()
    roof = valu_roofline(counters, "disp_band_9x9_sad_d256_2880x1988", units, OPS_PER_UNIT_WINDOWED, kernel_ms,
                         "disp_band_kernel<4,false>: disparity scan, %dx%d SAD, D=%d" % (WIN, WIN, D))
    c = counters.get("disp_band_9x9_sad_d256_2880x1988", {})
    # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this
    # command (null when profiles/kernel_counters.json has no entry)
    roof["traffic"] = (c["fetch_bytes"] + c["write_bytes"]) if "fetch_bytes" in c and "write_bytes" in c else None
    roof["traffic_source"] = c.get("source")
    roof["algorithmic_bytes"] = 3.0 * W * H
    roof["hbm_GBps_algorithmic"] = 3.0 * W * H / (kernel_ms * 1e-3) / 1e9
    out["roofline"] = roof

    # ---- secondary kernels (outside the timed region) ------------------------
    extra = {}
    single = world == 1 and not args.no_extra
    if single:
        ks, kw = max(3, min(25, args.steps // 4)), 1
        # reference mode: 1x1 window, whole-row search (bit-exact vs the reference).  The kernel
        # spends one v_sad_u32 per candidate and one v_min3_u32 per two: 1.5 lane-ops per unit is
        # the model of what it executes (SURVEY's 3-op model would credit it with more than it does)
        ms = event_time_ms(torch, lambda: ctx.disparity_dev(d_l, d_r, d_o, win=1, max_disp=0), ks, kw)
        u = W * W * H
        extra["disparity_reference_mode"] = {
            "workload": "%dx%d, 1x1 window, whole-row search (ComputeDepthImage as written)" % (W, H),
            "ms": ms, "Mpix_disp_per_s": u / (ms * 1e-3) / 1e6,
            "roofline": valu_roofline(counters, "disp_ref_2880x1988", u, 1.5, ms, "disp_ref_kernel"),
        }
        ms = event_time_ms(torch, lambda: ctx.disparity_dev(d_l, d_r, d_o, win=1, max_disp=D), ks, kw)
        extra["disparity_win1_d256"] = {
            "workload": "%dx%d, 1x1 window, %d disparities" % (W, H, D),
            "ms": ms, "Mpix_disp_per_s": units / (ms * 1e-3) / 1e6,
            "roofline": valu_roofline(counters, "disp_band_1x1_d256_2880x1988", units, OPS_PER_UNIT_WIN1, ms),
        }
        ms = event_time_ms(torch, lambda: ctx.disparity_dev(d_l, d_r, d_o, win=WIN, max_disp=D,
                                                            cost_mode=api.COST_SSD), ks, kw)
        extra["disparity_ssd_9x9_d256"] = {
            "workload": "%dx%d, %dx%d SSD window, %d disparities (same sliding-sum kernel, squared taps)" % (W, H, WIN, WIN, D),
            "ms": ms, "Mpix_disp_per_s": units / (ms * 1e-3) / 1e6,
            "roofline": valu_roofline(counters, "disp_band_9x9_ssd_d256_2880x1988", units, OPS_PER_UNIT_WINDOWED, ms),
        }
        # larger windows through the same sliding-sum kernel (two halo lanes per side up to 15x15; beyond
        # that the wave works as four 16-lane rows with row-DPP fetches)
        for wl, cm, nm in ((11, api.COST_SAD, "sad"), (13, api.COST_SAD, "sad"), (15, api.COST_SAD, "sad"),
                           (13, api.COST_SSD, "ssd"), (15, api.COST_SSD, "ssd"), (21, api.COST_SAD, "sad"),
                           (31, api.COST_SAD, "sad"), (31, api.COST_SSD, "ssd")):
            ms = event_time_ms(torch, lambda: ctx.disparity_dev(d_l, d_r, d_o, win=wl, max_disp=D, cost_mode=cm), 3, 1)
            extra["disparity_%s_%dx%d_d256" % (nm, wl, wl)] = {
                "workload": "%dx%d, %dx%d %s window, %d disparities" % (W, H, wl, wl, nm.upper(), D),
                "ms": ms, "Mpix_disp_per_s": units / (ms * 1e-3) / 1e6,
            }
        # rectification warp: batch of 8 x 1080p
        nimg, rh, rw = 8, 1080, 1920
        src_np = np.stack([synth.texture(70 + k, rh, rw) for k in range(2)])
        src = torch.from_numpy(src_np).cuda()
        src = src.repeat(nimg // 2, 1, 1).contiguous()
        dstt = torch.zeros_like(src)
        Hinv = np.stack([ctx.invert_homography(synth.homography(80 + k, rw, rh)).reshape(9)
                         for k in range(nimg)])
        ms = event_time_ms(torch, lambda: ctx.rectify_dev(src, dstt, Hinv, api.RECT_DST_IS_ZERO), ks, kw)
        gbs = 2.0 * nimg * rh * rw / (ms * 1e-3) / 1e9
        rr = valu_roofline(counters, "rectify_8x1080p", float(nimg * rh * rw), 50.0, ms, "rectify_kernel<true>")
        rr["model"] = ("~50 VALU lane-instructions per destination pixel (exact fp32 op order: two correctly rounded "
                       "divides, five conversions, four compares); x %d pixels" % (nimg * rh * rw))
        rr["hbm"] = {"achieved_GBps": gbs, "peak_GBps": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS, "bytes_per_pixel": 2,
                     "note": "algorithmic traffic only; the kernel is bound by VALU issue, not by HBM"}
        extra["rectify"] = {
            "workload": "%d x %dx%d u8, K R K^-1 homographies, zero-initialised destination" % (nimg, rw, rh),
            "ms": ms, "Mpix_per_s": nimg * rh * rw / (ms * 1e-3) / 1e6, "roofline": rr,
        }
        # FAST-12 corner detection on the left image of the headline pair
        xy = torch.empty((W * H // 4, 2), dtype=torch.float32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        ms = event_time_ms(torch, lambda: ctx.fast_detect_dev(d_l, xy, cnt, api.FAST_THRESHOLD), ks, kw)
        gbs = 1.0 * W * H / (ms * 1e-3) / 1e9
        fr = valu_roofline(counters, "fast12_2880x1988", float(W * H), 107.0, ms, "fast_flags_kernel + fast_scatter_kernel")
        fr["model"] = "~107 VALU lane-instructions per pixel (two 16-bit ring masks, run-of-12 test); x %d pixels" % (W * H)
        fr["hbm"] = {"achieved_GBps": gbs, "peak_GBps": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS, "bytes_per_pixel": 1,
                     "note": "algorithmic traffic only; the kernel is bound by VALU issue, not by HBM"}
        extra["fast12"] = {
            "workload": "%dx%d u8, threshold %d, corners compacted in raster order" % (W, H, api.FAST_THRESHOLD),
            "ms": ms, "Mpix_per_s": W * H / (ms * 1e-3) / 1e6, "corners": int(cnt.item()), "roofline": fr,
        }
        # Shi-Tomasi scoring + O(n^2) non-maximum suppression + score ordering of those corners
        ncorn = int(cnt.item())
        rng = np.random.RandomState(12)
        gxy_np = [np.clip(rng.randn(H, W) * 60, 0, 255).astype(np.uint8) for _ in range(2)]  # CV_8U Sobel stand-ins
        t_gx, t_gy = torch.from_numpy(gxy_np[0]).cuda(), torch.from_numpy(gxy_np[1]).cuda()
        kern = ctx.gaussian_kernel()
        s_xy = torch.empty((ncorn, 2), dtype=torch.float32, device="cuda")
        s_sc = torch.empty(ncorn, dtype=torch.float32, device="cuda")
        s_ix = torch.empty(ncorn, dtype=torch.int32, device="cuda")
        s_ct = torch.zeros(1, dtype=torch.int32, device="cuda")
        xy_in = xy[:ncorn].contiguous()
        ms = event_time_ms(torch, lambda: ctx.score_cluster_dev(t_gx, t_gy, xy_in, kern, api.ST_THRESH, api.NMS_WINDOW,
                                                                s_xy, s_sc, s_ix, s_ct), ks, kw)
        extra["score_nms"] = {
            "workload": "%d FAST corners of the %dx%d image: Shi-Tomasi score, threshold %.0f, O(n^2) suppression "
                        "within %d px, ordered by score" % (ncorn, W, H, api.ST_THRESH, api.NMS_WINDOW),
            "ms": ms, "features_per_s": ncorn / (ms * 1e-3), "survivors": int(s_ct.item()),
        }
        # SIFT-like descriptors of the survivors
        nsurv = int(s_ct.item())
        d_ang = torch.zeros(nsurv, dtype=torch.float32, device="cuda")
        d_desc = torch.empty((nsurv, 128), dtype=torch.float32, device="cuda")
        sxy = s_xy[:nsurv].contiguous()
        ms = event_time_ms(torch, lambda: ctx.sift_descriptors_dev(t_gx, t_gy, sxy, d_ang, d_desc), ks, kw)
        extra["sift_descriptors"] = {
            "workload": "128-float descriptors (orientation + 4x4 cells x 8 bins, normalise, clamp, renormalise) "
                        "of the %d surviving features" % nsurv,
            "ms": ms, "features_per_s": nsurv / (ms * 1e-3),
        }
        # RANSAC hypothesis scoring (the step between matching and rectification in config 5)
        from stereo_amd import ransac as rsc
        rn, rh_ = 200_000, api.FUNDAMENTAL_RANSAC_ITERATIONS
        rk1, rk2, rp1, rp2, rfm = synth.two_view_matches(77, rn)
        rrng = rsc.msvc_rand(1)
        recs, excl = [], []
        while len(recs) < rh_:
            ch = rsc.draw_eight(4000, rrng)
            okf, f8 = rsc.find_fundamental_matrix(rp1[ch], rp2[ch])
            rec = rsc.hypothesis_record(f8, rk1, rk2) if okf else None
            if rec is not None:
                recs.append(rec)
                excl.append(ch)
        recs, excl = np.stack(recs), np.array(excl, np.int32)
        t_p1, t_p2 = torch.from_numpy(rp1).cuda(), torch.from_numpy(rp2).cuda()
        r_inl = torch.zeros(rh_, dtype=torch.int32, device="cuda")
        r_sum = torch.zeros(rh_, dtype=torch.float32, device="cuda")
        ms = event_time_ms(torch, lambda: ctx.ransac_score_dev(recs, rk1, rk2, t_p1, t_p2, excl,
                                                               api.FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD,
                                                               r_inl, r_sum), ks, kw)
        extra["ransac_score"] = {
            "workload": "%d hypotheses x %d matched pairs: Lindstrom correction, triangulation, reprojection "
                        "error, inlier count and in-order error sum per hypothesis" % (rh_, rn),
            "ms": ms, "pair_evaluations_per_s": float(rh_) * rn / (ms * 1e-3),
            "best_inliers": int(r_inl.max().item()),
            "note": "includes the host-side record preparation and upload of each call",
        }
        ev = float(rh_) * rn
        ck = counters.get("ransac_error_200x200k")
        if ck and ck.get("valu_insts") and ck.get("kernel_ms"):
            rate = ck["valu_insts"] / (ck["kernel_ms"] * 1e-3)
            extra["ransac_score"]["roofline"] = {
                "bound": "valu", "kernel": "ransac_error_kernel", "kernel_ms": ck["kernel_ms"],
                "issue": {"valu_wave_instr_per_launch": ck["valu_insts"], "wave_instr_per_64_units": ck["valu_insts"] / (ev / 64),
                          "achieved_Ginstr_per_s": rate / 1e9, "ceiling_Ginstr_per_s": ISSUE_CEILING_WAVE_INSTR / 1e9,
                          "frac": rate / ISSUE_CEILING_WAVE_INSTR,
                          "vs_three_operand_stream": rate / ISSUE_RATE_3OP_WAVE_INSTR,
                          "source": "profiles/kernel_counters.json (%s): instruction count AND kernel time from the "
                                    "profile, not from this run" % ck.get("source", "?")}}

    # ---- north_star's multi-GPU workloads: run at every N ------------------------------
    def timed(fn, steps, warm=1):
        """Wall time per call of fn in ms, barrier + synchronize on both sides, max over ranks."""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el / steps * 1e3

    backend = dist.get_backend() if world > 1 else "none (one process)"
    # gloo (the shared-GPU rehearsal) moves CPU tensors only
    to_wire = (lambda t: t.cpu()) if (world > 1 and share) else (lambda t: t)
    if not args.no_extra:
        from stereo_amd import dist as sdist
        gathered = [None]

        # A leg that fails (an unexpected backend error on hardware this builder has never had: N > 1 ranks
        # on RCCL) must not take the headline line with it: the error is recorded in the leg's place.
        def run_leg(name, fn):
            try:
                fn()
            except Exception as e:  # noqa: BLE001 - reported, not swallowed
                gathered[0] = None
                torch.cuda.empty_cache()
                if rank == 0:
                    extra[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:500]), "n_ranks": world, "backend": backend}

        def leg_config3():
            # BASELINE config 3: one 4K pair, 512 disparities (two passes, u16 output), row bands + halo
            w3, h3, D3 = 3840, 2160, 512
            l3, r3 = synth.rectified_pair(303, h3, w3, D3)
            t_l3, t_r3 = torch.from_numpy(l3).cuda(), torch.from_numpy(r3).cuda()
            row0, rows, b0, bh = sdist.row_band(h3, rank, world, WIN // 2)
            counts3 = [sdist.row_band(h3, r_, world, WIN // 2)[1] for r_ in range(world)]
            band_out = torch.zeros((rows, w3), dtype=torch.int16, device="cuda")

            def band_step():
                if rows > 0:
                    ctx.disparity_rowband_dev(t_l3[b0:b0 + bh], t_r3[b0:b0 + bh], b0, h3, row0, rows, band_out,
                                              win=WIN, max_disp=D3)

            def band_gather_step():
                band_step()
                torch.cuda.current_stream().synchronize()
                g = sdist.gather_rows(to_wire(band_out), counts3, 0) if world > 1 else band_out
                gathered[0] = g

            # both legs from the same warm state: alternate them and keep each leg's best round (a single
            # 10-step round of a sub-millisecond call moves by several per cent while the clock settles)
            timed(band_gather_step, 5)
            rounds = [(timed(band_step, 10), timed(band_gather_step, 10)) for _ in range(3)]
            ms_c = min(r_[0] for r_ in rounds)
            ms_g = min(r_[1] for r_ in rounds)
            u3 = units_evaluated(w3, h3, D3)
            leg = {
                "workload": "BASELINE config 3: one %dx%d pair, %dx%d SAD, %d disparities, u16 map; %d row band(s) with "
                            "%d halo rows read from each rank's own copy, bands gathered to rank 0"
                            % (w3, h3, WIN, WIN, D3, world, WIN // 2),
                "n_ranks": world, "backend": backend, "scaling": "strong",
                "ms_compute": ms_c, "ms_with_gather": ms_g, "bytes_gathered": int((h3 - counts3[0]) * w3 * 2),
                "Mpix_disp_per_s": u3 / (ms_g * 1e-3) / 1e6, "Mpix_disp_per_s_compute_only": u3 / (ms_c * 1e-3) / 1e6,
            }
            if rank == 0:
                ref3 = torch.zeros((h3, w3), dtype=torch.int16, device="cuda")
                ctx.disparity_dev(t_l3, t_r3, ref3, win=WIN, max_disp=D3)
                torch.cuda.synchronize()
                leg["gathered_equals_single_gpu_map"] = bool(torch.equal(gathered[0].to(ref3.device), ref3))
                if world == 1:
                    # two launches (passes of 256 disparities) per call; best of three event-timed rounds
                    k3 = min(event_time_ms(torch, band_step, 10, 1) for _ in range(3))
                    leg["kernel_ms"] = k3
                    leg["roofline"] = valu_roofline(counters, "disp_band_9x9_sad_d512_3840x2160", u3, OPS_PER_UNIT_WINDOWED, k3,
                                                    "disp_band_kernel<4,false> x 2 passes")
                extra["disparity_config3_4k_d512"] = leg
                del ref3
            del t_l3, t_r3, band_out
            gathered[0] = None


        def leg_config4():
            # BASELINE config 4: 1024 1080p pairs, 192 disparities, 1024 / N pairs per rank
            w4, h4, D4, n4, distinct = 1920, 1080, 192, 1024, 4
            p4 = [synth.rectified_pair(400 + k, h4, w4, D4) for k in range(distinct)]
            lo4, hi4 = sdist.shard_range(n4, rank, world)
            counts4 = [sdist.shard_range(n4, r_, world)[1] - sdist.shard_range(n4, r_, world)[0] for r_ in range(world)]
            d_l4 = torch.from_numpy(np.stack([p[0] for p in p4])).cuda()
            d_r4 = torch.from_numpy(np.stack([p[1] for p in p4])).cuda()
            sel = torch.arange(lo4, hi4, device="cuda") % distinct      # pair k of the batch is distinct pair k mod 4
            t_l4, t_r4 = d_l4[sel].contiguous(), d_r4[sel].contiguous()
            t_o4 = torch.empty_like(t_l4)

            def batch_step():
                ctx.disparity_dev(t_l4, t_r4, t_o4, win=WIN, max_disp=D4)

            def batch_gather_step():
                batch_step()
                torch.cuda.current_stream().synchronize()
                gathered[0] = sdist.gather_rows(to_wire(t_o4), counts4, 0) if world > 1 else t_o4

            ms_c = timed(batch_step, 2)
            ms_g = timed(batch_gather_step, 2)
            u4 = units_evaluated(w4, h4, D4) * n4
            leg = {
                "workload": "BASELINE config 4: %d x %dx%d pairs (%d distinct, repeated), %dx%d SAD, %d disparities; "
                            "%d pairs per rank in one batched launch, maps gathered to rank 0"
                            % (n4, w4, h4, distinct, WIN, WIN, D4, hi4 - lo4),
                "n_ranks": world, "backend": backend, "scaling": "strong",
                "ms_compute": ms_c, "ms_with_gather": ms_g, "bytes_gathered": int((n4 - counts4[0]) * w4 * h4),
                "pairs_per_s": n4 / (ms_c * 1e-3), "pairs_per_s_with_gather": n4 / (ms_g * 1e-3),
                "Mpix_disp_per_s": u4 / (ms_c * 1e-3) / 1e6,
            }
            if rank == 0:
                single4 = torch.empty_like(d_l4)
                for k in range(distinct):
                    ctx.disparity_dev(d_l4[k], d_r4[k], single4[k], win=WIN, max_disp=D4)
                torch.cuda.synchronize()
                g = gathered[0].to(single4.device)
                leg["gathered_equals_single_pair_maps"] = bool(
                    (g.view(n4 // distinct, distinct, h4, w4) == single4.unsqueeze(0)).all().item())
                if world == 1:
                    # one launch per call: HIP events around two calls give the kernel time (ms_compute is wall clock)
                    k4 = event_time_ms(torch, batch_step, 2, 0)
                    leg["kernel_ms"] = k4
                    leg["roofline"] = valu_roofline(counters, "disp_band_9x9_sad_d192_1024x1080p", u4, OPS_PER_UNIT_WINDOWED, k4,
                                                    "disp_band_kernel<4,false>, grid z = 1024 pairs")
                extra["disparity_config4_batch1024"] = leg
                del single4, g
            gathered[0] = None
            del t_l4, t_r4, t_o4
            torch.cuda.empty_cache()

        run_leg("disparity_config3_4k_d512", leg_config3)
        run_leg("disparity_config4_batch1024", leg_config4)

    if not args.no_extra:
        # descriptor matching (BASELINE config 5).  With N > 1 ranks list1's queries are sharded
        # (list2 replicated, no data-path collective): strong scaling of the one 200k x 200k job.
        from stereo_amd.dist import shard_range
        n = args.match_n
        d1, s1, d2, s2 = synth.descriptor_sets(42, n, n)
        lo, hi = shard_range(n, rank, world)
        nq = hi - lo
        td1, ts1 = torch.from_numpy(d1[lo:hi]).cuda(), torch.from_numpy(s1[lo:hi]).cuda()
        td2, ts2 = torch.from_numpy(d2).cuda(), torch.from_numpy(s2).cuda()
        oc = torch.empty(nq, dtype=torch.int32, device="cuda")
        os_ = torch.empty(nq, dtype=torch.int32, device="cuda")
        om = torch.empty(nq, dtype=torch.float32, device="cuda")
        od = torch.empty(nq, dtype=torch.float32, device="cuda")
        ot = torch.empty(nq, dtype=torch.uint8, device="cuda")

        def mstep():
            ctx.match_dev(td1, ts1, td2, ts2, 1000.0, oc, os_, om, od, ot)

        msteps = 10
        ms = timed(mstep, msteps)
        counts_m = [shard_range(n, r_, world)[1] - shard_range(n, r_, world)[0] for r_ in range(world)]
        mg = [None]

        def mstep_gather():
            mstep()
            torch.cuda.current_stream().synchronize()
            if world > 1:   # five small per-query arrays, in list1 order (the order the reference emits pairs in)
                mg[0] = [sdist.gather_rows(to_wire(t_), counts_m, 0) for t_ in (oc, os_, om, od, ot)]
            else:
                mg[0] = [oc, os_, om, od, ot]

        gather_err = None
        try:
            ms_g = timed(mstep_gather, 3)
        except Exception as e:  # noqa: BLE001 - recorded in the leg (see run_leg)
            ms_g, gather_err, mg[0] = None, "%s: %s" % (type(e).__name__, str(e)[:500]), None
        mstats = ctx.match_stats()      # this rank's share: candidates the exact replay looked at, overflows
        acc = torch.tensor([float((ot == 3).sum().item())], device=red_dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(acc)
        same_m = None
        if rank == 0 and world > 1 and mg[0] is not None:
            # the single-GPU result of the whole job, on rank 0, against the gathered one
            fd1, fs1 = torch.from_numpy(d1).cuda(), torch.from_numpy(s1).cuda()
            fo = [torch.empty(n, dtype=t_.dtype, device="cuda") for t_ in (oc, os_, om, od, ot)]
            ctx.match_dev(fd1, fs1, td2, ts2, 1000.0, *fo)
            torch.cuda.synchronize()
            same_m = all(bool(torch.equal(g_.to("cuda").view(torch.uint8), f_.view(torch.uint8)))
                         for g_, f_ in zip(mg[0], fo))
            del fd1, fs1, fo
        flops = 2.0 * n * n * 128
        # the prefilter takes each fp32 dot product as ONE bf16 MFMA product (hi.hi) under a rigorous
        # error bound; the exact replay restores bit-exact results
        mfma_flops = 1.0 * flops
        extra["match"] = {
            "workload": "BASELINE config 5: %d x %d 128-d fp32 SIFT-like descriptors, bf16 MFMA prefilter "
                        "+ exact replay (bit-exact indices); list1 sharded over %d GPU(s)" % (n, n, world),
            "n_gpus": world, "n_ranks": world, "backend": backend, "scaling": "strong",
            "ms": ms, "matched_descriptors_per_s": n / (ms * 1e-3),
            "pair_evaluations_per_s": float(n) * n / (ms * 1e-3),
            "accepted": int(acc.item()),
            "candidates_per_query": mstats["candidates_per_query"], "candidates_max": mstats["candidates_max"],
            "overflow_queries": mstats["overflow_queries"],
            "workload_note": "synth.descriptor_sets(42): half of list2's rows are noisy copies (sigma 0.01) of list1 rows "
                             "drawn with replacement, the rest fresh; scores independent U[700,5000], so the score gate "
                             "rejects 59 % of the queries, the ratio test 27 %, 14 % are accepted; candidates_* and "
                             "overflow_queries are rank 0's share (stx_match_stats)",
            "ms_with_gather": ms_g, "bytes_gathered": int((n - counts_m[0]) * 17),
            "gathered_equals_single_gpu_result": same_m, "gather_error": gather_err,
            "roofline": {"bound": "mfma", "achieved": mfma_flops / (ms * 1e-3) / 1e12 / world,
                         "peak": MFMA_BF16_TFLOPS,
                         "unit": "TFLOP/s per GPU",
                         "frac": mfma_flops / (ms * 1e-3) / 1e12 / MFMA_BF16_TFLOPS / world,
                         "dtype": "bf16 (one MFMA product per fp32 product)",
                         "algorithmic_TFLOPs": flops / (ms * 1e-3) / 1e12,
                         "note": "whole call: prep + prefilter + replay + overflow pass; a register-only MFMA loop "
                                 "sustains ~2050 TFLOP/s on this chip with non-zero operands "
                                 "(tools/mfma_bench.hip, profiles/)"},
        }
        if world == 1:
            # the same size on descriptors with repeated structure: near-ties pile up in the scan, the
            # replay sees ~4x the candidates and the two large clusters' queries take the overflow pass
            c1, cs1, c2, cs2 = synth.descriptor_sets_clustered(77, n, n)
            tc1, tcs1 = torch.from_numpy(c1).cuda(), torch.from_numpy(cs1).cuda()
            tc2, tcs2 = torch.from_numpy(c2).cuda(), torch.from_numpy(cs2).cuda()
            co = [torch.empty(n, dtype=t_.dtype, device="cuda") for t_ in (oc, os_, om, od, ot)]

            def cstep():
                ctx.match_dev(tc1, tcs1, tc2, tcs2, 1000.0, *co)

            ms_cl = timed(cstep, 5)
            cst = ctx.match_stats()
            nchk = 256
            # the checker: the 16-lanes-per-query exact scan, which shares nothing with the prefilter path
            # (the one-lane-per-query kernel would take 0.4 s for any number of queries up to a full chip)
            ctx.set_match_algo(api.MATCH_SMALL)
            try:
                ce = [torch.empty(nchk, dtype=t_.dtype, device="cuda") for t_ in co]
                ctx.match_dev(tc1[:nchk].contiguous(), tcs1[:nchk].contiguous(), tc2, tcs2, 1000.0, *ce)
                torch.cuda.synchronize()
            finally:
                ctx.set_match_algo(api.MATCH_AUTO)
            same_c = all(bool(torch.equal(a_[:nchk].view(torch.uint8), b_.view(torch.uint8))) for a_, b_ in zip(co, ce))
            extra["match_clustered"] = {
                "workload": "%d x %d descriptors with repeated structure (synth.descriptor_sets_clustered(77): 2000 "
                            "centres, ~100 near-duplicates each at noise 1e-3, two centres with ~500 each)" % (n, n),
                "ms": ms_cl, "matched_descriptors_per_s": n / (ms_cl * 1e-3), "vs_ordinary_set": ms_cl / ms,
                "candidates_per_query": cst["candidates_per_query"], "candidates_max": cst["candidates_max"],
                "overflow_queries": cst["overflow_queries"],
                "first_%d_queries_equal_exact_gpu_scan" % nchk: same_c,   # STX_MATCH_SMALL over all of list2
            }
    if extra and rank == 0:
        out["extra"] = extra

    # ---- CPU baseline: the oracle restatement on a bounded sample ------------
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import oracle as orc
        rows = H  # the whole pair: ~4 s of single-thread work
        band_l, band_r = left[:rows], right[:rows]
        t0 = time.perf_counter()
        ref = orc.disparity_ext(band_l, band_r, WIN, D, 0, sliding=True)
        dt = time.perf_counter() - t0
        got = d_o  # last extra leg may have overwritten it: recompute the headline output
        ctx.disparity_dev(d_l, d_r, d_o, win=WIN, max_disp=D, cost_mode=api.COST_SAD)
        torch.cuda.synchronize()
        same = bool((d_o.cpu().numpy() == ref).all())
        out["cpu_baseline"] = {
            "value": units_evaluated(W, rows, D) / dt / 1e6,
            "unit": "Mpix*disp/s",
            "cores": 1,
            "kind": "port",
            "sample": "oracle_disparity_ext_sliding on %d of %d rows of the same pair "
                      "(%.1f s, single thread; the reference is single-threaded)" % (rows, H, dt),
            "matches_gpu_on_sample": same,
        }

        if not args.no_extra:
            # all host cores: row bands with halo on a thread pool (ctypes releases the GIL)
            from concurrent.futures import ThreadPoolExecutor
            cores = max(1, min(16, os.cpu_count() or 1))
            mt_rows = H
            halo = WIN // 2

            def band(k):
                lo, hi = k * mt_rows // cores, (k + 1) * mt_rows // cores
                b0, b1 = max(0, lo - halo), min(H, hi + halo)
                r = orc.disparity_ext(left[b0:b1], right[b0:b1], WIN, D, 0, sliding=True)
                return r[lo - b0:hi - b0]

            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                parts = list(ex.map(band, range(cores)))
            dt_mt = time.perf_counter() - t0
            same_mt = bool((np.concatenate(parts) == d_o.cpu().numpy()).all())
            out["extra"]["cpu_disparity_all_cores"] = {
                "value": units_evaluated(W, mt_rows, D) / dt_mt / 1e6, "unit": "Mpix*disp/s", "cores": cores,
                "sample": "all %d rows in %d row bands with halo, one thread each (%.1f s)" % (mt_rows, cores, dt_mt),
                "matches_gpu_on_sample": same_mt}
            # matching: oracle_match on a slice of list1 against the whole list2
            nq_cpu = max(1, min(n, int(2.0e8 // max(n, 1))))
            t0 = time.perf_counter()
            cm = orc.match(d1[:nq_cpu], s1[:nq_cpu], d2, s2, 1000.0)
            dt_m = time.perf_counter() - t0
            same_m = bool((cm[0] == oc[:nq_cpu].cpu().numpy()).all() and (cm[4] == ot[:nq_cpu].cpu().numpy()).all())
            out["extra"]["cpu_match"] = {
                "value": nq_cpu / dt_m, "unit": "matched descriptors/s", "cores": 1,
                "sample": "oracle_match, first %d queries x %d (%.1f s, single thread, no per-distance "
                          "heap allocation)" % (nq_cpu, n, dt_m),
                "matches_gpu_on_sample": same_m}
            # scoring + NMS: the oracle on the first 15000 corners
            nsc = min(ncorn, 15000)
            xy_host = xy_in[:nsc].cpu().numpy()
            t0 = time.perf_counter()
            cs = orc.score_and_cluster(gxy_np[0], gxy_np[1], xy_host, kern, api.ST_THRESH, api.NMS_WINDOW)
            dt_s = time.perf_counter() - t0
            gs = ctx.score_cluster(gxy_np[0], gxy_np[1], xy_host, kern, api.ST_THRESH, api.NMS_WINDOW)
            out["extra"]["cpu_score_nms"] = {
                "value": nsc / dt_s, "unit": "features/s", "cores": 1,
                "sample": "oracle_score_and_cluster, first %d corners (%.2f s, single thread; O(n^2), so the "
                          "rate falls with n)" % (nsc, dt_s),
                "matches_gpu_on_sample": bool(gs[2].tolist() == cs[2].tolist() and gs[1].tobytes() == cs[1].tobytes())}
            t0 = time.perf_counter()
            ca, cd = orc.sift_descriptors(gxy_np[0], gxy_np[1], sxy.cpu().numpy())
            dt_d = time.perf_counter() - t0
            ok_d = bool(np.array_equal(cd.view(np.uint32), d_desc.cpu().numpy().view(np.uint32)))
            out["extra"]["cpu_sift_descriptors"] = {
                "value": nsurv / dt_d, "unit": "features/s", "cores": 1,
                "sample": "oracle_sift_descriptors, all %d features (%.2f s, single thread)" % (nsurv, dt_d),
                "matches_gpu_on_sample": ok_d}
            # hypothesis scoring: the oracle on the first 64 hypotheses (OpenMP over hypotheses)
            nh_cpu = min(rh_, 64)
            t0 = time.perf_counter()
            ci, cs_, _, _ = orc.ransac_score(recs[:nh_cpu], rk1, rk2, rp1, rp2, excl[:nh_cpu],
                                             api.FUNDAMENTAL_REPROJECTION_ERROR_THRESHOLD)
            dt_h = time.perf_counter() - t0
            out["extra"]["cpu_ransac_score"] = {
                "value": nh_cpu * rn / dt_h, "unit": "pair evaluations/s", "cores": min(nh_cpu, cores),
                "sample": "oracle_ransac_score, first %d hypotheses x %d pairs, one thread per hypothesis "
                          "(%.2f s)" % (nh_cpu, rn, dt_h),
                "matches_gpu_on_sample": bool(np.array_equal(ci, r_inl[:nh_cpu].cpu().numpy())
                                              and np.array_equal(cs_.view(np.uint32),
                                                                 r_sum[:nh_cpu].cpu().numpy().view(np.uint32)))}
            # rectification: oracle_rectify on one image of the batch
            t0 = time.perf_counter()
            cr = orc.rectify(src_np[0], np.zeros((rh, rw), np.uint8), Hinv[0].reshape(3, 3))
            dt_r = time.perf_counter() - t0
            out["extra"]["cpu_rectify"] = {
                "value": rh * rw / dt_r / 1e6, "unit": "Mpix/s", "cores": 1,
                "sample": "oracle_rectify, one %dx%d image (%.2f s, single thread, inverse hoisted)" % (rw, rh, dt_r),
                "matches_gpu_on_sample": bool((cr == dstt[0].cpu().numpy()).all())}

            # BASELINE config 1 as a chain, stage by stage, GPU library beside the oracle
            out["extra"]["config1_chain"] = config1_chain(ctx, orc, api, synth)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
