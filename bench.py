#!/usr/bin/env python3
"""Headline benchmark: audio frames/s through the SpeechT5-base speech encoder, 30 s x batch 32 per GPU
(BASELINE.json configs[1]), synthetic 16 kHz audio resident in HBM, random-init weights of the named
architecture (no checkpoint is reachable offline).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one forward of the whole hot path (conv feature extractor -> 12-layer encoder) over this rank's
batch, plus -- for N > 1 -- the single RCCL all-gather of the [32, 1499, 768] embeddings that the
data-parallel design performs per step, issued asynchronously so that it overlaps the next step's
kernels; all K gathers complete inside the timed region.  Weak scaling: every rank encodes its own 32
clips.  Rank 0 prints ONE JSON line; `value` = frames encoded by all ranks / max-over-ranks time.

Extra objects on the line:
  roofline      the dominant kernel = the profiling bucket with the largest summed launch time (the projection GEMM at
                30 s x 32, attention at 10 min x 4; picked from the warm-up steps, whose every launch is bracketed):
                algorithmic FLOPs / its summed launch time, measured live with HIP events on the launch stream
                over the timed region -- in which only THIS bucket's launches carry events (two records around each of
                ~100 launches cost ~2 ms of a 52 ms step); peak = the dense MFMA
                rate of the instruction it issues (fp16: 2.5 PFLOP/s; fp32: 157.3 TFLOP/s; MI355X_MICROARCH.md);
                for f16x3 the 3-MFMAs-per-product issue rate is reported beside the algorithmic fraction;
                traffic = PMC HBM bytes when profiles/ holds them, else null.
  kernels       per-bucket launches / ms / rate per step, from up to 5 further steps after the timed region (every launch
                bracketed); with --warmup 0 the timed steps themselves are bracketed launch by launch, as before.
  alt_precision the other precision mode measured for 3 steps in the same process.
  multi_gpu     (N > 1 or --force-collective) what a first multi-GPU run needs to explain itself: per-rank ms_per_step (min / max /
                rank of max / all), how long the forward's stream stood still for the previous step's all-gather (hipEvents around
                every wait), gathered bytes per step, the world size RCCL reports, and a check that every rank's block of the
                gathered tensor equals a probe row that rank all-gathered separately.
  packed_reference_batches  the REFERENCE'S operating point (batches of two 2-6 s utterances, corpus order, …base…py:67-68) on the packed
                forward (loco_forward_packed, 32 batches per launch sequence): frames/s counting the frames the reference pickles.
  cpu_baseline  the CPU oracle (oracle/speecht5_oracle.py, torch fp32, all host cores) timed on a bounded
                sample of the same workload (8 clips of 30 s: 1 warm-up + 3 timed passes, median) on rank 0 at
                N = 1 -- reported, not targeted.
  embed_rel_l2  relative L2 of the GPU embeddings vs that oracle run on the sample clips (bar: 1e-3).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLIP_SECONDS = 30
BATCH_PER_GPU = 32
PEAK_F32_MFMA_TFLOPS = 157.3   # dense fp32-input MFMA, MI355X_MICROARCH.md
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense fp16/bf16 MFMA (the 5 PF headline includes 2:1 sparsity)
# What a register-resident loop of nothing but fp16 MFMAs reaches on RANDOM operands on this part: the socket power limit holds the
# clock at ~1.7 GHz (tools/mfma_probe/, profiles/r03_mfma_energy_probe.txt; 2 265-2 480 on all-zero operands).  Reported beside the
# contractual peak, never instead of it.
PRACTICAL_F16_MFMA_TFLOPS_RANDOM_OPERANDS = 1640.0
# profiling bucket (loco_api.hip kKernelNames) -> (kernel symbol, bound, peak, MFMAs issued per algorithmic product)
BUCKETS = {"gemm_f32": ("gemm_f32_kernel", "mfma", PEAK_F32_MFMA_TFLOPS, 1),
           "attention_f32": ("attention_kernel", "mfma", PEAK_F32_MFMA_TFLOPS, 1),
           "pos_conv_f32": ("pos_conv_kernel", "mfma", PEAK_F32_MFMA_TFLOPS, 1),
           "gemm_f16x3": ("gemm_f16x3_dma_kernel", "mfma", PEAK_F16_MFMA_TFLOPS, 3),
           "attention_f16x3": ("attention_f16x3_kernel", "mfma", PEAK_F16_MFMA_TFLOPS, 3),
           "pos_conv_f16x3_gemm": ("gemm_f16x3_dma_kernel (N = 48 form)", "mfma", PEAK_F16_MFMA_TFLOPS, 3)}
PEAK_HBM_GBPS = 8000.0


def host_cores() -> int:
    """Cores this process may actually use: min(affinity, cgroup quota, cpu_count).  A 1-GPU slice of the
    GPU box exposes all 256 hardware threads in cpu_count() but is entitled to a 16-CPU share; running the
    CPU baseline with 256 torch threads there oversubscribes ~16x and measures scheduler thrash."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    cap = os.environ.get("LOCO_CPU_BASELINE_THREADS")
    if cap:
        n = max(1, int(cap))
    elif n > 32:
        n = 16  # documented CPU share of a one-GPU box when no quota is visible
    return n


def traffic_for(stat_name, workload_key):
    """PMC L2-miss bytes per launch of this kernel bucket FOR THIS WORKLOAD, from the newest committed
    profiles/*_<bucket>_traffic.json whose "workload" field names the same shape (tools/pmc_traffic.py writes it) -> (bytes,
    file name); (None, None) when no pass was taken on this shape -- a figure measured at 30 s x 32 says nothing about 10 min x 4."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{stat_name}_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("workload", "30sx32") == workload_key and "hbm_bytes_per_launch" in d:
            return d["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def make_roofline(by, precision, steps, workload_key="30sx32"):
    """Roofline object of the DOMINANT kernel = the profiling bucket with the largest summed launch time in the timed region
    (the projection GEMM at 30 s x 32, attention at 10 min x 4): algorithmic FLOPs of its launches / their summed
    HIP-event duration against the dense MFMA peak of the instruction it issues; HBM-bound buckets against 8 TB/s."""
    cands = {k: v for k, v in by.items() if v["ms"] > 0}
    if not cands:
        return None
    stat_name = max(cands, key=lambda k: cands[k]["ms"])
    g = cands[stat_name]
    kernel, bound, peak, mfma_per_product = BUCKETS.get(stat_name, (stat_name, "hbm", PEAK_HBM_GBPS, 1))
    if precision == "f16x2" and stat_name in ("gemm_f16x3", "pos_conv_f16x3_gemm"):
        mfma_per_product = 2  # the W_lo term is dropped in that mode
    # PMC bytes per launch of the same kernel on the same workload, from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py);
    # counters cannot be read inside this process, so this is never a live figure: traffic_source names the file it came from
    traffic, traffic_source = traffic_for(stat_name, workload_key)
    if bound == "mfma":
        ach, unit = g["flops"] / (g["ms"] * 1e-3) / 1e12, "TFLOP/s"
    else:
        ach, unit = g["bytes"] / (g["ms"] * 1e-3) / 1e9, "GB/s"
    r = {"kernel": kernel, "bucket": stat_name, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
         "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_source, "launches_per_step": g["launches"] / steps,
         "avg_launch_ms": round(g["ms"] / g["launches"], 4), "share_of_kernel_time": round(g["ms"] / sum(v["ms"] for v in cands.values()), 3),
         "flops_per_launch_avg": g["flops"] / g["launches"], "algorithmic_bytes_per_launch_avg": g["bytes"] / g["launches"]}
    if mfma_per_product > 1:
        # the split algorithm issues 3 fp16 MFMAs per algorithmic product: matrix-pipe utilisation is 3x the algorithmic fraction
        r["mfma_flops_issued_per_algorithmic_flop"] = mfma_per_product
        r["mfma_issued_tflops"] = round(ach * mfma_per_product, 1)
        r["mfma_issued_frac_of_peak"] = round(ach * mfma_per_product / peak, 4)
        r["practical_peak_random_operands_tflops"] = PRACTICAL_F16_MFMA_TFLOPS_RANDOM_OPERANDS
        r["mfma_issued_frac_of_practical_peak"] = round(ach * mfma_per_product / PRACTICAL_F16_MFMA_TFLOPS_RANDOM_OPERANDS, 4)
    return r


def workload_label(clip_seconds, batch):
    """config.workload from the arguments; the BASELINE.json config it corresponds to, if any."""
    which = ""
    if abs(clip_seconds - 30) < 1e-9 and batch == 32:
        which = " (BASELINE.json configs[1])"
    elif abs(clip_seconds - 600) < 1e-9 and batch == 4:
        which = " (BASELINE.json configs[2])"
    return (f"SpeechT5-base speech encoder, synthetic 16 kHz {clip_seconds:g} s clips, batch {batch} per GPU{which}, "
            "random-init weights")


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks as CHILD processes through
    `python -m torch.distributed.run` (one rank per GPU over RCCL), relay their output -- rank 0 prints the JSON line --
    and return the launcher's exit code.  Called before anything in this process has touched the GPU (importing torch does
    not); the parent never initialises HIP, it only waits."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def flops_per_clip(T: int) -> float:
    """Algorithmic FLOPs of one clip: 284.2 MFLOP*T + 36 864 FLOP*T^2."""
    return T * 284.2e6 + 36864.0 * T * T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clip-seconds", type=float, default=CLIP_SECONDS)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU)
    ap.add_argument("--precision", choices=["f32", "f16x3", "f16x2"], default="f16x3",
                    help="contraction arithmetic: f16x3 = three fp16 MFMAs per fp32-class product (default; 3.5e-6 rel L2 of fp64), "
                         "f32 = exact fp32 MFMA (2.3e-6), ~2x slower")
    ap.add_argument("--no-alt", action="store_true", help="skip the short measurement of the other precision mode")
    ap.add_argument("--no-two-streams", action="store_true",
                    help="skip the short un-profiled run of the library's default schedule (two half-batches on two streams)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-packed", action="store_true",
                    help="skip the short measurement of the reference's own operating point (batches of two 2-6 s utterances) on the packed forward")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: create the RCCL process group of one rank anyway and issue the per-step all_gather_into_tensor on "
                         "the device embeddings (what every rank does at N > 1), then check the gathered tensor against the local one")
    ap.add_argument("--cpu-sample-clips", type=int, default=8,
                    help="clips of the same workload the CPU oracle is timed on (1 warm-up + --cpu-reps timed passes, median).  Default 8 "
                         "(batch 8: ~20 s per pass on 16 cores, so the default run stays within minutes); 32 = the full configuration "
                         "SURVEY.md 8d names (batch 32: ~80 s per pass, 17 GB of host memory) -- same frames/s within a few percent")
    ap.add_argument("--cpu-reps", type=int, default=3)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree "
                         f"(`python bench.py --gpus {args.gpus} ...` without a launcher starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the encoder has no CPU path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py rank {rank}: --gpus {args.gpus} needs {world} GPUs on this node, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.force_collective and "RANK" not in os.environ:  # started without the launcher: a group of one rank all the same
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    la = importlib.import_module("loco-asr_amd")
    dp = importlib.import_module("loco-asr_amd.dp")
    L = int(round(args.clip_seconds * 16000))
    B = args.batch
    T = la.synth.conv_out_length(L)

    sd = la.synth.encoder_state_dict(0)
    pre, enc_sd = la.synth.split_state_dict(sd)
    model = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                             {k: torch.from_numpy(v) for k, v in enc_sd.items()},
                                                             precision=args.precision).to(dev)
    enc = model.speecht5.encoder
    x_np, m_np = la.synth.batch([L] * B, first_index=rank * B)
    x = torch.from_numpy(x_np).to(dev)
    m = torch.from_numpy(m_np).to(dev)

    dp.FORCE_COLLECTIVE = bool(args.force_collective)
    collective = world > 1 or args.force_collective
    gatherer = dp.OverlappedGather(timed=collective)

    def step():
        # forward, then hand the embeddings to the one RCCL all-gather of the step; the collective runs on RCCL's
        # stream under the next step's kernels and is waited for before the next one is issued (and before the clock stops)
        out = enc(input_values=x, attention_mask=m).last_hidden_state
        gatherer.submit(out)
        return out

    # per-kernel events are on from the first warm-up step: they keep loco_forward on one stream, so every launch of the
    # run (and of a rocprofv3 trace of it) has the shape the roofline is quoted for
    enc.set_profiling(True)
    enc.profile_reset()
    for _ in range(args.warmup):
        y = step()
    gatherer.finish()
    torch.cuda.synchronize()
    # Two event records per launch cost ~1 ms of a 52 ms step when all ~100 launches carry them.  The warm-up steps are bracketed
    # launch by launch (the per-kernel table below comes from them); the TIMED steps carry events on the launches of the dominant
    # bucket only -- the kernel the roofline is quoted for, measured live over the timed region as the contract asks.
    warm_stats = enc.profile_read() if args.warmup > 0 else []
    dominant = max(warm_stats, key=lambda s_: s_["ms"])["name"] if warm_stats else None
    enc.set_profiling_filter(dominant)
    enc.profile_reset()
    gatherer.blocked_ms()  # forget the warm-up's waits
    gathers0, bytes0 = gatherer.gathers, gatherer.bytes_gathered
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    gathered = gatherer.finish()  # the last step's gather is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert gathered.shape[0] == world * B
    # ---- what the first multi-GPU run needs to explain itself: per-rank step time, how long the forward's stream
    # stood still for the collective, bytes gathered, and a check that every rank's block of the gathered tensor IS that rank's
    multi = None
    if collective:
        blocked_ms = gatherer.blocked_ms()
        mine = torch.tensor([elapsed / args.steps * 1e3, blocked_ms / args.steps], dtype=torch.float64, device=dev)
        every = torch.empty((world, 2), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every.view(-1), mine)
        probe = y[0, 0].contiguous()  # 768 floats: row 0 of this rank's first clip, as this rank computed it
        probes = torch.empty((world, probe.numel()), dtype=probe.dtype, device=dev)
        dist.all_gather_into_tensor(probes.view(-1), probe)
        blocks_ok = [bool(torch.equal(gathered[r * B, 0], probes[r])) for r in range(world)]
        every = every.cpu()
        step_ms = every[:, 0].tolist()
        multi = {"world_size_seen_by_rccl": dist.get_world_size(), "backend": dist.get_backend(),
                 "per_rank_ms_per_step": {"min": round(min(step_ms), 3), "max": round(max(step_ms), 3),
                                          "rank_of_max": int(max(range(world), key=lambda r: step_ms[r])),
                                          "all": [round(v, 3) for v in step_ms]},
                 "stream_blocked_by_gather_ms_per_step": {"this_rank": round(blocked_ms / args.steps, 4),
                                                         "max": round(float(every[:, 1].max()), 4),
                                                         "all": [round(v, 4) for v in every[:, 1].tolist()],
                                                         "note": "hipEvents around every wait for the previous step's all-gather on the "
                                                                 "stream that waits: 0 = the collective finished under the next forward"},
                 "gathers_in_timed_region": gatherer.gathers - gathers0,
                 "gathered_bytes_per_step": (gatherer.bytes_gathered - bytes0) // max(1, gatherer.gathers - gathers0),
                 "every_ranks_block_matches_its_probe_row": blocks_ok}
        if rank == 0:
            assert all(blocks_ok), f"gathered blocks do not match their ranks' own rows: {blocks_ok}"
    if args.force_collective and world == 1:  # the collective really ran (RCCL, device tensors) and returned the rank's own rows
        assert gathered.data_ptr() != y.data_ptr() and torch.equal(gathered, y), "all_gather_into_tensor at world size 1 changed the embeddings"
    # the default range policy ("fp32": re-run out-of-range batches on the exact-fp32 kernels) was active; a re-run inside the
    # timed region would make `value` a mixed-precision figure -- refuse to report one
    if enc.last_range_fallback:
        raise SystemExit("bench.py: the synthetic batch left the f16x3 activation range and was re-run in fp32 -- not a valid f16x3 measurement")
    stats = enc.profile_read()
    enc.set_profiling_filter(None)
    # the per-kernel table: a few more steps, every launch bracketed, on the warm chip (the warm-up steps include the first, cold one)
    table_stats, table_steps = [], 0
    if dominant:
        table_steps = max(1, min(args.steps, 5))
        enc.profile_reset()
        for _ in range(table_steps):
            step()
        gatherer.finish()
        torch.cuda.synchronize()
        table_stats = enc.profile_read()
    enc.set_profiling(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    frames_per_step = world * B * T
    value = frames_per_step * args.steps / elapsed

    result = None
    if rank == 0:
        by = {s["name"]: s for s in stats}
        table, table_steps = (table_stats, table_steps) if dominant else (stats, args.steps)
        kernels = {s["name"]: {"launches_per_step": s["launches"] / table_steps, "ms_per_step": s["ms"] / table_steps,
                               "tflops": (s["flops"] / (s["ms"] * 1e-3) / 1e12) if s["ms"] > 0 and s["flops"] else None,
                               "gbps": (s["bytes"] / (s["ms"] * 1e-3) / 1e9) if s["ms"] > 0 and s["bytes"] else None}
                   for s in table}
        wkey = f"{args.clip_seconds:g}sx{B}" if args.clip_seconds < 60 else f"{args.clip_seconds / 60:g}minx{B}"
        roofline = make_roofline(by, args.precision, args.steps, wkey)
        if roofline and dominant:  # the share among ALL kernels comes from the fully bracketed warm-up steps
            roofline["share_of_kernel_time"] = round(next(s_["ms"] for s_ in table if s_["name"] == dominant) / sum(s_["ms"] for s_ in table), 3)
            roofline["events"] = (f"timed region: hipEvents around the {roofline['launches_per_step']:g} launches per step of this bucket only; "
                                  f"`kernels` and share_of_kernel_time: {table_steps} further steps after it, every launch bracketed")
        whole = flops_per_clip(T) * B * world * args.steps / elapsed / 1e12
        result = {
            "metric": "audio frames/sec SpeechT5-base encoder, 30s×bs32 @1/2/4/8 GPU; embed L2 vs HF",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f16x3": "f16x3 (fp16 hi+lo operands, 3 MFMAs per product, fp32 accumulate)", "f32": "f32",
                      "f16x2": "f16x2 (weights rounded to fp16, activations hi+lo, 2 MFMAs per product; opt-in, ~9e-4: at the 1e-3 bar, no margin)"}[args.precision],
            "precision": args.precision, "range_policy": enc.range_policy, "data": "synthetic",
            "config": {"workload": workload_label(args.clip_seconds, B),
                       "cpu_baseline_sample": "skipped" if (args.no_cpu_baseline or world > 1) else
                                              f"{max(1, min(args.cpu_sample_clips, B))} of the {B} clips, as one batch",
                       "clip_seconds": args.clip_seconds, "batch_per_gpu": B, "global_batch": B * world, "frames_per_clip": T,
                       "parallelism": f"dp{world}",
                       "collective": "all_gather(embeddings)" if world > 1 else
                                     ("all_gather(embeddings), forced in a process group of one rank and checked" if args.force_collective else "none")},
            "whole_path_tflops": round(whole, 2),
            "roofline": roofline, "kernels": kernels,
        }
        if multi is not None:
            result["multi_gpu"] = multi
        # the other precision modes, short runs (3 steps), for reference: the exact-fp32 mode ("alt_precision") and the opt-in
        # two-term mode ("opt_in_precision": weights rounded to fp16, ~9e-4 instead of ~1e-6 -- never what `value` reports)
        if world == 1 and not args.no_alt:
            for key, alt in (("alt_precision", "f32" if args.precision != "f32" else "f16x3"),
                             ("opt_in_precision", "f16x2" if args.precision != "f16x2" else "f16x3")):
                enc.precision = alt
                enc(input_values=x, attention_mask=m)
                torch.cuda.synchronize()
                enc.set_profiling(True)
                enc.profile_reset()
                t1 = time.perf_counter()
                for _ in range(3):
                    enc(input_values=x, attention_mask=m)
                torch.cuda.synchronize()
                ealt = time.perf_counter() - t1
                st_alt = {s_["name"]: s_ for s_ in enc.profile_read()}
                enc.set_profiling(False)
                enc.precision = args.precision
                result[key] = {"precision": alt, "value": round(B * T * 3 / ealt, 1), "unit": "frames/s",
                               "ms_per_step": round(ealt / 3 * 1e3, 3), "steps": 3, "roofline": make_roofline(st_alt, alt, 3, wkey),
                               "kernel_ms_per_step": {k_: round(v_["ms"] / 3, 3) for k_, v_ in st_alt.items()}}
        # The timed region above runs with per-kernel HIP events, which keep loco_forward on ONE stream.  Without them the
        # library's default for big batches is two half-batches on two streams (bit-identical output, loco_set_streams):
        # the same workload, un-profiled, for reference.  `value` stays the single-stream figure the roofline belongs to.
        if world == 1 and not args.no_two_streams:
            legs = {}
            default_streams = int(enc.streams)
            for ns in (1, 2):
                enc.streams = ns
                enc(input_values=x, attention_mask=m)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    enc(input_values=x, attention_mask=m)
                torch.cuda.synchronize()
                e2 = time.perf_counter() - t1
                legs[f"streams_{ns}"] = {"value": round(B * T * args.steps / e2, 1), "ms_per_step": round(e2 / args.steps * 1e3, 3)}
            enc.streams = default_streams
            result["two_streams"] = {"unit": "frames/s", "steps": args.steps, "library_default_streams": default_streams, **legs,
                                     "note": "un-profiled forward passes of the same batch, same process; no all-gather (N = 1)"}
        # The reference's own operating point -- batch_size = 2 in corpus order (…base…py:67-68), SLURP-like 2-6 s utterances -- on the
        # packed forward (loco_forward_packed: G of those batches per launch sequence, every clip keeps its own batch's padded length;
        # tools/packed_bench.py is the long form).  Reported beside the headline, never instead of it.
        if world == 1 and not args.no_packed and abs(args.clip_seconds - CLIP_SECONDS) < 1e-9:
            npairs, G = 256, 32
            lens = la.synth.mixed_lengths(2 * npairs, 6 * 16000, min_fraction=2.0 / 6.0)
            fe = la.SpeechT5FeatureExtractorMI355X()
            order = sorted(range(npairs), key=lambda p_: max(lens[2 * p_], lens[2 * p_ + 1]))  # which batches share a pack changes no embedding
            packs, kept = [], 0
            for g0 in range(0, npairs, G):
                groups = [[la.synth.clip(20000 + 2 * p_ + j, lens[2 * p_ + j]) for j in (0, 1)] for p_ in order[g0:g0 + G]]
                packs.append(fe.pack_clips(groups, dev, la.synth.conv_out_length))
                kept += sum(nb * t for (_, nb, t) in packs[-1].spans)
            enc.set_inflight(3)
            default_streams = int(enc.streams)
            enc.streams = 1  # packs in flight fill each other's tails; the two half-batch schedule inside a forward would only add launches
            for pk in packs[:3]:
                enc.forward_packed_async(packed=pk)
            enc.drain()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            tickets = [enc.forward_packed_async(packed=pk) for pk in packs]
            for tk in tickets:
                tk.result()
            torch.cuda.synchronize()
            epk = time.perf_counter() - t1
            enc.streams = default_streams
            rows = sum(pk.wav.shape[0] * la.synth.conv_out_length(pk.wav.shape[1]) for pk in packs)
            result["packed_reference_batches"] = {
                "value": round(kept / epk, 1), "unit": "frames/s", "ms_per_pack": round(epk / len(packs) * 1e3, 3),
                "workload": f"{npairs} reference batches of two synthetic 2-6 s utterances (mean {sum(lens) / len(lens) / 16000:.2f} s), packed {G} batches "
                            f"per launch sequence, sorted by length, 3 packs in flight on one stream each; frames = the frames the reference pickles (each batch's own "
                            f"padded frames, {kept}); rows computed / frames kept = {rows / kept:.3f}",
                "fp32_reruns": sum(int(tk.used_fp32) for tk in tickets)}
            del tickets, packs
        # parity + CPU baseline on a bounded sample of the same workload (rank 0, N = 1 only)
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import speecht5_oracle as oracle  # the checker; never on the product path
            nc = max(1, min(args.cpu_sample_clips, B))
            threads = host_cores()
            torch.set_num_threads(threads)
            xs, ms = x_np[:nc], m_np[:nc]
            oracle.encode(xs[:1, :16000 * 2], None, sd)  # warm-up (thread pool, weight conversion caches)
            reps = []
            ref = None
            for _ in range(max(1, args.cpu_reps)):
                t1 = time.perf_counter()
                ref = oracle.encode(xs, ms, sd)
                reps.append(time.perf_counter() - t1)
            cpu_s = float(np.median(reps))
            yg = enc(input_values=x[:nc], attention_mask=m[:nc]).last_hidden_state.cpu()
            rel = float((yg.double() - ref.double()).norm() / ref.double().norm())
            result["cpu_baseline"] = {"value": round(nc * T / cpu_s, 1), "unit": "frames/s", "cores": torch.get_num_threads(),
                                      "kind": "port",
                                      "sample": f"{nc} clips x {args.clip_seconds:g} s (batch {nc}) of the same synthetic workload, "
                                                f"1 warm-up + median of {len(reps)} timed passes ({', '.join(f'{r:.2f}' for r in reps)} s), "
                                                f"{torch.get_num_threads()} torch threads, torch {torch.__version__} fp32",
                                      "clips": nc, "reps": len(reps)}
            result["embed_rel_l2"] = rel
            result["speedup_vs_cpu_baseline"] = round(value / (nc * T / cpu_s), 1)
        print(json.dumps(result), flush=True)
    if world > 1 or args.force_collective:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
