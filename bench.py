#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json: 5x5 x4-SR LF patches/sec, 32^2 -> 128^2).

Workload at every N: configs[1] "DistgSSR 5x5 x4 inference, batch 32 patches" per GPU; a step is one
forward of the HIP path over one batch of 32 synthetic patches already resident in HBM.  N > 1: one
process per GPU (torchrun env), patches are independent so ranks share no data-path collective
("weak" scaling: 32 patches per GPU); RCCL is used only for the barrier and the max-over-ranks clock.

One JSON line on rank 0.  `roofline` is for the dominant kernel (the per-view 3x3 64->64 conv, 77 % of
DistgSSR FLOPs, MFMA-bound in fp32): algorithmic FLOPs per launch = 2 * 576 * 64 * (B*25*32*32) (SURVEY 8d:
direct-conv 2 x MAC) divided by that op's average duration, measured with hipEvents recorded on the launch
stream around every launch inside the timed region.  The op runs in Winograd F(2x2,3x3) form, which issues
2.25x fewer MFMA flops than that count -- `achieved`/`peak` can therefore exceed 1; `mfma_util` is the physical
matrix-pipe utilisation (flops actually issued / time / peak).  `cpu_baseline` = the numpy oracle (fp32 mode)
timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

A, H, W, S, BATCH = 5, 32, 32, 4, 32
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
FLOP_PER_PATCH = 130.525e9      # SURVEY 8d, DistgSSR x4 forward, 2 x MAC


def cpu_baseline(sd, x1):
    """The oracle's torch-CPU form (stock ATen ops == what the reference's CPU path runs), fp32,
    one patch of the same geometry per call, bounded to ~10-20 s of CPU work."""
    import torch
    from oracle import lfsr_torch_port as T
    threads = torch.get_num_threads()
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    xt = torch.from_numpy(x1)
    T.distgssr_forward(xt, sdt, A, S)   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        T.distgssr_forward(xt, sdt, A, S)
        n += 1
        el = time.perf_counter() - t0
        if el > 12.0 or n >= 40:
            break
    return {"value": n / el, "unit": "patches/s", "cores": threads, "kind": "port",
            "sample": f"{n} x 1 patch (5x5 views of 32x32, x4) through oracle/lfsr_torch_port.py (stock torch CPU ops, fp32), {el:.1f} s"}


def bench_train(args, rank, world, dev, dist):
    """configs[3]: one optimisation step = forward + backward (HIP) + one RCCL all-reduce of the flat 14.3 MB gradient
    bucket + clip + AdamW, batch 8 patches per GPU (weak scaling), fp32."""
    import importlib
    from argparse import Namespace
    import torch
    from lfsr_amd import capi
    from lfsr_amd.synth import synth_input, synth_state_dict
    from lfsr_amd.train_step import broadcast_parameters, train_step
    sys.path.insert(0, capi._HERE)
    M = importlib.import_module("model.SR.DistgSSR")
    sys.path.remove(capi._HERE)
    Bt = 8
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"]["DistgSSR"]["full"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["spec"]], seed=0)
    net = M.get_model(Namespace(angRes_in=A, angRes_out=A, scale_factor=S))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(dev).train()
    broadcast_parameters(net)
    crit = M.get_loss(None)
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=1e-4)
    x = torch.from_numpy(synth_input((Bt, 1, A * H, A * W), seed=1 + rank)).to(dev)
    y = torch.from_numpy(synth_input((Bt, 1, A * H * S, A * W * S), seed=100 + rank)).to(dev)
    for _ in range(args.warmup):
        train_step(net, crit, opt, x, y)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = train_step(net, crit, opt, x, y)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "5x5 x4-SR LF patches/sec (32^2->128^2), DistgSSR training step (fwd+bwd+allreduce+AdamW)",
            "value": world * Bt * args.steps / el, "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "loss": float(loss),
            "config": {"workload": "configs[3]: DistgSSR 5x5 x4 training, batch 8 per GPU, data-parallel, one RCCL all-reduce of the flat gradient bucket",
                       "batch_per_gpu": Bt, "parallelism": f"dp{world}"},
            "model_tflops": 3 * FLOP_PER_PATCH * world * Bt * args.steps / el / 1e12}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["infer", "train"], default="infer",
                    help="infer = configs[1] (headline, default); train = configs[3]: DistgSSR x4 fp32 train step, batch 8 per GPU, RCCL bucket all-reduce")
    args = ap.parse_args()

    import numpy as np
    import torch
    from lfsr_amd import capi
    from lfsr_amd.synth import synth_input, synth_state_dict

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # rehearsal knobs for a one-GPU box (never set by the driver): all ranks on device 0, gloo instead of RCCL
    if os.environ.get("LFSR_BENCH_ONE_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("LFSR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.workload == "train":
        return bench_train(args, rank, world, dev, dist)
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"]["DistgSSR"]["full"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["spec"]], seed=0)
    rt = capi.DistgSSRRuntime(A, S)
    rt.load_state([(k, torch.from_numpy(v).to(dev)) for k, v in sd.items()], dev)
    x_np = synth_input((BATCH, 1, A * H, A * W), seed=1 + rank)          # each rank its own shard of patches
    x = torch.from_numpy(x_np).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        y = rt.forward(x)
    torch.cuda.synchronize()
    barrier()
    # live instrumentation inside the timed region: hipEvents on the launch stream around every 4th 3x3 conv op (the roofline
    # kernel; 53 ops per forward, so every layer is sampled once in four steps) and nothing else -- events around all six operator
    # classes cost 0.8 ms per step (tools/prof_overhead.py), around every conv 0.4-0.7 ms; the per-class breakdown comes from a
    # separate, untimed pass below
    rt.profile(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = rt.forward(x)
    torch.cuda.synchronize()
    barrier()
    el = time.perf_counter() - t0
    prof = rt.profile_read()
    rt.profile(1)
    nb = 3
    for _ in range(nb):
        rt.forward(x)
    torch.cuda.synchronize()
    prof_all = rt.profile_read()
    rt.profile(0)
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        assert torch.isfinite(y).all()
        conv_ms, conv_n = prof["conv3x3"]
        conv_avg_ms = conv_ms / max(conv_n, 1)
        M = BATCH * A * A * H * W
        conv_flop = 2.0 * 576 * 64 * M
        ach = conv_flop / (conv_avg_ms * 1e-3) / 1e12
        sel = os.environ.get("LFSR_CONV3X3", "")
        direct = sel[:1] in ("h", "g")
        wino2 = sel[:5] == "wino2"
        # flops the MFMA pipe actually executes per op: per 8 x 32-pixel tile F(4x4,3x3) runs 36 position-GEMMs of 16 tiles x 64 x 64
        # (2.25 multiplies per output and channel pair), F(2x2,3x3) 16 of 64 tiles x 64 x 64 (4), the direct form 9 taps (9)
        exec_flop = conv_flop if direct else conv_flop * (4.0 / 9.0 if wino2 else 2.25 / 9.0)
        kname = ("k_conv3x3_halo (direct 9-tap halo-tile kernel + channel-split tail launch)" if direct else
                 "k_conv3x3_wino (Winograd F(2x2,3x3): persistent 8x32 tiles, 16 position-GEMMs on fp32 MFMA 32x32x2, in-place halo streaming)" if wino2 else
                 "k_conv3x3_wino4 (per-view 3x3 64->64 in Winograd F(4x4,3x3) form: persistent 8x32-pixel tiles, 36 position-GEMMs on fp32 MFMA "
                 "16x16x4 with the inverse transform in registers; 4 MFMA consumer waves + 4 producer waves per CU, V through LDS, U streamed from L2)")
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_conv3x3.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        line = {
            "metric": "5x5 x4-SR LF patches/sec (32^2->128^2), DistgSSR inference",
            "value": world * BATCH * args.steps / el,
            "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: DistgSSR 5x5 x4 inference, batch 32 patches per GPU (5x5 views of 32x32 -> 128x128)",
                       "batch_per_gpu": BATCH, "parallelism": f"patch-sharded x{world}, no data-path collective",
                       "weights": "synthetic U(-1/sqrt(fan_in), 1/sqrt(fan_in)), numpy PCG64 seed 0"},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": kname + "; duration = one conv op",
                         "flop_per_launch": conv_flop, "executed_flop_per_launch": exec_flop,
                         "mfma_util": exec_flop / (conv_avg_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                         "note": "achieved = algorithmic direct-conv flops (SURVEY 8d) / time; the Winograd F(4x4,3x3) form issues 4x fewer MFMA "
                                 "flops (F(2x2,3x3): 2.25x), so frac may exceed 1 -- mfma_util is the matrix-pipe utilisation",
                         "avg_launch_ms": conv_avg_ms, "launches_timed": conv_n},
            "model_tflops": FLOP_PER_PATCH * world * BATCH * args.steps / el / 1e12,
            "kernel_ms_per_step": {k: v[0] / nb for k, v in prof_all.items()},
            "kernel_ms_per_step_note": f"separate untimed pass of {nb} steps with events around every operator class",
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, x_np[:1])
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
