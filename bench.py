#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json: 5x5 x4-SR LF patches/sec, 32^2 -> 128^2).

Workload at every N (default): configs[1] "DistgSSR 5x5 x4 inference, batch 32 patches" per GPU; a step is one forward of the
HIP path over one batch of 32 synthetic patches already resident in HBM.  N > 1: one process per GPU; patches are independent
so ranks share no data-path collective ("weak" scaling: 32 patches per GPU); RCCL carries only the barrier and the
max-over-ranks clock.  Other workloads (`--workload train|epit|lft`) are configs[3], [2], [4] on the same harness.

Launch forms (both give the same ranks):
  * `python bench.py --gpus N ...`                       -- the parent spawns its N ranks itself as FRESH child processes
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment) before it has imported
    torch or touched the GPU, relays rank 0's JSON line and exits with the worst child status;
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`   -- torchrun provides the environment.

One JSON line on rank 0.
  roofline          the dominant kernel (per-view 3x3 64->64 conv, 77 % of DistgSSR FLOPs, fp32-MFMA-bound): `achieved` = the
                    flops the op EXECUTES on the matrix pipe per launch (Winograd F(4x4,3x3): 36 position-GEMMs of
                    [16 tiles x 64] x [64 x 64] per 8x32-pixel tile = 2.25/9 of the direct 2 x 576 x 64 x pixels of SURVEY 8d)
                    / that op's average duration, measured with hipEvents on the launch stream inside the timed region;
                    `frac` = max(MFMA floor, HBM floor) / duration on that executed work -- a true roofline fraction (< 1).
                    The direct-conv-equivalent rate and the algorithmic speed-up over a direct conv are separate fields.
  roofline_classes  one line per other kernel class of the forward (and the stand-alone SAI<->MacPI / PixelShuffle kernels),
                    algorithmic bytes or flops / hipEvent time, against 8 TB/s (and the 6.3 TB/s a float4 copy reaches) or the
                    fp32 MFMA peak; taken in a separate untimed pass (events around every class cost ~0.8 ms per step).
  other_workloads   (N = 1) configs[2] EPIT B = 8, configs[3] DistgSSR training step B = 8, configs[4] LFT 64-patch scene, run after the headline timing:
                    value, ms_per_step, arithmetic, the dominant operator's in-run duration (event hooks of the library) and its executed-work
                    roofline fraction; EPIT / LFT with their GEMMs on the three-term bf16 pipe and, beside it, with every GEMM on fp32 MFMA.
  cpu_baseline      SURVEY 8d: oracle/lfsr_torch_port.py (stock torch CPU ops = what the reference's CPU path runs), fp32,
                    B = 4 chunks, thread count = a short sweep around the physical cores this process may use; plus one line
                    each for configs 1, 3, 5 (rank 0, N = 1 only; bounded to ~30-40 s).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

A, H, W, S, BATCH = 5, 32, 32, 4, 32
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 dense peak
HBM_PEAK_TBPS = 8.0             # MI355X_MICROARCH.md: HBM3E spec; 6.29 TB/s is what a float4 copy reaches
HBM_COPY_TBPS = 6.29
FLOP_PER_PATCH = {"distgssr": 130.525e9, "epit": 148.9e9, "lft": 62.4e9}   # SURVEY 8d (EPIT / LFT: windowed accounting -- the kernels visit valid keys only)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-check", action="store_true",
                    help="skip the B = 1 re-runs of the batch-independence check (profiling passes: keeps every launch of a kernel the same size)")
    ap.add_argument("--rank-timeout", type=float, default=900.0, help="--gpus N self-launch: seconds after which a still-running rank is killed (with the others) and the run fails")
    ap.add_argument("--arithmetic", choices=["default", "f32"], default="default",
                    help="default: the GEMMs that have an exact-three-term-bf16 form run it (EPI branch, fuse.0, the transformers' linears / FFN / tail); "
                         "f32: every GEMM on fp32 MFMA (lfsr_set_arithmetic)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the configs[2] / [3] / [4] lines (EPIT, training step, LFT scene) of the default run")
    ap.add_argument("--scene-size", type=int, default=128, help="--workload lft: the scene is 5x5 views of SIZE x SIZE (128 -> 64 patches, the default; 512 -> 1024 "
                    "patches, i.e. 128 per GPU at 8 GPUs: the strong-scaling leg with every GPU's convs at full tile rounds)")
    ap.add_argument("--batch", type=int, default=0, help="patches per GPU (default: 32 infer, 8 train / epit / lft)")
    ap.add_argument("--workload", choices=["infer", "train", "epit", "lft"], default="infer",
                    help="infer = configs[1] (headline, default); train = configs[3]: DistgSSR x4 fp32 train step, batch 8 per GPU, RCCL bucket "
                         "all-reduce; epit = configs[2]; lft = configs[4] (full scene through LFdivide / LFintegrate, patches sharded)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# parent: spawn the ranks (no torch, no GPU in this process)
# ------------------------------------------------------------------------------------------------------------------

def launch_ranks(args):
    """One FRESH child per rank (this process has not imported torch and never touches a GPU).  The parent polls: the first child that fails, or a
    child still running after --rank-timeout seconds, takes the others down with it (terminate, then kill) and the parent exits non-zero, so one
    hung rank cannot hold the caller's lease.  Rank 0's stdout goes through a temporary file (no pipe to fill up)."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + args.rank_timeout
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = f"rank {bad[0]} exited with status {rcs[bad[0]]}"
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            failed = f"rank(s) {[r for r, rc in enumerate(rcs) if rc is None]} still running after {args.rank_timeout} s"
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        sys.stderr.write(f"bench.py: {failed}; the other ranks were stopped\n")
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    return 1 if failed else 0


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------------------------

def physical_cores():
    """Physical cores this process may run on: unique (package, core) pairs of /proc/cpuinfo restricted to the affinity mask."""
    allowed = os.sched_getaffinity(0)
    cores, cur, phys, core = set(), None, 0, None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cur = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
                if cur in allowed:
                    cores.add((phys, core))
    except OSError:
        pass
    return max(1, len(cores) if cores else len(allowed)), len(allowed)


def _time_cpu(fn, budget_s, max_iter):
    fn()   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= max_iter:
            return n, el


def cpu_baseline(sd_np, x_np):
    import numpy as np
    import torch
    from oracle import lfsr_torch_port as T
    from lfsr_amd.synth import synth_input, synth_state_dict
    phys, logical = physical_cores()
    sdt = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    x4 = torch.from_numpy(np.ascontiguousarray(x_np[:4]))
    sweep = []
    # short sweep at and below the physical-core count (a GPU box hands a job a CPU share well under what the affinity mask shows, and
    # oversubscribed oneDNN threads collapse: 256 threads ran 40x slower than 64 on the round-2 box); a candidate whose warm-up already
    # takes > 8 s is not timed further
    cands = sorted({c for c in (8, 16, 32, 64, min(phys, 64)) if c <= phys})
    for nt in cands:
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        T.distgssr_forward(x4, sdt, A, S)
        warm = time.perf_counter() - t0
        if warm > 8.0:
            sweep.append({"threads": nt, "patches_per_s": 4 / warm, "iters": 0, "note": "warm-up only (slow)"})
            continue
        n, t0 = 0, time.perf_counter()
        while True:
            T.distgssr_forward(x4, sdt, A, S)
            n += 1
            el = time.perf_counter() - t0
            if el > 3.0 or n >= 3:
                break
        sweep.append({"threads": nt, "patches_per_s": 4 * n / el, "iters": n})
    best = max(sweep, key=lambda r: r["patches_per_s"])
    torch.set_num_threads(best["threads"])
    meta_all = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"]
    others = []
    for cfg, name, fn, s, flop in (("configs[0]: LF_InterNet 5x5 x2, one 32x32 patch", "LF_InterNet", T.internet_forward, 2, 84.653e9),
                                   ("configs[2]: EPIT 5x5 x4, B = 1", "EPIT", T.epit_forward, 4, 162.611e9),
                                   ("configs[4]: LFT 5x5 x4, B = 1 patch of the LFdivide list", "LFT", T.lft_forward, 4, 114.797e9)):
        spec = meta_all[name]["full"]["spec"]
        sdo = {k: torch.from_numpy(v) for k, v in synth_state_dict([(k, tuple(sh)) for k, sh in spec], seed=0).items()}
        xo = torch.from_numpy(synth_input((1, 1, A * H, A * W), seed=1))
        n, el = _time_cpu(lambda: fn(xo, sdo, A, s), 3.0, 3)
        others.append({"config": cfg, "value": n / el, "unit": "patches/s", "threads": best["threads"],
                       "sample": f"{n} x 1 patch through oracle/lfsr_torch_port.py:{fn.__name__}, {el:.1f} s", "reference_dense_gflop_per_patch": flop / 1e9})
    return {"value": best["patches_per_s"], "unit": "patches/s", "cores": best["threads"], "kind": "port",
            "sample": f"{best['iters']} x 4 patches (B = 4 chunks of the B = 32 batch; 5x5 views of 32x32, x4) through "
                      "oracle/lfsr_torch_port.py:distgssr_forward (stock torch CPU ops, fp32)",
            "physical_cores_available": phys, "logical_cpus_available": logical, "thread_sweep": sweep, "other_configs": others}


# ------------------------------------------------------------------------------------------------------------------
# rank body
# ------------------------------------------------------------------------------------------------------------------

def init_rank(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    return rank, world, dev, dist


def timed_loop(step, args, dev, dist):
    """W warm-ups, then exactly K steps bracketed by barrier + synchronize; returns the max-over-ranks seconds, the last result and (N > 1) what the
    ranks themselves saw of the job: the backend, the number of ranks a sum of ones over that backend returns, every rank's own step time and device."""
    import torch
    r = None
    for _ in range(args.warmup):
        r = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    seen = None
    if dist is not None:
        mine = el
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)                       # every rank that takes part in the collective adds one
        backend = dist.get_backend()
        xdev = dev if backend == "nccl" else torch.device("cpu")          # (gloo gathers host tensors only)
        prop = torch.cuda.get_device_properties(dev)
        uid = str(getattr(prop, "uuid", ""))[:40]
        rec = torch.zeros(3 + 40, dtype=torch.float64, device=xdev)
        rec[0], rec[1], rec[2] = mine / max(args.steps, 1) * 1e3, float(dev.index), float(os.getpid())
        if uid:
            rec[3:3 + len(uid)] = torch.tensor([float(ord(ch)) for ch in uid], dtype=torch.float64)
        allrec = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
        dist.all_gather(allrec, rec)
        seen = {"backend": backend + (" (RCCL)" if backend == "nccl" else ""), "ranks_seen": int(round(float(ones.item()))),
                "per_rank_ms_per_step": [float(a[0]) for a in allrec],
                "per_rank_device": [{"cuda_index": int(a[1]), "pid": int(a[2]), "name": prop.name,
                                     "uuid": "".join(chr(int(v)) for v in a[3:].tolist() if v > 0) or None} for a in allrec]}
    return el, r, seen


def finish(dist):
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def base_line(args, world, el, patches_per_step, metric, workload, extra_cfg, seen=None):
    line = _base_line(args, world, el, patches_per_step, metric, workload, extra_cfg)
    if seen is not None:      # N > 1: what the ranks saw -- the collective's own count of participants, every rank's clock and device
        line["multi_gpu"] = seen
    return line


def _base_line(args, world, el, patches_per_step, metric, workload, extra_cfg):
    return {"metric": metric, "value": world * patches_per_step * args.steps / el, "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict({"workload": workload, "batch_per_gpu": patches_per_step}, **extra_cfg)}


def train_arithmetic(all_f32):
    """(dtype, per-operator arithmetic) of the DistgSSR training step under the library's arithmetic selection (lfsr_set_arithmetic): by default the EPI branch's
    forward, fuse.0's forward and fuse.0's data gradient run with fp32 operands carried as three exact bf16 terms; everything else is fp32 MFMA."""
    if all_f32:
        return "f32", {k: "f32" for k in ("conv3x3", "conv3x3_dgrad", "conv3x3_wgrad", "epiconv", "fuse.0", "fuse.0_dgrad", "angconv", "branch gradients", "head")}
    return ("f32 (EPI branch forward, fuse.0 forward and fuse.0 data gradient: fp32 operands as three exact bf16 terms on the bf16 MFMA pipe, fp32 accumulation; "
            "3x3 convs and their gradients, angular branch, the branches' gradients, init, head: fp32 MFMA)",
            {"conv3x3": "f32", "conv3x3_dgrad": "f32", "conv3x3_wgrad": "f32", "epiconv": "bf16x3", "fuse.0": "bf16x3", "fuse.0_dgrad": "bf16x3", "angconv": "f32",
             "branch gradients": "f32", "head": "f32"})


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
    Bt = args.batch or 8
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"]["DistgSSR"]["full"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["spec"]], seed=0)
    net = M.get_model(Namespace(angRes_in=A, angRes_out=A, scale_factor=S))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(dev).train()
    broadcast_parameters(net)
    crit = M.get_loss(None)
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=1e-4, fused=True)   # (one multi-tensor launch per step instead of ~30 foreach launches)
    x = torch.from_numpy(synth_input((Bt, 1, A * H, A * W), seed=1 + rank)).to(dev)
    y = torch.from_numpy(synth_input((Bt, 1, A * H * S, A * W * S), seed=100 + rank)).to(dev)
    el, (loss, _), seen = timed_loop(lambda: train_step(net, crit, opt, x, y), args, dev, dist)
    if rank == 0:
        line = base_line(args, world, el, Bt, "5x5 x4-SR LF patches/sec (32^2->128^2), DistgSSR training step (fwd+bwd+allreduce+AdamW)",
                         "configs[3]: DistgSSR 5x5 x4 training, batch 8 per GPU, data-parallel, one RCCL all-reduce of the flat gradient bucket",
                         {"parallelism": f"dp{world}"}, seen)
        line["dtype"], line["config"]["gemm_arithmetic"] = train_arithmetic(args.arithmetic == "f32")
        line["loss"] = float(loss)
        line["model_tflops"] = 3 * FLOP_PER_PATCH["distgssr"] * world * Bt * args.steps / el / 1e12
        print(json.dumps(line), flush=True)
    finish(dist)


def bench_model(args, rank, world, dev, dist):
    """configs[2] (EPIT, a batch of patches per GPU) / configs[4] (LFT, one full scene per step: LFdivide -> sharded batched forward ->
    all-gather -> LFintegrate; every rank holds the scene, as the reference's test() does)."""
    import torch
    from lfsr_amd import capi
    from lfsr_amd.synth import synth_input, synth_state_dict
    name = args.workload
    key = {"epit": "EPIT", "lft": "LFT"}[name]
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"][key]["full"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["spec"]], seed=0)
    rt = capi.ModelRuntime(name, A, S, 5 if name == "epit" else 4, 64)
    rt.load_state([(k, torch.from_numpy(v).to(dev)) for k, v in sd.items()], dev)
    if name == "epit":
        Bm = args.batch or 8
        x = torch.from_numpy(synth_input((Bm, 1, A * H, A * W), seed=1 + rank)).to(dev)
        el, y, seen = timed_loop(lambda: rt.forward(x), args, dev, dist)
        per_step, wl = Bm, f"configs[2]: EPIT 5x5 x4 inference, batch {Bm} patches per GPU"
        par = f"patch-sharded x{world}, no data-path collective"
    else:
        from lfsr_amd.dispatch import sr_scene
        sz = args.scene_size
        scene = torch.from_numpy(synth_input((A * sz, A * sz), seed=3)).to(dev)      # 5x5x128x128 -> 64 patches -> (5,5,512,512); 5x5x512x512 -> 1024 patches
        n_patches = int(capi.lf_divide(scene, A, 32, 16).shape[:2].numel())
        el, y, seen = timed_loop(lambda: sr_scene(lambda t, info=None: rt.forward(t), scene, A, S, minibatch=args.batch or 32, dst=0), args, dev, dist)
        per_step, wl = float(n_patches) / world, f"configs[4]: LFT 5x5 x4 full-scene inference (5x5x{sz}x{sz} -> {n_patches} patches via LFdivide / LFintegrate)"
        par = (f"{n_patches} patches sharded over {world} rank(s); each rank crops its SR patches to the tiles LFintegrate keeps, one gather of the tiles "
               f"({n_patches * A * A * 64 * 64 * 4 / 1e6:.0f} MB per scene in total) to rank 0, which places them")
    if rank == 0:
        assert torch.isfinite(y).all()
        line = base_line(args, world, el, per_step, f"5x5 x4-SR LF patches/sec (32^2->128^2), {key} inference", wl, {"parallelism": par}, seen)
        if name == "lft":
            line["scaling"] = "strong"
            line["config"]["scene"] = {"size": [A, A, args.scene_size, args.scene_size], "patches": n_patches}
        line["model_tflops_windowed"] = FLOP_PER_PATCH[name] * line["value"] / 1e12
        # the arithmetic the transformer GEMMs run in (rowgemm_b3.hip / ffn_b3.hip / up_tail.hip): fp32 operands split EXACTLY into three bf16 terms, six
        # products on the bf16 MFMA pipe, fp32 accumulation -- error against fp64 below the fp32-MFMA kernels' (tools/b3_accuracy.py); LFSR_ROWGEMM=f32
        # LFSR_FFN=f32 LFSR_UPTAIL=v2 select the fp32-MFMA kernels
        lab = "LFSR_LAB" in os.environ      # (the A/B selectors are live only then)
        f32_sel = [args.arithmetic == "f32" or (lab and os.environ.get("LFSR_ROWGEMM", "")[:1] in ("f", "1")), args.arithmetic == "f32" or (lab and os.environ.get("LFSR_FFN", "")[:1] == "f"),
                   args.arithmetic == "f32" or (lab and os.environ.get("LFSR_UPTAIL", "")[:1] == "v")]
        line["dtype"] = "f32" if all(f32_sel) else "f32 (linear / FFN / tail GEMMs: fp32 operands as three exact bf16 terms on the bf16 MFMA pipe, fp32 accumulation; 3x3 convs and attention: fp32 MFMA)"
        line["config"]["gemm_arithmetic"] = {"rowgemm": "f32" if f32_sel[0] else "bf16x3", "ffn": "f32" if f32_sel[1] else "bf16x3", "up_tail": "f32" if f32_sel[2] else "bf16x3"}
        print(json.dumps(line), flush=True)
    finish(dist)


BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (the three-term GEMMs issue six bf16 products per fp32 product)


def _op_floor(op, a, b, npix, f32_gemms):
    """executed work of one launch of an instrumented operator at `npix` token rows -> (bound, floor_us, what) against the gfx950 peaks, or None"""
    if op in ("conv3x3", "conv3x3_dgrad"):       # Winograd F(4x4,3x3): 2.25 / 9 of the direct 2 x 576 x 64 flops per pixel, fp32 MFMA
        m = a * b
        fl, by = 2.0 * 576 * 64 * m * 2.25 / 9.0, 2.0 * m * 64 * 4
        t = max(fl / (FP32_MFMA_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_TBPS * 1e12))
        return "mfma-f32", t * 1e6, "k_conv3x3_wino4: 36 position-GEMMs per 8x32-pixel tile"
    if op == "conv3x3_wgrad":                    # adjoint of F(2x2,3x3): 16 products per 2x2 outputs and channel pair (4 MACs per pixel and pair instead of 9), fp32 MFMA
        m = a * b
        fl, by = 2.0 * 4 * 64 * 64 * m, 2.0 * m * 64 * 4
        t = max(fl / (FP32_MFMA_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_TBPS * 1e12))
        return "mfma-f32", t * 1e6, "k_wgrad_conv3_wino: 16 position-GEMMs with the tile index as the MFMA K dimension (+ k_wgrad_reduce)"
    if op in ("linear", "linear_ln"):
        fl, by = 2.0 * npix * a * b, npix * (a + b) * 4.0
        tm = fl / (FP32_MFMA_PEAK_TFLOPS * 1e12) if f32_gemms else 6.0 * fl / (BF16_MFMA_PEAK_TFLOPS * 1e12)
        th = by / (HBM_PEAK_TBPS * 1e12)
        return ("hbm" if th >= tm else ("mfma-f32" if f32_gemms else "mfma-bf16x3")), max(tm, th) * 1e6, f"row-GEMM K = {a}, N = {b}"
    if op == "ffn":
        fl, by = 4.0 * npix * a * b, 2.0 * npix * a * 4.0
        tm = fl / (FP32_MFMA_PEAK_TFLOPS * 1e12) if f32_gemms else 6.0 * fl / (BF16_MFMA_PEAK_TFLOPS * 1e12)
        th = by / (HBM_PEAK_TBPS * 1e12)
        return ("hbm" if th >= tm else ("mfma-f32" if f32_gemms else "mfma-bf16x3")), max(tm, th) * 1e6, f"fused LayerNorm + {a} -> {b} -> {a} feed-forward"
    if op == "window_attn":                      # q | k | v read + o written once, E = 8 heads x head dim; the windowed flops are far below the HBM floor
        by = 4.0 * npix * 8 * a * 4.0
        return "hbm", by / (HBM_PEAK_TBPS * 1e12) * 1e6, f"windowed attention, head dim {a}, {b} tokens per sequence"
    return None


def other_workloads(dev, budget_steps=(20, 10, 8)):
    """configs[2], [3], [4] on this GPU after the headline (rank 0, N = 1): EPIT B = 8 and the LFT 64-patch scene with their GEMMs on the three-term bf16
    pipe AND on fp32 MFMA, the DistgSSR training step at B = 8; each with its dominant operator's in-run duration (the library's event hooks,
    capi.op_profile, in a separate untimed pass of 2 steps) and that operator's executed-work roofline fraction.  Template: check_efficiency_official.py:306-330."""
    import importlib
    from argparse import Namespace
    import torch
    from lfsr_amd import capi
    from lfsr_amd.synth import synth_input, synth_state_dict
    meta_all = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"]
    out = []

    def timed(step, warm, steps):
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def dominant(step, npix, f32_gemms, nprof=2):
        step()
        torch.cuda.synchronize()
        capi.op_profile(True)
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        tab = capi.op_profile_read()
        capi.op_profile(False)
        nested = {"epiconv_hv"}                  # (outer hooks that only wrap another hooked launch)
        tab = {k: v for k, v in tab.items() if k[0] not in nested}
        if not tab:
            return None
        tot = sum(ms for ms, _ in tab.values())
        (op, a, b), (ms, n) = max(tab.items(), key=lambda kv: kv[1][0])
        e = {"operator": op, "tags": [a, b], "avg_launch_us": ms / n * 1e3, "launches_per_step": n // nprof, "ms_per_step": ms / nprof,
             "share_of_hooked_time": ms / tot, "hooked_ms_per_step": tot / nprof}
        fl = _op_floor(op, a, b, npix, f32_gemms)
        if fl:
            e.update(bound=fl[0], floor_us=fl[1], frac=fl[1] / (ms / n * 1e3), kernel=fl[2])
        e["by_operator_ms_per_step"] = {f"{k[0]}({k[1]},{k[2]})": v[0] / nprof for k, v in sorted(tab.items(), key=lambda kv: -kv[1][0])[:8]}
        return e


    # ---- configs[2] EPIT B = 8 and configs[4] LFT scene: both arithmetic selections (read when the runtime is built and at every launch) ----
    for name, key, steps in (("epit", "EPIT", budget_steps[0]), ("lft", "LFT", budget_steps[2])):
        sd = synth_state_dict([(k, tuple(sh)) for k, sh in meta_all[key]["full"]["spec"]], seed=0)
        lines, b1 = {}, {}
        for arith in ("bf16x3", "f32"):
            capi.set_arithmetic(capi.ARITH_F32 if arith == "f32" else capi.ARITH_DEFAULT)
            rt = capi.ModelRuntime(name, A, S, 5 if name == "epit" else 4, 64)
            rt.load_state([(k, torch.from_numpy(v).to(dev)) for k, v in sd.items()], dev)
            if name == "epit":
                x = torch.from_numpy(synth_input((8, 1, A * H, A * W), seed=1)).to(dev)
                step, per_step, npix = (lambda: rt.forward(x)), 8, 8 * A * A * H * W
                x1 = x[:1].contiguous()           # B in {1, 8}
                sec1 = timed(lambda: rt.forward(x1), 3, steps)
                b1[arith] = {"value": 1 / sec1, "ms_per_step": sec1 * 1e3}
                if arith == "bf16x3":             # the same forward replayed from a HIP graph: at B = 1 the host's launch rate is part of the step
                    gf = capi.GraphedForward(rt)
                    secg = timed(lambda: gf(x1), 3, steps)
                    b1[arith]["hipgraph_replay"] = {"value": 1 / secg, "ms_per_step": secg * 1e3,
                                                    "note": "capi.GraphedForward: the eager launches captured once per shape (torch.cuda.CUDAGraph = hipGraph), bit-equal output"}
            else:
                from lfsr_amd.dispatch import sr_scene
                scene = torch.from_numpy(synth_input((A * 128, A * 128), seed=3)).to(dev)
                step, per_step, npix = (lambda: sr_scene(lambda t, info=None: rt.forward(t), scene, A, S, minibatch=32)), 64, 32 * A * A * H * W
            sec = timed(step, 3, steps)
            lines[arith] = {"value": per_step / sec, "ms_per_step": sec * 1e3, "dominant": dominant(step, npix, arith == "f32")}
            del rt
        capi.set_arithmetic(capi.ARITH_DEFAULT)
        b3, f32 = lines["bf16x3"], lines["f32"]
        out.append({"config": "configs[2]: EPIT 5x5 x4 inference, batch 8 patches, 1 GPU" if name == "epit" else
                              "configs[4]: LFT 5x5 x4 full-scene inference (5x5x128x128 -> 64 patches via LFdivide / LFintegrate, minibatch 32), 1 GPU",
                    "value": b3["value"], "unit": "patches/s", "ms_per_step": b3["ms_per_step"], "steps": steps, "warmup": 3,
                    "dtype": "f32 (linear / FFN / tail GEMMs: fp32 operands as three exact bf16 terms on the bf16 MFMA pipe, fp32 accumulation; 3x3 convs and attention: fp32 MFMA)",
                    "gemm_arithmetic": "bf16x3", "dominant_kernel": b3["dominant"],
                    "all_fp32_mfma": {"value": f32["value"], "ms_per_step": f32["ms_per_step"], "dtype": "f32", "gemm_arithmetic": "f32", "dominant_kernel": f32["dominant"]},
                    "model_tflops_windowed": FLOP_PER_PATCH[name] * b3["value"] / 1e12})
        if name == "epit":
            out.append({"config": "configs[2]: EPIT 5x5 x4 inference, batch 1 patch, 1 GPU (latency point)", "value": b1["bf16x3"]["value"], "unit": "patches/s",
                        "ms_per_step": b1["bf16x3"]["ms_per_step"], "steps": steps, "warmup": 3, "dtype": out[-1]["dtype"], "gemm_arithmetic": "bf16x3",
                        "hipgraph_replay": b1["bf16x3"].get("hipgraph_replay"),
                        "all_fp32_mfma": dict(b1["f32"], dtype="f32", gemm_arithmetic="f32")})
    # ---- configs[3] DistgSSR training step, B = 8: default arithmetic (EPI branch, fuse.0 and fuse.0's data gradient on the three-term bf16 form) and all fp32 MFMA ----
    sys.path.insert(0, capi._HERE)
    M = importlib.import_module("model.SR.DistgSSR")
    sys.path.remove(capi._HERE)
    from lfsr_amd.train_step import train_step
    sd = synth_state_dict([(k, tuple(sh)) for k, sh in meta_all["DistgSSR"]["full"]["spec"]], seed=0)
    net = M.get_model(Namespace(angRes_in=A, angRes_out=A, scale_factor=S))
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(dev).train()
    crit = M.get_loss(None)
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=1e-4, fused=True)   # (one multi-tensor launch per step instead of ~30 foreach launches)
    x = torch.from_numpy(synth_input((8, 1, A * H, A * W), seed=1)).to(dev)
    y = torch.from_numpy(synth_input((8, 1, A * H * S, A * W * S), seed=100)).to(dev)
    step = lambda: train_step(net, crit, opt, x, y)
    sec = timed(step, 3, budget_steps[1])
    dom = dominant(step, 8 * A * A * H * W, True)
    capi.set_arithmetic(capi.ARITH_F32)
    sec_f32 = timed(step, 2, budget_steps[1])
    capi.set_arithmetic(capi.ARITH_DEFAULT)
    dtype, arith = train_arithmetic(False)
    out.append({"config": "configs[3]: DistgSSR 5x5 x4 training step (fwd + bwd + clip + AdamW; the RCCL bucket all-reduce is a no-op at N = 1), batch 8, 1 GPU",
                "value": 8 / sec, "unit": "patches/s", "ms_per_step": sec * 1e3, "steps": budget_steps[1], "warmup": 3, "dtype": dtype, "gemm_arithmetic": arith,
                "dominant_kernel": dom, "model_tflops": 3 * FLOP_PER_PATCH["distgssr"] * 8 / sec / 1e12,
                "all_fp32_mfma": {"value": 8 / sec_f32, "ms_per_step": sec_f32 * 1e3, "dtype": "f32", "gemm_arithmetic": "f32",
                                  "selection": "lfsr_set_arithmetic(LFSR_ARITH_F32), same process, same weights, timed right after"}})
    del net, opt
    torch.cuda.empty_cache()
    return out


def index_micro(dev):
    """stand-alone a1 / a2 / a3 at the bench geometry (B = 32, C = 64): algorithmic bytes = one read + one write of every element."""
    import torch
    from lfsr_amd import capi
    out = []

    def t_us(fn, n=10):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    x = torch.rand(BATCH, 64, A * H, A * W, device=dev)
    p = torch.rand(2, 64 * S * S, A * H, A * W, device=dev)
    for cls, kern, nbytes, fn in (
            ("a1 SAI2MacPI (32,64,160,160)", "k_sai_macpi", 2 * x.numel() * 4, lambda: capi.sai2macpi(x, A)),
            ("a2 MacPI2SAI (32,64,160,160)", "k_sai_macpi", 2 * x.numel() * 4, lambda: capi.macpi2sai(x, A)),
            ("a3 PixelShuffle(4) (2,1024,160,160)", "k_pixel_shuffle2d", 2 * p.numel() * 4, lambda: capi.pixel_shuffle2d(p, S)),
            ("float4 copy of the a1 bytes (reference point: torch clone)", "copy", 2 * x.numel() * 4, lambda: x.clone())):
        us = t_us(fn)
        out.append({"class": cls, "kernel": kern, "bound": "hbm", "bytes": nbytes, "us": us, "achieved": nbytes / us / 1e6, "unit": "TB/s",
                    "frac": nbytes / us / 1e6 / HBM_PEAK_TBPS, "frac_of_copy": nbytes / us / 1e6 / HBM_COPY_TBPS})
    return out


def bench_infer(args, rank, world, dev, dist):
    import numpy as np
    import torch
    from lfsr_amd import capi
    from lfsr_amd.synth import synth_input, synth_state_dict
    B = args.batch or BATCH
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "models.json")))["models"]["DistgSSR"]["full"]
    sd = synth_state_dict([(k, tuple(s)) for k, s in meta["spec"]], seed=0)
    rt = capi.DistgSSRRuntime(A, S)
    rt.load_state([(k, torch.from_numpy(v).to(dev)) for k, v in sd.items()], dev)
    x_np = synth_input((B, 1, A * H, A * W), seed=1 + rank)          # each rank its own shard of patches
    x = torch.from_numpy(x_np).to(dev)
    for _ in range(args.warmup):
        rt.forward(x)
    # live instrumentation inside the timed region: hipEvents on the launch stream around every 4th 3x3 conv op (the roofline
    # kernel; 53 ops per forward, so every layer is sampled once in four steps) and nothing else
    rt.profile(3)
    el, y, seen = timed_loop(lambda: rt.forward(x), argparse.Namespace(warmup=0, steps=args.steps), dev, dist)
    prof = rt.profile_read()
    # the same timed loop with every GEMM of the forward on fp32 MFMA (lfsr_set_arithmetic(LFSR_ARITH_F32): the F(2,5) fp32 kernel of the EPI branch, fuse.0 on the
    # fp32 row-GEMM): printed beside the headline whenever the headline uses the exact three-term bf16 form for those two operators
    lab = "LFSR_LAB" in os.environ          # (the library's A/B selectors are live only then)
    user_sel = {k: (os.environ.get(k) if lab else None) for k in ("LFSR_EPI", "LFSR_ROWGEMM")}
    b3_ops = [] if args.arithmetic == "f32" else [k for k, v in user_sel.items() if not (v and v[:1] in ("w", "d", "f", "g", "1"))]
    el_f32 = None
    if b3_ops:
        capi.set_arithmetic(capi.ARITH_F32)
        for _ in range(2):
            rt.forward(x)
        el_f32, _, _ = timed_loop(lambda: rt.forward(x), argparse.Namespace(warmup=0, steps=args.steps), dev, dist)
        capi.set_arithmetic(capi.ARITH_DEFAULT)
        rt.forward(x)
    rt.profile(1)
    nb = 3
    for _ in range(nb):
        rt.forward(x)
    torch.cuda.synchronize()
    prof_all = rt.profile_read()
    rt.profile(0)
    if rank == 0:
        assert torch.isfinite(y).all()
        # B = 32 forward == the same patches run one at a time (batch independence of the path; bit-equal expected)
        batch_split_max_abs_diff = None
        if not args.no_split_check:
            y1 = torch.cat([rt.forward(x[i:i + 1]) for i in (0, B - 1)], 0)
            batch_split_max_abs_diff = float((y1 - rt.forward(x)[[0, B - 1]]).abs().max())
        M = B * A * A * H * W
        conv_ms, conv_n = prof["conv3x3"]
        conv_s = conv_ms / max(conv_n, 1) * 1e-3
        direct_flop = 2.0 * 576 * 64 * M
        sel = os.environ.get("LFSR_CONV3X3", "")
        direct, wino2 = sel[:1] in ("h", "g"), sel[:5] == "wino2"
        exec_flop = direct_flop * (1.0 if direct else 4.0 / 9.0 if wino2 else 2.25 / 9.0)
        conv_bytes = (2 + 21.0 / 53.0) * M * 64 * 4            # in + out per launch, + the residual operand on 21 of the 53 ops
        mfma_floor, hbm_floor = exec_flop / (FP32_MFMA_PEAK_TFLOPS * 1e12), conv_bytes / (HBM_PEAK_TBPS * 1e12)
        kname = ("k_conv3x3_halo (direct 9-tap halo-tile kernel)" if direct else
                 "k_conv3x3_wino (Winograd F(2x2,3x3), 16 position-GEMMs on fp32 MFMA 32x32x2)" if wino2 else
                 "k_conv3x3_wino4 (per-view 3x3 64->64 in Winograd F(4x4,3x3) form: 36 position-GEMMs per 8x32-pixel tile on fp32 MFMA 16x16x4)")
        traffic, tsrc = None, None
        for cand in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "pmc_conv3x3.json"):
            pmc = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(pmc):
                j = json.load(open(pmc))
                traffic = j.get("conv3x3", j).get("hbm_bytes_per_launch")
                tsrc = f"profiles/{cand}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH_SIZE x 2 (gfx950 correction); not re-measured in this run"
                break
        line = base_line(args, world, el, B, "5x5 x4-SR LF patches/sec (32^2->128^2), DistgSSR inference",
                         f"configs[1]: DistgSSR 5x5 x4 inference, batch {B} patches per GPU (5x5 views of 32x32 -> 128x128)",
                         {"parallelism": f"patch-sharded x{world}, no data-path collective",
                          "weights": "synthetic U(-1/sqrt(fan_in), 1/sqrt(fan_in)), numpy PCG64 seed 0"}, seen)
        if b3_ops:
            line["dtype"] = ("f32 (EPI branch stage 1 / 2 and fuse.0: fp32 operands as three exact bf16 terms on the bf16 MFMA pipe, six products, fp32 accumulation -- "
                             "error against fp64 not above the fp32-MFMA kernels', tests/test_gpu_b3_accuracy.py; 3x3 convs, angular branch, init, head: fp32 MFMA)")
            line["config"]["gemm_arithmetic"] = {"conv3x3": "f32", "epiconv": "bf16x3" if "LFSR_EPI" in b3_ops else "f32", "fuse.0": "bf16x3" if "LFSR_ROWGEMM" in b3_ops else "f32",
                                                 "angconv": "f32", "head": "f32"}
            line["all_fp32_mfma"] = {"value": world * B * args.steps / el_f32, "unit": "patches/s", "ms_per_step": el_f32 / args.steps * 1e3, "dtype": "f32",
                                     "selection": "lfsr_set_arithmetic(LFSR_ARITH_F32) (same process, same weights, timed right after the headline loop)"}
        line["roofline"] = {
            "bound": "mfma", "achieved": exec_flop / conv_s / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": max(mfma_floor, hbm_floor) / conv_s, "traffic": traffic, "traffic_source": tsrc,
            "kernel": kname + "; duration = one conv op", "executed_flop_per_launch": exec_flop, "algorithmic_bytes_per_launch": conv_bytes,
            "mfma_floor_us": mfma_floor * 1e6, "hbm_floor_us": hbm_floor * 1e6, "avg_launch_us": conv_s * 1e6, "launches_timed": conv_n,
            "direct_conv_flop_per_launch": direct_flop, "direct_equivalent_tflops": direct_flop / conv_s / 1e12,
            "algorithmic_speedup_over_direct_form": direct_flop / exec_flop,
            "note": "frac is on the work the kernel executes (Winograd domain), so it is a true fraction; direct_equivalent_tflops = SURVEY 8d's "
                    "2 x 576 x 64 x pixels / time may exceed the peak and is not a roofline figure"}
        # per-class lines: algorithmic bytes (one read of every input + one write of every output, fp32) and flops per launch at this B
        npx = float(M)
        epi_b3, pw_b3 = "LFSR_EPI" in b3_ops, "LFSR_ROWGEMM" in b3_ops
        cls_spec = {   # class -> (kernel, bound, bytes, flop)
            # executed flops: stage 1 in Winograd F(2,5) form runs 6 products per 2 outputs instead of 10 (x 0.6; LFSR_EPI=direct: x 1.0); in the three-term bf16 form the
            # direct sums run as six bf16 products each (x 6 on the bf16 pipe)
            "epiconv": (("k_epi_b3 (both EPI passes, direct form, fp32 operands as three exact bf16 terms: six bf16 MFMA products per fp32 product; DistgSSR.py:91-97,108)", "mfma-bf16x3",
                         (64 + 64) * npx * 4, 6 * 2 * (0.524e9 + 0.052e9) * B) if epi_b3 else
                        ("k_epi_wino5 (both EPI passes: 1xA^2 conv 64->32 in F(2,5) form, LReLU, 1x1 32->160, LReLU, PixelShuffle1D; DistgSSR.py:91-97,108)", "mfma",
                         (64 + 64) * npx * 4, 2 * (0.524e9 * (1.0 if os.environ.get("LFSR_EPI", "")[:1] == "d" else 0.6) + 0.052e9) * B)),
            "pointwise": ("fuse.0 144->64 + LReLU (DistgSSR.py:98-100,109)" + (" on k_rowgemm_b3 (three-term bf16 operands, 144 of 160 operand columns valid)" if pw_b3 else ""), "hbm",
                          (144 + 64) * npx * 4, 0.472e9 * B),
            "angconv": ("k_ang_fused (AngConv.0 + LReLU + AngConv.2 + LReLU + PixelShuffle(A); DistgSSR.py:84-90)", "hbm", (64 + 16) * npx * 4, 0.065e9 * B),
            "init_conv": ("k_initconv (SAI2MacPI + 3x3 1->64; DistgSSR.py:22,31-32)", "hbm", (1 + 64) * npx * 4, 0.029e9 * B),
            "upsample_head": ("k_head<4> (MacPI2SAI + folded 64->16 1x1 + PixelShuffle(4) + bilinear skip; DistgSSR.py:24-35)", "hbm",
                              (64 + 1 + 16) * npx * 4, 2.0 * 64 * 16 * npx),
        }
        classes = []
        for cls, (kern, bound, nbytes, flop) in cls_spec.items():
            ms, n = prof_all[cls]
            if not n:
                continue
            us = ms / n * 1e3
            e = {"class": cls, "kernel": kern, "bound": bound, "launches_per_step": n // nb, "bytes": nbytes, "flop": flop, "us": us,
                 "TBps": nbytes / us / 1e6, "TFLOPs": flop / us / 1e6}
            if bound == "hbm":
                e.update(achieved=e["TBps"], unit="TB/s", frac=e["TBps"] / HBM_PEAK_TBPS, frac_of_copy=e["TBps"] / HBM_COPY_TBPS)
            elif bound == "mfma-bf16x3":
                e.update(achieved=e["TFLOPs"], unit="TFLOP/s (bf16 products executed)", frac=max(e["TFLOPs"] / BF16_MFMA_PEAK_TFLOPS, e["TBps"] / HBM_PEAK_TBPS),
                         f32_equivalent_TFLOPs=e["TFLOPs"] / 6.0)
            else:
                e.update(achieved=e["TFLOPs"], unit="TFLOP/s", frac=e["TFLOPs"] / FP32_MFMA_PEAK_TFLOPS)
            classes.append(e)
        line["roofline_classes"] = classes + index_micro(dev)
        line["model_tflops"] = FLOP_PER_PATCH["distgssr"] * line["value"] / 1e12
        line["kernel_ms_per_step"] = {k: v[0] / nb for k, v in prof_all.items()}
        line["kernel_ms_per_step_note"] = f"separate untimed pass of {nb} steps with events around every operator class"
        line["batch_split_max_abs_diff"] = batch_split_max_abs_diff
        if world == 1 and not args.no_other_workloads:
            del rt
            torch.cuda.empty_cache()
            line["other_workloads"] = other_workloads(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, x_np)
        print(json.dumps(line), flush=True)
    finish(dist)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))     # before torch is imported or anything touches the GPU
    hook = os.environ.get("LFSR_BENCH_TEST_RANK")     # launcher tests only (tests/test_bench_launcher.py): "ok" | "fail:<rank>" | "hang:<rank>", no torch, no GPU
    if hook and "WORLD_SIZE" in os.environ:
        me = int(os.environ["RANK"])
        kind, _, who = hook.partition(":")
        if kind == "ok":
            if me == 0:
                print(json.dumps({"ok": True, "world": int(os.environ["WORLD_SIZE"]), "port": os.environ["MASTER_PORT"]}), flush=True)
            sys.exit(0)
        if me == int(who):
            if kind == "fail":
                sys.exit(3)
            time.sleep(600)
        time.sleep(600 if kind == "fail" else 0)
        sys.exit(0)
    rank, world, dev, dist = init_rank(args)
    if args.arithmetic == "f32":
        from lfsr_amd import capi
        capi.set_arithmetic(capi.ARITH_F32)
    if args.workload == "train":
        return bench_train(args, rank, world, dev, dist)
    if args.workload in ("epit", "lft"):
        return bench_model(args, rank, world, dev, dist)
    return bench_infer(args, rank, world, dev, dist)


if __name__ == "__main__":
    main()
