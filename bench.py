"""Headline benchmark: try-on frames/sec (forward+backward+Adam) at 256x192, bs=4 per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is the chained warp -> try-on training step of SURVEY.md §8d (C4 at the BASELINE batch size):
WarpModel (GMM) training step, then UnetMaskModel (self-attention, num_attn=2, GELU, L1 + VGG + mask
loss) training step on the detached warped cloth, each with its Adam update; under N > 1 each rank works
on its own 4 frames (weak scaling) and gradients are mean-all-reduced over RCCL.
Inputs are synthetic (seed 420) and resident in HBM before the timed region; weights are random-init.

Rank 0 prints ONE JSON line.  `roofline` is the live HIP-event measurement of the dominant MFMA kernel
(algorithmic FLOPs / summed kernel time inside the timed region); `cpu_baseline` is the oracle (a CPU
restatement of the reference, kind "port") timed on this host's cores on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import shineon_virtual_tryon_amd as pkg  # noqa: E402
from shineon_virtual_tryon_amd.data import synthetic_batch  # noqa: E402
from shineon_virtual_tryon_amd.trainer import GradientAllReducer, broadcast_parameters, init_distributed  # noqa: E402
from shineon_virtual_tryon_amd.unet_mask_model import UnetMaskModel  # noqa: E402
from shineon_virtual_tryon_amd.warp_model import WarpModel  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
KEY_NAMES = [f"{m}_{t}" for m in ("fprop", "dgrad", "wgrad", "gemm")
             for t in ("64x64", "128x64", "64x128", "128x128", "128x128w8", "64x128w8", "thin4", "-")]


def hparams(**kw):
    base = dict(n_frames_total=1, cloth_inputs=["cloth"], is_train=True, ngf=64, grid_size=5, fine_height=256,
                fine_width=192, self_attn=True, num_attn=2, flow_warp=False, activation="gelu", display_count=10 ** 9,
                pen_flow_mask=1.0, lr=1e-4, keep_epochs=5, decay_epochs=5)
    base.update(kw)
    return argparse.Namespace(**base)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup quota, capped at 64
    (one socket's worth; PyTorch's CPU convolutions stop scaling well before that)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(batch_size, iters=2):
    """The oracle (CPU restatement of the reference path) on this host's cores: same chained step."""
    from oracle import shineon_oracle as oracle
    from oracle.procedural import procedural_state_dict, shapes_of

    torch.set_num_threads(usable_cores())
    log(f"cpu_baseline: {torch.get_num_threads()} threads")
    warp_sd = procedural_state_dict(shapes_of(WarpModel(hparams(person_inputs=["agnostic", "cocopose"])).state_dict()))
    unet_sd = procedural_state_dict(shapes_of(UnetMaskModel(hparams(person_inputs=["agnostic", "densepose"])).state_dict()))
    wp = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in warp_sd.items()}
    up = {k: v.clone().requires_grad_(k.startswith("unet.")) for k, v in unet_sd.items()}
    optw = torch.optim.Adam([v for v in wp.values() if v.requires_grad], 1e-4)
    optu = torch.optim.Adam([v for v in up.values() if v.requires_grad], 1e-4)
    batch = synthetic_batch(batch_size, "cpu")
    consts = oracle.tps_constants(256, 192, 5)
    whp = dict(person_inputs=["agnostic", "cocopose"], cloth_inputs=["cloth"])
    uhp = dict(n_frames_total=1, person_inputs=["agnostic", "densepose"], cloth_inputs=["cloth"], self_attn=True,
               num_attn=2, activation="gelu", flow_warp=False)

    def step():
        optw.zero_grad()
        out = oracle.warp_losses(wp, batch, whp, consts)
        out["loss/G"].backward()
        optw.step()
        b2 = dict(batch)
        b2["cloth"] = out["warped_cloth"].detach()
        optu.zero_grad()
        oracle.unet_mask_losses(up, b2, uhp)["loss/G"].backward()
        optu.step()

    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    log(f"cpu_baseline: warm-up step {warm:.2f} s")
    if warm > 15.0:  # bounded sample: keep the default run within minutes on a slow host
        dt, note = warm, "1 chained step (the warm-up itself; host too slow for more)"
    else:
        t0 = time.perf_counter()
        for _ in range(iters):
            step()
        dt = (time.perf_counter() - t0) / iters
        note = f"1 warm-up + {iters} timed chained steps"
    return {"value": batch_size / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{note} (WarpModel + UnetMaskModel fwd+bwd+Adam, bs={batch_size}, 256x192) "
                      f"with PyTorch CPU fp32, {dt * 1e3:.0f} ms/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="frames per GPU (BASELINE: 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly (no hipGraph replay)")
    ap.add_argument("--cpu-iters", type=int, default=2)
    ap.add_argument("--serial-wgrad", action="store_true", help="no side stream for weight gradients")
    ap.add_argument("--no-pipeline", action="store_true", help="two sequential graphs (warp, then try-on) instead of the "
                    "two-stream schedule that overlaps the warp backward pass with the try-on stage")
    ap.add_argument("--plans", default="", help="file with measured igemm plans: loaded if present (skips the "
                    "one-off tuning sweep, e.g. under a profiler), written back at the end")
    args = ap.parse_args()

    rank, world = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    L = pkg.lib()
    if args.plans and os.path.exists(args.plans):
        log(f"loaded {L.so_igemm_plans_load(args.plans.encode())} igemm plans from {args.plans}")

    torch.manual_seed(420)
    warp = WarpModel(hparams(person_inputs=["agnostic", "cocopose"])).to(dev).train()
    unet = UnetMaskModel(hparams(person_inputs=["agnostic", "densepose"])).to(dev).train()
    warp.global_step = unet.global_step = 1
    broadcast_parameters(warp)
    broadcast_parameters(unet)
    (optw,), _ = warp.configure_optimizers()
    (optu,), _ = unet.configure_optimizers()
    optw.zero_grad()
    optu.zero_grad()
    redw, redu = GradientAllReducer(optw.flat_grads), GradientAllReducer(optu.flat_grads)
    batch = synthetic_batch(args.batch, dev, seed=420, start=rank * args.batch)

    def eager_step():
        optw.zero_grad()
        res = warp.training_step(batch, 0)
        res.minimize.backward()
        optw.step(grad_scale=redw.all_reduce())
        b2 = dict(batch)
        b2["cloth"] = warp.warped_cloth.detach()
        optu.zero_grad()
        res = unet.training_step(b2, 0)
        res.minimize.backward()
        optu.step(grad_scale=redu.all_reduce())

    # first step: every new layer shape times its (tile, split-K) candidates; keep the streams serial meanwhile so
    # the measurements are not disturbed by the concurrent weight-gradient stream
    from shineon_virtual_tryon_amd import ops as so_ops

    concurrent = so_ops.CONCURRENT_WGRAD and not args.serial_wgrad
    so_ops.CONCURRENT_WGRAD = False
    eager_step()
    torch.cuda.synchronize()
    so_ops.CONCURRENT_WGRAD = concurrent
    log(f"{L.so_igemm_plan_count()} igemm plans measured; concurrent wgrad stream: {concurrent}")

    step = eager_step
    if not args.no_graph:
        # forward+backward of each model captured once as a hipGraph; Adam + RCCL all-reduce stay eager
        from shineon_virtual_tryon_amd.graphs import GraphedTrainStep

        eager_step()  # allocates workspaces, sets kernel attributes, plants the flat gradient views
        pending = {"unet": False}
        gp = gw = gu = None
        use_pipeline = not args.no_pipeline
        if use_pipeline and world > 1:
            # The two-stream schedule hides the warp model's gradient exchange but leaves the try-on model's exposed
            # between two try-on graphs; the sequential schedule hides both behind the other model's graph.  Measure the
            # try-on all-reduce on this node (same message sizes, scratch buffer) and keep the two-stream schedule only
            # where its gain on one GPU (~1.0 ms/step) exceeds that exposure.  All ranks take the same decision (MAX).
            scratch = torch.zeros_like(optu.flat_grads)
            probe = GradientAllReducer(scratch)
            for _ in range(2):
                probe.all_reduce()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                probe.all_reduce()
            e1.record()
            torch.cuda.synchronize()
            t_ms = torch.tensor([e0.elapsed_time(e1) / 5], dtype=torch.float64, device=dev)
            dist.all_reduce(t_ms, op=dist.ReduceOp.MAX)
            use_pipeline = float(t_ms.item()) < 1.0
            log(f"try-on gradient all-reduce ({scratch.numel() * 4 / 1e6:.1f} MB, {world} ranks): {float(t_ms.item()):.2f} ms -> "
                f"{'two-stream' if use_pipeline else 'sequential'} schedule")
            del scratch, probe
        if use_pipeline:
            from shineon_virtual_tryon_amd.graphs import GraphedChainedStep

            gp = GraphedChainedStep(warp, optw, unet, optu, batch)
        else:
            gw = GraphedTrainStep(warp, optw, batch)
            b2 = dict(batch)
            b2["cloth"] = warp.warped_cloth.detach()  # static output of the warp graph, consumed in place
            gu = GraphedTrainStep(unet, optu, b2, alias_keys=("cloth",))

        def flush():
            if pending["unet"]:
                optu.step(grad_scale=redu.finish())
                pending["unet"] = False

        def step_pipeline():
            # side stream: warp fwd -> warp bwd -> warp all-reduce -> warp Adam, all behind the try-on graph;
            # main stream: (previous try-on all-reduce, Adam) -> try-on fwd+bwd -> try-on all-reduce (hidden behind the
            # next step's warp forward).  The try-on stage only waits for the warped cloth.
            gp.launch_warp_forward()
            flush()
            gp.launch_tryon()
            gp.launch_warp_backward()
            with gp.on_side():
                redw.start()
                optw.step(grad_scale=redw.finish())
            redu.start()
            pending["unet"] = True

        def step():
            if gp is not None:
                return step_pipeline()
            # Both all-reduces are hidden behind compute: the warp gradients travel over xGMI while the try-on
            # graph runs, the try-on gradients while the NEXT step's warp graph runs (the two models share no
            # parameters, so the try-on Adam update only has to land before the next try-on forward).
            gw()
            redw.start()
            flush()
            gu()
            redu.start()
            pending["unet"] = True
            optw.step(grad_scale=redw.finish())

    log(f"rank {rank}/{world}: models built, warm-up {args.warmup} steps")
    for i in range(args.warmup):
        t_w = time.perf_counter()
        step()
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {1e3 * (time.perf_counter() - t_w):.1f} ms")
    if not args.no_graph:
        flush()  # the timed region starts with no update pending

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    if args.no_graph:
        L.so_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if not args.no_graph:
        flush()  # the last try-on Adam update belongs to the timed steps
    fence()
    elapsed = time.perf_counter() - t0
    L.so_prof_enable(0)
    log(f"timed {args.steps} steps: {1e3 * elapsed / args.steps:.2f} ms/step")
    prof_steps = args.steps
    if not args.no_graph:
        # Kernels inside a replayed graph cannot be bracketed individually, so the per-kernel HIP-event timing
        # (roofline figure) is taken on the SAME kernels launched eagerly right after the timed region.
        prof_steps = min(args.steps, 5)
        eager_step()
        fence()
        L.so_prof_enable(1)
        for _ in range(prof_steps):
            eager_step()
        fence()
        L.so_prof_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms = (ctypes.c_float * 32)()
    fl = (ctypes.c_float * 32)()
    cnt = (ctypes.c_int * 32)()
    L.so_prof_collect(ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(cnt))

    if rank == 0:
        kernels = {KEY_NAMES[k]: {"launches": cnt[k], "avg_us": 1e3 * ms[k] / cnt[k], "total_ms_per_step": ms[k] / prof_steps,
                                  "tflops": fl[k] / (ms[k] * 1e-3) / 1e12}
                   for k in range(32) if cnt[k] > 0}
        dom = max(range(32), key=lambda k: ms[k])
        traffic = None  # HBM bytes per launch of the dominant instantiation, from the committed PMC passes
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(KEY_NAMES[dom], {}).get("hbm_bytes_per_launch")
        achieved = fl[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
        mfma_ms = sum(ms) / prof_steps
        out = {
            "metric": "try-on frames/sec (fwd+bwd) at 256x192 bs=4",
            "value": world * args.batch * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp32",
            "data": "synthetic",
            "config": {
                "workload": "chained warp->try-on training step (SURVEY 8d C4 at bs=4/GPU): WarpModel (GMM) fwd+bwd+Adam, then "
                            "UnetMaskModel (self_attn, num_attn=2, gelu; L1+VGG19+mask loss) fwd+bwd+Adam on the warped cloth, "
                            "256x192",
                "launch": ("eager" if args.no_graph else "two sequential hipGraphs (warp, try-on); Adam and all-reduce eager" if gp is None
                           else "three hipGraphs on two streams: warp forward -> [try-on fwd+bwd || warp backward + its all-reduce + Adam]; "
                                "Adam and all-reduce eager"), "batch_per_gpu": args.batch, "global_batch": world * args.batch, "parallelism": f"dp{world}",
            },
            "roofline": {
                "bound": "mfma", "kernel": f"so_igemm_kernel<{KEY_NAMES[dom]}>", "achieved": achieved,
                "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                "traffic": traffic, "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, bytes per launch)" if traffic else None, "timing": ("hip events, eager launches in the timed region" if args.no_graph else f"hip events on the same kernels launched eagerly for {prof_steps} steps right after the graph-replayed timed region"), "avg_launch_us": 1e3 * ms[dom] / max(1, cnt[dom]),
                "mfma_ms_per_step": mfma_ms, "mfma_time_frac_of_step": mfma_ms / (1e3 * elapsed / args.steps),
                "all_mfma_tflops": sum(fl) / (sum(ms) * 1e-3) / 1e12 if sum(ms) > 0 else 0.0,
            },
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.batch, args.cpu_iters)
        print(json.dumps(out), flush=True)
    if args.plans and rank == 0:
        log(f"saved {L.so_igemm_plans_save(args.plans.encode())} igemm plans to {args.plans}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
