#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3]

Metric (BASELINE.json): Mpixels/s of one forward+backward `render_gaussians` frame.
Workloads (BASELINE.json configs / SURVEY.md 8d, synthetic seeded scene, random-init parameters):
    c2: 200k Gaussians, 1920x1080, SH deg 0, forward only
    c3: 1M Gaussians, 2048x2048, SH deg 3, forward+backward            (default; the headline)
    c4: c3 + depth / depth-variance feature render (F = 5)
    c5: 6M Gaussians, 4096x4096, SH deg 3, forward+backward
A step = one frame: forward, then backward from a fixed random dL/d(image) (+ depth terms for c4); inputs are
resident in HBM before the timed region.  N > 1 (launched by torch.distributed.run, one rank per
GPU over RCCL): the SAME frame is sharded by tile-row strips with one all-reduce of per-Gaussian
gradients (taichi_gaussian_rasterizer_amd/parallel.py) -> "scaling": "strong".

Rank 0 prints ONE JSON line.  Beyond the driver's contract it carries
  roofline     : the dominant kernel's achieved algorithmic bytes/s (HIP events on the launch stream
                 inside the timed region) against the 8 TB/s HBM peak,
  cpu_baseline : the CPU oracle (oracle/, "port" of the same algorithm, OpenMP) timed on this host,
  stages_ms_untimed_pass : per C-ABI entry point GPU time per step (separate untimed pass of 20 frames BEFORE the W
                 warm-up steps; then untimed frames until 1 s has passed since the first one, so that a process that
                 starts on an idle GPU measures its sustained clocks whatever W and K are; all of them are reported as
                 "pre_warm_frames" / "pre_warm_seconds").
Defaults: K = 100, W = 20 (SURVEY 8d asks for >= 50 iterations after >= 10 warm-ups); the default run takes about 20 s,
most of it scene generation and the CPU baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    "c2": dict(n=200_000, size=(1920, 1080), sh_degree=0, backward=False, depth=False),
    "c3": dict(n=1_000_000, size=(2048, 2048), sh_degree=3, backward=True, depth=False),
    "c4": dict(n=1_000_000, size=(2048, 2048), sh_degree=3, backward=True, depth=True),
    "c5": dict(n=6_000_000, size=(4096, 4096), sh_degree=3, backward=True, depth=False),
}
VALU_ISSUE_NS, NUM_SIMDS = 1.1, 1024  # tools/ubench/valu_rate.hip; 256 CUs x 4 SIMDs
STAGE_FRAMES = 20  # untimed frames of the per-stage table, run before the warm-up
PRE_WARM_SECONDS = 1.0  # ... followed by untimed frames until this much time has passed since the first frame
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def algorithmic_bytes(N, V, K, T, P, F, C, D):
    """Compulsory HBM bytes per launch of each stage (every stage reads its inputs
    once and writes its outputs once; atomics counted once per (tile, splat))."""
    return {
        "gs_project_fwd": 44 * N + 64 * N + 48 * V + 4 * N,          # inputs, staging w+r, compact rows, slot map
        "gs_sh_fwd": V * (8 + 12 + 4 * C * D + 4 * C),
        # sharded frame only: colours of the splats that reach the rank's rows (the caller scales by 1 / world), and the
        # exchange buffers: gradient row + forward colours read, 4 (7 + F) packed bytes written
        "gs_sh_fwd_shard": V * (28 + 8 + 12 + 4 * C * D + 4 * C),
        "gs_shard_pack_grads": V * (64 + 4 * F + 4 * (7 + F)),
        "gs_map_prepare": 28 * V + 16 * T,
        "gs_map_finish": 32 * V + 8 * K + 8 * K + 4 * K + 8 * T,     # query again, bucket w, sort r, order w
        "gs_raster_fwd": 8 * T + K * (4 + 28 + 4 * F) + 4 * P * (F + 1),
        "gs_raster_bwd": 8 * T + K * (4 + 28 + 4 * F) + 8 * P * F + 4 * (7 + F) * K,
        "gs_raster_bwd_unpack": V * (64 + 4 * (7 + F)),
        # dense adjoint: one 4CD-byte row written per Gaussian (zeros for culled ones), per visible Gaussian its
        # position, the forward colour (clamp mask) and the 64-B gradient row are read; coefficients are not re-read
        "gs_sh_bwd": 4 * C * D * N + 4 * N + V * (12 + 4 * C + 64),
        "gs_project_bwd": 44 * N + 4 * N + 32 * V + 44 * N,
    }


def survey_bytes(N, V, K, T, P, F, C, D, backward, kb=8):
    """SURVEY.md 8(d) whole-frame formula (B_fwd, B_bwd), general feature width F."""
    b_fwd = (44 * N + V * (40 + (20 + 4 * C * D + 4 * C) + 8 + 32 + 8 + 36)
             + K * ((kb + 4) + 2 * (kb + 4) + kb + (32 + 4 * F)) + 16 * T + 4 * P * (F + 1))
    b_bwd = (8 * T + K * (32 + 4 * F) + 8 * P * F + 4 * (7 + F) * K + 4 * (7 + F) * V
             + V * (20 + 4 * C * D + 4 * C) + 4 * C * D * N + 76 * V + 44 * N)
    return b_fwd, (b_bwd if backward else 0)


def launch_command(args, port=None):
    """The one-node multi-rank launch of this script: one process per GPU over RCCL (torch.distributed.run)."""
    import socket
    if port is None:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--workload", args.workload, "--cpu-frames", str(args.cpu_frames)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_kernel_timing:
        cmd.append("--no-kernel-timing")
    if args.interleave:
        cmd += ["--interleave", str(args.interleave)]
    cmd += ["--exchange", args.exchange, "--grad-mode", args.grad_mode]
    return cmd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--interleave", type=int, default=0,
                    help="N > 1: tile rows per band of the interleaved strip assignment (0 = one contiguous strip)")
    ap.add_argument("--exchange", default="dense", choices=["dense", "sparse"],
                    help="N > 1: how the ranks sum their partial per-Gaussian gradients (parallel.py)")
    ap.add_argument("--grad-mode", default="replicated", choices=["replicated", "sharded"],
                    help="N > 1: sharded = every rank ends with the gradients of its own index range only (sharded "
                         "optimizer; implies the sparse exchange)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="print the multi-rank launch command for --gpus N and exit")
    args = ap.parse_args()

    # --gpus N without a launcher around us: start the N ranks ourselves, as a CHILD process (never exec: this
    # process may already hold the GPU), relay its output and exit with its code.  No GPU call happens before this.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        cmd = launch_command(args)
        if args.dry_run_launch:
            print(" ".join(cmd))
            return 0
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("MASTER_ADDR", "127.0.0.1")
        return subprocess.run(cmd, env=env).returncode
    if args.dry_run_launch:
        print("single process: " + " ".join([sys.executable, os.path.abspath(__file__), "--gpus", "1"]))
        return 0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GS_BENCH_BACKEND=gloo + GS_BENCH_SHARE_GPU=1: rehearsal of the multi-rank code path on a one-GPU box
    # (all ranks on cuda:0, collectives staged through the host); never used for reported numbers
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    if os.environ.get("GS_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a mislabelled line",
              file=sys.stderr)
        return 2
    share = os.environ.get("GS_BENCH_SHARE_GPU") == "1"
    if not share and torch.cuda.device_count() < world:  # device_count() does not initialise the GPU
        print(f"bench.py: --gpus {world} needs {world} GPUs, this host shows {torch.cuda.device_count()}",
              file=sys.stderr)
        return 3
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path in the product"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import taichi_gaussian_rasterizer_amd as gs
    from taichi_gaussian_rasterizer_amd import _native as nv
    from taichi_gaussian_rasterizer_amd import RasterConfig, parallel, scenes

    wl = WORKLOADS[args.workload]
    W, H = wl["size"]
    cfg = RasterConfig()
    g_cpu, cam_cpu = scenes.benchmark_scene(wl["n"], wl["size"], sh_degree=wl["sh_degree"], seed=0)
    g = g_cpu.to(dev)
    owned = None
    sharded_grads = world > 1 and args.grad_mode == "sharded" and wl["backward"]
    if sharded_grads:
        owned = parallel.split_owned(g, rank, world).requires_grad_(True)
    elif wl["backward"]:
        g.requires_grad_(True)
    grad_holder = owned if owned is not None else g
    cam = cam_cpu.to(device=dev)
    gen = torch.Generator().manual_seed(1)
    G = torch.rand(H, W, 3, generator=gen).to(dev)
    Gd = torch.rand(H, W, generator=gen).to(dev) if wl["depth"] else None
    Gv = (torch.rand(H, W, generator=gen) * 0.1).to(dev) if wl["depth"] else None
    info = {}

    def step():
        if wl["backward"]:
            for _, t in grad_holder.items():
                t.grad = None
        if world > 1:
            r = parallel.render_gaussians_sharded(g, cam, cfg, use_sh=True, render_depth=wl["depth"],
                                                  interleave=args.interleave, exchange=args.exchange,
                                                  grad_mode=args.grad_mode if wl["backward"] else "replicated",
                                                  owned=owned)
        else:
            r = gs.render_gaussians(g, cam, cfg, use_sh=True, render_depth=wl["depth"])
        info["V"] = int(r.points_in_view.shape[0])
        if not wl["backward"]:
            return r
        # backward from a fixed random dL/d(outputs): the loss function is the caller's, not part of the path
        # (the reference benchmark uses image.sum(), i.e. an all-ones gradient: benchmarks/bench_rasterizer.py:83-85)
        if world > 1 and "rows" not in info:  # this rank's rows of the upstream gradients, gathered once
            rows = parallel.owned_pixel_rows(r.bands).to(dev)
            info["rows"] = rows
            info["G"] = [t[rows].contiguous() for t in (G, Gd, Gv) if t is not None]
        Gs = info["G"] if world > 1 else [t for t in (G, Gd, Gv) if t is not None]
        if wl["depth"]:
            torch.autograd.backward([r.image, r.depth, r.depth_var], Gs)
        else:
            r.image.backward(Gs[0])
        return r

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Inside the timed region only the dominant entry point is bracketed by HIP events (two event
    # records per step); bracketing all ~10 entry points costs ~0.05 ms of host time per step.  The
    # per-stage table is measured in a separate, untimed pass of STAGE_FRAMES frames FIRST: it also brings the GPU to
    # its sustained clocks (a cold run of 5 + 30 frames measures 4 % slower than 20 + 100), so that the W warm-up
    # steps and the K timed steps below see the steady state the metric asks for (>= 10 warm-ups, >= 50
    # iterations) whatever W and K are.
    stage_steps, stage_records = 0, {}
    pre_warm_extra = 0
    if not args.no_kernel_timing:
        t_first = time.perf_counter()
        for _ in range(STAGE_FRAMES // 2):  # first-frame allocations, buffer-size hints, clocks
            step()
        sync()
        nv.timer.reset()
        nv.timer.only = None
        nv.timer.enabled = True
        stage_steps = STAGE_FRAMES - STAGE_FRAMES // 2
        for _ in range(stage_steps):
            step()
        sync()
        nv.timer.enabled = False
        stage_records = nv.timer.summary() if nv.timer.records else {}
        # A process that starts on an idle GPU (the driver's: a fresh box, smoke(), then this) can find it in a low
        # power state for longer than 20 + W frames take: keep rendering, untimed, until PRE_WARM_SECONDS have passed
        # since the first frame.  Reported as part of "pre_warm_frames".  (Every rank runs the same number of frames:
        # a sharded step holds a collective.)
        while True:
            more = time.perf_counter() - t_first < PRE_WARM_SECONDS
            if world > 1:
                flag = torch.tensor([1 if more else 0], dtype=torch.int32, device=dev)
                dist.broadcast(flag, 0)
                more = bool(int(flag.item()))
            if not more:
                break
            for _ in range(25):
                step()
            sync()
            pre_warm_extra += 25
    for _ in range(args.warmup):
        step()
    sync()
    dominant = "gs_raster_bwd" if wl["backward"] else "gs_raster_fwd"
    nv.timer.reset()
    nv.timer.only = {dominant}
    nv.timer.enabled = not args.no_kernel_timing
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    nv.timer.enabled = False
    timed_records = nv.timer.summary() if nv.timer.records else {}
    allreduce = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if wl["backward"]:
            # the frame's one exchange step on its own: 4*(7+F) bytes per visible Gaussian, summed over the ranks
            F_ = 5 if wl["depth"] else 3
            buf = torch.zeros((info["V"], 7 + F_), dtype=torch.float32, device=dev)
            for _ in range(3):
                dist.all_reduce(buf)
            sync()
            t1 = time.perf_counter()
            for _ in range(10):
                dist.all_reduce(buf)
            sync()
            allreduce = dict(bytes=int(buf.numel() * 4), ms=round((time.perf_counter() - t1) * 100, 4),
                             note="gradient all-reduce timed alone, 10 back-to-back calls (inside a frame it overlaps "
                                  "the SH adjoint)")
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- scene statistics for the byte formulas (untimed)
    with torch.no_grad():
        from taichi_gaussian_rasterizer_amd.perspective.projection import project_with_ndc
        p2d, _, _, ndc = project_with_ndc(*[t.detach() for t in g.shape_tensors()], cam.T_camera_world,
                                          cam.projection, cam.image_size, cam.depth_range, cfg)
        o2p, ranges = gs.map_to_tiles(p2d, ndc, (W, H), cfg)
        V, K, T = int(p2d.shape[0]), int(o2p.shape[0]), int(ranges.shape[0] * ranges.shape[1])
    F = 5 if wl["depth"] else 3
    D = (wl["sh_degree"] + 1) ** 2
    by = algorithmic_bytes(wl["n"], V, K, T, W * H, F, 3, D)

    stages = {}
    roofline = None
    if stage_records and stage_steps:
        for name, (calls, total_ms) in stage_records.items():
            stages[name] = dict(calls_per_step=calls / stage_steps, ms_per_step=total_ms / stage_steps,
                                avg_launch_ms=total_ms / calls)
    if dominant in timed_records:
        calls, total_ms = timed_records[dominant]
        stages.setdefault(dominant, {})
        stages[dominant].update(avg_launch_ms=total_ms / calls)  # the live, in-timed-region measurement
        dom = dominant
        if stages and max(stages, key=lambda k: stages[k].get("ms_per_step", 0.0)) != dominant:
            print(f"# note: {max(stages, key=lambda k: stages[k].get('ms_per_step', 0.0))} outweighs {dominant}",
                  file=sys.stderr)
        if dom in by:
            # under sharding a launch covers 1/world of the tiles: scale the per-launch bytes accordingly
            per_launch = by[dom] / (world if dom.startswith(("gs_raster", "gs_map_finish", "gs_sh_fwd_shard")) else 1)
            achieved = per_launch / (stages[dom]["avg_launch_ms"] * 1e-3) / 1e9
            roofline = dict(bound="hbm", kernel=dom, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(achieved / HBM_PEAK_GBS, 5), traffic=None, traffic_source=None,
                            algorithmic_bytes_per_launch=int(per_launch),
                            avg_launch_ms=round(stages[dom]["avg_launch_ms"], 4))
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):  # HBM bytes per launch from rocprofv3 --pmc passes (see profiles/README.md)
                try:
                    rec = json.load(open(pmc)).get(args.workload, {}).get(dom)
                    if rec and world == 1:
                        roofline["traffic"] = rec["hbm_bytes_per_launch"]  # FETCH_SIZE x2 (gfx950) + WRITE_SIZE
                        roofline["traffic_source"] = ("profiles/pmc_traffic.json (static: rocprofv3 --pmc passes of "
                                                      "this workload, not collected by this run)")
                        roofline["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_launch"] / per_launch, 3)
                        if "valu_wave_insts_per_launch" in rec:
                            # the kernel is f32-VALU-issue-bound (no MFMA shape, HBM far from saturated): second
                            # roofline = measured wave64 VALU instructions against the chip's issue rate
                            insts = rec["valu_wave_insts_per_launch"]
                            floor_ms = insts * VALU_ISSUE_NS / NUM_SIMDS * 1e-6
                            roofline["valu"] = dict(wave_insts_per_launch=insts, issue_floor_ms=round(floor_ms, 4),
                                                    issue_frac=round(floor_ms / stages[dom]["avg_launch_ms"], 4))
                except Exception:
                    pass
    # only the entry points this frame actually ran (the fused frame consumes the gradient rows in place: no unpack)
    ran = set(stages) or {k for k in by if wl["backward"] or not k.endswith(("_bwd", "_unpack"))}
    whole = sum(v for k, v in by.items() if k in ran)
    sv = survey_bytes(wl["n"], V, K, T, W * H, F, 3, D, wl["backward"])

    # ---- the exchange step of an 8-rank frame of this workload, in bytes: exact without 8 GPUs (list lengths and V are
    # all it depends on); the lists are measured by running the 8 ranks' forward passes one after the other here
    exchange_model = None
    if rank == 0 and world == 1 and wl["backward"]:
        try:
            model_world = 8
            with torch.no_grad():
                touched = [int(parallel.render_gaussians_sharded(g.detach(), cam, cfg, use_sh=True,
                                                                 render_depth=wl["depth"], rank=q, world_size=model_world,
                                                                 interleave=args.interleave).touched_count)
                           for q in range(model_world)]
            exchange_model = dict(
                ranks=model_world, visible=V, touched_per_rank=touched, dense_rows_bytes=4 * (7 + F) * V,
                dense=parallel.exchanged_bytes("dense", "replicated", model_world, V, touched, F),
                sparse_replicated=parallel.exchanged_bytes("sparse", "replicated", model_world, V, touched, F),
                sparse_sharded=parallel.exchanged_bytes("sparse", "sharded", model_world, V, touched, F),
                note="payload bytes rank 0 sends / receives per frame; dense = ring all-reduce of the (V, 7 + F) rows")
        except Exception as e:
            exchange_model = dict(error=f"{type(e).__name__}: {e}")

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            check = step()  # one more, untimed frame: its image is compared with the oracle's below
            gpu_image = (check.image if hasattr(check, "image") else check).detach().float().cpu().numpy()
            cpu_baseline = run_cpu_baseline(g_cpu, cam_cpu, cfg, wl, args.cpu_frames, gpu_image=gpu_image,
                                            gpu_overlaps=K)
        except Exception as e:  # the oracle is test infrastructure; its absence must not fail the bench
            cpu_baseline = dict(error=f"{type(e).__name__}: {e}")

    if rank == 0:
        out = {
            "metric": "Mpixels/s fwd+bwd @ 2048x2048, 1M Gaussians" if args.workload == "c3"
                      else f"Mpixels/s ({args.workload})",
            "value": round(W * H / (ms_per_step * 1e-3) / 1e6, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "collective_backend": (dist.get_backend() if world > 1 else None),
            "allreduce": allreduce,
            "steps": args.steps,
            "warmup": args.warmup,
            # frames rendered BEFORE the W warm-up steps (the untimed per-stage pass, see the module docstring)
            "pre_warm_frames": 0 if args.no_kernel_timing else STAGE_FRAMES + pre_warm_extra,
            "pre_warm_seconds": 0.0 if args.no_kernel_timing else PRE_WARM_SECONDS,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if backend == "nccl" or world == 1 else f"synthetic (REHEARSAL: backend {backend}, ranks share one GPU)",
            "config": {"workload": f"{args.workload}: {wl['n']} Gaussians, {W}x{H}, SH deg {wl['sh_degree']}, "
                                   f"tile 16, {'fwd+bwd' if wl['backward'] else 'fwd'}"
                                   f"{', depth features' if wl['depth'] else ''}",
                       "visible": V, "overlaps": K, "tiles": T,
                       # not a reference field: the forward's early stop (deviation 6) is ON in this number
                       "forward_cut": cfg.forward_cut,
                       "backward_from": "fixed random dL/d(outputs), no loss kernels in the timed region"
                       if wl["backward"] else None,
                       "parallelism": "single GPU" if world == 1 else
                       (f"tile-row strips x{world} + " + ("grad all-reduce" if args.exchange == "dense" and
                                                          args.grad_mode == "replicated" else
                                                          f"sparse gradient exchange ({args.grad_mode})"))},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "exchange": dict(mode=args.exchange if args.grad_mode == "replicated" else "sparse",
                             grad_mode=args.grad_mode) if world > 1 else None,
            "exchange_model_8_ranks": exchange_model,
            "whole_path": {"algorithmic_bytes_per_frame": int(whole),
                           "hbm_frac": round(whole / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                           "survey_8d_bytes_per_frame": int(sum(sv)),
                           "survey_8d_hbm_frac": round(sum(sv) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                           "note": "first pair: sum of the per-stage byte table above (what this implementation's "
                                   "stages must move); second pair: SURVEY.md 8(d) B_fwd + B_bwd on the measured V, K"},
            # per entry point, from the untimed pre-warm pass with every entry point bracketed by events: NOT a
            # decomposition of ms_per_step (the bracketing itself costs host time; the sum is a few % above it)
            "stages_ms_untimed_pass": {k: round(v["ms_per_step"], 4) for k, v in sorted(stages.items())
                                       if "ms_per_step" in v},
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_cpu_baseline(g_cpu, cam_cpu, cfg, wl, frames, gpu_image=None, gpu_overlaps=None):
    """The CPU oracle (the C++/OpenMP restatement in oracle/) on the same scene, all host cores.  Its forward image of
    the full workload doubles as a parity check of the frame just benchmarked (`parity` in the returned object)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import parity_util as pu
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    orc.set_num_threads(cores)
    W, H = wl["size"]
    gi = np.random.default_rng(1).random((H, W, 3)).astype(np.float32)
    grads = dict(image=gi) if wl["backward"] else None
    if wl["depth"] and grads is not None:
        grads.update(depth=np.random.default_rng(2).random((H, W)).astype(np.float32),
                     depth_var=np.random.default_rng(3).random((H, W)).astype(np.float32) * 0.1)
    best = None
    for _ in range(max(1, frames)):
        t0 = time.perf_counter()
        out = pu.oracle_render(g_cpu, cam_cpu, cfg, use_sh=True, render_depth=wl["depth"], grads=grads, flips=False)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    result = dict(value=round(W * H / best / 1e6, 3), unit="Mpix/s", cores=orc.num_threads(), kind="port",
                  sample=f"the full workload, best of {max(1, frames)} frames, no warm-up "
                         f"({best:.2f} s/frame, OpenMP over {orc.num_threads()} threads)")
    if gpu_image is not None and gpu_image.shape == out["image"].shape:
        diff = np.abs(gpu_image - out["image"])
        beyond = diff > pu.ATOL + pu.RTOL * np.abs(out["image"])
        flip_bound = float(cfg.alpha_threshold) * float(np.abs(out["features"]).max())
        result["parity"] = dict(
            mean_abs_image_diff=float(diff.mean()), fraction_beyond_2e_5=float(beyond.mean()),
            max_abs_image_diff=float(diff.max()), single_threshold_flip_bound=flip_bound,
            overlaps_equal=bool(gpu_overlaps is None or int(out["o2p"].shape[0]) == int(gpu_overlaps)),
            note="informational: HIP frame vs the oracle's f32 frame of the whole workload, END TO END (the two "
                 "projections differ in the last bit, which moves alpha > 1/255 decisions downstream; one such flip "
                 "changes a pixel by at most alpha_threshold * |feature| = single_threshold_flip_bound).  The asserted "
                 "full-size comparison is tests/test_full_size_gpu.py: stage by stage, every out-of-tolerance pixel "
                 "proven a threshold flip (17 of 4.2 M pixels on this workload)")
    return result


if __name__ == "__main__":
    sys.exit(main() or 0)
