"""per-rank stage times of a sharded frame, emulated on one GPU (no process group: nothing is communicated).
usage: python tools/exp_shard.py [c3|c5] [interleave] [dense|sparse|sharded]
  dense   : the dense all-reduce path (pack kernel; the collective itself is a no-op here)
  sparse  : exchange="sparse", replicated gradients: pack of the touched rows + the add of world lists of that size
  sharded : grad_mode="sharded": per-owner lists, adjoints on the rank's own index range only (range-shaped gradients)"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import taichi_gaussian_rasterizer_amd as gs
from taichi_gaussian_rasterizer_amd import RasterConfig, scenes, parallel, _native as nv
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
interleave = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mode = sys.argv[3] if len(sys.argv) > 3 else "dense"
n, size = (1_000_000, (2048, 2048)) if wl == "c3" else (6_000_000, (4096, 4096))
g, cam = scenes.benchmark_scene(n, size, sh_degree=3, seed=0)
cam = cam.to(device='cuda:0'); cfg = RasterConfig()
G = torch.rand(size[1], size[0], 3, device='cuda:0')
gg = g.to('cuda:0')
rows_of = {}
base = None
for world in (1, 2, 4, 8):
    rank = world // 2
    parallel.EMULATED_WORLD = world if mode != "dense" else 0
    owned = parallel.split_owned(gg, rank, world).requires_grad_(True) if mode == "sharded" else None
    full = gg if mode == "sharded" else gg.detach().requires_grad_(True)
    holder = owned if owned is not None else full
    info = {}
    def step():
        for _, t in holder.items(): t.grad = None
        r = parallel.render_gaussians_sharded(full, cam, cfg, use_sh=True, rank=rank, world_size=world,
                                              interleave=interleave, exchange="dense" if mode == "dense" else "sparse",
                                              grad_mode="sharded" if mode == "sharded" else "replicated", owned=owned)
        if world not in rows_of:
            rows_of[world] = G[parallel.owned_pixel_rows(r.bands).cuda()].contiguous()
        r.image.backward(rows_of[world])
        info["V"] = int(r.points_in_view.shape[0])
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    nv.timer.reset(); nv.timer.only = None; nv.timer.enabled = True
    for _ in range(5): step()
    torch.cuda.synchronize(); nv.timer.enabled = False
    st = {k: round(v[1] / 5, 3) for k, v in nv.timer.summary().items()}
    gpu = sum(st.values())
    if base is None:
        base, base_gpu = dt, gpu
    print(f"{mode} world {world} rank {rank}: {dt:.3f} ms/frame wall ({base / dt:.2f}x; includes the Python of the emulation), "
          f"GPU stages {gpu:.3f} ms ({base_gpu / gpu:.2f}x), no collective ", st, flush=True)
