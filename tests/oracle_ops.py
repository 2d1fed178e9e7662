"""torch.autograd wrappers around the CPU oracle with the operator signatures of the package.

TEST INFRASTRUCTURE ONLY: lets the host-side logic of the package (render composition, tile-strip
sharding, the gradient all-reduce) and BASELINE config 1 (the 2D fit loop) run on CPU tensors in the
`-m "not gpu"` suite.  The product never imports this module.
"""
from types import SimpleNamespace

import numpy as np
import torch

from oracle import oracle as orc
from taichi_gaussian_rasterizer_amd.rasterizer.function import RasterOut


def _t(a, like):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype=like.dtype if a.dtype.kind == "f" else None)


class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, position, log_scaling, rotation, alpha_logit, T, proj, image_size, depth_range, config):
        p, d, idx = orc.project(position, log_scaling, rotation, alpha_logit, T, proj, image_size, depth_range,
                                blur_cov=config.blur_cov, clamp_margin=config.clamp_margin,
                                alpha_threshold=config.alpha_threshold)
        ndc = orc.ndc_depth(d.astype(np.float32), depth_range[0], depth_range[1])
        ctx.save_for_backward(position, log_scaling, rotation, alpha_logit, T, proj)
        ctx.meta = (image_size, config, idx)
        outs = (_t(p, position), _t(d, position), torch.from_numpy(idx), _t(ndc, position).reshape(-1, 1))
        ctx.mark_non_differentiable(outs[2], outs[3])
        return outs

    @staticmethod
    def backward(ctx, gp, gd, _gi, _gn):
        position, log_scaling, rotation, alpha_logit, T, proj = ctx.saved_tensors
        image_size, config, idx = ctx.meta
        gp = torch.zeros((idx.shape[0], 7), dtype=position.dtype) if gp is None else gp
        gd = torch.zeros((idx.shape[0], 1), dtype=position.dtype) if gd is None else gd
        grads = orc.project_backward(position, log_scaling, rotation, alpha_logit, T, proj, image_size, idx, gp, gd,
                                     blur_cov=config.blur_cov, clamp_margin=config.clamp_margin)
        return (*[_t(g, position) for g in grads], None, None, None)


def project_with_ndc(position, log_scaling, rotation, alpha_logit, T, proj, image_size, depth_range, config):
    return _Project.apply(position, log_scaling, rotation, alpha_logit, T, proj, tuple(image_size),
                          tuple(depth_range), config)


class _SH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, points, indexes, cam):
        ctx.save_for_backward(params, points, indexes, cam)
        return _t(orc.evaluate_sh_at(params, points, indexes, cam), params)

    @staticmethod
    def backward(ctx, go):
        params, points, indexes, cam = ctx.saved_tensors
        dp, dpts, dcam = orc.evaluate_sh_at_backward(params, points, indexes, cam, go.contiguous())
        return _t(dp, params), _t(dpts, params), None, _t(dcam, params)


def evaluate_sh_at(params, points, indexes, cam):
    return _SH.apply(params, points, indexes, cam)


def map_to_tiles(gaussians, depth, image_size, config, use_depth16=False):
    o2p, ranges = orc.map_to_tiles(gaussians.detach().float(), depth.detach().float(), image_size, config,
                                   use_depth16)
    return torch.from_numpy(o2p), torch.from_numpy(ranges)


class _Raster(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gaussians, features, o2p, ranges, image_size, config):
        image, alpha, vis = orc.rasterize_with_tiles(gaussians, features, o2p, ranges, image_size, config)
        v = gaussians.shape[0]
        heur = torch.zeros((v, 2), dtype=gaussians.dtype) if config.compute_point_heuristic \
            else torch.empty((0, 2), dtype=gaussians.dtype)
        vis_t = _t(vis, gaussians) if vis is not None else torch.empty((0,), dtype=gaussians.dtype)
        image_t, alpha_t = _t(image, gaussians), _t(alpha, gaussians)
        ctx.save_for_backward(gaussians, features, o2p, ranges, image_t)
        ctx.meta = (image_size, config, heur)
        ctx.mark_non_differentiable(alpha_t, heur, vis_t)
        return image_t, alpha_t, heur, vis_t

    @staticmethod
    def backward(ctx, gi, _a, _h, _v):
        gaussians, features, o2p, ranges, image = ctx.saved_tensors
        image_size, config, heur = ctx.meta
        gg, gf, h = orc.rasterize_backward(gaussians, features, o2p, ranges, image_size, image, gi.contiguous(),
                                           config)
        if h is not None:
            heur.copy_(_t(h, gaussians))
        return _t(gg, gaussians), _t(gf, gaussians), None, None, None, None


def rasterize_with_tiles(gaussians2d, features, overlap_to_point, tile_overlap_ranges, image_size, config):
    return RasterOut(*_Raster.apply(gaussians2d, features, overlap_to_point, tile_overlap_ranges,
                                    tuple(int(x) for x in image_size), config))


def rasterize(gaussians2d, depth, features, image_size, config, use_depth16=False):
    o2p, ranges = map_to_tiles(gaussians2d, depth, image_size, config, use_depth16)
    return rasterize_with_tiles(gaussians2d, features, o2p, ranges.view(-1, 2), image_size, config)


OPS = SimpleNamespace(project_with_ndc=project_with_ndc, evaluate_sh_at=evaluate_sh_at, map_to_tiles=map_to_tiles,
                      rasterize_with_tiles=rasterize_with_tiles)
