"""Dense O(pixels x splats) alpha-blend renderer in plain torch (autograd-differentiable).

Independent check of the tiled oracle and of the HIP rasterizer: no tiles, no culling, one
global depth order.  Valid because the reference's tile cull is exactly conservative for the
non-antialiased pdf: outside the oriented box at sigma*sqrt(2 ln(alpha/thr)) the splat's alpha is
below alpha_threshold and would be skipped anyway (taichi_lib/grid_query.py:76-77,
rasterizer/forward.py:101).  Blend recurrence: rasterizer/forward.py:96-108.  The clamp to
clamp_max_alpha passes gradients straight through, as the reference backward does
(rasterizer/backward.py:166-169).
"""
import torch


def gaussian_pdf(pix, g):
    """pix (P,2), g (V,7) -> (P,V)   taichi_lib/generic.py:311-317"""
    d = pix[:, None, :] - g[None, :, 0:2]
    ax, ay = g[None, :, 2], g[None, :, 3]
    tx = (d[..., 0] * ax + d[..., 1] * ay) / g[None, :, 4]
    ty = (d[..., 0] * -ay + d[..., 1] * ax) / g[None, :, 5]
    return torch.exp(-0.5 * (tx ** 2 + ty ** 2))


def _s_sig(x, sigma):
    z = x / sigma
    return 1 / (1 + torch.exp(-1.6 * z - 0.07 * z ** 3))


def gaussian_pdf_antialias(pix, g):
    """taichi_lib/generic.py:341-357"""
    d = pix[:, None, :] - g[None, :, 0:2]
    ax, ay = g[None, :, 2], g[None, :, 3]
    sx, sy = g[None, :, 4], g[None, :, 5]
    tx = d[..., 0] * ax + d[..., 1] * ay
    ty = d[..., 0] * -ay + d[..., 1] * ax
    return (2 * torch.pi * sx * (_s_sig(tx + 0.5, sx) - _s_sig(tx - 0.5, sx))
            * sy * (_s_sig(ty + 0.5, sy) - _s_sig(ty - 0.5, sy)))


def depth_order(depth):
    """stable order on the f32 bit pattern of the (non-negative) depth, ties by index"""
    bits = depth.detach().reshape(-1).to(torch.float32).contiguous().view(torch.int32).to(torch.int64)
    return torch.sort(bits, stable=True).indices


def render_dense(gaussians2d, depth, features, image_size, clamp_max_alpha=0.99, alpha_threshold=1 / 255.,
                 antialias=False, order=None, visible_mask=None):
    """-> image (H,W,F), alpha (H,W), weights (P,V in blend order) ; order = blend order of splats.
    visible_mask (P,V) optionally restricts which (pixel, splat) pairs may contribute (used to
    reproduce a tile cull that is not conservative, i.e. the antialiased pdf)."""
    W, H = image_size
    dt, dev = gaussians2d.dtype, gaussians2d.device
    if order is None:
        order = depth_order(depth)
    g = gaussians2d[order]
    f = features[order]
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1).to(dt) + 0.5
    pdf = gaussian_pdf_antialias(pix, g) if antialias else gaussian_pdf(pix, g)
    a = g[None, :, 6] * pdf
    a_c = a + (torch.clamp(a, max=clamp_max_alpha) - a).detach()
    mask = torch.clamp(a, max=clamp_max_alpha) > alpha_threshold
    if visible_mask is not None:
        mask = mask & visible_mask[:, order]
    a_eff = torch.where(mask, a_c, torch.zeros_like(a_c))
    T = torch.cumprod(1 - a_eff, dim=1)
    T_excl = torch.cat([torch.ones_like(T[:, :1]), T[:, :-1]], 1)
    w = a_eff * T_excl
    image = (w @ f).reshape(H, W, -1)
    alpha = w.sum(1).reshape(H, W)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel(), device=dev)
    return image, alpha, w[:, inv]
