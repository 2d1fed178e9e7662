"""torch helpers of the projection stage (the names of the reference's torch_lib/projection.py).

First part: what sits ON the render path there -- ndc_depth (:120-123, called from renderer.py:189), its inverse
(:126-129) and the point (un)projection helpers used to build synthetic scenes (:48-60).  Second part: a plain-torch
EWA projection with the reference oracle's entry points; the package never calls it (project.hip does the work), it
is a differentiable utility that tests/test_oracle_golden.py holds to the reference's own golden outputs.
"""
from __future__ import annotations

import torch


def ndc_depth(depth: torch.Tensor, near: float, far: float) -> torch.Tensor:
    """ndc from 0 (near) to 1 (far).  Same formula as the reference; the sort-key path inside
    render_gaussians uses the fused HIP version with a fixed f32 operation order."""
    return 1 - (1. / depth - 1. / far) / (1. / near - 1. / far)


def inverse_ndc_depth(ndc_depth: torch.Tensor, near: float, far: float) -> torch.Tensor:
    return 1.0 / ((1.0 - ndc_depth) * (1 / near - 1 / far) + 1 / far)


def make_homog(points):
    shape = list(points.shape)
    shape[-1] = 1
    return torch.cat([points, torch.ones(shape, dtype=points.dtype, device=points.device)], dim=-1)


def transform44(transform, points):
    return (transform.reshape(1, 4, 4) @ points.reshape(-1, 4, 1))[..., 0]


def project_points(transform, xyz):
    homog = transform44(transform, make_homog(xyz))
    depth = homog[..., 2:3]
    return homog[..., 0:2] / depth, depth


def unproject_points(uv, depth, transform):
    points = torch.cat([uv * depth, depth, torch.ones_like(depth)], dim=-1)
    transformed = transform44(torch.inverse(transform), points)
    return transformed[..., 0:3] / transformed[..., 3:4]


def inverse_sigmoid(x: torch.Tensor):
    return torch.log(x / (1 - x))


def quat_to_mat(quat: torch.Tensor) -> torch.Tensor:
    """xyzw quaternion -> rotation matrix (reference torch_lib/transforms.py:4-15)."""
    x, y, z, w = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    x2, y2, z2 = x * x, y * y, z * z
    return torch.stack([
        1 - 2 * y2 - 2 * z2, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y,
        2 * x * y + 2 * w * z, 1 - 2 * x2 - 2 * z2, 2 * y * z - 2 * w * x,
        2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x2 - 2 * y2], dim=-1).reshape(quat.shape[:-1] + (3, 3))


def join_rt(r, t):
    T = torch.eye(4, device=r.device, dtype=r.dtype)
    T[0:3, 0:3] = r
    T[0:3, 3] = t
    return T


# ---------------------------------------------------------------------------------------------------------------
# EWA projection of 3D Gaussians in plain torch: a differentiable stand-alone utility with the entry points of the
# reference's torch_lib/projection.py (eig :20-38, ellipse_bounds :41-43, covariance_in_camera :64-77,
# project_with_jacobian :80-100, project_perspective_gaussian :103-110, cov_to_conic :113-118, generalized_ndc
# :132-139, apply :156-191).  The renderer does this work in project.hip; tests/test_oracle_golden.py holds this
# version to the reference's own outputs and autograd gradients.
def eig(cov: torch.Tensor):
    """symmetric 2x2 (..., 2, 2) -> (sqrt of the eigenvalues, major first), unit major axis, unit minor axis"""
    assert cov.shape[-2:] == (2, 2), f"Expected ...x2x2 covariance matrix, got {cov.shape}"
    a, b, c = cov[..., 0, 0], cov[..., 0, 1], cov[..., 1, 1]
    trace = a + c
    gap = (trace * trace - 4.0 * (a * c - b * b)).clamp_min(0).sqrt()
    major, minor = 0.5 * (trace + gap), 0.5 * (trace - gap)
    v1 = torch.nn.functional.normalize(torch.stack((a - minor, b), dim=-1), dim=-1)
    v2 = torch.stack((-v1[..., 1], v1[..., 0]), dim=-1)
    return torch.stack((major, minor), dim=-1).sqrt(), v1, v2


def radii_from_cov(uv_cov: torch.Tensor) -> torch.Tensor:
    """largest standard deviation of each 2x2 covariance"""
    a, b, _, c = uv_cov.reshape(-1, 4).unbind(1)
    return (0.5 * (a + c + ((a - c) ** 2 + 4.0 * b * b).sqrt())).sqrt()


def ellipse_bounds(mean: torch.Tensor, v1: torch.Tensor, v2: torch.Tensor):
    """axis-aligned box of the ellipse with semi-axes v1, v2 (vectors) around mean"""
    half = (v1 * v1 + v2 * v2).sqrt()
    return mean - half, mean + half


def covariance_in_camera(T_camera_world: torch.Tensor, cov_rotation: torch.Tensor, cov_scale: torch.Tensor):
    """(N, 3, 3) covariance W R S S R^T W^T of unit quaternions `cov_rotation` and sigmas `cov_scale`"""
    factor = (T_camera_world[:3, :3] @ quat_to_mat(cov_rotation)) * cov_scale.unsqueeze(1)
    return factor @ factor.transpose(1, 2)


def project_with_jacobian(projection: torch.Tensor, position: torch.Tensor, image_size: torch.Tensor,
                          clamp_margin: float = 0.15):
    """camera-space points -> pixel means, depths and the (N, 2, 3) Jacobian of the perspective map, evaluated at the
    mean clamped to the image plus a margin (keeps the linearisation sane for splats far off screen)"""
    focal, centre = projection[:2], projection[2:]
    z = position[:, 2]
    uv = position[:, :2] * focal / z.unsqueeze(1) + centre
    at = torch.clamp(uv, -clamp_margin * image_size, (1.0 + clamp_margin) * (image_size - 1))
    J = position.new_zeros((position.shape[0], 2, 3))
    J[:, 0, 0], J[:, 1, 1] = focal[0] / z, focal[1] / z
    J[:, :, 2] = -(at - centre) / z.unsqueeze(1)
    return uv, z, J


def project_perspective_gaussian(J: torch.Tensor, cov_in_camera: torch.Tensor) -> torch.Tensor:
    return J @ cov_in_camera @ J.transpose(1, 2)


def cov_to_conic(cov: torch.Tensor) -> torch.Tensor:
    """(..., 2, 2) covariance -> the three distinct entries of its inverse"""
    a, b, c = cov[..., 0, 0], cov[..., 0, 1], cov[..., 1, 1]
    det = a * c - b * b
    return torch.stack((c / det, -b / det, a / det), dim=-1)


def generalized_ndc(depth: torch.Tensor, near: float, far: float, k: float) -> torch.Tensor:
    """power-law depth normalisation: k = 1 linear, k = -1 the inverse-depth ndc"""
    lo, hi = near ** k, far ** k
    return (depth.pow(k) - hi) / (hi - lo)


def unpack_activate(vec: torch.Tensor):
    """(..., 11) packed [position, log sigma, quaternion, alpha logit] -> activated parts"""
    quaternion = vec[..., 6:10]
    return (vec[..., 0:3], vec[..., 3:6].exp(), quaternion / quaternion.norm(dim=-1, keepdim=True),
            vec[..., 10:11].sigmoid())


def apply(position, log_scaling, rotation, alpha_logit, T_camera_world, projection, image_size, depth_range,
          blur_cov=0.0, clamp_margin=0.15, alpha_threshold=1. / 255.):
    """Project, cull and pack: returns (points (V, 7) [mean, major axis, sigmas, alpha], depth (V, 1), indexes (V))
    for the Gaussians whose alpha_threshold-level ellipse meets the image and whose depth is inside depth_range."""
    T_camera_world, projection = T_camera_world.reshape(4, 4), projection.reshape(4)
    size = torch.tensor(image_size, dtype=position.dtype, device=position.device)
    in_camera = position @ T_camera_world[:3, :3].T + T_camera_world[:3, 3]
    mean, z, J = project_with_jacobian(projection, in_camera, size, clamp_margin)
    spread = covariance_in_camera(T_camera_world, torch.nn.functional.normalize(rotation, dim=-1), log_scaling.exp())
    cov = project_perspective_gaussian(J, spread) + blur_cov * torch.eye(2, dtype=mean.dtype, device=mean.device)
    sigma, v1, v2 = eig(cov)
    alpha = alpha_logit.sigmoid()
    reach = sigma * (2.0 * torch.log(alpha / alpha_threshold)).sqrt()   # NaN below the threshold: culled below
    lower, upper = ellipse_bounds(mean, v1 * reach[:, 0:1], v2 * reach[:, 1:2])
    seen = (z > depth_range[0]) & (z < depth_range[1]) & (upper > 0).all(1) & (lower < size.unsqueeze(0)).all(1)
    packed = torch.cat((mean, v1, sigma, alpha), dim=-1)
    return packed[seen], z[seen].unsqueeze(1), seen.nonzero(as_tuple=True)[0]


def project_to_image(gaussians, camera_params, config):
    """torch counterpart of perspective.projection.project_to_image"""
    return apply(*gaussians.shape_tensors(), camera_params.T_camera_world, camera_params.projection,
                 camera_params.image_size, camera_params.depth_range, blur_cov=config.blur_cov,
                 clamp_margin=config.clamp_margin, alpha_threshold=config.alpha_threshold)
