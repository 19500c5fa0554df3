"""Dense, differentiable PyTorch-CPU restatement of the rasterizer.

TEST INFRASTRUCTURE ONLY.  Purpose: an *independent* derivation of the gradients -- autograd
through the forward, with the reference's gradient conventions (A.6) written as explicit
detach()/straight-through constructs -- against which the hand-derived analytic backward of
oracle/hgs_oracle.c (and through it the HIP kernels) is validated.  O(P * H * W) memory:
small cases only.  PARITY UNPINNED at the rasterizer boundary (see hgs_oracle.c header).

Call-site contract followed: /root/reference/hugs/renderer/gs_renderer.py:126-152.
"""
import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _sh_to_rgb(deg, sh, d):
    """sh [P,M,3], d [P,3] unit -> [P,3] (polynomial of spherical_harmonics.py:87-113)."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def _cov3d(scales, rots, mod):
    s = mod * scales
    r, x, y, z = rots[:, 0], rots[:, 1], rots[:, 2], rots[:, 3]  # NOT normalised (A.6 quirk 7)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    Mm = R * s[:, None, :]
    return Mm @ Mm.transpose(1, 2)


def rasterize(means3D, means2D, opacities, viewmatrix, projmatrix, campos, bg, tanfovx, tanfovy, H, W,
              shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0,
              scale_modifier=1.0):
    """Returns (color [3,H,W], radii [P] int32, aux dict). Differentiable w.r.t. the tensor inputs."""
    dt = means3D.dtype
    P = means3D.shape[0]
    if P == 0:
        return torch.zeros(3, H, W, dtype=dt), torch.zeros(0, dtype=torch.int32), {}
    V, F = viewmatrix.reshape(4, 4), projmatrix.reshape(4, 4)
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], 1)
    pv = (hom @ V)[:, :3]
    ph = hom @ F
    pw = 1.0 / (ph[:, 3] + 1e-7)
    # gradient sink: d(out)/d(means2D) == d(out)/d(ndc)   (A.6 quirk 6, quirk 9)
    sink = means2D[:, :2] - means2D[:, :2].detach()
    ndc = ph[:, :2] * pw[:, None] + sink
    visible = pv[:, 2] > 0.2

    if cov3D_precomp is not None:
        c = cov3D_precomp
        S3 = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                         -1).reshape(-1, 3, 3)
    else:
        S3 = _cov3d(scales, rotations, scale_modifier)

    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = torch.where(visible, pv[:, 2], torch.ones_like(pv[:, 2]))
    rx, ry = pv[:, 0] / tz, pv[:, 1] / tz
    x_in = (rx >= -limx) & (rx <= limx)
    y_in = (ry >= -limy) & (ry <= limy)
    # A.6 quirk 2: inside -> t.x is an independent variable; outside -> a constant
    tx = torch.where(x_in, pv[:, 0], (rx.clamp(-limx, limx) * tz).detach())
    ty = torch.where(y_in, pv[:, 1], (ry.clamp(-limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], -1).reshape(-1, 2, 3)
    Wr = V[:3, :3].t()
    Tm = J @ Wr
    S2 = Tm @ S3 @ Tm.transpose(1, 2)
    a, b, c_ = S2[:, 0, 0] + 0.3, S2[:, 0, 1], S2[:, 1, 1] + 0.3
    det = a * c_ - b * b
    ok = visible & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conx, cony, conz = c_ / det_s, -b / det_s, a / det_s
    with torch.no_grad():
        mid = 0.5 * (a + c_)
        sq = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        rad = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + sq, mid - sq)))
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        minx = ((px - rad) / 16).clamp(0, gx).to(torch.int64)
        maxx = ((px + rad + 15) / 16).clamp(0, gx).to(torch.int64)
        miny = ((py - rad) / 16).clamp(0, gy).to(torch.int64)
        maxy = ((py + rad + 15) / 16).clamp(0, gy).to(torch.int64)
        ok = ok & (maxx > minx) & (maxy > miny)
        radii = torch.where(ok, rad, torch.zeros_like(rad)).to(torch.int32)

    clamped = None
    if shs is not None:
        dvec = means3D - campos[None, :]
        dvec = dvec / dvec.norm(dim=1, keepdim=True)
        raw = _sh_to_rgb(sh_degree, shs, dvec) + 0.5
        clamped = raw < 0
        rgb = torch.clamp_min(raw, 0.0)
    else:
        rgb = colors_precomp

    # ---- dense blend: order by (fp32 depth bits, index) == stable sort on the 64-bit key ----
    with torch.no_grad():
        order = torch.sort(pv[:, 2].detach().to(torch.float32), stable=True).indices
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxl = xs.reshape(-1).to(dt)
    pyl = ys.reshape(-1).to(dt)
    txl, tyl = (xs.reshape(-1) // 16), (ys.reshape(-1) // 16)

    o = order
    dx = px[o][None, :] - pxl[:, None]
    dy = py[o][None, :] - pyl[:, None]
    power = -0.5 * (conx[o][None] * dx * dx + conz[o][None] * dy * dy) - cony[o][None] * dx * dy
    Gs = torch.exp(torch.clamp(power, max=0.0))
    araw = opacities.reshape(-1)[o][None, :] * Gs
    # A.6 quirk 1: straight-through over min(0.99, .)
    alpha = araw + (torch.clamp(araw, max=0.99) - araw).detach()
    with torch.no_grad():
        in_rect = (txl[:, None] >= minx[o][None]) & (txl[:, None] < maxx[o][None]) & \
                  (tyl[:, None] >= miny[o][None]) & (tyl[:, None] < maxy[o][None]) & ok[o][None]
        valid = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    T_incl = torch.cumprod(one_m, dim=1)
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], 1)
    with torch.no_grad():
        stop = (valid & (T_incl < 1e-4)).to(torch.int8)
        stopped = torch.cummax(stop, dim=1).values.bool()  # A.6 quirk 4
        contrib = valid & ~stopped
    w = torch.where(contrib, a_eff * T_excl, torch.zeros_like(a_eff))
    color = w @ rgb[o]  # [HW,3]
    T_final = torch.prod(torch.where(contrib, one_m, torch.ones_like(one_m)), dim=1)
    out = color + T_final[:, None] * bg[None, :]
    with torch.no_grad():
        idx1 = torch.arange(1, P + 1)[None, :].expand_as(contrib)
        n_contrib = torch.where(contrib, idx1, torch.zeros_like(idx1))
        # n_contrib counts entries of the *tile list* visited, not of the dense order
        in_list_rank = torch.cumsum(in_rect.to(torch.int64), 1)
        n_contrib = torch.where(contrib, in_list_rank, torch.zeros_like(in_list_rank)).max(dim=1).values
    aux = dict(final_T=T_final.reshape(H, W).detach(), n_contrib=n_contrib.reshape(H, W), clamped=clamped,
               xy=torch.stack([px, py], 1).detach(), conic=torch.stack([conx, cony, conz], 1).detach(),
               rgb=rgb.detach(), depth=pv[:, 2].detach())
    return out.t().reshape(3, H, W), radii, aux


def tanfov(fov):
    return math.tan(0.5 * fov)
