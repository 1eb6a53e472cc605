"""Executable specification of the C ABI in include/mrfa_hip.h  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Every entry point of libmrfa_hip.so restated with plain torch CPU ops on host memory addressed through the same raw
pointers / parameter structs.  Two uses, both in tests/ only:
  * `-m gpu` kernel tests run the HIP library and this emulator on the same seeded inputs and compare;
  * `-m "not gpu"` wiring tests inject it in place of the library (monkeypatching mrfa_amd.hip) so that the host-side
    engine and nn.Module programs can be checked against the reference goldens on a machine without a GPU.
mrfa_amd never imports this file; without libmrfa_hip.so the product raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F


STATS_SLOTS = 32        # MRFA_STATS_SLOTS (include/mrfa_hip.h)


def _flat(ptr: int, n: int, dtype=torch.float32) -> torch.Tensor:
    ct = {torch.float32: C.c_float, torch.float64: C.c_double, torch.int32: C.c_int}[dtype]
    return torch.frombuffer((ct * n).from_address(ptr), dtype=dtype)


def mat(ptr: int, rows: int, ld: int, cols: int) -> torch.Tensor:
    """[rows, cols] strided view (row stride ld) of host memory at ptr."""
    return torch.as_strided(_flat(ptr, (rows - 1) * ld + cols), (rows, cols), (ld, 1))


def nhwc(ptr, N, H, W, ld, Cc) -> torch.Tensor:
    return mat(ptr, N * H * W, ld, Cc).view(N, H, W, Cc) if ld == Cc else \
        torch.as_strided(_flat(ptr, (N * H * W - 1) * ld + Cc), (N, H, W, Cc), (H * W * ld, W * ld, ld, 1))


def vec(ptr, n, dtype=torch.float32):
    return _flat(ptr, n, dtype) if ptr else None


def _groups(p) -> int:
    """statistic groups of a parameter block (v7): 0 / 1 / absent = one"""
    return max(int(getattr(p, "groups", 0) or 0), 1)


def _obj(ref):
    return ref._obj if hasattr(ref, "_obj") else ref


class Emulator:
    def __init__(self):
        self._err = b""

    # ---------------------------------------------------------------- K14-K17: prior-motion stage (csrc/prior.hip)
    @staticmethod
    def _grid(H, W):
        xs = 2.0 * (torch.arange(W, dtype=torch.float32) / (W - 1)) - 1.0
        ys = 2.0 * (torch.arange(H, dtype=torch.float32) / (H - 1)) - 1.0
        return torch.stack([xs.view(1, W).expand(H, W), ys.view(H, 1).expand(H, W)], dim=-1)          # (H,W,2) as (x,y)

    def _gauss(self, kp, H, W, variance):
        d = self._grid(H, W).view(1, 1, H, W, 2) - kp.view(kp.shape[0], kp.shape[1], 1, 1, 2)
        return torch.exp(-0.5 * (d * d).sum(-1) / variance)                                           # (B,K,H,W)

    def mrfa_kp_gaussian_fwd(self, stream, kp, pos, B, K, H, W, variance, out, ldo):
        g = self._gauss(_flat(kp, B * K * 2).view(B, K, 2), H, W, variance)
        if pos:
            g = g + _flat(pos, K * H * W).view(1, K, H, W)
        nhwc(out, B, H, W, ldo, K).copy_(g.permute(0, 2, 3, 1))
        return 0

    def mrfa_kp_gaussian_bwd(self, stream, kp, B, K, H, W, variance, dout, lddo, dkp, dpos):
        k = _flat(kp, B * K * 2).view(B, K, 2).clone().requires_grad_(True)
        with torch.enable_grad():
            g = self._gauss(k, H, W, variance)
        d = nhwc(dout, B, H, W, lddo, K).permute(0, 3, 1, 2).contiguous()
        if dkp:
            (gk,) = torch.autograd.grad(g, k, d)
            _flat(dkp, B * K * 2).view(B, K, 2).add_(gk)
        if dpos:
            _flat(dpos, K * H * W).view(K, H, W).add_(d.sum(0))
        return 0

    def _prior(self, p, kd, ks, jd, js, bg):
        """-> heat (B,K1,H,W), motions (B,K1,H,W,2), warp (B,K1,C,H,W) exactly as dense_motion.py:36-85 computes them"""
        B, K, H, W, Cc = p.B, p.K, p.H, p.W, p.C
        var = 1.0 / p.inv_var
        heat = self._gauss(kd, H, W, var) - self._gauss(ks, H, W, var)
        heat = torch.cat([torch.zeros_like(heat[:, :1]), heat], dim=1)
        ident = self._grid(H, W).view(1, 1, H, W, 2)
        z = ident - kd.view(B, K, 1, 1, 2)
        if jd is not None:
            a, b_, c_, d = jd[..., 0], jd[..., 1], jd[..., 2], jd[..., 3]
            det = a * d - b_ * c_
            inv = torch.stack([torch.stack([d, -b_], -1), torch.stack([-c_, a], -1)], -2) / det[..., None, None]
            jac = torch.matmul(js.view(B, K, 2, 2), inv)
            z = torch.einsum("bkij,bkhwj->bkhwi", jac, z)
        d2s = z + ks.view(B, K, 1, 1, 2)
        bgg = ident.expand(B, 1, H, W, 2)
        if bg is not None:
            hom = torch.cat([bgg, torch.ones_like(bgg[..., :1])], -1)
            hom = torch.matmul(bg.view(B, 1, 1, 1, 3, 3), hom.unsqueeze(-1)).squeeze(-1)
            bgg = hom[..., :2] / hom[..., 2:3]
        motions = torch.cat([bgg, d2s], dim=1)
        src = nhwc(p.src, B, H, W, p.lds, Cc).permute(0, 3, 1, 2)
        rep = src.unsqueeze(1).expand(B, K + 1, Cc, H, W).reshape(B * (K + 1), Cc, H, W)
        warp = F.grid_sample(rep, motions.reshape(B * (K + 1), H, W, 2), mode="bilinear", padding_mode="zeros", align_corners=False)
        return heat, motions, warp.view(B, K + 1, Cc, H, W)

    def _prior_inputs(self, p, grad=False):
        B, K = p.B, p.K
        f = lambda ptr, n, shp: (_flat(ptr, n).view(*shp).clone().requires_grad_(grad) if ptr else None)
        return (f(p.kd, B * K * 2, (B, K, 2)), f(p.ks, B * K * 2, (B, K, 2)), f(p.jd, B * K * 4, (B, K, 4)), f(p.js, B * K * 4, (B, K, 4)),
                f(p.bg, B * 9, (B, 9)))

    def mrfa_prior_motion_fwd(self, stream, pref):
        p = _obj(pref)
        B, K, H, W, Cc = p.B, p.K, p.H, p.W, p.C
        K1 = K + 1
        heat, motions, warp = self._prior(p, *self._prior_inputs(p))
        nhwc(p.motions, B * K1, H, W, p.ldm, 2).copy_(motions.reshape(B * K1, H, W, 2))
        inp = nhwc(p.inp, B, H, W, p.ldi, K1 * (Cc + 1))
        inter = torch.cat([heat.unsqueeze(2), warp], dim=2)                                           # (B,K1,C+1,H,W)
        inp.copy_(inter.reshape(B, K1 * (Cc + 1), H, W).permute(0, 2, 3, 1))
        if p.sparse:
            _flat(p.sparse, B * K1 * Cc * H * W).view(B, K1, Cc, H, W).copy_(warp)
        return 0

    def mrfa_prior_motion_bwd(self, stream, pref):
        p = _obj(pref)
        B, K, H, W, Cc = p.B, p.K, p.H, p.W, p.C
        K1 = K + 1
        ins = self._prior_inputs(p, grad=True)
        with torch.enable_grad():
            heat, motions, warp = self._prior(p, *ins)
        dinp = nhwc(p.dinp, B, H, W, p.lddi, K1 * (Cc + 1)).permute(0, 3, 1, 2).reshape(B, K1, Cc + 1, H, W)
        dwarp = dinp[:, :, 1:].clone()
        if p.dsparse:
            dwarp = dwarp + _flat(p.dsparse, B * K1 * Cc * H * W).view(B, K1, Cc, H, W)
        outs, grads = [heat, warp], [dinp[:, :, 0].contiguous(), dwarp]
        if p.dmotions:
            outs.append(motions)
            grads.append(nhwc(p.dmotions, B * K1, H, W, p.ldm, 2).reshape(B, K1, H, W, 2).clone())
        live = [t for t in ins if t is not None]
        g = torch.autograd.grad(outs, live, grads, allow_unused=True)
        it = iter(g)
        for t, dst, n in zip(ins, (p.dkd, p.dks, p.djd, p.djs, p.dbg), (B * K * 2, B * K * 2, B * K * 4, B * K * 4, B * 9)):
            if t is None:
                continue
            gt = next(it)
            if dst and gt is not None:
                _flat(dst, n).add_(gt.reshape(-1))
        return 0

    def mrfa_softmax_combine_fwd(self, stream, logit, ldl, motions, ldm, B, H, W, K1, deformation, mask, logit_nchw):
        l = nhwc(logit, B, H, W, ldl, K1)
        m = F.softmax(l, dim=-1)
        mo = nhwc(motions, B * K1, H, W, ldm, 2).reshape(B, K1, H, W, 2).permute(0, 2, 3, 1, 4)
        _flat(deformation, B * H * W * 2).view(B, H, W, 2).copy_((mo * m.unsqueeze(-1)).sum(3))
        _flat(mask, B * K1 * H * W).view(B, K1, H, W).copy_(m.permute(0, 3, 1, 2))
        _flat(logit_nchw, B * K1 * H * W).view(B, K1, H, W).copy_(l.permute(0, 3, 1, 2))
        return 0

    def mrfa_softmax_combine_bwd(self, stream, motions, ldm, B, H, W, K1, mask, ddeformation, dmask, dlogit_nchw, dlogit, lddl, dmotions):
        m = _flat(mask, B * K1 * H * W).view(B, K1, H, W).permute(0, 2, 3, 1)                        # (B,H,W,K1)
        mo = nhwc(motions, B * K1, H, W, ldm, 2).reshape(B, K1, H, W, 2).permute(0, 2, 3, 1, 4)
        dd = _flat(ddeformation, B * H * W * 2).view(B, H, W, 2) if ddeformation else torch.zeros(B, H, W, 2)
        dm = (mo * dd.unsqueeze(3)).sum(-1)
        if dmask:
            dm = dm + _flat(dmask, B * K1 * H * W).view(B, K1, H, W).permute(0, 2, 3, 1)
        dl = m * (dm - (m * dm).sum(-1, keepdim=True))
        if dlogit_nchw:
            dl = dl + _flat(dlogit_nchw, B * K1 * H * W).view(B, K1, H, W).permute(0, 2, 3, 1)
        nhwc(dlogit, B, H, W, lddl, K1).add_(dl)
        if dmotions:
            nhwc(dmotions, B * K1, H, W, ldm, 2).add_((m.unsqueeze(-1) * dd.unsqueeze(3)).permute(0, 3, 1, 2, 4).reshape(B * K1, H, W, 2))
        return 0

    def _kp_head(self, lg, jm, temperature):
        B, H, W, K = lg.shape
        heat = F.softmax(lg.reshape(B, H * W, K) / temperature, dim=1)
        kp = torch.einsum("bpk,pc->bkc", heat, self._grid(H, W).reshape(H * W, 2))
        jac = torch.einsum("bpk,bpj->bkj", heat, jm.reshape(B, H * W, 4)) if jm is not None else None
        return kp, jac, heat

    def mrfa_kp_head_fwd(self, stream, logits, ldl, jm, ldj, B, H, W, K, temperature, kp, jac, stat):
        lg = nhwc(logits, B, H, W, ldl, K)
        j = nhwc(jm, B, H, W, ldj, 4) if jm else None
        k, ja, _ = self._kp_head(lg, j, temperature)
        _flat(kp, B * K * 2).view(B, K, 2).copy_(k)
        if jm:
            _flat(jac, B * K * 4).view(B, K, 4).copy_(ja)
        s = lg.reshape(B, H * W, K) / temperature
        mx = s.max(dim=1).values
        st = _flat(stat, B * K * 2).view(B, K, 2)
        st[..., 0] = mx
        st[..., 1] = 1.0 / torch.exp(s - mx.unsqueeze(1)).sum(1)
        return 0

    def mrfa_kp_head_bwd(self, stream, logits, ldl, jm, ldj, B, H, W, K, temperature, kp, jac, stat, dkp, djac, dlogits, lddl, djm, lddj):
        lg = nhwc(logits, B, H, W, ldl, K).clone().requires_grad_(True)
        j = nhwc(jm, B, H, W, ldj, 4).clone().requires_grad_(True) if jm else None
        with torch.enable_grad():
            k, ja, _ = self._kp_head(lg, j, temperature)
        outs, grads = [], []
        if dkp:
            outs.append(k); grads.append(_flat(dkp, B * K * 2).view(B, K, 2).clone())
        if jm and djac:
            outs.append(ja); grads.append(_flat(djac, B * K * 4).view(B, K, 4).clone())
        if not outs:
            return 0
        ins = [lg] + ([j] if (jm and djac) else [])
        g = torch.autograd.grad(outs, ins, grads, allow_unused=True)
        nhwc(dlogits, B, H, W, lddl, K).add_(g[0])
        if jm and djac and djm and g[1] is not None:
            nhwc(djm, B, H, W, lddj, 4).add_(g[1])
        return 0

    # ---------------------------------------------------------------- K22: training losses
    def mrfa_maxpool2_fwd(self, stream, x, ldx, N, H, W, Cc, y, ldy):
        v = F.max_pool2d(nhwc(x, N, H, W, ldx, Cc).permute(0, 3, 1, 2), 2)
        nhwc(y, N, H // 2, W // 2, ldy, Cc).copy_(v.permute(0, 2, 3, 1))
        return 0

    def mrfa_maxpool2_bwd(self, stream, x, ldx, N, H, W, Cc, dy, lddy, dx, lddx):
        xx = nhwc(x, N, H, W, ldx, Cc).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        with torch.enable_grad():
            out = F.max_pool2d(xx, 2)
        (g,) = torch.autograd.grad(out, xx, nhwc(dy, N, H // 2, W // 2, lddy, Cc).permute(0, 3, 1, 2).contiguous())
        nhwc(dx, N, H, W, lddx, Cc).add_(g.permute(0, 2, 3, 1))
        return 0

    def mrfa_maxpool3s2_fwd(self, stream, x, ldx, N, H, W, Cc, y, ldy):
        v = F.max_pool2d(nhwc(x, N, H, W, ldx, Cc).permute(0, 3, 1, 2), 3, 2, 1)
        nhwc(y, N, v.shape[2], v.shape[3], ldy, Cc).copy_(v.permute(0, 2, 3, 1))
        return 0

    def mrfa_maxpool3s2_bwd(self, stream, x, ldx, N, H, W, Cc, dy, lddy, dx, lddx):
        xx = nhwc(x, N, H, W, ldx, Cc).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        with torch.enable_grad():
            out = F.max_pool2d(xx, 3, 2, 1)
        (g,) = torch.autograd.grad(out, xx, nhwc(dy, N, out.shape[2], out.shape[3], lddy, Cc).permute(0, 3, 1, 2).contiguous())
        nhwc(dx, N, H, W, lddx, Cc).add_(g.permute(0, 2, 3, 1))
        return 0

    def mrfa_l1_diff_fwd(self, stream, x, ldx, y, ldy, rows, Cc, coef, out_sum):
        vec(out_sum, 1, torch.float64).add_(coef * (mat(x, rows, ldx, Cc) - mat(y, rows, ldy, Cc)).abs().double().sum())
        return 0

    def mrfa_l1_diff_bwd(self, stream, x, ldx, y, ldy, rows, Cc, gscale, coef, dx, lddx):
        g = (float(vec(gscale, 1)[0]) if gscale else 1.0) * coef
        mat(dx, rows, lddx, Cc).add_(g * torch.sign(mat(x, rows, ldx, Cc) - mat(y, rows, ldy, Cc)))
        return 0

    def mrfa_antialias_down_bwd(self, stream, dy, lddy, N, Cc, H, W, kern, k, stride, dx):
        ker = _flat(kern, k * k).view(1, 1, k, k).expand(Cc, 1, k, k)
        ka = k // 2
        img = torch.zeros(N, Cc, H, W, requires_grad=True)
        with torch.enable_grad():
            v = F.conv2d(F.pad(img, (ka, ka, ka, ka)), ker, groups=Cc)[:, :, ::stride, ::stride]
        (g,) = torch.autograd.grad(v, img, nhwc(dy, N, H // stride, W // stride, lddy, Cc).permute(0, 3, 1, 2).contiguous())
        _flat(dx, N * Cc * H * W).view(N, Cc, H, W).add_(g)
        return 0

    # ---------------------------------------------------------------- K21: MTIA prior (TokenPose_B)
    def mrfa_subsample_fwd(self, stream, x, ldx, N, H, W, Cc, stride, y, ldy):
        nhwc(y, N, H // stride, W // stride, ldy, Cc).copy_(nhwc(x, N, H, W, ldx, Cc)[:, ::stride, ::stride])
        return 0

    def mrfa_subsample_bwd(self, stream, dy, lddy, N, H, W, Cc, stride, dx, lddx):
        nhwc(dx, N, H, W, lddx, Cc)[:, ::stride, ::stride].add_(nhwc(dy, N, H // stride, W // stride, lddy, Cc))
        return 0

    def mrfa_upsample_add_act_fwd(self, stream, lo, ldl, N, Hl, Wl, Cc, f, base, ldb, relu, y, ldy):
        v = nhwc(lo, N, Hl, Wl, ldl, Cc).repeat_interleave(f, dim=1).repeat_interleave(f, dim=2) + nhwc(base, N, Hl * f, Wl * f, ldb, Cc)
        nhwc(y, N, Hl * f, Wl * f, ldy, Cc).copy_(F.relu(v) if relu else v)
        return 0

    def mrfa_upsample_add_act_bwd(self, stream, y, ldy, dy, lddy, N, Hl, Wl, Cc, f, relu, dlo, lddl, dbase, lddb):
        g = nhwc(dy, N, Hl * f, Wl * f, lddy, Cc).clone()
        if relu:
            g = torch.where(nhwc(y, N, Hl * f, Wl * f, ldy, Cc) > 0, g, torch.zeros_like(g))
        if dbase:
            nhwc(dbase, N, Hl * f, Wl * f, lddb, Cc).add_(g)
        if dlo:
            nhwc(dlo, N, Hl, Wl, lddl, Cc).add_(g.view(N, Hl, f, Wl, f, Cc).sum(dim=(2, 4)))
        return 0

    def mrfa_layernorm_fwd(self, stream, x, ldx, rows, Cc, gamma, beta, eps, y, ldy, mean, rstd):
        xx = mat(x, rows, ldx, Cc)
        m = xx.mean(1)
        r = 1.0 / torch.sqrt(xx.var(1, unbiased=False) + eps)
        mat(y, rows, ldy, Cc).copy_((xx - m[:, None]) * r[:, None] * vec(gamma, Cc) + vec(beta, Cc))
        vec(mean, rows).copy_(m)
        vec(rstd, rows).copy_(r)
        return 0

    def mrfa_layernorm_bwd(self, stream, x, ldx, dy, lddy, rows, Cc, gamma, mean, rstd, dx, lddx, dgamma, dbeta, scratch=None):
        if scratch:                  # v8: the caller's promise -- zeroed, fresh per call (the library leaves its partial sums and the ticket count in it)
            sc = vec(scratch, 16 * 2 * Cc + 1)
            assert float(sc.abs().max()) == 0.0, "layernorm_bwd: scratch must be zeroed"
            sc.fill_(1.0)
        xh = (mat(x, rows, ldx, Cc) - vec(mean, rows)[:, None]) * vec(rstd, rows)[:, None]
        d = mat(dy, rows, lddy, Cc)
        g = d * vec(gamma, Cc)
        mat(dx, rows, lddx, Cc).add_(vec(rstd, rows)[:, None] * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True)))
        if dgamma:
            vec(dgamma, Cc).add_((d * xh).sum(0))
        if dbeta:
            vec(dbeta, Cc).add_(d.sum(0))
        return 0

    def mrfa_gelu_fwd(self, stream, x, ldx, rows, Cc, y, ldy):
        mat(y, rows, ldy, Cc).copy_(F.gelu(mat(x, rows, ldx, Cc)))
        return 0

    def mrfa_gelu_bwd(self, stream, x, ldx, dy, lddy, rows, Cc, dx, lddx):
        v = mat(x, rows, ldx, Cc)
        cdf = 0.5 * (1 + torch.erf(v * 0.7071067811865476))
        pdf = 0.3989422804014327 * torch.exp(-0.5 * v * v)
        mat(dx, rows, lddx, Cc).add_(mat(dy, rows, lddy, Cc) * (cdf + v * pdf))
        return 0

    @staticmethod
    def _qkv(qkv, ld, B, n, heads, d):
        t = mat(qkv, B * n, ld, 3 * heads * d).view(B, n, 3, heads, d)
        return [t[:, :, i].permute(0, 2, 1, 3) for i in range(3)]            # (B, heads, n, d) each

    def mrfa_attention_fwd(self, stream, qkv, ld, B, n, heads, d, scale, out, ldo, lse):
        q, k, v = self._qkv(qkv, ld, B, n, heads, d)
        s = torch.einsum("bhid,bhjd->bhij", q, k) * scale
        vec(lse, B * heads * n).copy_(torch.logsumexp(s, -1).reshape(-1))
        o = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v)
        mat(out, B * n, ldo, heads * d).copy_(o.permute(0, 2, 1, 3).reshape(B * n, heads * d))
        return 0

    def mrfa_attention_bwd(self, stream, qkv, ld, out, ldo, dout, lddo, lse, delta, B, n, heads, d, scale, dqkv, lddq):
        q, k, v = self._qkv(qkv, ld, B, n, heads, d)
        p = (torch.einsum("bhid,bhjd->bhij", q, k) * scale).softmax(-1)
        do = mat(dout, B * n, lddo, heads * d).view(B, n, heads, d).permute(0, 2, 1, 3)
        o = mat(out, B * n, ldo, heads * d).view(B, n, heads, d).permute(0, 2, 1, 3)
        dl = (do * o).sum(-1)
        vec(delta, B * heads * n).copy_(dl.reshape(-1))
        dp = torch.einsum("bhid,bhjd->bhij", do, v)
        ds = p * (dp - dl[..., None])
        dq = torch.einsum("bhij,bhjd->bhid", ds, k) * scale
        dk = torch.einsum("bhij,bhid->bhjd", ds, q) * scale
        dv = torch.einsum("bhij,bhid->bhjd", p, do)
        g = torch.stack([t.permute(0, 2, 1, 3) for t in (dq, dk, dv)], dim=2).reshape(B * n, 3 * heads * d)
        mat(dqkv, B * n, lddq, 3 * heads * d).add_(g)
        return 0

    # ---------------------------------------------------------------- misc
    def mrfa_version(self):
        return 9              # MRFA_ABI_VERSION of include/mrfa_hip.h

    def mrfa_last_error(self):
        return self._err

    def mrfa_conv2d_last_config(self):
        return 0

    def mrfa_set_mfma_mode(self, mode):
        self._mfma = mode            # the specification is the same fp32 convolution in every mode
        return 0

    def mrfa_set_tuning(self, key, value):
        """kernel-selection knobs change which kernel runs, never the result: nothing to emulate"""
        return 0

    def mrfa_get_mfma_mode(self):
        return getattr(self, "_mfma", 0)

    def mrfa_build_ktab(self, tab, Cc, R, S, pad, flip):
        K = R * S * Cc
        KP = (K + 31) // 32 * 32
        for k in range(KP):
            if k >= K:
                tab[k] = -1
                continue
            tap, c = divmod(k, Cc)
            r, s = divmod(tap, S)
            tab[k] = ((r - pad) + 128) | (((s - pad) + 128) << 8) | (c << 16)
        return 0

    def mrfa_pack_conv_weight(self, stream, src, dst, Cout, Cin, R, S, mode):
        T = R * S
        overwrite = bool(mode & 16)
        mode &= 15
        if mode == 4:
            g = _flat(src, T * Cout * Cin).view(T, Cout, Cin)
            d = _flat(dst, Cout * Cin * T).view(Cout, Cin, T)
            d.copy_(g.permute(1, 2, 0) if overwrite else d + g.permute(1, 2, 0))
            return 0
        if mode == 6:
            g = _flat(src, T * Cout * Cin).view(Cout, T, Cin)
            d = _flat(dst, Cout * Cin * T).view(Cout, Cin, T)
            d.copy_(g.permute(0, 2, 1) if overwrite else d + g.permute(0, 2, 1))
            return 0
        w = _flat(src, Cout * Cin * T).view(Cout, Cin, T)
        if mode == 5:
            _flat(dst, Cout * T * Cin).view(Cout, T, Cin).copy_(w.permute(0, 2, 1))
            return 0
        if mode == 7:
            _flat(dst, Cin * T * Cout).view(Cin, T, Cout).copy_(w.flip(2).permute(1, 2, 0))
            return 0
        if mode == 0:
            cop, cip = (Cout + 127) // 128 * 128, (Cin + 31) // 32 * 32
            d = _flat(dst, T * cop * cip).view(T, cop, cip)
            d.zero_()
            d[:, :Cout, :Cin] = w.permute(2, 0, 1)
        elif mode == 1:
            cop, kp = (Cout + 127) // 128 * 128, (T * Cin + 31) // 32 * 32
            d = _flat(dst, cop * kp).view(cop, kp)
            d.zero_()
            d[:Cout, :T * Cin] = w.permute(0, 2, 1).reshape(Cout, T * Cin)
        elif mode == 2:
            cip, cop = (Cin + 127) // 128 * 128, (Cout + 31) // 32 * 32
            d = _flat(dst, T * cip * cop).view(T, cip, cop)
            d.zero_()
            d[:, :Cin, :Cout] = w.flip(2).permute(2, 1, 0)
        elif mode == 3:
            cip, kp = (Cin + 127) // 128 * 128, (T * Cout + 31) // 32 * 32
            d = _flat(dst, cip * kp).view(cip, kp)
            d.zero_()
            d[:Cin, :T * Cout] = w.flip(2).permute(1, 2, 0).reshape(Cin, T * Cout)
        else:
            return 1
        return 0

    # ---------------------------------------------------------------- conv / gemm
    @staticmethod
    def _prep_input(p, x_ptr):
        x = nhwc(x_ptr, p.N, p.Hin, p.Win, p.ldx, p.Cin).permute(0, 3, 1, 2).clone()
        if p.in_scale:
            G = max(int(getattr(p, "groups", 0)), 1)          # v9: prologue vectors [groups][Cin], sample n uses row n // (N / groups)
            sc = vec(p.in_scale, G * p.Cin).view(G, 1, p.Cin, 1, 1)
            sh = vec(p.in_shift, G * p.Cin).view(G, 1, p.Cin, 1, 1)
            x = (x.view(G, p.N // G, p.Cin, p.Hin, p.Win) * sc + sh).view(p.N, p.Cin, p.Hin, p.Win)
            if p.in_relu:
                x = F.relu(x)
        if p.ups == 1:
            x = F.interpolate(x, scale_factor=2)
        return x

    def mrfa_conv2d_nhwc(self, stream, pref):
        p = _obj(pref)
        T = p.R * p.S
        nb = max(p.nbatch, 1)
        for b in range(nb):
            xp = p.x + 4 * b * p.x_bs
            wp = p.w + 4 * b * p.w_bs
            yp = p.y + 4 * b * p.y_bs
            x = self._prep_input(p, xp)
            if p.kflat > 0:
                wm = mat(wp, p.Cout, p.w_ld, p.kflat)
                w = wm.reshape(p.Cout, T, p.Cin).permute(0, 2, 1).reshape(p.Cout, p.Cin, p.R, p.S)
            else:
                wt = torch.as_strided(_flat(wp, (T - 1) * p.w_tap + (p.Cout - 1) * p.w_ld + p.Cin), (T, p.Cout, p.Cin),
                                      (p.w_tap, p.w_ld, 1))
                w = wt.permute(1, 2, 0).reshape(p.Cout, p.Cin, p.R, p.S)
            acc = F.conv2d(x, w.contiguous(), None, padding=p.pad, stride=max(p.stride, 1)) * p.alpha
            if p.ups == 2:           # data gradient of a fused-upsample layer: the 3x3 data gradient on the high-resolution grid, 2x2 sum-pooled
                acc = F.avg_pool2d(acc, 2) * 4.0
            assert acc.shape[2] == p.Hout and acc.shape[3] == p.Wout, (acc.shape, p.Hout, p.Wout)
            v = acc.permute(0, 2, 3, 1)
            y = nhwc(yp, p.N, p.Hout, p.Wout, p.ldy, p.Cout)
            if b == 0 and not p.accumulate and self.mrfa_conv2d_split_k(pref) > 1:
                # v8: what a caller promises to a launch that splits K
                if p.y_zero:
                    assert float(y.abs().max()) == 0.0, "y_zero = 1 but y does not hold zeros"
                if p.sk_ticket:
                    nt = -(-p.N * p.Hout * p.Wout // 32) * -(-p.Cout // 32) * nb
                    tk = vec(p.sk_ticket, nt, torch.int32)
                    assert int(tk.abs().max()) == 0, "sk_ticket words must be zero on entry (fresh per call)"
                    tk.fill_(2)                      # (the library leaves the tickets it drew in them)
            if p.bias:
                v = v + vec(p.bias, p.Cout)
            if p.out_scale:
                v = v * vec(p.out_scale, p.Cout) + vec(p.out_shift, p.Cout)
            if p.res:
                v = v + nhwc(p.res, p.N, p.Hout, p.Wout, p.ldr, p.Cout)
            if p.relu:
                v = F.relu(v)
            if p.mask:               # fused ReLU backward of the producer of the tensor whose gradient this launch writes
                v = v * (nhwc(p.mask, p.N, p.Hout, p.Wout, p.ldm, p.Cout) > 0)
            if p.accumulate:
                v = v + y
            y.copy_(v)
            # v7, statistic groups: samples [g N / G, (g + 1) N / G) accumulate into statistics block g, with group g's bst_* vectors
            G = _groups(p)
            assert p.N % G == 0 and (G == 1 or nb == 1)
            n = p.N // G
            for g in range(G):
                vg = v[g * n:(g + 1) * n]
                if p.stats and p.bst_x:  # v6: first phase of the BatchNorm backward whose output's gradient this launch wrote (mrfa_conv_params.bst_*)
                    st = vec(p.stats + 8 * g * STATS_SLOTS * 2 * p.Cout, 2 * p.Cout, torch.float64)
                    xr = nhwc(p.bst_x, p.N, p.Hout, p.Wout, p.bst_ldx, p.Cout)[g * n:(g + 1) * n]
                    gv = lambda ptr: vec(ptr + 4 * g * p.Cout, p.Cout)
                    u = xr * gv(p.bst_scale) + gv(p.bst_shift)
                    du = torch.where(u > 0, vg, torch.zeros_like(vg)) if p.bst_relu else vg
                    xhat = (xr - gv(p.bst_mean)) * gv(p.bst_invstd)
                    st[:p.Cout] += du.reshape(-1, p.Cout).double().sum(0)
                    st[p.Cout:] += (du * xhat).reshape(-1, p.Cout).double().sum(0)
                elif p.stats:
                    st = vec(p.stats + 8 * g * STATS_SLOTS * 2 * p.Cout, 2 * p.Cout, torch.float64)
                    flat = vg.reshape(-1, p.Cout).double()
                    st[:p.Cout] += flat.sum(0)
                    st[p.Cout:] += (flat * flat).sum(0)
        if p.fin_scale:              # v6: the BatchNorm that follows is finished inside the call (mrfa_conv_params.fin_*)
            assert p.stats and p.fin_counter and p.fin_count > 0
            return self.mrfa_bn_finalize_groups(stream, p.stats, p.fin_count, p.fin_gamma, p.fin_beta, p.fin_rmean, p.fin_rvar, p.fin_momentum,
                                                p.fin_eps, p.Cout, _groups(p), p.fin_scale, p.fin_shift, p.fin_mean, p.fin_invstd)
        return 0

    def mrfa_conv2d_split_k(self, pref):
        """v8.  The emulator never splits anything; it ANSWERS like a library that splits K on launches with few output tiles, so that callers' handling of
        sk_ticket / y_zero runs on the CPU too (mrfa_conv2d_nhwc below checks their promises)"""
        p = _obj(pref)
        if p.splitk > 1:
            return p.splitk
        M = p.N * p.Hout * p.Wout
        K = p.kflat if p.kflat > 0 else p.R * p.S * p.Cin
        small = (p.kflat == 0 and not p.ups and not p.in_scale and p.Cin % 16 == 0 and M <= 65536 and p.Cout <= 640
                 and K <= 1152 and 2.0 * M * p.Cout * K <= 1.3e9)
        tiles = -(-M // 128) * -(-p.Cout // 128) * max(p.nbatch, 1)
        return 2 if (p.splitk == 0 and not small and tiles < 384 and K >= 128 and p.stride <= 1) else 1

    @staticmethod
    def _lean_shape(p, Cin, Cout, Hout, Wout, stride) -> bool:
        """the shapes conv_lean.hip / wgrad_lean.hip take in the library's default (split-operand) matrix mode: the keypoint encoder's 3x3 stride-1 layers
        with 32, 64 or 128 channels on maps that tile into their patches"""
        if p.kflat > 0 or p.ups or p.R != 3 or p.S != 3 or p.pad != 1 or stride > 1 or p.nbatch > 1:
            return False
        if 2.0 * p.N * Hout * Wout * Cout * 9.0 * Cin > 2.6e9 or Cout % 32 or Cout > 128:
            return False
        return (Cin in (32, 64) and Wout % 32 == 0) or (Cin == 128 and Wout % 16 == 0)

    def mrfa_conv2d_reads_fp32_weights(self, pref):
        """the specification always computes from the fp32 layout"""
        return 1

    def mrfa_conv2d_wgrad_lean_supported(self, pref):
        p = _obj(pref)
        return int(self._lean_shape(p, p.Cin, p.Cout, p.Hout, p.Wout, p.stride) and not p.dbias and ((p.Cin == 32 and p.Cout == 32) or (p.Cin % 64 == 0 and p.Cout % 64 == 0)))

    def mrfa_conv2d_wgrad_groups_supported(self, pref):
        """the library's rule: a prologue with one vector pair per statistic group exists in wgrad_lean.hip only"""
        p = _obj(pref)
        if p.groups <= 1 or not p.in_scale:
            return 1
        ok = self._lean_shape(p, p.Cin, p.Cout, p.Hout, p.Wout, p.stride) and not p.dbias and p.N % p.groups == 0
        return int(ok and ((p.Cin == 32 and p.Cout == 32) or (p.Cin % 64 == 0 and p.Cout % 64 == 0)))

    def mrfa_conv2d_groups_supported(self, pref):
        """the library's rule (the emulator itself honours `groups` for every shape)"""
        p = _obj(pref)
        if p.groups <= 1:
            return 1
        if p.in_scale:                                   # prologue vectors per group: conv_lean.hip only
            return int(bool(p.w_split) and p.splitk <= 1 and not p.mask and p.N % p.groups == 0 and self._lean_shape(p, p.Cin, p.Cout, p.Hout, p.Wout, p.stride))
        if not (p.stats or p.fin_scale or p.bst_x):
            return 1
        rows = p.N // p.groups * p.Hout * p.Wout
        small = (p.kflat == 0 and not p.ups and not p.in_scale and p.Cin % 16 == 0 and p.N * p.Hout * p.Wout <= 65536 and p.Cout <= 640
                 and p.R * p.S * p.Cin <= 1152 and 2.0 * p.N * p.Hout * p.Wout * p.Cout * p.R * p.S * p.Cin <= 1.3e9)
        return int(p.nbatch <= 1 and p.splitk <= 1 and p.N % p.groups == 0 and (rows % 128 == 0 or (small and rows % 64 == 0)))

    def mrfa_conv2d_wgrad_nhwc(self, stream, pref):
        p = _obj(pref)
        T = p.R * p.S
        nb = max(p.nbatch, 1)
        for b in range(nb):
            x = self._prep_input(p, p.x + 4 * b * p.x_bs)
            dy = nhwc(p.dy + 4 * b * p.dy_bs, p.N, p.Hout, p.Wout, p.ldy, p.Cout).permute(0, 3, 1, 2)
            with torch.enable_grad():
                w = torch.zeros(p.Cout, p.Cin, p.R, p.S, requires_grad=True)
                y = F.conv2d(x, w, None, padding=p.pad, stride=max(p.stride, 1))
                (gw,) = torch.autograd.grad(y, w, dy.contiguous())
            dw = _flat(p.dw + 4 * b * p.dw_bs, T * p.Cout * p.Cin).view(T, p.Cout, p.Cin)
            dw += p.alpha * gw.reshape(p.Cout, p.Cin, T).permute(2, 0, 1)
            if p.dbias:
                vec(p.dbias, p.Cout).add_(dy.sum(dim=(0, 2, 3)))
        return 0

    def mrfa_conv_fewout_fwd(self, stream, x, ldx, N, H, W, Cin, w, bias, y, ldy, Cout, R, pad, accumulate):
        T = R * R
        xx = nhwc(x, N, H, W, ldx, Cin).permute(0, 3, 1, 2)
        ww = _flat(w, Cout * T * Cin).view(Cout, T, Cin).permute(0, 2, 1).reshape(Cout, Cin, R, R)
        v = F.conv2d(xx, ww.contiguous(), vec(bias, Cout) if bias else None, padding=pad).permute(0, 2, 3, 1)
        Ho, Wo = H + 2 * pad - R + 1, W + 2 * pad - R + 1
        o = nhwc(y, N, Ho, Wo, ldy, Cout)
        o.copy_(o + v if accumulate else v)
        return 0

    def mrfa_conv_fewout_wgrad(self, stream, x, ldx, N, H, W, Cin, dy, lddy, Cout, R, pad, dw, dbias):
        T = R * R
        Ho, Wo = H + 2 * pad - R + 1, W + 2 * pad - R + 1
        xx = nhwc(x, N, H, W, ldx, Cin).permute(0, 3, 1, 2).contiguous()
        g = nhwc(dy, N, Ho, Wo, lddy, Cout).permute(0, 3, 1, 2).contiguous()
        with torch.enable_grad():
            ww = torch.zeros(Cout, Cin, R, R, requires_grad=True)
            (gw,) = torch.autograd.grad(F.conv2d(xx, ww, None, padding=pad), ww, g)
        _flat(dw, Cout * T * Cin).view(Cout, T, Cin).add_(gw.reshape(Cout, Cin, T).permute(0, 2, 1))
        if dbias:
            vec(dbias, Cout).add_(g.sum(dim=(0, 2, 3)))
        return 0

    def mrfa_conv2d_stride_supported(self, pref):
        """shape rule of the library's one-wave-per-tile kernel (the emulator itself honours `stride` everywhere)"""
        p = _obj(pref)
        if p.stride == 2 and p.kflat > 0:        # flat-K launches: the fp32 tile kernel's strided gather
            return int(not p.ups and p.nbatch <= 1 and p.splitk <= 1 and bool(p.ktab))
        return int(p.stride == 2 and p.kflat == 0 and not p.ups and not p.in_scale and p.Cin % 16 == 0 and p.ldx % 4 == 0 and p.nbatch <= 1
                   and p.N * p.Hout * p.Wout <= 65536 and p.Cout <= 640 and p.R * p.S * p.Cin <= 1152)

    def mrfa_conv2d_wgrad_stride_supported(self, pref):
        p = _obj(pref)
        return int(p.stride == 2 and p.kflat == 0 and not p.ups and not p.in_scale and p.Cin % 32 == 0 and p.Cout % 32 == 0 and p.nbatch <= 1
                   and p.N * p.Hout * p.Wout <= 65536 and p.Cout <= 640 and p.Cin <= 640)

    def mrfa_conv2d_mask_supported(self, pref):
        """shape rule of the library's patch-tiled kernel (the emulator itself honours `mask` everywhere)"""
        p = _obj(pref)
        return int(p.kflat == 0 and p.R == 3 and p.S == 3 and p.pad == 1 and p.Wout % 32 == 0 and p.Cin % 32 == 0 and p.Cout >= 32
                   and p.ldy % 4 == 0 and p.nbatch <= 1)

    def mrfa_conv_fewout_dgrad(self, stream, dy, lddy, N, H, W, Cout, w, dx, lddx, Cin, R, pad, accumulate, mask=None, ldm=0):
        T = R * R
        g = nhwc(dy, N, H, W, lddy, Cout).permute(0, 3, 1, 2).contiguous()
        ww = _flat(w, Cout * T * Cin).view(Cout, T, Cin).permute(0, 2, 1).reshape(Cout, Cin, R, R).contiguous()
        v = F.conv_transpose2d(g, ww, padding=pad).permute(0, 2, 3, 1)
        if mask:
            v = v * (nhwc(mask, N, H, W, ldm, Cin) > 0)
        o = nhwc(dx, N, H, W, lddx, Cin)
        o.copy_(o + v if accumulate else v)
        return 0

    def mrfa_conv2d_phase_dgrad_supported(self, pref):
        """mirrors the library's answer for shapes, independent of tuning knobs (the emulator computes the same result either way)"""
        p = _obj(pref)
        return int(p.ups == 2 and bool(p.w_phase) and p.R == 3 and p.S == 3 and p.pad == 1 and p.Hin == 2 * p.Hout and p.Win == 2 * p.Wout
                   and p.Wout % 32 == 0 and p.Hout % 8 == 0 and p.Cin % 32 == 0 and not p.in_scale)

    def mrfa_conv2d_bwdstats_supported(self, pref):
        p = _obj(pref)
        return 1 if (p.stats and not p.fin_scale and p.stride >= 0 and p.kflat == 0) else 0

    def mrfa_conv_fewout_dgrad_supported(self, Cin, Cout, R, pad, W, lddx):
        return int(R == 3 and pad == 1 and Cout in (1, 2) and Cin in (64, 128, 256) and lddx % 4 == 0 and W >= 4)

    # ---------------------------------------------------------------- batch norm
    def mrfa_bn_stats(self, stream, x, ldx, rows, Cc, stats):
        v = mat(x, rows, ldx, Cc).double()
        st = vec(stats, 2 * Cc, torch.float64)
        st[:Cc] += v.sum(0)
        st[Cc:] += (v * v).sum(0)
        return 0

    def mrfa_bn_finalize(self, stream, stats, count, gamma, beta, rmean, rvar, momentum, eps, Cc, train, scale, shift, mean_out,
                         invstd_out):
        g, b = vec(gamma, Cc), vec(beta, Cc)
        if train:
            # [MRFA_STATS_SLOTS][2C]: the kernels spread their atomics over the slots (mrfa_hip.h); this specification's producers write
            # slot 0 only, the consumer sums all of them
            st = vec(stats, STATS_SLOTS * 2 * Cc, torch.float64).view(STATS_SLOTS, 2 * Cc).sum(0)
            m = st[:Cc] / count
            var = (st[Cc:] / count - m * m).clamp_min(0)
            mean = m.float()
            invstd = (1.0 / torch.sqrt(var + eps)).float()
            if rmean:
                unb = var * count / (count - 1) if count > 1 else var
                rm, rv = vec(rmean, Cc), vec(rvar, Cc)
                rm.mul_(1 - momentum).add_(momentum * mean)
                rv.mul_(1 - momentum).add_(momentum * unb.float())
        else:
            mean = vec(rmean, Cc).clone()
            invstd = 1.0 / torch.sqrt(vec(rvar, Cc) + eps)
        sc = g * invstd
        vec(scale, Cc).copy_(sc)
        vec(shift, Cc).copy_(b - mean * sc)
        if mean_out:
            vec(mean_out, Cc).copy_(mean)
        if invstd_out:
            vec(invstd_out, Cc).copy_(invstd)
        return 0

    def mrfa_bn_finalize_groups(self, stream, stats, count, gamma, beta, rmean, rvar, momentum, eps, Cc, groups, scale, shift, mean_out, invstd_out):
        """v7: `groups` train-mode finalizes, one after the other -- statistics block g -> row g of the outputs; the running statistics take the
        groups' momentum updates in that order (what successive calls of the module do: reference model.py:185-186,234)"""
        for g in range(groups):
            off = 4 * g * Cc
            rc = self.mrfa_bn_finalize(stream, stats + 8 * g * STATS_SLOTS * 2 * Cc, count, gamma, beta, rmean, rvar, momentum, eps, Cc, 1,
                                       scale + off, shift + off, mean_out + off if mean_out else mean_out, invstd_out + off if invstd_out else invstd_out)
            if rc:
                return rc
        return 0

    @staticmethod
    def _per_sample(ptr, p, Cc):
        """(N, 1, 1, C) view of a [groups][C] per-channel array: sample n uses row n // (N / groups)  (v7, statistic groups)"""
        G = _groups(p)
        assert p.N % G == 0
        return vec(ptr, G * Cc).view(G, 1, 1, 1, Cc).expand(G, p.N // G, 1, 1, Cc).reshape(p.N, 1, 1, Cc)

    def mrfa_bn_act_fwd(self, stream, pref):
        p = _obj(pref)
        x = nhwc(p.x, p.N, p.H, p.W, p.ldx, p.C)
        u = x * self._per_sample(p.scale, p, p.C) + self._per_sample(p.shift, p, p.C)
        if getattr(p, "res", None):
            u = u + nhwc(p.res, p.N, p.H, p.W, p.ldr, p.C)
        if p.relu:
            u = F.relu(u)
        Ho, Wo = p.H, p.W
        if p.pool:
            u = F.avg_pool2d(u.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
            Ho, Wo = p.H // 2, p.W // 2
        if p.blend_a:
            o = nhwc(p.occ, p.N, Ho, Wo, p.ldo, 1)
            u = nhwc(p.blend_a, p.N, Ho, Wo, p.lda, p.C) * o + u * (1 - o)
        nhwc(p.y, p.N, Ho, Wo, p.ldy, p.C).copy_(u)
        return 0

    def mrfa_bn_act_bwd(self, stream, pref):
        p = _obj(pref)
        assert p.phase in (1, 2)
        Cc = p.C
        G = _groups(p)               # v7, statistic groups: per-sample rows of the [G][C] arrays, per-group sums and counts
        n = p.N // G
        x = nhwc(p.x, p.N, p.H, p.W, p.ldx, Cc)
        sc, sh = self._per_sample(p.scale, p, Cc), self._per_sample(p.shift, p, Cc)
        u = x * sc + sh
        if getattr(p, "res", None):
            u = u + nhwc(p.res, p.N, p.H, p.W, p.ldr, Cc)
        a = F.relu(u) if p.relu else u
        Ho, Wo = (p.H // 2, p.W // 2) if p.pool else (p.H, p.W)
        dy = nhwc(p.dy, p.N, Ho, Wo, p.lddy, Cc)
        if p.pool:
            da = 0.25 * dy.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
        else:
            da = dy.clone()
        if p.blend_a:
            o = nhwc(p.occ, p.N, Ho, Wo, p.ldo, 1)
            A = nhwc(p.blend_a, p.N, Ho, Wo, p.lda, Cc)
            if p.phase == 1:
                if p.dblend_a:
                    nhwc(p.dblend_a, p.N, Ho, Wo, p.ldda, Cc).add_(da * o)
                if p.docc:
                    nhwc(p.docc, p.N, Ho, Wo, p.lddo, 1).add_((da * (A - a)).sum(-1, keepdim=True))
            da = da * (1 - o)
        du = torch.where(u > 0, da, torch.zeros_like(da)) if p.relu else da
        if p.phase == 1 and getattr(p, "dres", None):
            nhwc(p.dres, p.N, p.H, p.W, p.lddr, Cc).add_(du)
        mean = self._per_sample(p.mean, p, Cc) if p.mean else torch.zeros(Cc)
        invstd = self._per_sample(p.invstd, p, Cc) if p.invstd else torch.zeros(Cc)
        xhat = (x - mean) * invstd
        slots = vec(p.red, G * STATS_SLOTS * 2 * Cc, torch.float64).view(G, STATS_SLOTS, 2 * Cc)   # slotted like the statistics buffers, per group
        if p.phase == 1:
            for g in range(G):
                slots[g, 1, :Cc] += du[g * n:(g + 1) * n].reshape(-1, Cc).double().sum(0)          # any slot: phase 2 sums them all
                slots[g, 1, Cc:] += (du * xhat)[g * n:(g + 1) * n].reshape(-1, Cc).double().sum(0)
            return 0
        red = slots.sum(1)                                                    # [G][2C]
        rows = n * p.H * p.W * max(int(getattr(p, "red_world", 0)), 1)        # v6: `red` summed over that many ranks (SyncBatchNorm)
        if p.train:
            k1 = (red[:, :Cc] / rows).float().view(G, 1, 1, 1, Cc).expand(G, n, 1, 1, Cc).reshape(p.N, 1, 1, Cc)
            k2 = (red[:, Cc:] / rows).float().view(G, 1, 1, 1, Cc).expand(G, n, 1, 1, Cc).reshape(p.N, 1, 1, Cc)
            dx = vec(p.gamma, Cc) * invstd * (du - k1 - xhat * k2)
        else:
            dx = du * sc
        tgt = nhwc(p.dx, p.N, p.H, p.W, p.lddx, Cc)
        if getattr(p, "dx_overwrite", 0):
            tgt.copy_(dx)
        else:
            tgt.add_(dx)
        if p.dbeta:
            vec(p.dbeta, Cc).add_(red[:, :Cc].sum(0).float())
        if p.dgamma:
            vec(p.dgamma, Cc).add_(red[:, Cc:].sum(0).float())
        return 0

    def mrfa_bn_param_grad(self, stream, red, Cc, dgamma, dbeta):
        return self.mrfa_bn_param_grad_groups(stream, red, Cc, 1, dgamma, dbeta)

    def mrfa_bn_param_grad_groups(self, stream, red, Cc, groups, dgamma, dbeta):
        r = vec(red, groups * STATS_SLOTS * 2 * Cc, torch.float64).view(groups * STATS_SLOTS, 2 * Cc).sum(0)
        if dbeta:
            vec(dbeta, Cc).add_(r[:Cc].float())
        if dgamma:
            vec(dgamma, Cc).add_(r[Cc:].float())
        return 0

    # ---------------------------------------------------------------- samplers
    @staticmethod
    def _norm_grid(grid, mode, Hi, Wi):
        """-> (normalised grid, align_corners) exactly as the reference builds it."""
        if mode == 0:
            return grid, False
        n, ho, wo, _ = grid.shape
        ys, xs = torch.meshgrid(torch.arange(ho, dtype=grid.dtype), torch.arange(wo, dtype=grid.dtype), indexing="ij")
        px = grid[..., 0] + xs
        py = grid[..., 1] + ys
        return torch.stack([2 * px / (Wi - 1) - 1, 2 * py / (Hi - 1) - 1], dim=-1), True

    def _gs_inputs(self, inp, ldi, in_bstride, in_rep, Hi, Wi, Cc, N):
        n_in = (N + in_rep - 1) // in_rep
        assert in_bstride == Hi * Wi * ldi
        x = nhwc(inp, n_in, Hi, Wi, ldi, Cc).permute(0, 3, 1, 2)
        return x, n_in

    def mrfa_grid_sample_fwd(self, stream, inp, ldi, in_bstride, in_rep, Hi, Wi, Cc, grid, ldg, N, Ho, Wo, out, ldo, mode):
        x, n_in = self._gs_inputs(inp, ldi, in_bstride, in_rep, Hi, Wi, Cc, N)
        xr = x.repeat_interleave(in_rep, dim=0)[:N]
        g, ac = self._norm_grid(nhwc(grid, N, Ho, Wo, ldg, 2), mode, Hi, Wi)
        y = F.grid_sample(xr, g, mode="bilinear", padding_mode="zeros", align_corners=ac)
        nhwc(out, N, Ho, Wo, ldo, Cc).copy_(y.permute(0, 2, 3, 1))
        return 0

    def mrfa_warp_frame_reflect(self, stream, inp, N, Cc, H, W, grid, Ho, Wo, out):
        x = _flat(inp, N * Cc * H * W).view(N, Cc, H, W)
        g = _flat(grid, N * Ho * Wo * 2).view(N, Ho, Wo, 2)
        _flat(out, N * Cc * Ho * Wo).view(N, Cc, Ho, Wo).copy_(F.grid_sample(x, g, mode="bilinear", padding_mode="reflection", align_corners=False))
        return 0

    def mrfa_grid_sample_bwd(self, stream, inp, ldi, in_bstride, in_rep, Hi, Wi, Cc, grid, ldg, N, Ho, Wo, dout, lddo, mode, din, lddi,
                             din_bstride, dgrid, lddg):
        x, n_in = self._gs_inputs(inp, ldi, in_bstride, in_rep, Hi, Wi, Cc, N)
        gy = nhwc(dout, N, Ho, Wo, lddo, Cc).permute(0, 3, 1, 2)
        with torch.enable_grad():
            xl = x.clone().requires_grad_(True)
            gl = nhwc(grid, N, Ho, Wo, ldg, 2).clone().requires_grad_(True)
            g, ac = self._norm_grid(gl, mode, Hi, Wi)
            y = F.grid_sample(xl.repeat_interleave(in_rep, dim=0)[:N], g, mode="bilinear", padding_mode="zeros", align_corners=ac)
            gx, gg = torch.autograd.grad(y, [xl, gl], gy.contiguous())
        if din:
            nhwc(din, n_in, Hi, Wi, lddi, Cc).add_(gx.permute(0, 2, 3, 1))
        if dgrid:
            nhwc(dgrid, N, Ho, Wo, lddg, 2).add_(gg)
        return 0

    def mrfa_resize_bilinear_fwd(self, stream, inp, ldi, N, Hi, Wi, Cc, out, ldo, Ho, Wo, mul, acc):
        x = nhwc(inp, N, Hi, Wi, ldi, Cc).permute(0, 3, 1, 2)
        y = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=True).permute(0, 2, 3, 1) * mul
        o = nhwc(out, N, Ho, Wo, ldo, Cc)
        o.copy_(o + y if acc else y)
        return 0

    def mrfa_resize_sum_multi(self, stream, descs, n):
        """v8: dst (=|+=) sum_k mul_k resize(src_k), records in table order, terms in record order"""
        for i in range(n):
            d = descs[i]
            o = nhwc(d.dst, d.N, d.Hd, d.Wd, d.ldd, d.C)
            v = None if d.overwrite else o.clone()
            for k in range(d.nterm):
                t = d.term[k]
                x = nhwc(t.src, d.N, t.Hs, t.Ws, t.lds, d.C)
                if (t.Hs, t.Ws) != (d.Hd, d.Wd):
                    x = F.interpolate(x.permute(0, 3, 1, 2), size=(d.Hd, d.Wd), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
                r = x * t.mul
                v = r if v is None else v + r
            o.copy_(v)
        return 0

    def mrfa_resize_sum_multi_bwd(self, stream, descs, n):
        """v8: dst (the gradient of an input, Hd x Wd) += sum_k mul_k adjoint-resize(src_k = the gradient of an output of size Hs x Ws >= Hd x Wd)"""
        for i in range(n):
            d = descs[i]
            o = nhwc(d.dst, d.N, d.Hd, d.Wd, d.ldd, d.C)
            v = o.clone()
            for k in range(d.nterm):
                t = d.term[k]
                assert t.Hs >= d.Hd and t.Ws >= d.Wd, "resize_sum_multi_bwd: up-sampling (or same-size) terms only"
                g = nhwc(t.src, d.N, t.Hs, t.Ws, t.lds, d.C)
                if (t.Hs, t.Ws) != (d.Hd, d.Wd):
                    with torch.enable_grad():
                        xl = torch.zeros(d.N, d.C, d.Hd, d.Wd, requires_grad=True)
                        y = F.interpolate(xl, size=(t.Hs, t.Ws), mode="bilinear", align_corners=True)
                        (gx,) = torch.autograd.grad(y, xl, g.permute(0, 3, 1, 2).contiguous())
                    g = gx.permute(0, 2, 3, 1)
                v = v + g * t.mul
            o.copy_(v)
        return 0

    def mrfa_resize_bilinear_bwd(self, stream, dout, lddo, N, Hi, Wi, Cc, din, lddi, Ho, Wo, mul):
        with torch.enable_grad():
            xl = torch.zeros(N, Cc, Hi, Wi, requires_grad=True)
            y = F.interpolate(xl, size=(Ho, Wo), mode="bilinear", align_corners=True) * mul
            (gx,) = torch.autograd.grad(y, xl, nhwc(dout, N, Ho, Wo, lddo, Cc).permute(0, 3, 1, 2).contiguous())
        nhwc(din, N, Hi, Wi, lddi, Cc).add_(gx.permute(0, 2, 3, 1))
        return 0

    @staticmethod
    def _lookup(v0, v1, coords, radius):
        Q = coords.shape[0]
        d = torch.linspace(-radius, radius, 2 * radius + 1)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * radius + 1, 2 * radius + 1, 2)
        outs = []
        for lvl, vol in enumerate((v0, v1)):
            hh, ww = vol.shape[-2:]
            c = coords.view(Q, 1, 1, 2) / (2 ** lvl) + delta
            g = torch.stack([2 * c[..., 0] / (ww - 1) - 1, 2 * c[..., 1] / (hh - 1) - 1], dim=-1)
            outs.append(F.grid_sample(vol, g, align_corners=True).view(Q, -1))
        return torch.cat(outs, dim=1)

    def mrfa_corr_lookup_fwd(self, stream, vol0, vol1, Hs, Ws, coords, ldc, Q, radius, out, ldo):
        v0 = _flat(vol0, Q * Hs * Ws).view(Q, 1, Hs, Ws)
        v1 = _flat(vol1, Q * (Hs // 2) * (Ws // 2)).view(Q, 1, Hs // 2, Ws // 2)
        r = self._lookup(v0, v1, mat(coords, Q, ldc, 2), radius)
        mat(out, Q, ldo, r.shape[1]).copy_(r)
        return 0

    def mrfa_corr_lookup_bwd(self, stream, vol0, vol1, Hs, Ws, coords, ldc, Q, radius, dout, lddo, dvol0, dvol1, dcoords, lddc):
        with torch.enable_grad():
            v0 = _flat(vol0, Q * Hs * Ws).view(Q, 1, Hs, Ws).clone().requires_grad_(True)
            v1 = _flat(vol1, Q * (Hs // 2) * (Ws // 2)).view(Q, 1, Hs // 2, Ws // 2).clone().requires_grad_(True)
            cl = mat(coords, Q, ldc, 2).clone().requires_grad_(True)
            r = self._lookup(v0, v1, cl, radius)
            g0, g1, gc = torch.autograd.grad(r, [v0, v1, cl], mat(dout, Q, lddo, r.shape[1]).contiguous())
        if dvol0:
            _flat(dvol0, Q * Hs * Ws).add_(g0.reshape(-1))
            _flat(dvol1, Q * (Hs // 2) * (Ws // 2)).add_(g1.reshape(-1))
        if dcoords:
            mat(dcoords, Q, lddc, 2).add_(gc)
        return 0

    # ---------------------------------------------------------------- layout / elementwise
    def mrfa_nchw_to_nhwc(self, stream, src, dst, ldd, N, Cc, H, W, acc):
        s = _flat(src, N * Cc * H * W).view(N, Cc, H, W).permute(0, 2, 3, 1)
        d = nhwc(dst, N, H, W, ldd, Cc)
        d.copy_(d + s if acc else s)
        return 0

    def mrfa_nhwc_to_nchw(self, stream, src, lds, dst, N, Cc, H, W, acc):
        s = nhwc(src, N, H, W, lds, Cc).permute(0, 3, 1, 2)
        d = _flat(dst, N * Cc * H * W).view(N, Cc, H, W)
        d.copy_(d + s if acc else s)
        return 0

    def mrfa_copy_view(self, stream, x, ldx, rows, Cc, y, ldy, mul, acc):
        s = mat(x, rows, ldx, Cc) * mul
        d = mat(y, rows, ldy, Cc)
        d.copy_(d + s if acc else s)
        return 0

    def mrfa_timestamp(self, stream, dst):
        import time
        C.c_ulonglong.from_address(dst).value = time.perf_counter_ns() // 10          # the device clock's 100 MHz
        return 0

    def mrfa_avgpool2_fwd(self, stream, x, ldx, N, H, W, Cc, y, ldy):
        v = F.avg_pool2d(nhwc(x, N, H, W, ldx, Cc).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        nhwc(y, N, H // 2, W // 2, ldy, Cc).copy_(v)
        return 0

    def mrfa_sumpool2_acc(self, stream, x, ldx, N, Ho, Wo, Cc, y, ldy, mul):
        v = 4 * F.avg_pool2d(nhwc(x, N, 2 * Ho, 2 * Wo, ldx, Cc).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        nhwc(y, N, Ho, Wo, ldy, Cc).add_(mul * v)
        return 0

    def mrfa_unpool2_acc(self, stream, dy, lddy, N, Ho, Wo, Cc, dx, lddx, mul):
        g = nhwc(dy, N, Ho, Wo, lddy, Cc).repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
        nhwc(dx, N, 2 * Ho, 2 * Wo, lddx, Cc).add_(mul * g)
        return 0

    def mrfa_bias_act(self, stream, x, ldx, rows, Cc, bias, act, y, ldy, stats):
        v = mat(x, rows, ldx, Cc).clone()
        if bias:
            v = v + vec(bias, Cc)
        if act == 1:
            v = F.relu(v)
        elif act == 2:
            v = torch.sigmoid(v)
        mat(y, rows, ldy, Cc).copy_(v)
        if stats:
            self.mrfa_bn_stats(stream, y, ldy, rows, Cc, stats)
        return 0

    def mrfa_act_bwd(self, stream, y, ldy, dy, lddy, rows, Cc, act, dx, lddx, acc):
        yy = mat(y, rows, ldy, Cc)
        g = mat(dy, rows, lddy, Cc).clone()
        if act == 1:
            g = torch.where(yy > 0, g, torch.zeros_like(g))
        elif act == 2:
            g = g * yy * (1 - yy)
        d = mat(dx, rows, lddx, Cc)
        d.copy_(d + g if acc else g)
        return 0

    def mrfa_blend_fwd(self, stream, a, lda, b, ldb, occ, ldo, rows, Cc, y, ldy):
        o = mat(occ, rows, ldo, 1)
        v = mat(a, rows, lda, Cc) * o
        if b:
            v = v + mat(b, rows, ldb, Cc) * (1 - o)
        mat(y, rows, ldy, Cc).copy_(v)
        return 0

    def mrfa_blend_bwd(self, stream, a, lda, b, ldb, occ, ldo, dy, lddy, rows, Cc, da, ldda, db, lddb, docc, lddo):
        o = mat(occ, rows, ldo, 1)
        g = mat(dy, rows, lddy, Cc)
        av = mat(a, rows, lda, Cc)
        bv = mat(b, rows, ldb, Cc) if b else torch.zeros_like(av)
        if da:
            mat(da, rows, ldda, Cc).add_(g * o)
        if db:
            mat(db, rows, lddb, Cc).add_(g * (1 - o))
        if docc:
            mat(docc, rows, lddo, 1).add_((g * (av - bv)).sum(1, keepdim=True))
        return 0

    def mrfa_colsum(self, stream, x, ldx, rows, Cc, out):
        vec(out, Cc).add_(mat(x, rows, ldx, Cc).sum(0))
        return 0

    def mrfa_antialias_down(self, stream, x, N, Cc, H, W, kern, k, stride, y, ldy):
        img = _flat(x, N * Cc * H * W).view(N, Cc, H, W)
        ker = _flat(kern, k * k).view(1, 1, k, k).expand(Cc, 1, k, k)
        ka = k // 2
        v = F.conv2d(F.pad(img, (ka, ka, ka, ka)), ker, groups=Cc)[:, :, ::stride, ::stride]
        nhwc(y, N, H // stride, W // stride, ldy, Cc).copy_(v.permute(0, 2, 3, 1))
        return 0

    # ---------------------------------------------------------------- K20: flat clip + Adam
    def mrfa_adam_prepare(self, stream, state, ngroups, beta1, beta2):
        st = _flat(state, 8 * ngroups).view(ngroups, 8)
        for g in range(ngroups):
            t = float(st[g, 0]) + 1.0
            st[g, 0] = t
            st[g, 1] = float(st[g, 3]) / (1.0 - beta1 ** t)
            st[g, 2] = 1.0 / (1.0 - beta2 ** t) ** 0.5
            st[g, 4:8] = 0.0
        return 0

    def mrfa_grad_absmax(self, stream, g, n, state, clip_slot):
        if n:
            st = _flat(state, 8)
            st[4 + clip_slot] = max(float(st[4 + clip_slot]), float(_flat(g, n).abs().max()))
        return 0

    def mrfa_adam_flat(self, stream, w, g, m, v, n, state, beta1, beta2, eps, gscale, clip_slot, max_norm):
        if not n:
            return 0
        st = _flat(state, 8)
        coef = gscale
        if clip_slot >= 0:
            coef = coef * min(1.0, max_norm / (float(st[4 + clip_slot]) * gscale + 1e-6))
        W, G, M, V = _flat(w, n), _flat(g, n) * coef, _flat(m, n), _flat(v, n)
        M.lerp_(G, 1.0 - beta1)
        V.mul_(beta2).addcmul_(G, G, value=1.0 - beta2)
        W.addcdiv_(M, V.sqrt() * float(st[2]) + eps, value=-float(st[1]))
        return 0

    # ---------------------------------------------------------------- batched (un)packing
    @staticmethod
    def _chunk_major(flat, taps, rows, cols):
        """[taps][rows][cols] row-major -> the k16-chunk-major order of the bf16 planes: [taps][cols / 16][rows][16]"""
        return flat.view(taps, rows, cols // 16, 16).permute(0, 2, 1, 3).reshape(-1)

    def mrfa_pack_conv_weights_multi(self, stream, descs, n):
        for i in range(n):
            d = descs[i]
            for k in range(d.ndst):
                if d.mode[k] in (8, 9):              # three bf16 pieces of the mode 0 / 2 layout (exact split by chopping)
                    T = d.R * d.S
                    if d.mode[k] == 8:
                        n = T * ((d.Cout + 127) // 128 * 128) * ((d.Cin + 31) // 32 * 32)
                    else:
                        n = T * ((d.Cin + 127) // 128 * 128) * ((d.Cout + 31) // 32 * 32)
                    tmp = torch.zeros(n, dtype=torch.float32)
                    rc = self.mrfa_pack_conv_weight(stream, d.src, tmp.data_ptr(), d.Cout, d.Cin, d.R, d.S, d.mode[k] - 8 if d.mode[k] == 8 else 2)
                    if rc:
                        return rc
                    out = torch.frombuffer((C.c_short * (3 * n)).from_address(d.dst[k]), dtype=torch.int16).view(3, n)
                    rows_, cols_ = ((d.Cout + 127) // 128 * 128, (d.Cin + 31) // 32 * 32) if d.mode[k] == 8 else \
                        ((d.Cin + 127) // 128 * 128, (d.Cout + 31) // 32 * 32)
                    r = self._chunk_major(tmp, T, rows_, cols_)
                    for pc in range(3):
                        bits = r.view(torch.int32) & -65536
                        out[pc] = (bits >> 16).to(torch.int16)
                        r = r - bits.view(torch.float32)
                    continue
                if d.mode[k] in (14, 15):            # one bf16 plane, round-to-nearest-even, in the layout of mode 8 / 9
                    T = d.R * d.S
                    if d.mode[k] == 14:
                        n = T * ((d.Cout + 127) // 128 * 128) * ((d.Cin + 31) // 32 * 32)
                    else:
                        n = T * ((d.Cin + 127) // 128 * 128) * ((d.Cout + 31) // 32 * 32)
                    tmp = torch.zeros(n, dtype=torch.float32)
                    rc = self.mrfa_pack_conv_weight(stream, d.src, tmp.data_ptr(), d.Cout, d.Cin, d.R, d.S, 0 if d.mode[k] == 14 else 2)
                    if rc:
                        return rc
                    out = torch.frombuffer((C.c_short * n).from_address(d.dst[k]), dtype=torch.int16)
                    rows_, cols_ = ((d.Cout + 127) // 128 * 128, (d.Cin + 31) // 32 * 32) if d.mode[k] == 14 else \
                        ((d.Cin + 127) // 128 * 128, (d.Cout + 31) // 32 * 32)
                    out.copy_(self._chunk_major(tmp, T, rows_, cols_).to(torch.bfloat16).view(torch.int16))
                    continue
                if d.mode[k] in (12, 13):            # phase weights of nearest-x2 + 3x3 (see mrfa_conv_params.w_phase), split into three bf16 pieces
                    assert d.R == 3 and d.S == 3     # 13: transposed (rows = input channels) for the phase data gradient
                    tr = d.mode[k] == 13
                    cop, cip = ((d.Cin + 127) // 128 * 128, (d.Cout + 31) // 32 * 32) if tr else ((d.Cout + 127) // 128 * 128, (d.Cin + 31) // 32 * 32)
                    w = _flat(d.src, d.Cout * d.Cin * 9).view(d.Cout, d.Cin, 3, 3)
                    sets = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}        # (phase bit, tap bit) -> 3x3 rows / columns summed
                    full = torch.zeros(16, cop, cip, dtype=torch.float32)
                    for py in range(2):
                        for px in range(2):
                            for a_ in range(2):
                                for b_ in range(2):
                                    acc = torch.zeros(d.Cout, d.Cin, dtype=torch.float32)
                                    for r in sets[(py, a_)]:          # same summation order as the kernel: r outer, s inner, fp32
                                        for s_ in sets[(px, b_)]:
                                            acc = acc + w[:, :, r, s_]
                                    if tr:
                                        full[(py * 2 + px) * 4 + a_ * 2 + b_, :d.Cin, :d.Cout] = acc.t()
                                    else:
                                        full[(py * 2 + px) * 4 + a_ * 2 + b_, :d.Cout, :d.Cin] = acc
                    nn_ = 16 * cop * cip
                    out = torch.frombuffer((C.c_short * (3 * nn_)).from_address(d.dst[k]), dtype=torch.int16).view(3, nn_)
                    r_ = self._chunk_major(full.reshape(-1), 16, cop, cip)
                    for pc in range(3):
                        bits = r_.view(torch.int32) & -65536
                        out[pc] = (bits >> 16).to(torch.int16)
                        r_ = r_ - bits.view(torch.float32)
                    continue
                rc = self.mrfa_pack_conv_weight(stream, d.src, d.dst[k], d.Cout, d.Cin, d.R, d.S, d.mode[k])
                if rc:
                    return rc
        return 0

    def mrfa_conv2d_wgrad_multi(self, stream, ps, n):
        """v6: n weight gradients in as few launches as possible == n calls of mrfa_conv2d_wgrad_nhwc"""
        for i in range(n):
            rc = self.mrfa_conv2d_wgrad_nhwc(stream, ps[i])
            if rc:
                return rc
        return 0

    def mrfa_unpack_wgrads_multi(self, stream, descs, n):
        for i in range(n):
            d = descs[i]
            R = int(round(d.T ** 0.5))
            assert R * R == d.T
            rc = self.mrfa_pack_conv_weight(stream, d.src, d.dst, d.Cout, d.Cin, R, R, 6 if d.fewout else 4)
            if rc:
                return rc
        return 0

