"""Token transformer of the MTIA prior on the HIP engine.  reference: modules/transformer/tokenpose_base.py
(Residual / PreNorm 14-38, FeedForward 46-58, Attention 60-94, Transformer 137-158, TokenPose_TB_base 230-468).

Parameter containers mirror the reference's nesting (transformer.layers.<i>.<0|1>.fn.norm / .fn.fn.to_qkv / .fn.fn.to_out.0 /
.fn.fn.net.<0|3>, patch_to_embedding, keypoint_token, pos_embedding, mlp_head, mlp_head_jacobian) so its state_dict loads.
Computation per layer on (B*tokens) x dim row matrices:
  LayerNorm (K21) -> to_qkv as a 1x1 convolution (K1) -> fused attention (K21, softmax never leaves the CU) -> to_out
  convolution with the residual added in its epilogue -> LayerNorm -> fc1 -> GELU (K21) -> fc2 + residual.
Token assembly (patches, keypoint tokens, positional embedding) is a tiny torch island; the two heads run LayerNorm / Linear as engine
kernels on the gathered keypoint-token rows and leave only slicing + 2*sigmoid-1 to torch.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from ...engine import Ctx, View, run_program

MIN_NUM_PATCHES = 16
BN_MOMENTUM = 0.1


def trunc_normal_(t, std=.02):
    """timm.models.layers.weight_init.trunc_normal_ (tokenpose_base.py:5) is torch's since 1.8"""
    return nn.init.trunc_normal_(t, std=std, a=-2.0, b=2.0)


class Residual(nn.Module):
    def __init__(self, fn, num_keypoints=10):
        super().__init__()
        self.fn = fn
        self.num_keypoints = num_keypoints


class PreNorm(nn.Module):
    def __init__(self, dim, fn, fusion_factor=1):
        super().__init__()
        self.norm = nn.LayerNorm(dim * fusion_factor)
        self.fn = fn


class FeedForward(nn.Module):
    """Linear -> GELU -> Linear (dropout 0 in every reference config).  reference: tokenpose_base.py:46-58"""

    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        if dropout != 0.:
            raise NotImplementedError("dropout is 0 in the reference configs (vox1.yaml / celebvhq.yaml)")
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden_dim, dim), nn.Dropout(dropout))


class Attention(nn.Module):
    """multi-head self-attention, `b n (h d)` head layout.  reference: tokenpose_base.py:60-94"""

    def __init__(self, dim, heads=8, dropout=0., num_keypoints=None, scale_with_head=False, fix_img2motion_attention=False, num_img_tokens=None):
        super().__init__()
        if dropout != 0. or fix_img2motion_attention:
            raise NotImplementedError("dropout / FIX_IMG2MOTION_ATTENTION are off in the reference configs")
        self.heads = heads
        self.scale = (dim // heads) ** -0.5 if scale_with_head else dim ** -0.5
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim, dim), nn.Dropout(dropout))
        self.num_keypoints = num_keypoints
        self.fix_img2motion_attention = fix_img2motion_attention


class Transformer(nn.Module):
    """depth x (x += attn(LN(x)); x += ff(LN(x))); with 'sine-full' embeddings the positional code is re-added to the
    image tokens before every layer but the first.  reference: tokenpose_base.py:137-158"""

    def __init__(self, dim, depth, heads, mlp_dim, dropout, num_keypoints=None, all_attn=False, scale_with_head=False,
                 fix_img2motion_attention=False, num_patches=256):
        super().__init__()
        self.all_attn = all_attn
        self.num_keypoints = num_keypoints
        self.fix_img2motion_attention = fix_img2motion_attention
        self.layers = nn.ModuleList([nn.ModuleList([
            Residual(PreNorm(dim, Attention(dim, heads=heads, dropout=dropout, num_keypoints=num_keypoints, scale_with_head=scale_with_head,
                                            fix_img2motion_attention=fix_img2motion_attention, num_img_tokens=num_patches)),
                     num_keypoints=num_keypoints),
            Residual(PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout)))]) for _ in range(depth)])

    def run(self, e: Ctx, x: View, pos_rows: View = None) -> View:
        """x: (B,1,tokens,dim) residual stream (modified in place by the positional re-add); pos_rows: same shape, zero in
        the keypoint-token rows"""
        for idx, (attn, ff) in enumerate(self.layers):
            if idx > 0 and self.all_attn:
                e.add_const(pos_rows, x)                                   # x[:, num_keypoints:] += pos   (:155)
            a = attn.fn.fn
            h = e.layernorm(x, attn.fn.norm)
            qkv = e.conv(h, a.to_qkv)
            o = e.attention(qkv, a.heads, a.scale)
            x = e.conv(o, a.to_out[0], res=x)                              # Residual: fn(x) + x
            f = ff.fn.fn.net
            h = e.layernorm(x, ff.fn.norm)
            g = e.gelu(e.conv(h, f[0]))
            x = e.conv(g, f[3], res=x)
        return x


class TokenPose_TB_base(nn.Module):
    """HRNet feature map -> 4x4 patch tokens + 2*K learned keypoint / Jacobian tokens -> Transformer -> per-token heads:
    kp = 2*sigmoid(Linear(LN(t))) - 1, jacobian = Linear(LN(t)) as 2x2.  reference: tokenpose_base.py:230-468"""

    def __init__(self, *, feature_size, patch_size, num_keypoints, dim, depth, heads, mlp_dim, apply_init=False, apply_multi=True,
                 hidden_heatmap_dim=64 * 6, heatmap_dim=64 * 64, heatmap_size=[64, 64], channels=3, dropout=0., emb_dropout=0.,
                 pos_embedding_type="sine-full", estimate_jacobian=True, temperature=0.1, spatial_kp_head=False, jacobian_token=True,
                 hidden_dim=False, affine_jacobian=False, fix_img2motion_attention=False):
        super().__init__()
        assert isinstance(feature_size, list) and isinstance(patch_size, list), 'image_size and patch_size should be list'
        assert feature_size[0] % patch_size[0] == 0 and feature_size[1] % patch_size[1] == 0, \
            'Image dimensions must be divisible by the patch size.'
        if spatial_kp_head or emb_dropout != 0.:
            raise NotImplementedError("spatial_kp_head / emb_dropout are off in the reference configs")
        h, w = feature_size[0] // patch_size[0], feature_size[1] // patch_size[1]
        num_patches = h * w
        patch_dim = channels * patch_size[0] * patch_size[1]
        self.inplanes = 64
        self.patch_size = patch_size
        self.heatmap_size = heatmap_size
        self.num_patches = num_patches
        self.pos_embedding_type = pos_embedding_type
        self.all_attn = (self.pos_embedding_type == "sine-full")
        self.jacobian_token = jacobian_token
        if jacobian_token:
            num_keypoints = 2 * num_keypoints
        self.num_keypoints = num_keypoints
        self.keypoint_token = nn.Parameter(torch.zeros(1, self.num_keypoints, dim))
        self._make_position_embedding(w, h, dim, pos_embedding_type)
        self.patch_to_embedding = nn.Linear(patch_dim, dim)
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = Transformer(dim, depth, heads, mlp_dim, dropout, num_keypoints=num_keypoints, all_attn=self.all_attn,
                                       scale_with_head=True, fix_img2motion_attention=fix_img2motion_attention, num_patches=num_patches)
        self.to_keypoint_token = nn.Identity()
        self.spatial_kp_head = spatial_kp_head

        def head(n_out):
            if hidden_dim:
                return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim), nn.LayerNorm(dim), nn.Linear(dim, n_out))
            return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, n_out))
        self.mlp_head = head(2)
        trunc_normal_(self.keypoint_token, std=.02)
        if apply_init:
            self.apply(self._init_weights)
        self.mlp_head_jacobian = None
        if estimate_jacobian:
            self.mlp_head_jacobian = head(4)
            for m in list(self.mlp_head_jacobian)[:-1]:
                self._init_weights(m)
            self.mlp_head_jacobian[-1].weight.data.zero_()
            self.mlp_head_jacobian[-1].bias.data.copy_(torch.tensor([1, 0, 0, 1], dtype=torch.float))
        self.affine_jacobian = affine_jacobian

    # -- positional embedding: tokenpose_base.py:317-362
    def _make_position_embedding(self, w, h, d_model, pe_type='sine'):
        assert pe_type in ['none', 'learnable', 'sine', 'sine-full']
        self.pe_h, self.pe_w = h, w
        if pe_type == 'none':
            self.pos_embedding = None
        elif pe_type == 'learnable':
            self.pos_embedding = nn.Parameter(torch.zeros(1, self.num_patches + self.num_keypoints, d_model))
            trunc_normal_(self.pos_embedding, std=.02)
        else:
            self.pos_embedding = nn.Parameter(self._make_sine_position_embedding(d_model), requires_grad=False)

    def _make_sine_position_embedding(self, d_model, temperature=10000, scale=2 * math.pi):
        """2-D sine code (1, h*w, d_model): [sin/cos interleaved over y | same over x].  reference: tokenpose_base.py:340-362"""
        h, w = self.pe_h, self.pe_w
        half, eps = d_model // 2, 1e-6
        ys = torch.arange(1, h + 1, dtype=torch.float32) / (h + eps) * scale
        xs = torch.arange(1, w + 1, dtype=torch.float32) / (w + eps) * scale
        k = torch.arange(half, dtype=torch.float32)
        freq = temperature ** (2 * torch.div(k, 2, rounding_mode='floor') / half)

        def code(v):                                                     # (len,) -> (len, half): sin, cos, sin, cos, ...
            a = v[:, None] / freq
            return torch.stack((a[:, 0::2].sin(), a[:, 1::2].cos()), dim=2).flatten(1)
        py = code(ys)[:, None, :].expand(h, w, half)
        px = code(xs)[None, :, :].expand(h, w, half)
        return torch.cat((py, px), dim=2).reshape(1, h * w, d_model).contiguous()

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # -- computation
    def run(self, e: Ctx, feature: View):
        """feature: (B,H,W,C) NHWC view -> IslandOuts (kp (B,K,2)[, jacobian (B,K,2,2)])"""
        p1, p2 = self.patch_size
        B, H, W, Cc = feature.N, feature.H, feature.W, feature.C
        hh, ww = H // p1, W // p2
        n = hh * ww

        def patchify(f):                                  # 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' on the NHWC tensor   (:408)
            return [f.reshape(B, hh, p1, ww, p2, Cc).permute(0, 1, 3, 2, 4, 5).reshape(B, 1, n, p1 * p2 * Cc)]
        patches = e.island(patchify, [feature])[0]
        emb = e.conv(patches.view(), self.patch_to_embedding)                                   # (B,1,n,dim)
        kind = self.pos_embedding_type
        nk = self.num_keypoints

        def assemble(em, kp_tok, *pos):                   # tokenpose_base.py:411-422
            if kind in ("sine", "sine-full"):
                em = em + pos[0][:, :n].unsqueeze(0)
            x = torch.cat((kp_tok.expand(B, -1, -1).unsqueeze(1), em), dim=2)
            if kind == "learnable":
                x = x + pos[0][:, :n + nk].unsqueeze(0)
            return [x]
        pos_in = [] if self.pos_embedding is None else [self.pos_embedding]
        x = e.island(assemble, [emb, self.keypoint_token] + pos_in)[0]
        pos_rows = None
        if self.all_attn:
            # the constant position code tiled over the batch, zero in the keypoint-token rows; cached (it is a constant and
            # building it inside a hipGraph capture would record a memset node)
            key = (B, str(e.dev), self.pos_embedding.data_ptr(), self.pos_embedding._version)
            if getattr(self, "_pos_rows_key", None) != key:
                pr = torch.zeros((B, 1, nk + n, self.pos_embedding.shape[-1]), dtype=torch.float32, device=e.dev)
                pr[:, 0, nk:] = self.pos_embedding.detach()[0, :n]
                object.__setattr__(self, "_pos_rows", pr)
                object.__setattr__(self, "_pos_rows_key", key)
            pos_rows = e.wrap_nhwc(self._pos_rows)
        xv = self.transformer.run(e, x.view(), pos_rows)
        jac_tok, affine = self.jacobian_token, self.affine_jacobian
        # heads (tokenpose_base.py:424-466): the 2K keypoint / Jacobian token rows are gathered once (a copy), then LayerNorm and the
        # Linear layers run as the engine's own kernels (K21 / the few-output direct convolution) -- not as torch layer_norm + rocBLAS GEMMs
        # on 10-token matrices; what is left to torch is slicing and the closing element-wise maps on (B, K, 2 | 4) values
        (toks,) = e.island(lambda xt: [xt[:, :, :nk]], [xv])

        def apply(hd, t: View) -> View:
            for m in hd:
                t = e.layernorm(t, m) if isinstance(m, nn.LayerNorm) else e.conv(t, m)
            return t
        head_outs = [apply(self.mlp_head, toks.view())]
        if self.mlp_head_jacobian is not None:
            head_outs.append(apply(self.mlp_head_jacobian, toks.view()))

        def finish(o_kp, *o_jac):                          # (B,1,2K,2) [, (B,1,2K,4)]
            o_kp = o_kp[:, 0]
            outs = [2 * torch.sigmoid(o_kp[:, 0:nk // 2] if jac_tok else o_kp[:, 0:nk]) - 1]
            if o_jac:
                jac = o_jac[0][:, 0]
                jac = jac[:, nk // 2:nk] if jac_tok else jac[:, 0:nk]
                if affine:
                    theta = jac[:, :, 0:2]
                    theta = theta / (torch.norm(theta, p=2, dim=-1, keepdim=True) + 1e-10)
                    c, s_ = theta[:, :, 0:1], theta[:, :, 1:2]
                    rot = torch.cat((c, -s_, s_, c), dim=-1).reshape(B, -1, 2, 2)
                    sc = 1 / (torch.tanh(jac[:, :, 2:]) * 0.9 + 1)
                    zero = torch.zeros_like(sc[:, :, 0:1])
                    scm = torch.cat((sc[:, :, 0:1], zero, zero, sc[:, :, 1:2]), dim=-1).reshape(B, -1, 2, 2)
                    jac = torch.matmul(rot, scm)
                else:
                    jac = jac.reshape(B, -1, 2, 2)
                outs.append(jac)
            return outs
        return e.island(finish, head_outs)

    def forward(self, feature, mask=None):
        assert mask is None
        outs = run_program(self, self._program, [feature])
        out = {'kp': outs[0]}
        if self.mlp_head_jacobian is not None:
            out['jacobian'] = outs[1]
        return out

    def _program(self, e: Ctx, feature: torch.Tensor):
        fv = e.from_nchw(feature)
        outs = self.run(e, fv)
        return tuple(o.t for o in outs), tuple(o.add_grad for o in outs), (lambda: e.grad_to_nchw(fv),)
