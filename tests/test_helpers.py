"""The reference's small callables under their own names (VERDICT r1 'missing' item 5): bilinear_sampler, batch_bilinear_sampler,
coords_grid (modules/util.py:26-56), CorrBlock (modules/raft.py:12-48), BasicMotionEncoder.forward (raft.py:60-68),
RefineFlow.forward (raft.py:80-88) -- forward values and autograd gradients against tests/golden/helpers.npz, recorded from the
reference's own functions by tools/make_goldens.py:g10_helpers.  CPU: the product code through the ABI emulator; GPU: the kernels."""
import os

import numpy as np
import pytest
import torch

from mrfa_amd.modules import BasicMotionEncoder, CorrBlock, RefineFlow, batch_bilinear_sampler, bilinear_sampler, coords_grid
from mrfa_amd.utils.prng import det_uniform
from tests import cases
from tests.emu import emulated_hip


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "helpers.npz"))


def _cmp(t, ref, tol, what):
    d = np.abs(t.detach().cpu().numpy() - ref).max()
    assert d <= tol, (what, d)


def _check(g, dev):
    u = lambda tag, shp, lo, hi: det_uniform(tag, shp, lo, hi).to(dev)
    # bilinear_sampler (+ mask), gradients w.r.t. the image and the pixel coordinates
    img = u("h/bs_img", (2, 4, 6, 7), -1, 1).requires_grad_(True)
    xy = u("h/bs_xy", (2, 5, 6, 2), -1.5, 7.5).requires_grad_(True)
    y, m = bilinear_sampler(img, xy, mask=True)
    (y * u("h/bs_w", tuple(y.shape), -1, 1)).sum().backward()
    _cmp(y, g["bs_out"], 1e-6, "bilinear_sampler")
    _cmp(m, g["bs_mask"], 0, "bilinear_sampler mask")
    _cmp(img.grad, g["bs_dimg"], 1e-5, "d img")
    _cmp(xy.grad, g["bs_dxy"], 1e-5, "d coords")
    _cmp(batch_bilinear_sampler(u("h/bbs_img", (32, 1, 6, 6), -1, 1), u("h/bbs_xy", (32, 7, 7, 2), -1.0, 6.0), h=4, w=4, mini_batch=1),
         g["bbs_out"], 1e-6, "batch_bilinear_sampler")
    _cmp(coords_grid(2, 3, 5, dev), g["coords_grid"], 0, "coords_grid")
    # CorrBlock: same constructor / call as raft.py:238-240
    maps = u("h/corr_maps", (40, 1, 8, 8), -1, 1).requires_grad_(True)
    cxy = u("h/corr_xy", (2, 2, 4, 5), -2.0, 9.0).requires_grad_(True)
    c = CorrBlock(maps)(cxy)
    (c * u("h/corr_w", tuple(c.shape), -1, 1)).sum().backward()
    _cmp(c, g["cb_out"], 1e-6, "CorrBlock")
    _cmp(maps.grad, g["cb_dmaps"], 1e-5, "d corr")
    _cmp(cxy.grad, g["cb_dxy"], 2e-5, "d coords (CorrBlock)")
    # BasicMotionEncoder.forward / RefineFlow.forward as stand-alone modules
    enc = BasicMotionEncoder()
    enc.load_state_dict(cases.weights_for(enc.state_dict(), "h/enc"))
    enc.to(dev)
    flow = u("h/enc_flow", (2, 2, 8, 8), -3, 3).requires_grad_(True)
    corr = u("h/enc_corr", (2, 98, 8, 8), -1, 1).requires_grad_(True)
    o = enc(flow, corr)
    (o * u("h/enc_w", tuple(o.shape), -1, 1)).sum().backward()
    _cmp(o, g["enc_out"], 2e-5, "BasicMotionEncoder")
    _cmp(flow.grad, g["enc_dflow"], 1e-4, "d flow")
    _cmp(corr.grad, g["enc_dcorr"], 1e-4, "d corr (encoder)")
    norms = np.array([p.grad.norm().item() for _, p in enc.named_parameters()], np.float32)
    assert np.abs(norms - g["enc_pgrad_norms"]).max() <= 1e-3 * g["enc_pgrad_norms"].max(), norms
    ref = RefineFlow()
    ref.load_state_dict(cases.weights_for(ref.state_dict(), "h/ref"))
    ref.to(dev)
    mf = u("h/ref_mf", (2, 128, 8, 8), -1, 1).requires_grad_(True)
    wf = u("h/ref_wf", (2, 192, 8, 8), -1, 1).requires_grad_(True)
    d, inp = ref(mf, wf)
    ((d * u("h/ref_w", tuple(d.shape), -1, 1)).sum() + (inp * u("h/ref_wi", tuple(inp.shape), -0.1, 0.1)).sum()).backward()
    _cmp(d, g["ref_out"], 2e-5, "RefineFlow out")
    _cmp(inp, g["ref_inp"], 2e-5, "RefineFlow inp")
    _cmp(mf.grad, g["ref_dmf"], 1e-4, "d m_f")
    _cmp(wf.grad, g["ref_dwf"], 1e-4, "d warp_f")
    norms = np.array([p.grad.norm().item() for _, p in ref.named_parameters()], np.float32)
    assert np.abs(norms - g["ref_pgrad_norms"]).max() <= 1e-3 * g["ref_pgrad_norms"].max(), norms


def test_helpers_through_abi_emulator(golden_dir):
    with emulated_hip():
        _check(_g(golden_dir), "cpu")


@pytest.mark.gpu
def test_helpers_gpu(golden_dir):
    _check(_g(golden_dir), "cuda:0")
