"""Streaming animation of ONE source by many driving frames (the reference's make_animation loop, demo.py:47-73 /
animate_ddp.py:88-105) with everything that depends only on the source computed once: KPDetector(source), the 1/4-scale
source, the generator's feature pyramid and the source structure keys (~38 GF of 375 per frame and
one encoder pass).  Optionally the per-frame program is replayed as a hipGraph."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn


class Animator:
    def __init__(self, model: nn.Module, graph: bool = False):
        """model: mrfa_amd.train.HotPath or mrfa_amd.modules.model.MRFA (attributes encoder / dense_motion / decoder / down)"""
        self.m = model.eval()
        self.use_graph = graph
        self.source = None
        self._g: Optional[torch.cuda.CUDAGraph] = None

    @torch.no_grad()
    def set_source(self, source: torch.Tensor):
        m = self.m
        self.source = source
        self.kp_s = m.encoder(source)
        self.img_down = m.down(source)
        self.cache = m.decoder.encode_source(self.kp_s["kp"], self.img_down, source)
        self._g = None

    @torch.no_grad()
    def _frame(self, driving):
        m = self.m
        kp_d = m.encoder(driving)
        dm = m.dense_motion(self.source, kp_d, self.kp_s)
        out, _, _ = m.decoder(self.kp_s["kp"], kp_d["kp"], dm, img=self.img_down, img_full=self.source, source_cache=self.cache)
        return out

    @torch.no_grad()
    def __call__(self, driving: torch.Tensor) -> torch.Tensor:
        assert self.source is not None, "call set_source(source) first"
        if not self.use_graph:
            return self._frame(driving)
        if self._g is None:                                   # capture the per-frame program once per source
            self._drv = driving.clone()
            self._frame(self._drv)                            # packs / tables outside the graph
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._out = self._frame(self._drv)
            self._g = g
        self._drv.copy_(driving)
        self._g.replay()
        return self._out
