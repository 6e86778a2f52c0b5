"""Independent PyTorch-CPU restatement of the HF-Net graph (TEST INFRASTRUCTURE).

Second, structurally different implementation of the same reference sources as
oracle/hfnet_oracle.c (hfnet/models/hf_net.py:13-96,184-237; hfnet/models/utils/layers.py:6-109;
hfnet/models/backbones/utils/conv_blocks.py:163-312), written with library ops
(F.conv2d / F.max_pool2d / F.pixel_shuffle / F.grid-free gather) in float64, NCHW.  It exists
only so that tests/test_oracle_vs_torch.py can check the C oracle against something that does
not share its loops; it is never on a product path.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from hfnet_slam_amd.spec import BN_EPS, NetSpec, same_pad

DT = torch.float64


def _t(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.asarray(a)).to(DT)


def _conv_w(w: np.ndarray) -> torch.Tensor:         # HWIO -> OIHW
    return _t(w).permute(3, 2, 0, 1).contiguous()


def _pad_same(x: torch.Tensor, k: int, stride: int, value: float = 0.0) -> torch.Tensor:
    h, w = x.shape[-2:]
    _, pt, pb = same_pad(h, k, stride)
    _, pl, pr = same_pad(w, k, stride)
    return F.pad(x, (pl, pr, pt, pb), value=value)


class TorchHFNet:
    def __init__(self, weights: Dict[str, np.ndarray], spec: NetSpec):
        self.w, self.spec = weights, spec

    def _bn(self, x, scope):
        g, b = _t(self.w[f"{scope}/BatchNorm/gamma"]), _t(self.w[f"{scope}/BatchNorm/beta"])
        m, v = _t(self.w[f"{scope}/BatchNorm/moving_mean"]), _t(self.w[f"{scope}/BatchNorm/moving_variance"])
        return F.batch_norm(x, m, v, g, b, training=False, eps=BN_EPS)

    def _conv_bn(self, x, scope, k, stride, act=True, wname="weights"):
        x = F.conv2d(_pad_same(x, k, stride), _conv_w(self.w[f"{scope}/{wname}"]), stride=stride)
        x = self._bn(x, scope)
        return torch.clamp(x, 0.0, 6.0) if act else x

    def _dw_bn(self, x, scope, stride):
        w = _t(self.w[f"{scope}/depthwise_weights"]).permute(2, 3, 0, 1).contiguous()  # [3,3,C,1] -> [C,1,3,3]
        x = F.conv2d(_pad_same(x, 3, stride), w, stride=stride, groups=x.shape[1])
        return torch.clamp(self._bn(x, scope), 0.0, 6.0)

    def _block(self, x, b):
        y = x
        if b.expand > b.cin:
            y = self._conv_bn(y, f"{b.scope}/expand", 1, 1)
        y = self._dw_bn(y, f"{b.scope}/depthwise", b.stride)
        y = self._conv_bn(y, f"{b.scope}/project", 1, 1, act=False)
        return y + x if b.residual else y

    def backbone(self, img_u8: np.ndarray):
        h, w = img_u8.shape
        hc, wc = h // 8 * 8, w // 8 * 8
        x = (_t(img_u8[:hc, :wc].astype(np.float64)) - 128.0) / 128.0
        x = x[None, None]
        feats = {}
        x = self._conv_bn(x, "MobilenetV2/Conv", 3, 2)
        feats[1] = x
        for b in self.spec.blocks:
            x = self._block(x, b)
            feats[b.index] = x
        return feats

    def tail_from_intermediate(self, inter_hwc: np.ndarray):
        x = _t(inter_hwc).permute(2, 0, 1)[None]
        for b in self.spec.blocks:
            if b.index >= 8:
                x = self._block(x, b)
        return x

    def local_head(self, f7):
        w = self.w
        d = self._conv_bn(f7, "local_head/descriptor/Conv", 3, 1)
        d = F.conv2d(d, _conv_w(w["local_head/descriptor/Conv_1/weights"]), _t(w["local_head/descriptor/Conv_1/biases"]))
        d = d * torch.rsqrt(torch.clamp((d * d).sum(1, keepdim=True), min=1e-12))
        s = self._conv_bn(f7, "local_head/detector/Conv", 3, 1)
        logits = F.conv2d(s, _conv_w(w["local_head/detector/Conv_1/weights"]), _t(w["local_head/detector/Conv_1/biases"]))
        prob = torch.softmax(logits, dim=1)[:, :-1]
        dense = F.pixel_shuffle(prob, 8)[0, 0]
        return d[0].permute(1, 2, 0), logits[0].permute(1, 2, 0), dense

    @staticmethod
    def simple_nms(scores: torch.Tensor, radius=4, iterations=2) -> torch.Tensor:
        size = 2 * radius + 1

        def mp(x):
            return F.max_pool2d(F.pad(x[None, None], (radius,) * 4, value=float("-inf")), size, stride=1)[0, 0]

        zeros = torch.zeros_like(scores)
        max_mask = scores == mp(scores)
        for _ in range(iterations - 1):
            supp_mask = mp(max_mask.to(scores.dtype)) > 0
            supp_scores = torch.where(supp_mask, zeros, scores)
            new_max_mask = supp_scores == mp(supp_scores)
            max_mask = max_mask | (new_max_mask & ~supp_mask)
        return torch.where(max_mask, scores, zeros)

    def global_head(self, f18):
        w = self.w
        feat = f18[0].permute(1, 2, 0).reshape(-1, f18.shape[1])                 # [P, D]
        mem = self._conv_bn(f18, "global_head/vlad/memberships", 1, 1, act=False)
        mem = torch.softmax(mem, dim=1)[0].permute(1, 2, 0).reshape(feat.shape[0], -1)   # [P, K]
        clusters = _t(w["global_head/vlad/clusters"])                              # [K, D]
        desc = (mem.t()[:, :, None] * (clusters[:, None, :] - feat[None, :, :])).sum(1)  # [K, D]
        desc = desc * torch.rsqrt(torch.clamp((desc * desc).sum(0, keepdim=True), min=1e-12))
        v = desc.reshape(-1)
        v = v * torch.rsqrt(torch.clamp((v * v).sum(), min=1e-12))
        vlad = v.clone()
        v = v * torch.rsqrt(torch.clamp((v * v).sum(), min=1e-12))
        g = v @ _t(w["global_head/dimensionality_reduction/weights"]) + _t(w["global_head/dimensionality_reduction/biases"])
        g = g * torch.rsqrt(torch.clamp((g * g).sum(), min=1e-12))
        return g, vlad, mem

    def run(self, img_u8: np.ndarray):
        feats = self.backbone(img_u8)
        desc_map, logits, dense = self.local_head(feats[7])
        g, vlad, mem = self.global_head(feats[18])
        return {"feats": feats, "desc_map": desc_map, "logits": logits, "scores_dense": dense,
                "scores_nms": self.simple_nms(dense), "global": g, "vlad": vlad, "memberships": mem}
