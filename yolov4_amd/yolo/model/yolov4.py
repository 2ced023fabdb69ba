# -*- coding: utf-8 -*-
"""YOLOv4 detector graph (CSPDarknet53 + SPP + FPN/PAN + 3 heads) behind the
reference's module API (yolo/model/yolov4.py:26-324); same attribute tree, hence
the same 648 state_dict keys."""
import os
from collections import OrderedDict
from typing import Dict

import torch
from torch import nn

from ... import ops
from ...darknet.darknet import ConvBNAct, CSPDownSample0, CSPDownSample, chain, geo_of, observed, plan_for, soft, takes_planes
from .yololayer import YOLOLayer

L = 'leaky_relu'


def _five(big, small):
    return nn.Sequential(ConvBNAct(big, small, 1, 1, act=L), ConvBNAct(small, big, 3, 1, act=L),
                         ConvBNAct(big, small, 1, 1, act=L), ConvBNAct(small, big, 3, 1, act=L),
                         ConvBNAct(big, small, 1, 1, act=L))


class Backbone(nn.Module):

    def __init__(self):
        super().__init__()
        self.stem = ConvBNAct(3, 32, 3, 1, act='mish')
        self.stage1 = CSPDownSample0(32, 64, 3, 2, act='mish')
        self.stage2 = CSPDownSample(64, 128, 3, 2, num_blocks=2, act='mish')
        self.stage3 = CSPDownSample(128, 256, 3, 2, num_blocks=8, act='mish')
        self.stage4 = CSPDownSample(256, 512, 3, 2, num_blocks=8, act='mish')
        self.stage5 = CSPDownSample(512, 1024, 3, 2, num_blocks=4, act='mish')

    def forward(self, x, readers=((), (), ())):
        """readers: the ConvBNAct modules outside the backbone that read x3 / x4 / x5 (YOLOv4.forward names the neck's), so
        that a stage's transition conv can write its result pre-split where its readers take it so (the reference's
        forward has no such argument; hooks on this module or on a stage keep everything fp32: darknet.observed)."""
        quiet = not observed(self)
        x = self.stage1(self.stem(x), readers=[self.stage2.base] if quiet and not observed(self.stage2) else None)
        x = self.stage2(x, readers=[self.stage3.base] if quiet and not observed(self.stage3) else None)
        x3 = self.stage3(x, readers=[self.stage4.base] + list(readers[0]))
        x3a, x3b = ops.fork(x3)
        x4 = self.stage4(x3a, readers=[self.stage5.base] + list(readers[1]))
        x4a, x4b = ops.fork(x4)
        x5 = self.stage5(x4a, readers=list(readers[2]) if quiet else None)
        return x3b, x4b, x5


class SPPBlock(nn.Module):

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Sequential(ConvBNAct(1024, 512, 1, 1, act=L), ConvBNAct(512, 1024, 3, 1, act=L),
                                   ConvBNAct(1024, 512, 1, 1, act=L))
        # kept for module-tree parity; pooling runs in the fused SPP kernel sequence (5, 9, 5: the
        # reference never uses max_pool3, yolov4.py:68-70)
        self.max_pool1 = nn.MaxPool2d(5, 1, 5 // 2)
        self.max_pool2 = nn.MaxPool2d(9, 1, 9 // 2)
        self.max_pool3 = nn.MaxPool2d(13, 1, 13 // 2)
        self.conv2 = ConvBNAct(2048, 512, 1, 1, act=L)

    def forward(self, x, out_planes=False):
        """out_planes: the sole consumer (fpn.module1[0]) takes a pre-split input; the reference has no such argument."""
        y = chain(self.conv1, x)
        from ...darknet import darknet as D
        # the pooled concat has one reader, conv2: where that takes planes and nobody is looking it is split once, pre-split
        want = bool(D._CAT_PLANES and takes_planes(self.conv2, geo=geo_of(y)) and not observed(self, self.conv1, self.conv2))
        return self.conv2(ops.spp_pool_cat(y, planes=want), out_planes=soft(out_planes, self))


class Upsample(nn.Module):

    def forward(self, x, target_size, out=None):
        """out: optional destination (a CatBuffer slot); the reference has no such argument."""
        assert x.dim() == 4
        slot = ops.Slot(out) if out is not None else None
        Ht, Wt = int(target_size[2]), int(target_size[3])
        if Ht == 2 * x.shape[2] and Wt == 2 * x.shape[3]:
            if getattr(x, 'y4_planes', False):       # pre-split, into its slot of a pre-split concat buffer (FPNBlock)
                return ops.as_planes(ops.Upsample2xFn.apply(x, slot), getattr(x, 'y4_amax', None))
            return self._tag(ops.Upsample2xFn.apply(x, slot), x, out)      # the YOLOv4 neck at S % 32 == 0
        # any other target (e.g. S = 600: 19 -> 38 -> 75): train = F.interpolate(size=target, nearest) (yolov4.py:85);
        # eval = integer-factor expand whose final view() needs target % input == 0 (yolov4.py:87-90)
        if not self.training and (Ht % x.shape[2] or Wt % x.shape[3]):
            raise RuntimeError(f"shape '[{x.shape[0]}, {x.shape[1]}, {Ht}, {Wt}]' is invalid for input of size "
                               f"{x.shape[0] * x.shape[1] * (Ht // x.shape[2]) * x.shape[2] * (Wt // x.shape[3]) * x.shape[3]}")
        return self._tag(ops.UpsampleNearestFn.apply(x, Ht, Wt, not self.training, slot), x, out)

    @staticmethod
    def _tag(y, x, out):
        """conv mode 3: max|upsampled| = max|x|; written into a concat slot it folds into that buffer's shared cell"""
        cell, shared = ops.amax_of(x), ops.amax_of(out)
        if shared is not None and y.data_ptr() == out.data_ptr():
            if cell is None:
                cell = ops.amax_raw(x)
            ops.amax_merge(shared, cell)
            return ops.tag_amax(y, shared)
        return ops.tag_amax(y, cell)


class FPNBlock(nn.Module):

    def __init__(self):
        super().__init__()
        self.module1 = nn.Sequential(ConvBNAct(512, 1024, 3, 1, act=L), ConvBNAct(1024, 512, 1, 1, act=L))
        self.conv3 = ConvBNAct(512, 256, 1, 1)
        self.upsample1 = Upsample()
        self.conv4 = ConvBNAct(512, 256, 1, 1, act=L)
        self.module2 = _five(512, 256)
        self.conv10 = ConvBNAct(256, 128, 1, 1)
        self.upsample2 = Upsample()
        self.conv11 = ConvBNAct(256, 128, 1, 1, act=L)
        self.module3 = _five(256, 128)

    def forward(self, x3, x4, x5, head_planes=False):
        """head_planes: head.yolo1[0] (one of the two consumers of f1) takes a pre-split input."""
        f3 = chain(self.module1, x5)
        f3a, f3b = ops.fork(f3)
        pn = self._cat_planes(self.conv4, self.conv3, self.upsample1, self.module2, x4, f3a)
        cb = ops.cat_buffer(x4, [256, 256], planes_norms=pn)     # [conv4(x4) | upsampled conv3(f3)], written in place
        up = self.upsample1(self.conv3(f3a, out_planes=bool(pn), scale_from=cb if pn else None), x4.size(), out=cb.slot(1))
        x4 = self.conv4(x4, out=cb.slot(0))
        assert up.shape[2:] == x4.shape[2:]
        f2 = chain(self.module2, ops.cat([x4, up], into=cb))
        f2a, f2b = ops.fork(f2)
        pn = self._cat_planes(self.conv11, self.conv10, self.upsample2, self.module3, x3, f2a)
        cb = ops.cat_buffer(x3, [128, 128], planes_norms=pn)
        up = self.upsample2(self.conv10(f2a, out_planes=bool(pn), scale_from=cb if pn else None), x3.size(), out=cb.slot(1))
        x3 = self.conv11(x3, out=cb.slot(0))
        assert up.shape[2:] == x3.shape[2:]
        f1 = chain(self.module3, ops.cat([x3, up], into=cb), last='both' if head_planes else False)
        return f1, f2b, f3b

    def _cat_planes(self, lateral, reduce, up, five, x, f):
        """planes_norms for the concat [lateral(x) | up(reduce(f))] in front of the block `five`, or None: where its first conv
        takes planes, the upsample is the exact x2 one and nobody is looking, both producers write the concat buffer pre-split
        under one joint scale (the reducing conv's BatchNorm ran over the SMALL map: its bound uses that pixel count) and the
        concat-fed 1x1 conv runs on the DMA kernels (darknet._CAT_PLANES, ops.CatBuffer)."""
        from ...darknet import darknet as D
        if not (D._CAT_PLANES and x.shape[2] == 2 * f.shape[2] and x.shape[3] == 2 * f.shape[3]
                and takes_planes(five[0], geo=geo_of(x)) and lateral.training and reduce.training
                and not observed(self, lateral, reduce, up, five, five[0])):
            return None
        return (lateral.norm, (reduce.norm, int(f.shape[0]) * int(f.shape[2]) * int(f.shape[3])))


class PANBlock(nn.Module):

    def __init__(self):
        super().__init__()
        self.conv1 = ConvBNAct(128, 256, 3, 2, act=L)
        self.module1 = _five(512, 256)
        self.conv7 = ConvBNAct(256, 512, 3, 2, act=L)
        self.module2 = _five(1024, 512)

    def forward(self, f1, f2, f3, head_planes=(False, False)):
        """head_planes: head.yolo2[0] / head.yolo3[0] take a pre-split input (p2 has a second, fp32 consumer; p3 none)."""
        p1, f1b = ops.fork(f1)
        cb = ops.cat_buffer(f2, [256, f2.shape[1]], planes_norms=self._cat_planes(self.conv1, self.module1, f2))
        p2 = self.conv1(f1b, out=cb.slot(0))
        assert p2.shape[2:] == f2.shape[2:]
        p2 = chain(self.module1, ops.cat([p2, f2], into=cb),
                   last='both' if (head_planes[0] or takes_planes(self.conv7, p2.shape[2:], geo_of(p2))) else False)
        p2a, p2b = ops.fork(p2)
        cb = ops.cat_buffer(f3, [512, f3.shape[1]], planes_norms=self._cat_planes(self.conv7, self.module2, f3))
        p3 = self.conv7(p2a, out=cb.slot(0))
        assert p3.shape[2:] == f3.shape[2:]
        p3 = chain(self.module2, ops.cat([p3, f3], into=cb), last=soft(head_planes[1], self))
        return p1, p2b, p3


    def _cat_planes(self, down, five, f):
        """planes_norms for the concat [down(.) | f] in front of the block `five`, or None: where its first conv takes planes and
        nobody is looking, the stride-2 conv writes its slot of the concat buffer pre-split and cat() SPLITS the lateral
        tensor f into the other slot (instead of copying it), both under one joint scale (darknet._CAT_PLANES, ops.CatBuffer)."""
        from ...darknet import darknet as D
        pm = ops.planes_mode()
        if not (D._CAT_PLANES and pm is not None and takes_planes(five[0], geo=geo_of(f)) and down.training and down.has_bn
                and not getattr(f, 'y4_planes', False) and f.shape[1] % (64 if pm == 'bf16' else 32) == 0
                and not observed(self, down, five, five[0])):
            return None
        return (down.norm, f)


class Neck(nn.Module):

    def __init__(self):
        super().__init__()
        self.spp = SPPBlock()
        self.fpn = FPNBlock()
        self.pan = PANBlock()

    def forward(self, x3, x4, x5, head_planes=(False, False, False)):
        """head_planes: which of the three head 3x3 convs take pre-split inputs (YOLOv4.forward asks them)."""
        x5 = self.spp(x5, out_planes=soft(takes_planes(self.fpn.module1[0], geo=geo_of(x5)), self.fpn, self.fpn.module1, self.fpn.module1[0]))
        # f1 is read by head.yolo1[0] and by the stride-2 conv pan.conv1: either taking planes asks for the pre-split twin
        f1_twin = head_planes[0] or takes_planes(self.pan.conv1, x3.shape[2:], geo_of(x3))
        return self.pan(*self.fpn(x3, x4, x5, head_planes=f1_twin), head_planes=head_planes[1:])


class Head(nn.Module):

    def __init__(self, cfg: Dict, device=None):
        super().__init__()
        oc = (4 + 1 + cfg['N_CLASSES']) * 3
        self.yolo1 = nn.Sequential(ConvBNAct(128, 256, 3, 1, act=L),
                                   ConvBNAct(256, oc, 3, 1, bias=True, bn=False, act='linear'),
                                   YOLOLayer(cfg, layer_no=0, device=device))
        self.yolo2 = nn.Sequential(ConvBNAct(256, 512, 3, 1, act=L),
                                   ConvBNAct(512, oc, 1, 1, bias=True, bn=False, act='linear'),
                                   YOLOLayer(cfg, layer_no=1, device=device))
        self.yolo3 = nn.Sequential(ConvBNAct(512, 1024, 3, 1, act=L),
                                   ConvBNAct(1024, oc, 1, 1, bias=True, bn=False, act='linear'),
                                   YOLOLayer(cfg, layer_no=2, device=device))

    def logits(self, p1, p2, p3):
        assert p1.shape[1] == 128 and p2.shape[1] == 256 and p3.shape[1] == 512
        # the 3x3 conv's activation has ONE reader, the output conv: where that conv takes planes (training, ops.ConvBNActFn
        # "a conv WITHOUT BatchNorm over a pre-split input") and nobody is looking, it leaves pre-split
        out = []
        for h, p in ((self.yolo1, p1), (self.yolo2, p2), (self.yolo3, p3)):
            want = takes_planes(h[1], geo=geo_of(p)) and not observed(self, h, h[0], h[1])
            out.append(h[1](h[0](p, out_planes=bool(want))))
        return out

    def forward(self, p1, p2, p3):
        if observed(self.yolo1, self.yolo2, self.yolo3):
            return tuple(h(p) for h, p in ((self.yolo1, p1), (self.yolo2, p2), (self.yolo3, p3)))    # hooks on a Sequential: call it
        return tuple(h[2](lg) for h, lg in zip((self.yolo1, self.yolo2, self.yolo3), self.logits(p1, p2, p3)))


class YOLOv4(nn.Module):

    def __init__(self, cfg: Dict, device=None):
        super().__init__()
        assert cfg['TYPE'] == 'YOLOv4'
        self.backbone = Backbone()
        self.neck = Neck()
        self.head = Head(cfg, device=device)
        self._init(ckpt_path=cfg['BACKBONE_PRETRAINED'])

    def _init(self, ckpt_path=None):
        # yolov4.py:283-294: Kaiming-normal(fan_out, relu) conv weights, zero conv bias, BN gamma ~ N(0, 0.01)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)
        if ckpt_path is not None and os.path.isfile(ckpt_path):
            # files we did not write are only ever read with weights_only=True
            ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=True)['state_dict']
            ckpt = OrderedDict((k.replace("module.backbone.", ""), v) for k, v in ckpt.items() if 'backbone' in k)
            self.backbone.load_state_dict(ckpt, strict=True)

    def forward(self, x):
        if x.dtype != torch.float32:
            x = x.float()              # Transform hands float64 images; apex O0 casts them
        # (True: the head conv may be the SOLE reader of a pre-split tensor; 'both' where somebody may be looking on the way)
        B, H, W = geo_of(x)                      # the three heads read maps of stride 8 / 16 / 32
        hp = tuple(soft(takes_planes(h[0], geo=(B, -(-H // st), -(-W // st))), self.neck, self.neck.pan, self.head, h, h[0])
                   for h, st in ((self.head.yolo1, 8), (self.head.yolo2, 16), (self.head.yolo3, 32)))
        # who reads the backbone's three results: fpn.conv11 (x3), fpn.conv4 (x4), the first conv of the SPP block (x5, alone)
        seen = observed(self.backbone, self.neck, self.neck.spp, self.neck.spp.conv1, self.neck.fpn)
        readers = ((self.neck.fpn.conv11,), (self.neck.fpn.conv4,), () if seen else (self.neck.spp.conv1[0],))
        p1, p2, p3 = self.neck(*self.backbone(x, readers=readers), head_planes=hp)
        if self.training:
            return list(self.head(p1, p2, p3))
        if observed(self.head, self.head.yolo1, self.head.yolo2, self.head.yolo3):
            return torch.cat(list(self.head(p1, p2, p3)), dim=1)     # hooked head: the reference's call sequence (yolov4.py:316-324)
        # eval: the three decodes write into one [B, N, 5+C] buffer (the cat of yolov4.py:324, no copy)
        logits = self.head.logits(p1, p2, p3)
        layers = (self.head.yolo1[2], self.head.yolo2[2], self.head.yolo3[2])
        counts = [ly.n_anchors * lg.shape[2] * lg.shape[3] for ly, lg in zip(layers, logits)]
        n_total = sum(counts)
        out = torch.empty((x.shape[0], n_total, 5 + layers[0].n_classes), device=x.device, dtype=torch.float32)
        off = 0
        for ly, lg, n in zip(layers, logits, counts):
            ly.decode_into(lg, out, n_total, off)
            off += n
        return out
