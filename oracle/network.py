# -*- coding: utf-8 -*-
"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- torch-CPU-fp32 functional
restatement of the reference's conv stacks on a state_dict keyed exactly like the
reference's 648 keys.  Differentiable through torch autograd (the oracle for
dgrad / wgrad / BN-backward parity).

Follows darknet/darknet.py:14-138 (Mish, ConvBNAct, ResBlock, CSPDownSample0,
CSPDownSample) and yolo/model/yolov4.py:26-324 (Backbone, SPPBlock, Upsample,
FPNBlock, PANBlock, Neck, Head, YOLOv4.forward).
"""
import collections

import numpy as np
import torch
import torch.nn.functional as F

from . import head as H


def mish(x):
    return x * torch.tanh(F.softplus(x))                      # darknet.py:18-20


ACTS = {
    'mish': mish,
    'leaky_relu': lambda x: F.leaky_relu(x, 0.1),             # darknet.py:44
    'relu': F.relu,
    'linear': lambda x: x,
}


class RefNet:
    """Holds params (requires_grad leaves) + buffers; `sd` uses reference key names."""

    def __init__(self, sd, cfg=None):
        self.p = collections.OrderedDict()
        for k, v in sd.items():
            v = v.detach().clone()
            if v.is_floating_point() and not ('running_' in k):
                v.requires_grad_(True)
            self.p[k] = v
        self.cfg = cfg
        self.training = False
        self.momentum = 0.1

    # -- darknet.py:23-58: conv -> BN -> act
    def cba(self, x, pre, k, s, act, bn=True):
        p = self.p
        x = F.conv2d(x, p[pre + '.conv.weight'], p.get(pre + '.conv.bias'), stride=s, padding=(k - 1) // 2)
        if bn:
            if self.training:
                p[pre + '.norm.num_batches_tracked'] += 1
            x = F.batch_norm(x, p[pre + '.norm.running_mean'], p[pre + '.norm.running_var'],
                             p[pre + '.norm.weight'], p[pre + '.norm.bias'],
                             training=self.training, momentum=self.momentum, eps=1e-5)
        return ACTS[act](x)

    def resblock(self, x, pre, n, act='mish'):                # darknet.py:61-81
        for i in range(n):
            h = self.cba(x, f'{pre}.module_list.{i}.0', 1, 1, act)
            h = self.cba(h, f'{pre}.module_list.{i}.1', 3, 1, act)
            x = x + h
        return x

    def csp0(self, x, pre):                                   # darknet.py:84-113
        x = self.cba(x, pre + '.base', 3, 2, 'mish')
        x1 = self.cba(x, pre + '.part1', 1, 1, 'mish')
        x211 = self.cba(x, pre + '.part2_1_1', 1, 1, 'mish')
        h = self.cba(x211, pre + '.part2_1_2.0', 1, 1, 'mish')
        h = self.cba(h, pre + '.part2_1_2.1', 3, 1, 'mish')
        x2 = self.cba(x211 + h, pre + '.part2_2', 1, 1, 'mish')
        return self.cba(torch.cat([x2, x1], 1), pre + '.transition', 1, 1, 'mish')

    def csp(self, x, pre, n):                                 # darknet.py:116-138
        x = self.cba(x, pre + '.base', 3, 2, 'mish')
        x1 = self.cba(x, pre + '.part1', 1, 1, 'mish')
        x2 = self.cba(x, pre + '.part2.0', 1, 1, 'mish')
        x2 = self.resblock(x2, pre + '.part2.1', n)
        x2 = self.cba(x2, pre + '.part2.2', 1, 1, 'mish')
        return self.cba(torch.cat([x2, x1], 1), pre + '.transition', 1, 1, 'mish')

    def backbone(self, x):                                    # yolov4.py:26-47
        x = self.cba(x, 'backbone.stem', 3, 1, 'mish')
        x = self.csp0(x, 'backbone.stage1')
        x = self.csp(x, 'backbone.stage2', 2)
        x3 = self.csp(x, 'backbone.stage3', 8)
        x4 = self.csp(x3, 'backbone.stage4', 8)
        x5 = self.csp(x4, 'backbone.stage5', 4)
        return x3, x4, x5

    def seq(self, x, pre, spec):
        for i, (k, act) in enumerate(spec):
            x = self.cba(x, f'{pre}.{i}', k, 1, act)
        return x

    def spp(self, x):                                         # yolov4.py:50-74 (pools 5, 9, 5: quirk D7)
        L = 'leaky_relu'
        x = self.seq(x, 'neck.spp.conv1', [(1, L), (3, L), (1, L)])
        m1 = F.max_pool2d(x, 5, 1, 2)
        m2 = F.max_pool2d(x, 9, 1, 4)
        m3 = F.max_pool2d(x, 5, 1, 2)
        return self.cba(torch.cat([m3, m2, m1, x], 1), 'neck.spp.conv2', 1, 1, L)

    @staticmethod
    def up(x, size):                                          # yolov4.py:77-90 (nearest)
        return F.interpolate(x, size=size, mode='nearest')

    def neck(self, x3, x4, x5):                               # yolov4.py:93-224
        L = 'leaky_relu'
        five = [(1, L), (3, L), (1, L), (3, L), (1, L)]
        s = self.spp(x5)
        f3 = self.seq(s, 'neck.fpn.module1', [(3, L), (1, L)])
        f2 = self.up(self.cba(f3, 'neck.fpn.conv3', 1, 1, L), x4.shape[2:])
        f2 = self.seq(torch.cat((self.cba(x4, 'neck.fpn.conv4', 1, 1, L), f2), 1), 'neck.fpn.module2', five)
        f1 = self.up(self.cba(f2, 'neck.fpn.conv10', 1, 1, L), x3.shape[2:])
        f1 = self.seq(torch.cat((self.cba(x3, 'neck.fpn.conv11', 1, 1, L), f1), 1), 'neck.fpn.module3', five)
        p2 = self.cba(f1, 'neck.pan.conv1', 3, 2, L)
        p2 = self.seq(torch.cat((p2, f2), 1), 'neck.pan.module1', five)
        p3 = self.cba(p2, 'neck.pan.conv7', 3, 2, L)
        p3 = self.seq(torch.cat((p3, f3), 1), 'neck.pan.module2', five)
        return f1, p2, p3

    def head_logits(self, p1, p2, p3):                        # yolov4.py:227-268 minus YOLOLayer
        L = 'leaky_relu'
        outs = []
        for name, x, k in (('head.yolo1', p1, 3), ('head.yolo2', p2, 1), ('head.yolo3', p3, 1)):
            x = self.cba(x, name + '.0', 3, 1, L)
            outs.append(self.cba(x, name + '.1', k, 1, 'linear', bn=False))
        return outs

    def logits(self, x):
        return self.head_logits(*self.neck(*self.backbone(x)))

    def forward_eval(self, x):
        """yolov4.py:304-324 eval branch: cat of the three decoded heads."""
        self.training = False
        with torch.no_grad():
            lg = self.logits(x)
        return np.concatenate([H.yolo_decode(t.numpy(), l, self.cfg, False) for l, t in enumerate(lg)], 1)

    def forward_train(self, x):
        """Train-mode forward (BN batch stats, running stats updated).  Returns the
        three head-logit tensors still attached to the autograd graph."""
        self.training = True
        return self.logits(x)

    def train_step(self, x, labels, ignore_thresh=0.7):
        """fwd -> loss -> bwd, the call order of yolo/engine/build.py:60-65.
        Returns (loss, logits list); parameter grads are left in self.p[k].grad."""
        lg = self.forward_train(x)
        loss, glog = H.yolo_loss([t.detach().numpy() for t in lg], np.asarray(labels), self.cfg, ignore_thresh)
        torch.autograd.backward(lg, [torch.from_numpy(g) for g in glog])
        return loss, lg

    def calibrate(self, x_cal):
        """recipe.calibrate_bn_ for this class (momentum 1.0 train forward)."""
        old = self.momentum
        self.momentum = 1.0
        with torch.no_grad():
            self.forward_train(x_cal)
        self.momentum = old
        self.training = False

    def state_dict(self):
        return collections.OrderedDict((k, v.detach()) for k, v in self.p.items())


def yolov4_state_dict_spec(n_classes=80):
    """The 648 (name, shape) pairs of the reference's YOLOv4.state_dict(), derived
    from the architecture (Appendix A of SURVEY.md), in the reference's order."""
    spec = []

    def cba(pre, cin, cout, k, bn=True):
        spec.append((pre + '.conv.weight', (cout, cin, k, k)))
        if bn:
            spec.extend([(pre + '.norm.weight', (cout,)), (pre + '.norm.bias', (cout,)),
                         (pre + '.norm.running_mean', (cout,)), (pre + '.norm.running_var', (cout,)),
                         (pre + '.norm.num_batches_tracked', ())])
        else:
            spec.append((pre + '.conv.bias', (cout,)))

    cba('backbone.stem', 3, 32, 3)
    p = 'backbone.stage1'
    cba(p + '.base', 32, 64, 3); cba(p + '.part1', 64, 64, 1); cba(p + '.part2_1_1', 64, 64, 1)
    cba(p + '.part2_1_2.0', 64, 32, 1); cba(p + '.part2_1_2.1', 32, 64, 3)
    cba(p + '.part2_2', 64, 64, 1); cba(p + '.transition', 128, 64, 1)
    for st, cin, cout, n in ((2, 64, 128, 2), (3, 128, 256, 8), (4, 256, 512, 8), (5, 512, 1024, 4)):
        p = f'backbone.stage{st}'
        h = cout // 2
        cba(p + '.base', cin, cout, 3); cba(p + '.part1', cout, h, 1); cba(p + '.part2.0', cout, h, 1)
        for i in range(n):
            cba(f'{p}.part2.1.module_list.{i}.0', h, h, 1)
            cba(f'{p}.part2.1.module_list.{i}.1', h, h, 3)
        cba(p + '.part2.2', h, h, 1); cba(p + '.transition', cout, cout, 1)
    cba('neck.spp.conv1.0', 1024, 512, 1); cba('neck.spp.conv1.1', 512, 1024, 3); cba('neck.spp.conv1.2', 1024, 512, 1)
    cba('neck.spp.conv2', 2048, 512, 1)
    cba('neck.fpn.module1.0', 512, 1024, 3); cba('neck.fpn.module1.1', 1024, 512, 1)
    cba('neck.fpn.conv3', 512, 256, 1); cba('neck.fpn.conv4', 512, 256, 1)

    def five(pre, cbig, csmall):
        cba(pre + '.0', cbig, csmall, 1); cba(pre + '.1', csmall, cbig, 3); cba(pre + '.2', cbig, csmall, 1)
        cba(pre + '.3', csmall, cbig, 3); cba(pre + '.4', cbig, csmall, 1)

    five('neck.fpn.module2', 512, 256)
    cba('neck.fpn.conv10', 256, 128, 1); cba('neck.fpn.conv11', 256, 128, 1)
    five('neck.fpn.module3', 256, 128)
    cba('neck.pan.conv1', 128, 256, 3); five('neck.pan.module1', 512, 256)
    cba('neck.pan.conv7', 256, 512, 3); five('neck.pan.module2', 1024, 512)
    oc = 3 * (5 + n_classes)
    cba('head.yolo1.0', 128, 256, 3); cba('head.yolo1.1', 256, oc, 3, bn=False)
    cba('head.yolo2.0', 256, 512, 3); cba('head.yolo2.1', 512, oc, 1, bn=False)
    cba('head.yolo3.0', 512, 1024, 3); cba('head.yolo3.1', 1024, oc, 1, bn=False)
    return spec


def empty_state_dict(n_classes=80):
    sd = collections.OrderedDict()
    for k, shp in yolov4_state_dict_spec(n_classes):
        sd[k] = torch.zeros(shp, dtype=torch.int64 if k.endswith('num_batches_tracked') else torch.float32)
    return sd
