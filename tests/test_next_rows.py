# -*- coding: utf-8 -*-
"""SURVEY 8(f) "next" rows: optimizer factory / groups, LR schedule, checkpoint I/O (CPU side),
fused Adam against torch.optim.Adam (GPU)."""
import math
import os

import numpy as np
import pytest
import torch

import recipe
from yolov4_amd.yolo.model.yolov4 import YOLOv4
from yolov4_amd.yolo.optim.lr_schedulers.build import adjust_learning_rate, build_lr_scheduler
from yolov4_amd.yolo.optim.optimizers.build import FusedAdam, build_optimizer, filter_weight
from yolov4_amd.yolo.util.checkpoint import load_checkpoint, save_checkpoint

CFG = dict(recipe.FULL_CFG)
CFG['OPTIMIZER'] = {'TYPE': 'ADAM', 'LR': '3e-4', 'NO_BIAS': True, 'NO_NORM': True}
CFG['LR_SCHEDULER'] = {'TYPE': 'MultiStepLR', 'MILESTONES': [60, 90, 110], 'GAMMA': 0.1, 'IS_WARMUP': True,
                       'WARMUP_EPOCH': 5, 'MULTIPLIER': 1.0}
CFG['TRAIN'] = {'IMGSIZE': 608, 'MAX_EPOCHS': 120, 'ACCUMULATION_STEPS': 4}


def test_param_groups_match_reference_rule():
    m = YOLOv4(recipe.MODEL_CFG)
    groups = filter_weight(CFG, m)
    assert len(groups[0]['params']) == 110                 # every conv weight decays (config/yolov4_default.cfg:29-30)
    assert len(groups[1]['params']) == 3 + 2 * 107         # 3 head biases + gamma/beta of 107 BatchNorms
    assert groups[1]['weight_decay'] == 0.
    opt = build_optimizer(CFG, m)
    assert isinstance(opt, FusedAdam) and opt.defaults['lr'] == 3e-4 and opt.defaults['betas'] == (0.9, 0.999)
    with pytest.raises(ValueError):
        build_optimizer(dict(CFG, OPTIMIZER=dict(CFG['OPTIMIZER'], TYPE='LAMB')), m)


def test_warmup_and_multistep_schedule():
    lin = torch.nn.Linear(2, 2)
    opt = FusedAdam(lin.parameters(), lr=3e-4)
    len_epoch = 100
    adjust_learning_rate(CFG, opt, epoch=0, step=0, len_epoch=len_epoch)
    assert math.isclose(opt.param_groups[0]['lr'], 3e-4 * 1 / 500)
    adjust_learning_rate(CFG, opt, epoch=2, step=49, len_epoch=len_epoch)
    assert math.isclose(opt.param_groups[0]['lr'], 3e-4 * 250 / 500)
    adjust_learning_rate(CFG, opt, epoch=5, step=0, len_epoch=len_epoch)
    assert math.isclose(opt.param_groups[0]['lr'], 3e-4)
    sch = build_lr_scheduler(CFG, opt)
    assert list(sch.milestones) == [55, 85, 105]           # milestones shifted by the warm-up epochs


def test_checkpoint_roundtrip_reference_format(tmp_path):
    m = YOLOv4(recipe.MODEL_CFG)
    sd = m.state_dict()
    recipe.fill_state_dict_(sd, 5)
    m.load_state_dict(sd)
    state = {'epoch': 3, 'ap50': 0.1, 'ap50_95': 0.05, 'best_ap50': 0.1, 'best_ap50_95': 0.05,
             'state_dict': {'module.' + k: v for k, v in m.state_dict().items()},      # as saved under DDP
             'optimizer': {}, 'lr_scheduler': {}}
    path = save_checkpoint(state, True, 'checkpoint_3.pth.tar', str(tmp_path))
    assert os.path.isfile(os.path.join(str(tmp_path), 'model_best.pth.tar'))
    raw = torch.load(path, map_location='cpu', weights_only=True)
    w = raw['state_dict']['module.backbone.stage1.base.conv.weight']
    assert w.is_contiguous() and tuple(w.shape) == (64, 32, 3, 3)                   # OIHW on the wire
    m2 = YOLOv4(recipe.MODEL_CFG)
    ck = load_checkpoint(m2, path)
    assert ck['epoch'] == 3
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    assert m2.backbone.stage1.base.conv.weight.is_contiguous(memory_format=torch.channels_last)


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam():
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 32, 3, 3), (255,), (128,), (32, 3, 3, 3), (1024, 512, 1, 1)]
    ref_p = [torch.randn(s, generator=g) for s in shapes]
    hip_p = [torch.nn.Parameter((p.clone().contiguous(memory_format=torch.channels_last) if p.dim() == 4 else p.clone()).to(dev))
             for p in ref_p]
    ref_p = [torch.nn.Parameter(p) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=3e-4, betas=(0.9, 0.999), eps=1e-8)
    hip = FusedAdam(hip_p, lr=3e-4)
    for step in range(4):
        for rp, hp in zip(ref_p, hip_p):
            gr = torch.randn(rp.shape, generator=g) * (10.0 ** (step - 2))
            rp.grad = gr.clone()
            hp.grad = (gr.contiguous(memory_format=torch.channels_last) if gr.dim() == 4 else gr).to(dev)
        if step == 2:
            for o in (ref, hip):
                for gp in o.param_groups:
                    gp['lr'] = 1e-4
        ref.step(); hip.step()
    for rp, hp in zip(ref_p, hip_p):
        np.testing.assert_allclose(hp.detach().cpu().numpy(), rp.detach().numpy(), rtol=2e-6, atol=2e-7)


# ------------------------------------------------------------------ row 4: eval input pipeline + COCO records
def test_coco_class_table_and_records_golden():
    from yolov4_amd.yolo.util.utils import COCO_CLASS_IDS, detections_to_coco
    assert len(COCO_CLASS_IDS) == 80 and COCO_CLASS_IDS[0] == 1 and COCO_CLASS_IDS[-1] == 90
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'coco.npz'))
    assert COCO_CLASS_IDS == list(g['class_ids'])                              # the reference's table, as data
    for ci in range(3):
        det = torch.from_numpy(g[f'c{ci}.det'])
        recs = detections_to_coco(det, list(g[f'c{ci}.info']), image_id=42 + ci)
        assert len(recs) == det.shape[0]
        for r, bb, sc, d in zip(recs, g[f'c{ci}.bbox'], g[f'c{ci}.score'], g[f'c{ci}.det']):
            assert r['bbox'] == list(bb) and r['score'] == sc                    # double arithmetic: bit-exact
            assert r['category_id'] == COCO_CLASS_IDS[int(d[6])] and r['image_id'] == 42 + ci
    assert detections_to_coco(None, [1, 1, 1, 1], 0) == []


def _float_bilinear(img, S):
    x = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
    return torch.nn.functional.interpolate(x, size=(S, S), mode='bilinear', align_corners=False)[0].permute(1, 2, 0).numpy()


@pytest.mark.parametrize('shape,S', [((48, 64), 32), ((37, 53), 64), ((64, 64), 64), ((128, 128), 64), ((5, 7), 16)])
def test_resize_oracle_within_one_lsb_of_float_bilinear(shape, S):
    """cv2 is absent (parity unpinned): the fixed-point restatement must agree with exact bilinear sampling at
    the same half-pixel-centre coordinates to 1 LSB (the box average of the 2x case equals bilinear there)."""
    from oracle import preprocess as OP
    img = np.random.RandomState(shape[0] * 131 + S).randint(0, 256, shape + (3,)).astype(np.uint8)
    got = OP.resize_linear_u8(img, S).astype(np.float64)
    assert np.abs(got - _float_bilinear(img, S)).max() <= 1.0
    if shape == (S, S):
        assert np.array_equal(got, img)


@pytest.mark.gpu
@pytest.mark.parametrize('shape,S', [((480, 640), 416), ((333, 500), 608), ((1216, 1216), 608), ((608, 608), 608),
                                     ((9, 1300), 64), ((31, 17), 416)])
def test_gpu_val_transform_bit_exact_vs_oracle(shape, S):
    from oracle import preprocess as OP
    from yolov4_amd.yolo.data.transform import val_batch, val_transform
    img = np.random.RandomState(shape[1] + S).randint(0, 256, shape + (3,)).astype(np.uint8)
    ref, info_ref = OP.val_input(img, S)
    out, info = val_transform(img, S)
    assert info == info_ref
    assert np.array_equal(out.cpu().numpy(), ref)
    batch, infos = val_batch([img, img[:, ::-1]], S)                         # strided view as second image
    assert np.array_equal(batch[0].cpu().numpy(), ref)
    assert np.array_equal(batch[1].cpu().numpy(), OP.val_input(np.ascontiguousarray(img[:, ::-1]), S)[0])


@pytest.mark.gpu
def test_training_loop_end_to_end_loss_falls():
    """The reference's train() sequence (yolo/engine/build.py:56-69) on one fixed synthetic batch: model -> YOLOLoss ->
    backward through BucketedDDP (gradients written into the flat bucket slots) -> fused Adam.  The loss must stay
    finite and fall; the parameters must actually move and the BN running statistics must be tracked."""
    from yolov4_amd.ddp import BucketedDDP
    from yolov4_amd.yolo.model.yololoss import YOLOLoss
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    m = YOLOv4(recipe.MODEL_CFG, device=dev).to(dev).train()
    w0 = m.backbone.stem.conv.weight.detach().clone()
    ddp = BucketedDDP(m)
    opt = build_optimizer(CFG, m)
    crit = YOLOLoss(recipe.MODEL_CFG, 0.7, device=dev)
    x = recipe.randn((4, 3, 160, 160), 80).to(dev)
    labels = recipe.synth_labels(4, 160, 81)
    losses = []
    for _ in range(15):
        ddp.zero_grad()
        loss = crit(ddp(x), {'padded_labels': labels})
        loss.backward()
        ddp.finish_backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(math.isfinite(v) for v in losses), losses
    assert losses[-1] < 0.98 * losses[0] and min(losses[8:]) < min(losses[:4]), losses
    assert not torch.equal(w0, m.backbone.stem.conv.weight.detach())
    assert int(m.backbone.stem.norm.num_batches_tracked) == 15
