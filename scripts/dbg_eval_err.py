import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch, recipe, yolov4_amd
from oracle import network as NW
from yolov4_amd.yolo.model.yolov4 import YOLOv4
dev = torch.device('cuda:0'); cfg = recipe.MODEL_CFG
model = YOLOv4(cfg, device=dev).to(dev)
for seed in (7, 1234, 99):
    sd = NW.empty_state_dict(); recipe.fill_state_dict_(sd, seed)
    x_cal = recipe.randn((8, 3, 64, 64), 3); x = recipe.randn((2, 3, 64, 64), 1)
    refs = {}
    for dt in (torch.float32, torch.float64):
        net = NW.RefNet({k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}, cfg)
        net.calibrate(x_cal.to(dt)); net.training = False
        with torch.no_grad(): lg = net.logits(x.to(dt))
        refs[dt] = [t.double() for t in lg]
    for mode in ('f32', 'bf16x3'):
        yolov4_amd.set_conv_mode(mode)
        model.load_state_dict(sd); recipe.calibrate_bn_(model, x_cal.to(dev)); model.eval()
        with torch.no_grad():
            lg = model.head.logits(*model.neck(*model.backbone(x.to(dev))))
        e64 = max(float((a.double().cpu() - b).abs().max()) for a, b in zip(lg, refs[torch.float64]))
        e32 = max(float((a.double().cpu() - b).abs().max()) for a, b in zip(lg, refs[torch.float32]))
        c32 = max(float((a - b).abs().max()) for a, b in zip(refs[torch.float32], refs[torch.float64]))
        print(f'seed {seed} mode {mode}: logits max|hip-fp64| {e64:.2e}  max|hip-cpu32| {e32:.2e}  max|cpu32-fp64| {c32:.2e}')
