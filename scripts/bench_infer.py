"""Config 2 (BASELINE.json configs[1]): 1xMI355X inference, 608x608 bs=32: eval forward (BN folded into the conv
epilogues) + postprocess on calibrated random weights.  Not the headline metric; numbers go to DESIGN.md."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, recipe, yolov4_amd
from yolov4_amd.yolo.model.yolov4 import YOLOv4
from yolov4_amd.yolo.util.utils import postprocess
dev = torch.device('cuda:0')
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 608
m = YOLOv4(recipe.MODEL_CFG, device=dev); sd = m.state_dict(); recipe.fill_state_dict_(sd, 1234); m.load_state_dict(sd); m = m.to(dev)
recipe.calibrate_bn_(m, recipe.randn((8, 3, S, S), 77).to(dev))
m.eval()
x = recipe.randn((B, 3, S, S), 78).to(dev)
with torch.no_grad():
    for _ in range(2): out = m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
sc = (out[..., 4:5] * out[..., 5:]).flatten()
thr = float(torch.quantile(sc[torch.randint(0, sc.numel(), (1000000,), device=dev)], 1 - 500.0 / (out.shape[1] * 80)))  # ~500 candidates / image
torch.cuda.synchronize(); t0 = time.perf_counter()
det = postprocess(out.clone(), 80, thr, 0.4)
torch.cuda.synchronize(); dp = time.perf_counter() - t0
# eval input pipeline: B camera-sized BGR uint8 images already on the device -> [B,3,S,S]
import numpy as np
from yolov4_amd.yolo.data.transform import val_batch
from yolov4_amd.yolo.util.utils import detections_to_coco
rng = np.random.RandomState(3)
imgs = [torch.from_numpy(rng.randint(0, 256, (480, 640, 3)).astype(np.uint8)).to(dev) for _ in range(B)]
val_batch(imgs, S, dev); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): xb, infos = val_batch(imgs, S, dev)
torch.cuda.synchronize(); dpre = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
recs = [r for i, d in enumerate(det) for r in detections_to_coco(d, infos[i], i)]
drec = time.perf_counter() - t0
print(json.dumps({'preprocess_ms_per_batch': dpre * 1e3, 'coco_records': len(recs), 'coco_records_ms': drec * 1e3,
                  'batch': B, 'eval_forward_ms': dt * 1e3, 'img_per_s_forward': B / dt, 'conv_tflops': B / dt * 134.422e9 / 1e12,
                  'postprocess_ms': dp * 1e3, 'conf_thre': thr, 'survivors_per_img': sum(0 if d is None else len(d) for d in det) / B,
                  'conv_mode': yolov4_amd.get_conv_mode()}))
