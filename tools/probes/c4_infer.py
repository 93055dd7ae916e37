"""The serving leg of bench.py's `extra.config4` alone (EfficientNet-B3 640x640 batch 8, PerClassSoftNMS, class logits at
std 1): python tools/probes/c4_infer.py [--infer-batch 8]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--infer-batch", type=int, default=8)
ap.add_argument("--logit-std", type=float, default=1.0)
a = ap.parse_args()
print(json.dumps(bench.run_config4_infer(a, torch.device("cuda:0"))))
