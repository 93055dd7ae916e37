"""ResNet50-640 inference (BASELINE configs[1]) eager vs HIP-graph replay.  python tools/bench_infer.py [--batch 8]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--big-min-tiles", type=int, default=0, help="override the 256-row kernels' minimum tile count")
    ap.add_argument("--split-target", type=int, default=0, help="rn_launch_opts.splitk_target_blocks (128-row split-K)")
    a = ap.parse_args()
    from retinanet.cfg import default_params
    from retinanet.model import ModelBuilder
    dev = torch.device("cuda:0")
    p = default_params(input_size=a.size, inference_batch=a.batch)
    b = ModelBuilder(p, "val", device=dev, seed=1337)
    model = b()
    model.launch_opts = dict(conv_big_min_tiles=a.big_min_tiles or 0, splitk_target_blocks=a.split_target)
    x = torch.randn((a.batch, a.size, a.size, 3), device=dev)
    preds = model(x)
    std = torch.cat([preds["class-predictions"][l].reshape(-1) for l in "34567"]).std().item()
    model.variables["class-head/class-head-prediction-conv2d/kernel"].mul_(1.0 / max(std, 1e-12))
    model._refresh()
    for graph in (False, True):
        infer = b.add_post_processing_stage(model, capture_graph=graph)
        for _ in range(5):
            out = infer(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            out = infer(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        print(f"graph={graph}: {dt * 1e3:.3f} ms -> {a.batch / dt:.1f} img/s  valid {out['valid_detections'].tolist()[:4]}")


if __name__ == "__main__":
    main()
