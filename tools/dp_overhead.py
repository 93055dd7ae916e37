"""extra.dp_overhead of bench.py on its own: python tools/dp_overhead.py  (GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


class A:
    train_batch, size = 32, 640


if __name__ == "__main__":
    torch.cuda.set_device(0)
    print(json.dumps(bench.run_dp_overhead(A(), torch.device("cuda", 0)), indent=1))
