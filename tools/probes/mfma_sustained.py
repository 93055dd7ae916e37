"""prints bench.mfma_sustained() (csrc/rn_probe.hip: MFMA-only kernel, random vs zero operands, core clock) as JSON"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

print(json.dumps(bench.mfma_sustained(torch.device("cuda:0")), indent=1))
