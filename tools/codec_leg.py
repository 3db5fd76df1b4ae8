"""bench.py's codec leg alone (8 windows of 375 codes, full-depth decoder): for rocprofv3 runs."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

print(json.dumps(bench.codec_leg(torch.device("cuda:0"), reps=int(os.environ.get("MTTS_LEG_REPS", "3")))))
