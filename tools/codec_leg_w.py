"""bench.codec_leg with WINDOWS windows per call (env), for rocprofv3 runs of the small-call regime."""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "moss-ttsd_amd")); sys.path.insert(0, ROOT)
import torch, bench
print(json.dumps(bench.codec_leg(torch.device("cuda:0"), windows=int(os.environ.get("WINDOWS", "1")), reps=int(os.environ.get("MTTS_LEG_REPS", "3")))))
