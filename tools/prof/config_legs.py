"""BASELINE configs[2] / configs[3] alone (bench.config_legs) - the command the round's config-3 / config-4 profiles are taken on:
    rocprofv3 --kernel-trace --stats -d gpurun_out/<tag>_config_stats -o run --output-format csv -- python3 tools/prof/config_legs.py
    rocprofv3 --pmc <counters> --kernel-trace ...                                               -- python3 tools/prof/config_legs.py
Prints the same `configs` object the benchmark's line carries."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import bench  # noqa: E402
from kzg_rs_amd import api  # noqa: E402

torch.cuda.set_device(0)
st = api.KzgSettings.load_trusted_setup_file()
print(json.dumps(bench.config_legs(st, torch, torch.device("cuda", 0), no_cpu="--no-cpu" in sys.argv)))
