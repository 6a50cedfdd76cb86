#!/usr/bin/env python3
"""Profiling target for BASELINE configs[3] (the batched CircuitLens pass): builds the bench's config-4 world once
and runs the pass a few times.  Used under rocprofv3 --kernel-trace --stats; not a test."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "transcoder-sae-circuits_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n_prompts = int(sys.argv[2]) if len(sys.argv) > 2 else 512
print(json.dumps(bench.config4_side_object(torch.device("cuda", 0), n_prompts=n_prompts, reps=reps)))
