#!/usr/bin/env python3
"""Wall time of one steady-state epoch of the trainer CLI on the config-2 dataset (100 classes x 1000 instances,
L = 1400, batch 64: 1562 full batches + a tail of 32, DF accuracy pass over the epoch's generated traces included).
Steady-state epochs only (log line to log line, epochs 3..5), so start-up, graph capture and the epoch-1 checkpoint
drop out."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXTRA = sys.argv[1:]          # extra trainer flags, e.g. --operands bf16x3
import bench  # noqa: E402


def main():
    r = bench.cli_epoch(torch.device("cuda:0"), epochs=5, extra=EXTRA)
    r["extra_args"] = EXTRA
    print(json.dumps(r))


if __name__ == "__main__":
    main()
