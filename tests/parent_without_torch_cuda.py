"""Test helper: runs ``bench.py``'s ``main`` in a parent process in which EVERY ``torch.cuda`` entry that could open
the device raises -- the N > 1 launcher must not need any of them (its children are fresh processes)."""
import os
import sys

import torch


def _refuse(*a, **k):
    raise AssertionError("the launcher's parent process touched torch.cuda")


for _name in ("device_count", "is_available", "init", "current_device", "set_device", "synchronize", "mem_get_info"):
    setattr(torch.cuda, _name, _refuse)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.main(sys.argv[1:])
