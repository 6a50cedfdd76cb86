"""Name -> torch dtype table of the z-SAE configs (counterpart of reference ``src/dtypes.py:3``)."""
import torch

DTYPES = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}
