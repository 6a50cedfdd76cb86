"""Test-only stand-in for bench.py's engine (``--engine-module cpu_bench_engine``): the CPU oracle behind the
five methods bench.py calls, so that the N > 1 launcher, the process group, the all-gather + merge and the
verification step can be driven end to end without a GPU.  Never used by the product or by a real bench run."""
import time

import torch

from oracle.encode_ref import encode_ref
from oracle.topk_ref import empty_state, topk_merge_ref, topk_stream_ref


class _Timer:
    def start(self):
        self.t0 = time.perf_counter()

    def stop(self):
        self.t1 = time.perf_counter()

    def ms(self):
        return (self.t1 - self.t0) * 1e3


class Engine:
    name = "oracle-cpu (test stand-in)"

    def __init__(self, local_rank):
        self.device = torch.device("cpu")
        torch.set_num_threads(2)

    def make_dict(self, W_enc, b_enc, b_dec):
        return {"W_enc": W_enc, "b_enc": b_enc, "b_dec": b_dec}

    def make_state(self, d_sae, k):
        return {"lists": empty_state(d_sae, k), "k": k}

    def encode_topk(self, d, x, base, st):
        st["lists"] = topk_stream_ref(x, d, k=st["k"], token_id_base=base, state=st["lists"])

    def encode_dense(self, d, x):
        return encode_ref(x, d["W_enc"], d["b_enc"], d["b_dec"])

    def finalize(self, st):
        return st["lists"]

    def merge(self, v, i):
        return topk_merge_ref([(v[r], i[r]) for r in range(v.shape[0])], k=v.shape[-1])

    def sync(self):
        pass

    def timer(self):
        return _Timer()

    def free_bytes(self):
        return 1 << 30
