"""A few seconds of the randomised comparisons in ``tools/`` (shapes, list lengths, slot counts, staging variants,
duplicated rows that the fixed tests do not enumerate): the fused top-k against the exact ranking of the library's own
dense activations, the bf16-screened top-k against the fp32 one, bit for bit, and the lens pass (plain: exactly the
sorted positives of the dense activations; screened: no active feature above the fp32 rounding noise lost or invented)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seed,cases", [("fuzz_topk.py", 11, 500), ("fuzz_screened.py", 12, 150), ("fuzz_lens.py", 13, 300),
                                             ("fuzz_merge.py", 14, 500), ("fuzz_encode.py", 15, 1000), ("fuzz_aux.py", 16, 1000), ("fuzz_grouped.py", 17, 300)])
def test_randomised_list_checks(tool, seed, cases):
    """A fixed number of cases from a fixed seed (the same cases on every machine); the time budget is only a guard."""
    p = subprocess.run([sys.executable, os.path.join(REPO, "tools", tool), "90", str(seed), str(cases)],
                       capture_output=True, text=True, timeout=115)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    last = p.stdout.strip().splitlines()[-1]
    assert last.startswith("ok:") and f"{cases} cases" in last, last
