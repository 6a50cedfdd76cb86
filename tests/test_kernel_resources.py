"""The compiler's own report of every kernel (build/<file>.remarks, written by the Makefile with
-Rpass-analysis=kernel-resource-usage): the kernels of the measured paths run at the register limit, and a change
that makes one of them spill costs 25-50 % of its speed without failing any parity test (round 3: four live
registers more in the epilogue, emit 725 -> 917 us; a register-pair constraint on the queue write, 670 -> 1010 us).
Runs wherever the library was built in-tree (the remarks do not travel to the GPU box: skipped there)."""
import glob
import os
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "transcoder-sae-circuits_amd", "csrc")


def _kernels():
    out = {}
    for path in glob.glob(os.path.join(CSRC, "build", "*.remarks")):
        name = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                out[name] = {}
                continue
            m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+) \[-Rpass-analysis", line)
            if m and name:
                out[name][m.group(1).strip()] = m.group(2)
    return out


def _built_after_sources():
    headers = glob.glob(os.path.join(CSRC, "*.h"))
    for src in glob.glob(os.path.join(CSRC, "*.hip")):
        rem = os.path.join(CSRC, "build", os.path.basename(src)[:-4] + ".remarks")
        if not os.path.exists(rem) or os.path.getmtime(rem) < max(os.path.getmtime(p) for p in [src] + headers) - 1:
            return False
    return True


@pytest.mark.skipif(not _built_after_sources(), reason="no up-to-date build/*.remarks (make -C transcoder-sae-circuits_amd/csrc)")
def test_measured_kernels_do_not_spill():
    ks = _kernels()
    assert len(ks) > 40, "the remarks of the six source files were not all found"
    # LDS-DMA variants of the encode body (template flags KALIGNED, DMA = true, true) in all three modes and both tiles,
    # the grouped screen emit, and every kernel of the lens tail / screened top-k
    hot = [n for n in ks if re.search(r"encode_f32_kernel.*ELi[012]ELb1ELb1ELb[01]ELb[01]EEE", n) or "encode_group_kernel" in n
           or re.search(r"sparse_finish|lens_assemble|rowdot|contrib_many|convert_x|row_bound|screen_recheck|topk_merge", n)]
    assert len(hot) >= 20, hot
    bad = {n: ks[n] for n in hot if int(ks[n]["ScratchSize [bytes/lane]"]) != 0 or int(ks[n]["VGPRs Spill"]) != 0}
    assert not bad, f"kernels of the measured paths spill: {bad}"
    # one workgroup of the wide tile per CU: its LDS image must fit in 160 KiB
    for n in hot:
        assert int(ks[n]["LDS Size [bytes/block]"]) <= 160 * 1024, (n, ks[n])
