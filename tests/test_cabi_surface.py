"""CPU-side checks of the drop-in boundary: the shared library builds in-tree, loads, and exports
exactly the entry points ``include/tsc_hip.h`` declares.  No compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, "include", "tsc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsc_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import tsc_hip

    if not os.path.exists(tsc_hip.library_path()):
        pytest.skip("libtsc_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(tsc_hip.library_path())
    declared = _declared()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in tsc_hip.h but not exported"
    assert sorted(tsc_hip.EXPORTED_SYMBOLS) == declared, "ctypes binding and header disagree"
    assert tsc_hip.lib().tsc_version() >= 100


def test_argument_errors_without_gpu():
    import tsc_hip

    if not os.path.exists(tsc_hip.library_path()):
        pytest.skip("libtsc_hip.so not built")
    lib = tsc_hip.lib()
    handle = ctypes.c_void_p()
    # null weights and bad dimensions are rejected before any HIP call
    assert lib.tsc_dict_create(ctypes.byref(handle), 768, 24576, 768, 0, None, None, None, None, None, None) == -1
    assert b"required" in lib.tsc_last_error()
    assert lib.tsc_dict_create(ctypes.byref(handle), 770, 24576, 768, 0, 16, 16, 16, None, None, None) == -3
    assert lib.tsc_encode_dense(None, None, 4, None, None) == -1
    assert lib.tsc_topk_recommended_slots(24576) == 8
    assert lib.tsc_topk_recommended_slots(49152) == 8   # (4 also fills the chip; 8 keeps an XCD's W_enc slabs within its L2)
    assert lib.tsc_topk_recommended_slots(131072) == 8


def test_product_path_has_no_cpu_fallback():
    import torch
    import tsc_hip

    w = torch.zeros(8, 16)
    with pytest.raises(tsc_hip.TscError):
        tsc_hip.DeviceDict(w, torch.zeros(16), torch.zeros(8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "transcoder-sae-circuits_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_header_is_plain_c(tmp_path):
    """include/tsc_hip.h is the contract a non-Python caller compiles against: it must be valid C99 on its own
    and the state struct must have the layout the ctypes binding assumes."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "tsc_hip.h"\n'
        "int main(void){ printf(\"%zu %zu %zu %zu\\n\", sizeof(tsc_topk_state_t), offsetof(tsc_topk_state_t, d_sae),"
        " offsetof(tsc_topk_state_t, flags), offsetof(tsc_topk_state_t, floor)); return 0; }\n")
    exe = tmp_path / "t"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{inc}", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    import ctypes

    import tsc_hip

    st = tsc_hip._TopkStateStruct
    assert [int(v) for v in out] == [ctypes.sizeof(st), st.d_sae.offset, st.flags.offset, st.floor.offset]
