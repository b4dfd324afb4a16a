"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/xq_hip.h declares (no compute without a GPU), and the product path refuses to run without it."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "xq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xq_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from xiangqi_alphazero_amd import hip
    hip.build()
    lib = hip.lib()
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/xq_hip.h but not exported"
    assert sorted(hip.EXPORTS) == names
    assert b"gfx950" in lib.xq_version()


def test_struct_sizes_match_header():
    from xiangqi_alphazero_amd import engine, hip
    assert engine.SAMPLE_DTYPE.itemsize == 640 and engine.RESULT_DTYPE.itemsize == 16
    assert ctypes.sizeof(hip.EngineStats) == 32 * 8
    cfg = engine.make_config(8192, 800)
    need = hip.lib().xq_engine_workspace_bytes(ctypes.byref(cfg))
    # tree arenas: G * (1 + (S+1)*128) nodes * 24 B dominate
    nodes = 8192 * (1 + 801 * 128)
    assert nodes * 24 < need < nodes * 24 * 1.15
    bad = engine.make_config(0, 800)
    assert hip.lib().xq_engine_workspace_bytes(ctypes.byref(bad)) == 0


def test_argument_errors_without_gpu():
    from xiangqi_alphazero_amd import hip
    lib = hip.lib()
    assert lib.xq_movegen_batch(None, None, -1, None, None, None, None, None) == -1
    assert lib.xq_movegen_batch(None, None, 0, None, None, None, None, None) == 0     # empty batch is a no-op
    assert lib.xq_encode_batch(None, None, 3, None, None) == -1
    assert lib.xq_engine_select(None, None, None) == -1
    # round-2 entry points: same conventions (null pointers / negative sizes -> XQ_ERR_ARG, empty batches are no-ops)
    assert lib.xq_policy_head_legal(None, None, None, None, None, 4, None, None) == -1
    assert lib.xq_policy_head_legal(None, None, None, None, None, 0, None, None) == 0
    assert lib.xq_value_head(None, None, None, None, None, 2, None, None) == -1
    assert lib.xq_value_head(None, None, None, None, None, 0, None, None) == 0
    assert lib.xq_engine_expand_legal(None, None, None, None) == -1
    assert lib.xq_engine_requests(None, None, None) == -1
    assert lib.xq_engine_drain_device(None, None, 0, None, None, 0, None, None) == -1
    assert lib.xq_wino_conv3x3(None, None, None, None, None, 8, 256, 1, None) == -1


def test_product_path_has_no_cpu_fallback():
    import torch
    from xiangqi_alphazero_amd import engine, hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hip.XqError):
        engine.SelfPlayEngine(engine.make_config(4, 8))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "xiangqi-alphazero_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "xq_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f
