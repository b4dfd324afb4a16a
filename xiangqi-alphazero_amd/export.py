"""Model export -- the reference's `export_model.py` (training/export_model.py:17-85) for checkpoints written by either
trainer: `export_to_torchscript(model_path, output_path)` and `export_to_onnx(model_path, output_path)` with the
reference's signatures, input name / output names and dynamic batch axis.

The checkpoint is read with `weights_only=True` (nothing in the file is executed; the reference calls a bare
`torch.load`).  TorchScript round-trips are checked against the reference network's recorded outputs
(tests/test_host_logic.py).  ONNX needs the `onnx` package, which this image does not ship: the function is provided
for deployments that have it and its parity is unpinned here (no runtime to check the file against).
"""
from __future__ import annotations

import torch

from .model import ACTION_SPACE, COLS, ROWS, XiangqiNet, load_reference_checkpoint


def _load(model_path: str) -> XiangqiNet:
    net = load_reference_checkpoint(model_path)        # best_model.pt / checkpoint_iter*.pt, train.py:537-567
    net.eval()
    return net


def export_to_torchscript(model_path: str, output_path: str) -> str:
    """training/export_model.py:71-85: trace the eval-mode module on a [1,15,10,9] input and save it."""
    net = _load(model_path)
    traced = torch.jit.trace(net, torch.randn(1, 15, ROWS, COLS))
    traced.save(output_path)
    return output_path


def export_to_onnx(model_path: str, output_path: str) -> str:
    """training/export_model.py:17-49: opset 13, constant folding, input 'state', outputs 'policy' / 'value', dynamic
    batch axis on all three."""
    net = _load(model_path)
    try:
        import onnx  # noqa: F401
    except ImportError as e:
        raise RuntimeError("export_to_onnx needs the `onnx` package (not installed here); "
                           "export_to_torchscript has no extra dependency") from e
    torch.onnx.export(net, torch.randn(1, 15, ROWS, COLS), output_path, export_params=True, opset_version=13,
                      do_constant_folding=True, input_names=["state"], output_names=["policy", "value"],
                      dynamic_axes={"state": {0: "batch_size"}, "policy": {0: "batch_size"}, "value": {0: "batch_size"}})
    return output_path


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="export a checkpoint (reference CLI: export_model.py:88-101)")
    ap.add_argument("--model", required=True)
    ap.add_argument("--output", default="model.onnx")
    ap.add_argument("--format", default="onnx", choices=["onnx", "torchscript"])
    a = ap.parse_args()
    (export_to_onnx if a.format == "onnx" else export_to_torchscript)(a.model, a.output)
    print("exported %s (input (batch,15,%d,%d) -> policy (batch,%d), value (batch,1))" % (a.output, ROWS, COLS, ACTION_SPACE))
