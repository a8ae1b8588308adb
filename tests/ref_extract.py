"""Load the reference aggregator classes straight from /root/reference WITHOUT copying them.

TEST INFRASTRUCTURE (in-container only: /root/reference does not exist on the GPU box).

PathNet_run.py cannot be imported (argparse + a full training run at import time,
PathNet_run.py:44-64 and :406-485, and it imports torch_geometric which is not installed), so we
parse the file with ``ast`` and exec only the ClassDef nodes we need in a namespace where
``MessagePassing`` is ``nn.Module`` (the classes never call propagate()) and the
module globals ``device`` / ``dropout`` that forward() reads (PathNet_run.py:176,:194) are ours.
"""
import ast
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("PN_REFERENCE_ROOT", "/root/reference")


def have_reference():
    return os.path.exists(os.path.join(REF_ROOT, "PathNet_run.py"))


def _extract(path, names, dropout):
    with open(path) as f:
        tree = ast.parse(f.read())
    ns = {"torch": torch, "nn": nn, "F": F, "MessagePassing": nn.Module, "device": "cpu", "dropout": dropout}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
    return ns


def reference_classes(dropout=0.0):
    """-> dict with the reference's PathNet, PathNet_homo (PathNet_run.py:150-278) and PAGG
    (baseline/GPRGNN/src/copy.py:299-359) classes, running on CPU."""
    ns = _extract(os.path.join(REF_ROOT, "PathNet_run.py"), {"PathNet", "PathNet_homo"}, dropout)
    ns2 = _extract(os.path.join(REF_ROOT, "baseline", "GPRGNN", "src", "copy.py"), {"PAGG"}, dropout)
    return {"PathNet": ns["PathNet"], "PathNet_homo": ns["PathNet_homo"], "PAGG": ns2["PAGG"], "_ns": (ns, ns2)}
