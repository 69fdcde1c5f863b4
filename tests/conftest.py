import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# Collection order of the -m gpu suite (the driver runs it with -x): every file that compares a HIP kernel with the
# oracle or with the reference's golden fixtures runs FIRST, the product-against-itself files after them, and the one
# statistical test (a training run) LAST — a red test there must never hide a parity row.  Files not listed keep their
# alphabetical place in front of the list (the CPU suites).
GPU_ORDER = ["test_gpu_kernels", "test_gpu_model", "test_gpu_configs", "test_gpu_lowp", "test_gpu_p3", "test_gpu_train",
             "test_gpu_train_lp", "test_gpu_wgrad_det", "test_gpu_bn_fusion", "test_gpu_bn_pool", "test_gpu_convergence"]


def _file_rank(item):
    stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    return GPU_ORDER.index(stem) if stem in GPU_ORDER else (-1 if not stem.startswith("test_gpu") else len(GPU_ORDER) - 2)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_file_rank)                     # stable: the order inside a file is kept
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
