"""Evaluation harness parity: eval.py:146-215 on top of the GVCNN engine.

Per batch the reference runs partial_run #1 (view scores), the host group_scheme / group_weight, partial_run #2
(accuracy, confusion matrix) and reports the MEAN OF THE PER-BATCH ACCURACIES (eval.py:204,213 — not the pooled
accuracy: a short last batch weighs as much as a full one) and the summed confusion matrix.  Here the grouping stays
on the device (`fused=True`) or follows the two-phase protocol with the reference-shaped host functions; argmax,
correct count and confusion matrix are one kernel (gv_eval_metrics).  Dataset sizes are counted, not the
hard-coded constants of eval.py:26.
"""
import torch

from . import _lib
from . import model as _model


class Evaluator:
    def __init__(self, engine, num_classes=None):
        self.eng = engine
        self.C = num_classes if num_classes is not None else engine.num_classes
        dev = engine.device
        self.confusion = torch.zeros((self.C, self.C), dtype=torch.int32, device=dev)
        self._correct = torch.zeros(1, dtype=torch.int32, device=dev)
        self._pred = torch.empty(engine.N, dtype=torch.int64, device=dev)
        self.batch_accuracies = []
        self.num_shapes = 0

    def add_batch(self, views, labels, fused=True, valid=None):
        """views [N,V,H,W,3], labels [N] (int64).  Returns this batch's accuracy (reads one int back).  valid: the
        number of real shapes of a padded last batch (ViewBatcher(remainder="pad"): labels of the padding are -1 and are
        ignored by gv_eval_metrics); the batch then weighs as one batch of `valid` shapes, like eval.py:204."""
        eng = self.eng
        with torch.cuda.device(eng.device):
            return self._add_batch(views, labels, fused, eng.N if valid is None else int(valid))

    def _add_batch(self, views, labels, fused, valid):
        eng = self.eng
        if fused:
            _, _, logits = eng.forward(views)
        else:                                               # eval.py:176-198, the two-partial_run protocol
            scores = eng.forward_phase1(views)
            g_scheme = _model.group_scheme([scores.cpu().numpy()], eng.G, eng.V)
            g_weight = _model.group_weight(g_scheme)
            _, logits = eng.forward_phase2(g_scheme, g_weight)
        lab = labels.to(device=eng.device, dtype=torch.int64).contiguous()
        self._correct.zero_()
        _lib.check(_lib.load().gv_eval_metrics(logits.data_ptr(), lab.data_ptr(), eng.N, self.C, self._pred.data_ptr(),
                                               self.confusion.data_ptr(), self._correct.data_ptr(), _model._st()),
                   "gv_eval_metrics")
        acc = float(self._correct.item()) / valid
        self.batch_accuracies.append(acc)
        self.num_shapes += valid
        return acc

    def result(self):
        """(mean of the per-batch accuracies — eval.py:213, confusion matrix [C,C] int32 on the host, #shapes)."""
        acc = sum(self.batch_accuracies) / max(len(self.batch_accuracies), 1)
        return acc, self.confusion.cpu().numpy(), self.num_shapes
