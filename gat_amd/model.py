"""The reference's network and training loop (/root/reference/gat/model.py:18-133) on
gat_amd.GATConv: same constructor, forward(data), train_model(train, val) -> Metrics and
test_model(data).  The three layer calls are the HIP hot path; ELU / BatchNorm / dropout /
the optimizer stay stock PyTorch-ROCm ops; ELU/BatchNorm/dropout and log_softmax/nll_loss
run fused (gat_amd/glue.py; `fused_glue = False` restores the stock ops).

Differences a caller can see, all forced by the move to a GPU or to torch 2.10:
  * the model lives on the GPU (`device`, default cuda:LOCAL_RANK) and moves `data` there
    once (cached), the reference has no device at all;
  * ReduceLROnPlateau is built without `verbose=True` (removed in torch 2.10); the learning
    rate is still printed every epoch as at model.py:112;
  * precision / recall / F1 come from an on-device confusion matrix (one tiny copy per
    epoch) with scikit-learn's weighted-average arithmetic, instead of three sklearn calls
    on .cpu() tensors per epoch (model.py:61-63,76-79);
  * the checkpoint file name is an attribute (`checkpoint_path`, default "best_model.pth"
    as at model.py:118) so concurrent trials do not collide;
  * `hipgraph=True` (default off): from its second epoch on, the training step (forward, loss,
    backward) and the validation forward are each ONE captured HIP graph that is replayed -- on
    small graphs (the reference's own data, Cora-shaped) an epoch is some sixty kernel launches
    whose host cost exceeds their run time.  The dropout seeds then live on the device
    (GATMI_MASK_HASH_DEV), advanced inside the graph, so every replay draws new masks.
"""
from __future__ import annotations

import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from .data import Data, Metrics
from .glue import elu_bn_dropout, log_softmax_nll
from .layer import GATConv


def default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("gat_amd.model.GAT needs a GPU: the GAT layer has no CPU path")
    n = max(torch.cuda.device_count(), 1)
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % n)


def confusion_counts(y_true: torch.Tensor, y_pred: torch.Tensor, num_classes: int) -> torch.Tensor:
    """[K, K] int64, rows = true class, columns = predicted (sklearn.confusion_matrix)."""
    flat = y_true.to(torch.int64) * num_classes + y_pred.to(torch.int64)
    return torch.bincount(flat, minlength=num_classes * num_classes).view(num_classes, num_classes)


def weighted_prf(cm: torch.Tensor):
    """precision, recall, F1 with average="weighted" and zero_division -> 0, the arithmetic
    of sklearn.metrics.precision_score & co. as called at model.py:61-63."""
    cm = cm.to(torch.float64)
    tp = cm.diag()
    support, predicted = cm.sum(1), cm.sum(0)
    total = support.sum().clamp(min=1)
    prec = torch.where(predicted > 0, tp / predicted.clamp(min=1), torch.zeros_like(tp))
    rec = torch.where(support > 0, tp / support.clamp(min=1), torch.zeros_like(tp))
    den = prec + rec
    f1 = torch.where(den > 0, 2 * prec * rec / den.clamp(min=1e-300), torch.zeros_like(tp))
    w = support / total
    return float((prec * w).sum()), float((rec * w).sum()), float((f1 * w).sum())


def print_epoch_stats(epoch, seconds, tr, va, cm):
    """Same lines as /root/reference/gat/analysis.py:112-122."""
    print(f"Epoch {epoch}, Time: {seconds:.2f}s")
    print(f"Train Loss: {tr[0]:.4f}, Train Accuracy: {tr[1]:.4f}")
    print(f"Val Loss: {va[0]:.4f}, Val Accuracy: {va[1]:.4f}")
    print(f"Train Precision: {tr[2]:.4f}, Train Recall: {tr[3]:.4f}, Train F1-Score: {tr[4]:.4f}")
    print(f"Validation Precision: {va[2]:.4f}, Validation Recall: {va[3]:.4f}, "
          f"Validation F1-Score: {va[4]:.4f}")
    print(f"Confusion Matrix:\n{cm}")
    print("--------------------------------------------------")


class GAT(nn.Module):
    def __init__(self, optimizer, num_features, num_classes, weight_decay=1e-3, dropout=0.7,
                 hidden_dim=16, epochs=30, lr=0.005, patience=3, device=None, verbose=True,
                 feature_dtype=None, hipgraph=False):
        super().__init__()
        self.epochs, self.patience, self.num_classes = epochs, patience, num_classes
        self.verbose = verbose
        self.checkpoint_path = "best_model.pth"
        # elu -> bn -> dropout between the layers as one fused op (gat_amd/glue.py) instead of
        # three stock ones; same modules, same state_dict
        self.fused_glue = True
        # feature_dtype=torch.bfloat16 (not in the reference; default None = f32 everywhere): the
        # layers on the projected-row path whose shape the bf16 kernels cover keep their projected
        # rows in bf16 (gat_amd/layer.py); with the reference's widths that is conv2 when
        # hidden_dim is a multiple of 8.  The parity claims of this package are for the default.
        kw = dict(dropout=dropout, feature_dtype=feature_dtype)
        self.conv1 = GATConv(num_features, hidden_dim, heads=8, **kw)
        self.bn1 = nn.BatchNorm1d(hidden_dim * 8)
        self.conv2 = GATConv(hidden_dim * 8, hidden_dim, heads=8, **kw)
        self.bn2 = nn.BatchNorm1d(hidden_dim * 8)
        self.conv3 = GATConv(hidden_dim * 8, num_classes, heads=1, concat=True, **kw)
        self.dropout = nn.Dropout(p=dropout)
        self.device = torch.device(device) if device is not None else default_device()
        self.to(self.device)
        self.optimizer = optimizer(self.parameters(), lr=lr, weight_decay=weight_decay)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(
            self.optimizer, mode="min", factor=0.1, patience=self.patience)
        self._resident = {}
        self.hipgraph = bool(hipgraph)
        self._graphs, self._eager_done = {}, set()
        self._glue_seeds = [None, None]
        if self.hipgraph:
            base = int(torch.empty((), dtype=torch.int64).random_().item())
            mk = lambda i: torch.tensor([(base + 0x632BE59BD9B4E019 * i) % (1 << 63)], dtype=torch.int64,
                                        device=self.device)
            self.conv1.device_seed, self.conv2.device_seed, self.conv3.device_seed = mk(0), mk(1), mk(2)
            self._glue_seeds = [mk(3), mk(4)]

    # -- data residency: move each Data object to the GPU once
    def _on_device(self, data) -> Data:
        if data.x.device == self.device:
            return data
        hit = self._resident.get(id(data))
        if hit is None or hit[0] is not data:
            hit = (data, Data(data.x, data.edge_index, data.y).to(self.device))
            self._resident[id(data)] = hit
        return hit[1]

    def _logits(self, data):
        x, edge_index = data.x, data.edge_index
        x = self.dropout(x)
        if self.fused_glue:
            p = self.dropout.p
            x = elu_bn_dropout(self.conv1(x, edge_index), self.bn1, p, self.training, self._glue_seeds[0])
            x = elu_bn_dropout(self.conv2(x, edge_index), self.bn2, p, self.training, self._glue_seeds[1])
        else:
            x = self.bn1(F.elu(self.conv1(x, edge_index)))
            x = self.dropout(x)
            x = self.bn2(F.elu(self.conv2(x, edge_index)))
            x = self.dropout(x)
        return self.conv3(x, edge_index)

    def forward(self, data):
        return F.log_softmax(self._logits(self._on_device(data)), dim=1)

    def _out_and_loss(self, data):
        """(log-probabilities, mean nll loss): model.py:52-53, fused when `fused_glue`."""
        if self.fused_glue:
            return log_softmax_nll(self._logits(data), data.y)
        out = self(data)
        return out, F.nll_loss(out, data.y)

    def _scores(self, out, y, loss):
        preds = out.argmax(dim=1)
        k = max(self.num_classes, int(out.size(1)))
        # confusion matrix and loss cross to the host in ONE copy (one synchronisation per call;
        # the counts are exact in float64)
        packed = torch.cat([confusion_counts(y, preds, k).flatten().to(torch.float64),
                            loss.detach().to(torch.float64).view(1)]).cpu()
        cm = packed[:-1].to(torch.int64).view(k, k)
        accuracy = float(cm.diag().sum()) / max(int(y.size(0)), 1)
        return (float(packed[-1]), accuracy) + weighted_prf(cm), cm

    # -- HIP-graph replay of a step (hipgraph=True)
    def _captured_step(self, kind: str, data):
        """(out, loss) of one training step (forward + loss + backward; parameter .grad filled) or
        one validation forward.  The first call of each kind on a Data object runs eagerly (it builds
        the graph layout, which synchronises, and sizes every workspace); the second captures the
        same calls into a HIP graph; from then on the graph is replayed.  Outputs and gradients
        live in the graph's memory pool and are overwritten by every replay."""
        key = (kind, id(data))
        ent = self._graphs.get(key)
        if ent is None or ent["data"] is not data:
            if key not in self._eager_done:
                self._eager_done.add(key)
                return self._eager_step(kind, data)
            torch.cuda.synchronize(self.device)
            if kind == "train":
                self.optimizer.zero_grad(set_to_none=True)    # .grad is created inside the capture
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out, loss = self._eager_step(kind, data, zero=False)
            ent = self._graphs[key] = dict(data=data, graph=graph, out=out, loss=loss)
            if kind == "train":     # the gradient tensors THIS graph writes (they live in its pool)
                ent["grads"] = [p.grad for p in self.parameters()]
        ent["graph"].replay()
        if kind == "train":
            # Every captured training graph fills its own gradient tensors.  An eager step or the
            # capture of another train set in between re-binds p.grad elsewhere; the optimizer must
            # read what this replay has just written, not the other graph's stale values.
            for p, g in zip(self.parameters(), ent["grads"]):
                p.grad = g
        return ent["out"], ent["loss"]

    def _eager_step(self, kind: str, data, zero: bool = True):
        if kind == "train":
            if zero:
                self.optimizer.zero_grad()
            out, loss = self._out_and_loss(data)
            loss.backward()
            return out.detach(), loss.detach()
        with torch.no_grad():
            return self._out_and_loss(data)

    def train_epoch(self, data):
        data = self._on_device(data)
        self.train()
        if self.hipgraph:
            out, loss = self._captured_step("train", data)
        else:
            out, loss = self._eager_step("train", data)
        self.optimizer.step()
        stats, _ = self._scores(out, data.y, loss)
        return stats

    def validate_epoch(self, data):
        data = self._on_device(data)
        self.eval()
        if self.hipgraph:
            out, loss = self._captured_step("eval", data)
        else:
            out, loss = self._eager_step("eval", data)
        stats, cm = self._scores(out, data.y, loss)
        return stats + (cm.numpy(),)

    def train_model(self, train_data, val_data) -> Metrics:
        metrics = Metrics()
        best, stale = float("inf"), 0
        for epoch in range(1, self.epochs + 1):
            t0 = time.time()
            tr = self.train_epoch(train_data)
            *va, cm = self.validate_epoch(val_data)
            for name, value in zip(("loss", "accuracy", "precision", "recall", "f1"), tr):
                getattr(metrics, "train_" + name).append(value)
            for name, value in zip(("loss", "accuracy", "precision", "recall", "f1"), va):
                getattr(metrics, "val_" + name).append(value)
            metrics.epoch_values.append(epoch)
            if self.verbose:
                print_epoch_stats(epoch, time.time() - t0, tr, va, cm)
            self.scheduler.step(va[0])
            if self.verbose:
                print(f"Epoch {epoch}, Current Learning Rate: "
                      f'{self.scheduler.optimizer.param_groups[0]["lr"]}')
            if va[0] < best:
                best, stale = va[0], 0
                torch.save(self.state_dict(), self.checkpoint_path)
            else:
                stale += 1
                if stale >= self.patience:
                    if self.verbose:
                        print("Early stopping")
                    break
        if os.path.exists(self.checkpoint_path):
            self.load_state_dict(torch.load(self.checkpoint_path, map_location=self.device))
        plot_metrics(metrics)
        return metrics

    def test_model(self, data):
        self.eval()
        return self(data).argmax(dim=1)


def plot_metrics(metrics: Metrics, path: str = "./results/accuracy_over_epochs.png") -> None:
    """Accuracy-over-epoch plot of /root/reference/gat/analysis.py:124-157; skipped quietly
    when matplotlib or the results directory is missing (reporting is out of the hot path)."""
    if not os.path.isdir(os.path.dirname(path)):
        return
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:  # noqa: BLE001
        return
    _, ax = plt.subplots(figsize=(12, 8))
    ax.plot(metrics.epoch_values, metrics.train_accuracy, "o-r", label="Train Accuracy")
    ax.plot(metrics.epoch_values, metrics.val_accuracy, "s--r", label="Val Accuracy")
    ax.set_xlabel("Epoch"), ax.set_ylabel("Metrics"), ax.set_title("Accuracy over Epoch")
    ax.grid(True, linestyle="--", linewidth=0.7), ax.legend()
    plt.tight_layout(), plt.savefig(path), plt.close()
