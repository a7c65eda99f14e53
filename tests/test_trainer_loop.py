"""CPU: control flow of the training loop under gradient accumulation and the HF-readable `checkpoint-<step>/` layout.

The loop is exercised with the REAL `DeSTA25Trainer.train / _train_pass / _accumulating_step` on a host-only stand-in for the
model (a 16-float gradient arena; micro-batch k writes 2**k into it), so every optimizer step's gradient names exactly the
micro-batches that fed it.  HF semantics being matched (TF:trainer.py:1715-1813, 2356-2360): ceil(len(dataloader) / GA) update
steps per epoch, a shorter LAST window per epoch, no window across an epoch boundary, evaluation once per due step."""
import json
import os
import types

import pytest
import torch


class _Arena:
    def __init__(self):
        self.grads = torch.zeros(16)
        self.params = torch.zeros(16)


class _Model:
    device = torch.device("cpu")
    training = True

    def __init__(self):
        self.arena = _Arena()
        self._fwd_count, self.dropout_seed, self._weights_dirty = 0, 0, False
        self.cur = None

    def train(self, mode=True):
        self.training = mode
        return self

    def backward(self):
        self.arena.grads.zero_()
        self.arena.grads[0] = float(2 ** self.cur["id"])

    def drop_prefetched(self):
        pass


class _Stream:
    """train_dataset with `.batches(epoch)`: micro-batch ids are consecutive over epochs."""

    def __init__(self, per_epoch):
        self.per_epoch = per_epoch

    def batches(self, epoch):
        for i in range(self.per_epoch):
            yield {"id": epoch * self.per_epoch + i, "epoch": epoch}


def _trainer(monkeypatch, ga, per_epoch=None, **kw):
    from desta import _hip as H
    from desta.trainer import desta_trainer as T
    monkeypatch.setattr(H, "add_f32", lambda acc, g: acc.add_(g))
    tr = T.DeSTA25Trainer.__new__(T.DeSTA25Trainer)
    tr.model = _Model()
    tr.args = T.TrainingArguments(gradient_accumulation_steps=ga, overlap_comm=False, logging_steps=10 ** 9,
                                  steps_per_epoch=per_epoch, **kw)
    tr.cfg, tr.train_dataset, tr.eval_dataset, tr.data_collator, tr.processing_class = None, None, None, None, None
    if per_epoch is not None:
        tr.train_dataset = _Stream(per_epoch)
    tr._micro, tr._acc, tr.global_step, tr._total_steps = 0, None, 0, None
    tr.world, tr.rank, tr._side, tr._enc_stream, tr._side_done = 1, 0, None, None, None
    tr._log_buffer, tr.log_history, tr.prediction_step_outputs = [], [], []
    tr.updates, tr.evals, tr.saves = [], [], []

    def compute_loss(self, model, inputs, return_outputs=False, **kwargs):
        model.cur = inputs
        return torch.tensor(float(inputs["id"]))
    tr.compute_loss = types.MethodType(compute_loss, tr)
    tr._reduce_and_update = types.MethodType(lambda self, lr: self.updates.append(int(self.model.arena.grads[0])), tr)
    tr._sync = types.MethodType(lambda self: None, tr)
    tr._can_evaluate = types.MethodType(lambda self: True, tr)
    tr.evaluate = types.MethodType(lambda self, *a, **k: self.evals.append((self.global_step, self._micro)), tr)
    tr.save_checkpoint = types.MethodType(lambda self, d: self.saves.append(os.path.basename(d)), tr)
    return tr


def _ids(mask):
    return [k for k in range(64) if mask >> k & 1]


def test_ga2_two_epochs_of_six_micro_batches(monkeypatch):
    tr = _trainer(monkeypatch, ga=2, per_epoch=6, num_train_epochs=2, max_steps=-1, save_strategy="epoch", eval_strategy="epoch")
    assert tr.steps_per_epoch() == 3 and tr.total_steps == 6
    losses = tr.train()
    assert len(losses) == 12 and tr.global_step == 6
    assert [_ids(u) for u in tr.updates] == [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 11]]
    assert tr.saves == ["checkpoint-3", "checkpoint-6"] and [e[0] for e in tr.evals] == [3, 6]   # exactly two epoch ends


def test_ga2_max_steps_closes_its_last_window(monkeypatch):
    tr = _trainer(monkeypatch, ga=2, per_epoch=6, num_train_epochs=5, max_steps=4)
    losses = tr.train()
    assert tr.global_step == 4 and len(losses) == 8 and tr._micro == 0       # no dangling half window
    assert [_ids(u) for u in tr.updates] == [[0, 1], [2, 3], [4, 5], [6, 7]]  # step 4 = first window of epoch 1


def test_ga2_over_a_plain_iterable_with_max_steps(monkeypatch):
    tr = _trainer(monkeypatch, ga=2, max_steps=100)
    data = [{"id": i} for i in range(10)]
    losses = tr.train(data, max_steps=3)
    assert tr.global_step == 3 and len(losses) == 6 and tr._micro == 0
    assert [_ids(u) for u in tr.updates] == [[0, 1], [2, 3], [4, 5]]


def test_remainder_window_at_the_end_of_every_epoch(monkeypatch):
    # 5 micro-batches per epoch, GA 2 -> ceil(5 / 2) = 3 update steps per epoch, the third from ONE micro-batch (HF `remainder`)
    tr = _trainer(monkeypatch, ga=2, per_epoch=5, num_train_epochs=2, max_steps=-1)
    assert tr.steps_per_epoch() == 3 and tr.total_steps == 6
    tr.train()
    assert [_ids(u) for u in tr.updates] == [[0, 1], [2, 3], [4], [5, 6], [7, 8], [9]]


def test_eval_steps_fire_once_per_due_optimizer_step(monkeypatch):
    tr = _trainer(monkeypatch, ga=3, per_epoch=12, num_train_epochs=1, max_steps=-1, eval_strategy="steps", eval_steps=2)
    tr.train()
    assert tr.global_step == 4
    assert tr.evals == [(2, 0), (4, 0)]                                       # never at step 0, never mid-window, once each


def test_resume_mid_epoch_skips_whole_windows(monkeypatch):
    tr = _trainer(monkeypatch, ga=2, per_epoch=6, num_train_epochs=2, max_steps=-1)
    tr.global_step = 4                                                        # as restored from checkpoint-4: epoch 1, one window done
    tr.train()
    assert [_ids(u) for u in tr.updates] == [[8, 9], [10, 11]] and tr.global_step == 6


def test_steps_per_epoch_counts_optimizer_steps():
    from desta.trainer.desta_trainer import TrainingArguments, micro_batches_per_epoch, steps_per_epoch
    a = TrainingArguments(per_device_train_batch_size=8, gradient_accumulation_steps=4)
    assert micro_batches_per_epoch(a, 1000, 8) == 16 and steps_per_epoch(a, 1000, 8) == 4
    a = TrainingArguments(per_device_train_batch_size=8, gradient_accumulation_steps=3)
    assert steps_per_epoch(a, 1000, 8) == 6                                   # 16 micro-batches: 5 full windows + a remainder of 1
    assert steps_per_epoch(TrainingArguments(steps_per_epoch=32, gradient_accumulation_steps=2), None, 1) == 16


def test_checkpoint_directory_is_readable_by_hf_trainer_state(tmp_path, monkeypatch):
    """`trainer_state.json` holds TrainerState fields only (TF:trainer_callback.py `load_from_json` = cls(**json)); the library's
    own resume data sits in a sidecar; rng_state.pth and training_args.bin exist (TF:trainer.py `_save_checkpoint`)."""
    from transformers.trainer_callback import TrainerState
    from desta.trainer import desta_trainer as T
    tr = _trainer(monkeypatch, ga=1, per_epoch=6, num_train_epochs=2, max_steps=-1, eval_steps=3)
    tr.global_step, tr.log_history = 9, [{"train/loss": 1.5, "train/learning_rate": 1e-4}]
    tr.model._fwd_count, tr.model.dropout_seed = 9, 1
    tr.model.save_pretrained = lambda d: open(os.path.join(d, "model.safetensors"), "wb").close()
    tr.optimizer = types.SimpleNamespace(hf_state_dict=lambda names, lr, wd: {"state": {}, "param_groups": []})
    tr.model.config = types.SimpleNamespace(target_layer_ids=[0], qformer_num_hidden_layers=0)
    ck = tmp_path / "checkpoint-9"
    T.DeSTA25Trainer.save_checkpoint(tr, str(ck))
    names = set(os.listdir(ck))
    assert {"trainer_state.json", "desta_hip_state.json", "rng_state.pth", "training_args.bin", "optimizer.pt", "scheduler.pt"} <= names
    st = TrainerState.load_from_json(str(ck / "trainer_state.json"))
    assert st.global_step == 9 and st.max_steps == 12 and abs(st.epoch - 1.5) < 1e-12 and st.num_train_epochs == 2
    assert st.log_history == tr.log_history and st.train_batch_size == 8
    st.save_to_json(str(tmp_path / "roundtrip.json"))                         # and HF can write it back unchanged
    assert json.load(open(tmp_path / "roundtrip.json")) == json.load(open(ck / "trainer_state.json"))
    assert "desta_hip" not in json.load(open(ck / "trainer_state.json"))
    assert json.load(open(ck / "desta_hip_state.json"))["forward_count"] == 9
    rng = torch.load(ck / "rng_state.pth", weights_only=False)
    assert {"python", "numpy", "cpu"} <= set(rng)
    args = torch.load(ck / "training_args.bin", weights_only=True)
    assert args["learning_rate"] == tr.args.learning_rate
    sched = torch.load(ck / "scheduler.pt", weights_only=True)
    assert sched["last_epoch"] == 9
