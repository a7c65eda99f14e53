"""CPU: host logic of the trainer that needs no device — schedule length and LR for epochs-only configs pinned against
transformers' own scheduler (the shipped full-size YAMLs use max_steps: -1 with max_epochs: 5)."""
import math

import pytest
import torch


def test_epochs_only_schedule_matches_hf_linear_schedule():
    from transformers.optimization import get_linear_schedule_with_warmup
    from desta.optim import linear_warmup_lr
    from desta.trainer.desta_trainer import TrainingArguments, resolve_total_steps, steps_per_epoch
    # 1000 samples, 8 ranks, per-device batch 8 -> ceil(125 / 8) = 16 steps per epoch, 5 epochs -> 80 steps
    args = TrainingArguments(learning_rate=1e-4, warmup_steps=10, max_steps=-1, num_train_epochs=5, per_device_train_batch_size=8)
    spe = steps_per_epoch(args, 1000, 8)
    assert spe == 16
    total = resolve_total_steps(args, spe)
    assert total == 80
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-4)
    sched = get_linear_schedule_with_warmup(opt, 10, total)
    for step in range(total + 3):
        assert abs(linear_warmup_lr(step, 1e-4, 10, total) - sched.get_last_lr()[0]) < 1e-15, step
        opt.step()
        sched.step()
    assert linear_warmup_lr(total, 1e-4, 10, total) == 0.0                    # decays to zero, not constant after warm-up
    # max_steps > 0 takes precedence over epochs (YAML comment "precedence over max_epochs")
    assert resolve_total_steps(TrainingArguments(max_steps=7, num_train_epochs=5), 16) == 7
    # fractional epochs round up; an unsized stream needs steps_per_epoch or max_steps (HF's own error)
    assert resolve_total_steps(TrainingArguments(max_steps=-1, num_train_epochs=1.5), 5) == 8
    assert steps_per_epoch(TrainingArguments(steps_per_epoch=12), None, 4) == 12
    with pytest.raises(ValueError, match="max_steps must be set to a positive value"):
        resolve_total_steps(TrainingArguments(max_steps=-1), None)
