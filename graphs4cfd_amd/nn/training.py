"""Training driver behind `GNN.fit` (what the reference does inside nn/model.py:152-301).

The contract with the reference is the behaviour, not the loop: Adam (+ optional ReduceLROnPlateau), one optimiser step per rollout step
with the previous prediction fed back detached, gradient clipping after a given epoch, validation over the longest rollout length,
a curriculum of rollout lengths that advances (with a fresh optimiser) when the monitored loss falls below a tolerance, and `.chk`
files the reference can read (`GNN.save_checkpoint`).  The pieces:

    RolloutCurriculum   the `num_steps` list as a cursor (current length, longest length, advance, fast-forward on resume)
    OptimiserFactory    Adam in its single-launch form on the GPU + the plateau scheduler of the config
    CheckpointFile      where the run's `.chk` goes; an older file of the same name is kept as `.bck`
    Trainer             epochs: train pass, range check of the fp16 arithmetic, validation pass, bookkeeping
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
from torch import nn, optim

from .. import ops


class RolloutCurriculum:
    """The rollout lengths of `TrainConfig.num_steps`, in the order they are trained."""

    def __init__(self, lengths: Iterable[int]):
        self.lengths: List[int] = [int(n) for n in lengths]
        if not self.lengths:
            raise ValueError("TrainConfig.num_steps is empty")
        self.stage = 0

    @property
    def n_out(self) -> int:
        return self.lengths[self.stage]

    @property
    def longest(self) -> int:
        return self.lengths[-1]

    @property
    def finished(self) -> bool:
        return self.stage == len(self.lengths) - 1

    def advance(self) -> bool:
        if self.finished:
            return False
        self.stage += 1
        return True

    def fast_forward(self, n_out: int) -> None:
        """Resume: skip the stages shorter than the rollout length the checkpoint was written at."""
        while self.n_out < n_out and self.advance():
            pass


class OptimiserFactory:
    def __init__(self, model: nn.Module, scheduler_cfg: Optional[dict]):
        self.model, self.cfg = model, scheduler_cfg

    def adam(self, lr: float) -> optim.Optimizer:
        """torch.optim.Adam; on the GPU its `fused` form (one launch for the 306 parameter tensors of a 3-scale model: 0.47 instead of
        1.35 ms per step, same update rule)."""
        params = list(self.model.parameters())
        if params and params[0].is_cuda:
            try:
                return optim.Adam(params, lr=lr, fused=True)
            except (TypeError, RuntimeError):
                pass
        return optim.Adam(params, lr=lr)

    def plateau(self, optimiser: optim.Optimizer):
        if not self.cfg or self.cfg.get('patience') is None:
            return None
        return optim.lr_scheduler.ReduceLROnPlateau(optimiser, factor=self.cfg['factor'], patience=self.cfg['patience'], eps=0.)


class CheckpointFile:
    def __init__(self, folder: str, name: str):
        self.path = os.path.join(folder, name + ".chk")
        if os.path.exists(self.path):
            backup = self.path + ".bck"
            print(f"[fit] {self.path} exists: kept as {backup}")
            os.rename(self.path, backup)

    def write(self, model, n_out: int, epoch: int, optimiser, scheduler) -> None:
        model.save_checkpoint(self.path, n_out, epoch, optimiser, scheduler=scheduler)


class Trainer:
    def __init__(self, model, cfg, train_loader, val_loader=None):
        self.model, self.cfg, self.train_loader, self.val_loader = model, cfg, train_loader, val_loader
        if cfg['device'] is not None and torch.device(cfg['device']) != model.device:
            model.to(cfg['device'])
        self.curriculum = RolloutCurriculum(cfg['num_steps'])
        self.factory = OptimiserFactory(model, cfg['scheduler'])
        self.first_epoch = 1
        resume = cfg['checkpoint']
        if resume is not None and os.path.exists(resume):
            self._resume(resume)
        else:
            if resume is not None:
                print(f"[fit] no checkpoint at {resume}: starting from the model's current weights")
            self.optimiser = self.factory.adam(cfg['lr'])
            self.scheduler = self.factory.plateau(self.optimiser)
        self.file = CheckpointFile(cfg["folder"], cfg["name"])
        self.board = None
        if cfg['tensor_board'] is not None:
            from torch.utils.tensorboard import SummaryWriter
            self.board = SummaryWriter(os.path.join(cfg["tensor_board"], cfg["name"]))

    def _resume(self, path: str) -> None:
        m = self.model
        state = torch.load(path, map_location=m.device, weights_only=False)
        print(f"[fit] resuming from {path} (epoch {state['epoch']}, rollout length {state['n_out']})")
        m.load_state_dict(state['weights'])
        self.optimiser = self.factory.adam(state['lr'])
        self.optimiser.load_state_dict(state['optimiser'])
        self.scheduler = self.factory.plateau(self.optimiser)
        if self.scheduler is not None and 'scheduler' in state:
            self.scheduler.load_state_dict(state['scheduler'])
        self.curriculum.fast_forward(state['n_out'])
        self.first_epoch = state['epoch'] + 1

    @property
    def lr(self) -> float:
        return self.optimiser.param_groups[0]['lr']

    # ------------------------------------------------------------------ one pass over a loader
    def _targets(self, data, t: int):
        nf = self.model.num_fields
        return data.target[:, nf * t: nf * (t + 1)]

    def train_epoch(self, epoch: int):
        m, cfg, n_out = self.model, self.cfg, self.curriculum.n_out
        loss_fn, clip = cfg['training_loss'], cfg['grad_clip']
        clipping = clip is not None and epoch > clip["epoch"]
        m.train()
        loss_sum, norm_sum, batches = 0., 0., 0
        for data in self.train_loader:
            data = data.to(m.device)
            pred = None
            for t in range(n_out):                      # one optimiser step per rollout step, the prediction fed back detached
                if t:
                    data.field = m.shift_and_replace(data.field, pred.detach())
                pred = m.forward(data, t)
                loss = loss_fn(data, pred, self._targets(data, t))
                loss.backward()
                loss_sum += loss.item() / n_out
                norm_sum += m.grad_norm2() / n_out
                if clipping:
                    nn.utils.clip_grad_norm_(m.parameters(), clip["limit"])
                self.optimiser.step()
                self.optimiser.zero_grad()
                m.invalidate_packed()                   # the step changed the weights, whatever path the gradients came from
            batches += 1
            # (the plan caches are LRU- and byte-bounded (plan._Cache, G4C_PLAN_CACHE_MB): a loader whose graphs recur keeps its
            # plans, a stream of fresh batches evicts the oldest — no global clear per iteration)
        return loss_sum / max(batches, 1), norm_sum / max(batches, 1)

    def validate(self) -> Optional[float]:
        if self.val_loader is None:
            return None
        m, loss_fn, n_out = self.model, self.cfg['validation_loss'], self.curriculum.longest
        m.eval()
        total, batches = 0., 0
        with torch.no_grad():
            for data in self.val_loader:
                data = data.to(m.device)
                pred = None
                for t in range(n_out):
                    if t:
                        data.field = m.shift_and_replace(data.field, pred)
                    pred = m.forward(data, t)
                    total += loss_fn(data, pred, self._targets(data, t)).item() / n_out
                batches += 1
        return total / max(batches, 1)

    @staticmethod
    def _pick(which: str, train_loss: float, val_loss: Optional[float]) -> float:
        """The loss a config entry monitors: 'training...' or 'validation...' (the reference's selectors)."""
        if which[:2] == 'tr':
            return train_loss
        if which[:3] == 'val':
            if val_loss is None:
                raise ValueError("a validation loss is monitored but no val_loader was given")
            return val_loss
        raise NameError(f"Invalid loss selector {which!r} (expected 'training' or 'validation').")

    # ------------------------------------------------------------------ the run
    def run(self) -> None:
        m, cfg, cur = self.model, self.cfg, self.curriculum
        if cfg['mixed_precision']:
            print(f"[fit] mixed_precision: MLP products run in ops.mlp_precision() = {ops.mlp_precision()!r}; gradients stay fp32, no loss scaling")
        print(f"[fit] device {m.device}, {m.num_params} trainable parameters, rollout lengths {cur.lengths}")
        m.history = []
        for epoch in range(self.first_epoch, cfg['epochs'] + 1):
            if self.lr < cfg['stopping']:
                print(f"[fit] learning rate {self.lr:g} fell below the stopping value {cfg['stopping']:g}: done")
                self.file.write(m, cur.n_out, epoch, self.optimiser, self.scheduler)
                break
            print(f"[fit] epoch {epoch}: rollout length {cur.n_out}, lr {self.lr:g}")
            train_loss, grad_norm = self.train_epoch(epoch)
            print(f"[fit] epoch {epoch}: training loss {train_loss:.4e}, gradient norm {grad_norm:.4e}")
            if ops.mlp_precision() == "f16x3":
                ops.check_f16_range(m.device, f"fit(), epoch {epoch}")
            val_loss = self.validate()
            if val_loss is not None:
                print(f"[fit] epoch {epoch}: validation loss {val_loss:.4e} (rollout length {cur.longest})")
            m.history.append({'epoch': epoch, 'n_out': cur.n_out, 'training_loss': train_loss, 'validation_loss': val_loss,
                              'gradients_norm': grad_norm, 'lr': self.lr})
            if self.board is not None:
                self.board.add_scalar('Loss/train', train_loss, epoch)
                if val_loss is not None:
                    self.board.add_scalar('Loss/test', val_loss, epoch)
            if self.scheduler is not None:
                self.scheduler.step(self._pick(cfg['scheduler']['loss'], train_loss, val_loss))
            if epoch % cfg["chk_interval"] == 0:
                self.file.write(m, cur.n_out, epoch, self.optimiser, self.scheduler)
                print(f"[fit] epoch {epoch}: checkpoint written to {self.file.path}")
            grow = cfg['add_steps']
            if self._pick(grow['loss'], train_loss, val_loss) < grow['tolerance'] and cur.advance():
                # a longer rollout is a new problem for the optimiser: fresh moments, the configured learning rate again
                self.optimiser = self.factory.adam(cfg["lr"])
                self.scheduler = self.factory.plateau(self.optimiser)
        if self.board is not None:
            self.board.close()
        print("[fit] done")
