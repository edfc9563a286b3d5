"""Trainer base: epoch loop, Adam, and the static loss helpers, on the HIP kernels.

Public surface of the reference's utils/trainer.py:16-413 (constructor,
train_model, loss_and_acc_on_epoch, cuda/zero_grad/step, load_model, the abstract
hooks and the static loss helpers with their names and argument meaning).
Differences a caller can observe:
  * the optimizer is arvae_amd.optim.FlatAdam (same update rule, one kernel);
  * epoch means are accumulated on the device (no per-step D2H sync);
  * logging tolerates writer=None (the reference crashes in epoch 2 with --no_log).
"""
import datetime
import os
import time
from abc import ABC, abstractmethod

import torch

from . import ops
from .optim import FlatAdam


def _is_standard_prior(prior_dist):
    return prior_dist is None or getattr(prior_dist, '_arvae_standard', False)


class Trainer(ABC):
    def __init__(self, dataset, model, lr=1e-4):
        self.dataset = dataset
        self.model = model
        # the order of the flat arena is the model's to choose (Model.arena_parameters: tensors that one launch reads as ONE
        # matrix sit side by side -- the two directions' input projections of MeasureVAE's encoder); default: parameters()
        order = self.model.arena_parameters() if hasattr(self.model, 'arena_parameters') else self.model.parameters()
        self.optimizer = FlatAdam((p for p in order if p.requires_grad), lr=lr)
        self.global_iter = 0
        self.trainer_config = ''
        self.writer = None
        self.data_parallel = None          # arvae_amd.parallel.DataParallel when world_size > 1

    # -- epoch loop (reference utils/trainer.py:39-154) ------------------------------------------------
    def train_model(self, batch_size, num_epochs, log=False):
        if log and (self.data_parallel is None or self.data_parallel.rank == 0):
            from .logging_utils import make_writer
            stamp = datetime.datetime.fromtimestamp(time.time()).strftime('%Y-%m-%d_%H:%M:%S')
            self.writer = make_writer(os.path.join('runs', repr(self.model) + stamp))
        dp = self.data_parallel
        chief = dp is None or dp.rank == 0                      # one rank logs, prints and writes checkpoints
        if dp is not None:
            # batch_size is the PER-GPU batch (weak scaling, as bench.py): every rank iterates its own rows of the same
            # shuffled global batches (data/loaders.py)
            train_loader, val_loader, _ = self.dataset.data_loaders(batch_size=batch_size, split=(0.70, 0.20),
                                                                    shard=(dp.rank, dp.world_size, dp.comm))
            dp.broadcast_parameters(self.model)
        else:
            train_loader, val_loader, _ = self.dataset.data_loaders(batch_size=batch_size, split=(0.70, 0.20))
        if chief:
            print('Num Train Batches: ', len(train_loader))
            print('Num Valid Batches: ', len(val_loader))
        for epoch in range(num_epochs):
            self.update_scheduler(epoch)
            self.model.train()
            loss_tr, acc_tr = self.loss_and_acc_on_epoch(train_loader, epoch_num=epoch, train=True)
            self.model.eval()
            with torch.no_grad():
                loss_va, acc_va = self.loss_and_acc_on_epoch(val_loader, epoch_num=epoch, train=False)
            if dp is not None:                                   # epoch means over the global batch
                loss_tr, acc_tr, loss_va, acc_va = dp.mean_stats([loss_tr, acc_tr, loss_va, acc_va],
                                                                 next(self.model.parameters()).device)
            if not chief:
                continue
            self.eval_model(data_loader=val_loader, epoch_num=epoch)
            if log and self.writer is not None:
                self.writer.add_scalar('loss/train', loss_tr, epoch)
                self.writer.add_scalar('loss/valid', loss_va, epoch)
                self.writer.add_scalar('acc/train', acc_tr, epoch)
                self.writer.add_scalar('acc/valid', acc_va, epoch)
            self.print_epoch_stats(epoch, num_epochs, loss_tr, acc_tr, loss_va, acc_va)
            self.model.save()

    # Training steps replayed from HIP graphs (ar-vae_amd/graphed.py).  Worth it where the step is many small launches and
    # the host sets the pace (MeasureVAE: 3.9 -> 1.5 ms per step); subclasses switch it on.  Batches of another shape
    # (the last one of an epoch) and CPU models take the eager path; changing beta / gamma / delta / reg_dim (e.g. from
    # update_scheduler) re-captures.  Data-parallel steps replay too when their collectives are the library's (recorded by
    # the capture like any launch, parallel.LibraryComm); over torch.distributed (parallel.TorchComm) they stay eager.
    use_graph_replay = False

    def _replay_step(self, batch):
        """-> (loss, accuracy) of zero_grad + loss + backward replayed from a captured graph, or None (run it eagerly)."""
        if not self.use_graph_replay or not torch.cuda.is_available():
            return None
        if not next(self.model.parameters()).is_cuda:
            return None
        if self.data_parallel is not None and not self.data_parallel.capturable:
            return None
        from .graphed import GraphedStep
        graphed = getattr(self, '_graphed', None)
        if graphed is not None and graphed.hyper != GraphedStep.hyper_of(self):
            graphed = self._graphed = None                     # beta / gamma / delta / reg_dim are baked into the captured kernels
        if graphed is None:
            self.model.train()
            try:
                graphed = self._graphed = GraphedStep(self, batch)
            except RuntimeError as e:                          # capture not possible here: stay eager from now on (under data
                print(f'graph replay disabled: {e}')           # parallelism GraphedStep raises on EVERY rank or on none)
                self.use_graph_replay = False
                return None
        data = graphed.accepts(batch)                          # processed once; None = a batch of another shape (the last one
        if data is None:                                       # of an epoch): eager
            return None
        return graphed(data=data)                              # errors of the step itself propagate

    def loss_and_acc_on_epoch(self, data_loader, epoch_num=None, train=True):
        """-> (mean loss, mean accuracy) of one pass over `data_loader` (utils/trainer.py:114-154).  A pass that reported a
        device-side failure (DeviceStatusError: an in-launch hand-off gave up, csrc/midcluster.hip) is run AGAIN, once: the
        update kernel withheld every step from the failing one on (optim.py "Guard slot"), so the weights, the moments and
        the step count are those of the last good step, and the trainer has left the kernels that can fail.  What the
        repeated epoch costs is the good steps of the first attempt applied once more."""
        from .fused import DeviceStatusError
        try:
            return self._epoch(data_loader, epoch_num, train)
        except DeviceStatusError as e:
            print(f'epoch {epoch_num}: {e}\n  -> repeating the epoch on the row kernels', flush=True)
            self._graphed = None                                 # a captured step has the failed kernels baked in
        return self._epoch(data_loader, epoch_num, train)

    def _epoch(self, data_loader, epoch_num, train):
        loss_sum = acc_sum = None
        count = 0
        for batch_num, batch in enumerate(data_loader):
            replayed = self._replay_step(batch) if train else None
            if replayed is not None:
                loss, accuracy = replayed
                self.step()
                if batch_num == 0 and hasattr(self, 'log_loss_split'):
                    self.log_loss_split(epoch_num)
            else:
                batch_data = self.process_batch_data(batch)
                self.zero_grad()
                loss, accuracy = self.loss_and_acc_for_batch(batch_data, epoch_num, batch_num, train=train)
                if train:
                    self.backward(loss)
                    self.step()
            # the accumulators start from COPIES: under graph replay `loss` / `accuracy` are the captured step's static
            # output buffers, which the next replay overwrites
            l = loss.detach().mean()
            loss_sum = l.clone() if loss_sum is None else loss_sum + l
            if accuracy is not None:
                a = accuracy.detach()
                acc_sum = a.clone() if acc_sum is None else acc_sum + a
            count += 1
        n = max(count, 1)
        if self.data_parallel is not None:
            # the first host wait of an epoch sits behind that epoch's collectives: bounded, with the communicator's
            # asynchronous error polled (a peer that died shows as RuntimeError here, not as a hang in float() below)
            self.data_parallel.comm.wait_idle()
        mean_loss = float(loss_sum) / n if loss_sum is not None else 0.0       # single sync per epoch
        mean_acc = float(acc_sum) / n if acc_sum is not None else 0.0
        self.check_device_status()                                             # (the stream is idle here: no extra wait)
        return mean_loss, mean_acc

    def check_device_status(self):
        """Raises DeviceStatusError (a RuntimeError) when a pass of this trainer reported a device-side failure in its status
        word (the reference reads its loss on the host every step, utils/trainer.py:145-147; this build once per epoch, here;
        the update kernel reads the same word every step and withholds the update, include/arvae_hip.h arvae_adam_step)."""
        fused = getattr(self, '_fused', None)
        if fused is not None and hasattr(fused, 'check_status'):
            fused.check_status()

    def backward(self, loss):
        """loss.backward() (utils/trainer.py:140) with a cached unit seed gradient: a bare `loss.backward()` makes torch fill a
        ones tensor on the device first, one more launch per step."""
        seed = getattr(self, '_seed_grad', None)
        if seed is None or seed.shape != loss.shape or seed.device != loss.device or seed.dtype != loss.dtype:
            seed = self._seed_grad = torch.ones_like(loss)
        torch.autograd.backward(loss, grad_tensors=seed)

    def cuda(self):
        self.model.cuda()

    def zero_grad(self):
        self.optimizer.zero_grad()

    def step(self):
        if self.data_parallel is not None:
            self.data_parallel.reduce_gradients(self.optimizer)
        self.optimizer.step()
        self.global_iter += 1

    def eval_model(self, data_loader, epoch_num):
        pass

    def load_model(self):
        on_gpu = torch.cuda.is_available()
        self.model.load(cpu=not on_gpu)
        if on_gpu:
            self.model.cuda()

    @abstractmethod
    def loss_and_acc_for_batch(self, batch, epoch_num=None, batch_num=None, train=True):
        """-> (loss tensor with grad_fn, accuracy tensor or None)"""

    @abstractmethod
    def process_batch_data(self, batch):
        """-> device tensors for loss_and_acc_for_batch"""

    def update_scheduler(self, epoch_num):
        pass

    @staticmethod
    def print_epoch_stats(epoch_index, num_epochs, mean_loss_train, mean_accuracy_train, mean_loss_val,
                          mean_accuracy_val):
        print(f'Train Epoch: {epoch_index + 1}/{num_epochs}')
        print(f'\tTrain Loss: {mean_loss_train}\tTrain Accuracy: {mean_accuracy_train * 100} %')
        print(f'\tValid Loss: {mean_loss_val}\tValid Accuracy: {mean_accuracy_val * 100} %')

    # -- static loss helpers (same names / argument meaning as utils/trainer.py:247-413) ---------------
    @staticmethod
    def mean_crossentropy_loss(weights, targets):
        """weights (B, T, V), targets (B, T) int64 -> mean cross entropy (utils/trainer.py:247-264)."""
        if weights.size(0) != targets.size(0) or weights.size(1) != targets.size(1):
            raise AssertionError('weights / targets shape mismatch')
        return ops.token_recon(weights, targets)[0]

    @staticmethod
    def mean_accuracy(weights, targets):
        """top-1 accuracy over B*T rows (utils/trainer.py:266-282)."""
        return ops.token_recon(weights.detach(), targets)[1]

    @staticmethod
    def compute_kld_loss(z_dist, prior_dist, beta, c=0.0):
        """beta * |KL(z_dist || prior).sum(1).mean() - c|   (utils/trainer.py:354-367)."""
        if torch.is_tensor(c):
            cap = c
        else:
            cap = None if c == 0.0 else torch.tensor([float(c)], device=z_dist.loc.device)
        if _is_standard_prior(prior_dist):
            out = ops.kld_loss(z_dist.loc, z_dist.scale, beta, cap)
        else:
            out = ops.kld_loss(z_dist.loc, z_dist.scale, beta, cap, prior_dist.loc.contiguous(),
                               prior_dist.scale.contiguous())
        return out if (torch.is_tensor(c) or cap is None) else out[0]

    @staticmethod
    def compute_reg_loss(z, labels, reg_dim, gamma, factor=1.0):
        """gamma * reg_loss_sign(z[:, reg_dim], labels)   (utils/trainer.py:369-376)."""
        return gamma * Trainer.reg_loss_sign(z[:, reg_dim], labels, factor=factor)

    @staticmethod
    def reg_loss_sign(latent_code, attribute, factor=1.0):
        """mean_ij |tanh(factor (x_i - x_j)) - sign(a_i - a_j)|   (utils/trainer.py:378-403)."""
        x = latent_code.reshape(-1, 1)
        a = attribute.reshape(-1, 1).to(torch.float32)
        return ops.reg_loss(x, a, (0,), 1.0, factor)

    @staticmethod
    def get_save_dir(model, sub_dir_name='results'):
        path = os.path.join(os.path.dirname(model.filepath), sub_dir_name)
        os.makedirs(path, exist_ok=True)
        return path
