"""MeasureVAETrainer: the AR-VAE loss step for 24-tick music measures on the HIP kernels.

Loss recipe and API of the reference's measurevae/measure_vae_trainer.py:23-186,399-400:
    loss = CE_mean(weights, score) + beta*|KL| + sum_{d in reg_dim} gamma * reg(z[:,d], attr[:,d])
The four attribute labels (rhythmic complexity, pitch range, note density, contour) come from one integer
kernel over the score batch instead of B x 24 Python loops with music21 lookups
(data/dataloaders/bar_dataset.py:338-500).
"""
from typing import Tuple

import numpy as np
import torch

from . import ops
from .trainer import Trainer

MUSIC_REG_TYPE = {'rhy_complexity': 0, 'pitch_range': 1, 'note_density': 2, 'contour': 3}

# reference data/dataloaders/bar_dataset_helpers.py:21-30
RHY_COMPLEXITY_COEFFS = [0.20, 1, 2, 0.5, 2, 1, 0.67, 1, 2, 0.5, 2, 1, 0.25, 1, 2, 0.5, 2, 1, 0.67, 1, 2, 0.5, 2, 1]
_NON_NOTES = ('__', 'START', 'END', 'rest', None)
_PITCH = {'C': 0, 'D': 2, 'E': 4, 'F': 5, 'G': 7, 'A': 9, 'B': 11}


def pitch_to_midi(name):
    """'G3', 'F#4', 'B-4' / 'Bb4' -> MIDI number (what music21.pitch.Pitch(name).midi gives)."""
    step, rest = name[0].upper(), name[1:]
    acc = 0
    while rest and rest[0] in '#-b':
        acc += 1 if rest[0] == '#' else -1
        rest = rest[1:]
    return 12 * (int(rest) + 1) + _PITCH[step] + acc


def build_measure_tables(dataset, device):
    """(midi int32[V], is_note uint8[V], is_density_note uint8[V]) from the dataset's index2note dict."""
    index2note = dataset.index2note_dicts
    v = len(index2note)
    midi, is_note, is_dens = np.zeros(v, np.int32), np.zeros(v, np.uint8), np.zeros(v, np.uint8)
    for i, sym in index2note.items():
        if sym in _NON_NOTES:
            is_dens[i] = 1 if sym is None else 0          # note density does not exclude `None` (bar_dataset.py:348-356)
            continue
        midi[i] = pitch_to_midi(sym)
        is_note[i] = is_dens[i] = 1
    return tuple(torch.from_numpy(a).to(device) for a in (midi, is_note, is_dens))


class MeasureVAETrainer(Trainer):
    def __init__(self, dataset, model, lr=1e-4, reg_type: Tuple[str] = None, reg_dim: Tuple[int] = 0, beta=0.001,
                 gamma=1.0, capacity=0.0, rand=0, delta=10.0):
        super().__init__(dataset, model, lr)
        kind = dataset.class_name[5:9]
        if kind == 'Chor':
            self.dataset_type = 'bach'
        elif kind == 'Folk':
            self.dataset_type = 'folk'
        else:
            raise ValueError('Dataset Type not recognized')
        self.attr_dict = MUSIC_REG_TYPE
        self.reverse_attr_dict = {v: k for k, v in self.attr_dict.items()}
        self.metrics = {}
        self.beta = beta
        self.capacity = torch.tensor([capacity], dtype=torch.float32)
        self.gamma = 0.0
        self.delta = 0.0
        self.cur_epoch_num = 0
        self.warm_up_epochs = 10
        self.reg_type = reg_type if reg_type is not None else ()
        self.reg_dim = ()
        self.use_reg_loss = False
        self.rand_seed = rand
        torch.manual_seed(self.rand_seed)
        np.random.seed(self.rand_seed)
        self.trainer_config = f'_r_{self.rand_seed}_b_{self.beta}_'
        if capacity != 0.0:
            self.trainer_config += f'c_{capacity}_'
        if len(self.reg_type) != 0:
            self.use_reg_loss = True
            self.reg_dim = reg_dim
            self.gamma = gamma
            self.delta = delta
            self.trainer_config += f'g_{self.gamma}_d_{self.delta}_' + '_'.join(self.reg_type) + '_'
        self.model.update_trainer_config(self.trainer_config)
        self._tables = None
        self.last_terms = {}
        self.use_graph_replay = True          # a few hundred small launches per step: replayed from HIP graphs (trainer.py)

    def process_batch_data(self, batch):
        score, metadata = batch
        n_bars = getattr(self.dataset, 'n_bars', None)
        if n_bars is not None:
            b = score.size(0)
            score = score.view(b * n_bars, -1)
            metadata = metadata.view(b * n_bars, -1)
        dev = next(self.model.parameters()).device
        return score.to(dev, torch.int64).contiguous(), metadata.to(dev, torch.int64).contiguous()

    def _attr_tables(self, device):
        if self._tables is None or self._tables[0][0].device != device:
            w = torch.tensor(RHY_COMPLEXITY_COEFFS, dtype=torch.float64).float()
            self._tables = (build_measure_tables(self.dataset, device), w.to(device), float(w.sum()))
        return self._tables

    def compute_attribute_labels(self, score, attr_list=None):
        """(B, 24) int64 -> (B, len(attr_list)) attribute values (all four by default)."""
        tables, w, norm = self._attr_tables(score.device)
        labels = ops.measure_attributes(score, tables, w, norm)
        if attr_list is None:
            return labels
        cols = []
        for name in attr_list:
            if name not in self.attr_dict:
                raise ValueError('Invalid regularization attribute')
            cols.append(self.attr_dict[name])
        return labels[:, cols]

    def loss_and_acc_for_batch(self, batch, epoch_num=None, batch_num=None, train=True):
        first_of_epoch = self.cur_epoch_num != epoch_num
        if first_of_epoch:
            self.cur_epoch_num = epoch_num
        score, metadata = batch
        weights, samples, z_dist, prior_dist, z_tilde, _ = self.model(measure_score_tensor=score,
                                                                       measure_metadata_tensor=metadata, train=train)
        recons_loss, accuracy = ops.token_recon(weights, score)
        dist_loss = self.compute_kld_loss(z_dist, prior_dist, self.beta)
        loss = recons_loss + dist_loss
        reg_loss = None
        if self.use_reg_loss:
            if type(self.reg_dim) != tuple:
                raise TypeError('Regularization dimension must be a tuple of integers')
            attr_labels = self.compute_attribute_labels(score)
            if self.data_parallel is not None:
                reg_loss = self.data_parallel.reg_loss(z_tilde, attr_labels, self.reg_dim, self.gamma, self.delta)
            else:
                reg_loss = ops.reg_loss(z_tilde, attr_labels, self.reg_dim, self.gamma, self.delta)
            loss = loss + reg_loss
        self.last_terms = {'recons': recons_loss.detach(), 'dist': dist_loss.detach(),
                           'reg': None if reg_loss is None else reg_loss.detach()}
        if first_of_epoch and self.writer is not None and not torch.cuda.is_current_stream_capturing():
            self.log_loss_split(epoch_num)
        return loss, accuracy

    def log_loss_split(self, epoch_num):
        """the loss terms of the last step to the summary writer (measure_vae_trainer.py:116-127); also called by the
        epoch loop after the first graph-replayed step of an epoch, whose terms live in the captured step's outputs."""
        t = self.last_terms
        if self.writer is None or not t:
            return
        self.writer.add_scalar('loss_split/recons_loss', t['recons'].item(), epoch_num)
        self.writer.add_scalar('loss_split/dist_loss', (t['dist'] / self.beta).item(), epoch_num)
        if t['reg'] is not None:
            self.writer.add_scalar('loss_split/reg_loss', (t['reg'] / self.gamma).item(), epoch_num)

    # -- evaluation-only inference (measure_vae_trainer.py:188-212) ---------------------------------------------------
    def compute_representations(self, data_loader, num_batches=None):
        """-> (latent codes, attribute labels (n, 4), attribute names) over at most num_batches + 1 batches."""
        num_batches = 200 if num_batches is None else num_batches
        codes, attrs = [], []
        self.model.eval()
        with torch.no_grad():
            for i, batch in enumerate(data_loader):
                score, metadata = self.process_batch_data(batch)
                codes.append(self.model(score, metadata, train=False)[4])
                attrs.append(self.compute_attribute_labels(score))
                if i == num_batches:
                    break
        if not codes:
            raise ValueError('compute_representations: the loader produced no batch (split smaller than the batch size?)')
        return torch.cat(codes).cpu().numpy(), torch.cat(attrs).cpu().numpy(), list(self.attr_dict)

    def test_model(self, batch_size):
        _, _, loader = self.dataset.data_loaders(batch_size)
        self.model.eval()
        with torch.no_grad():
            loss, acc = self.loss_and_acc_on_epoch(loader, epoch_num=0, train=False)
        return {'test_loss': loss, 'test_acc': acc}

    @staticmethod
    def reconstruction_loss(x, x_recons):
        return Trainer.mean_crossentropy_loss(weights=x_recons, targets=x)
