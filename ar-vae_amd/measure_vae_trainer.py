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
        ops.rng_reseed(self.rand_seed)                 # torch.manual_seed + restart of the library's Philox stream position
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
        self.use_graph_replay = True          # about a hundred small launches per step: replayed from HIP graphs (trainer.py)
        # forward + loss terms and backward as one library call each (fused_measure.py, csrc/plan_measure.hip) where the executor
        # is built for the model; False keeps the per-layer autograd path (the same launches issued one by one)
        self.use_fused_step = True
        self._fused = None

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

    def _fused_binding(self):
        """the whole-model executor bound to this trainer's arena and hyper-parameters, or None when the steps take the per-layer
        path: debug checks, CPU models, unsupported shapes.  Data-parallel steps run it too (forward, one grouped all-gather of z
        and the attribute labels, arvae_measure_vae_finish)."""
        if not self.use_fused_step or ops.checks_enabled():
            return None
        if self.use_reg_loss and type(self.reg_dim) != tuple:
            return None                                            # (the per-layer path raises the reference's TypeError)
        from .fused_measure import FusedMeasureVAE
        from .measure_vae import _use_sequence_kernels
        reg_dims = tuple(self.reg_dim) if self.use_reg_loss else ()
        key = (reg_dims, float(self.beta), float(self.gamma), float(self.delta))
        if self._fused is None or self._fused[0] != key:
            usable = (_use_sequence_kernels(self.model.encoder.rnn_hidden_size) and next(self.model.parameters()).is_cuda
                      and FusedMeasureVAE.supports(self.model, self.optimizer, reg_dims) is None)
            self._fused = (key, FusedMeasureVAE(self.model, self.optimizer, reg_dims, self.beta, self.gamma, self.delta) if usable else None)
        return self._fused[1]

    def fused_executor(self, score):
        """-> the executor for a step on `score` (B, 24), or None: this step takes the per-layer path"""
        fused = self._fused_binding() if score.is_cuda else None
        if fused is None or not fused.fits(score.shape[0]):
            return None
        enc, dec = self.model.encoder, self.model.decoder
        if bool(enc._mask_queue) != bool(dec._mask_queue):
            return None                                            # explicit keep-masks for one half only: per-layer path
        return fused

    def _replay_step(self, batch):
        # the executor issues a step's launches from two library calls (three and one collective under data parallelism): a captured
        # graph has no host work left to save and its nodes cost more than the stream launches they replace (B = 256: 1.09 ms eager,
        # 1.11 ms replayed)
        fused = self._fused_binding()
        if fused is not None:
            score = batch[0]
            rows = score.shape[0] * (getattr(self.dataset, 'n_bars', None) or 1) if torch.is_tensor(score) else -1
            if rows > 0 and fused.fits(rows):
                return None
        return super()._replay_step(batch)                   # (a batch the executor does not take: the per-layer path, replayed)

    def _fused_loss_and_acc(self, fused, score, epoch_num, first_of_epoch, train):
        from .fused_measure import ACC, DIST, RECON, REG
        enc, dec = self.model.encoder, self.model.decoder
        eps = enc._eps_queue.popleft() if enc._eps_queue else enc.static_eps
        masks = None
        if self.model.training and enc._mask_queue:
            masks = (enc._mask_queue.popleft(),) + tuple(dec._mask_queue.popleft())
        tables = fused.tables(self, score.device) if self.use_reg_loss else None
        loss, scalars, accuracy, *_ = fused.run(score, train, None, tables, eps, masks, self.data_parallel)
        self.last_terms = {'recons': scalars[RECON], 'dist': scalars[DIST], 'reg': scalars[REG] if self.use_reg_loss else None}
        if first_of_epoch and self.writer is not None and not torch.cuda.is_current_stream_capturing():
            self.log_loss_split(epoch_num)
        return loss.reshape(()), accuracy

    def loss_and_acc_for_batch(self, batch, epoch_num=None, batch_num=None, train=True):
        first_of_epoch = self.cur_epoch_num != epoch_num
        if first_of_epoch:
            self.cur_epoch_num = epoch_num
        score, metadata = batch
        fused = self.fused_executor(score)
        if fused is not None:
            return self._fused_loss_and_acc(fused, score, epoch_num, first_of_epoch, train)
        weights, samples, z_dist, prior_dist, z_tilde, _ = self.model(measure_score_tensor=score, measure_metadata_tensor=metadata,
                                                                       train=train, need_prior_sample=False)
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

    def save_representations(self, path, data_loader=None, batch_size=256, num_batches=None):
        """Write the record the reference's compute_eval_metrics consumes (measure_vae_trainer.py:217-243): latent codes,
        attribute labels and attribute names as JSON, from the encoder-only pass.  The disentanglement metrics themselves
        (utils/evaluation.py: sklearn / scipy on the host) are out of scope: they run unchanged on
        this file's arrays."""
        import json
        if data_loader is None:
            _, _, data_loader = self.dataset.data_loaders(batch_size=batch_size)
        codes, attrs, names = self.compute_representations(data_loader, num_batches)
        with open(path, 'w') as f:
            json.dump({'latent_codes': codes.tolist(), 'attributes': attrs.tolist(), 'attr_list': list(names)}, f)
        return codes, attrs, names

    def compute_eval_metrics(self, batch_size=256):
        """results_dict.json next to the checkpoint, as in the reference (measure_vae_trainer.py:217-243): loaded when it exists,
        otherwise created with what this path computes on the device -- the test loss / accuracy -- and the file the host-side
        metric suite reads (representations.json)."""
        import json
        import os
        folder = os.path.dirname(self.model.filepath)
        results_fp = os.path.join(folder, 'results_dict.json')
        if os.path.exists(results_fp):
            with open(results_fp) as f:
                self.metrics = json.load(f)
            return self.metrics
        os.makedirs(folder, exist_ok=True)
        rep_fp = os.path.join(folder, 'representations.json')
        self.save_representations(rep_fp, batch_size=batch_size)
        self.metrics = {'representations': rep_fp}
        self.metrics.update(self.test_model(batch_size=batch_size))
        with open(results_fp, 'w') as f:
            json.dump(self.metrics, f, indent=2)
        return self.metrics

    def decode_latent_codes(self, latent_codes):
        """(n, z_dim) latent codes -> (music21 score or None, note indices (n, 1, 24) int64): the decoder alone, free-running
        (measure_vae_trainer.py:281-288).  The score object needs the dataset's music21 converter (`tensor_to_m21score`);
        datasets without one (the device-resident loaders here) give None."""
        dev = next(self.model.parameters()).device
        z = torch.as_tensor(latent_codes, dtype=torch.float32, device=dev).contiguous()
        dummy = torch.zeros(z.size(0), self.model.num_ticks_per_measure, dtype=torch.int64, device=dev)
        with torch.no_grad():
            _, tensor_score = self.model.decoder(z, dummy, False)
        to_score = getattr(self.dataset, 'tensor_to_m21score', None)
        return (to_score(tensor_score) if callable(to_score) else None), tensor_score

    def compute_latent_interpolations(self, latent_code, original_score=None, dim1=0, num_points=5):
        """Sweep latent dimension dim1 over [-4, 4] (measure_vae_trainer.py:290-308): all num_points codes are decoded in ONE
        batch instead of one decoder call per point.  -> (concatenated music21 score or None, note indices (num_points, 24))."""
        if num_points % 2 != 1:
            raise AssertionError('num_points must be odd')
        dev = next(self.model.parameters()).device
        z = torch.as_tensor(latent_code, dtype=torch.float32, device=dev).reshape(1, -1).repeat(num_points, 1)
        z[:, dim1] = torch.linspace(-4.0, 4.0, num_points, device=dev)
        scores, tensor_score = self.decode_latent_codes(z)
        tensor_score = tensor_score.squeeze(1)
        concat = getattr(self.dataset, 'concatenate_scores', None)
        to_score = getattr(self.dataset, 'tensor_to_m21score', None)
        score = None
        if callable(concat) and callable(to_score):
            parts = [to_score(tensor_score[i:i + 1, None, :]) for i in range(num_points)]
            if original_score is not None:
                parts[num_points // 2] = original_score
            score = concat(parts)
        return score, tensor_score

    def loss_and_acc_test(self, data_loader):
        """mean RECONSTRUCTION loss (no KL / regulariser terms) and mean top-1 accuracy over the loader's batches
        (measure_vae_trainer.py:367-397); accumulated on the device, one host sync at the end."""
        loss_sum = acc_sum = None
        count = 0
        with torch.no_grad():
            for batch in data_loader:
                score, metadata = self.process_batch_data(batch)
                fused = self.fused_executor(score)
                if fused is not None:                          # the executor's forward pass leaves both numbers in its scalars
                    from .fused_measure import ACC, RECON
                    enc = self.model.encoder
                    eps = enc._eps_queue.popleft() if enc._eps_queue else enc.static_eps
                    tables = fused.tables(self, score.device) if self.use_reg_loss else None
                    scalars = fused.run(score, False, None, tables, eps, None)[1]
                    loss, acc = scalars[RECON], scalars[ACC]
                else:
                    weights = self.model(measure_score_tensor=score, measure_metadata_tensor=metadata, train=False)[0]
                    loss, acc = ops.token_recon(weights, score)
                loss_sum = loss.detach().clone() if loss_sum is None else loss_sum + loss.detach()
                acc_sum = acc.detach().clone() if acc_sum is None else acc_sum + acc.detach()
                count += 1
        n = max(count, 1)
        return (float(loss_sum) / n if count else 0.0), (float(acc_sum) / n if count else 0.0)

    def test_model(self, batch_size):
        _, _, loader = self.dataset.data_loaders(batch_size)
        mean_loss, mean_acc = self.loss_and_acc_test(loader)
        print('Test Epoch:')
        print('\tTest Loss: ', mean_loss, '\n\tTest Accuracy: ', mean_acc * 100)
        return {'test_loss': mean_loss, 'test_acc': mean_acc}

    @staticmethod
    def reconstruction_loss(x, x_recons):
        return Trainer.mean_crossentropy_loss(weights=x_recons, targets=x)
