"""ImageVAETrainer: the AR-VAE loss step for the conv VAEs on the HIP kernels.

Loss recipe and API of the reference's imagevae/image_vae_trainer.py:65-217,
623-655:   loss = recon + beta*|KL - c| + sum_{d in reg_dim} gamma * reg(z[:,d], labels[:,d])
with the R-dimension Python loop replaced by one all-pairs kernel launch and the
two passes over the logits (BCE, accuracy) by one fused reduction.
"""
from typing import Tuple

import numpy as np
import os

import torch

from . import ops
from .trainer import Trainer

MNIST_REG_TYPES = {'digit_identity': 0, 'area': 1, 'length': 2, 'thickness': 3, 'slant': 4, 'width': 5,
                   'height': 6}
DSPRITES_REG_TYPE = {'color': 0, 'shape': 1, 'scale': 2, 'orientation': 3, 'posx': 4, 'posy': 5}
DATASET_REG_TYPE_DICT = {'mnist': MNIST_REG_TYPES, 'dsprites': DSPRITES_REG_TYPE}


def get_reg_dim(attr_dict):
    return tuple(v for k, v in attr_dict.items() if k not in ('digit_identity', 'color'))


class ImageVAETrainer(Trainer):
    def __init__(self, dataset, model, lr=1e-4, reg_type: Tuple[str] = None, reg_dim: Tuple[int] = 0,
                 dec_dist='bernoulli', beta=4.0, gamma=10.0, capacity=0.0, rand=0, delta=1.0):
        super().__init__(dataset, model, lr)
        kind = dataset.__class__.__name__
        if kind == 'MorphoMnistDataset':
            self.dataset_type = 'mnist'
        elif kind == 'DspritesDataset':
            self.dataset_type = 'dsprites'
        else:
            raise ValueError(f'Dataset type not recognized: {kind}')
        self.attr_dict = DATASET_REG_TYPE_DICT[self.dataset_type]
        self.reverse_attr_dict = {v: k for k, v in self.attr_dict.items()}
        self.metrics = {}
        self.beta = beta
        self.capacity = torch.tensor([capacity], dtype=torch.float32)
        self._capacity_nonzero = float(capacity) != 0.0
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
        self.dec_dist = dec_dist
        if len(self.reg_type) != 0:
            self.use_reg_loss = True
            self.reg_dim = reg_dim
            self.gamma = gamma
            self.delta = delta
            self.trainer_config += f'g_{self.gamma}_d_{self.delta}_' + '_'.join(self.reg_type) + '_'
        self.model.update_trainer_config(self.trainer_config)
        self.last_terms = {}
        self.use_fused = True          # whole-model C calls (arvae_amd.fused); False = one autograd node per layer
        self._fused = None
        # A TRAINING step (train=True, gradients enabled) leaves the forward pass's finishing step -- the sums that become the
        # loss, its split and the accuracy -- to the first launch of loss.backward() (ARVAE_VAE_DEFER_FINISH): those scalars are
        # final once backward() has run and read as NaN before.  The reference's loop (utils/trainer.py:136-147) and this
        # build's read them after step(); set False for code that looks at the loss between the two calls.
        self.defer_loss_finish = os.environ.get('ARVAE_DEFER_FINISH', '1') != '0'      # (the variable: same-box A/B runs)

    def cuda(self):
        super().cuda()
        self.capacity = self.capacity.cuda()

    def process_batch_data(self, batch):
        if self.dataset_type == 'mnist':
            inputs, _, labels = batch
        else:
            inputs, labels = batch
        dev = next(self.model.parameters()).device
        return (inputs.to(dev, torch.float32, non_blocking=True).contiguous(),
                labels.to(dev, torch.float32, non_blocking=True).contiguous())

    def loss_and_acc_for_batch(self, batch, epoch_num=None, batch_num=None, train=True):
        first_of_epoch = self.cur_epoch_num != epoch_num
        if first_of_epoch:
            self.cur_epoch_num = epoch_num
        inputs, labels = batch
        if self.capacity.device != inputs.device:
            self.capacity = self.capacity.to(inputs.device)
        if self.use_fused and inputs.is_cuda:
            return self._fused_loss_and_acc(inputs, labels, first_of_epoch, epoch_num, batch_num, train)

        outputs, z_dist, prior_dist, z_tilde, _ = self.model(inputs)
        recons_loss, accuracy = ops.image_recon(outputs, inputs, self.dec_dist)
        cap = self.capacity
        if self.data_parallel is not None and self._capacity_nonzero:   # |KL - c| needs the global KL mean (parallel.py)
            kl_local = self.compute_kld_loss(z_dist, prior_dist, beta=1.0, c=0.0).detach()
            cap = self.data_parallel.shifted_capacity(kl_local, self.capacity)
        dist_loss = self.compute_kld_loss(z_dist, prior_dist, beta=self.beta, c=cap)
        loss = recons_loss + dist_loss
        reg_loss = None
        if self.use_reg_loss:
            if type(self.reg_dim) != tuple:
                raise TypeError('Regularization dimension must be a tuple of integers')
            if self.data_parallel is not None:
                reg_loss = self.data_parallel.reg_loss(z_tilde, labels, self.reg_dim, self.gamma, self.delta)
            else:
                reg_loss = ops.reg_loss(z_tilde, labels, self.reg_dim, self.gamma, self.delta)
            loss = loss + reg_loss
        self.last_terms = {'recons': recons_loss.detach(), 'dist': dist_loss.detach(),
                           'reg': None if reg_loss is None else reg_loss.detach()}
        if first_of_epoch and self.writer is not None:
            self.writer.add_scalar('loss_split/recons_loss', recons_loss.item(), epoch_num)
            self.writer.add_scalar('loss_split/dist_loss', (dist_loss / self.beta).item(), epoch_num)
            if reg_loss is not None:
                self.writer.add_scalar('loss_split/reg_loss', (reg_loss / self.gamma).item(), epoch_num)
        if not train and batch_num == 0 and self.writer is not None:
            from .logging_utils import image_grid
            n = min(inputs.size(0), 16)
            self.writer.add_image('reconstruction',
                                  image_grid(torch.cat([inputs[:n], torch.sigmoid(outputs[:n].detach())]).cpu(), n),
                                  epoch_num)
        return loss, accuracy

    def _fused_loss_and_acc(self, inputs, labels, first_of_epoch, epoch_num, batch_num, train):
        """Same loss as above through arvae_image_vae_forward / _backward (one C call per pass)."""
        from .fused import DIST, RECON, REG, FusedImageVAE
        if type(self.reg_dim) != tuple and self.use_reg_loss:
            raise TypeError('Regularization dimension must be a tuple of integers')
        model = self.model
        if self._fused is None:
            self._fused = FusedImageVAE(model, self.optimizer, self.reg_dim if self.use_reg_loss else (), self.beta,
                                        self.gamma, self.delta, self.dec_dist)
        n = inputs.size(0)
        x = inputs.contiguous().view(n, model.image_hw, model.image_hw, 1)
        masks = model._next_masks(n, inputs.device)
        if model._eps_queue:                                     # explicit noise (parity runs)
            eps, draw = model._noise(torch.empty(n, model.z_dim, device=inputs.device)), False
        else:                                                    # drawn inside the fused heads kernel, written for backward
            eps, draw = torch.empty(n, model.z_dim, device=inputs.device), True
        dp = self.data_parallel
        # (the epoch's first batch is logged from the host right below: it finishes inside the forward pass)
        defer = train and self.defer_loss_finish and not (first_of_epoch and self.writer is not None)
        loss, scalars, accuracy, z, mu, sigma, logits = self._fused.run(x, labels, eps, masks, self.capacity, dp=dp,
                                                                       capacity_nonzero=self._capacity_nonzero, draw_eps=draw,
                                                                       defer_finish=defer)
        reg_loss = scalars[REG].detach() if self.use_reg_loss else None
        self.last_terms = {'recons': scalars[RECON].detach(), 'dist': scalars[DIST].detach(), 'reg': reg_loss}
        self.last_outputs = {'logits': logits.view(inputs.size()), 'z': z, 'mu': mu, 'sigma': sigma}
        if first_of_epoch and self.writer is not None:
            self.writer.add_scalar('loss_split/recons_loss', scalars[RECON].item(), epoch_num)
            self.writer.add_scalar('loss_split/dist_loss', scalars[DIST].item() / self.beta, epoch_num)
            if reg_loss is not None:
                self.writer.add_scalar('loss_split/reg_loss', reg_loss.item() / self.gamma, epoch_num)
        if not train and batch_num == 0 and self.writer is not None:
            from .logging_utils import image_grid
            k = min(n, 16)
            recon = torch.sigmoid(logits.view(inputs.size())[:k].detach())
            self.writer.add_image('reconstruction', image_grid(torch.cat([inputs[:k], recon]).cpu(), k), epoch_num)
        return loss, accuracy

    # -- evaluation-only inference (image_vae_trainer.py:264-288,381-403): encoder / decoder passes on the forward
    #    kernels; the metrics fed from here (utils/evaluation.py) stay host-side and are out of scope ---------------------
    def _extract_relevant_attributes(self, attributes):
        attr_list = [a for a in self.attr_dict if a not in ('digit_identity', 'color')]
        return attributes[:, [self.attr_dict[a] for a in attr_list]], attr_list

    def compute_representations(self, data_loader, num_batches=200):
        """-> (latent codes (n, z_dim), attribute columns (n, k), attribute names); stops after num_batches + 1 batches
        like the reference (`if sample_id == 200: break` comes after the append)."""
        codes, attrs = [], []
        self.model.eval()
        with torch.no_grad():
            for i, batch in enumerate(data_loader):
                inputs, labels = self.process_batch_data(batch)
                codes.append(self.model(inputs)[3])
                attrs.append(labels)
                if i == num_batches:
                    break
        codes = torch.cat(codes).cpu().numpy()             # one device->host copy for the whole sweep
        attrs, names = self._extract_relevant_attributes(torch.cat(attrs).cpu().numpy())
        return codes, attrs, names

    def compute_latent_interpolations(self, latent_code, dim1=0, num_points=10):
        """Decoder sweep of one latent dimension over [-4, 4]: (num_points, 1, H, W) probabilities on the device."""
        dev = next(self.model.parameters()).device
        z = torch.as_tensor(latent_code, dtype=torch.float32, device=dev).reshape(1, -1).repeat(num_points, 1)
        z[:, dim1] = torch.linspace(-4.0, 4.0, num_points, device=dev)
        with torch.no_grad():
            return torch.sigmoid(self.model.decode(z.contiguous()))

    def compute_latent_interpolations2d(self, latent_code, dim1=0, dim2=1, num_points=10):
        dev = next(self.model.parameters()).device
        x = torch.linspace(-4.0, 4.0, num_points, device=dev)
        z1, z2 = torch.meshgrid([x, x], indexing='ij')
        z = torch.as_tensor(latent_code, dtype=torch.float32, device=dev).reshape(1, -1).repeat(num_points * num_points, 1)
        z[:, dim1], z[:, dim2] = z1.reshape(-1), z2.reshape(-1)
        with torch.no_grad():
            return torch.sigmoid(self.model.decode(z.contiguous()))

    def save_representations(self, path, data_loader=None, batch_size=128):
        """Write the record the reference's compute_eval_metrics consumes (image_vae_trainer.py:289-317): latent codes,
        attribute columns and attribute names as JSON, from the encoder-only pass.  The disentanglement metrics themselves
        (utils/evaluation.py: sklearn / scipy on the host) are out of scope: they run unchanged on
        this file's arrays."""
        import json
        if data_loader is None:
            _, _, data_loader = self.dataset.data_loaders(batch_size=batch_size)
        codes, attrs, names = self.compute_representations(data_loader)
        with open(path, 'w') as f:
            json.dump({'latent_codes': codes.tolist(), 'attributes': np.asarray(attrs).tolist(), 'attr_list': list(names)}, f)
        return codes, attrs, names

    def compute_eval_metrics(self, batch_size=128):
        """results_dict.json next to the checkpoint, as in the reference (image_vae_trainer.py:289-317): loaded when it exists,
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

    def loss_and_acc_test(self, data_loader):
        """mean RECONSTRUCTION loss (no KL / regulariser terms) and mean pixel accuracy over the loader's batches
        (image_vae_trainer.py:595-621); accumulated on the device, one host sync at the end."""
        loss_sum = acc_sum = None
        count = 0
        with torch.no_grad():
            for batch in data_loader:
                inputs, _ = self.process_batch_data(batch)
                outputs = self.model(inputs)[0]
                loss, acc = ops.image_recon(outputs, inputs, self.dec_dist)
                loss_sum = loss.detach().clone() if loss_sum is None else loss_sum + loss.detach()
                acc_sum = acc.detach().clone() if acc_sum is None else acc_sum + acc.detach()
                count += 1
        n = max(count, 1)
        return (float(loss_sum) / n if count else 0.0), (float(acc_sum) / n if count else 0.0)

    def test_model(self, batch_size):
        """Mean reconstruction loss / accuracy over the evaluation split (image_vae_trainer.py:582-593)."""
        _, _, loader = self.dataset.data_loaders(batch_size)
        mean_loss, mean_acc = self.loss_and_acc_test(loader)
        print('Test Epoch:')
        print('\tTest Loss: ', mean_loss, '\n\tTest Accuracy: ', mean_acc * 100)
        return {'test_loss': mean_loss, 'test_acc': mean_acc}

    # -- static helpers (image_vae_trainer.py:623-655) ---------------------------------------------------
    @staticmethod
    def reconstruction_loss(x, x_recons, dist):
        if dist not in ('bernoulli', 'gaussian'):
            raise AttributeError('invalid dist')
        return ops.image_recon(x_recons, x, dist)[0]

    @staticmethod
    def mean_accuracy(weights, targets):
        """weights are probabilities (sigmoid already applied by the caller, as in the reference)."""
        logits = torch.logit(weights.detach().clamp(0.0, 1.0))     # p >= .5  <=>  logit >= 0
        return ops.image_recon(logits.nan_to_num(posinf=1e30, neginf=-1e30), targets, 'bernoulli')[1]
