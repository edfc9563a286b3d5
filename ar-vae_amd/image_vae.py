"""Conv AR-VAEs for dSprites and Morpho-MNIST on the HIP kernels.

Same class names, constructor, forward contract and state_dict keys/layouts as
the reference (imagevae/mnist_vae.py:7-105, imagevae/dsprites_vae.py:7-55):
    forward(x) -> (logits like x, z_dist, prior_dist, z_tilde, z_prior)
Activations are channels-last inside; a 1-channel image is identical in both
layouts, and the flatten between conv and dense stacks is honoured by a channel
permutation inside the dense kernels (no transpose pass).
"""
from collections import deque

import torch
from torch import distributions

from . import ops
from .model import LayerStack, Model, ParamLayer
from .ops import ACT_NONE, ACT_RELU, ACT_SELU, Link


def _conv(cin, cout):          # nn.Conv2d weight [cout, cin, 4, 4]
    return ParamLayer((cout, cin, 4, 4), cout, cin * 16)


def _deconv(cin, cout):        # nn.ConvTranspose2d weight [cin, cout, 4, 4]
    return ParamLayer((cin, cout, 4, 4), cout, cout * 16)


def _dense(fin, fout):
    return ParamLayer((fout, fin), fout, fin)


class MnistVAE(Model):
    """Morpho-MNIST VAE: 3x Conv(k4,s1)+SELU+Dropout(.5) -> Linear+SELU -> (mu, log_std) heads;
    2x Linear+SELU -> 3x ConvTranspose (SELU+Dropout between)."""
    z_dim_default = 16

    def __init__(self):
        super().__init__()
        self.input_size = 784
        self.z_dim = 16
        self.inter_dim = 19
        self.enc_conv = LayerStack([(0, _conv(1, 64)), (3, _conv(64, 64)), (6, _conv(64, 8))])
        self.enc_lin = LayerStack([(0, _dense(2888, 256))])
        self.enc_mean = _dense(256, 16)
        self.enc_log_std = _dense(256, 16)
        self.dec_lin = LayerStack([(0, _dense(16, 256)), (2, _dense(256, 2888))])
        self.dec_conv = LayerStack([(0, _deconv(8, 64)), (3, _deconv(64, 64)), (6, _deconv(64, 1))])
        self._plan_mnist()
        self._init_common()

    def _plan_mnist(self):
        self.hidden_act = ACT_SELU
        self.dropout_p = 0.5
        # (stack index, link) in execution order; sizes 28 -> 25 -> 22 -> 19
        self.enc_conv_plan = [(0, Link(28, 28, 1, 25, 25, 64, 4, 4, 1, 0)),
                              (3, Link(25, 25, 64, 22, 22, 64, 4, 4, 1, 0)),
                              (6, Link(22, 22, 64, 19, 19, 8, 4, 4, 1, 0))]
        self.enc_lin_plan = [(0, Link.dense(2888, 256, in_perm=(8, 361)))]
        self.dec_lin_plan = [(0, Link.dense(16, 256)), (2, Link.dense(256, 2888, out_perm=(8, 361)))]
        self.dec_conv_plan = [(0, Link(22, 22, 64, 19, 19, 8, 4, 4, 1, 0)),
                              (3, Link(25, 25, 64, 22, 22, 64, 4, 4, 1, 0)),
                              (6, Link(28, 28, 1, 25, 25, 64, 4, 4, 1, 0))]
        self.dec_grid = (19, 19, 8)
        self.image_hw = 28

    def _init_common(self):
        self.head_link = Link.dense(256, self.z_dim)
        self.xavier_initialization()
        self.update_filepath()
        self._eps_queue = deque()      # explicit noise for parity runs (see push_noise)
        self._mask_queue = deque()     # explicit dropout keep-masks

    def __repr__(self):
        return 'MnistVAE' + self.trainer_config

    # -- explicit randomness (parity tests); default = device RNG ----------------------------------
    def push_noise(self, eps):
        """Use `eps` (B, z_dim) for the next rsample instead of drawing it."""
        self._eps_queue.append(eps)

    def push_dropout_masks(self, masks):
        """Use these uint8 keep-masks (channels-last, encoder then decoder order) for the next forward."""
        self._mask_queue.append(list(masks))

    def _noise(self, like):
        if self._eps_queue:
            return self._eps_queue.popleft().to(like.device, torch.float32).contiguous()
        return ops.normal_noise(like.shape, like.device)

    def _next_masks(self, n, device):
        count = sum(1 for _ in self.enc_conv_plan) + len(self.dec_conv_plan) - 1
        if self.dropout_p == 0.0 or not self.training:
            return [None] * count
        if self._mask_queue:
            return [m.to(device).contiguous() for m in self._mask_queue.popleft()]
        shapes = [l.lo_shape(n) for _, l in self.enc_conv_plan] + [l.hi_shape(n) for _, l in self.dec_conv_plan[:-1]]
        return ops.keep_masks(shapes, self.dropout_p, device)        # (one launch for the five Dropout layers' masks)

    # -- encoder / decoder --------------------------------------------------------------------------
    def _encode_params(self, x, masks):
        n = x.size(0)
        h = x.contiguous().view(n, self.image_hw, self.image_hw, 1)
        for k, (idx, link) in enumerate(self.enc_conv_plan):
            layer = self.enc_conv[idx]
            h = ops.conv_down(h, layer.weight, layer.bias, link, self.hidden_act, masks[k])
        h = h.view(n, -1)
        for idx, link in self.enc_lin_plan:
            layer = self.enc_lin[idx]
            h = ops.dense(h, layer.weight, layer.bias, link, self.hidden_act)
        mu = ops.dense(h, self.enc_mean.weight, self.enc_mean.bias, self.head_link)
        log_std = ops.dense(h, self.enc_log_std.weight, self.enc_log_std.bias, self.head_link)
        return mu, log_std

    def encode(self, x):
        """-> Normal(mu, exp(log_std)); the reparameterised sample is computed by the same fused kernel
        and travels with the distribution object (used by reparametrize)."""
        self._masks = self._next_masks(x.size(0), x.device)
        mu, log_std = self._encode_params(x, self._masks)
        eps = self._noise(mu)
        sigma, z = ops.latent_head(mu, log_std, eps)
        z_dist = distributions.Normal(loc=mu, scale=sigma, validate_args=False)
        z_dist._arvae_sample = z
        return z_dist

    def decode(self, z):
        n = z.size(0)
        masks = getattr(self, '_masks', None) or self._next_masks(n, z.device)
        h = z.contiguous()
        for idx, link in self.dec_lin_plan:
            layer = self.dec_lin[idx]
            h = ops.dense(h, layer.weight, layer.bias, link, self.hidden_act)
        gh, gw, gc = self.dec_grid
        h = h.view(n, gh, gw, gc)
        n_enc = len(self.enc_conv_plan)
        last = len(self.dec_conv_plan) - 1
        for k, (idx, link) in enumerate(self.dec_conv_plan):
            layer = self.dec_conv[idx]
            if k < last:
                h = ops.conv_up(h, layer.weight, layer.bias, link, self.hidden_act, masks[n_enc + k])
            else:
                h = ops.conv_up(h, layer.weight, layer.bias, link, ACT_NONE, None)
        self._masks = None
        return h.view(n, 1, self.image_hw, self.image_hw)

    def reparametrize(self, z_dist):
        z_tilde = getattr(z_dist, '_arvae_sample', None)
        if z_tilde is None:
            _, z_tilde = ops.latent_head(z_dist.loc, torch.log(z_dist.scale), self._noise(z_dist.loc))
        prior_dist = distributions.Normal(loc=torch.zeros_like(z_dist.loc), scale=torch.ones_like(z_dist.scale),
                                          validate_args=False)
        prior_dist._arvae_standard = True
        z_prior = ops.normal_noise(z_dist.loc.shape, z_dist.loc.device)   # the reference's second, unused draw (mnist_vae.py:86)
        return z_tilde, z_prior, prior_dist

    def forward(self, x):
        z_dist = self.encode(x)
        z_tilde, z_prior, prior_dist = self.reparametrize(z_dist)
        output = self.decode(z_tilde).view(x.size())
        return output, z_dist, prior_dist, z_tilde, z_prior


class DspritesVAE(MnistVAE):
    """dSprites VAE: 4x Conv(k4,s2,p1)+ReLU -> 2x Linear+ReLU -> heads; 3x Linear+ReLU -> 4x ConvTranspose(k4,s2,p1)."""

    def __init__(self):
        Model.__init__(self)
        self.input_size = 4096
        self.z_dim = 10
        self.inter_dim = 4
        self.enc_conv = LayerStack([(0, _conv(1, 32)), (2, _conv(32, 32)), (4, _conv(32, 32)), (6, _conv(32, 32))])
        self.enc_lin = LayerStack([(0, _dense(512, 256)), (2, _dense(256, 256))])
        self.enc_mean = _dense(256, 10)
        self.enc_log_std = _dense(256, 10)
        self.dec_lin = LayerStack([(0, _dense(10, 256)), (2, _dense(256, 256)), (4, _dense(256, 512))])
        self.dec_conv = LayerStack([(0, _deconv(32, 32)), (2, _deconv(32, 32)), (4, _deconv(32, 32)),
                                    (6, _deconv(32, 1))])
        self.hidden_act = ACT_RELU
        self.dropout_p = 0.0

        def s2(hi, chi, clo):
            return Link(hi, hi, chi, hi // 2, hi // 2, clo, 4, 4, 2, 1)
        self.enc_conv_plan = [(0, s2(64, 1, 32)), (2, s2(32, 32, 32)), (4, s2(16, 32, 32)), (6, s2(8, 32, 32))]
        self.enc_lin_plan = [(0, Link.dense(512, 256, in_perm=(32, 16))), (2, Link.dense(256, 256))]
        self.dec_lin_plan = [(0, Link.dense(10, 256)), (2, Link.dense(256, 256)),
                             (4, Link.dense(256, 512, out_perm=(32, 16)))]
        self.dec_conv_plan = [(0, s2(8, 32, 32)), (2, s2(16, 32, 32)), (4, s2(32, 32, 32)), (6, s2(64, 1, 32))]
        self.dec_grid = (4, 4, 32)
        self.image_hw = 64
        self._init_common()

    def __repr__(self):
        return 'DspritesVAE' + self.trainer_config
