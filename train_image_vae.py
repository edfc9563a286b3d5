#!/usr/bin/env python3
"""Train / evaluate the image AR-VAEs on MI355X: counterpart of the reference's train_image_vae.py (same flags and
defaults, same `models/<repr>/<repr>.pt` and `runs/` side effects, same reg_type -> reg_dim mapping and seed loop).

Differences a user can observe: the dataset lives in HBM (arvae_amd.data), a single `-r <name>` works (the
reference indexes its attribute table with the whole tuple there and raises KeyError), `--no_log` survives the
second epoch, and after training the script prints the representation summary it can compute on the device
(latent codes / attributes of the evaluation split, test loss and accuracy) instead of the sklearn metric suite and
the GIF plots, which are outside this build's scope.
"""
import json
import os
import sys

import click
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from arvae_amd.data import DspritesDataset, MorphoMnistDataset  # noqa: E402
from arvae_amd.image_vae import DspritesVAE, MnistVAE  # noqa: E402
from arvae_amd.image_vae_trainer import DSPRITES_REG_TYPE, MNIST_REG_TYPES, ImageVAETrainer  # noqa: E402


def reg_dims_for(reg_type, attr_dict, skip=('digit_identity', 'color')):
    """The reference's --reg_type -> latent dimension mapping (train_image_vae.py:72-91)."""
    if len(reg_type) == 0:
        return (0,)
    if len(reg_type) == 1 and reg_type[0] == 'all':
        return tuple(v for k, v in attr_dict.items() if k not in skip)
    return tuple(attr_dict[r] for r in reg_type)


# flag surface of the reference script (names, defaults and help strings are the drop-in contract), kept as a table
IMAGE_FLAGS = [
    (('--dataset_type', '-d'), dict(default='mnist', help='dataset to be used, `mnist` or `dsprites`')),
    (('--batch_size',), dict(default=128, help='training batch size')),
    (('--num_epochs',), dict(default=100, help='number of training epochs')),
    (('--lr',), dict(default=1e-4, help='learning rate')),
    (('--beta',), dict(default=4.0, help='parameter for weighting KLD loss')),
    (('--capacity',), dict(default=0.0, help='parameter for beta-VAE capacity')),
    (('--gamma',), dict(default=10.0, help='parameter for weighting regularization loss')),
    (('--delta',), dict(default=1.0, help='parameter for controlling the spread')),
    (('--dec_dist',), dict(default='bernoulli', help='distribution of the decoder')),
    (('--train/--test',), dict(default=True, help='train or test the specified model')),
    (('--log/--no_log',), dict(default=False, help='log the results for tensorboard')),
    (('--rand',), dict(default=None, help='random seed for the random number generator')),
    (('--reg_type', '-r'), dict(default=None, multiple=True, help='attribute name string to be used for regularization')),
]


def with_options(fn):
    for names, kwargs in reversed(IMAGE_FLAGS):
        fn = click.option(*names, **kwargs)(fn)
    return click.command()(fn)


@with_options
def main(dataset_type, batch_size, num_epochs, lr, beta, capacity, gamma, delta, dec_dist, train, log, rand, reg_type):
    if dataset_type == 'mnist':
        dataset, attr_dict = MorphoMnistDataset(), MNIST_REG_TYPES
    elif dataset_type == 'dsprites':
        dataset, attr_dict = DspritesDataset(), DSPRITES_REG_TYPE
    else:
        raise ValueError('Invalid dataset_type. Choose between mnist and dsprites')
    reg_dim = reg_dims_for(reg_type, attr_dict)
    seeds = range(0, 10) if rand is None else [int(rand)]
    # launched by torch.distributed.run: one process per GPU, minibatch rows sharded over the ranks, gradients all-reduced
    # over RCCL (arvae_amd.parallel); --batch_size is then the per-GPU batch
    from arvae_amd.parallel import init_from_env
    dp = init_from_env() if train else None
    chief = dp is None or dp.rank == 0

    def build(seed):
        # the reference builds the model BEFORE the trainer seeds torch (train_image_vae.py:97-109, image_vae_trainer.py:103):
        # its initial weights differ from run to run.  Here the run's seed covers them too (same run, same weights)
        torch.manual_seed(seed)
        model = MnistVAE() if dataset_type == 'mnist' else DspritesVAE()
        trainer = ImageVAETrainer(dataset=dataset, model=model, lr=lr, reg_type=reg_type, reg_dim=reg_dim, beta=beta,
                                  capacity=capacity, gamma=gamma, delta=delta, dec_dist=dec_dist, rand=seed)
        return model, trainer

    # every rank trains every seed first; rank 0 evaluates afterwards, once the process group is gone -- a rank that waits in
    # a pending RCCL collective while rank 0 evaluates would be killed by the group's watchdog timeout (advisor, round 2)
    if train:
        if not torch.cuda.is_available():
            raise SystemExit('training needs a GPU: the AR-VAE path has no CPU fallback')
        try:
            for seed in seeds:
                model, trainer = build(seed)
                trainer.cuda()
                if dp is not None:
                    dp.attach(trainer)
                trainer.train_model(batch_size=batch_size, num_epochs=num_epochs, log=log)
                trainer.data_parallel = None
            if dp is not None:
                dp.finish()
        except BaseException:
            # a rank that fails must not leave its peers waiting in a collective: give the communicator up without waiting for
            # them (ncclCommAbort) and let the error end this process non-zero -- the launcher then ends the other ranks, whose
            # own bounded waits (parallel.LibraryComm.wait_idle) raise in the meantime
            if dp is not None:
                dp.abort()
            raise
    if not chief:
        return
    for seed in seeds:
        model, trainer = build(seed)
        trainer.load_model()
        trainer.writer = None
        eval_bs = min(128, batch_size)                  # the reference evaluates with 128; smaller runs keep their own size
        _, _, eval_loader = dataset.data_loaders(batch_size=eval_bs)
        codes, attrs, names = trainer.compute_representations(eval_loader)
        summary = {'model': repr(model), 'num_codes': int(codes.shape[0]), 'attributes': names,
                   'latent_mean_abs': [float(v) for v in abs(codes).mean(0)]}
        summary.update(trainer.test_model(batch_size=eval_bs))
        print(json.dumps(summary, indent=2))


if __name__ == '__main__':
    main()
