#!/usr/bin/env python3
"""Train / evaluate the MeasureVAE on MI355X: counterpart of the reference's train_measure_vae.py (same flags and
defaults).  Only the `folk` one-bar dataset in its pre-built tensor form is supported (arvae_amd.data.FolkNBarDataset):
building it from ABC files with music21, and the `bach` chorales, are offline steps outside this build."""
import json
import os
import sys

import click
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from arvae_amd.data import FolkNBarDataset  # noqa: E402
from arvae_amd.measure_vae import MeasureVAE  # noqa: E402
from arvae_amd.measure_vae_trainer import MUSIC_REG_TYPE, MeasureVAETrainer  # noqa: E402


def reg_dims_for(reg_type, attr_dict):
    if len(reg_type) == 0:
        return (0,)
    if len(reg_type) == 1 and reg_type[0] == 'all':
        return tuple(attr_dict.values())
    return tuple(attr_dict[r] for r in reg_type)


# flag surface of the reference script (names, defaults and help strings are the drop-in contract), kept as a table
MEASURE_FLAGS = [
    (('--dataset_type', '-d'), dict(default='folk', help='dataset to be used, `bach` or `folk`')),
    (('--note_embedding_dim',), dict(default=10, help='size of the note embeddings')),
    (('--metadata_embedding_dim',), dict(default=2, help='size of the metadata embeddings')),
    (('--num_encoder_layers',), dict(default=2, help='number of layers in encoder RNN')),
    (('--encoder_hidden_size',), dict(default=128, help='hidden size of the encoder RNN')),
    (('--encoder_dropout_prob',), dict(default=0.5, help='float, amount of dropout prob between encoder RNN layers')),
    (('--has_metadata',), dict(default=False, help='bool, True if data contains metadata')),
    (('--latent_space_dim',), dict(default=32, help='int, dimension of latent space parameters')),
    (('--num_decoder_layers',), dict(default=2, help='int, number of layers in decoder RNN')),
    (('--decoder_hidden_size',), dict(default=128, help='int, hidden size of the decoder RNN')),
    (('--decoder_dropout_prob',), dict(default=0.5, help='float, amount got dropout prob between decoder RNN layers')),
    (('--batch_size',), dict(default=256, help='training batch size')),
    (('--num_epochs',), dict(default=30, help='number of training epochs')),
    (('--lr',), dict(default=1e-4, help='learning rate')),
    (('--beta',), dict(default=0.001, help='parameter for weighting KLD loss')),
    (('--capacity',), dict(default=0.0, help='parameter for beta-VAE capacity')),
    (('--gamma',), dict(default=1.0, help='parameter for weighting regularization loss')),
    (('--delta',), dict(default=10.0, help='parameter for controlling the spread')),
    (('--train/--test',), dict(default=True, help='train or test the specified model')),
    (('--log/--no_log',), dict(default=False, help='log the results for tensorboard')),
    (('--rand',), dict(default=None, help='random seed for the random number generator')),
    (('--reg_type', '-r'), dict(default=None, multiple=True, help='attribute name string to be used for regularization')),
]


def with_options(fn):
    for names, kwargs in reversed(MEASURE_FLAGS):
        fn = click.option(*names, **kwargs)(fn)
    return click.command()(fn)


@with_options
def main(dataset_type, note_embedding_dim, metadata_embedding_dim, num_encoder_layers, encoder_hidden_size,
         encoder_dropout_prob, latent_space_dim, num_decoder_layers, decoder_hidden_size, decoder_dropout_prob,
         has_metadata, batch_size, num_epochs, lr, beta, capacity, gamma, delta, train, log, rand, reg_type):
    if dataset_type == 'folk':
        dataset = FolkNBarDataset(dataset_type='train', is_short=False, num_bars=1)
    elif dataset_type == 'bach':
        raise SystemExit('the `bach` chorale dataset needs music21 preprocessing, which is outside this build')
    else:
        raise ValueError('Invalid dataset_type. Choose between `folk` and `bach`')
    reg_dim = reg_dims_for(reg_type, MUSIC_REG_TYPE)
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
        model = MeasureVAE(dataset=dataset, note_embedding_dim=note_embedding_dim,
                           metadata_embedding_dim=metadata_embedding_dim, num_encoder_layers=num_encoder_layers,
                           encoder_hidden_size=encoder_hidden_size, encoder_dropout_prob=encoder_dropout_prob,
                           latent_space_dim=latent_space_dim, num_decoder_layers=num_decoder_layers,
                           decoder_hidden_size=decoder_hidden_size, decoder_dropout_prob=decoder_dropout_prob,
                           has_metadata=has_metadata, dataset_type=dataset_type)
        trainer = MeasureVAETrainer(dataset=dataset, model=model, lr=lr, reg_type=reg_type, reg_dim=reg_dim, beta=beta,
                                    capacity=capacity, gamma=gamma, delta=delta, rand=seed)
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
        eval_bs = min(256, batch_size)                  # the reference evaluates with 256; smaller runs keep their own size
        _, _, eval_loader = dataset.data_loaders(batch_size=eval_bs)
        codes, attrs, names = trainer.compute_representations(eval_loader)
        summary = {'model': repr(model), 'num_codes': int(codes.shape[0]), 'attributes': names,
                   'attribute_means': [float(v) for v in attrs.mean(0)]}
        summary.update(trainer.test_model(batch_size=eval_bs))
        print(json.dumps(summary, indent=2))


if __name__ == '__main__':
    main()
