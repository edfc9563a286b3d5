"""CPU oracle for the AR-VAE training path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (PyTorch-CPU functional code and
numpy float64 closed forms) of the algorithm the reference runs on its training
hot path.  It exists to CHECK the HIP path; it is never the product:

  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import it; nothing under ar-vae_amd/ does, and the product raises if its
    HIP library is missing instead of falling back to anything here;
  * parity is PINNED: tests/test_oracle_golden.py checks every function here
    against tests/golden/*.npz, which tests/golden/make_goldens.py produced by
    importing and running the reference itself (/root/reference) on PyTorch-CPU
    in the build container.  The reference ships no tests or golden vectors of
    its own, so those fixtures are the pin.

The reference's arithmetic lives in PyTorch (third party; environment.yml pins
pytorch=1.0.0, this image has 2.10.0): nn.Conv2d / ConvTranspose2d / Linear /
GRU / Embedding, distributions.Normal + kl_divergence, BCE-with-logits,
CrossEntropyLoss, L1Loss(tanh, sign), optim.Adam.  The restatement below calls
the stateless torch.nn.functional conv/linear primitives on CPU tensors and
writes everything else (GRU cells, KL, losses, Adam, attribute labels) out by
hand; each function cites the reference file:line it follows.
"""
