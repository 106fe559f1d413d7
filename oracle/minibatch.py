"""Epoch / minibatch index order of IterateWithMinibatches (CPU oracle).

Follows derl/runners/onpolicy.py:44-62: before every epoch the WHOLE interaction dict is
shuffled in place by a fresh ``np.random.permutation``, so shuffles compose across
epochs; minibatches are contiguous slices of size ``n // num_minibatches`` (a remainder
gives an extra short minibatch).
"""
import numpy as np


def minibatch_indices(sample_size, num_epochs, num_minibatches, rng=None,
                      shuffle_before_epoch=True):
  """Yields (epoch, start, indices-into-the-ORIGINAL-arrays) in the reference's order.
  ``rng`` must offer ``permutation(n)``; default is the global ``np.random`` stream
  the reference uses (onpolicy.py:47)."""
  rng = np.random if rng is None else rng
  order = np.arange(sample_size)
  for epoch in range(num_epochs):
    if shuffle_before_epoch:
      order = order[rng.permutation(sample_size)]
    mbsize = sample_size // num_minibatches
    for start in range(0, sample_size, mbsize):
      yield epoch, start, order[start:min(start + mbsize, sample_size)]
