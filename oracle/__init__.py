"""CPU oracle for the derl on-policy hot path (rollout act -> GAE -> PPO/A2C update).

TEST INFRASTRUCTURE ONLY.  Nothing in ``derl_amd/`` (the product) imports this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and only as the checker / the reported CPU baseline -- never as the thing shipped.

It is an independent restatement (NumPy + torch-CPU functional ops, plus a plain-C GAE
in ``gae.c``) of what mknbv/derl computes on this path, written from the reference's
behaviour; every function cites the reference file:line it follows.  Parity status:
PINNED -- ``tests/test_oracle_golden.py`` checks it against (i) golden vectors produced
by importing the unmodified reference in the build container
(``tests/golden/generate.py``) and (ii) the reference's own fixtures that still
reproduce (``testdata/models/dqn-base-outputs.npy``, ``testdata/ppo/pybullet/*``,
``testdata/a2c/atari/*``).
"""
from .gae import gae_advantages, merge_time_batch, normalize_advantages
from .models import (nature_cnn_forward, mlp_forward, mujoco_forward,
                     init_nature_cnn, init_mujoco, NATURE_CNN_KEYS, mujoco_keys)
from .distributions import (categorical_log_prob_entropy, diag_normal_log_prob_entropy,
                            categorical_sample_from_uniform)
from .losses import (ppo_loss_terms, a2c_loss_terms, ppo_head_grads, a2c_head_grads,
                     ppo_loss_and_grads, a2c_loss_and_grads)
from .optim import clip_grad_norm, adam_step, rmsprop_step, linear_anneal
from .minibatch import minibatch_indices
