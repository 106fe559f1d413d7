"""derl_amd -- MI355X-native rollout + PPO/A2C update engine behind derl's Python API.

Import surface mirrors ``derl/__init__.py`` for the on-policy path (PPO / A2C); the DQN and
SAC families of the reference are out of scope."""
from . import distributed, env, summary
from .alg import Alg, Loss, Trainer, PPO, PPOLoss, A2C, A2CLoss
from .anneal import AnnealingVariable, LinearAnneal, TorchSched
from .factory import Factory, KwargsDict, PPOFactory, A2CFactory
from .models import NatureCNNBase, NatureCNNModel, make_model, GatheredRows
from .mlp_models import MLP, MuJoCoModel, MLPCategoricalModel
from .policies import Policy, ActorCriticPolicy
from .runners import (EnvRunner, RunnerWrapper, TransformInteractions, IterateWithMinibatches,
                      ppo_runner_wrap, make_ppo_runner, PeriodicSummaries, GAE, MergeTimeBatch,
                      NormalizeAdvantages, Take)
from .scripts import (get_simple_parser, get_defaults_parser, get_parser, log_args,
                      get_args_from_defaults, get_args)
